from . import adaptive_tempered, tempered  # noqa: F401  (bblackjax/smc/__init__.py:1-3)

__all__ = ["adaptive_tempered", "tempered"]
