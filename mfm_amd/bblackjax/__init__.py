"""Kernel API of the reference's vendored ``bblackjax`` for the MFM hot path (MALA only)."""
from .mcmc.mala import mala  # noqa: F401
