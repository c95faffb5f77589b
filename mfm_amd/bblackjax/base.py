"""``bblackjax/base.py:76-103``: a sampling algorithm is a pair of functions."""
from typing import Callable, NamedTuple


class SamplingAlgorithm(NamedTuple):
    init: Callable
    step: Callable
