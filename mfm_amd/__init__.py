"""mfm_amd -- MI355X-native Markovian Flow Matching inner loop (drop-in for the hot path of albcab/mfm).

Host side: Python mirrors of the reference's entry points (``multi_modal.py``, ``exe_flow_matching.py``) and of the
``bblackjax.mcmc.mala`` kernel API, calling a C-ABI shared library (``include/mfm.h``) of hand-written HIP kernels
for gfx950 through ctypes.  There is NO CPU fallback: importing the device layer without the built library, or
creating a context without a GPU, raises.
"""
__version__ = "0.1.0"
