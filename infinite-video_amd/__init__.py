"""MI355X-native continuous-time long-term-memory (LTM) consolidation for infinity-Video.

Host side (Python on PyTorch-ROCm) of the one hot path this repository accelerates; the
arithmetic runs in hand-written HIP kernels behind the C ABI of ``include/infv_ltm.h``
(``csrc/``, built to ``libinfv_ltm.so``).  There is no CPU fallback: every operator raises
if the library cannot be loaded.
"""
__version__ = "0.1.0"
