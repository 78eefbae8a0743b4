"""chronoclust_amd — MI355X (gfx950) implementation of ChronoClust's per-timestep hot path.

Same Python face as the reference package (`app.run`, `clustering.hddstream.HDDStream`,
`tracking.cluster_tracker.*`); the arithmetic runs in hand-written HIP kernels reached through the C-ABI of
include/chronoclust_hip.h via ctypes (chronoclust_amd/_lib.py).  There is no CPU fallback.
"""
__version__ = "0.1.0"
