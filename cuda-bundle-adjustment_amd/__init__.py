"""MI355X-native bundle adjustment behind the cuba::CudaBundleAdjustment API.

Layout:  csrc/   hand-written HIP kernels + the C ABI (libcuba_hip.so, include/cuba_hip.h)
         host/   C++ host layer mirroring cuba::CudaBundleAdjustment (include/cuda_bundle_adjustment.h)
         capi.py ctypes binding of the C ABI (used by tests / bench.py; torch is only plumbing)
         graph.py, synth.py  graph containers, flattening, synthetic KITTI-shaped graphs
Importing this package does not import torch and does not touch the GPU.
"""
__version__ = "0.1.0"
