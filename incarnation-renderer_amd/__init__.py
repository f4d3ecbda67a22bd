"""incarnation-renderer_amd: MI355X-native render hot path of Incarnation-Renderer.

The product is the C-ABI shared library `csrc/libincr_hip.so` (hand-written HIP kernels for gfx950 +
host-side scene/BVH code, declared in include/incr.h).  This package is the thin Python plumbing
around it used by tests and bench.py: ctypes struct mirrors (ctypes_api), scene packing (scene_io,
synth) and the library loader (runtime).  There is no CPU fallback: `runtime.load()` raises if the
HIP library has not been built, and every launch entry point fails without a GPU.
"""
