"""Times incr_prefilter_normal_roughness on a 2048 x 2048 pair (device-resident inputs; the call blocks, so wall time = GPU time
+ one hipMalloc/hipFree of the pyramid).  Usage: python tools/vmf_time.py [size] [reps]"""
import ctypes as C
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: F401  (must precede libincr_hip.so)

rt = importlib.import_module("incarnation-renderer_amd.runtime")
A = importlib.import_module("incarnation-renderer_amd.ctypes_api")

size = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ctx = rt.Context(0)
lib = rt.load()
rng = np.random.default_rng(0)
n = rng.integers(0, 256, (size, size, 4), dtype=np.uint8); m = rng.integers(0, 256, (size, size, 4), dtype=np.uint8)
mips = lib.incr_compressed_mip_count(size, size); texels = lib.incr_mip_chain_texels(size, size, mips)
b = [ctx.buffer(size * size * 4), ctx.buffer(size * size * 4), ctx.buffer(texels * 4), ctx.buffer(texels * 4)]
ctx.upload(b[0], n); ctx.upload(b[1], m)
d = A.PrefilterNormalRoughnessDesc(normalmap=b[0], metallic_roughness=b[1], width=size, height=size, num_mips=mips, out_normalmap=b[2], out_metallic_roughness=b[3])
for _ in range(3):
    rt.check(lib.incr_prefilter_normal_roughness(ctx.handle, C.byref(d), None))
t0 = time.perf_counter()
for _ in range(reps):
    rt.check(lib.incr_prefilter_normal_roughness(ctx.handle, C.byref(d), None))
dt = (time.perf_counter() - t0) / reps
alg = size * size * 8 + texels * 8
print("vmf prefilter %dx%d, %d levels: %.3f ms per pair (wall, incl. pyramid malloc/free); algorithmic %.1f MB -> %.0f GB/s" % (size, size, mips, dt * 1e3, alg / 1e6, alg / dt / 1e9))
