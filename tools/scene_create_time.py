"""Developer tool (GPU box): wall time of incr_scene_create for the bench scene (texture chains + vMF prefilter + tiling + interleave +
host BVH build + uploads), and of the exact-order vMF prefilter of one pair.  usage: python tools/scene_create_time.py [sponza|bistro] [texture_size]"""
import ctypes as C, importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa
rt = importlib.import_module("incarnation-renderer_amd.runtime"); A = importlib.import_module("incarnation-renderer_amd.ctypes_api")
syn = importlib.import_module("incarnation-renderer_amd.synth")
name = sys.argv[1] if len(sys.argv) > 1 else "sponza"; ts = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
packed = syn.bistro_scale() if name == "bistro" else syn.sponza_stat(texture_size=ts)
ctx = rt.Context(0); lib = rt.load()
for rep in range(2):
    t0 = time.perf_counter(); sc = ctx.create_scene(packed); ctx.synchronize(); dt = time.perf_counter() - t0
    info = sc.info()
    print("%s: incr_scene_create %.3f s (%d triangles, %d nodes, depth %d, %.0f MB textures)" % (name, dt, info.num_triangles, info.num_bvh_nodes, info.bvh_max_depth, info.texture_bytes / 1e6))
    sc.close()
size = ts
rng = np.random.default_rng(0)
n = rng.integers(0, 256, (size, size, 4), dtype=np.uint8); m = rng.integers(0, 256, (size, size, 4), dtype=np.uint8)
mips = lib.incr_compressed_mip_count(size, size); texels = lib.incr_mip_chain_texels(size, size, mips)
b = [ctx.buffer(size * size * 4), ctx.buffer(size * size * 4), ctx.buffer(texels * 4), ctx.buffer(texels * 4)]
ctx.upload(b[0], n); ctx.upload(b[1], m)
d = A.PrefilterNormalRoughnessDesc(normalmap=b[0], metallic_roughness=b[1], width=size, height=size, num_mips=mips, out_normalmap=b[2], out_metallic_roughness=b[3])
d.reserved = 1
for _ in range(2):
    rt.check(lib.incr_prefilter_normal_roughness(ctx.handle, C.byref(d), None)); ctx.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    rt.check(lib.incr_prefilter_normal_roughness(ctx.handle, C.byref(d), None))
ctx.synchronize()
print("exact-order vMF prefilter %dx%d, %d levels: %.3f ms per pair" % (size, size, mips, (time.perf_counter() - t0) / 5 * 1e3))
