"""Generates the committed regression fixtures under tests/golden/ from the CPU oracle.

The reference has no tests and cannot run here, so these vectors pin the ORACLE against
accidental change; they are not outputs of the reference.  Run: python tools/make_golden.py"""
import ctypes as C
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "incarnation-renderer_amd"
A = importlib.import_module(PKG + ".ctypes_api")
sio = importlib.import_module(PKG + ".scene_io")
rt = importlib.import_module(PKG + ".runtime")
from oracle import binding as ob

G = os.path.join(ROOT, "tests", "golden")
os.makedirs(G, exist_ok=True)
L = ob.lib()

# 1. sampler tables
out = (C.c_float * 2)()
rows = []
for num_pixels, pixels in ((512 * 512, (0, 1, 511, 512 * 256 + 100, 512 * 512 - 1)), (3840 * 2160, (0, 3839, 4147200, 8294399))):
    for pixel in pixels:
        for sample in (1, 2, 3, 8, 255):
            for st in (0, 1, 2, 4):
                L.orc_sample_point(pixel, sample, st, 16, num_pixels, out)
                rows.append(dict(pixel=pixel, sample=sample, set=st, num_pixels=num_pixels,
                                 x_bits=int(np.float32(out[0]).view(np.uint32)), y_bits=int(np.float32(out[1]).view(np.uint32))))
perm = [dict(i=i, l=l, p=p, out=int(L.orc_cmj_permute(i, l, p))) for l in (16, 256) for p in (0, 0x51633e2d, 0x9e3779b9, 0xffffffff) for i in (0, 1, 7, l - 1)]
with open(os.path.join(G, "sampler.json"), "w") as f:
    json.dump(dict(sample_point=rows, permute=perm), f)

# 2. helmet 64x64: camera rays, primary hits, 1-spp target, 4-sample accumulation
s = sio.load_gltf(os.path.join(ROOT, "assets/damaged-helmet/DamagedHelmet.gltf"))
osc = ob.OracleScene(s)
W = H = 64
view = sio.FView().reset(s, aspect=1.0)
d = rt.make_pathtrace_desc(view, W, H, 1)
ys, xs = np.mgrid[0:H, 0:W]
o, dr = osc.camera_rays(d, xs.ravel(), ys.ravel())
ids, tuv = osc.trace_rays(o, dr, 0.001, 10000.0)
r = osc.pathtrace(d, num_threads=1)
target1 = r["target"].copy()
hist = r["history"]
stats = [[r["stats"].camera_rays, r["stats"].bounce_rays, r["stats"].sky_shadow_rays, r["stats"].light_shadow_rays, r["stats"].shaded_hits]]
for n in (2, 3, 4):
    d.current_sample_index = n
    r = osc.pathtrace(d, history=hist, num_threads=1)
    stats.append([r["stats"].camera_rays, r["stats"].bounce_rays, r["stats"].sky_shadow_rays, r["stats"].light_shadow_rays, r["stats"].shaded_hits])
np.savez_compressed(os.path.join(G, "helmet_64.npz"), origins=o, dirs=dr, ids=ids, tuv=tuv, target1=target1, history4=hist,
                    stats=np.array(stats, dtype=np.int64), view=np.asarray(view.view_transform), proj=np.asarray(view.projection_transform))
# 2b. MeshUtils::Meshletize of the helmet primitive from the oracle restatement (oracle/oracle_meshlet.cpp): meshlet ranges,
#     bounding spheres, vertex lists and packed triangles -- the ids the visibility buffer encodes
p0 = s.mesh_primitives[0][0]
midx = s.accessor_array(p0["index"], np.uint16, 1, p0["index_count"]).astype(np.uint32).ravel()
mpos = np.ascontiguousarray(s.accessor_array(p0["position"], np.float32, 3, p0["vertex_count"]))
mm = ob.meshletize(midx, mpos, 64, 126)
np.savez_compressed(os.path.join(G, "helmet_meshlets.npz"), **mm)
print("golden written:", os.listdir(G))

# 3. scalar known-answer tables: float results stored as uint32 bit patterns
def bits(v):
    return int(np.float32(v).view(np.uint32))


kat = {}
f3 = lambda *v: (C.c_float * 3)(*v)
rng = np.random.default_rng(0x6B41)
rows = []
for NoH, r in ((1.0, 0.5), (0.9, 0.1), (0.5, 0.9), (0.25, 0.35), (0.001, 1.0)):
    rows.append(dict(NoH=NoH, r=r, ggx=bits(L.orc_ggx(NoH, r)), pdf=bits(L.orc_ggx_pdf(NoH, r))))
kat["ggx"] = rows
kat["g_smith"] = [dict(NoV=a, NoL=b, r=c, direct=bits(L.orc_g_smith_direct(a, b, c)), correlated=bits(L.orc_g_smith_ggx_correlated(a, b, c)))
                  for a, b, c in ((1.0, 1.0, 0.5), (0.3, 0.8, 0.2), (0.05, 0.5, 0.9), (0.7, 0.001, 0.04))]
rows = []
for u, f0 in ((1.0, (0.04, 0.04, 0.04)), (0.5, (0.04, 0.04, 0.04)), (0.0, (0.04, 0.04, 0.04)), (0.2, (1.0, 0.71, 0.29))):
    o3 = f3(0, 0, 0); L.orc_f_schlick(u, f3(*f0), o3); rows.append(dict(u=u, f0=list(f0), out=[bits(x) for x in o3]))
kat["f_schlick"] = rows
kat["exposure"] = [dict(ev100=e, out=bits(L.orc_exposure(e))) for e in (0, 9, 13, 15)]
rows = []
for x in ((0.0, 0.18, 1.0), (2.5, 10.0, 1e4), (0.01, 0.5, 0.75)):
    o3 = f3(0, 0, 0); L.orc_aces_film(f3(*x), o3); rows.append(dict(x=list(x), out=[bits(v) for v in o3]))
kat["aces_film"] = rows
sun = sio.sun_direction_from_time_of_day()            # ToD defaults 11.0 h / day 200 / lat 42.5 (inc/common.h:43-46)
rows = []
for e in ((0, 1, 0), (1, 0.2, 0), (-0.5, 0.5, 0.7), (0.3, 0.05, -0.9), tuple(float(v) for v in sun), (0, -0.2, 1)):
    en = np.array(e, np.float64); en = (en / np.linalg.norm(en)).astype(np.float32)
    o3 = f3(0, 0, 0); L.orc_sky_radiance(f3(*sun), f3(*en), 2.0, o3)
    rows.append(dict(dir=[bits(v) for v in en], out=[bits(v) for v in o3]))
kat["preetham"] = dict(sun=[bits(v) for v in sun], turbidity=2.0, samples=rows)
rows = []
for _ in range(24):
    n = rng.normal(size=3); n = (n / np.linalg.norm(n)).astype(np.float32)
    enc, dec = (C.c_float * 2)(), f3(0, 0, 0)
    L.orc_oct_roundtrip(f3(*n), enc, dec)
    rows.append(dict(n=[bits(v) for v in n], enc=[bits(v) for v in enc], dec=[bits(v) for v in dec]))
kat["oct"] = rows
kat["visibility_encoding"] = dict(clear=0xFFFFF000, meshlet=[dict(id=i, tri=t, value=(i << 7) | t) for i, t in ((0, 0), (1, 5), (3042, 125), (0xFFFFFF, 127))],
                                 primitive=[dict(id=i, tri=t, value=(i << 20) | t) for i, t in ((0, 0), (24, 99999), (2046, 0xFFFFF))])
# cluster grid 16x9x24, near 1, extent 200: slice depths, the pixel->slice scale / bias, frustum planes of two clusters
import math
sp = synth = importlib.import_module(PKG + ".synth")
R = importlib.import_module(PKG + ".raster")
scene3 = synth.sponza_stat(texture_size=8, tri_scale=1 / 64.0)
view3 = sio.FView().reset(scene3, aspect=16 / 9)
vc3 = R.make_view_constants(view3, 1920, 1080)
grid = (C.c_uint32 * 3)(16, 9, 24)
planes = {}
for ci in ((0, 0, 0), (15, 8, 23), (7, 4, 11)):
    out24 = (C.c_float * 24)()
    L.orc_cluster_frustum((C.c_uint32 * 3)(*ci), grid, 1.0, 200.0, C.byref(vc3.lightcull_proj_transform), C.byref(vc3.lightcull_inv_view_proj), out24)
    planes["%d,%d,%d" % ci] = [bits(v) for v in out24]
scale = np.float32(24.0) / np.float32(math.log(200.0 / 1.0)); bias = -scale * np.float32(math.log(1.0))
kat["clusters"] = dict(grid=[16, 9, 24], near=1.0, extent=200.0, slice_depths=[bits(np.float32(1.0) * np.float32(np.float32(200.0) ** np.float32(k / 24.0))) for k in (0, 1, 12, 24)],
                       scale=bits(scale), bias=bits(bias), planes=planes)
with open(os.path.join(G, "scalar_kats.json"), "w") as f:
    json.dump(kat, f)

# 4. raster crop (item 9): helmet 64x64 frame without TAA -- visibility ids, depth, G-buffers, colour target
class NoTaa(A.FConfig):
    EnableTAA = False
s = R.generate_meshlets(s)
osc = ob.OracleScene(s)                                # a new oracle scene: the first one was packed before the meshlets existed
hs = ob.HostSurfaces(W, H, s.num_meshlets, 16 * 9 * 24, 64)
hs.arr["sh9"][:, :3] = 0.25
vcr = R.make_view_constants(view, W, H)
for name, dsc in R.build_descs(None, s, vcr, hs.ptr, 0, NoTaa):
    ob.run_pass(osc, name, dsc)
np.savez_compressed(os.path.join(G, "helmet_raster_64.npz"), vis=hs.arr["vis"], depth=hs.arr["depth"], g0=hs.arr["gbuffer_basecolor"],
                    g1=hs.arr["gbuffer_normals"], g2=hs.arr["gbuffer_mra"], color=hs.arr["color"], counts=hs.arr["counts"])

# ---- sky / IBL precompute at 16^2 (FScene::UpdateDynamicSky): lat-long chain, source cube, GGX-prefiltered cube (its taps cross face
#      edges: seamless cube filtering), SH9
sky = ob.update_dynamic_sky((0.35, 0.8, 0.48), 2.0, resolution=16)
np.savez_compressed(os.path.join(G, "sky_ibl_16.npz"), latlong=sky["latlong"], source_cube=sky["source_cube"], envmap=sky["envmap"], sh9=sky["sh9"])
print("golden written:", sorted(os.listdir(G)))
