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
print("golden written:", os.listdir(G))
