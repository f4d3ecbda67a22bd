"""Dev tool: BVH traversal statistics (nodes visited / triangles tested per closest-hit ray).
Needs a library built with -DINCR_TRAV_STATS:  make -C incarnation-renderer_amd/csrc clean && make -C ... EXTRA=-DINCR_TRAV_STATS"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = "incarnation-renderer_amd"
sio = importlib.import_module(pkg + ".scene_io"); syn = importlib.import_module(pkg + ".synth"); rt = importlib.import_module(pkg + ".runtime")
s = syn.sponza_stat(texture_size=64)
ctx = rt.Context(0); sc = ctx.create_scene(s)
i = sc.info(); print("tris", i.num_triangles, "nodes", i.num_bvh_nodes, "depth", i.bvh_max_depth)
rng = np.random.default_rng(1)
n = 400000
# incoherent rays from inside the atrium
o = np.stack([rng.uniform(-12, 11, n), rng.uniform(0.5, 9, n), rng.uniform(-7, 7, n)], 1).astype(np.float32)
d = rng.standard_normal((n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
ids, tuv = sc.trace_rays(o, d, 0.001, 10000.0)
print("incoherent closest: hit %.3f  nodes/ray %.1f  tris/ray %.1f  (max nodes %d, max tris %d)" % (ids[:, 0].mean(), ids[:, 1].mean(), ids[:, 2].mean(), ids[:, 1].max(), ids[:, 2].max()))
v = sio.FView().reset(s, aspect=16 / 9)
d2 = np.tile(v.look, (n, 1)).astype(np.float32) + rng.uniform(-0.4, 0.4, (n, 3)).astype(np.float32); d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
ids, tuv = sc.trace_rays(np.tile(v.position, (n, 1)).astype(np.float32), d2, 0.001, 10000.0)
print("camera-like closest: hit %.3f  nodes/ray %.1f  tris/ray %.1f" % (ids[:, 0].mean(), ids[:, 1].mean(), ids[:, 2].mean()))
