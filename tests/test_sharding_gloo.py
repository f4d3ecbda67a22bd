"""N > 1 path on CPU: two gloo ranks render interleaved row-block shards (with the oracle standing in for the
kernel), all_gather them and un-interleave; the result must equal the single-process frame bit for bit."""
import os
import subprocess
import sys

import numpy as np

from conftest import ROOT

WORKER = r'''
import importlib, os, sys
import numpy as np
import torch
import torch.distributed as dist
ROOT = sys.argv[1]; out_path = sys.argv[2]
sys.path.insert(0, ROOT)
PKG = "incarnation-renderer_amd"
A = importlib.import_module(PKG + ".ctypes_api"); sio = importlib.import_module(PKG + ".scene_io"); rt = importlib.import_module(PKG + ".runtime")
from oracle import binding as ob
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
W, H, h = 80, 52, 8                       # 52 rows = 6.5 blocks: exercises the ragged last block + padding rows
s = sio.load_gltf(os.path.join(ROOT, "assets/damaged-helmet/DamagedHelmet.gltf"), max_texture_size=256)
view = sio.FView().reset(s, aspect=W / H)
osc = ob.OracleScene(s)
shard = A.Shard(h, rank, world)
d = rt.make_pathtrace_desc(view, W, H, 1, shard)
r = osc.pathtrace(d)
hist = r["history"]
d.current_sample_index = 2
r = osc.pathtrace(d, history=hist)
local = torch.from_numpy(hist.view(np.int32).copy())
gathered = [torch.zeros_like(local) for _ in range(world)]
dist.all_gather(gathered, local)
rays = torch.tensor([r["stats"].total_rays()], dtype=torch.int64)
dist.all_reduce(rays)
if rank == 0:
    g = np.stack([t.numpy() for t in gathered]).view(np.uint32)      # [world][local_rows][W]
    full = np.zeros((H, W), dtype=np.uint32)
    for y in range(H):
        blk, rr = divmod(y, h)
        full[y] = g[blk % world, (blk // world) * h + rr]
    np.savez(out_path, full=full, rays=rays.numpy())
dist.destroy_process_group()
'''


def test_two_rank_shards_equal_single_frame(tmp_path, ob, sio, rt):
    out = str(tmp_path / "gathered.npz")
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29617", str(script), ROOT, out]
    subprocess.run(cmd, check=True, env=env, timeout=600, cwd=ROOT)
    g = np.load(out)
    W, H = 80, 52
    s = sio.load_gltf(os.path.join(ROOT, "assets/damaged-helmet/DamagedHelmet.gltf"), max_texture_size=256)
    view = sio.FView().reset(s, aspect=W / H)
    osc = ob.OracleScene(s)
    d = rt.make_pathtrace_desc(view, W, H, 1)
    r = osc.pathtrace(d)
    hist = r["history"]
    d.current_sample_index = 2
    r = osc.pathtrace(d, history=hist)
    np.testing.assert_array_equal(g["full"], hist)
    assert int(g["rays"][0]) == r["stats"].total_rays()
