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


RASTER_WORKER = r'''
import importlib, os, sys
import numpy as np
import torch
import torch.distributed as dist
ROOT = sys.argv[1]; out_path = sys.argv[2]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
PKG = "incarnation-renderer_amd"
A = importlib.import_module(PKG + ".ctypes_api"); sio = importlib.import_module(PKG + ".scene_io"); R = importlib.import_module(PKG + ".raster")
syn = importlib.import_module(PKG + ".synth")
from oracle import binding as ob
from test_raster_host_and_oracle import DEFAULT_SH9
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
W, H = 96, 54                              # 54 rows over 2 ranks = 27 each; with 4 ranks the last band would be ragged
s = R.generate_meshlets(syn.sponza_stat(texture_size=32, tri_scale=1.0 / 32.0))
view = sio.FView().reset(s, aspect=W / H)
osc = ob.OracleScene(s)
rows = (H + world - 1) // world
band = (rank * rows, min((rank + 1) * rows, H))
hs = ob.HostSurfaces(W, H, s.num_meshlets, 16 * 9 * 24, 64)
hs.arr["sh9"][:, :3] = DEFAULT_SH9
pad = np.zeros((rows * world, W), dtype=np.uint32)             # TAA surfaces padded to equal bands
prev = None
for frame in range(3):
    vc = R.make_view_constants(view, W, H, R.pixel_jitter(frame, W, H), prev)
    for name, d in R.build_descs(None, s, vc, hs.ptr, frame, band=band):
        ob.run_pass(osc, name, d)
    # the one exchange step: all_gather the TAA bands into the full accumulation surface of every rank
    key = "taa%d" % (frame & 1)
    pad[:H] = hs.arr[key]
    local = torch.from_numpy(pad[rank * rows:(rank + 1) * rows].view(np.int32).copy())
    out = torch.zeros((world * rows, W), dtype=torch.int32)
    dist.all_gather_into_tensor(out, local)
    hs.arr[key][:] = out.numpy().view(np.uint32)[:H]
    prev = R.matrix_to_numpy(vc.unjittered_view_proj)
if rank == 0:
    np.savez(out_path, taa=hs.arr[key], vis_band=hs.arr["vis"][band[0]:band[1]], band=np.array(band))
dist.destroy_process_group()
'''


def test_two_rank_raster_row_bands_equal_single_frame(tmp_path, ob, sio, R, A, synth):
    """Sort-first raster sharding with the oracle standing in for the kernels: each rank draws and shades
    its row band (+ one halo row), resolves TAA for the band, and an all_gather of the bands rebuilds the history that the
    next frame reprojects from.  Three frames later the gathered history must equal the single-process one bit for bit."""
    from test_raster_host_and_oracle import run_oracle_frame
    out = str(tmp_path / "raster.npz")
    script = tmp_path / "raster_worker.py"
    script.write_text(RASTER_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29631", str(script), ROOT, out]
    subprocess.run(cmd, check=True, env=env, timeout=600, cwd=ROOT)
    g = np.load(out)
    W, H = 96, 54
    s = R.generate_meshlets(synth.sponza_stat(texture_size=32, tri_scale=1.0 / 32.0))
    view = sio.FView().reset(s, aspect=W / H)
    osc = ob.OracleScene(s)
    prev, hs = None, None
    for frame in range(3):
        vc, hs = run_oracle_frame(ob, R, A, s, osc, view, W, H, frame_index=frame, prev=prev, hs=hs)
        prev = R.matrix_to_numpy(vc.unjittered_view_proj)
    np.testing.assert_array_equal(g["taa"], hs.arr["taa0"])
    b0, b1 = g["band"]
    np.testing.assert_array_equal(g["vis_band"], hs.arr["vis"][b0:b1])
    osc.close()


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
