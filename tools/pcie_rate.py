"""PCIe-inclusive rate of the path-trace step (task statement 4: reported in DESIGN.md, never as the bench value): one 1-spp
4K frame + tonemap, with the 33 MB LDR back buffer (or the 33 MB R11G11B10F history) copied to host memory every frame.
python tools/pcie_rate.py [--frames 8]"""
import argparse, ctypes as C, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
PKG = "incarnation-renderer_amd"
A = importlib.import_module(PKG + ".ctypes_api"); sio = importlib.import_module(PKG + ".scene_io")
synth = importlib.import_module(PKG + ".synth"); rt = importlib.import_module(PKG + ".runtime")

ap = argparse.ArgumentParser(); ap.add_argument("--frames", type=int, default=8); ap.add_argument("--texture-size", type=int, default=1024)
a = ap.parse_args()
W, H = 3840, 2160
scene = synth.sponza_stat(texture_size=a.texture_size)
ctx = rt.Context(0); gs = ctx.create_scene(scene)
view = sio.FView().reset(scene, aspect=W / H)
pt = rt.PathTracer(ctx, gs, W, H)
ldr = ctx.buffer(W * H * 4)
host_pageable = np.zeros((H, W), np.uint32)
host_pinned = torch.zeros((H, W), dtype=torch.int32).pin_memory()


def frame(sample, download):
    pt.render(view, sample)
    t = A.TonemapDesc(); t.hdr_r11g11b10, t.ldr_rgba8, t.res_x, t.rows, t.exposure, t.viewmode = pt.history, ldr, W, H, 13.0, 0
    ctx.tonemap(t, None)
    if download == "pageable":
        rt.check(ctx.lib.incr_buffer_download(ctx.handle, host_pageable.ctypes.data, ldr, W * H * 4, None))
    elif download == "pinned":
        rt.check(ctx.lib.incr_buffer_download(ctx.handle, host_pinned.data_ptr(), ldr, W * H * 4, None))


for mode in (None, "pageable", "pinned"):
    for s in range(2):
        frame(1 + s, mode)
    ctx.synchronize()
    t0 = time.perf_counter()
    for s in range(a.frames):
        frame(3 + s, mode)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / a.frames
    rays = 78.26e6
    print("download=%-9s %.2f ms/frame  %.0f Mray/s" % (mode, dt * 1e3, rays / dt / 1e6))
