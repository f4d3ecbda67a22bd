"""Time the raster frame (BASELINE config 3 shape) on the GPU: python tools/raster_time.py [--res 3840x2160] [--frames 20]."""
import argparse, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
A = importlib.import_module("incarnation-renderer_amd.ctypes_api")
sio = importlib.import_module("incarnation-renderer_amd.scene_io")
synth = importlib.import_module("incarnation-renderer_amd.synth")
R = importlib.import_module("incarnation-renderer_amd.raster")
rt = importlib.import_module("incarnation-renderer_amd.runtime")

ap = argparse.ArgumentParser()
ap.add_argument("--res", default="3840x2160"); ap.add_argument("--frames", type=int, default=20); ap.add_argument("--texture-size", type=int, default=256)
ap.add_argument("--hbao", action="store_true"); ap.add_argument("--prims", action="store_true")
ap.add_argument("--host", action="store_true", help="the C++ Renderer::Render mirror with its fused passes (deferred_lighting_kernel, tonemap inside the TAA resolve) and the sky IBL inputs: the frame bench.py times")
a = ap.parse_args()
class Cfg(A.FConfig):
    EnableHBAO = a.hbao
    UseBentNormals = a.hbao
    UseMeshlets = not a.prims
W, H = map(int, a.res.split("x"))
t = time.time(); scene = synth.sponza_stat(texture_size=a.texture_size); print("scene %.1fs" % (time.time() - t))
t = time.time(); scene = R.generate_meshlets(scene); print("meshlets %d in %.1fs" % (scene.num_meshlets, time.time() - t))
ctx = rt.Context(0)
t = time.time(); gs = ctx.create_scene(scene); print("upload %.1fs" % (time.time() - t))
view = sio.FView().reset(scene, aspect=W / H)
if a.host:
    ibl = importlib.import_module("incarnation-renderer_amd.ibl")
    sky = ibl.SkyIbl(ctx, gs); lut = ibl.generate_env_brdf(ctx, 512, 512)
    hr = R.HostRenderer(ctx, gs, W, H, cfg=Cfg, ibl=R.IblInputs.from_sky(sky, lut, (512, 512)))
    for _ in range(3):
        hr.render(view)
    ctx.synchronize()
    t = time.time()
    for _ in range(a.frames):
        hr.render(view)
    ctx.synchronize()
    print("raster frame (C++ renderer, fused passes) %.3f ms (%dx%d, %d meshlets)" % ((time.time() - t) / a.frames * 1e3, W, H, scene.num_meshlets))
    hr.set_profiling(True); hr.render(view); print("passes", [(n, round(t_, 4)) for n, t_ in hr.pass_times()]); hr.close()
    sys.exit(0)
rr = R.RasterRenderer(ctx, gs, W, H, Cfg)
if os.environ.get("INCR_NO_RAY_STATS"):
    rr.surf.ptr["ray_stats_real"], rr.surf.ptr["ray_stats"] = rr.surf.ptr["ray_stats"], None      # passes take NULL = no counting
rr.set_sh9(np.ones((9, 3), np.float32) * 0.1)
for _ in range(3):
    rr.render(view)
ctx.synchronize()
t = time.time()
for _ in range(a.frames):
    rr.render(view)
ctx.synchronize()
dt = (time.time() - t) / a.frames
print("raster frame %.3f ms (%dx%d, %d meshlets)" % (dt * 1e3, W, H, scene.num_meshlets))
print("counts", rr.surf.download("counts"), "stats", rr.surf.download("render_stats"), "shadow rays", rr.surf.download("ray_stats").light_shadow_rays)
import ctypes as C
st = (C.c_uint32 * 4)()
rt.check(ctx.lib.incr_visibility_get_stats(ctx.handle, st))
print("visibility queues: screen triangles %d, band items %d, small %d, whole %d" % tuple(st))
