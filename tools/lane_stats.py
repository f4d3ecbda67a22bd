"""Developer tool (GPU box): rebuilds the path-tracing kernels with -DINCR_WF_LANE_STATS, renders Sponza-stat frames and prints
where the lanes of the persistent traversal waves go (node steps / triangle steps / idle, steady state against the drain after
the cursor ran dry).  "extend" covers bounces 1-3: the camera rays go through the packet kernel, which has no idle lanes to count.  usage: python tools/lane_stats.py [--res 3840x2160] [--frames 2]   (restores the normal build afterwards)"""
import argparse, ctypes as C, importlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser(); ap.add_argument("--res", default="3840x2160"); ap.add_argument("--frames", type=int, default=2)
ap.add_argument("--extra", default="")
a = ap.parse_args()
csrc = os.path.join(ROOT, "incarnation-renderer_amd", "csrc")
def build(extra):
    os.remove(os.path.join(csrc, "pathtrace_wf.o")) if os.path.exists(os.path.join(csrc, "pathtrace_wf.o")) else None
    subprocess.run(["make", "-s", "-j16", "EXTRA=" + extra], cwd=csrc, check=True, stdout=subprocess.DEVNULL)
build("-DINCR_WF_LANE_STATS " + a.extra)
try:
    pkg = importlib.import_module("incarnation-renderer_amd")
    synth = importlib.import_module("incarnation-renderer_amd.synth")
    sio = importlib.import_module("incarnation-renderer_amd.scene_io")
    rt = importlib.import_module("incarnation-renderer_amd.runtime")
    W, H = map(int, a.res.split("x"))
    s = synth.sponza_stat()
    ctx = rt.Context(0)
    scene = rt.Scene(ctx, s)
    view = sio.FView().reset(s, aspect=W / H)
    r = rt.PathTracer(ctx, scene, W, H)
    lib = ctx.lib
    out = (C.c_ulonglong * 16)()
    for f in range(a.frames):
        r.render(view, sample_index=f + 1)
    ctx.synchronize()
    lib.incr_debug_lane_stats.argtypes = [C.POINTER(C.c_ulonglong)]
    lib.incr_debug_lane_stats(out)
    for k, name in enumerate(("extend", "shadow")):
        it, live, node, tri, dit, dlive, refills, pure = [out[k * 8 + i] / a.frames for i in range(8)]
        print("%s: wave-iterations %.3gM  live lane-iterations %.4gM  live lanes/iter %.1f (taking a node step %.1f, a triangle step %.1f)  single-kind iterations %.1f %%" % (name, it / 1e6, live / 1e6, live / it, node / it, tri / it, 100 * pure / it))
        print("   drain (cursor dry): %.1f %% of iterations at %.1f live lanes; steady state %.1f live lanes; refills %.3gM" % (100 * dit / it, dlive / max(dit, 1), (live - dlive) / max(it - dit, 1), refills / 1e6))
finally:
    build(a.extra)
