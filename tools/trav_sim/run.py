"""Developer tool: builds tools/trav_sim/trav_sim.cpp against csrc/scene_build.cpp and runs the CPU traversal model
on Sponza-stat (or --bistro).  No GPU needed.  usage: python tools/trav_sim/run.py [--bistro] [--w 240 --h 135]"""
import argparse, ctypes as C, importlib, math, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("incarnation-renderer_amd")
synth = importlib.import_module("incarnation-renderer_amd.synth")
sio = importlib.import_module("incarnation-renderer_amd.scene_io")
A = importlib.import_module("incarnation-renderer_amd.ctypes_api")
HERE = os.path.dirname(os.path.abspath(__file__))
ap = argparse.ArgumentParser(); ap.add_argument("--bistro", action="store_true"); ap.add_argument("--w", type=int, default=240); ap.add_argument("--h", type=int, default=135)
ap.add_argument("--extra", default="")
a = ap.parse_args()
so = os.path.join(HERE, "_trav_sim.so")
csrc = os.path.join(ROOT, "incarnation-renderer_amd", "csrc")
subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off"] + a.extra.split() + ["-o", so, os.path.join(HERE, "trav_sim.cpp"), os.path.join(csrc, "scene_build.cpp")], check=True)
L = C.CDLL(so)
s = synth.bistro_scale(texture_size=8) if a.bistro else synth.sponza_stat(texture_size=8)
v = sio.FView().reset(s, aspect=16.0 / 9.0)
desc = s.to_desc()
cam = np.concatenate([v.position, v.right, v.up, v.look]).astype(np.float32)
L.trav_sim.argtypes = [C.POINTER(A.SceneDesc), C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_int]
sys.stdout.flush()
L.trav_sim(C.byref(desc), cam.ctypes.data, math.tan(float(v.fov) / 2), 16.0 / 9.0, a.w, a.h)
