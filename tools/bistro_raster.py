import importlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
PKG="incarnation-renderer_amd"
A = importlib.import_module(PKG + ".ctypes_api"); sio = importlib.import_module(PKG + ".scene_io"); syn = importlib.import_module(PKG + ".synth")
R = importlib.import_module(PKG + ".raster"); rt = importlib.import_module(PKG + ".runtime")
t=time.time(); s = syn.bistro_scale(); print("gen %.1fs"%(time.time()-t))
t=time.time(); s = R.generate_meshlets(s); print("meshlets %d %.1fs"%(s.num_meshlets, time.time()-t))
ctx = rt.Context(0); t=time.time(); gs = ctx.create_scene(s); print("create %.1fs"%(time.time()-t))
W,H=3840,2160
view = sio.FView().reset(s, aspect=W/H)
hr = R.HostRenderer(ctx, gs, W, H, sh9_rgb=np.full((9,3),0.1,np.float32))
for _ in range(3): hr.render(view)
ctx.synchronize(); t=time.time()
for _ in range(10): hr.render(view)
ctx.synchronize(); print("bistro raster %.3f ms/frame" % ((time.time()-t)/10*1e3))
import ctypes as C
st=(C.c_uint32*4)(); rt.check(ctx.lib.incr_visibility_get_stats(ctx.handle, st)); print("queues", list(st))
vis = hr.download(A.SURFACE_VISIBILITY); print("covered", (vis!=0xFFFFF000).mean(), "counts", ctx.download(hr.surface_ptr(A.SURFACE_BATCH_COUNTS),(2,),np.uint32))
