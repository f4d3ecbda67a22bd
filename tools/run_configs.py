"""Run the five BASELINE.json configurations as far as one MI355X allows and print one JSON object
(results table of BASELINE.md).  python tools/run_configs.py > gpurun_out/configs.json"""
import ctypes as C
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402  (its HIP runtime must be loaded before libincr_hip.so)
PKG = "incarnation-renderer_amd"
A = importlib.import_module(PKG + ".ctypes_api")
sio = importlib.import_module(PKG + ".scene_io")
syn = importlib.import_module(PKG + ".synth")
R = importlib.import_module(PKG + ".raster")
rt = importlib.import_module(PKG + ".runtime")
from oracle import binding as ob  # noqa: E402


def psnr(a, b, peak):
    fin = np.isfinite(a) & np.isfinite(b)
    mse = float(np.mean((np.clip(a[fin], 0, peak) - np.clip(b[fin], 0, peak)) ** 2))
    return float("inf") if mse == 0 else 10.0 * np.log10(peak * peak / mse)


def time_pathtrace(ctx, scene, view, W, H, spp, warm=2, in_flight=1):
    """spp samples, `in_flight` of them per incr_pathtrace call (incr_pathtrace_desc::num_samples: same history, fewer launch tails)"""
    pt = rt.PathTracer(ctx, scene, W, H)
    for n in range(1, warm + 1):
        pt.render(view, n, num_samples=in_flight)
    ctx.synchronize()
    ctx.clear(pt.stats, C.sizeof(A.RayStats))
    ctx.clear(pt.history, pt.rows * W * 4)
    t0 = time.perf_counter()
    n = 1
    while n <= spp:
        S = min(in_flight, spp + 1 - n)
        pt.render(view, n, num_samples=S); n += S
    ctx.synchronize()
    dt = time.perf_counter() - t0
    out = pt.read()
    pt.close()
    return dt, out


def main():
    res = {}
    cores = os.cpu_count() or 1
    helmet = R.generate_meshlets(sio.load_gltf(os.path.join(ROOT, "assets/damaged-helmet/DamagedHelmet.gltf")))
    ctx = rt.Context(0)

    # config 1: CPU scalar reference path tracer (the oracle), helmet 512x512 1 spp
    W = H = 512
    view = sio.FView().reset(helmet, aspect=1.0)
    osc = ob.OracleScene(helmet)
    d = rt.make_pathtrace_desc(view, W, H, 1)
    t0 = time.perf_counter(); r1 = osc.pathtrace(d, num_threads=1); t1 = time.perf_counter() - t0
    t0 = time.perf_counter(); rN = osc.pathtrace(d, num_threads=cores); tN = time.perf_counter() - t0
    rays = r1["stats"].total_rays()
    res["config1"] = {"scene": "Damaged Helmet 512x512 1 spp, CPU oracle", "rays": rays,
                      "threads_1": {"s_per_frame": round(t1, 3), "Mray_s": round(rays / t1 / 1e6, 3)},
                      "threads_all": {"threads": cores, "s_per_frame": round(tN, 3), "Mray_s": round(rays / tN / 1e6, 3)}}
    # same frame on the GPU: parity + PSNR vs the oracle
    gs = ctx.create_scene(helmet)
    dt, g = time_pathtrace(ctx, gs, view, W, H, 1)
    gt = g["target"].view(np.float16).astype(np.float32)[..., :3]; ot = rN["target"].view(np.float16).astype(np.float32)[..., :3]
    same = float((g["target"] == rN["target"]).all(axis=-1).mean())
    res["config1"]["gpu_vs_oracle"] = {"bit_identical_pixels": round(same, 6), "psnr_hdr_db": psnr(gt, ot, 65504.0),
                                       "psnr_ldr_db": psnr(ob.tonemap(g["history"]).astype(np.float32), ob.tonemap(rN["history"]).astype(np.float32), 255.0),
                                       "rays_equal": g["stats"].total_rays() == rays}
    # config 2: helmet 1080p 1 spp on the GPU (average over 16 samples)
    W, H = 1920, 1080
    view = sio.FView().reset(helmet, aspect=W / H)
    dt, g = time_pathtrace(ctx, gs, view, W, H, 16)
    res["config2"] = {"scene": "Damaged Helmet 1920x1080, 1 spp/frame", "ms_per_frame": round(dt / 16 * 1e3, 3),
                      "Mray_s": round(g["stats"].total_rays() / dt / 1e6, 1), "rays_per_frame": g["stats"].total_rays() // 16}
    hr = R.HostRenderer(ctx, gs, W, H)
    for _ in range(3):
        hr.render(view)
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        hr.render(view)
    ctx.synchronize()
    res["config2"]["raster_ms_per_frame"] = round((time.perf_counter() - t0) / 20 * 1e3, 3)
    hr.close(); gs.close(); osc.close()

    # config 3 / 4: Sponza-stat 4K -- raster frame; path trace 64 spp on this one GPU (config 4 is the same job over 8 GPUs)
    sponza = R.generate_meshlets(syn.sponza_stat(texture_size=1024))
    W, H = 3840, 2160
    view = sio.FView().reset(sponza, aspect=W / H)
    gs = ctx.create_scene(sponza)
    hr = R.HostRenderer(ctx, gs, W, H, sh9_rgb=np.full((9, 3), 0.1, np.float32))
    for _ in range(3):
        hr.render(view)
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        hr.render(view)
    ctx.synchronize()
    res["config3"] = {"scene": "Sponza-stat 3840x2160 raster + clustered deferred", "ms_per_frame": round((time.perf_counter() - t0) / 20 * 1e3, 3),
                      "meshlets": sponza.num_meshlets}
    hr.close()
    dt, g = time_pathtrace(ctx, gs, view, W, H, 64)
    res["config4_on_1_gpu"] = {"scene": "Sponza-stat 3840x2160 64 spp path trace", "seconds": round(dt, 3), "ms_per_sample_frame": round(dt / 64 * 1e3, 3),
                               "Mray_s": round(g["stats"].total_rays() / dt / 1e6, 1)}
    dt8, g8 = time_pathtrace(ctx, gs, view, W, H, 64, in_flight=8)
    res["config4_on_1_gpu"]["samples_in_flight_8"] = {"seconds": round(dt8, 3), "ms_per_sample_frame": round(dt8 / 64 * 1e3, 3), "Mray_s": round(g8["stats"].total_rays() / dt8 / 1e6, 1),
                                                      "history_bit_identical_to_one_sample_per_call": bool((g8["history"] == g["history"]).all())}
    gs.close()

    # config 5 on one GPU: Bistro-scale 4K 16 spp
    t0 = time.perf_counter(); bistro = syn.bistro_scale(); tgen = time.perf_counter() - t0
    t0 = time.perf_counter(); gs = ctx.create_scene(bistro); tbuild = time.perf_counter() - t0
    info = gs.info()
    view = sio.FView().reset(bistro, aspect=W / H)
    dt, g = time_pathtrace(ctx, gs, view, W, H, 16, in_flight=8)
    res["config5_on_1_gpu"] = {"samples_in_flight": 8, "scene": "Bistro-scale synthetic 3840x2160 16 spp path trace", "triangles": int(info.num_triangles), "instances": int(info.num_instances),
                               "scene_generate_s": round(tgen, 2), "bvh_build_upload_s": round(tbuild, 2), "seconds": round(dt, 3),
                               "ms_per_sample_frame": round(dt / 16 * 1e3, 3), "Mray_s": round(g["stats"].total_rays() / dt / 1e6, 1)}
    gs.close(); ctx.close()
    res["host"] = {"threads": cores}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
