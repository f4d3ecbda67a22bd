"""8-GPU behaviour of the path tracer measured on ONE GPU: the 4K frame is rendered as N framebuffer shards
(incr_shard{8, r, N}: interleaved 8-row blocks, exactly what rank r of an N-GPU run renders) one after the other, and
the time of every shard is compared with the full frame.  On N GPUs the shards run concurrently, so
  projected N-GPU frame time = max over shards (+ one RCCL all_gather of 33 MB per gathered frame),
  projected speed-up = full-frame ms / max shard ms.
Writes a JSON line (also to --out).  usage: python tools/shard_emulation.py [--shards 8] [--res 3840x2160] [--steps 8]"""
import argparse, ctypes as C, importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "incarnation-renderer_amd"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shards", type=int, default=8); ap.add_argument("--res", default="3840x2160")
    ap.add_argument("--steps", type=int, default=8); ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scene", default="sponza"); ap.add_argument("--texture-size", type=int, default=1024)
    ap.add_argument("--out", default=None)
    ap.add_argument("--samples-in-flight", type=int, default=1,
                    help="incr_pathtrace_desc::num_samples of the SHARD runs (the full frame stays one sample per call): a shard then traces S samples of its "
                         "pixels as one wavefront; times are per sample")
    ap.add_argument("--pipelines", type=int, default=0, help="incr_set_pathtrace_pipelines (0 = the library's choice by size)")
    ap.add_argument("--overlap", default="auto", choices=["auto", "on", "off"], help="incr_set_pathtrace_overlap (auto = the library's default)")
    ap.add_argument("--shard-list", default="", help="comma-separated shard indices to time instead of all (a quick A/B)")
    a = ap.parse_args()
    import torch  # noqa: F401  (first HIP runtime in the process, as in bench.py)
    A = importlib.import_module(PKG + ".ctypes_api"); sio = importlib.import_module(PKG + ".scene_io")
    rt = importlib.import_module(PKG + ".runtime"); bench = importlib.import_module("bench")
    W, H = (int(v) for v in a.res.lower().split("x"))
    packed = bench.build_scene(a.scene, a.texture_size)
    view = sio.FView().reset(packed, aspect=W / H)
    ctx = rt.Context(0)
    scene = ctx.create_scene(packed)
    ctx.set_pathtrace_pipelines(a.pipelines)
    if a.overlap != "auto":
        ctx.set_pathtrace_overlap(a.overlap == "on")

    def timed(shard, S=1):
        pt = rt.PathTracer(ctx, scene, W, H, shard=shard)
        n = 1
        for i in range(a.warmup):
            pt.render(view, n, count_rays=False, num_samples=S); n += S
        ctx.synchronize()
        ctx.clear(pt.stats, C.sizeof(A.RayStats))
        t0 = time.perf_counter()
        for i in range(a.steps):
            pt.render(view, n, num_samples=S); n += S
        ctx.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / (a.steps * S)          # per sample
        rays = pt.read()["stats"].total_rays() / (a.steps * S)
        # per-kernel HIP-event times of the same frames (a second pass: the events serialise the host side a little)
        ctx.profile_begin()
        for i in range(a.steps):
            pt.render(view, n, num_samples=S); n += S
        ctx.synchronize()
        p = ctx.profile_end()
        kernels = {k: round(getattr(p, k + "_ms") / (a.steps * S), 4) for k in ("generate", "extend", "shade", "shadow_gen", "shadow", "accumulate")}
        pt.close()
        return ms, rays, kernels

    full_ms, full_rays, full_kernels = timed(None)
    shard_ms, shard_rays, shard_kernels = [], [], None
    for r in ([int(v) for v in a.shard_list.split(",")] if a.shard_list else range(a.shards)):
        # best of two timings per shard: the shards of a real run share nothing, a hiccup of this process is not theirs
        ms, rays, kernels = min((timed(A.Shard(8, r, a.shards), a.samples_in_flight) for _ in range(2)), key=lambda t: t[0])
        shard_ms.append(round(ms, 4)); shard_rays.append(int(rays))
        if shard_kernels is None:
            shard_kernels = kernels
    out = {"tool": "shard_emulation", "scene": a.scene, "resolution": [W, H], "shards": a.shards, "steps": a.steps, "samples_in_flight": a.samples_in_flight, "pipelines": a.pipelines, "overlap": a.overlap,
           "full_frame_ms": round(full_ms, 4), "full_frame_rays": int(full_rays), "shard_ms": shard_ms, "shard_rays": shard_rays,
           "max_shard_ms": max(shard_ms), "sum_shard_ms": round(sum(shard_ms), 4),
           "projected_speedup_compute_only": round(full_ms / max(shard_ms), 3),
           "shard_overhead_vs_full": round(sum(shard_ms) / full_ms, 4),
           "kernel_ms_full_frame": full_kernels, "kernel_ms_shard0": shard_kernels,
           "note": "one GPU, shards rendered sequentially, best of two timings per shard; times per sample (a shard call traces samples_in_flight samples as one wavefront); excludes the RCCL all_gather (33 MB per gathered 4K frame, once per --spp-per-gather samples)"}
    line = json.dumps(out)
    print(line, flush=True)
    if a.out:
        with open(a.out, "w") as f:
            f.write(line + "\n")


if __name__ == "__main__":
    main()
