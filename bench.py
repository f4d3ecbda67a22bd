#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native Incarnation render hot path.

Metric (BASELINE.json): Mray/s (+ ms/frame) at 4K, 1 spp, Sponza, path tracer.  A "step" is one
1-spp frame of the path tracer at 3840x2160 on the deterministic "Sponza-stat" scene
(Sponza.bin is absent from the reference checkout) accumulated into the
R11G11B10F history, exactly what RenderJob::PathTracing::Execute does per displayed frame.

  python bench.py --gpus N --steps K --warmup W          (N > 1: under torch.distributed.run, or alone -- it then starts its N ranks itself)

N > 1 shards the framebuffer in interleaved 8-row blocks (one process per GPU, scene replicated) and
gathers the HDR history to every rank with one RCCL all_gather per frame (strong scaling: the frame is fixed).
Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "incarnation-renderer_amd"

# Algorithmic bytes (SURVEY.md 8d / DESIGN.md "Measurement"): per traversal query 32 + ceil(log2 T)*32 + 36 + 16,
# per shaded hit 378, per pixel-sample 24.
B_HIT, B_PX = 378, 24
# cycles a SIMD-32 spends issuing one wave64 vector instruction: the hardware guide's rate, and the slot cost measured on the traversal loop's mix at 7 waves per SIMD
# (profiles/r05_issue_rate.txt)
VALU_CYCLES_SIMD32 = 2.0
VALU_CYCLES_MEASURED_MIX = 1.18
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)


def kernel_source_hash():
    """Stamp of the path-tracing kernel sources + BVH layout: ties a PMC traffic measurement to the code it was taken on."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, PKG, "csrc")
    for f in ("pathtrace_wf.hip", "pt_traverse.h", "pt_shade.h", "pt_device.h", "pt_params.h", "device_types.h", "scene_build.cpp", "api.hip", "api_internal.h", "Makefile"):
        with open(os.path.join(csrc, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def bytes_per_ray(num_tris):
    return 32 + int(np.ceil(np.log2(max(num_tris, 2)))) * 32 + 36 + 16


def build_scene(name, texture_size):
    sio = importlib.import_module(PKG + ".scene_io")
    syn = importlib.import_module(PKG + ".synth")
    if name == "sponza":
        return syn.sponza_stat(texture_size=texture_size)
    if name == "helmet":
        return sio.load_gltf(os.path.join(ROOT, "assets/damaged-helmet/DamagedHelmet.gltf"))
    if name == "bistro":
        return syn.bistro_scale()
    raise SystemExit("unknown scene " + name)


def cpu_baseline(packed, view, res, seconds_hint, envmap=None):
    """Oracle (scalar CPU restatement, kind "port") on a bounded sample of the same workload."""
    from oracle import binding as ob
    rt = importlib.import_module(PKG + ".runtime")
    sio = importlib.import_module(PKG + ".scene_io")
    cores = os.cpu_count() or 1
    osc = ob.OracleScene(packed)
    # probe at 1/16 of the requested sample, then size the sample for ~seconds_hint of CPU work (bounded by 4K)
    W, H = res
    pw, ph = max(W // 4, 16), max(H // 4, 16)
    t0 = time.time()
    osc.pathtrace(rt.make_pathtrace_desc(sio.FView().reset(packed, aspect=pw / ph), pw, ph, 1, envmap=envmap), num_threads=cores)
    probe = max(time.time() - t0, 1e-3)
    scale = min(max((seconds_hint / (probe * 16.0)) ** 0.5, 1.0), 3840.0 / W)
    W, H = int(W * scale) // 8 * 8, int(H * scale) // 8 * 8
    view = sio.FView().reset(packed, aspect=W / H)
    d = rt.make_pathtrace_desc(view, W, H, 1, envmap=envmap)
    t0 = time.time()
    r = osc.pathtrace(d, num_threads=cores)
    dt = time.time() - t0
    rays = r["stats"].total_rays()
    osc.close()
    return {"value": round(rays / dt / 1e6, 4), "unit": "Mray/s", "cores": cores, "kind": "port",
            "sample": "oracle/oracle_pt.cpp, same scene and camera, %dx%d 1 spp (%d rays) in %.2f s on %d threads" % (W, H, rays, dt, cores),
            "seconds": round(dt, 3)}


# Raster frame (BASELINE config 3): algorithmic HBM bytes per pixel as fixed by SURVEY.md 8d: visibility write 4 + 4;
# G-buffer read 4, write 16; direct read 16, rw 8; clustered read 24, rw 8; sky read 17, rw 8; TAA read 12, write 4;
# tonemap read 4, write 4 => 133 B/pixel (geometry, light lists and textures are extra and small).
RASTER_B_PX = 133


class _Collectives:
    """The path's collectives.  Over RCCL they run on the device tensors as they are.  In the one-GPU rehearsal of the N > 1 path
    (--share-gpu: every rank renders on GPU 0, torch.distributed over gloo) the same calls stage through the host: the rank
    choreography, the shard / band arithmetic and every buffer shape are the ones the RCCL run uses, only the transport differs."""

    def __init__(self, dist, torch, staged):
        self.dist, self.torch, self.staged = dist, torch, staged
        self.ReduceOp = dist.ReduceOp

    def all_gather_into_tensor(self, out, inp):
        if not self.staged:
            return self.dist.all_gather_into_tensor(out, inp)
        host = self.torch.empty(out.numel(), dtype=out.dtype)          # gloo wants the concatenation as one flat tensor
        self.dist.all_gather_into_tensor(host, inp.cpu().contiguous().view(-1))
        out.copy_(host.view(out.shape))

    def all_reduce(self, t, op):
        if not self.staged:
            return self.dist.all_reduce(t, op=op)
        host = t.cpu()
        self.dist.all_reduce(host, op=op)
        t.copy_(host)

    def barrier(self):
        self.dist.barrier()

    def destroy_process_group(self):
        self.dist.destroy_process_group()


class _DeviceWords:
    """CUDA array interface over a raw device pointer: lets torch.distributed collectives run on a library-owned surface."""

    def __init__(self, ptr, words):
        self.__cuda_array_interface__ = {"shape": (words,), "typestr": "<i4", "data": (int(ptr), False), "version": 2}


def raster_leg(ctx, scene, packed, view, W, H, steps, warmup, stream_handle, torch, dist, rank, world, dev, ibl_inputs, collective=False, checksums=False):
    """ms/frame of the visibility-buffer raster + clustered deferred frame (batch cull .. tonemap), inputs resident in HBM.
    world > 1: every rank renders its row band; one RCCL all_gather of the TAA bands per frame; rank 0 tonemaps."""
    R = importlib.import_module(PKG + ".raster")
    A = importlib.import_module(PKG + ".ctypes_api")
    rr = R.HostRenderer(ctx, scene, W, H, stream=stream_handle, ibl=ibl_inputs)      # C++ Renderer::Render (csrc/host/renderer.cpp)
    rows = (H + world - 1) // world
    tensors = {}
    collective = collective or world > 1
    if collective:
        rr.set_band(rank * rows, min((rank + 1) * rows, H))

    def frame():
        rr.render(view)
        if collective:
            ptr = rr.surface_ptr(A.SURFACE_TAA_ACCUMULATION)
            if ptr not in tensors:
                tensors[ptr] = torch.as_tensor(_DeviceWords(ptr, H * W), device=dev)
            surface = tensors[ptr]
            if world * rows == H:            # equal bands: gather straight into the surface
                band = surface[rank * rows * W:(rank + 1) * rows * W].clone()
                dist.all_gather_into_tensor(surface, band)
            else:                            # the last band is shorter: gather padded bands, copy the frame's rows back
                band = torch.zeros(rows * W, dtype=surface.dtype, device=dev)
                mine = surface[rank * rows * W:min((rank + 1) * rows, H) * W]
                band[:mine.numel()] = mine
                padded = tensors.setdefault("padded", torch.empty(world * rows * W, dtype=surface.dtype, device=dev))
                dist.all_gather_into_tensor(padded, band)
                surface.copy_(padded[:H * W])
            if rank == 0:
                rr.tonemap()

    def barrier():
        if collective:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(warmup):
        frame()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        frame()
    barrier()
    tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    if collective:
        dist.all_reduce(tt, dist.ReduceOp.MAX)
    ms = float(tt.item()) / steps * 1e3
    taa_crc = None
    if checksums:
        import zlib
        taa_crc = zlib.crc32(ctx.download(rr.surface_ptr(A.SURFACE_TAA_ACCUMULATION), (H * W,), np.uint32).tobytes())
    # per-pass GPU time of one more frame (events on the submission stream) and the algorithmic bytes each pass moves per pixel
    # (SURVEY.md 8d): visibility 8 (id + depth written), G-buffer 4 + 16, fused lighting 16 read + 8 colour rw (unfused: 3 x 24),
    # dynamic sky 4 + 4, TAA 12 + 4 (+ 4 LDR when the tonemap is fused), tonemap 4 + 4
    per_pass = None
    iso_ms = None
    if world == 1:
        rr.set_profiling(True); rr.render(view); times = rr.pass_times(); rr.set_profiling(False)
        if (int(packed.max_anisotropy) or 16) != 1:          # the same frame through the isotropic sampler of rounds 1-2, for comparison with their records
            rt_mod = importlib.import_module(PKG + ".runtime")
            rt_mod.check(ctx.lib.incr_scene_set_max_anisotropy(ctx.handle, scene.handle, 1))
            for _ in range(2):
                frame()
            barrier(); t1 = time.perf_counter()
            for _ in range(steps):
                frame()
            barrier(); iso_ms = (time.perf_counter() - t1) / steps * 1e3
            rt_mod.check(ctx.lib.incr_scene_set_max_anisotropy(ctx.handle, scene.handle, int(packed.max_anisotropy)))
        bpp = {"visibility": 8, "gbuffer": 20, "deferred_lighting": 24, "sky_lighting": 24, "direct_lighting": 24, "clustered_lighting": 24,
               "dynamic_sky": 8, "environmentmap": 8, "taa_resolve": 20, "tonemap": 8, "hbao": 13}
        per_pass = {}
        for name, t in times:
            e = per_pass.setdefault(name, {"ms": 0.0})
            e["ms"] = round(e["ms"] + t, 4)
        for name, e in per_pass.items():
            if name in bpp and e["ms"] > 0:
                e["algorithmic_GBps"] = round(float(W) * H * bpp[name] / (e["ms"] * 1e-3) / 1e9, 1)
    host_s, g_frames, g_updates, g_rebuilds = rr.frame_info()
    counts = ctx.download(rr.surface_ptr(A.SURFACE_BATCH_COUNTS), (2,), np.uint32)
    raw = ctx.download(rr.surface_ptr(A.SURFACE_RAY_STATS), (C.sizeof(A.RayStats),), np.uint8)
    rays = A.RayStats.from_buffer_copy(raw.tobytes()).light_shadow_rays // (steps + warmup)
    rr.close()
    alg = float(W) * H * RASTER_B_PX
    return {"ms_per_frame": round(ms, 4), "fps": round(1e3 / ms, 1), "n_gpus": world, **({"ms_per_frame_isotropic_sampler": round(iso_ms, 4)} if iso_ms is not None else {}),
            "sharding": "none" if not collective else "sort-first row bands of %d rows + RCCL all_gather of the TAA bands" % rows,
            "workload": "visibility-buffer raster + clustered deferred shading, %dx%d, %d meshlets (%d + %d drawn), SH9 + split-sum specular sky lighting, 1 sun shadow ray/pixel, dynamic sky, TAA, material sampler anisotropic x%d"
                        % (W, H, packed.num_meshlets, int(counts[0]), int(counts[1]), int(packed.max_anisotropy) or 16),
            "passes": "batch_cull, light_cull, visibility, gbuffer, deferred_lighting (sky + sun + clustered fused), dynamic_sky, taa_resolve (+ tonemap fused)",
            "per_pass": per_pass, "host_us_per_frame": round(host_s * 1e6, 1),
            "sun_shadow_rays_per_frame": int(rays), "algorithmic_bytes_per_frame": int(alg),
            "achieved_GBps": round(alg / (ms * 1e-3) / 1e9, 2), "frac_of_hbm_peak": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
            **({"taa_accumulation_crc32": taa_crc} if taa_crc is not None else {})}


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes of this script (one per GPU, rendezvous on
    127.0.0.1), before this process has touched the GPU, and wait for them.  Rank 0's stdout is the parent's (the ONE JSON line);
    the first non-zero exit code is propagated and the remaining ranks are stopped (a dead peer would leave them in a collective)."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=(r == 0)))

    def forward():      # rank 0's JSON line -> stdout; anything else a library printed on its stdout (gloo's connection banner) -> stderr
        for line in procs[0].stdout:
            (sys.stdout if line.startswith('{"metric"') else sys.stderr).write(line)
            sys.stdout.flush()
    import threading
    fw = threading.Thread(target=forward, daemon=True)
    fw.start()
    code = 0
    live = list(procs)
    while live and code == 0:
        time.sleep(0.05)
        for pr in list(live):
            rc = pr.poll()
            if rc is None:
                continue
            live.remove(pr)
            if rc != 0:
                code = rc
                break
    if code != 0:
        for pr in live:
            pr.terminate()
        deadline = time.time() + 10
        for pr in live:
            try:
                pr.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                pr.kill()
                pr.wait()
    fw.join(timeout=5)
    return code if code >= 0 else 1


def dry_run(args, rank, world):
    """--dry-run (CPU, gloo): the rank choreography of the real run -- rendezvous, the shard arithmetic, barrier-bracketed timing
    with the max over ranks, one gather of the per-rank history shards per step, rank 0's single JSON line -- with no render
    work in it.  Exists so that the launch path (`--gpus N` with and without a launcher) is testable without a GPU."""
    import torch
    import torch.distributed as dist
    W, H = (int(v) for v in args.res.lower().split("x"))
    if world > 1:
        dist.init_process_group(backend=args.backend)
    if args.fail_rank == rank:
        raise RuntimeError("--fail-rank %d: injected failure" % rank)
    block = 8
    blocks = (H + block - 1) // block
    rows = ((blocks - rank + world - 1) // world) * block if world > 1 else H      # interleaved 8-row blocks, as incr_shard_local_rows (padding rows included)
    rows_max = ((blocks + world - 1) // world) * block if world > 1 else H
    history = torch.full((rows_max, W), rank, dtype=torch.int32)
    gathered = torch.zeros((world * rows_max, W), dtype=torch.int32) if world > 1 else None      # gloo wants the concatenated shape

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        pass
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if world > 1:
            dist.all_gather_into_tensor(gathered, history)
    barrier()
    tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    px = torch.tensor([rows * W], dtype=torch.int64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(px, op=dist.ReduceOp.SUM)
        assert all(int(gathered[r * rows_max, 0]) == r for r in range(world))
    if rank == 0:
        print(json.dumps({"metric": "Mray/s", "value": 0.0, "unit": "Mray/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(float(tt.item()) / max(args.steps, 1) * 1e3, 4), "higher_is_better": True, "scaling": "strong",
                          "vs_baseline": None, "dtype": "f32", "data": "synthetic", "dry_run": True,
                          "config": {"workload": "dry run of the rank choreography (no render work)", "resolution": [W, H], "backend": args.backend,
                                     "local_rows_sum": int(px.item()) // W}}), flush=True)
    barrier()
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scene", default="sponza", choices=["sponza", "helmet", "bistro"])
    ap.add_argument("--res", default="3840x2160")
    ap.add_argument("--texture-size", type=int, default=1024)
    ap.add_argument("--cpu-res", default="960x540")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-direct-lighting", action="store_true", help="diagnostic: DIRECT_LIGHTING=0 (not the headline workload)")
    ap.add_argument("--viewmode", type=int, default=0, help="diagnostic: FConfig::Viewmode")
    ap.add_argument("--max-anisotropy", type=int, default=0, help="raster leg: MaxAnisotropy of the material sampler (0 = the reference's 16; 1 = isotropic trilinear, a diagnostic)")
    ap.add_argument("--block-compress", action="store_true", help="diagnostic: hand the textures over as BC3 / BC5 chains (bc.py's encoder) instead of RGBA8")
    ap.add_argument("--no-raster", action="store_true", help="skip the raster-frame leg (ms/frame of BASELINE config 3; N = 1 only)")
    ap.add_argument("--pipelines", type=int, default=0,
                    help="concurrent bands of tile rows a path-traced frame is issued as (incr_set_pathtrace_pipelines; results are bit-identical; "
                         "0 = the library's choice: 1 with the side stream of --overlap, else 2 for a frame and 4 for a framebuffer shard of fewer than 1.5 M pixels). "
                         "With more than one, launches of different bands overlap and stretch each other: the per-kernel times then price launches "
                         "that share the GPU; --pipelines 1 times every kernel alone (profiles/ holds both)")
    ap.add_argument("--overlap", default="auto", choices=["auto", "on", "off"],
                    help="incr_set_pathtrace_overlap: the any-hit launches (and the deferred summation) of a pipeline run on a side stream beside its main chain "
                         "primary -> shade -> extend -> shade ...; auto = the library's default (on), except with an explicit --pipelines 1, whose purpose is to time every kernel alone")
    ap.add_argument("--spp-per-gather", type=int, default=64,
                    help="N > 1: samples accumulated in each rank's history shard between two RCCL gathers of the HDR buffer "
                         "(BASELINE config 4 = 64 spp per gathered frame, SURVEY.md 8e); the last timed step always gathers")
    ap.add_argument("--samples-in-flight", type=int, default=0,
                    help="samples (= steps) a rank traces as ONE wavefront (incr_pathtrace_desc::num_samples; the history is bit-identical to one "
                         "call per sample).  0 = 1 at N = 1 (a step is then exactly one 1-spp frame, the headline) and 8 at N > 1, where a rank's "
                         "1/N framebuffer shard alone is too short a ray list to fill the GPU between two launch tails (BASELINE config 4: 64 spp per gathered frame)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend (nccl = RCCL; gloo only with --dry-run)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal of the N > 1 path on ONE GPU: every rank renders on GPU 0 and the collectives go over gloo through the host "
                                                             "(with --backend gloo; the numbers are not a scaling measurement and the JSON line says so)")
    ap.add_argument("--force-collectives", action="store_true",
                    help="N = 1 only: run the N > 1 code path at world size 1 over RCCL -- init_process_group('nccl', device_id), the 8-row-block shard descriptor, "
                         "all_gather_into_tensor of the history + incr_unshard per gathered frame, the TAA band gather on the library-owned surface, "
                         "destroy_process_group.  Results equal the plain run bit for bit (--checksums); proves the transport branch on one GPU")
    ap.add_argument("--checksums", action="store_true", help="add CRC-32s of the final history (and of the raster leg's TAA accumulation surface) to the JSON line")
    ap.add_argument("--dry-run", action="store_true", help="CPU rehearsal of the multi-rank choreography (rendezvous, barriers, gather, the JSON line) without render work")
    ap.add_argument("--fail-rank", type=int, default=-1, help="test hook (--dry-run): this rank raises after the rendezvous")
    args = ap.parse_args()

    # `python bench.py --gpus N` as the driver invokes it (no launcher): this process becomes the launcher -- BEFORE anything
    # touches the GPU (no torch.cuda call, no rt.Context yet; a process that has initialised the GPU must not re-exec) -- and the
    # N ranks are fresh child processes.  Under torch.distributed.run WORLD_SIZE is already set and this branch is not taken.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node %d (or without a launcher)" % (args.gpus, world, args.gpus))
    if args.dry_run:
        return dry_run(args, rank, world)
    if (args.backend != "nccl") != bool(args.share_gpu):
        raise SystemExit("--backend gloo goes with --dry-run (no render work) or with --share-gpu (all ranks on GPU 0); the measured path runs over RCCL, one GPU per rank")
    import torch
    A = importlib.import_module(PKG + ".ctypes_api")
    sio = importlib.import_module(PKG + ".scene_io")
    rt = importlib.import_module(PKG + ".runtime")

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if args.force_collectives and (world != 1 or args.share_gpu):
        raise SystemExit("--force-collectives is the one-rank run of the N > 1 path: use it with --gpus 1")
    collective = world > 1 or args.force_collectives      # the framebuffer goes through the shard / gather / unshard path
    if collective:
        import torch.distributed as dist_mod
        if args.share_gpu:
            dist_mod.init_process_group(backend="gloo")
        elif world == 1 and "MASTER_ADDR" not in os.environ:      # no launcher: a rendezvous of one on the loopback
            import socket
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                port = so.getsockname()[1]
            dist_mod.init_process_group(backend=args.backend, device_id=dev, init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
        else:
            dist_mod.init_process_group(backend=args.backend, device_id=dev)
        dist = _Collectives(dist_mod, torch, staged=bool(args.share_gpu))

    W, H = (int(v) for v in args.res.lower().split("x"))
    packed = build_scene(args.scene, args.texture_size)
    packed.max_anisotropy = args.max_anisotropy
    if args.block_compress:
        packed.compress_textures()
    packed = importlib.import_module(PKG + ".raster").generate_meshlets(packed)       # FScene::GenerateMeshlets at load time
    view = sio.FView().reset(packed, aspect=W / H)
    ctx = rt.Context(local_rank)
    ctx.set_pathtrace_pipelines(max(0, min(args.pipelines, 8)))
    if args.overlap != "auto" or args.pipelines == 1:
        ctx.set_pathtrace_overlap(args.overlap == "on")
    scene = ctx.create_scene(packed)
    info = scene.info()
    # load-time precompute, as the reference does once per scene / sun change (FScene::UpdateDynamicSky at
    # EnvmapResolution 256, Renderer::GenerateEnvBrdfTexture 512x512): feeds the sky NEE of the path tracer and sky lighting
    ibl_mod = importlib.import_module(PKG + ".ibl")
    t_ibl = time.perf_counter()
    sky = ibl_mod.SkyIbl(ctx, scene)
    env_brdf = ibl_mod.generate_env_brdf(ctx, 512, 512)
    ctx.synchronize()
    t_ibl = time.perf_counter() - t_ibl
    envmap = (sky.info.envmap_r11g11b10, sky.info.cubemap_res, sky.info.num_mips)
    ibl_inputs = importlib.import_module(PKG + ".raster").IblInputs.from_sky(sky, env_brdf, (512, 512))

    shard = A.Shard(8, rank, world) if collective else A.Shard(0, 0, 1)
    rows = ctx.lib.incr_shard_local_rows(H, C.byref(shard))
    target = torch.zeros((rows, W, 4), dtype=torch.int16, device=dev)        # RGBA16F bits
    history = torch.zeros((rows, W), dtype=torch.int32, device=dev)          # R11G11B10F bits
    stats = torch.zeros(8, dtype=torch.int64, device=dev)
    gathered = torch.zeros((world, rows, W), dtype=torch.int32, device=dev) if collective else None
    full = torch.zeros((H, W), dtype=torch.int32, device=dev) if collective else None
    stream = torch.cuda.current_stream(dev)

    sif = args.samples_in_flight if args.samples_in_flight > 0 else (8 if world > 1 else 1)
    gather_events = []          # (start, end) around every all_gather + incr_unshard, on the render stream

    def step(sample_index, e0=None, e1=None, gather=True, num_samples=1):
        d = rt.make_pathtrace_desc(view, W, H, sample_index, shard, envmap=envmap, num_samples=num_samples)
        d.scene = scene.handle
        if args.no_direct_lighting:
            d.direct_lighting = 0
        d.viewmode = args.viewmode
        d.target_rgba16f, d.history_r11g11b10, d.ray_stats = target.data_ptr(), history.data_ptr(), stats.data_ptr()
        if e0 is not None:
            e0.record(stream)
        ctx.pathtrace(d, stream.cuda_stream)
        if e1 is not None:
            e1.record(stream)
        if collective and gather:   # the one exchange step of the path: gather disjoint framebuffer shards (final HDR of the accumulation)
            g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            g0.record(stream)
            dist.all_gather_into_tensor(gathered, history)
            ctx.unshard(full.data_ptr(), gathered.data_ptr(), W, H, shard.row_block_height, world, 4, stream.cuda_stream)
            g1.record(stream)
            gather_events.append((g0, g1))

    def barrier():
        if collective:
            dist.barrier()
        torch.cuda.synchronize(dev)

    sample = 1
    # warm-up in the SAME call shape as the timed steps (`sif` samples per call): the library sizes its path-state workspace by the samples
    # in flight, and the first call of a shape allocates it
    w = 0
    while w < args.warmup:
        S = min(sif, max(args.steps, 1))
        step(sample, num_samples=S); sample += S; w += S
    barrier()
    stats.zero_()
    gather_events.clear()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    ctx.profile_begin()        # HIP events around every kernel of incr_pathtrace, recorded on the launch stream
    barrier()
    t0 = time.perf_counter()
    spg = max(args.spp_per_gather, 1)
    i = calls = 0
    while i < args.steps:      # the K steps (= K samples of the frame), `sif` of them per incr_pathtrace call, never across a gather
        S = min(sif, args.steps - i, spg - (i % spg))
        step(sample, ev[calls][0], ev[calls][1], gather=((i + S) % spg == 0 or i + S == args.steps), num_samples=S)
        sample += S; i += S; calls += 1
    ev = ev[:calls]
    barrier()
    dt = time.perf_counter() - t0
    prof = ctx.profile_end()
    # max over ranks of the wall time, sum over ranks of the rays traced
    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    local_stats = stats.clone()
    if collective:
        dist.all_reduce(tt, dist.ReduceOp.MAX)
        dist.all_reduce(stats, dist.ReduceOp.SUM)
    dt = float(tt.item())
    s = stats.cpu().numpy()
    ls = local_stats.cpu().numpy()
    rays = int(s[0] + s[1] + s[2] + s[3]); hits = int(s[4])
    call_ms = float(np.sum([a.elapsed_time(b) for a, b in ev])) / args.steps   # HIP events around the whole incr_pathtrace call, per step (sample)

    # The same kernels timed ALONE (one pipeline, no bounce overlap: every launch has the GPU to itself), a few extra steps outside the timed
    # region: the record then carries both the shared-GPU interval of the dominant kernel and its stand-alone one (roofline.alone)
    alone = None
    if world == 1 and not collective and (int(prof.pipelines) > 1 or prof.overlap):
        ctx.set_pathtrace_pipelines(1); ctx.set_pathtrace_overlap(False)
        stats_a = torch.zeros(8, dtype=torch.int64, device=dev)
        hist_a = history.clone()                                      # the timed accumulation stays what the K steps left
        n_alone = max(2, min(4, args.steps))

        def step_alone(si):
            d = rt.make_pathtrace_desc(view, W, H, si, shard, envmap=envmap)
            d.scene = scene.handle
            if args.no_direct_lighting:
                d.direct_lighting = 0
            d.viewmode = args.viewmode
            d.target_rgba16f, d.history_r11g11b10, d.ray_stats = target.data_ptr(), hist_a.data_ptr(), stats_a.data_ptr()
            ctx.pathtrace(d, stream.cuda_stream)
        step_alone(sample); torch.cuda.synchronize(dev); stats_a.zero_()
        ctx.profile_begin()
        for k in range(n_alone):
            step_alone(sample + 1 + k)
        torch.cuda.synchronize(dev)
        alone = (ctx.profile_end(), stats_a.cpu().numpy(), n_alone)
        ctx.set_pathtrace_pipelines(max(0, min(args.pipelines, 8)))
        if args.overlap != "auto":
            ctx.set_pathtrace_overlap(args.overlap == "on")
        else:
            ctx.set_pathtrace_overlap(True)

    # second half of the metric ("Mray/s + ms/frame"): the raster frame of config 3, outside the timed path-trace region;
    # at N > 1 every rank renders its row band and the TAA bands are all-gathered (sort-first)
    raster_out = None
    if not args.no_raster:
        try:
            raster_out = raster_leg(ctx, scene, packed, view, W, H, args.steps, args.warmup, stream.cuda_stream, torch, dist, rank, world, dev, ibl_inputs,
                                    collective=collective, checksums=args.checksums)
            raster_out["sky_ibl_precompute_ms"] = round(t_ibl * 1e3, 2)      # once per scene / sun change, not per frame
        except Exception as e:      # the headline (path trace) line must survive a failure of the secondary leg
            if collective:          # ... but not at N > 1: the leg contains collectives, a rank that skipped them would hang the others
                raise
            raster_out = {"error": "%s: %s" % (type(e).__name__, e)}

    if rank == 0:
        b_ray = bytes_per_ray(int(info.num_triangles))
        answered = int(prof.shadow_rays_answered)          # over the profiled steps, like the ray statistics `ls`
        # Roofline of the DOMINANT kernel on this rank.  incr_pathtrace is 16 launches per pipeline (4 x extend / shade / shadow / accumulate; bounce 0's
        # extend makes the camera rays itself and shade sets its vertices' shadow rays up: the "generate" and "shadow_gen" intervals of rounds 1-2 are empty);
        # the per-kernel HIP-event sums say which one dominates.  Traversal kernels move
        # bytes_per_ray per query, the shading kernels bytes_per_hit per shaded hit, shadow_gen 32 B per queued ray + the request mask.
        per_kernel = {"incr::wf_extend_kernel": (prof.extend_ms, max(prof.launches[0], 1), float(ls[0] + ls[1]) * b_ray),
                      "incr::wf_shade_kernel": (prof.shade_ms, max(prof.launches[1], 1), float(ls[4]) * B_HIT),
                      # ... less the rays the sun cover map answered at their set-up: they have no queue entry, the any-hit kernel never sees them
                      "incr::wf_shadow_kernel": (prof.shadow_ms, max(prof.launches[3], 1), float(ls[2] + ls[3] - answered) * b_ray),
                      "incr::wf_accumulate_kernel": (prof.accumulate_ms, max(prof.launches[4], 1), float(ls[4]) * 128.0 + rows * W * B_PX * args.steps)}
        # With more than one pipeline in flight the event intervals of the streaming kernels (generate, shade, shadow_gen, accumulate)
        # mostly measure how long their workgroups WAITED for the CUs the other band's persistent traversal kernel holds (2-4 x their
        # stand-alone durations, the traversal kernels 1.3-1.4 x: profiles/r02_bench_pipelines1.json), so the sums no longer rank the
        # kernels by GPU time: the dominant kernel is then chosen among the two traversal kernels, which dominate when timed alone.
        # Between those two the overlapped intervals are within a few per cent of each other and trade places from run to run, while alone
        # the any-hit kernel is the longer one (7.6 vs 7.0 ms per frame) because it traces more rays: with pipelines > 1 the one that moves
        # more algorithmic bytes is taken -- a choice that does not flip with timing noise.
        if prof.pipelines <= 1 and not prof.overlap:
            dominant = max(per_kernel, key=lambda k: per_kernel[k][0])
        else:
            dominant = max(("incr::wf_extend_kernel", "incr::wf_shadow_kernel"), key=lambda k: per_kernel[k][2])
        dom_ms, dom_launches, dom_bytes = per_kernel[dominant]
        kernel_ms = dom_ms / dom_launches if dom_ms > 0 else call_ms
        alg_bytes_per_launch = dom_bytes / dom_launches if dom_ms > 0 else 0.0
        achieved = alg_bytes_per_launch / (kernel_ms * 1e-3) / 1e9
        l_rays = int(ls[0] + ls[1] + ls[2] + ls[3]); l_hits = int(ls[4])
        # an answered ray moves its 16-byte record twice (written by the set-up, read by the summation) and nothing else
        frame_alg_bytes = ((l_rays - answered) * b_ray + answered * 32 + l_hits * B_HIT) / args.steps + rows * W * B_PX
        out = {
            "metric": "Mray/s", "value": round(rays / dt / 1e6, 3), "unit": "Mray/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "path trace %s %dx%d 1 spp/step (BASELINE config 4 scene: deterministic Sponza-stat, 262,267 tris, 65 lights)"
                                   % (args.scene, W, H) if args.scene == "sponza" else "path trace %s %dx%d 1 spp/step" % (args.scene, W, H),
                       "scene": args.scene, "resolution": [W, H], "spp_per_step": 1, "triangles": int(info.num_triangles),
                       "lights": len(packed.global_lights), "texture_size": args.texture_size, **({"textures": "block-compressed chains (BC3 / BC5)"} if args.block_compress else {}), "pipelines": int(prof.pipelines), "bounce_overlap": bool(prof.overlap), "samples_in_flight": sif,
                       "sharding": "none" if not collective else "interleaved 8-row blocks + RCCL all_gather of the R11G11B10F history every %d samples and after the last step" % spg,
                       "rays_per_step": rays // args.steps, "shaded_hits_per_step": hits // args.steps,
                       "ms_per_frame": round(dt / args.steps * 1e3, 4),
                       **({"rehearsal": "--share-gpu: %d ranks on ONE GPU, collectives over gloo through the host -- exercises the N > 1 code path, measures nothing" % world} if args.share_gpu else {})},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                         "kernel": dominant, "kernel_ms": round(kernel_ms, 4), "kernel_launches_per_step": dom_launches / args.steps,
                         "kernel_timing": ("HIP events around every launch on the stream it is issued on; %d pipeline(s)%s: launches share the GPU while they are timed "
                                           "(the sum of kernel_ms_per_step exceeds ms_per_step), so the dominant kernel is picked among the two traversal kernels by the bytes they "
                                           "move; roofline.alone times the same kernel with the GPU to itself" % (prof.pipelines, ", the any-hit launches on a side stream beside the main chain (deferred accumulation)" if prof.overlap else ""))
                                          if (prof.pipelines > 1 or prof.overlap) else "HIP events around every launch, one pipeline, everything on one stream: every kernel has the GPU to itself",
                         "algorithmic_bytes_per_launch": int(alg_bytes_per_launch),
                         "bytes_per_ray": b_ray, "bytes_per_hit": B_HIT, "bytes_per_pixel_sample": B_PX,
                         "whole_call": {"ms": round(call_ms, 4), "algorithmic_bytes": int(frame_alg_bytes),
                                        "achieved": round(frame_alg_bytes / (call_ms * 1e-3) / 1e9, 2)},
                         "kernel_ms_per_step": {k.split("::")[1]: round(v[0] / args.steps, 4) for k, v in per_kernel.items()}},
        }
        # the shadow-ray queue: rays beyond its capacity are traced inline by the summation kernel (same image, slower) -- a silent cliff until round 5
        out["config"]["shadow_queue"] = {"rays_asked_per_step": int(prof.shadow_rays_asked) // max(int(prof.frames), 1), "rays_inline_per_step": int(prof.shadow_rays_inline) // max(int(prof.frames), 1),
                                         "rays_answered_by_sun_cover_per_step": int(prof.shadow_rays_answered) // max(int(prof.frames), 1),
                                         "overflowed_calls": int(prof.shadow_queue_overflows)}
        if prof.shadow_queue_overflows:
            print("bench.py: WARNING: the shadow-ray queue overflowed in %d call(s): %d rays per step were traced inline" % (prof.shadow_queue_overflows, int(prof.shadow_rays_inline) // max(int(prof.frames), 1)), file=sys.stderr)
        if alone is not None:
            pa, sa, na = alone
            alone_kernel = {"incr::wf_extend_kernel": (pa.extend_ms, max(pa.launches[0], 1), float(sa[0] + sa[1]) * b_ray),
                            "incr::wf_shade_kernel": (pa.shade_ms, max(pa.launches[1], 1), float(sa[4]) * B_HIT),
                            "incr::wf_shadow_kernel": (pa.shadow_ms, max(pa.launches[3], 1), float(sa[2] + sa[3] - int(pa.shadow_rays_answered)) * b_ray),
                            "incr::wf_accumulate_kernel": (pa.accumulate_ms, max(pa.launches[4], 1), float(sa[4]) * 128.0 + rows * W * B_PX * na)}
            a_ms, a_launches, a_bytes = alone_kernel[dominant]
            if a_ms > 0:
                a_kernel_ms = a_ms / a_launches
                a_ach = a_bytes / a_launches / (a_kernel_ms * 1e-3) / 1e9
                out["roofline"]["alone"] = {"kernel_ms": round(a_kernel_ms, 4), "algorithmic_bytes_per_launch": int(a_bytes / a_launches), "achieved": round(a_ach, 2),
                                            "frac": round(a_ach / HBM_PEAK_GBS, 5), "launches_per_step": a_launches / na, "steps": na,
                                            "kernel_ms_per_step": {k.split("::")[1]: round(v[0] / na, 4) for k, v in alone_kernel.items()},
                                            "note": "%d extra steps outside the timed region with one pipeline and no bounce overlap: every launch has the GPU to itself (a launch then covers the whole frame)" % na}
        # measured HBM bytes of the dominant kernel: only from a PMC run (tools/pmc.sh) of THESE kernel sources on this workload
        traffic_file = os.path.join(ROOT, "profiles", "hbm_traffic_latest.json")
        if os.path.exists(traffic_file):
            try:
                with open(traffic_file) as f:
                    tj = json.load(f)
                entry = tj.get("kernels", {}).get(dominant)
                default_workload = args.scene == "sponza" and [W, H] == [3840, 2160] and not args.no_direct_lighting and args.viewmode == 0
                if entry and tj.get("kernel_source_hash") == kernel_source_hash() and default_workload and tj.get("pipelines") == int(prof.pipelines) and bool(tj.get("bounce_overlap", False)) == bool(prof.overlap):
                    out["roofline"]["traffic"] = entry.get("hbm_bytes_per_launch")
                    out["roofline"]["traffic_source"] = tj.get("source")
                    # Vector-ALU issue of the dominant kernel, priced twice (round 5; rounds 2-4 priced a wave64 instruction at 4 cycles per SIMD -- what ONE wave sustains -- and
                    # reported a utilisation above 1): against the SIMD-32 rate of the hardware guide (2 cycles per wave64 instruction) and against the slot cost
                    # measured with the loop's own instruction mix at the kernel's occupancy (tools/microbench/issue_rate.hip, profiles/r05_issue_rate.txt).  Neither may
                    # exceed 1: a fraction above 1 means the model is wrong, and the run fails.
                    if entry.get("SQ_INSTS_VALU_per_launch") and entry.get("GRBM_GUI_ACTIVE_per_launch"):
                        insts_per_simd = entry["SQ_INSTS_VALU_per_launch"] / 1024.0
                        elapsed = entry["GRBM_GUI_ACTIVE_per_launch"] / 8.0          # GRBM_GUI_ACTIVE sums the 8 XCDs
                        frac_simd32 = insts_per_simd * VALU_CYCLES_SIMD32 / elapsed
                        frac_measured = insts_per_simd * VALU_CYCLES_MEASURED_MIX / elapsed
                        out["roofline"]["valu"] = {"insts_per_launch": int(entry["SQ_INSTS_VALU_per_launch"]), "elapsed_cycles": int(elapsed),
                                                   "cycles_per_inst_simd32": VALU_CYCLES_SIMD32, "frac_simd32": round(frac_simd32, 4),
                                                   "cycles_per_inst_measured": VALU_CYCLES_MEASURED_MIX, "frac_measured": round(frac_measured, 4),
                                                   "measured_with": "tools/microbench/issue_rate.hip: the traversal loop's mix (cvt_ubyte, fma, min / max, cndmask) at 7 waves per SIMD, every CU loaded",
                                                   "lane_utilisation": round(entry.get("SQ_THREAD_CYCLES_VALU_per_launch", 0.0) / max(entry.get("SQ_ACTIVE_INST_VALU_per_launch", 1.0), 1.0) / 64.0, 4)}
                        if max(frac_simd32, frac_measured) > 1.0:
                            raise RuntimeError("roofline.valu: a vector-issue fraction above 1 (%.3f / %.3f): the issue model does not describe this kernel" % (frac_simd32, frac_measured))
                else:
                    out["roofline"]["traffic_note"] = "profiles/hbm_traffic_latest.json was measured on other kernel sources, another workload or another pipeline count: not attached"
            except Exception:
                pass
        if collective and gather_events:
            gms = [a.elapsed_time(b) for a, b in gather_events]
            out["config"]["gather"] = {"gathers": len(gms), "gather_ms": round(sum(gms) / len(gms), 4), "gather_ms_max": round(max(gms), 4), "bytes_gathered": int(world * rows * W * 4),
                                       "what": "%s all_gather_into_tensor of the R11G11B10F history shards + incr_unshard into the %dx%d frame, HIP events on the render stream (rank 0)" % (args.backend, W, H)}
        if args.force_collectives:
            out["config"]["force_collectives"] = ("world size 1 over %s: init_process_group(device_id), all_gather_into_tensor of the history shard + incr_unshard, "
                                                  "the TAA band gather on the library-owned surface, all_reduce of time and ray counts, destroy_process_group" % args.backend)
        if args.checksums:
            import zlib
            final = (full if collective else history[:H]).cpu().numpy()
            out["config"]["history_crc32"] = zlib.crc32(np.ascontiguousarray(final).tobytes())
        if raster_out is not None:
            out["raster"] = raster_out
        if world == 1 and not args.no_cpu_baseline:
            cw, ch = (int(v) for v in args.cpu_res.lower().split("x"))
            host_env = np.ascontiguousarray(sky.download()["envmap"])      # the same sky cube, on the host, for the CPU port
            out["cpu_baseline"] = cpu_baseline(packed, sio.FView().reset(packed, aspect=cw / ch), (cw, ch), 20,
                                               envmap=(host_env.ctypes.data, sky.info.cubemap_res, sky.info.num_mips))
        print(json.dumps(out), flush=True)
    barrier()
    if collective:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
