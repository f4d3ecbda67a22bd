"""GPU parity tests added in round 3 (run with -m gpu on an MI355X)."""
import ctypes as C
import os

import numpy as np
import pytest

from test_gpu_pathtrace import pixel_agreement, stats_tuple

pytestmark = pytest.mark.gpu


def _render(gpu_ctx, rt, sio, packed, W, H, samples=(1,), env=None):
    old = {}
    for k, v in (env or {}).items():
        old[k] = os.environ.get(k); os.environ[k] = v
    try:
        sc = gpu_ctx.create_scene(packed)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    view = sio.FView().reset(packed, aspect=W / H)
    pt = rt.PathTracer(gpu_ctx, sc, W, H)
    for s in samples:
        pt.render(view, s)
    g = pt.read()
    pt.close(); sc.close()
    return g


# ------------------------------------------------------------------------------------------------ interleaved material texels
@pytest.mark.parametrize("scene_name", ["helmet", "sponza_small"])
def test_interleaved_material_texels_are_bit_identical(gpu_ctx, rt, sio, request, scene_name):
    """chsMain samples base colour, normal and metallic-roughness at one uv (material/common.hlsli:28-71).  The interleaved level-0
    image (device_types.h MaterialTexels: one tap, four 8-byte loads) must give the bytes of the texture-by-texture path."""
    packed = request.getfixturevalue(scene_name)
    W, H = 320, 180
    a = _render(gpu_ctx, rt, sio, packed, W, H, samples=(1, 2))
    b = _render(gpu_ctx, rt, sio, packed, W, H, samples=(1, 2), env={"INCR_MATERIAL_TEXELS": "0"})
    assert stats_tuple(a["stats"]) == stats_tuple(b["stats"])
    np.testing.assert_array_equal(a["target"], b["target"])
    np.testing.assert_array_equal(a["history"], b["history"])
    assert a["stats"].shaded_hits > 0


def test_texture_download_detiles_the_resident_chain(gpu_ctx, rt, helmet):
    """incr_scene_texture_download returns the row-major chain although only the tiled copy stays resident (level 0 = the bytes handed over)."""
    sc = gpu_ctx.create_scene(helmet)
    for ti in (0, len(helmet.textures) - 1):
        w, h, mips, chain = sc.texture_chain(ti)
        img, _fmt = helmet.textures[ti]
        assert (h, w) == img.shape[:2] and mips >= 2
        np.testing.assert_array_equal(chain[: w * h].view(np.uint8).reshape(h, w, 4), img)
    sc.close()


# ------------------------------------------------------------------------------------------------ samples in flight
@pytest.mark.parametrize("scene_name,W,H", [("helmet", 200, 120), ("sponza_small", 256, 144)])
def test_samples_in_flight_equal_successive_calls(gpu_ctx, rt, sio, A, request, scene_name, W, H):
    """incr_pathtrace_desc::num_samples = S traces samples n .. n + S - 1 as one wavefront; the R11G11B10 history (a running mean
    re-quantised at every sample, pathtrace-integrate.hlsl:16-31) and the ray counts must equal S successive calls, the target the
    last sample's.  16 samples as 1 + 3 + 8 + 4 against 16 single calls; also on a framebuffer shard (2 of 3 ranks, ragged rows)."""
    packed = request.getfixturevalue(scene_name)
    sc = gpu_ctx.create_scene(packed)
    view = sio.FView().reset(packed, aspect=W / H)
    for shard in (None, A.Shard(8, 1, 3)):
        a = rt.PathTracer(gpu_ctx, sc, W, H, shard=shard)
        b = rt.PathTracer(gpu_ctx, sc, W, H, shard=shard)
        for n in range(1, 17):
            a.render(view, n)
        n = 1
        for S in (1, 3, 8, 4):
            b.render(view, n, num_samples=S); n += S
        ra, rb = a.read(), b.read()
        assert stats_tuple(ra["stats"]) == stats_tuple(rb["stats"])
        np.testing.assert_array_equal(ra["history"], rb["history"])
        np.testing.assert_array_equal(ra["target"], rb["target"])
        assert ra["stats"].camera_rays == 16 * a.rows * W - 16 * _padding_pixels(gpu_ctx, H, shard, a.rows) * W
        a.close(); b.close()
    sc.close()


def _padding_pixels(ctx, H, shard, rows):
    """rows of a shard's local frame that lie beyond the frame (the last 8-row block of a ragged frame)"""
    if shard is None:
        return 0
    blocks = (H + 7) // 8
    mine = [k for k in range(shard.row_block_first, blocks, shard.row_block_stride)]
    real = sum(min(8, H - 8 * k) for k in mine)
    return rows - real


# ------------------------------------------------------------------------------------------------ parity evidence, tightened
def _outliers(g_target, o_target):
    gt = g_target.view(np.float16).astype(np.float32)
    ot = o_target.view(np.float16).astype(np.float32)
    same = (g_target == o_target).all(axis=-1)
    with np.errstate(invalid="ignore"):
        rel = np.abs(gt - ot) / np.maximum(np.abs(ot), 1e-3)
    return ~(same | (rel <= 2.0 ** -7).all(axis=-1))


def _classify(g_target, o_target, g_chain, o_chain, NOT_TRACED):
    """outliers of the colour tolerance, pixels whose path or shadow-query outcomes differ, and the number of vertices of every path"""
    bad = _outliers(g_target, o_target)
    differs = (g_chain != o_chain).any(axis=(-1, -2))
    vertices = (g_chain[..., 0] != NOT_TRACED).sum(axis=-1)
    return bad, differs, vertices


@pytest.mark.parametrize("scene_name,W,H,cap", [("helmet", 512, 512, 1e-4), ("sponza_small", 480, 270, 3e-4)])
def test_pixels_outside_the_tolerance_are_accounted_for(gpu_ctx, rt, sio, ob, request, scene_name, W, H, cap):
    """The colour tolerance of the path-trace parity tests (2^-7 relative on >= 99.5 % of the pixels) leaves 0.5 % unconstrained.
    Measured (tools/diag_outliers.py): 0 of 262,144 helmet pixels and 11 of 129,600 Sponza-stat pixels are outside 2^-7.  They are of
    two kinds: paths that took another triangle or whose shadow query flipped (device vs glibc ulps in sin / cos / sqrt move a ray
    across an edge; visible in the hit chain), and full-length paths over the same triangles whose colour went through three
    BRDF-sampled bounces (normalize(L + V) of nearly opposite vectors, GGX of a near-mirror lobe: ulp differences amplified to
    1-20 %).  So: every outlier has another chain OR at least three path vertices, and the share of outliers is capped at 1e-4 / 3e-4
    instead of 5e-3."""
    packed = request.getfixturevalue(scene_name)
    osc = request.getfixturevalue(scene_name + "_oracle")
    sc = gpu_ctx.create_scene(packed)
    view = sio.FView().reset(packed, aspect=W / H)
    pt = rt.PathTracer(gpu_ctx, sc, W, H)
    pt.render(view, 1, hit_chain=True)
    g = pt.read(); gc = pt.read_chain()
    o = osc.pathtrace(rt.make_pathtrace_desc(view, W, H, 1), num_threads=os.cpu_count(), want_chain=True)
    bad, differs, vertices = _classify(g["target"], o["target"], gc, o["chain"], rt.PathTracer.NOT_TRACED)
    # ray counts: equal, or off by what the pixels with another path can account for (<= 3 bounce rays + 4 x 66 shadow rays each)
    assert g["stats"].camera_rays == o["stats"].camera_rays == W * H
    for a, b in zip(stats_tuple(g["stats"]), stats_tuple(o["stats"])):
        assert abs(a - b) <= 70 * int(differs.sum())
    assert vertices.min() >= 1                                                   # every pixel traced its camera ray
    print("%s %dx%d: %d pixels outside 2^-7 (%.5f %%): %d with another chain, %d same chain with >= 3 vertices; %d pixels with another chain in all"
          % (scene_name, W, H, int(bad.sum()), 100 * bad.mean(), int((bad & differs).sum()), int((bad & ~differs & (vertices >= 3)).sum()), int(differs.sum())))
    assert not (bad & ~differs & (vertices <= 2)).any(), "a short path over the oracle's triangles is outside the colour tolerance"
    assert bad.mean() <= cap, bad.mean()
    assert differs.mean() <= 4 * cap
    pt.close(); sc.close()


def test_bench_configuration_shaded_strips_4k(gpu_ctx, rt, sio, synth, ob, A):
    """The bench's exact configuration -- Sponza-stat with 1024^2 textures at 3840x2160, sky cube from the IBL precompute, 65 lights --
    SHADED, against the oracle: three 8-row strips of the frame (30,720 pixels each, rendered through incr_shard so that the sampler
    keys on full-frame pixel indices), ray counts equal up to edge flips, colours within 2^-7 except for flips and compounded deep paths (see above)."""
    from conftest import mod
    W, H = 3840, 2160
    packed = synth.sponza_stat(texture_size=1024)
    sc = gpu_ctx.create_scene(packed)
    ibl = mod("ibl")
    sky = ibl.SkyIbl(gpu_ctx, sc)
    gpu_ctx.synchronize()
    envmap = (sky.info.envmap_r11g11b10, sky.info.cubemap_res, sky.info.num_mips)
    host_env = np.ascontiguousarray(sky.download()["envmap"])
    osc = ob.OracleScene(packed)
    view = sio.FView().reset(packed, aspect=W / H)
    blocks = (H + 7) // 8
    total_bad = total = 0
    for block in (40, 135, 230):                       # upper arcade, nave, floor
        shard = A.Shard(8, block, blocks)              # exactly one 8-row block
        pt = rt.PathTracer(gpu_ctx, sc, W, H, shard=shard)
        assert pt.rows == 8
        pt.render(view, 1, envmap=envmap, hit_chain=True)
        g = pt.read(); gc = pt.read_chain()
        d = rt.make_pathtrace_desc(view, W, H, 1, shard, envmap=(host_env.ctypes.data, sky.info.cubemap_res, sky.info.num_mips))
        o = osc.pathtrace(d, num_threads=os.cpu_count(), want_chain=True)
        assert g["stats"].light_shadow_rays > 0 and g["stats"].shaded_hits > 8 * W // 2
        bad, differs, vertices = _classify(g["target"], o["target"], gc, o["chain"], rt.PathTracer.NOT_TRACED)
        assert g["stats"].camera_rays == o["stats"].camera_rays == 8 * W
        for a, b in zip(stats_tuple(g["stats"]), stats_tuple(o["stats"])):
            assert abs(a - b) <= 70 * int(differs.sum())
        assert not (bad & ~differs & (vertices <= 2)).any()
        total_bad += int(bad.sum()); total += bad.size
        pt.close()
    print("4K bench configuration, 3 strips: %d of %d pixels outside 2^-7" % (total_bad, total))
    assert total_bad <= 5e-4 * total
    sky.close(); osc.close(); sc.close()


def test_config4_64_samples_accumulation_vs_oracle(gpu_ctx, rt, sio, sponza_small, sponza_small_oracle):
    """BASELINE config 4 in small: 64 samples (sampleIdx 1 .. 64 of MaxSampleCount 256, Q4) accumulated into the R11G11B10 history, eight
    samples in flight per call, against 64 oracle passes at 96x54.  A pixel's running mean can differ where one of its 64 paths
    flipped an edge; the decoded means must stay within a 6-bit mantissa step of each other almost everywhere."""
    W, H = 96, 54
    sc = gpu_ctx.create_scene(sponza_small)
    view = sio.FView().reset(sponza_small, aspect=W / H)
    pt = rt.PathTracer(gpu_ctx, sc, W, H)
    for n in range(1, 65, 8):
        pt.render(view, n, num_samples=8)
    g = pt.read()
    hist = None
    rays = 0
    for n in range(1, 65):
        o = sponza_small_oracle.pathtrace(rt.make_pathtrace_desc(view, W, H, n), history=hist, num_threads=os.cpu_count())
        hist = o["history"]; rays += o["stats"].total_rays()
    assert abs(g["stats"].total_rays() - rays) <= 1e-4 * rays
    same = (g["history"] == hist)
    from oracle import binding
    gh, oh = binding.unpack_r11g11b10(g["history"]), binding.unpack_r11g11b10(hist)
    with np.errstate(invalid="ignore"):
        rel = np.abs(gh - oh) / np.maximum(np.abs(oh), 1e-3)
    close = (rel <= 2.0 ** -4).all(axis=-1)
    print("64-sample history: %.3f %% bit-identical, %.3f %% within one 5-bit mantissa step" % (100 * same.mean(), 100 * close.mean()))
    assert same.mean() >= 0.99 and close.mean() >= 0.999
    pt.close(); sc.close()


# ------------------------------------------------------------------------------------------------ top-tree rebuild (UpdateTLASPass)
def test_update_bvh_rebuilds_the_top_tree_for_scattered_instances(gpu_ctx, rt, sio, synth):
    """update-tlas.inl:24-89 rebuilds the TLAS from the meshes' current transforms every frame.  Here: incr_update_bvh re-transforms
    and refits the meshes' subtrees on the device and REBUILDS the top tree over their new boxes.  256 instances are scattered far
    from where the scene was built (a refit alone would leave a top tree of scene-sized boxes): hits equal a scene created with the new
    transforms bit for bit, the tree is as deep, and a path-traced frame costs the same (ray counts equal, time within 10 %)."""
    import copy, time
    s = synth.bistro_scale(num_instances=256, tris_per_mesh=700, texture_size=16)
    sc = gpu_ctx.create_scene(s)
    rng = np.random.default_rng(11)
    t2 = []
    for i, m in enumerate(s.mesh_transforms):
        if i == 0:
            t2.append(m.copy()); continue                       # (mesh 0 = the ground)
        ang = rng.uniform(0, 2 * np.pi)
        rot = sio.mat_from_quaternion(0.0, np.sin(ang / 2), 0.0, np.cos(ang / 2))
        t2.append(sio.mat_mul(sio.mat_mul(m, rot), sio.mat_translation(*rng.uniform(-60, 60, 3) * np.array([1, 0.1, 1]))))
    sc.update_bvh(t2)
    s2 = copy.copy(s); s2.mesh_transforms = t2; s2._desc = None
    fresh = gpu_ctx.create_scene(s2)
    assert abs(int(sc.info().bvh_max_depth) - int(fresh.info().bvh_max_depth)) <= 1
    n = 200_000
    org = (rng.uniform(-70, 70, (n, 3)) * np.array([1, 0.1, 1]) + np.array([0, 2.0, 0])).astype(np.float32)
    d = rng.standard_normal((n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    ui, ut = sc.trace_rays(org, d, 0.001, 10000.0); fi, ft = fresh.trace_rays(org, d, 0.001, 10000.0)
    np.testing.assert_array_equal(ui, fi); np.testing.assert_array_equal(ut.view(np.uint32), ft.view(np.uint32))
    ua, _ = sc.trace_rays(org, d, 0.1, 1000.0, any_hit=True); fa, _ = fresh.trace_rays(org, d, 0.1, 1000.0, any_hit=True)
    np.testing.assert_array_equal(ua[:, 0], fa[:, 0])
    assert 0.05 < ui[:, 0].mean() < 0.99 and len(np.unique(ui[ui[:, 0] == 1, 1])) > 100
    W, H = 1920, 1080
    view = sio.FView().reset(s2, aspect=W / H)
    res = {}
    for name, scene in (("updated", sc), ("fresh", fresh)):
        pt = rt.PathTracer(gpu_ctx, scene, W, H)
        pt.render(view, 1); gpu_ctx.synchronize()
        best = None
        for rep in range(3):                    # best of three rounds of six frames: a wall clock on a shared box hiccups
            t0 = time.perf_counter()
            for k in range(2, 8):
                pt.render(view, k + 6 * rep, count_rays=False)
            gpu_ctx.synchronize()
            dt = (time.perf_counter() - t0) / 6
            best = dt if best is None else min(best, dt)
        res[name] = (best, pt.read())
        pt.close()
    (tu, u), (tf, f) = res["updated"], res["fresh"]
    assert stats_tuple(u["stats"]) == stats_tuple(f["stats"])
    np.testing.assert_array_equal(u["history"], f["history"])
    print("scattered instances: %.3f ms per frame after incr_update_bvh, %.3f ms on a fresh scene" % (tu * 1e3, tf * 1e3))
    assert tu <= 1.15 * tf
    sc.close(); fresh.close()


# ------------------------------------------------------------------------------------------------ generated tangents
def test_helmet_without_tangents_renders_like_the_oracle(gpu_ctx, rt, sio, ob):
    """MeshUtils::FixupMeshes path end to end: the helmet loaded with its TANGENT accessor ignored (MikkTSpace tangents generated by
    the library), path traced on the GPU and by the oracle."""
    from conftest import ROOT
    packed = sio.load_gltf(os.path.join(ROOT, "assets/damaged-helmet/DamagedHelmet.gltf"), ignore_tangents=True)
    W = H = 256
    view = sio.FView().reset(packed, aspect=1.0)
    sc = gpu_ctx.create_scene(packed)
    pt = rt.PathTracer(gpu_ctx, sc, W, H)
    pt.render(view, 1)
    g = pt.read()
    osc = ob.OracleScene(packed)
    o = osc.pathtrace(rt.make_pathtrace_desc(view, W, H, 1), num_threads=os.cpu_count())
    assert stats_tuple(g["stats"]) == stats_tuple(o["stats"])
    ok, same = pixel_agreement(g["target"], o["target"])
    assert ok >= 0.9995 and same >= 0.98, (ok, same)
    osc.close(); pt.close(); sc.close()


# ------------------------------------------------------------------------------------------------ typed-UAV codec
def test_small_float_codec_matches_the_generic_one(gpu_ctx, rt):
    """The RGBA16F / R11G11B10F stores and loads run on v_cvt_pkrtz_f16_f32 / v_cvt_f32_f16 (pt_device.h INCR_FAST_SMALL_FLOAT).  Against the
    bit-level functions -- the ones the oracle restates (D3D11.3 3.2.3.3 round-toward-zero stores) -- over EVERY float32 pattern and every
    storage pattern: no mismatch."""
    out = (C.c_uint64 * 4)()
    rt.check(gpu_ctx.lib.incr_selftest_small_float(gpu_ctx.handle, out))
    assert list(out) == [0, 0, 0, 0], list(out)


@pytest.mark.parametrize("ranks,res", [(2, "640x360"), (3, "640x368")])
def test_bench_ranks_rehearsal_on_one_gpu(ranks, res):
    """`python bench.py --gpus N` end to end with render work: the N > 1 branch of bench.py (its own rank processes, interleaved
    8-row shards, the history gather + incr_unshard, 8 samples in flight, the raster row bands + the TAA band gather, the
    max-over-ranks clock, the one JSON line) on the one GPU there is -- all ranks on GPU 0, collectives over gloo (--share-gpu).
    Three ranks over 368 rows: 46 row blocks and 123-row raster bands, neither a multiple of the rank count (ragged shards, the padded
    band gather).  The RCCL transport itself needs a second GPU; everything above it runs here."""
    import json, os, subprocess, sys
    from conftest import ROOT
    W, H = (int(v) for v in res.split("x"))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--backend", "gloo", "--share-gpu", "--steps", "10", "--warmup", "1",
           "--res", res, "--texture-size", "64", "--no-cpu-baseline", "--spp-per-gather", "4"]
    env = dict(os.environ); env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == ranks and d["steps"] == 10 and d["value"] > 0 and d["scaling"] == "strong"
    assert d["config"]["samples_in_flight"] == 8 and "rehearsal" in d["config"] and "RCCL" in d["config"]["sharding"]
    # every rank's rays are in the total: the frame sends W x H camera rays per sample, whatever the split
    assert d["config"]["rays_per_step"] > 2 * W * H
    ra = d["raster"]
    assert "error" not in ra and ra["n_gpus"] == ranks and ra["ms_per_frame"] > 0 and "row bands" in ra["sharding"]
