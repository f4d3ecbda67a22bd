"""Round 4, GPU side: the RCCL branch of bench.py at world size 1, raster ids at the bench configuration's real size, fidelity items."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _bench(*extra, res="640x360", steps=6, spg=4):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", str(steps), "--warmup", "1", "--res", res, "--texture-size", "64",
           "--no-cpu-baseline", "--checksums", "--spp-per-gather", str(spg)] + list(extra)
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_force_collectives_runs_the_rccl_branch_bit_identically():
    """`bench.py --gpus 1 --force-collectives`: the transport branch of the N > 1 path (_Collectives(staged=False)) on the one GPU there is --
    init_process_group("nccl", device_id), all_gather_into_tensor of the R11G11B10 history shard and incr_unshard after every 4th sample and
    after the last, the TAA band gather on the library-owned surface through __cuda_array_interface__, all_reduce of the clock and the ray
    counts, destroy_process_group.  The gathered history and the TAA accumulation surface equal the plain run's bit for bit: RCCL's stream
    is ordered against the caller's stream and the library's pipeline streams (fork / join inside incr_pathtrace) without a host sync."""
    plain = _bench()
    coll = _bench("--force-collectives")
    assert "force_collectives" in coll["config"] and "RCCL" in coll["config"]["sharding"] and "row bands" in coll["raster"]["sharding"]
    assert plain["config"]["sharding"] == "none"
    assert coll["config"]["history_crc32"] == plain["config"]["history_crc32"]
    assert coll["raster"]["taa_accumulation_crc32"] == plain["raster"]["taa_accumulation_crc32"]
    assert coll["config"]["rays_per_step"] == plain["config"]["rays_per_step"]
    assert coll["n_gpus"] == 1 and coll["value"] > 0


def test_bench_force_collectives_at_4k_reports_the_gather():
    """The exchange step at BASELINE config 4's size: one RCCL all_gather of the 33 MB R11G11B10F history + incr_unshard per gathered frame, timed with HIP events on
    the render stream and reported as config.gather.gather_ms; the gathered 4K history equals the plain run's bit for bit."""
    plain = _bench("--no-raster", res="3840x2160", steps=8, spg=8)
    coll = _bench("--no-raster", "--force-collectives", res="3840x2160", steps=8, spg=8)
    assert coll["config"]["history_crc32"] == plain["config"]["history_crc32"] and coll["config"]["rays_per_step"] == plain["config"]["rays_per_step"]
    g = coll["config"]["gather"]
    assert g["gathers"] == 1 and g["bytes_gathered"] == 3840 * 2160 * 4 and 0.0 < g["gather_ms"] < 5.0, g
    assert "gather" not in plain["config"]
    print("4K gather: %.3f ms for %.1f MB" % (g["gather_ms"], g["bytes_gathered"] / 1e6))


def test_eight_shards_of_the_4k_frame(gpu_ctx, rt, sio, synth, A):
    """tools/shard_emulation.py as a test: the 4K frame rendered as the 8 framebuffer shards of an 8-GPU run (incr_shard{8, r, 8}) -- every shard's history and target
    rows are the full frame's rows bit for bit, and with 8 samples in flight a shard takes less than a sixth of the full frame's time per sample (the compute-only
    projection of the north star's >= 6x at 8 GPUs; measured x7.8).  One GPU: the shards run one after the other."""
    import ctypes as C
    import time
    W, H, S = 3840, 2160, 8
    scene = synth.sponza_stat(texture_size=64)
    view = sio.FView().reset(scene, aspect=W / H)
    sc = gpu_ctx.create_scene(scene)
    try:
        def run(shard, samples, steps):
            pt = rt.PathTracer(gpu_ctx, sc, W, H, shard=shard)
            pt.render(view, 1, num_samples=samples, count_rays=False)          # sample 1 (.. S): what is compared, and the warm-up of the timed calls
            first = pt.read()
            gpu_ctx.synchronize()
            t0 = time.perf_counter()
            n = 1 + samples
            for _ in range(steps):
                pt.render(view, n, num_samples=samples, count_rays=False); n += samples
            gpu_ctx.synchronize()
            ms = (time.perf_counter() - t0) * 1e3 / (steps * samples)
            pt.close()
            return first, ms
        full1, _ = run(None, 1, 1)
        fullS, full_ms = run(None, 1, 6)                     # the full frame at one sample per call: what one GPU does
        worst = 0.0
        for r in range(8):
            rows = np.concatenate([np.arange(b * 8, min(b * 8 + 8, H)) for b in range(r, (H + 7) // 8, 8)])
            one, _ = run(A.Shard(8, r, 8), 1, 1)
            assert one["history"].shape[0] == 272 and len(rows) in (264, 272)          # every rank holds 34 blocks; the last one of ranks 6 and 7 is padding (cleared)
            np.testing.assert_array_equal(one["history"][:len(rows)], full1["history"][rows], err_msg="shard %d history" % r)
            np.testing.assert_array_equal(one["target"][:len(rows)], full1["target"][rows], err_msg="shard %d target" % r)
            assert not one["target"][len(rows):].any()
            _, ms = min((run(A.Shard(8, r, 8), S, 3) for _ in range(2)), key=lambda t: t[1])
            worst = max(worst, ms)
        print("full frame %.2f ms per sample, slowest 1/8 shard with %d samples in flight %.2f ms per sample: projected x%.2f" % (full_ms, S, worst, full_ms / worst))
        assert full_ms / worst >= 6.0, (full_ms, worst)
    finally:
        sc.close()


# ------------------------------------------------------------------------------------------------ raster ids at config 3's real size
def test_raster_4k_bench_configuration_bands_vs_oracle(gpu_ctx, rt, R, sio, synth, ob, A):
    """BASELINE config 3 at its real size -- full Sponza-stat (262,267 triangles, 1024^2 textures, the x16 anisotropic sampler, 65 lights,
    the sky's SH9 / prefiltered cube / env-BRDF LUT from the IBL precompute) at 3840x2160 -- against oracle_raster.h: the GPU renders the whole
    frame pass by pass, the oracle the same passes on three scissor bands of 16 rows (upper arcade, nave, floor; every raster pass of the
    oracle honours incr_view_constants::scissor_y0/y1).  Visibility ids (the north star's one exact requirement: visibility-pass.hlsl:96-114,
    encoding.hlsli:4-28), depth and the three G-buffers must be bit-identical on the bands, the lit colour within colour_close, the draw lists
    and the light grid identical."""
    from conftest import mod
    from test_gpu_raster import colour_close
    W, H = 3840, 2160
    scene = synth.sponza_stat(texture_size=1024)
    scene = R.generate_meshlets(scene)
    assert sum(p["index_count"] // 3 for prims in scene.mesh_primitives for p in prims) == 262267 and len(scene.global_lights) == 65
    view = sio.FView().reset(scene, aspect=W / H)
    gscene = gpu_ctx.create_scene(scene)
    osc = ob.OracleScene(scene)
    ibl = mod("ibl")
    LW = LH = 512
    try:
        sky = ibl.SkyIbl(gpu_ctx, gscene)
        lut = ibl.generate_env_brdf(gpu_ctx, LW, LH)
        rr = R.RasterRenderer(gpu_ctx, gscene, W, H, A.FConfig)
        rr.ibl = R.IblInputs.from_sky(sky, lut, (LW, LH))
        vc = rr.render(view); gpu_ctx.synchronize()
        g = {n: rr.surf.download(n) for n in ("vis", "depth", "gbuffer_basecolor", "gbuffer_normals", "gbuffer_mra", "color", "counts", "default_list", "double_sided_list",
                                              "light_grid", "light_lists", "light_count")}
        prod = sky.download()
        lut_host = gpu_ctx.download(lut, (LH, LW), np.uint32)
        env_host, sh_host = np.ascontiguousarray(prod["envmap"]), np.ascontiguousarray(prod["sh9"])
        ibl_host = R.IblInputs(sh_host.ctypes.data, env_host.ctypes.data, sky.info.cubemap_res, sky.info.num_mips, lut_host.ctypes.data, (LW, LH))
        hs = ob.HostSurfaces(W, H, scene.num_meshlets, 16 * 9 * 24, 64)
        assert (g["vis"] != 0xFFFFF000).mean() > 0.95
        for y0, y1 in ((320, 336), (1080, 1096), (1840, 1856)):
            for name, d in R.build_descs(None, scene, R.with_scissor(vc, y0, y1), hs.ptr, 0, A.FConfig, ibl=ibl_host):
                if name != "taa_resolve":
                    ob.run_pass(osc, name, d)
            o = hs.arr
            band = slice(y0, y1)
            np.testing.assert_array_equal(g["counts"], o["counts"])
            for name, n in (("default_list", o["counts"][0]), ("double_sided_list", o["counts"][1])):
                np.testing.assert_array_equal(g[name][:n], o[name][:n])
            np.testing.assert_array_equal(g["light_grid"], o["light_grid"])
            np.testing.assert_array_equal(g["vis"][band], o["vis"][band], err_msg="visibility ids rows %d-%d" % (y0, y1))
            np.testing.assert_array_equal(g["depth"][band].view(np.uint32), o["depth"][band].view(np.uint32), err_msg="depth rows %d-%d" % (y0, y1))
            for name in ("gbuffer_basecolor", "gbuffer_normals", "gbuffer_mra"):
                np.testing.assert_array_equal(g[name][band], o[name][band], err_msg="%s rows %d-%d" % (name, y0, y1))
            frac = colour_close(ob, g["color"][band], o["color"][band], "lit colour rows %d-%d" % (y0, y1))
            assert frac > 0.995, frac
            ids = np.unique(g["vis"][band] >> 7)
            print("rows %d-%d: %d distinct meshlets, %.1f %% covered, colour bit-identical on %.4f of the pixels" % (y0, y1, len(ids), 100 * (g["vis"][band] != 0xFFFFF000).mean(), frac))
            assert len(ids) > 20
        rr.close(); sky.close(); gpu_ctx.free(lut)
    finally:
        osc.close(); gscene.close()


# ------------------------------------------------------------------------------------------------ path-tracer view modes 6-13
def test_pathtrace_viewmode_7_mirror_bounce_vs_oracle(gpu_ctx, rt, sio, A, sponza_small, sponza_small_oracle, helmet, helmet_oracle):
    """VIEWMODE 7, Reflections (pathtracing.hlsl:212-224): the first hit sends one mirror ray normalize(reflect(-V, N)) with the same payload and
    returns -- no light loop, no sky sample, no SamplePoint at that vertex; from the second hit on the path is the normal one.  Against the
    oracle with every outlier accounted for; the first vertex requests no shadow ray (its shadow word is 0) and every first hit continues."""
    from parity_helpers import accounted_parity, NOT_TRACED

    class Reflections(A.FConfig):
        Viewmode = 7
    for scene, osc, W, H, cap in ((sponza_small, sponza_small_oracle, 320, 180, 4e-4), (helmet, helmet_oracle, 256, 256, 2e-4)):
        view = sio.FView().reset(scene, aspect=W / H)
        sc = gpu_ctx.create_scene(scene)
        pt = rt.PathTracer(gpu_ctx, sc, W, H)
        g, o, _ = accounted_parity(rt, pt, osc, view, W, H, 3, cap=cap, label="viewmode 7", cfg=Reflections, min_outliers_allowed=3)
        chain = pt.read_chain()
        hit0 = (chain[..., 0, 0] != 0xffffffff) & (chain[..., 0, 0] != NOT_TRACED)
        assert hit0.mean() > 0.3
        assert (chain[..., 0, 2][hit0] == 0).all()                                  # no shadow query at the mirror vertex
        assert (chain[..., 1, 0][hit0] != NOT_TRACED).all()                         # the mirror ray is always traced
        assert g["stats"].bounce_rays >= int(hit0.sum())
        # the normal mode's image differs (lights and sky at the first vertex)
        pt.render(view, 3); n = pt.read()
        assert not np.array_equal(n["target"], g["target"])
        pt.close(); sc.close()


@pytest.mark.parametrize("viewmode", [6, 8, 10, 13])
def test_pathtrace_viewmodes_without_a_shader_branch_are_the_normal_path(gpu_ctx, rt, sio, A, sponza_small, viewmode):
    """pathtracing.hlsl names VIEWMODE 1-5 and 7 only: compiled with 6 (NaN check, a tonemap-side view) or 8-13 (raster debug views) it is
    the normal path tracer (path-tracing.inl:31-35 passes any FConfig::Viewmode).  Bit-identical target, history and ray counts."""
    from parity_helpers import stats_tuple
    W, H = 200, 112
    view = sio.FView().reset(sponza_small, aspect=W / H)
    sc = gpu_ctx.create_scene(sponza_small)

    class Cfg(A.FConfig):
        Viewmode = viewmode
    a = rt.PathTracer(gpu_ctx, sc, W, H); b = rt.PathTracer(gpu_ctx, sc, W, H)
    for n in (1, 2):
        a.render(view, n); b.render(view, n, cfg=Cfg)
    ra, rb = a.read(), b.read()
    assert stats_tuple(ra["stats"]) == stats_tuple(rb["stats"]) and ra["stats"].light_shadow_rays > 0
    np.testing.assert_array_equal(ra["target"], rb["target"]); np.testing.assert_array_equal(ra["history"], rb["history"])
    with pytest.raises(rt.IncrError):
        b.render(view, 1, cfg=type("Bad", (A.FConfig,), {"Viewmode": 14}))
    a.close(); b.close(); sc.close()


# ------------------------------------------------------------------------------------------------ primitive path beyond 2,047 primitives
def test_primitive_path_uses_all_12_bits_of_primitive_id(gpu_ctx, rt, R, sio, synth, ob, A):
    """EncodePrimitiveVisibility keeps 12 bits of primitive id (encoding.hlsli:4-11).  Rounds 1-3 spent one of them on the draw-list order
    of equal depths; the list bit now lives beside the depth in the 64-bit depth || list || id word.  A scene of 3,601 primitives (one per mesh)
    through FConfig::UseMeshlets = false: ids, depth and G-buffers bit-exact against the oracle, ids above 2,047 on screen; 4,097 are refused."""
    from test_gpu_raster import gpu_frame
    from test_raster_host_and_oracle import run_oracle_frame

    class Prims(A.FConfig):
        UseMeshlets = False
    scene = synth.bistro_scale(num_instances=3600, tris_per_mesh=64, num_materials=12, texture_size=8)
    assert sum(len(p) for p in scene.mesh_primitives) == 3601
    scene = R.generate_meshlets(scene)               # sizes the draw lists (one entry per meshlet >= one per primitive)
    W, H = 384, 216
    view = sio.FView()                                  # above the street, looking down the grid: thousands of props on screen
    view.position = np.array([0, 45, -110], dtype=np.float32); view.look = np.array([0, -0.45, 1], dtype=np.float32); view.right = np.array([1, 0, 0], dtype=np.float32)
    view.update_view_transform()
    view.projection_transform = sio.reverse_z_infinite_perspective_fov_lh(0.8, W / H, A.FConfig.CameraNearPlane)
    gscene = gpu_ctx.create_scene(scene)
    osc = ob.OracleScene(scene)
    try:
        (vc, g), = gpu_frame(rt, R, gpu_ctx, gscene, view, W, H, Prims, frames=1)
        _, hs = run_oracle_frame(ob, R, A, scene, osc, view, W, H, cfg=Prims)
        o = hs.arr
        np.testing.assert_array_equal(g["counts"], o["counts"])
        np.testing.assert_array_equal(g["vis"], o["vis"])
        np.testing.assert_array_equal(g["depth"].view(np.uint32), o["depth"].view(np.uint32))
        for name in ("gbuffer_basecolor", "gbuffer_normals", "gbuffer_mra"):
            np.testing.assert_array_equal(g[name], o[name], err_msg=name)
        covered = g["vis"] != 0xFFFFF000
        prim = g["vis"][covered] >> 20
        print("primitive ids on screen: %d distinct, max %d, %.1f %% of the covered pixels above 2047" % (len(np.unique(prim)), prim.max(), 100 * (prim > 2047).mean()))
        assert prim.max() == 3600 and len(np.unique(prim)) > 2000 and ((prim > 2047) & (prim < 3600)).mean() > 0.3
    finally:
        osc.close(); gscene.close()
    big = synth.bistro_scale(num_instances=4096, tris_per_mesh=8, num_materials=4, texture_size=8)          # 4,097 primitives
    gs2 = gpu_ctx.create_scene(big)
    try:
        rr = R.RasterRenderer(gpu_ctx, gs2, 64, 36, Prims)
        with pytest.raises(rt.IncrError):
            rr.render(sio.FView().reset(big, aspect=64 / 36))
        rr.close()
    finally:
        gs2.close()


# ------------------------------------------------------------------------------------------------ bounce overlap
@pytest.mark.parametrize("scene_name,W,H,pipes", [("sponza_small", 320, 180, 1), ("sponza_small", 203, 117, 2), ("helmet", 256, 256, 4)])
def test_bounce_overlap_is_bit_identical(gpu_ctx, rt, sio, A, request, scene_name, W, H, pipes):
    """incr_set_pathtrace_overlap: the any-hit launches on the pipeline's side stream beside the main chain primary -> shade -> extend -> ..., the four summation
    launches at the end of the main stream or each behind the any-hit launch of its bounce (incr_set_pathtrace_accumulate_placement 0 / 1: a 4K frame takes 0 by
    size, these small frames 1 -- both are forced here).  Same kernels on the same inputs in another issue order: target, history, ray counts and the hit chains
    equal the sequential issue bit for bit, over several samples (the side streams are reused across calls), with samples in flight and on a framebuffer shard."""
    from parity_helpers import stats_tuple
    packed = request.getfixturevalue(scene_name)
    sc = gpu_ctx.create_scene(packed)
    view = sio.FView().reset(packed, aspect=W / H)
    gpu_ctx.set_pathtrace_pipelines(pipes)
    try:
        res = []
        for overlap, placement in ((False, -1), (True, 0), (True, 1)):
            gpu_ctx.set_pathtrace_overlap(overlap); gpu_ctx.set_pathtrace_accumulate_placement(placement)
            out = []
            for shard in (None, A.Shard(8, 1, 3)):
                pt = rt.PathTracer(gpu_ctx, sc, W, H, shard=shard)
                pt.render(view, 1, hit_chain=True)
                pt.render(view, 2, num_samples=3)
                pt.render(view, 5)
                r = pt.read(); r["chain"] = pt.read_chain()
                out.append(r); pt.close()
            res.append(out)
        for other in res[1:]:
            for a, b in zip(res[0], other):
                assert stats_tuple(a["stats"]) == stats_tuple(b["stats"]) and a["stats"].bounce_rays > 0
                for k in ("target", "history", "chain"):
                    np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    finally:
        gpu_ctx.set_pathtrace_pipelines(0); gpu_ctx.set_pathtrace_overlap(True); gpu_ctx.set_pathtrace_accumulate_placement(-1)
        sc.close()


# ------------------------------------------------------------------------------------------------ pinned shader division
def test_device_reciprocal_is_the_correctly_rounded_one(gpu_ctx):
    """Shader divisions are a * rcp(b) with rcp correctly rounded (oracle: 1.f / b; device: v_rcp_f32 + two fused Newton steps + v_div_fixup,
    pt_device.h rcp_cr).  Over all 2^32 float patterns the device reciprocal must equal the IEEE quotient except on a vanishing set (Markstein's
    sequence can miss by one ulp when the mantissa of x is all ones, and where the quotient is denormal), never by more than one ulp."""
    import ctypes as C
    out = (C.c_uint64 * 3)()
    gpu_ctx.lib.incr_selftest_reciprocal.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    assert gpu_ctx.lib.incr_selftest_reciprocal(gpu_ctx.handle, out) == 0
    print("rcp_cr vs 1.f / x: %d of the patterns with x and 1 / x normal differ, %d of them by more than one ulp; %d differ outside that domain" % (out[0], out[1], out[2]))
    assert out[1] == 0
    assert out[0] <= 2 ** 32 * 1e-5


# ------------------------------------------------------------------------------------------------ interleaved material chains (raster)
@pytest.mark.gpu
@pytest.mark.parametrize("scene_name,hook", [("sponza_small", "INCR_MATERIAL_CHAINS"), ("helmet", "INCR_MATERIAL_CHAINS"), ("sponza_small", "INCR_ALPHA_QUADS")])
def test_interleaved_material_chains_are_bit_identical(gpu_ctx, R, sio, A, request, scene_name, hook):
    """The raster passes' anisotropic Sample of a material's base-colour, normal and metallic-roughness maps out of the interleaved chain
    (device_types.h MaterialTexels::chain: four 8-byte loads per tap and level) must give the bytes of the three-texture path: G-buffers,
    lit colour and TAA history of three frames, with the scene built once with and once without the chains (INCR_MATERIAL_CHAINS=0).
    Likewise the visibility scan's alpha test out of the alpha quads (texture_mips.hip alpha_quads_kernel: one word per tap and level,
    INCR_ALPHA_QUADS=0 samples the colour texture): ids and everything behind them."""
    packed = request.getfixturevalue(scene_name)          # (the session fixtures carry their meshlets: regenerating them here would pull the arrays from under the session's oracle scenes)
    W, H = 384, 216
    view = sio.FView().reset(packed, aspect=W / H)
    view.position = (view.position + 0.35 * view.right).astype(np.float32); view.update_view_transform()          # off-axis: anisotropic footprints, several levels
    out = []
    for chains in ("1", "0"):
        old = os.environ.get(hook)
        os.environ[hook] = chains
        try:
            sc = gpu_ctx.create_scene(packed)
        finally:
            if old is None: os.environ.pop(hook, None)
            else: os.environ[hook] = old
        hr = R.HostRenderer(gpu_ctx, sc, W, H)
        for _ in range(3): hr.render(view)
        gpu_ctx.synchronize()
        out.append({n: hr.download(w).copy() for n, w in (("basecolor", A.SURFACE_GBUFFER_BASECOLOR), ("normals", A.SURFACE_GBUFFER_NORMALS), ("mra", A.SURFACE_GBUFFER_MRA),
                                                        ("hdr", A.SURFACE_HDR_RASTER), ("taa", A.SURFACE_TAA_ACCUMULATION), ("vis", A.SURFACE_VISIBILITY))})
        hr.close(); sc.close()
    assert (out[0]["vis"] != 0xFFFFF000).mean() > 0.2
    for k in out[0]:
        np.testing.assert_array_equal(out[0][k], out[1][k], err_msg=k)
