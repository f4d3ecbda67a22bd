"""GPU parity tests (run with -m gpu on an MI355X): the HIP path through the C ABI vs the CPU oracle on the
same seeded inputs, the committed golden fixtures, and size-independent properties at the BASELINE sizes.

Tolerances (SURVEY.md 8d "Image quality"): integer stages, ray counts, primary-hit identity and the hit
parameters t/u/v are bit-exact; the RGBA16F 1-spp output must agree within 2^-7 relative per channel
on at least 99.5 % of the pixels (differences come only from ulp-level sin/cos/exp/acos differences between
the device math library and glibc, amplified at geometric discontinuities)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

REL_TOL = 2.0 ** -7


def pixel_agreement(gpu_target, cpu_target):
    gt = gpu_target.view(np.float16).astype(np.float32)
    ot = cpu_target.view(np.float16).astype(np.float32)
    same = (gpu_target == cpu_target).all(axis=-1)
    with np.errstate(invalid="ignore"):
        rel = np.abs(gt - ot) / np.maximum(np.abs(ot), 1e-3)
    ok = same | (rel <= REL_TOL).all(axis=-1)
    return float(ok.mean()), float(same.mean())


def stats_tuple(st):
    return (st.camera_rays, st.bounce_rays, st.sky_shadow_rays, st.light_shadow_rays, st.shaded_hits)


@pytest.fixture(scope="module")
def helmet_gpu(gpu_ctx, helmet):
    sc = gpu_ctx.create_scene(helmet)
    yield sc
    sc.close()


@pytest.fixture(scope="module")
def sponza_small_gpu(gpu_ctx, sponza_small):
    sc = gpu_ctx.create_scene(sponza_small)
    yield sc
    sc.close()


def test_native_library_is_the_one_running(rt, gpu_ctx):
    assert rt.LIB_PATH.endswith("incarnation-renderer_amd/csrc/libincr_hip.so") and os.path.exists(rt.LIB_PATH)
    assert rt.load().incr_device_count() >= 1
    with open("/proc/self/maps") as f:
        assert "libincr_hip.so" in f.read()


def test_primary_hits_bit_exact_vs_golden_and_oracle(gpu_ctx, helmet_gpu, helmet_oracle):
    g = np.load(os.path.join(GOLDEN, "helmet_64.npz"))
    ids, tuv = helmet_gpu.trace_rays(g["origins"], g["dirs"], 0.001, 10000.0)
    np.testing.assert_array_equal(ids, g["ids"])                                   # hit flag, InstanceIndex, GeometryIndex, PrimitiveIndex
    np.testing.assert_array_equal(tuv.view(np.uint32), g["tuv"].view(np.uint32))   # t, u, v bit for bit
    # 200k incoherent rays from inside the scene bounds, closest hit and any hit
    rng = np.random.default_rng(1234)
    n = 200_000
    o = rng.uniform(-1.5, 1.5, (n, 3)).astype(np.float32)
    d = rng.standard_normal((n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    gi, gt = helmet_gpu.trace_rays(o, d, 0.001, 10000.0)
    oi, ot = helmet_oracle.trace_rays(o, d, 0.001, 10000.0)
    np.testing.assert_array_equal(gi, oi)
    np.testing.assert_array_equal(gt.view(np.uint32), ot.view(np.uint32))
    ga, _ = helmet_gpu.trace_rays(o, d, 0.1, 1000.0, any_hit=True)
    oa, _ = helmet_oracle.trace_rays(o, d, 0.1, 1000.0, any_hit=True)
    np.testing.assert_array_equal(ga[:, 0], oa[:, 0])
    assert 0.05 < gi[:, 0].mean() < 0.95


def test_edge_case_rays(gpu_ctx, helmet_gpu, helmet_oracle):
    """Axis-parallel directions (infinite reciprocal), zero-length interval, rays starting far outside, NaN direction."""
    o = np.array([[0, 0, -5], [0, 0, -5], [0, 5, 0], [100, 100, 100], [0, 0, 0], [0, 0, -5]], dtype=np.float32)
    d = np.array([[0, 0, 1], [0, 0, -1], [0, -1, 0], [-0.57735, -0.57735, -0.57735], [1, 0, 0], [np.nan, 0, 1]], dtype=np.float32)
    gi, gt = helmet_gpu.trace_rays(o, d, 0.001, 10000.0)
    oi, ot = helmet_oracle.trace_rays(o, d, 0.001, 10000.0)
    np.testing.assert_array_equal(gi, oi)
    np.testing.assert_array_equal(gt.view(np.uint32), ot.view(np.uint32))
    assert gi[0, 0] == 1 and gi[1, 0] == 0 and gi[5, 0] == 0
    gi2, _ = helmet_gpu.trace_rays(o, d, 5.0, 5.0)       # empty interval: nothing can be hit
    assert not gi2[:, 0].any()
    empty_ids, empty_tuv = helmet_gpu.trace_rays(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32), 0.0, 1.0)
    assert empty_ids.shape == (0, 4) and empty_tuv.shape == (0, 3)


def test_helmet_1spp_parity_config1(gpu_ctx, rt, sio, helmet, helmet_gpu, helmet_oracle):
    """BASELINE config 1 (Damaged Helmet 512x512 1 spp): HIP vs the CPU scalar reference path tracer."""
    W = H = 512
    view = sio.FView().reset(helmet, aspect=1.0)
    pt = rt.PathTracer(gpu_ctx, helmet_gpu, W, H)
    pt.render(view, 1)
    g = pt.read()
    o = helmet_oracle.pathtrace(rt.make_pathtrace_desc(view, W, H, 1), num_threads=os.cpu_count())
    assert stats_tuple(g["stats"]) == stats_tuple(o["stats"])              # every ray category identical
    ok, same = pixel_agreement(g["target"], o["target"])
    assert ok >= 0.9995 and same >= 0.98, (ok, same)
    assert (g["history"] == o["history"]).mean() >= 0.9995
    pt.close()


def test_golden_64_accumulation(gpu_ctx, rt, sio, helmet, helmet_gpu):
    g = np.load(os.path.join(GOLDEN, "helmet_64.npz"))
    view = sio.FView().reset(helmet, aspect=1.0)
    pt = rt.PathTracer(gpu_ctx, helmet_gpu, 64, 64)
    for n in (1, 2, 3, 4):
        gpu_ctx.clear(pt.stats, C.sizeof(rt.A.RayStats))
        pt.render(view, n)
        r = pt.read()
        if n == 1:
            ok, same = pixel_agreement(r["target"], g["target1"])
            assert ok >= 0.9995
        assert list(stats_tuple(r["stats"])) == list(g["stats"][n - 1])
    # 4-sample R11G11B10 running mean: a pixel can differ where ONE of its 4 samples differed
    assert (r["history"] == g["history4"]).mean() >= 0.98
    pt.close()


@pytest.mark.parametrize("viewmode", [2, 3, 4, 5])
def test_viewmodes_material_channels(gpu_ctx, rt, sio, A, helmet, helmet_gpu, helmet_oracle, viewmode):
    class Cfg(A.FConfig):
        Viewmode = viewmode
    view = sio.FView().reset(helmet, aspect=1.0)
    pt = rt.PathTracer(gpu_ctx, helmet_gpu, 96, 96)
    pt.render(view, 1, cfg=Cfg)
    g = pt.read()
    o = helmet_oracle.pathtrace(rt.make_pathtrace_desc(view, 96, 96, 1, cfg=Cfg))
    # texture taps only (no transcendental): identical except for pixels whose hit differs by an ulp-induced edge flip
    assert (g["target"] == o["target"]).all(axis=-1).mean() >= 0.999
    pt.close()


def test_sponza_small_lights_and_alpha_mask(gpu_ctx, rt, sio, sponza_small, sponza_small_gpu, sponza_small_oracle):
    """65 lights (sun + 64 point, light-loop + shadow rays) and MASK materials (ahsMain alpha test) vs the oracle."""
    W, H = 320, 180
    view = sio.FView().reset(sponza_small, aspect=W / H)
    pt = rt.PathTracer(gpu_ctx, sponza_small_gpu, W, H)
    from parity_helpers import accounted_parity
    # every pixel outside 2^-7 is a path over other triangles / with a flipped shadow query, or a path of >= 3 vertices (parity_helpers.py)
    g, o, _ = accounted_parity(rt, pt, sponza_small_oracle, view, W, H, 5, cap=3e-4, label="sponza_small lights + alpha mask")
    assert g["stats"].camera_rays == W * H
    assert g["stats"].light_shadow_rays > W * H                         # the light loop really ran
    # masked (alpha-tested) rays: closest-hit identity against the oracle on rays aimed through the plants
    rng = np.random.default_rng(99)
    n = 100_000
    org = np.tile(view.position.astype(np.float32), (n, 1))
    d = np.tile(view.look.astype(np.float32), (n, 1)) + rng.uniform(-0.6, 0.6, (n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    gi, gt = sponza_small_gpu.trace_rays(org, d, 0.001, 10000.0)
    oi, ot = sponza_small_oracle.trace_rays(org, d, 0.001, 10000.0)
    np.testing.assert_array_equal(gi, oi)
    np.testing.assert_array_equal(gt.view(np.uint32), ot.view(np.uint32))
    masked_geoms = {0, 1, 20}
    assert np.isin(gi[gi[:, 0] == 1, 2], list(masked_geoms)).any()    # some closest hits are on alpha-tested geometry
    pt.close()


def test_sun_cover_map_answers_shadow_rays_identically(gpu_ctx, rt, sio, sponza_small, monkeypatch):
    """Shadow rays toward the sun whose origin the sun cover map proves clear are queued as null rays (pathtrace_wf.hip shadow_gen_block).
    Accumulation, targets and every ray counter must be what they are without the map."""
    W, H = 320, 180
    view = sio.FView().reset(sponza_small, aspect=W / H)

    def run(cover):
        if cover: monkeypatch.delenv("INCR_SUN_COVER", raising=False)
        else: monkeypatch.setenv("INCR_SUN_COVER", "0")
        gscene = gpu_ctx.create_scene(sponza_small)
        pt = rt.PathTracer(gpu_ctx, gscene, W, H)
        for s in range(1, 4):
            pt.render(view, s)
        g = pt.read()
        pt.close(); gscene.close()
        return g
    a, b = run(True), run(False)
    assert stats_tuple(a["stats"]) == stats_tuple(b["stats"]) and a["stats"].light_shadow_rays > W * H
    np.testing.assert_array_equal(a["history"], b["history"])
    np.testing.assert_array_equal(a["target"], b["target"])


def test_shards_are_bit_identical_to_the_full_frame(gpu_ctx, rt, sio, A, helmet, helmet_gpu):
    """sampler keys use full-frame coordinates, so a shard equals the same rows of the full frame."""
    W, H, h, n = 200, 116, 8, 3                    # 116 rows = 14.5 blocks over 3 ranks: ragged block + padding rows
    view = sio.FView().reset(helmet, aspect=W / H)
    full = rt.PathTracer(gpu_ctx, helmet_gpu, W, H)
    full.render(view, 2)
    f = full.read()
    rows = gpu_ctx.lib.incr_shard_local_rows(H, C.byref(A.Shard(h, 0, n)))
    gathered_ptr = gpu_ctx.buffer(n * rows * W * 4)
    total = 0
    for r in range(n):
        sh = rt.PathTracer(gpu_ctx, helmet_gpu, W, H, shard=A.Shard(h, r, n))
        assert sh.rows == rows
        sh.render(view, 2)
        s = sh.read()
        total += s["stats"].total_rays()
        gpu_ctx.upload(gathered_ptr + r * rows * W * 4, s["history"])
        for lr in range(rows):
            blk, rr = divmod(lr, h)
            y = (r + blk * n) * h + rr
            if y < H:
                np.testing.assert_array_equal(s["target"][lr], f["target"][y])
            else:
                assert not s["target"][lr].any()                       # padding rows stay zero
        sh.close()
    assert total == f["stats"].total_rays()
    out_ptr = gpu_ctx.buffer(H * W * 4)
    gpu_ctx.unshard(out_ptr, gathered_ptr, W, H, h, n, 4)
    gpu_ctx.synchronize()
    np.testing.assert_array_equal(gpu_ctx.download(out_ptr, (H, W), np.uint32), f["history"])
    gpu_ctx.free(out_ptr); gpu_ctx.free(gathered_ptr); full.close()


def test_tonemap_matches_oracle(gpu_ctx, rt, sio, A, ob, helmet, helmet_gpu):
    W = H = 128
    view = sio.FView().reset(helmet, aspect=1.0)
    pt = rt.PathTracer(gpu_ctx, helmet_gpu, W, H)
    pt.render(view, 1)
    g = pt.read()
    ldr = gpu_ctx.buffer(W * H * 4)
    d = A.TonemapDesc(pt.history, ldr, W, H, 13.0, 0)
    gpu_ctx.tonemap(d)
    gpu_ctx.synchronize()
    got = gpu_ctx.download(ldr, (H, W, 4), np.uint8).astype(np.int32)
    want = ob.tonemap(g["history"]).astype(np.int32)
    np.testing.assert_array_equal(got, want)                   # the sRGB encode steps where the oracle's powf makes it step
    gpu_ctx.free(ldr); pt.close()


@pytest.mark.parametrize("viewmode,exposure", [(0, 13.0), (0, 9.0), (0, 16.0), (2, 13.0)])
def test_tonemap_every_texel_value(gpu_ctx, A, ob, viewmode, exposure):
    """Every value an R11G11B10 channel can hold (2048 / 2048 / 1024 patterns incl. denormals, infinities and NaNs), through the ACES curve at three
    exposures and passed through (debug views): the sRGB encode must give the oracle's byte for each."""
    r = np.arange(2048, dtype=np.uint32)
    img = np.stack([r | (r << 11) | ((r >> 1) << 22), r | (((r * 7) & 2047) << 11) | (((r * 5) & 1023) << 22)]).astype(np.uint32)      # 2 x 2048
    hdr = gpu_ctx.buffer(img.nbytes); ldr = gpu_ctx.buffer(img.nbytes)
    gpu_ctx.upload(hdr, img)
    gpu_ctx.tonemap(A.TonemapDesc(hdr, ldr, img.shape[1], img.shape[0], exposure, viewmode))
    gpu_ctx.synchronize()
    got = gpu_ctx.download(ldr, img.shape + (4,), np.uint8)
    np.testing.assert_array_equal(got, ob.tonemap(img, exposure, viewmode))
    gpu_ctx.free(hdr); gpu_ctx.free(ldr)


def test_rejects_bad_arguments(gpu_ctx, rt, sio, A, helmet, helmet_gpu):
    view = sio.FView().reset(helmet, aspect=1.0)
    d = rt.make_pathtrace_desc(view, 64, 64, 1)
    d.scene = None
    assert gpu_ctx.lib.incr_pathtrace(gpu_ctx.handle, C.byref(d), None) == -1
    d.scene = helmet_gpu.handle
    d.env_sky_mode = 0
    assert gpu_ctx.lib.incr_pathtrace(gpu_ctx.handle, C.byref(d), None) == -1
    assert b"env_sky_mode" in gpu_ctx.lib.incr_last_error()
    bad = sio.PackedScene()
    with pytest.raises(rt.IncrError):
        import copy
        s2 = copy.copy(helmet)
        s2.mesh_primitives = [[dict(helmet.mesh_primitives[0][0], material=7)]]
        gpu_ctx.create_scene(s2)


def test_empty_scene_renders_sky_only(gpu_ctx, rt, sio, ob):
    s = sio.PackedScene()                       # no meshes at all: every camera ray misses
    sc = gpu_ctx.create_scene(s)
    view = sio.FView().reset(None, aspect=1.0)
    pt = rt.PathTracer(gpu_ctx, sc, 64, 64)
    pt.render(view, 1)
    g = pt.read()
    assert stats_tuple(g["stats"]) == (64 * 64, 0, 0, 0, 0)
    o = ob.OracleScene(s).pathtrace(rt.make_pathtrace_desc(view, 64, 64, 1))
    ok, same = pixel_agreement(g["target"], o["target"])
    assert ok == 1.0
    pt.close(); sc.close()


def test_full_size_properties_4k(gpu_ctx, rt, sio, synth, A, ob):
    """BASELINE size (3840x2160, full Sponza-stat): properties that do not need the oracle at full size."""
    W, H = 3840, 2160
    s = synth.sponza_stat(texture_size=256)
    sc = gpu_ctx.create_scene(s)
    assert sc.info().num_triangles == 262267
    view = sio.FView().reset(s, aspect=W / H)
    pt = rt.PathTracer(gpu_ctx, sc, W, H)
    pt.render(view, 1); a = pt.read()
    assert a["stats"].camera_rays == W * H                                  # one camera ray per pixel
    assert a["stats"].bounce_rays <= 3 * W * H and a["stats"].shaded_hits <= 4 * W * H     # MAX_RECURSION_DEPTH 4
    assert a["stats"].light_shadow_rays <= 65 * a["stats"].shaded_hits and a["stats"].sky_shadow_rays <= a["stats"].shaded_hits
    gpu_ctx.clear(pt.history, pt.rows * W * 4); gpu_ctx.clear(pt.stats, C.sizeof(A.RayStats))
    pt.render(view, 1); b = pt.read()
    np.testing.assert_array_equal(a["target"], b["target"])                 # deterministic: same bits on a re-run
    assert stats_tuple(a["stats"]) == stats_tuple(b["stats"])
    # integrate is the running mean of the stored texels (checked on the host from the two buffers)
    h1 = b["history"].copy()
    gpu_ctx.clear(pt.stats, C.sizeof(A.RayStats))
    pt.render(view, 2); c = pt.read()
    cur = c["target"].view(np.float16).astype(np.float32)[..., :3]
    want = (ob.unpack_r11g11b10(h1) + cur) / np.float32(2.0)
    got = ob.unpack_r11g11b10(c["history"])
    assert (got[np.isfinite(want) & (want < 0)] == 0).all()                 # R11G11B10 has no sign: negative radiance stores 0
    fin = np.isfinite(want) & (want < 60000) & (want >= 0)
    assert (got[fin] <= want[fin]).all() and (want[fin] - got[fin] <= np.maximum(want[fin] * 2.0 ** -5, 2e-6)).all()
    # the way the frame is issued does not change a bit of it: one pipeline / the default two / four, at the bench's own size
    for pipelines in (1, 4):
        gpu_ctx.set_pathtrace_pipelines(pipelines)
        gpu_ctx.clear(pt.history, pt.rows * W * 4); gpu_ctx.clear(pt.stats, C.sizeof(A.RayStats))
        pt.render(view, 1); e = pt.read()
        np.testing.assert_array_equal(a["target"], e["target"])
        np.testing.assert_array_equal(a["history"], e["history"])
        assert stats_tuple(a["stats"]) == stats_tuple(e["stats"])
    gpu_ctx.set_pathtrace_pipelines(0)
    # a 64x64 crop of the 4K frame against the oracle (global pixel indices key the sampler)
    ys, xs = np.mgrid[1000:1064, 1900:1964]
    d = rt.make_pathtrace_desc(view, W, H, 1)
    osc = ob.OracleScene(s)
    org, dirs = osc.camera_rays(d, xs.ravel(), ys.ravel())
    gi, gt = sc.trace_rays(org, dirs, 0.001, 10000.0)
    oi, ot = osc.trace_rays(org, dirs, 0.001, 10000.0)
    np.testing.assert_array_equal(gi, oi)
    np.testing.assert_array_equal(gt.view(np.uint32), ot.view(np.uint32))
    pt.close(); sc.close()
