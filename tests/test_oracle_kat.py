"""Known-answer and property tests that pin the oracle's building blocks.

The reference ships no golden vectors (parity unpinned), so the anchors are
(a) published properties of the algorithms the HLSL implements (Kensler 2013 CMJ stratification, IEEE
half-float conversion, D3D small-float formats, Halton radical inverse) and (b) regression vectors generated
by tools/make_golden.py from this oracle (tests/golden/*.json)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN


def _f(n):
    return (C.c_float * n)()


def test_cmj_permute_is_a_permutation(ob):
    L = ob.lib()
    # powers of two (the renderer only uses l = 16 and l = 256): a permutation for every 32-bit pattern;
    # other l: only while (i + p) does not wrap 2^32 (the final "(i + p) % l" is not wrap-safe, as in Kensler 2013)
    for l, ps in ((1, (0, 7)), (2, (0, 1, 0xffffffff)), (16, (0, 1, 0x51633e2d, 0xdeadbeef, 0xffffffff)),
                  (256, (0, 1, 0x51633e2d, 0xdeadbeef, 0xffffffff)), (3, (0, 1, 0x51633e2d)), (255, (0, 5, 0x68bc21eb))):
        for p in ps:
            vals = sorted(L.orc_cmj_permute(i, l, p) for i in range(l))
            assert vals == list(range(l)), (l, p)


def test_cmj_is_stratified_like_kensler_2013(ob):
    """Correlated multi-jittered sampling: the N = n*n samples of one pattern occupy every one of the N
    columns and N rows of the fine grid exactly once, and every cell of the n x n coarse grid exactly once."""
    L = ob.lib()
    n = 16
    N = n * n
    out = _f(2)
    for pattern in (0, 1, 12345, 512 * 512 * 3 + 77, 3840 * 2160 * 2 + 8294399):
        pts = []
        for s in range(N):
            L.orc_cmj_sample(s, n, n, pattern, out)
            pts.append((out[0], out[1]))
        pts = np.array(pts, dtype=np.float64)
        assert (pts >= 0).all() and (pts < 1).all()
        assert sorted(np.floor(pts[:, 0] * N).astype(int)) == list(range(N))
        assert sorted(np.floor(pts[:, 1] * N).astype(int)) == list(range(N))
        cells = set(zip(np.floor(pts[:, 0] * n).astype(int), np.floor(pts[:, 1] * n).astype(int)))
        assert len(cells) == N


def test_rand_float_range_and_scale(ob):
    L = ob.lib()
    vals = [L.orc_cmj_randfloat(i, p) for i in range(0, 4000, 7) for p in (0, 0x967a889b, 0xffffffff)]
    assert min(vals) >= 0.0 and max(vals) < 1.0
    # the scale constant is 1/4294967808 (not 2^-32): the largest uint32 maps just below 1
    assert np.float32(0xffffffff) * np.float32(1.0 / 4294967808.0) < 1.0


def test_halton(ob):
    L = ob.lib()
    assert [L.orc_halton(i, 2) for i in range(1, 9)] == pytest.approx([0.5, 0.25, 0.75, 0.125, 0.625, 0.375, 0.875, 0.0625], abs=1e-7)
    assert [L.orc_halton(i, 3) for i in range(1, 5)] == pytest.approx([1 / 3, 2 / 3, 1 / 9, 4 / 9], abs=1e-6)


def test_half_float_rne_matches_ieee(ob):
    L = ob.lib()
    rng = np.random.default_rng(7)
    xs = np.concatenate([rng.standard_normal(4000).astype(np.float32) * np.float32(10.0) ** rng.integers(-9, 6, 4000).astype(np.float32),
                         np.array([0, -0.0, 1, -1, 65504, 65520, 65519.99, 1e-8, 5.96e-8, 6.1e-5, np.inf, -np.inf], dtype=np.float32)])
    for x in xs:
        got = L.orc_f32_to_f16(float(x), 1)
        with np.errstate(over="ignore"):
            want = int(np.array([x], dtype=np.float32).astype(np.float16).view(np.uint16)[0])
        assert got == want, (x, hex(got), hex(want))
    assert L.orc_f32_to_f16(float("nan"), 1) & 0x7c00 == 0x7c00 and L.orc_f32_to_f16(float("nan"), 1) & 0x3ff


def test_half_float_rtz_brackets_value(ob):
    """Round-toward-zero (D3D11.3 functional spec 3.2.3.3): |decode(encode(x))| <= |x| < |next representable|."""
    L = ob.lib()
    rng = np.random.default_rng(11)
    xs = rng.standard_normal(3000).astype(np.float32) * np.float32(10.0) ** rng.integers(-7, 5, 3000).astype(np.float32)
    for x in xs:
        h = L.orc_f32_to_f16(float(x), 0)
        d = L.orc_f16_to_f32(h)
        assert abs(d) <= abs(x) and np.sign(d) in (0, np.sign(x))
        nxt = L.orc_f16_to_f32((h & 0x8000) | ((h & 0x7fff) + 1))
        assert abs(nxt) > abs(x) or np.isinf(nxt)
    assert L.orc_f32_to_f16(1e9, 0) == 0x7bff          # overflow saturates to the largest finite half under RTZ
    assert L.orc_f32_to_f16(1e9, 1) == 0x7c00          # ... and to +inf under nearest-even


def test_r11g11b10_roundtrip_and_range(ob):
    L = ob.lib()
    out = _f(3)
    # exactly representable values survive; negatives clamp to 0; max finite 65024 / 64512
    for rgb in ([1.0, 0.5, 0.25], [65024.0, 65024.0, 64512.0], [0.0, 3.0, 7.0], [2.0 ** -14, 2.0 ** -15, 2.0 ** -19]):
        p = L.orc_pack_r11g11b10((C.c_float * 3)(*rgb), 0)
        L.orc_unpack_r11g11b10(p, out)
        assert list(out) == rgb
    p = L.orc_pack_r11g11b10((C.c_float * 3)(-1.0, 1e9, float("inf")), 0)
    L.orc_unpack_r11g11b10(p, out)
    assert out[0] == 0.0 and out[1] == 65024.0 and np.isinf(out[2])
    # truncation: value between two codes maps to the lower one (6-bit mantissa: step 1/64 at 1.0)
    p = L.orc_pack_r11g11b10((C.c_float * 3)(1.0 + 1.9 / 64, 1.0 + 1.9 / 64, 1.0 + 1.9 / 32), 0)
    L.orc_unpack_r11g11b10(p, out)
    assert list(out) == [1.0 + 1 / 64, 1.0 + 1 / 64, 1.0 + 1 / 32]
    p = L.orc_pack_r11g11b10((C.c_float * 3)(1.0 + 1.9 / 64, 1.0 + 1.5 / 64, 1.0 + 0.4 / 32), 1)
    L.orc_unpack_r11g11b10(p, out)
    assert list(out) == [1.0 + 2 / 64, 1.0 + 2 / 64, 1.0]
    # the numpy decoder used by the tests agrees with the C one
    ps = np.random.default_rng(3).integers(0, 2 ** 32, 2000, dtype=np.uint64).astype(np.uint32)
    dec = ob.unpack_r11g11b10(ps)
    for i in range(0, 2000, 37):
        L.orc_unpack_r11g11b10(int(ps[i]), out)
        np.testing.assert_array_equal(np.nan_to_num(dec[i], nan=-1), np.nan_to_num(np.array(list(out), dtype=np.float32), nan=-1))


def test_brdf_and_tonemap_scalars(ob):
    L = ob.lib()
    assert L.orc_exposure(13) == pytest.approx(1.0 / (8192 * 1.2), rel=1e-7)
    out = _f(3)
    L.orc_aces_film((C.c_float * 3)(0.0, 1.0, 100.0), out)
    assert out[0] == 0.0 and out[1] == pytest.approx((2.51 + 0.03) / (2.43 + 0.59 + 0.14), rel=1e-6) and out[2] == 1.0
    # GGX normalisation: D(h) at NoH = 1 is 1/(pi a^2) with a = roughness^2
    for r in (0.5, 0.7, 0.9):
        assert L.orc_ggx(1.0, r) == pytest.approx(1.0 / (np.pi * r ** 4), rel=1e-5)
    assert L.orc_ggx(1.0, 0.2) == pytest.approx(0.2 ** 4 / 1e-4, rel=1e-5)      # denominator clamp max(denom, 1e-4)
    assert L.orc_g_smith_direct(1.0, 1.0, 0.7) == pytest.approx(1.0)
    L.orc_f_schlick(1.0, (C.c_float * 3)(0.04, 0.5, 1.0), out)
    assert list(out) == pytest.approx([0.04, 0.5, 1.0])
    L.orc_f_schlick(0.0, (C.c_float * 3)(0.04, 0.5, 1.0), out)
    assert list(out) == pytest.approx([1.0, 1.0, 1.0])


def test_hemisphere_sampling_is_unit_and_cosine(ob):
    L = ob.lib()
    out = _f(3)
    zs = []
    for i in range(64):
        for j in range(64):
            L.orc_cosine_hemisphere((C.c_float * 2)((i + 0.5) / 64, (j + 0.5) / 64), out)
            v = np.array(list(out))
            assert abs(np.linalg.norm(v) - 1) < 1e-5 and v[2] >= 0
            zs.append(v[2])
    assert np.mean(zs) == pytest.approx(2.0 / 3.0, abs=5e-3)        # E[cos] under a cosine-weighted pdf


def test_preetham_sky_is_finite_everywhere(ob, sio):
    """The horizon / below-horizon directions are where cos(acos(0)) would give 0 * inf; pinned to stay finite."""
    L = ob.lib()
    out = _f(3)
    sun = sio.sun_direction_from_time_of_day()
    assert np.linalg.norm(sun) == pytest.approx(1.0, abs=1e-6)
    assert list(sun) == pytest.approx([0.40592, 0.90698, -0.11229], abs=2e-5)        # 11h, day 200, latitude 42.5
    for e in ([0, 1, 0], [1, 0, 0], [0, -1, 0], [0.6, -0.8, 0], list(sun), [-0.3, 0.2, 0.93]):
        e = np.array(e, dtype=np.float64); e /= np.linalg.norm(e)
        L.orc_sky_radiance((C.c_float * 3)(*sun), (C.c_float * 3)(*e), 2.0, out)
        v = np.array(list(out))
        assert np.isfinite(v).all() and (v > -1e-3).all(), (e, v)
    L.orc_sky_radiance((C.c_float * 3)(*sun), (C.c_float * 3)(*sun), 2.0, out)
    towards_sun = out[1]
    L.orc_sky_radiance((C.c_float * 3)(*sun), (C.c_float * 3)(-sun[0], sun[1], -sun[2]), 2.0, out)
    assert towards_sun > out[1]                                   # circumsolar brightening


def test_matrix_inverse(ob, A):
    L = ob.lib()
    rng = np.random.default_rng(5)
    for _ in range(20):
        m = (rng.standard_normal((4, 4)) + 3 * np.eye(4)).astype(np.float32)
        src, dst = A.Matrix(), A.Matrix()
        for i in range(16):
            src.m[i] = float(m.reshape(16)[i])
        L.orc_mat_inverse(C.byref(src), C.byref(dst))
        inv = np.array(list(dst.m), dtype=np.float64).reshape(4, 4)
        np.testing.assert_allclose(m.astype(np.float64) @ inv, np.eye(4), atol=2e-5)


def test_golden_sampler_tables(ob):
    """Regression vectors (tools/make_golden.py): SamplePoint at 512^2 and 4K pixel indices, integer stages exact."""
    with open(os.path.join(GOLDEN, "sampler.json")) as f:
        g = json.load(f)
    L = ob.lib()
    out = _f(2)
    for row in g["sample_point"]:
        L.orc_sample_point(row["pixel"], row["sample"], row["set"], 16, row["num_pixels"], out)
        assert np.float32(out[0]).view(np.uint32) == row["x_bits"] and np.float32(out[1]).view(np.uint32) == row["y_bits"]
    for row in g["permute"]:
        assert L.orc_cmj_permute(row["i"], row["l"], row["p"]) == row["out"]


def _bits(v):
    return int(np.float32(v).view(np.uint32))


def _unbits(b):
    return float(np.uint32(b).view(np.float32))


def test_golden_scalar_kats(ob, sio, R, A):
    """tests/golden/scalar_kats.json: the oracle reproduces every stored bit pattern, and
    each table is checked against the closed form it restates, evaluated independently in float64."""
    with open(os.path.join(GOLDEN, "scalar_kats.json")) as f:
        g = json.load(f)
    L = ob.lib()
    out = _f(3)
    for row in g["ggx"]:
        assert _bits(L.orc_ggx(row["NoH"], row["r"])) == row["ggx"] and _bits(L.orc_ggx_pdf(row["NoH"], row["r"])) == row["pdf"]
        a2 = row["r"] ** 4
        d = max((row["NoH"] * row["NoH"]) * (a2 - 1.0) + 1.0, 0.0)
        assert _unbits(row["ggx"]) == pytest.approx(a2 / max(np.pi * d * d, 1e-4), rel=2e-5)         # bxdf-sampling.hlsli GGX, clamped denominator
        assert _unbits(row["pdf"]) == pytest.approx(_unbits(row["ggx"]) * row["NoH"], rel=2e-6)
    for row in g["g_smith"]:
        assert _bits(L.orc_g_smith_direct(row["NoV"], row["NoL"], row["r"])) == row["direct"]
        assert _bits(L.orc_g_smith_ggx_correlated(row["NoV"], row["NoL"], row["r"])) == row["correlated"]
        assert 0.0 <= _unbits(row["direct"]) <= 1.0
    for row in g["f_schlick"]:
        L.orc_f_schlick(row["u"], (C.c_float * 3)(*row["f0"]), out)
        assert [_bits(v) for v in out] == row["out"]
        want = [f0 + (1.0 - f0) * (1.0 - row["u"]) ** 5 for f0 in row["f0"]]
        assert [_unbits(b) for b in row["out"]] == pytest.approx(want, rel=2e-6)
    for row in g["exposure"]:
        assert _bits(L.orc_exposure(row["ev100"])) == row["out"]
        assert _unbits(row["out"]) == pytest.approx(1.0 / (2.0 ** row["ev100"] * 1.2), rel=1e-6)       # lighting/pbr.hlsli:78-81
    for row in g["aces_film"]:
        L.orc_aces_film((C.c_float * 3)(*row["x"]), out)
        assert [_bits(v) for v in out] == row["out"]
        want = [min(max(x * (2.51 * x + 0.03) / (x * (2.43 * x + 0.59) + 0.14), 0.0), 1.0) for x in row["x"]]
        assert [_unbits(b) for b in row["out"]] == pytest.approx(want, rel=3e-6, abs=1e-7)
    # Preetham at the time-of-day defaults
    sun = sio.sun_direction_from_time_of_day()
    assert [_bits(v) for v in sun] == g["preetham"]["sun"]
    for row in g["preetham"]["samples"]:
        e = (C.c_float * 3)(*[_unbits(b) for b in row["dir"]])
        L.orc_sky_radiance((C.c_float * 3)(*sun), e, g["preetham"]["turbidity"], out)
        assert [_bits(v) for v in out] == row["out"] and all(np.isfinite(v) and v >= 0 for v in out)
    # octahedral encoding
    for row in g["oct"]:
        n = [_unbits(b) for b in row["n"]]
        enc, dec = _f(2), _f(3)
        L.orc_oct_roundtrip((C.c_float * 3)(*n), enc, dec)
        assert [_bits(v) for v in enc] == row["enc"] and [_bits(v) for v in dec] == row["dec"]
        s = sum(abs(v) for v in n); p = [n[0] / s, n[1] / s]
        if n[2] < 0:
            p = [(1 - abs(n[1] / s)) * (1 if p[0] >= 0 else -1), (1 - abs(n[0] / s)) * (1 if p[1] >= 0 else -1)]
        assert [_unbits(b) for b in row["enc"]] == pytest.approx([p[0] * 0.5 + 0.5, p[1] * 0.5 + 0.5], abs=2e-7)     # encoding.hlsli:37-43
    # visibility ids (encoding.hlsli:4-28) and the clear value (visibility-pass.inl:68-74)
    v = g["visibility_encoding"]
    assert v["clear"] == 0xFFFFF000
    for row in v["meshlet"]:
        assert row["value"] >> 7 == row["id"] and row["value"] & 127 == row["tri"]
    for row in v["primitive"]:
        assert row["value"] >> 20 == row["id"] and row["value"] & 0xFFFFF == row["tri"]
    # light clusters (cluster-culling.hlsli:57-60; clustered-lighting.inl:88,102)
    c = g["clusters"]
    for k, b in zip((0, 1, 12, 24), c["slice_depths"]):
        assert _unbits(b) == pytest.approx(200.0 ** (k / 24.0), rel=1e-6)
    assert _unbits(c["scale"]) == pytest.approx(24.0 / np.log(200.0), rel=1e-6) and _unbits(c["bias"]) == 0.0
    synth = __import__("importlib").import_module("incarnation-renderer_amd.synth")
    scene = synth.sponza_stat(texture_size=8, tri_scale=1 / 64.0)
    view = sio.FView().reset(scene, aspect=16 / 9)
    vc = R.make_view_constants(view, 1920, 1080)
    grid = (C.c_uint32 * 3)(16, 9, 24)
    for key, want in c["planes"].items():
        ci = (C.c_uint32 * 3)(*[int(t) for t in key.split(",")])
        got = (C.c_float * 24)()
        L.orc_cluster_frustum(ci, grid, 1.0, 200.0, C.byref(vc.lightcull_proj_transform), C.byref(vc.lightcull_inv_view_proj), got)
        assert [_bits(x) for x in got] == want
        pl = np.array([_unbits(b) for b in want]).reshape(6, 4)
        # the cluster's own centre lies inside all six planes (cluster-culling.hlsli:116-129 sign convention)
        eye = np.array(list(vc.eye_pos)); look = np.array(list(view.to_ctypes().look), dtype=np.float64)
        assert np.isfinite(pl).all() and (np.linalg.norm(pl[:, :3], axis=1) > 0).all()


def test_golden_raster_crop(ob, helmet, sio, R, A):
    """tests/golden/helmet_raster_64.npz: 64x64 visibility ids, depth, G-buffers and colour target of the
    helmet frame (TAA off), regenerated by the oracle bit for bit."""
    g = np.load(os.path.join(GOLDEN, "helmet_raster_64.npz"))
    W = H = 64

    class NoTaa(A.FConfig):
        EnableTAA = False
    osc = ob.OracleScene(helmet)
    view = sio.FView().reset(helmet, aspect=1.0)
    hs = ob.HostSurfaces(W, H, helmet.num_meshlets, 16 * 9 * 24, 64)
    hs.arr["sh9"][:, :3] = 0.25
    vc = R.make_view_constants(view, W, H)
    for name, d in R.build_descs(None, helmet, vc, hs.ptr, 0, NoTaa):
        ob.run_pass(osc, name, d)
    for key, name in (("vis", "vis"), ("g0", "gbuffer_basecolor"), ("g1", "gbuffer_normals"), ("g2", "gbuffer_mra"), ("color", "color"), ("counts", "counts")):
        np.testing.assert_array_equal(hs.arr[name], g[key], err_msg=key)
    np.testing.assert_array_equal(hs.arr["depth"].view(np.uint32), g["depth"].view(np.uint32))
    covered = g["vis"] != 0xFFFFF000
    assert 0.1 < covered.mean() < 0.9 and (g["depth"][~covered] == 0).all()
    osc.close()


def test_example_hdr_decoder_matches_the_oracle(ob):
    """examples/render.py --dump-hdr decodes R11G11B10_FLOAT in numpy: every bit pattern class against the oracle's decoder."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    import render
    w = np.concatenate([np.random.default_rng(3).integers(0, 2 ** 32, 100000, dtype=np.uint64).astype(np.uint32),
                        np.array([0, 0xffffffff, 0x7c0 | (0x7c0 << 11) | (0x3e0 << 22), 1 | (1 << 11) | (1 << 22)], np.uint32)])
    a, b = render.decode_r11g11b10(w), ob.unpack_r11g11b10(w)
    fin = np.isfinite(b)
    assert np.array_equal(a[fin], b[fin]) and (np.isnan(a) == np.isnan(b)).all() and (np.isinf(a) == np.isinf(b)).all()


def test_example_exr_writer_round_trip(tmp_path):
    """examples/render.py --dump-hdr x.exr: an uncompressed scan-line OpenEXR (magic 20000630, version 2, channels B G R FLOAT); the
    reader of tools/psnr_capture.py gets the array back bit for bit, specials included."""
    import struct, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    import render
    rng = np.random.default_rng(4)
    img = rng.uniform(0, 60000, (7, 13, 3)).astype(np.float32)
    img[0, 0] = [0.0, np.inf, 1e-30]; img[6, 12] = [np.nan, 65504.0, 1.0]
    p = os.path.join(tmp_path, "t.exr")
    render.write_exr(p, img)
    raw = open(p, "rb").read()
    assert struct.unpack_from("<II", raw, 0) == (20000630, 2) and b"channels\0chlist\0" in raw[:64] and b"compression\0compression\0\1\0\0\0\0" in raw
    assert len(raw) == raw.index(b"screenWindowWidth") + len(b"screenWindowWidth\0float\0") + 4 + 4 + 1 + 8 * 7 + 7 * (8 + 13 * 12)
    back = render.read_exr(p)
    np.testing.assert_array_equal(back.view(np.uint32), img.view(np.uint32))
