"""g_anisoSampler of the raster passes (gbuffer-compute.hlsl:20 / visibility-pass.hlsl:11 / gbuffer-raster.hlsl:11: FILTER_ANISOTROPIC
with the static sampler's default MaxAnisotropy 16) as the oracle pins it: N = min(ceil(rho_max / rho_min), MaxAnisotropy) trilinear
taps on the line of the longer derivative at the LOD of the (clamped) shorter one.  The taps themselves are checked through the
ISOTROPIC sampler -- a footprint with two equal perpendicular derivatives takes one tap at log2 of their length -- so the rule is
verified against the part of the sampler the earlier rounds already pinned."""
import ctypes as C

import numpy as np
import pytest


def _scene(synth, sio, A, max_anisotropy, size=64):
    s = synth.triangle_soup(3, 64)
    s.textures = []
    rng = np.random.default_rng(11)
    tex = rng.integers(0, 256, (size, size, 4), dtype=np.uint8)
    tex[..., 0] = (np.arange(size)[None, :] * 255 // (size - 1)).astype(np.uint8)        # a ramp along u in red
    s.add_texture(tex, A.TEX_RGBA8_UNORM)
    for m in s.materials:
        for n in A._MAT_TEX:
            setattr(m, n + "TextureIndex", -1)
    s.max_anisotropy = max_anisotropy
    s._desc = None
    return s


def _sampler(ob, osc):
    L = ob.lib()
    L.orc_sample_texture_grad.argtypes = [C.c_void_p, C.c_int] + [C.POINTER(C.c_float)] * 4
    L.orc_sample_texture_grad.restype = None
    out = (C.c_float * 4)()

    def sample(uv, ddx, ddy):
        L.orc_sample_texture_grad(osc.handle, 0, (C.c_float * 2)(*uv), (C.c_float * 2)(*ddx), (C.c_float * 2)(*ddy), out)
        return np.array(list(out), dtype=np.float32)
    return sample


def _line_of_taps(iso, uv, along, taps, minor):
    """1/N * sum of isotropic trilinear taps (footprint `minor` in uv units) at uv + along * (i / (N + 1) - 1/2), float32 as the oracle sums."""
    acc = np.zeros(4, dtype=np.float32)
    for i in range(1, taps + 1):
        where = np.float32(i) / np.float32(taps + 1) - np.float32(0.5)
        p = (np.float32(uv[0]) + np.float32(along[0]) * where, np.float32(uv[1]) + np.float32(along[1]) * where)
        acc = acc + iso(p, (minor, 0.0), (0.0, minor))
    return acc / np.float32(taps)


@pytest.fixture(scope="module")
def samplers(ob, synth, sio, A):
    made = []
    out = {}
    for k in (1, 4, 16):
        osc = ob.OracleScene(_scene(synth, sio, A, k)); made.append(osc)
        out[k] = _sampler(ob, osc)
    # max_anisotropy 0 = the reference's default
    osc = ob.OracleScene(_scene(synth, sio, A, 0)); made.append(osc)
    out[0] = _sampler(ob, osc)
    yield out
    for o in made:
        o.close()


def test_default_is_sixteen_and_isotropic_footprints_take_one_tap(samplers):
    rng = np.random.default_rng(5)
    for _ in range(50):
        uv = rng.random(2) * 3 - 1
        m = float(2.0 ** rng.uniform(-8, -2))
        c, s = np.cos(rng.uniform(0, 6.3)), np.sin(rng.uniform(0, 6.3))
        ddx, ddy = (m, 0.0), (0.0, m)
        a = samplers[1](uv, ddx, ddy)
        for k in (0, 4, 16):
            np.testing.assert_array_equal(samplers[k](uv, ddx, ddy), a)
        # an elongated footprint: the default equals MaxAnisotropy 16 and differs from the isotropic sampler
        ddx = (m * 6.5, 0.0)
        np.testing.assert_array_equal(samplers[0](uv, ddx, ddy), samplers[16](uv, ddx, ddy))


@pytest.mark.parametrize("ratio,max_aniso,taps", [(1.5, 16, 2), (1.99, 16, 2), (2.01, 16, 3), (7.3, 16, 8), (15.9, 16, 16), (40.0, 16, 16), (7.3, 4, 4), (2.9, 4, 3)])
def test_taps_lie_on_the_longer_derivative_at_the_lod_of_the_shorter(samplers, ratio, max_aniso, taps):
    iso = samplers[1]
    rng = np.random.default_rng(int(ratio * 100) + max_aniso)
    for trial in range(12):
        uv = rng.random(2)
        minor = float(2.0 ** rng.uniform(-5.5, -3.0))                  # 1 to 8 texels of a 64^2 texture: LOD 0.5 .. 3
        major = minor * ratio
        # the footprint after the ratio clamp: minor axis = major / MaxAnisotropy when the ratio exceeds it
        eff_minor = max(minor, major / max_aniso)
        if trial % 2 == 0:
            ddx, ddy, along = (major, 0.0), (0.0, minor), (major, 0.0)
        else:
            ddx, ddy, along = (0.0, minor), (0.0, -major), (0.0, -major)       # the longer derivative may be ddy, any direction
        got = samplers[max_aniso](uv, ddx, ddy)
        want = _line_of_taps(iso, uv, along, taps, eff_minor)
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-6, err_msg="ratio %g max %d" % (ratio, max_aniso))


def test_anisotropic_sampling_keeps_detail_an_isotropic_footprint_blurs(samplers):
    """Along the ramp (u) the footprint is 1 texel, across it 12: the isotropic sampler reads level 3-4 and returns the average of a
    12-texel-wide block of the ramp; the anisotropic one keeps the ramp's value at u."""
    size = 64
    for u in (0.2, 0.4, 0.6, 0.8):
        uv = (u, 0.37)
        ddx, ddy = (1.0 / size, 0.0), (0.0, 12.0 / size)
        exact = (u * size - 0.5) / (size - 1)                         # the ramp under bilinear filtering
        a16 = samplers[16](uv, ddx, ddy)[0]
        assert abs(a16 - exact) < 0.012
        # along the ramp instead: both samplers must average over the long footprint
        ddx, ddy = (12.0 / size, 0.0), (0.0, 1.0 / size)
        assert abs(samplers[16](uv, ddx, ddy)[0] - exact) < 0.03 and abs(samplers[1](uv, ddx, ddy)[0] - exact) < 0.05


def test_degenerate_footprints(samplers):
    for k in (1, 16):
        base = samplers[k]((0.3, 0.6), (0.0, 0.0), (0.0, 0.0))                     # no footprint: level 0, one tap
        np.testing.assert_array_equal(base, samplers[k]((0.3, 0.6), (float("nan"), 0.0), (0.0, 0.0)))
        np.testing.assert_array_equal(base, samplers[k]((0.3, 0.6), (float("inf"), 0.0), (0.0, 1e-3)))
        assert np.isfinite(samplers[k]((0.3, 0.6), (0.05, 0.0), (0.0, 0.0))).all()  # a line footprint: rho_min = 0 -> MaxAnisotropy taps
    # rho_min = 0 is the clamped case: 16 taps at log2(rho_max / 16)
    got = samplers[16]((0.3, 0.6), (0.25, 0.0), (0.0, 0.0))
    want = _line_of_taps(samplers[1], (0.3, 0.6), (0.25, 0.0), 16, 0.25 / 16)
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-6)


def test_scene_validation_rejects_more_than_sixteen(rt, synth, sio, A):
    lib = rt.load()
    lib.incr_scene_validate_host.restype = C.c_int
    lib.incr_scene_validate_host.argtypes = [C.POINTER(A.SceneDesc), C.POINTER(A.SceneInfo)]
    d = _scene(synth, sio, A, 16).to_desc()
    assert lib.incr_scene_validate_host(C.byref(d), C.byref(A.SceneInfo())) == 0, lib.incr_last_error()
    d = _scene(synth, sio, A, 17).to_desc()
    assert lib.incr_scene_validate_host(C.byref(d), C.byref(A.SceneInfo())) == -1
    assert b"max_anisotropy" in lib.incr_last_error()
