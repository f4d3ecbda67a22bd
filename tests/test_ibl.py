"""Sky / IBL precompute: oracle properties on the CPU, HIP-vs-oracle parity on the GPU.
The chain goes through sinf/cosf/expf/tanf/acosf/atan2f/log2f, which differ in the last bit between glibc and the device
math library, so radiance surfaces are compared within one R11G11B10 storage step (2^-5 relative on blue, 2^-6 on red /
green) instead of bit for bit; everything that is pure arithmetic on given inputs is compared exactly."""
import ctypes as C
import os

import numpy as np
import pytest


def levels_2d(flat, res):
    out, off = [], 0
    w, h = 2 * res, res
    while h >= 1:
        out.append(flat[off:off + w * h].reshape(h, w)); off += w * h
        w, h = max(w >> 1, 1), h >> 1
    return out


def levels_cube(flat, res):
    out, off, n = [], 0, res
    while n >= 1:
        out.append(flat[off:off + 6 * n * n].reshape(6, n, n)); off += 6 * n * n
        n >>= 1
    return out


def env_dir(face, u, v):
    """split-sum-approx/prefilter.hlsl GetEnvDir."""
    vx, vy = 2 * u - 1, -2 * v + 1
    d = [(1, vy, -vx), (-1, vy, vx), (vx, 1, -vy), (vx, -1, vy), (vx, vy, 1), (-vx, vy, -1)][face]
    d = np.array(d, dtype=np.float64)
    return d / np.linalg.norm(d)


SUN = (0.35, 0.8, 0.48)


@pytest.fixture(scope="module")
def sky16(ob):
    return ob.update_dynamic_sky(SUN, 2.0, resolution=16)


def test_oracle_latlong_and_mips(ob, sky16):
    res = 16
    lv = levels_2d(sky16["latlong"], res)
    assert [l.shape for l in lv] == [(16, 32), (8, 16), (4, 8), (2, 4), (1, 2)] and sky16["mips"] == 5
    # level 0 = 0.04 * Preetham along the lat-long direction of the texel centre
    rad = ob.unpack_r11g11b10(lv[0])
    sun = np.array(SUN) / np.linalg.norm(SUN)
    for (y, x) in ((2, 3), (5, 17), (7, 30)):
        theta, phi = np.pi * (y + 0.5) / res, np.pi * ((x + 0.5) / (2 * res) * 2 - 1)
        d = np.array([np.sin(theta) * np.sin(phi), np.cos(theta), np.sin(theta) * np.cos(phi)])
        out = (C.c_float * 3)()
        ob.lib().orc_sky_radiance((C.c_float * 3)(*sun), (C.c_float * 3)(*d), 2.0, out)
        want = 0.04 * np.array(out[:])
        assert (rad[y, x] <= want * (1 + 1e-5)).all() and (rad[y, x] >= want * (1 - 2.0 ** -5)).all()      # truncating store
    # every level is the 2x2 box filter of the one above (one storage step of slack)
    for a, b in zip(lv[:-1], lv[1:]):
        fa = ob.unpack_r11g11b10(a)
        box = 0.25 * (fa[0::2, 0::2] + fa[0::2, 1::2] + fa[1::2, 0::2] + fa[1::2, 1::2])
        fb = ob.unpack_r11g11b10(b)
        assert (fb <= box * (1 + 1e-5)).all() and (fb >= box * (1 - 2.0 ** -5) - 1e-6).all()


def test_oracle_cube_conventions(ob, sky16):
    """The cube sampler (D3D face table) and GetEnvDir (the writer's face table) agree: sampling along the direction of a
    texel centre returns that texel."""
    res = 16
    cube = levels_cube(sky16["source_cube"], res)
    flat = np.ascontiguousarray(sky16["source_cube"])
    rng = np.random.default_rng(1)
    for _ in range(200):
        face, x, y = int(rng.integers(6)), int(rng.integers(res)), int(rng.integers(res))
        d = env_dir(face, (x + 0.5) / res, (y + 0.5) / res).astype(np.float32)
        out = (C.c_float * 3)()
        ob.lib().orc_cube_sample(flat.ctypes.data, res, 5, (C.c_float * 3)(*d), 0.0, out)
        want = ob.unpack_r11g11b10(cube[0][face, y, x])
        np.testing.assert_allclose(np.array(out[:]), want, rtol=2e-5, atol=1e-7)
    # +Y face looks at the zenith: brighter than the -Y face of a daytime sky
    up, down = ob.unpack_r11g11b10(cube[0][2]).mean(), ob.unpack_r11g11b10(cube[0][3])
    assert np.isfinite(up) and up > 0


def test_oracle_prefilter_level0_and_sh(ob, sky16):
    res = 16
    src, env = levels_cube(sky16["source_cube"], res), levels_cube(sky16["envmap"], res)
    flat = np.ascontiguousarray(sky16["source_cube"])
    # roughness 0: all 1024 samples point along R = GetEnvDir(face, x / n, y / n) -> the prefiltered texel is the bilinear
    # source sample at that (corner) direction
    for face, x, y in ((0, 3, 4), (2, 8, 8), (5, 15, 1), (3, 0, 9)):
        d = env_dir(face, x / res, y / res).astype(np.float32)
        out = (C.c_float * 3)()
        ob.lib().orc_cube_sample(flat.ctypes.data, res, 5, (C.c_float * 3)(*d), 0.0, out)
        got = ob.unpack_r11g11b10(env[0][face, y, x])
        want = np.array(out[:])
        assert (got <= want * (1 + 1e-4)).all() and (got >= want * (1 - 2.0 ** -5 - 1e-4)).all()
    # rougher levels are smoother: the spread over a face shrinks
    spread = [float(np.ptp(ob.unpack_r11g11b10(l[4])[..., 1])) for l in env[:3]]
    assert spread[2] <= spread[0] + 1e-6
    # SH9: coefficient i = sum over texels of radiance * basis_i * sin(theta) dTheta dPhi
    lv0 = ob.unpack_r11g11b10(levels_2d(sky16["latlong"], res)[0]).astype(np.float64)
    H, W = res, 2 * res
    ys, xs = np.mgrid[0:H, 0:W]
    theta, phi = np.pi * (ys + 0.5) / H, np.pi * ((xs + 0.5) / W * 2 - 1)
    d = np.stack([np.sin(theta) * np.sin(phi), np.cos(theta), np.sin(theta) * np.cos(phi)], -1)
    K = [0.282095, 0.488603, 0.488603, 0.488603, 1.092548, 1.092548, 0.315392, 1.092548, 0.546274]
    basis = [K[0] + 0 * theta, K[1] * d[..., 1], K[2] * d[..., 2], K[3] * d[..., 0], K[4] * d[..., 0] * d[..., 1], K[5] * d[..., 1] * d[..., 2],
             K[6] * (3 * d[..., 2] ** 2 - 1), K[7] * d[..., 0] * d[..., 2], K[8] * (d[..., 0] ** 2 - d[..., 1] ** 2)]
    w = np.sin(theta) * (np.pi / H) * (2 * np.pi / W)
    for i in range(9):
        want = (lv0 * (basis[i] * w)[..., None]).sum(axis=(0, 1))
        np.testing.assert_allclose(sky16["sh9"][i, :3], want, rtol=2e-4, atol=2e-5)


def test_oracle_env_brdf(ob):
    W = H = 32
    lut = ob.env_brdf(W, H)
    a = (lut & 0xffff).astype(np.uint16).view(np.float16).astype(np.float32)
    b = (lut >> 16).astype(np.uint16).view(np.float16).astype(np.float32)
    assert np.isfinite(a).all() and np.isfinite(b).all() and (a >= 0).all() and (b >= 0).all()
    assert (a[:, 0] == 0).all() and (b[:, 0] == 0).all()                   # NoV = 0 column: no sample passes NoV > 0
    assert ((a + b)[:, W // 2:] <= 1.02).all()                             # away from grazing angles scale + bias stay below 1
    # smooth (y = H-1 is roughness 1/H), facing the surface (NoV -> 1): scale ~ 1, bias ~ 0
    assert a[H - 1, W - 1] > 0.9 and b[H - 1, W - 1] < 0.05
    # independent float64 evaluation of the same integral for one texel
    x, y, n = 20, 12, 1024
    NoV, rough = x / W, 1 - y / H
    V = np.array([np.sqrt(1 - NoV * NoV), 0, NoV])
    A = B = 0.0
    for i in range(n):
        bits = int("{:032b}".format(i)[::-1], 2)
        u1, u2 = i / n, bits / 2.0 ** 32
        al = rough * rough
        ct = np.sqrt((1 - u2) / (1 + (al * al - 1) * u2)); st = np.sqrt(1 - ct * ct); ph = 2 * np.pi * u1
        Hh = np.array([st * np.sin(ph), st * np.cos(ph), ct])
        L = 2 * np.dot(V, Hh) * Hh - V
        L /= np.linalg.norm(L)
        NoL, NoH, VoH = min(max(L[2], 0), 1), min(max(Hh[2], 0), 1), min(max(np.dot(V, Hh), 0), 1)
        if NoL > 0 and NoV > 0:
            k = 0.5 * al * al
            G = (NoV / (NoV * (1 - k) + k)) * (NoL / (NoL * (1 - k) + k))
            gv = G * VoH / (NoH * NoV); fc = (1 - VoH) ** 5
            A += (1 - fc) * gv; B += fc * gv
    assert a[y, x] == pytest.approx(A / n, rel=2e-3) and b[y, x] == pytest.approx(B / n, rel=2e-3, abs=1e-4)


# ------------------------------------------------------------------------------------------------ GPU parity
def radiance_close(ob, got, want, what, steps=1.0):
    fa, fb = ob.unpack_r11g11b10(got), ob.unpack_r11g11b10(want)
    assert (np.isfinite(fa) == np.isfinite(fb)).all(), what
    fin = np.isfinite(fa) & np.isfinite(fb)
    tol = np.array([2.0 ** -6, 2.0 ** -6, 2.0 ** -5]) * steps
    err = np.abs(fa - fb)
    ref = np.maximum(np.abs(fb), 1e-6)
    bad = fin & (err > ref * tol + 1e-7)
    assert not bad.any(), "%s: %d texels outside %.1f storage step(s), worst rel %g" % (what, int(bad.sum()), steps, float((err / ref)[fin].max()))
    return float((got == want).mean())


@pytest.mark.gpu
def test_gpu_dynamic_sky_chain_matches_oracle(rt, ob, A, gpu_ctx, helmet):
    ibl_mod = __import__("importlib").import_module("incarnation-renderer_amd.ibl")
    res = 32
    gscene = gpu_ctx.create_scene(helmet)
    try:
        sky = ibl_mod.SkyIbl(gpu_ctx, gscene, resolution=res, turbidity=2.0)
        g = sky.download()
        o = ob.update_dynamic_sky(helmet.sun_dir, 2.0, resolution=res)
        assert sky.info.num_mips == o["mips"] == 6 and g["latlong"].size == o["latlong"].size and g["envmap"].size == o["envmap"].size
        same = radiance_close(ob, g["latlong"], o["latlong"], "lat-long sky + mips")
        assert same > 0.98
        assert radiance_close(ob, g["source_cube"], o["source_cube"], "source cube") > 0.97
        # the prefilter sums 1024 samples of already-quantised texels: two storage steps
        assert radiance_close(ob, g["envmap"], o["envmap"], "prefiltered envmap", steps=2.0) > 0.9
        np.testing.assert_allclose(g["sh9"][:, :3], o["sh9"][:, :3], rtol=2e-3, atol=2e-4)
        # refreshing after the sun moved reuses the buffers and changes the result
        before = g["sh9"].copy()
        sun = (C.c_float * 3)(0.1, 0.3, -0.9)
        rt.check(gpu_ctx.lib.incr_scene_set_sun_dir(gpu_ctx.handle, gscene.handle, sun))
        ptr = sky.info.envmap_r11g11b10
        sky.update()
        assert sky.info.envmap_r11g11b10 == ptr and not np.allclose(sky.download()["sh9"], before)
        o2 = ob.update_dynamic_sky((0.1, 0.3, -0.9), 2.0, resolution=res)
        np.testing.assert_allclose(sky.download()["sh9"][:, :3], o2["sh9"][:, :3], rtol=2e-3, atol=2e-4)
        sky.close()
        # bad arguments
        h = C.c_void_p()
        assert gpu_ctx.lib.incr_update_dynamic_sky(gpu_ctx.handle, gscene.handle, 48, 2.0, 0, None, C.byref(h)) == -1
        assert b"power of two" in gpu_ctx.lib.incr_last_error()
    finally:
        gscene.close()


@pytest.mark.gpu
def test_gpu_env_brdf_matches_oracle(rt, ob, gpu_ctx):
    ibl_mod = __import__("importlib").import_module("incarnation-renderer_amd.ibl")
    W, H = 64, 48
    buf = ibl_mod.generate_env_brdf(gpu_ctx, W, H)
    gpu_ctx.synchronize()
    g = gpu_ctx.download(buf, (H, W), np.uint32)
    gpu_ctx.free(buf)
    o = ob.env_brdf(W, H)
    for shift in (0, 16):
        a = ((g >> shift) & 0xffff).astype(np.uint16).view(np.float16).astype(np.float32)
        b = ((o >> shift) & 0xffff).astype(np.uint16).view(np.float16).astype(np.float32)
        np.testing.assert_allclose(a, b, rtol=2.0 ** -9, atol=2.0 ** -14)         # sinf/cosf of the sample azimuth: within two f16 steps
    assert (g == o).mean() > 0.9


@pytest.mark.gpu
def test_gpu_frame_with_specular_ibl(rt, R, ob, A, sio, gpu_ctx, sponza_small):
    """Raster frame with the precomputed sky inputs wired in (SH9 probe, prefiltered environment cube, env-BRDF LUT): the
    sky-lighting pass of the library against the oracle's, both fed the SAME products (downloaded from the GPU)."""
    from test_raster_host_and_oracle import run_oracle_frame
    from test_gpu_raster import colour_close
    ibl_mod = __import__("importlib").import_module("incarnation-renderer_amd.ibl")
    W, H, LW, LH = 256, 144, 64, 64
    view = sio.FView().reset(sponza_small, aspect=W / H)
    gscene = gpu_ctx.create_scene(sponza_small)
    osc = ob.OracleScene(sponza_small)
    try:
        sky = ibl_mod.SkyIbl(gpu_ctx, gscene, resolution=32)
        lut = ibl_mod.generate_env_brdf(gpu_ctx, LW, LH)
        hr = R.HostRenderer(gpu_ctx, gscene, W, H, ibl=R.IblInputs.from_sky(sky, lut, (LW, LH)))
        hr.render(view); gpu_ctx.synchronize()
        got = hr.download(A.SURFACE_HDR_RASTER)
        # the same frame without the specular term is darker on metallic / smooth pixels
        hr2 = R.HostRenderer(gpu_ctx, gscene, W, H, ibl=R.IblInputs(sh9=sky.info.sh9_rgba))
        hr2.render(view); gpu_ctx.synchronize()
        base = hr2.download(A.SURFACE_HDR_RASTER)
        assert (ob.unpack_r11g11b10(got).sum(-1) >= ob.unpack_r11g11b10(base).sum(-1) * 0.999).mean() > 0.99 and not np.array_equal(got, base)
        # oracle with the GPU's products as inputs
        prod = sky.download()
        lut_host = gpu_ctx.download(lut, (LH, LW), np.uint32)
        env_host, sh_host = np.ascontiguousarray(prod["envmap"]), np.ascontiguousarray(prod["sh9"])
        hs = ob.HostSurfaces(W, H, sponza_small.num_meshlets, 16 * 9 * 24, 64)
        vc = R.make_view_constants(view, W, H, R.pixel_jitter(0, W, H))
        ibl_host = R.IblInputs(sh_host.ctypes.data, env_host.ctypes.data, 32, sky.info.num_mips, lut_host.ctypes.data, (LW, LH))
        for name, d in R.build_descs(None, sponza_small, vc, hs.ptr, 0, ibl=ibl_host):
            ob.run_pass(osc, name, d)
        frac = colour_close(ob, got, hs.arr["color"], "colour target with specular IBL")
        assert frac > 0.995, frac
        hr.close(); hr2.close(); sky.close(); gpu_ctx.free(lut)
    finally:
        osc.close(); gscene.close()


@pytest.mark.gpu
def test_gpu_pathtrace_with_sky_cube(rt, ob, A, sio, gpu_ctx, helmet, helmet_oracle):
    """The sky next-event estimate of chsMain samples level 0 of the prefiltered sky cube (pathtracing.hlsl:240-243).  With
    the cube supplied, the HIP path tracer and the oracle (fed the same texels) must agree as they do with the analytic sky."""
    ibl_mod = __import__("importlib").import_module("incarnation-renderer_amd.ibl")
    W = H = 256
    view = sio.FView().reset(helmet, aspect=1.0)
    gscene = gpu_ctx.create_scene(helmet)
    try:
        sky = ibl_mod.SkyIbl(gpu_ctx, gscene, resolution=64)
        env = (sky.info.envmap_r11g11b10, sky.info.cubemap_res, sky.info.num_mips)
        pt = rt.PathTracer(gpu_ctx, gscene, W, H)
        pt.render(view, 1, envmap=env)
        g = pt.read()
        pt2 = rt.PathTracer(gpu_ctx, gscene, W, H)
        pt2.render(view, 1)
        analytic = pt2.read()
        assert not np.array_equal(g["target"], analytic["target"])                       # the texture path is really taken
        # the cube is built from 0.04 * Preetham: the two images are close, not equal
        a = g["target"].view(np.float16).astype(np.float32)[..., :3]; b = analytic["target"].view(np.float16).astype(np.float32)[..., :3]
        fin = np.isfinite(a) & np.isfinite(b)
        assert np.abs(a - b)[fin].mean() < 0.25 * np.abs(b)[fin].mean()
        host_env = np.ascontiguousarray(sky.download()["envmap"])
        d = rt.make_pathtrace_desc(view, W, H, 1, envmap=(host_env.ctypes.data, sky.info.cubemap_res, sky.info.num_mips))
        o = helmet_oracle.pathtrace(d, num_threads=8)
        assert g["stats"].total_rays() == o["stats"].total_rays()
        gt = g["target"].view(np.float16).astype(np.float32); ot = o["target"].view(np.float16).astype(np.float32)
        same = (g["target"] == o["target"]).all(axis=-1)
        rel = np.abs(gt - ot) / np.maximum(np.abs(ot), 1e-3)
        ok = same | (rel <= 2.0 ** -7).all(axis=-1)          # the tolerance of the analytic-sky parity test
        assert ok.mean() > 0.999 and same.mean() > 0.995, (ok.mean(), same.mean())
        pt.close(); pt2.close(); sky.close()
    finally:
        gscene.close()


@pytest.mark.gpu
def test_gpu_hdri_sky_mode(rt, R, ob, A, sio, gpu_ctx, helmet, helmet_oracle):
    """EnvSkyMode::HDRI (mode 0): the raster background comes from RenderJob::EnvironmentmapPass (envmap.hlsl) and the path
    tracer's miss shader is msEnvmap -- both sample level 0 of the scene's environment cube.  Any cube in the packed level
    layout serves; the prefiltered dynamic-sky cube is the one the precompute produces here."""
    from test_raster_host_and_oracle import run_oracle_frame
    ibl_mod = __import__("importlib").import_module("incarnation-renderer_amd.ibl")
    class Hdri(A.FConfig):
        EnvSkyMode = 0
        MaxSampleCount = 16
    W = H = 192
    view = sio.FView().reset(helmet, aspect=1.0)
    gscene = gpu_ctx.create_scene(helmet)
    try:
        sky = ibl_mod.SkyIbl(gpu_ctx, gscene, resolution=64)
        host_env = np.ascontiguousarray(sky.download()["envmap"])
        dev_ibl = R.IblInputs(None, sky.info.envmap_r11g11b10, sky.info.cubemap_res, sky.info.num_mips)
        host_ibl = R.IblInputs(None, host_env.ctypes.data, sky.info.cubemap_res, sky.info.num_mips)
        # raster: pass by pass and through the C++ renderer against the oracle
        rr = R.RasterRenderer(gpu_ctx, gscene, W, H, Hdri); rr.ibl = dev_ibl
        vc = rr.render(view); gpu_ctx.synchronize()
        hr = R.HostRenderer(gpu_ctx, gscene, W, H, cfg=Hdri, ibl=dev_ibl)
        hr.render(view); gpu_ctx.synchronize()
        hs = ob.HostSurfaces(W, H, helmet.num_meshlets, 16 * 9 * 24, 64)
        for name, d in R.build_descs(None, helmet, vc, hs.ptr, 0, Hdri, ibl=host_ibl):
            ob.run_pass(helmet_oracle, name, d)
        color = rr.surf.download("color")
        background = hs.arr["vis"] == 0xFFFFF000
        assert 0.2 < background.mean() < 0.9
        np.testing.assert_array_equal(color[background], hs.arr["color"][background])       # cube taps + one multiply: exact
        np.testing.assert_array_equal(hr.download(A.SURFACE_HDR_RASTER), color)
        lum = ob.unpack_r11g11b10(color[background]).sum(-1)
        assert np.isfinite(lum).all() and lum.min() > 0 and lum.std() > 0
        # the dynamic-sky mode paints something else there
        rr1 = R.RasterRenderer(gpu_ctx, gscene, W, H); rr1.render(view); gpu_ctx.synchronize()
        assert not np.array_equal(rr1.surf.download("color")[background], color[background])
        rr.close(); rr1.close(); hr.close()
        # a missing cube is an error, not a silent fallback
        rr2 = R.RasterRenderer(gpu_ctx, gscene, W, H, Hdri)
        with pytest.raises(ValueError):
            rr2.render(view)
        rr2.close()
        # path tracer: msEnvmap
        env_dev = (sky.info.envmap_r11g11b10, sky.info.cubemap_res, sky.info.num_mips)
        pt = rt.PathTracer(gpu_ctx, gscene, W, H)
        pt.render(view, 1, cfg=Hdri, envmap=env_dev)
        g = pt.read()
        d = rt.make_pathtrace_desc(view, W, H, 1, cfg=Hdri, envmap=(host_env.ctypes.data, sky.info.cubemap_res, sky.info.num_mips))
        o = helmet_oracle.pathtrace(d, num_threads=8)
        assert g["stats"].total_rays() == o["stats"].total_rays()
        gt = g["target"].view(np.float16).astype(np.float32); ot = o["target"].view(np.float16).astype(np.float32)
        same = (g["target"] == o["target"]).all(axis=-1)
        rel = np.abs(gt - ot) / np.maximum(np.abs(ot), 1e-3)
        ok = same | (rel <= 2.0 ** -7).all(axis=-1)
        assert ok.mean() > 0.999 and same.mean() > 0.995, (ok.mean(), same.mean())
        # primary rays that miss see the cube, not the analytic sky
        pt1 = rt.PathTracer(gpu_ctx, gscene, W, H); pt1.render(view, 1, envmap=env_dev)
        assert not np.array_equal(pt1.read()["target"][0, 0], g["target"][0, 0])
        with pytest.raises(rt.IncrError):
            pt.render(view, 1, cfg=Hdri)                                                  # mode 0 without a cube
        pt.close(); pt1.close(); sky.close()
    finally:
        gscene.close()


# ------------------------------------------------------------------------------------------------ seamless cube filtering (round 4)
def _random_cube(res, seed):
    """a packed cube chain (levels res .. 1) of random high-contrast R11G11B10 texels"""
    rng = np.random.default_rng(seed)
    n, parts = res, []
    while n >= 1:
        # 11/11/10-bit floats with exponents 12..17: values between 2^-3 and 4
        e = rng.integers(12, 18, (6 * n * n, 3), dtype=np.uint32)
        m = rng.integers(0, 32, (6 * n * n, 3), dtype=np.uint32)
        parts.append((e[:, 0] << 6 | m[:, 0]) | (e[:, 1] << 6 | m[:, 1]) << 11 | (e[:, 2] << 5 | (m[:, 2] & 31)) << 22)
        n >>= 1
    return np.ascontiguousarray(np.concatenate(parts).astype(np.uint32))


def _cube_point(face, sc, tc):
    """cube_face_uv inverted: the point of the cube surface with face coordinates (sc, tc) in [-1, 1]"""
    return np.array([(1, -tc, -sc), (-1, -tc, sc), (sc, 1, tc), (sc, -1, -tc), (sc, -tc, 1), (-sc, -tc, -1)][face], dtype=np.float64)


def _sample(ob, flat, res, mips, d, level=0.0):
    out = (C.c_float * 3)()
    ob.lib().orc_cube_sample(flat.ctypes.data, res, mips, (C.c_float * 3)(*[float(np.float32(v)) for v in d]), float(level), out)
    return np.array(out[:], dtype=np.float64)


def test_oracle_cube_filtering_is_seamless(ob):
    """D3D11.3 TextureCube edge handling (consumers: pathtracing.hlsl:243, 308; sky-lighting.hlsl:123; envmap.hlsl:47): a bilinear tap beyond a
    face edge comes from the adjacent face.  Checked against an INDEPENDENT adjacency: the neighbour of an edge texel is the texel of another
    face whose centre is nearest (in 3-D) to the point one texel step beyond the edge.  On the edge itself, level with a texel centre, the
    sample must be the mean of the two texels either side; approaching an edge from its two faces gives the same value; at a cube corner
    the sample is the mean of the three corner texels (the missing fourth tap = mean of the other three)."""
    res, mips = 8, 4
    flat = _random_cube(res, 7)
    lv0 = ob.unpack_r11g11b10(levels_cube(flat, res)[0]).astype(np.float64)              # [6, n, n, 3]
    n = res
    # all texel centres of the cube, for the brute-force neighbour search
    centres = np.array([_cube_point(f, (2 * x + 1 - n) / n, (2 * y + 1 - n) / n) for f in range(6) for y in range(n) for x in range(n)])
    index = [(f, y, x) for f in range(6) for y in range(n) for x in range(n)]
    checked = 0
    for f in range(6):
        for side in range(4):                       # x beyond n-1, x beyond 0, y beyond n-1, y beyond 0
            for k in range(n):
                x, y = ((n - 1, k), (0, k), (k, n - 1), (k, 0))[side]
                dx, dy = ((1, 0), (-1, 0), (0, 1), (0, -1))[side]
                beyond = _cube_point(f, (2 * (x + dx) + 1 - n) / n, (2 * (y + dy) + 1 - n) / n)
                dist = np.linalg.norm(centres - beyond, axis=1)
                dist[[i for i, (ff, _, _) in enumerate(index) if ff == f]] = np.inf
                nf, ny, nx = index[int(np.argmin(dist))]
                # the direction of the edge point level with the texel centre
                sc, tc = (2 * x + 1 - n) / n + dx / n, (2 * y + 1 - n) / n + dy / n
                got = _sample(ob, flat, res, mips, _cube_point(f, sc, tc))
                want = 0.5 * lv0[f, y, x] + 0.5 * lv0[nf, ny, nx]
                np.testing.assert_allclose(got, want, rtol=2e-6, err_msg="face %d side %d texel %d: neighbour (%d, %d, %d)" % (f, side, k, nf, nx, ny))
                checked += 1
    assert checked == 6 * 4 * n
    # continuity: the same edge point approached from either face
    rng = np.random.default_rng(3)
    for f in range(6):
        for side in range(4):
            for _ in range(8):
                t = rng.uniform(-0.98, 0.98)
                sc, tc = ((1.0, t), (-1.0, t), (t, 1.0), (t, -1.0))[side]
                p = _cube_point(f, sc, tc)
                inward = _cube_point(f, 0.0, 0.0)                       # the face centre: moving towards it stays on face f
                a = _sample(ob, flat, res, mips, p + 1e-4 * (inward - p))
                axis = int(np.argmax(np.abs(inward)))
                q = p.copy(); q[axis] *= (1 - 1e-4)                      # shrinking the major component: the neighbouring face becomes the major one
                b = _sample(ob, flat, res, mips, q)
                # (the two approaches land on neighbouring 1/256-texel steps of the fixed-point weights: up to one step of the local contrast apart)
                assert np.abs(a - b).max() <= 8e-3 * max(a.max(), b.max()), (f, side, t, a, b)
    # corners: mean of the three corner texels, on every level
    for lvl, nn in enumerate((8, 4, 2, 1)):
        lv = ob.unpack_r11g11b10(levels_cube(flat, res)[lvl]).astype(np.float64)
        for sx in (-1, 1):
            for sy in (-1, 1):
                for sz in (-1, 1):
                    got = _sample(ob, flat, res, mips, (sx, sy, sz), level=lvl)
                    three = []
                    for f in range(6):
                        for (yy, xx) in ((0, 0), (0, nn - 1), (nn - 1, 0), (nn - 1, nn - 1)):
                            c = _cube_point(f, (2 * xx + 1 - nn) / nn, (2 * yy + 1 - nn) / nn)
                            if np.sign(c[0]) == sx and np.sign(c[1]) == sy and np.sign(c[2]) == sz and (nn > 1 or True):
                                if nn == 1:
                                    continue
                                three.append(lv[f, yy, xx])
                    if nn == 1:          # 1 x 1 faces: the three faces that meet at the corner
                        three = [lv[0 if sx > 0 else 1, 0, 0], lv[2 if sy > 0 else 3, 0, 0], lv[4 if sz > 0 else 5, 0, 0]]
                    assert len(three) == 3
                    np.testing.assert_allclose(got, np.mean(three, axis=0), rtol=3e-6, err_msg="corner (%d, %d, %d) level %d" % (sx, sy, sz, lvl))


@pytest.mark.gpu
def test_gpu_cube_edges_and_corners_equal_the_oracle(rt, R, ob, A, sio, gpu_ctx):
    """The device sampler on every edge and corner of a random cube: EnvSkyMode 0 background (envmap.hlsl:47 -> cube level 0) of an empty
    scene seen from the origin towards each of the 8 cube corners (a corner view holds the three edges that meet there) and the 6 face
    centres, bit for bit against the oracle's pass; the path tracer's msEnvmap (pathtracing.hlsl:308) on the same views."""
    res, mips = 8, 4
    flat = _random_cube(res, 11)
    dev = gpu_ctx.buffer(flat.nbytes); gpu_ctx.upload(dev, flat)

    class Hdri(A.FConfig):
        EnvSkyMode = 0
        EnableTAA = False
    s = sio.PackedScene()
    gscene = gpu_ctx.create_scene(s)
    osc = ob.OracleScene(s)
    W = H = 96
    try:
        looks = [(sx, sy, sz) for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)] + [(1, 0, 0), (-1, 0, 0), (0, 1, 0.001), (0, -1, 0.001), (0, 0, 1), (0, 0, -1)]
        for look in looks:
            view = sio.FView()
            view.position = np.zeros(3, dtype=np.float32)
            view.look = np.array(look, dtype=np.float32)
            view.right = np.cross(np.array([0.0, 1.0, 0.0]), view.look).astype(np.float32)
            view.update_view_transform()
            view.projection_transform = sio.reverse_z_infinite_perspective_fov_lh(1.2, 1.0, A.FConfig.CameraNearPlane)
            rr = R.RasterRenderer(gpu_ctx, gscene, W, H, Hdri); rr.ibl = R.IblInputs(None, dev, res, mips)
            vc = rr.render(view); gpu_ctx.synchronize()
            hs = ob.HostSurfaces(W, H, 0, 16 * 9 * 24, 64)
            for name, d in R.build_descs(None, s, vc, hs.ptr, 0, Hdri, ibl=R.IblInputs(None, flat.ctypes.data, res, mips)):
                ob.run_pass(osc, name, d)
            color = rr.surf.download("color")
            np.testing.assert_array_equal(color, hs.arr["color"], err_msg="look %s" % (look,))
            assert len(np.unique(color)) > 50                           # texels of several faces, filtered
            rr.close()
            pt = rt.PathTracer(gpu_ctx, gscene, W, H)
            pt.render(view, 1, cfg=Hdri, envmap=(dev, res, mips)); g = pt.read()
            o = osc.pathtrace(rt.make_pathtrace_desc(view, W, H, 1, cfg=Hdri, envmap=(flat.ctypes.data, res, mips)), num_threads=8)
            # the camera ray goes through sin / cos (lens sample): a direction may differ by an ulp between device and host, the filtered cube
            # value with it -- continuous across texel and face boundaries now, so every pixel stays within the colour tolerance
            gt = g["target"].view(np.float16).astype(np.float32); ot = o["target"].view(np.float16).astype(np.float32)
            same = (g["target"] == o["target"]).all(axis=-1)
            assert same.mean() > 0.999, (look, same.mean())
            assert (np.abs(gt - ot) <= 2.0 ** -7 * np.maximum(np.abs(ot), 1e-3)).all(), "msEnvmap, look %s" % (look,)
            pt.close()
    finally:
        gpu_ctx.free(dev); osc.close(); gscene.close()


def test_golden_sky_ibl(ob, sky16):
    """tests/golden/sky_ibl_16.npz (tools/make_golden.py): the oracle's FScene::UpdateDynamicSky products at 16^2 -- lat-long chain, source
    cube, prefiltered cube (1024 trilinear cube taps per texel, many across face edges) and SH9 -- reproduce bit for bit."""
    from conftest import ROOT
    g = np.load(os.path.join(ROOT, "tests/golden/sky_ibl_16.npz"))
    for k in ("latlong", "source_cube", "envmap"):
        np.testing.assert_array_equal(sky16[k], g[k], err_msg=k)
    np.testing.assert_array_equal(sky16["sh9"].view(np.uint32), g["sh9"].view(np.uint32))
