"""Texture LOD of the raster passes (material/common.hlsli:6-8 TEX_SAMPLE = Sample; scene.cpp:1072-1090 mip generation;
scene.cpp:1157-1320 vMF chains for normal + metallic-roughness pairs): the device mip chains equal the oracle's byte for byte,
the G-buffer sampled through them equals the oracle's bit for bit, and the levels are really used (a grazing floor differs
from level-0 sampling, minified regions read coarse levels)."""
import copy

import numpy as np
import pytest

from test_gpu_raster import gpu_frame
from test_raster_host_and_oracle import run_oracle_frame

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("which", ["helmet", "sponza_small"])
def test_mip_chains_equal_the_oracle(request, gpu_ctx, ob, A, which):
    scene = request.getfixturevalue(which)
    gs = gpu_ctx.create_scene(scene)
    osc = ob.OracleScene(scene)
    pairs = set()
    for m in scene.materials:
        if m.normalTextureIndex >= 0 and m.metallicRoughnessTextureIndex >= 0:
            pairs.update((m.normalTextureIndex, m.metallicRoughnessTextureIndex))
    assert pairs, "scene has a vMF pair"
    for ti in range(len(scene.textures)):
        w, h, mips, chain = gs.texture_chain(ti)
        omips, ochain = osc.texture_chain(ti, chain.size)
        assert mips == omips and mips == max(1, min(int(np.log2(w)), int(np.log2(h))) - 1), (ti, w, h, mips, omips)
        np.testing.assert_array_equal(chain, ochain, err_msg="texture %d (%dx%d, %d levels, %s)" % (ti, w, h, mips, "vMF" if ti in pairs else "box"))
        # level 0 is the texture as handed over
        np.testing.assert_array_equal(chain[:w * h], np.ascontiguousarray(scene.textures[ti][0]).view(np.uint32).ravel())
    osc.close(); gs.close()


def _floor_scene(synth, sio, R, A, texture_size=256):
    """A large textured floor seen at a grazing angle: strong minification towards the horizon, anisotropic footprints."""
    s = sio.PackedScene()
    P, N, T4, UV, I = synth.assemble([(synth.plane([-40, 0, -40], [80, 0, 0], [0, 0, 80], 0.0, (1, 1)), 1.0, 1.0, (24.0, 24.0))], 800)
    prims = synth._pack_geometry(s, [(P, N, T4, UV, I)], lambda k: 0)
    s.mesh_names.append("floor"); s.mesh_primitives.append(prims); s.mesh_transforms.append(sio.mat_identity())
    base, mr, nm = synth.make_material_textures(synth.SPONZA_SEED, 3, texture_size, False)
    yy, xx = np.mgrid[0:texture_size, 0:texture_size]
    base = base.copy(); base[..., :3] = np.where(((xx + yy) & 1)[..., None] == 1, 240, 15).astype(np.uint8)      # 1-texel checker: every level >= 1 is one constant
    m = A.Material()
    for n in A._MAT_TEX:
        setattr(m, n + "TextureIndex", -1); setattr(m, n + "SamplerIndex", -1)
    for i in range(3):
        m.baseColorFactor[i] = 1.0; m.emissiveFactor[i] = 0.0
    m.metallicFactor, m.roughnessFactor, m.aoStrength = 1.0, 1.0, 1.0
    m.baseColorTextureIndex = s.add_texture(base, A.TEX_RGBA8_SRGB)
    m.metallicRoughnessTextureIndex = s.add_texture(mr, A.TEX_BGRA8_SWIZZLE_RG)
    m.normalTextureIndex = s.add_texture(nm, A.TEX_RG_SNORM)
    m.alphaMode, m.doubleSided = A.ALPHA_OPAQUE, 1
    s.materials.append(m)
    cam = sio.mat_create_world((0.0, 1.5, -30.0), (0.0, -0.12, 1.0), (0, 1, 0))
    s.cameras.append({"transform": cam, "yfov": 0.8, "aspect": 16 / 9})
    synth.add_light_rig(s, synth.SPONZA_SEED, 4, ((-10.0, 10.0), (1.0, 4.0), (-10.0, 10.0)))
    return R.generate_meshlets(s)


def test_gbuffer_through_mip_chain_equals_oracle_and_uses_the_levels(gpu_ctx, rt, R, sio, synth, A, ob):
    W, H = 320, 180
    s = _floor_scene(synth, sio, R, A)
    view = sio.FView().reset(s, aspect=W / H)
    gs = gpu_ctx.create_scene(s)
    osc = ob.OracleScene(s)

    class Cfg(A.FConfig):
        EnableTAA = False
    (vc, g), = gpu_frame(rt, R, gpu_ctx, gs, view, W, H, Cfg, frames=1)
    _, hs = run_oracle_frame(ob, R, A, s, osc, view, W, H, cfg=Cfg)
    o = hs.arr
    np.testing.assert_array_equal(g["vis"], o["vis"])
    for name in ("gbuffer_basecolor", "gbuffer_normals", "gbuffer_mra"):
        np.testing.assert_array_equal(g[name], o[name], err_msg=name)
    covered = g["vis"] != 0xFFFFF000
    assert covered.mean() > 0.3
    # level 0 is a 1-texel checker, so every coarser level is ONE constant: wherever the footprint exceeds two texels (LOD >= 1,
    # the far part of the floor) the G-buffer base colour must be exactly that constant, while level-0 sampling would alias
    # between the two checker values; near the camera (LOD < 1) the checker is still visible
    w, h, mips, chain = gs.texture_chain(0)
    assert mips >= 5 and len(np.unique(chain[w * h:] & 0xffffff)) == 1
    lin = ob.lib()     # the G-buffer stores the LINEAR colour: decode the constant through the oracle's sampler
    import ctypes as C
    lin.orc_sample_texture_grad.argtypes = [C.c_void_p, C.c_int] + [C.POINTER(C.c_float)] * 4
    out4 = (C.c_float * 4)()
    lin.orc_sample_texture_grad(osc.handle, 0, (C.c_float * 2)(0.3, 0.7), (C.c_float * 2)(0.1, 0.0), (C.c_float * 2)(0.0, 0.1), out4)
    const = int(np.floor(np.float32(out4[0]) * np.float32(255.0) + np.float32(0.5)))
    bc = g["gbuffer_basecolor"] & 0xff
    rows = np.nonzero(covered.all(axis=1))[0]
    order = rows[np.argsort(g["depth"][rows, W // 2])]          # reverse Z: smallest depth = farthest
    far, near = order[:8], order[-2:]
    assert (bc[far] == const).all(), (const, np.unique(bc[far]))
    assert len(np.unique(bc[near])) > 4                          # LOD < 1 next to the camera: level 0 still contributes
    osc.close(); gs.close()


def test_sample_grad_matches_oracle_pointwise(gpu_ctx, rt, R, sio, synth, A, ob):
    """Frame-level parity on the regular scenes with the LOD path active on every material fetch: helmet (2048^2 textures, LOD 3-4
    at this size) -- G-buffers, colour and the alpha-tested visibility ids."""
    W, H = 256, 256
    scene = R.generate_meshlets(sio.load_gltf(__import__("os").path.join(__import__("conftest").ROOT, "assets/damaged-helmet/DamagedHelmet.gltf")))
    view = sio.FView().reset(scene, aspect=1.0)
    gs = gpu_ctx.create_scene(scene); osc = ob.OracleScene(scene)
    (vc, g), = gpu_frame(rt, R, gpu_ctx, gs, view, W, H, A.FConfig, frames=1)
    _, hs = run_oracle_frame(ob, R, A, scene, osc, view, W, H)
    for name in ("vis", "gbuffer_basecolor", "gbuffer_normals", "gbuffer_mra"):
        np.testing.assert_array_equal(g[name], hs.arr[name], err_msg=name)
    w, h, mips, chain = gs.texture_chain(0)
    assert (w, h, mips) == (2048, 2048, 10)
    osc.close(); gs.close()


def test_max_anisotropy_setter_equals_a_scene_created_with_it(gpu_ctx, rt, R, sio, synth, A):
    """incr_scene_set_max_anisotropy switches the material sampler of later frames: 1 = the isotropic rule, and the floor scene's grazing
    part really depends on it."""
    W, H = 320, 180
    s = _floor_scene(synth, sio, R, A)
    view = sio.FView().reset(s, aspect=W / H)

    class Cfg(A.FConfig):
        EnableTAA = False
    frames = {}
    for created_with in (0, 1):
        s.max_anisotropy = created_with; s._desc = None
        gs = gpu_ctx.create_scene(s)
        (vc, g), = gpu_frame(rt, R, gpu_ctx, gs, view, W, H, Cfg, frames=1)
        frames[created_with] = g["gbuffer_basecolor"].copy()
        if created_with == 0:
            rt.check(gpu_ctx.lib.incr_scene_set_max_anisotropy(gpu_ctx.handle, gs.handle, 1))
            (vc, g), = gpu_frame(rt, R, gpu_ctx, gs, view, W, H, Cfg, frames=1)
            switched = g["gbuffer_basecolor"].copy()
            assert gpu_ctx.lib.incr_scene_set_max_anisotropy(gpu_ctx.handle, gs.handle, 17) != 0
        gs.close()
    np.testing.assert_array_equal(switched, frames[1])
    assert (frames[0] != frames[1]).mean() > 0.05
