"""Block-compressed texture chains at the boundary (include/incr_types.h formats 4-7; the reference uploads BC3_UNORM_SRGB / BC4_UNORM /
BC5_UNORM / BC5_SNORM chains, src/scene.cpp:902-904, 932-933, 1100, 1162-1163, 1401): the oracle's block expansion against hand-built
blocks, bc.py's encoder against it, the .dds container, and -- on the GPU -- the library's decode and every pass that samples the
result against the oracle."""
import ctypes as C
import importlib
import os

import numpy as np
import pytest

PKG = "incarnation-renderer_amd"


@pytest.fixture(scope="module")
def bc():
    return importlib.import_module(PKG + ".bc")


def _one_texture_scene(synth, A, image):
    s = synth.triangle_soup(3, 16)
    s.textures = []
    s.add_texture_bc(image)
    for m in s.materials:
        for n in A._MAT_TEX:
            setattr(m, n + "TextureIndex", -1)
    s._desc = None
    return s


def _oracle_chain(ob, synth, A, image):
    osc = ob.OracleScene(_one_texture_scene(synth, A, image))
    n = sum(max(image.width >> l, 1) * max(image.height >> l, 1) for l in range(image.mip_levels))
    mips, words = osc.texture_chain(0, n)
    osc.close()
    assert mips == image.mip_levels
    return words


def _bc4_block(e0, e1, selectors):
    bits = 0
    for i, s in enumerate(selectors):
        bits |= int(s) << (3 * i)
    return bytes([e0 & 0xff, e1 & 0xff]) + bits.to_bytes(6, "little")


def test_bc4_bc5_palettes_are_kept_exactly(ob, synth, A, bc):
    # eight-value mode (e0 > e1): selectors 0..7 in the first eight texels; six-value mode (e0 <= e1) with the 0 / 1 specials in the rest
    blk = _bc4_block(200, 60, [0, 1, 2, 3, 4, 5, 6, 7] * 2)
    words = _oracle_chain(ob, synth, A, bc.BcImage(np.frombuffer(blk, np.uint8), 4, 4, 1, A.TEX_BC4_UNORM))
    want = [200 * 35, 60 * 35] + [5 * ((7 - k) * 200 + k * 60) for k in range(1, 7)]
    assert [int(w) & 0xffff for w in words[:8]] == want and all(int(w) >> 16 == 0 for w in words)
    assert want[2] / 8925 == pytest.approx((6 * 200 + 60) / 7 / 255, rel=1e-12)                       # (6 r0 + r1) / 7, D3D11.3 19.5
    blk = _bc4_block(60, 200, [0, 1, 2, 3, 4, 5, 6, 7] * 2)
    words = _oracle_chain(ob, synth, A, bc.BcImage(np.frombuffer(blk, np.uint8), 4, 4, 1, A.TEX_BC4_UNORM))
    assert [int(w) & 0xffff for w in words[:8]] == [60 * 35, 200 * 35] + [7 * ((5 - k) * 60 + k * 200) for k in range(1, 5)] + [0, 8925]
    # BC5 SNORM: two blocks, signed endpoints, -128 reads as -127, specials are -1 and +1
    r = _bc4_block(100, -50, [0, 1, 2, 7] * 4); g = _bc4_block(-128, 127, [0, 1, 6, 7] * 4)
    words = _oracle_chain(ob, synth, A, bc.BcImage(np.frombuffer(r + g, np.uint8), 4, 4, 1, A.TEX_BC5_SNORM))
    lo = words[:4].astype(np.uint32) & 0xffff; hi = words[:4].astype(np.uint32) >> 16
    s16 = lambda v: [int(np.int16(np.uint16(x))) for x in v]
    assert s16(lo) == [3500, -1750, 5 * (6 * 100 - 50), 5 * (100 - 6 * 50)]
    assert s16(hi) == [-127 * 35, 127 * 35, -4445, 4445]


def test_bc3_block_expands_to_rgba8(ob, synth, A, bc):
    # colour endpoints pure 5:6:5 red (31, 0, 0) and blue (0, 0, 31); alpha block 255 / 0 in eight-value mode
    c0, c1 = 31 << 11, 31
    sel = 0
    for i in range(16):
        sel |= (i % 4) << (2 * i)
    colour = bytes([c0 & 0xff, c0 >> 8, c1 & 0xff, c1 >> 8]) + sel.to_bytes(4, "little")
    alpha = _bc4_block(255, 0, [0, 1, 2, 7] * 4)
    words = _oracle_chain(ob, synth, A, bc.BcImage(np.frombuffer(alpha + colour, np.uint8), 4, 4, 1, A.TEX_BC3_SRGB))
    px = [(int(w) & 0xff, (int(w) >> 8) & 0xff, (int(w) >> 16) & 0xff, int(w) >> 24) for w in words[:4]]
    assert px[0] == (255, 0, 0, 255) and px[1] == (0, 0, 255, 0)
    assert px[2] == ((2 * 255 + 1) // 3, 0, (255 + 1) // 3, round(6 * 255 / 7))                      # 2/3 c0 + 1/3 c1, rounded to nearest
    assert px[3] == ((255 + 1) // 3, 0, (2 * 255 + 1) // 3, round(255 / 7))


@pytest.mark.parametrize("fmt_name", ["TEX_BC3_SRGB", "TEX_BC4_UNORM", "TEX_BC5_UNORM", "TEX_BC5_SNORM"])
def test_encoder_round_trip_and_ragged_sizes(ob, synth, A, bc, fmt_name):
    """bc.compress -> oracle expansion stays within the format's quantisation of the source (smooth image), for a size that is not a
    multiple of four on any level."""
    fmt = getattr(A, fmt_name)
    h, w = 22, 37
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([(xx * 6) % 256, (yy * 11) % 256, ((xx + yy) * 4) % 256, 255 - (xx * 5) % 200], -1).astype(np.uint8)
    image = bc.compress(img, fmt, levels=3)
    assert image.mip_levels == 3 and image.blocks.size == bc.chain_bytes(w, h, 3, fmt)
    words = _oracle_chain(ob, synth, A, image)[:w * h].reshape(h, w)
    if fmt == A.TEX_BC3_SRGB:
        got = np.stack([(words >> s) & 0xff for s in (0, 8, 16, 24)], -1).astype(np.int32)
        err = np.abs(got - img.astype(np.int32))
        assert err[..., 3].max() <= 20 and np.median(err[..., :3]) <= 8             # 3-bit alpha ramp, 5:6:5 endpoints + 2-bit ramp per block
    else:
        lo = (words & 0xffff).astype(np.int64); hi = (words >> 16).astype(np.int64)
        if fmt == A.TEX_BC5_SNORM:
            lo = np.where(lo >= 32768, lo - 65536, lo); hi = np.where(hi >= 32768, hi - 65536, hi)
            want = (2.0 * img[..., :2].astype(np.float64) / 255.0 - 1.0)
            got = np.stack([lo, hi], -1) / 4445.0
        elif fmt == A.TEX_BC5_UNORM:
            want = img[..., [2, 1]].astype(np.float64) / 255.0                       # R = metal (the glTF image's B), G = roughness
            got = np.stack([lo, hi], -1) / 8925.0
        else:
            want = img[..., :1].astype(np.float64) / 255.0
            got = (lo / 8925.0)[..., None]
            assert (hi == 0).all()
        assert np.abs(got - want).max() <= 0.09 and np.abs(got - want).mean() <= 0.03


def test_dds_container_round_trip(tmp_path, A, bc):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (16, 32, 4), dtype=np.uint8)
    for fmt in (A.TEX_BC3_SRGB, A.TEX_BC4_UNORM, A.TEX_BC5_UNORM, A.TEX_BC5_SNORM):
        a = bc.compress(img, fmt)
        assert a.mip_levels == 3                                  # 16 x 32 -> 8 x 16 -> 4 x 8 (down to 4, scene.cpp:1072-1081)
        p = os.path.join(tmp_path, "t%d.dds" % fmt)
        bc.write_dds(p, a)
        b = bc.read_dds(p)
        assert (b.width, b.height, b.mip_levels, b.format) == (32, 16, 3, fmt) and np.array_equal(a.blocks, b.blocks)
        raw = open(p, "rb").read()
        assert raw[:4] == b"DDS " and raw[84:88] == b"DX10" and len(raw) == 148 + a.blocks.size


def test_scene_validation_of_compressed_chains(rt, synth, A, bc):
    lib = rt.load()
    lib.incr_scene_validate_host.restype = C.c_int
    lib.incr_scene_validate_host.argtypes = [C.POINTER(A.SceneDesc), C.POINTER(A.SceneInfo)]
    img = np.zeros((8, 8, 4), np.uint8)
    s = _one_texture_scene(synth, A, bc.compress(img, A.TEX_BC5_UNORM))
    d = s.to_desc()
    assert d.textures[0].mip_levels == 2 and d.textures[0].format == A.TEX_BC5_UNORM
    assert lib.incr_scene_validate_host(C.byref(d), C.byref(A.SceneInfo())) == 0, lib.incr_last_error()


# ------------------------------------------------------------------------------------------------ GPU
def _compressed_sponza(synth, R):
    s = synth.sponza_stat(texture_size=64, tri_scale=1.0 / 16.0)
    s.compress_textures()
    return R.generate_meshlets(s)


@pytest.mark.gpu
def test_gpu_decode_equals_the_oracle(gpu_ctx, ob, synth, R, A):
    scene = _compressed_sponza(synth, R)
    assert {f for _, f in scene.textures} == {A.TEX_BC3_SRGB, A.TEX_BC5_SNORM, A.TEX_BC5_UNORM}
    gs = gpu_ctx.create_scene(scene); osc = ob.OracleScene(scene)
    for ti, (img, fmt) in enumerate(scene.textures):
        w, h, mips, chain = gs.texture_chain(ti)
        omips, ochain = osc.texture_chain(ti, chain.size)
        assert (w, h, mips) == (img.width, img.height, img.mip_levels) and omips == mips
        np.testing.assert_array_equal(chain, ochain, err_msg="texture %d format %d" % (ti, fmt))
    osc.close(); gs.close()


@pytest.mark.gpu
def test_gpu_frames_on_compressed_textures_match_the_oracle(gpu_ctx, rt, ob, synth, R, A, sio):
    """The raster frame (anisotropic Sample of the expanded chains: ids through the alpha test, G-buffers bit-exact) and the path tracer
    (SampleLevel 0) on a scene whose every texture arrives block-compressed."""
    from test_gpu_raster import gpu_frame
    from test_raster_host_and_oracle import run_oracle_frame
    scene = _compressed_sponza(synth, R)
    W, H = 256, 144
    view = sio.FView().reset(scene, aspect=W / H)
    gs = gpu_ctx.create_scene(scene); osc = ob.OracleScene(scene)

    class Cfg(A.FConfig):
        EnableTAA = False
    (vc, g), = gpu_frame(rt, R, gpu_ctx, gs, view, W, H, Cfg, frames=1)
    _, hs = run_oracle_frame(ob, R, A, scene, osc, view, W, H, cfg=Cfg)
    for name in ("vis", "gbuffer_basecolor", "gbuffer_normals", "gbuffer_mra"):
        np.testing.assert_array_equal(g[name], hs.arr[name], err_msg=name)
    # path tracer: one sample, every ray category equal, colours within the suite's tolerance on nearly every pixel
    from test_gpu_pathtrace import pixel_agreement, stats_tuple
    pt = rt.PathTracer(gpu_ctx, gs, 128, 72)
    view2 = sio.FView().reset(scene, aspect=128 / 72)
    from parity_helpers import accounted_parity
    got, want, _ = accounted_parity(rt, pt, osc, view2, 128, 72, 1, cap=3e-4, label="block-compressed scene", min_outliers_allowed=3)
    assert got["stats"].camera_rays == 128 * 72
    ok, same = pixel_agreement(got["target"], want["target"])
    assert same >= 0.95, (ok, same)
    pt.close(); osc.close(); gs.close()


def test_loader_takes_the_reference_texture_cache(tmp_path, sio, A, bc, helmet):
    """`<cache>/<image stem>.dds` (what the reference's LoadTexture leaves behind, scene.cpp:1103-1115) replaces the decoded image when it
    holds the format the slot compresses to; other files are ignored."""
    import json
    from conftest import ROOT
    path = os.path.join(ROOT, "assets/damaged-helmet/DamagedHelmet.gltf")
    g = json.load(open(path))
    m = helmet.materials[0]
    small = lambda ti: np.ascontiguousarray(helmet.textures[ti][0][::32, ::32])          # 64 x 64 stand-ins: any valid chain will do
    stem = lambda ti: os.path.splitext(os.path.basename(g["images"][g["textures"][ti]["source"]]["uri"]))[0]
    gl = g["materials"][0]
    base_t, normal_t, mr_t = gl["pbrMetallicRoughness"]["baseColorTexture"]["index"], gl["normalTexture"]["index"], gl["pbrMetallicRoughness"]["metallicRoughnessTexture"]["index"]
    bc.write_dds(os.path.join(tmp_path, stem(base_t) + ".dds"), bc.compress(small(m.baseColorTextureIndex), A.TEX_BC3_SRGB))
    bc.write_dds(os.path.join(tmp_path, stem(normal_t) + ".dds"), bc.compress(small(m.normalTextureIndex), A.TEX_BC5_SNORM))
    bc.write_dds(os.path.join(tmp_path, stem(mr_t) + ".dds"), bc.compress(small(m.metallicRoughnessTextureIndex), A.TEX_BC4_UNORM))      # wrong format for the slot
    s = sio.load_gltf(path, texture_cache_dir=str(tmp_path))
    m2 = s.materials[0]
    assert s.textures[m2.baseColorTextureIndex][1] == A.TEX_BC3_SRGB and s.textures[m2.baseColorTextureIndex][0].width == 64
    assert s.textures[m2.normalTextureIndex][1] == A.TEX_BC5_SNORM
    assert s.textures[m2.metallicRoughnessTextureIndex][1] == A.TEX_BGRA8_SWIZZLE_RG and s.textures[m2.metallicRoughnessTextureIndex][0].shape[0] == 2048
    assert s.textures[m2.emissiveTextureIndex][1] == A.TEX_RGBA8_SRGB
    d = s.to_desc()
    assert d.textures[m2.baseColorTextureIndex].mip_levels == 5 and d.textures[m2.normalTextureIndex].format == A.TEX_BC5_SNORM
