"""Texture ingest, GPU step: vMF prefilter of normal + metallic-roughness maps.
CPU tests pin the oracle restatement of prefilter-normal-roughness.hlsl by closed forms; GPU tests compare libincr_hip.so's
summed-pyramid kernels with it.  Tolerance (float path, RGBA8 result): the pyramid re-associates the footprint sum, so a
texel may land on the other side of a rounding boundary -- at most 1 LSB, on at most 1 % of texels; mips 0 and 1 add in
the reference's order and must be byte-identical."""
import importlib
import math

import numpy as np
import pytest


def make_maps(w, h, seed, bump=0.35):
    """Random but plausible content: normals around +z, roughness and metalness anywhere, junk in the unused channels."""
    rng = np.random.default_rng(seed)
    v = rng.normal(0, bump, (h, w, 3)).astype(np.float32); v[..., 2] = 1.0
    v /= np.linalg.norm(v, axis=-1, keepdims=True)
    n = np.zeros((h, w, 4), np.uint8); n[..., :3] = np.clip(np.rint((v * 0.5 + 0.5) * 255), 0, 255); n[..., 3] = rng.integers(0, 256, (h, w))
    m = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    return n, m


def test_mip_count_and_level0(ob):
    L = ob.lib()
    assert [L.orc_compressed_mip_count(w, h) for w, h in [(4, 4), (3, 8), (8, 4), (64, 64), (2048, 1024), (37, 21)]] == [1, 0, 1, 5, 9, 3]
    n, m = make_maps(16, 8, 1)
    on, om = ob.prefilter_normal_roughness(n, m)
    assert [a.shape for a in on] == [(8, 16, 4), (4, 8, 4)]
    assert np.array_equal(on[0], n)                                    # hlsl:40
    assert np.array_equal(om[0], m[..., [2, 1, 0, 3]])                 # hlsl:41  .bgra


def test_flat_map_keeps_normal_and_roughness(ob):
    """Identical texels: the mean vMF vector is the texel's own, so every level must return the same normal and (up to the
    closed-form round trip r -> 1/lambda, graphicrants 'vMF part 3') nearly the same roughness; metalness is a box filter."""
    for rough in (64, 128, 200, 255):
        n = np.zeros((32, 32, 4), np.uint8); n[...] = (128, 128, 255, 255)
        m = np.zeros((32, 32, 4), np.uint8); m[..., 1] = rough; m[..., 2] = 77
        on, om = ob.prefilter_normal_roughness(n, m)
        a = rough / 255.0
        il = 0.5 * a * a
        coth = (1 + math.exp(-2 / il)) / (1 - math.exp(-2 / il)) if il > 0.1 else 1.0
        r = coth - il
        want = math.sqrt(2 * (1 / r) * (1 - r * r) / (3 - r * r))
        for k in range(1, len(on)):
            assert np.all(on[k][..., :3] == on[1][0, 0, :3]) and np.all(on[k][..., 3] == 0)
            assert abs(int(on[k][0, 0, 0]) - 128) <= 1 and abs(int(on[k][0, 0, 1]) - 128) <= 1 and on[k][0, 0, 2] == 255
            assert np.all(om[k][..., 0] == 77) and np.all(om[k][..., 2:] == 0)
            assert np.all(np.abs(om[k][..., 1].astype(int) - round(want * 255)) <= 1)
        assert abs(want - a) < 0.12                                    # the round trip is approximate by construction


def test_bumpy_map_widens_roughness(ob):
    """Normal variation inside a footprint shortens the mean vector: roughness must rise with the level (the point of the pass)."""
    n, m = make_maps(64, 64, 3, bump=0.6)
    m[..., 1] = 40
    on, om = ob.prefilter_normal_roughness(n, m)
    means = [float(l[..., 1].mean()) for l in om]
    assert means[1] > means[0] + 10 and all(means[k + 1] >= means[k] - 0.5 for k in range(1, len(means) - 1))
    # opposing normals cancel: r -> 0, normalize(0) is NaN and the UNORM store writes 0; roughness saturates
    n2 = np.zeros((4, 4, 4), np.uint8); n2[..., :3] = (255, 128, 128); n2[:, 1::2, 0] = 0; n2[:, 1::2, 1:3] = 127
    m2 = np.zeros((4, 4, 4), np.uint8)
    on2, om2 = ob.prefilter_normal_roughness(np.tile(n2, (2, 2, 1)), np.tile(m2, (2, 2, 1)))
    assert np.all(om2[1][..., 1] == 255)


def _compare(got, want, exact_levels=2):
    for k, (g, w) in enumerate(zip(got, want)):
        assert g.shape == w.shape
        d = np.abs(g.astype(int) - w.astype(int))
        if k < exact_levels:
            assert d.max() == 0, "level %d differs (max %d)" % (k, d.max())
        else:
            assert d.max() <= 1, "level %d: max diff %d" % (k, d.max())
            assert (d != 0).mean() <= 0.01, "level %d: %.3f %% texels differ" % (k, 100 * (d != 0).mean())


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,seed", [(64, 32, 11), (37, 21, 12), (256, 256, 13), (4, 4, 14), (8, 5, 15), (130, 66, 16)])
def test_gpu_prefilter_matches_oracle(rt, ob, gpu_ctx, w, h, seed):
    content = importlib.import_module("incarnation-renderer_amd.content")
    n, m = make_maps(w, h, seed)
    gn, gm = content.prefilter_normal_roughness(gpu_ctx, n, m)
    on, om = ob.prefilter_normal_roughness(n, m)
    assert len(gn) == len(on) == content.mip_count(w, h)
    _compare(gn, on); _compare(gm, om)


@pytest.mark.gpu
def test_gpu_prefilter_full_size_properties_and_errors(rt, A, gpu_ctx):
    """2048 x 2048 (the reference's Sponza texture size): size-independent properties instead of the O(texels x mips) oracle."""
    import ctypes as C
    content = importlib.import_module("incarnation-renderer_amd.content")
    W = H = 2048
    n = np.zeros((H, W, 4), np.uint8); n[...] = (128, 128, 255, 9)
    m = np.zeros((H, W, 4), np.uint8); m[..., 0] = 3; m[..., 1] = 150; m[..., 2] = np.arange(W, dtype=np.uint32)[None, :] % 256
    gn, gm = content.prefilter_normal_roughness(gpu_ctx, n, m)
    assert len(gn) == 10 and gn[-1].shape == (4, 4, 4)
    assert np.array_equal(gn[0], n) and np.array_equal(gm[0], m[..., [2, 1, 0, 3]])
    for k in range(1, 10):
        assert np.all(gn[k][..., :3] == gn[1][0, 0, :3])                                  # flat normals stay flat
        assert np.all(gm[k][..., 1] == gm[1][0, 0, 1])                                    # and so does the roughness
        box = m[..., 2].astype(np.float64).reshape(H >> k, 1 << k, W >> k, 1 << k).mean(axis=(1, 3))
        assert np.abs(gm[k][..., 0].astype(float) - box).max() <= 0.5 + 1e-3             # metalness: box filter, round to nearest
    lib = rt.load()                                                                      # INCR_ERR_INVALID_ARGUMENT = -1 (include/incr.h)
    d = A.PrefilterNormalRoughnessDesc(normalmap=1 << 20, metallic_roughness=1 << 20, width=64, height=64, num_mips=4, out_normalmap=1 << 20, out_metallic_roughness=1 << 20)
    assert lib.incr_prefilter_normal_roughness(gpu_ctx.handle, C.byref(d), None) == -1      # 64x64 has 5 levels
    d.width = d.height = 2; d.num_mips = 0
    assert lib.incr_prefilter_normal_roughness(gpu_ctx.handle, C.byref(d), None) == -1
