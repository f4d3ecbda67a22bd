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
