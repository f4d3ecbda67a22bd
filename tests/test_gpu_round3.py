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
