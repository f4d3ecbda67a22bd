"""Degenerate inputs through the whole HIP path against the oracle: zero-area and duplicate triangles, a coplanar soup
(depth ties on every pixel: the draw-order tie rule decides the ids), 32-bit indices, one-triangle scenes."""
import numpy as np
import pytest

from test_raster_host_and_oracle import DEFAULT_SH9, run_oracle_frame
from test_gpu_raster import colour_close, gpu_frame

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind,ntri", [("plain", 300), ("degenerate", 300), ("flat", 200), ("u32", 129), ("plain", 1)])
def test_triangle_soups(rt, R, ob, A, sio, synth, gpu_ctx, kind, ntri):
    W, H = 96, 96
    scene = R.generate_meshlets(synth.triangle_soup(0xC0FFEE + ntri, ntri, kind))
    view = sio.FView().reset(None, aspect=1.0)               # default camera at (0, 0, -15) looking down +z
    gscene = gpu_ctx.create_scene(scene)
    osc = ob.OracleScene(scene)
    try:
        for prims in (False, True):
            class Cfg(A.FConfig):
                UseMeshlets = not prims
                EnableTAA = False
            (vc, gs), = gpu_frame(rt, R, gpu_ctx, gscene, view, W, H, Cfg)
            _, hs = run_oracle_frame(ob, R, A, scene, osc, view, W, H, cfg=Cfg)
            o = hs.arr
            np.testing.assert_array_equal(gs["counts"], o["counts"])
            np.testing.assert_array_equal(gs["vis"], o["vis"], err_msg="%s prims=%s" % (kind, prims))
            np.testing.assert_array_equal(gs["depth"].view(np.uint32), o["depth"].view(np.uint32))
            for name in ("gbuffer_basecolor", "gbuffer_normals", "gbuffer_mra"):
                np.testing.assert_array_equal(gs[name], o[name], err_msg=name)
            assert colour_close(ob, gs["color"], o["color"], "soup colour") > 0.99
            if ntri > 1:
                assert (o["vis"] != 0xFFFFF000).any()
        # path tracer on the same soup: ray counts equal, pixels within the float tolerance
        pt = rt.PathTracer(gpu_ctx, gscene, W, H)
        pt.render(view, 1)
        g = pt.read()
        d = rt.make_pathtrace_desc(view, W, H, 1)
        orc = osc.pathtrace(d, num_threads=4)
        assert g["stats"].total_rays() == orc["stats"].total_rays()
        gt = g["target"].view(np.float16).astype(np.float32); ot = orc["target"].view(np.float16).astype(np.float32)
        same = (g["target"] == orc["target"]).all(axis=-1)
        rel = np.abs(gt - ot) / np.maximum(np.abs(ot), 1e-3)
        assert (same | (rel <= 2.0 ** -7).all(axis=-1)).mean() > 0.999
        pt.close()
    finally:
        osc.close(); gscene.close()
