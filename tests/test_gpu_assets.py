"""GPU parity on the reference's other complete assets (assets/suzanne, assets/transmission-test: content/models/* of the reference checkout): primary hits bit-exact,
path trace with every pixel accounted for, raster ids / depth / G-buffers bit-exact, the forward pass (the one shader that reads KHR_materials_transmission)."""
import numpy as np
import pytest

from parity_helpers import accounted_parity
from test_gpu_raster import INT_SURFACES, colour_close, gpu_frame
from test_raster_host_and_oracle import run_oracle_frame

pytestmark = pytest.mark.gpu

ASSETS = ["suzanne", "transmission_test"]


@pytest.mark.parametrize("which", ASSETS)
def test_asset_primary_hits_and_path_trace(request, rt, sio, ob, gpu_ctx, which):
    scene = request.getfixturevalue(which)
    gscene = gpu_ctx.create_scene(scene)
    osc = ob.OracleScene(scene)
    try:
        # camera rays of a 128 x 128 frame + 50 k incoherent rays through the scene's bounds: hit flag, instance / geometry / primitive ids and t, u, v bit for bit
        W = H = 128
        view = sio.FView().reset(scene, aspect=1.0)
        d = rt.make_pathtrace_desc(view, W, H, 1)
        ys, xs = np.mgrid[0:H, 0:W]
        o, dr = osc.camera_rays(d, xs.ravel(), ys.ravel())
        gi, gt = gscene.trace_rays(o, dr, 0.001, 10000.0)
        oi, ot = osc.trace_rays(o, dr, 0.001, 10000.0)
        np.testing.assert_array_equal(gi, oi)
        np.testing.assert_array_equal(gt.view(np.uint32), ot.view(np.uint32))
        assert 0.2 < gi[:, 0].mean() <= 1.0
        if which == "transmission_test":
            assert len(np.unique(gi[gi[:, 0] == 1, 1])) >= 10          # many of the 22 instances are seen
        rng = np.random.default_rng(77)
        n = 50_000
        c = np.asarray(view.position, np.float32)
        ro = (c + rng.uniform(-1.0, 1.0, (n, 3)) * 0.5 * np.linalg.norm(c)).astype(np.float32)
        rd = rng.standard_normal((n, 3)).astype(np.float32); rd /= np.linalg.norm(rd, axis=1, keepdims=True).astype(np.float32)
        gi, gt = gscene.trace_rays(ro, rd, 0.001, 10000.0); oi, ot = osc.trace_rays(ro, rd, 0.001, 10000.0)
        np.testing.assert_array_equal(gi, oi)
        np.testing.assert_array_equal(gt.view(np.uint32), ot.view(np.uint32))
        ga, _ = gscene.trace_rays(ro, rd, 0.1, 1000.0, any_hit=True); oa, _ = osc.trace_rays(ro, rd, 0.1, 1000.0, any_hit=True)
        np.testing.assert_array_equal(ga[:, 0], oa[:, 0])
        # 256 x 256, 1 spp: 2^-7 per channel, every pixel outside it explained by its hit chain (MASK materials go through ahsMain's alpha test)
        W = H = 256
        view = sio.FView().reset(scene, aspect=1.0)
        pt = rt.PathTracer(gpu_ctx, gscene, W, H)
        accounted_parity(rt, pt, osc, view, W, H, 1, 3e-4, which)
        pt.close()
    finally:
        osc.close(); gscene.close()


@pytest.mark.parametrize("which", ASSETS)
def test_asset_raster_frame(request, rt, R, ob, A, sio, gpu_ctx, which):
    W, H = 384, 216
    scene = request.getfixturevalue(which)
    view = sio.FView().reset(scene, aspect=W / H)
    gscene = gpu_ctx.create_scene(scene)
    osc = ob.OracleScene(scene)
    try:
        g = gpu_frame(rt, R, gpu_ctx, gscene, view, W, H, A.FConfig, frames=2)
        prev, hs = None, None
        for frame, (vc, gs) in enumerate(g):
            vco, hs = run_oracle_frame(ob, R, A, scene, osc, view, W, H, frame_index=frame, prev=prev, hs=hs)
            prev = R.matrix_to_numpy(vco.unjittered_view_proj)
            o = hs.arr
            np.testing.assert_array_equal(gs["counts"], o["counts"])
            for name, n in (("default_list", o["counts"][0]), ("double_sided_list", o["counts"][1])):
                np.testing.assert_array_equal(gs[name][:n], o[name][:n])
            np.testing.assert_array_equal(gs["vis"], o["vis"])
            np.testing.assert_array_equal(gs["depth"].view(np.uint32), o["depth"].view(np.uint32))
            for name in INT_SURFACES[1:]:
                np.testing.assert_array_equal(gs[name], o[name], err_msg=name)
            assert colour_close(ob, gs["color"], o["color"], "colour target") > 0.999
            assert colour_close(ob, gs["taa%d" % (frame & 1)], o["taa%d" % (frame & 1)], "taa") > 0.995
            assert (o["vis"] != 0).mean() > 0.05
        if which == "transmission_test":
            assert o["counts"][1] > 0          # the double-sided draw list is exercised: 12 of the 14 materials are double sided
    finally:
        osc.close(); gscene.close()


def test_transmission_reaches_the_forward_pass(rt, R, ob, A, sio, gpu_ctx, transmission_test):
    """forward-pass.hlsl:160: albedo = (1 - metallic)(1 - transmission) basecolor, transmission = factor x the map's red channel (material/common.hlsli:83-88)."""
    class Forward(A.FConfig):
        ForwardLighting = True
    W, H = 256, 144
    scene = transmission_test
    view = sio.FView().reset(scene, aspect=W / H)
    gscene = gpu_ctx.create_scene(scene)
    osc = ob.OracleScene(scene)
    try:
        (vc, gs), = gpu_frame(rt, R, gpu_ctx, gscene, view, W, H, Forward, frames=1)
        vco, hs = run_oracle_frame(ob, R, A, scene, osc, view, W, H, frame_index=0, prev=None, hs=None, cfg=Forward)
        np.testing.assert_array_equal(gs["depth"].view(np.uint32), hs.arr["depth"].view(np.uint32))
        assert colour_close(ob, gs["color"], hs.arr["color"], "forward colour") > 0.995
    finally:
        osc.close(); gscene.close()
