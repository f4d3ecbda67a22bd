"""GPU parity of the raster path: every pass of libincr_hip.so against the CPU oracle on the same seeded inputs.
Integer / id / byte surfaces must be bit-identical; float surfaces (R11G11B10 colour) may differ where device
transcendental functions (cosf of spot-light angles, logf of the cluster slice) round differently: the bound is one
code step of the storage format, stated in each test."""
import ctypes as C

import numpy as np
import pytest

from test_raster_host_and_oracle import DEFAULT_SH9, run_oracle_frame

pytestmark = pytest.mark.gpu

INT_SURFACES = ("vis", "gbuffer_basecolor", "gbuffer_normals", "gbuffer_mra")


def gpu_frame(rt, R, ctx, gscene, view, W, H, cfg, frames=1, sh9=DEFAULT_SH9, storage_rounding=0):
    rr = R.RasterRenderer(ctx, gscene, W, H, cfg)
    rr.set_sh9(sh9)
    out = []
    for _ in range(frames):
        vc = rr.render(view, storage_rounding=storage_rounding)
        ctx.synchronize()
        names = list(R.RasterSurfaces.SPECS) + ["default_list", "double_sided_list", "counts", "render_stats", "light_lists", "light_grid", "light_count", "ray_stats"]
        out.append((vc, {n: rr.surf.download(n) for n in names}))
    rr.close()
    return out


def colour_close(ob, a, b, what):
    """R11G11B10 surfaces: identical bits, or within one 5-bit-mantissa step on pixels that went through cosf / logf."""
    same = a == b
    if same.all():
        return 1.0
    fa, fb = ob.unpack_r11g11b10(a), ob.unpack_r11g11b10(b)
    fin = np.isfinite(fa) & np.isfinite(fb)
    assert (np.isfinite(fa) == np.isfinite(fb)).all(), what
    err = np.abs(fa - fb)[fin]
    ref = np.maximum(np.abs(fb)[fin], 1e-3)
    assert (err <= ref * 2.0 ** -4).all(), "%s: max rel err %g" % (what, (err / ref).max())
    return same.mean()


@pytest.mark.parametrize("which,W,H", [("helmet", 256, 256), ("sponza_small", 320, 180), ("sponza_small", 203, 117)])
def test_raster_frame_parity(request, rt, R, ob, A, sio, gpu_ctx, which, W, H):
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
            # culling: bit-exact lists, counts, statistics
            np.testing.assert_array_equal(gs["counts"], o["counts"])
            for name, n in (("default_list", o["counts"][0]), ("double_sided_list", o["counts"][1])):
                np.testing.assert_array_equal(gs[name][:n], o[name][:n])
            np.testing.assert_array_equal(gs["render_stats"], o["render_stats"])
            if len(scene.global_lights) > 1:
                np.testing.assert_array_equal(gs["light_grid"], o["light_grid"])
                np.testing.assert_array_equal(gs["light_count"], o["light_count"])
                for c in np.nonzero(o["light_grid"][:, 1])[0]:
                    np.testing.assert_array_equal(gs["light_lists"][c, :o["light_grid"][c, 1]], o["light_lists"][c, :o["light_grid"][c, 1]])
            # visibility + depth + G-buffer: bit-exact
            np.testing.assert_array_equal(gs["vis"], o["vis"])
            np.testing.assert_array_equal(gs["depth"].view(np.uint32), o["depth"].view(np.uint32))
            for name in INT_SURFACES[1:]:
                np.testing.assert_array_equal(gs[name], o[name], err_msg=name)
            # lighting + TAA
            frac = colour_close(ob, gs["color"], o["color"], "colour target")
            assert frac > 0.999, frac
            frac = colour_close(ob, gs["taa%d" % (frame & 1)], o["taa%d" % (frame & 1)], "taa")
            assert frac > 0.995, frac
            assert gs["ray_stats"].light_shadow_rays == int(o["ray_stats"][3])
            # tonemap of the GPU's own TAA output: bit-exact (the sRGB encode steps where the oracle's powf makes it step)
            ldr = ob.tonemap(gs["taa%d" % (frame & 1)], A.FConfig.Exposure, 0)
            np.testing.assert_array_equal(gs["ldr"].view(np.uint8).reshape(H, W, 4), ldr)
    finally:
        osc.close(); gscene.close()


def test_raster_options_and_edges(rt, R, ob, A, sio, gpu_ctx, sponza_small, synth):
    W, H = 160, 90
    view = sio.FView().reset(sponza_small, aspect=W / H)

    class Cfg(A.FConfig):
        FrustumCulling = False
        EnableTAA = False
        Viewmode = 1                  # LightingOnly: basecolor 0.5, no jitter
    gscene = gpu_ctx.create_scene(sponza_small)
    osc = ob.OracleScene(sponza_small)
    try:
        (vc, gs), = gpu_frame(rt, R, gpu_ctx, gscene, view, W, H, Cfg, storage_rounding=1)
        assert vc.jitter[0] == 0 and vc.jitter[1] == 0
        vco, hs = run_oracle_frame(ob, R, A, sponza_small, osc, view, W, H, cfg=Cfg)
        # the oracle run above used storage_rounding 0: redo the float stores with rounding 1
        for name, d in R.build_descs(None, sponza_small, vco, hs.ptr, 0, Cfg, storage_rounding=1):
            ob.run_pass(osc, name, d)
        assert gs["render_stats"][0] == 0 and gs["counts"].sum() == sponza_small.num_meshlets
        np.testing.assert_array_equal(gs["vis"], hs.arr["vis"])
        np.testing.assert_array_equal(gs["gbuffer_normals"], hs.arr["gbuffer_normals"])
        assert colour_close(ob, gs["color"], hs.arr["color"], "colour (RNE stores)") > 0.999
    finally:
        osc.close(); gscene.close()
    # hidden meshes draw nothing: every surface keeps its clear value
    hidden = synth.sponza_stat(texture_size=8, tri_scale=1 / 64.0, with_lights=False)
    hidden = R.generate_meshlets(hidden)
    hidden.mesh_visibility = [0] * len(hidden.mesh_primitives)
    gscene = gpu_ctx.create_scene(hidden)
    try:
        (vc, gs), = gpu_frame(rt, R, gpu_ctx, gscene, view, W, H, A.FConfig)
        assert gs["counts"].sum() == 0 and gs["render_stats"][0] == 0
        assert (gs["vis"] == 0xFFFFF000).all() and (gs["depth"] == 0).all() and (gs["gbuffer_basecolor"] == 0).all()
    finally:
        gscene.close()


@pytest.mark.parametrize("which,W,H", [("helmet", 200, 200), ("sponza_small", 320, 180)])
def test_raster_primitive_path_parity(request, rt, R, ob, A, sio, gpu_ctx, which, W, H):
    """FConfig::UseMeshlets == false (cs_primitive_cull_main, vs/ps_primitive_main, GetPrimitiveTriangleData): bit-exact
    against the oracle's primitive variants, through the pass-by-pass path and through the C++ host renderer; a row band of
    the primitive path equals the same rows of the full frame."""
    class Prims(A.FConfig):
        UseMeshlets = False
    scene = request.getfixturevalue(which)
    view = sio.FView().reset(scene, aspect=W / H)
    gscene = gpu_ctx.create_scene(scene)
    osc = ob.OracleScene(scene)
    try:
        g = gpu_frame(rt, R, gpu_ctx, gscene, view, W, H, Prims, frames=2)
        hr = R.HostRenderer(gpu_ctx, gscene, W, H, cfg=Prims, sh9_rgb=DEFAULT_SH9)
        band = R.HostRenderer(gpu_ctx, gscene, W, H, cfg=Prims, sh9_rgb=DEFAULT_SH9)
        y0, y1 = H // 3, 2 * H // 3
        band.set_band(y0, y1)
        prev, hs = None, None
        for frame, (vc, gs) in enumerate(g):
            vco, hs = run_oracle_frame(ob, R, A, scene, osc, view, W, H, frame_index=frame, prev=prev, hs=hs, cfg=Prims)
            prev = R.matrix_to_numpy(vco.unjittered_view_proj)
            o = hs.arr
            np.testing.assert_array_equal(gs["counts"], o["counts"])
            assert o["counts"].sum() + o["render_stats"][0] == sum(len(p) for p in scene.mesh_primitives)
            for name, n in (("default_list", o["counts"][0]), ("double_sided_list", o["counts"][1])):
                np.testing.assert_array_equal(gs[name][:n], o[name][:n])
            np.testing.assert_array_equal(gs["render_stats"], o["render_stats"])
            np.testing.assert_array_equal(gs["vis"], o["vis"])
            assert (gs["vis"] != 0xFFFFF000).mean() > 0.3
            np.testing.assert_array_equal(gs["depth"].view(np.uint32), o["depth"].view(np.uint32))
            for name in INT_SURFACES[1:]:
                np.testing.assert_array_equal(gs[name], o[name], err_msg=name)
            assert colour_close(ob, gs["color"], o["color"], "colour target") > 0.999
            assert colour_close(ob, gs["taa%d" % (frame & 1)], o["taa%d" % (frame & 1)], "taa") > 0.995
            hr.render(view); band.render(view); gpu_ctx.synchronize()
            for which_s, name in ((A.SURFACE_VISIBILITY, "vis"), (A.SURFACE_DEPTH, "depth"), (A.SURFACE_GBUFFER_NORMALS, "gbuffer_normals"),
                                  (A.SURFACE_HDR_RASTER, "color"), (A.SURFACE_TAA_ACCUMULATION, "taa%d" % (frame & 1))):
                full = hr.download(which_s)
                np.testing.assert_array_equal(full, gs[name].view(np.uint32), err_msg="C++ renderer frame %d %s" % (frame, name))
                if frame == 0:      # later frames need the gathered history (test_row_band_renderers_equal_full_frame)
                    np.testing.assert_array_equal(band.download(which_s)[y0:y1], full[y0:y1], err_msg="band %s" % name)
        hr.close(); band.close()
    finally:
        osc.close(); gscene.close()


@pytest.mark.parametrize("which,W,H", [("helmet", 160, 160), ("sponza_small", 256, 144)])
def test_hbao_parity(request, rt, R, ob, A, sio, gpu_ctx, which, W, H):
    """incr_hbao (hbao.hlsl): the pass evaluates its trigonometry with pinned polynomials and its ray queries are
    order-independent any-hit tests, so occlusion factors, bent normals and the ray count are bit-exact against the oracle;
    sky lighting with USE_BENT_NORMALS on top stays within the colour tolerance.  Also through the C++ host renderer."""
    class Hbao(A.FConfig):
        EnableHBAO = True
        UseBentNormals = True
    scene = request.getfixturevalue(which)
    view = sio.FView().reset(scene, aspect=W / H)
    gscene = gpu_ctx.create_scene(scene)
    osc = ob.OracleScene(scene)
    try:
        rr = R.RasterRenderer(gpu_ctx, gscene, W, H, Hbao); rr.set_sh9(DEFAULT_SH9)
        hr = R.HostRenderer(gpu_ctx, gscene, W, H, cfg=Hbao, sh9_rgb=DEFAULT_SH9)
        rr.render(view); hr.render(view); gpu_ctx.synchronize()
        _, hs = run_oracle_frame(ob, R, A, scene, osc, view, W, H, cfg=Hbao)
        o = hs.arr
        np.testing.assert_array_equal(rr.surf.download("vis"), o["vis"])
        ao, bent = rr.surf.download("ao"), rr.surf.download("bent_normals")
        np.testing.assert_array_equal(ao, o["ao"])
        np.testing.assert_array_equal(bent, o["bent_normals"])
        assert rr.surf.download("ray_stats").light_shadow_rays == int(o["ray_stats"][3])
        covered = o["vis"] != 0xFFFFF000
        assert 0 < (ao[covered] < 255).mean() < 1
        assert colour_close(ob, rr.surf.download("color"), o["color"], "colour with HBAO") > 0.999
        np.testing.assert_array_equal(hr.download(A.SURFACE_AMBIENT_OCCLUSION, dtype=np.uint8), ao)
        np.testing.assert_array_equal(hr.download(A.SURFACE_BENT_NORMALS), bent)
        np.testing.assert_array_equal(hr.download(A.SURFACE_HDR_RASTER), rr.surf.download("color"))
        # a row band of the pass equals the same rows of the full frame
        band = R.HostRenderer(gpu_ctx, gscene, W, H, cfg=Hbao, sh9_rgb=DEFAULT_SH9)
        y0, y1 = H // 4, H // 2
        band.set_band(y0, y1); band.render(view); gpu_ctx.synchronize()
        np.testing.assert_array_equal(band.download(A.SURFACE_AMBIENT_OCCLUSION, dtype=np.uint8)[y0:y1], ao[y0:y1])
        np.testing.assert_array_equal(band.download(A.SURFACE_HDR_RASTER)[y0:y1], rr.surf.download("color")[y0:y1])
        rr.close(); hr.close(); band.close()
        # bad arguments
        d = A.HbaoDesc(); d.scene = gscene.handle
        vc = R.make_view_constants(view, W, H); d.view = C.pointer(vc)
        assert gpu_ctx.lib.incr_hbao(gpu_ctx.handle, C.byref(d), None) == -1 and b"NULL" in gpu_ctx.lib.incr_last_error()
    finally:
        osc.close(); gscene.close()


def test_debug_view_modes_parity(rt, R, ob, A, sio, gpu_ctx, sponza_small):
    """RenderJob::DebugVizPass: every Viewmode the shader implements (and the ones it answers with 0), bit-exact against
    the oracle except LightClusterSlices, whose per-pixel logf may move a pixel across a slice boundary."""
    W, H = 192, 108
    view = sio.FView().reset(sponza_small, aspect=W / H)
    gscene = gpu_ctx.create_scene(sponza_small)
    osc = ob.OracleScene(sponza_small)
    try:
        seen = {}
        for mode in (2, 3, 4, 5, 8, 9, 10, 11, 12, 13):
            class Cfg(A.FConfig):
                Viewmode = mode
                EnableHBAO = mode in (12, 13)
                UseBentNormals = mode in (12, 13)
                EnableTAA = False
            (vc, gs), = gpu_frame(rt, R, gpu_ctx, gscene, view, W, H, Cfg)
            _, hs = run_oracle_frame(ob, R, A, sponza_small, osc, view, W, H, cfg=Cfg)
            if mode == 11:
                assert (gs["color"] == hs.arr["color"]).mean() > 0.999
            else:
                np.testing.assert_array_equal(gs["color"], hs.arr["color"], err_msg="viewmode %d" % mode)
            # tonemap passes debug views through (tonemap.hlsl:35-37): LDR = sRGB-encoded colour target, bit-exact
            ldr = ob.tonemap(gs["color"], Cfg.Exposure, mode)
            np.testing.assert_array_equal(gs["ldr"].view(np.uint8).reshape(H, W, 4), ldr)
            seen[mode] = gs["color"]
            # the C++ renderer takes the same branch
            hr = R.HostRenderer(gpu_ctx, gscene, W, H, cfg=Cfg, sh9_rgb=DEFAULT_SH9)
            hr.render(view); gpu_ctx.synchronize()
            np.testing.assert_array_equal(hr.download(A.SURFACE_HDR_RASTER), gs["color"], err_msg="C++ renderer viewmode %d" % mode)
            hr.close()
        assert (seen[5] == 0).all()                                                  # Emissive is not a raster debug view: 0.xxxx
        vis = hs.arr["vis"]
        for mode, key in ((8, vis >> 7), (9, vis & 127)):                            # one colour per id, several ids
            cols = {}
            for k, c in zip(key.ravel()[::7], seen[mode].ravel()[::7]):
                assert cols.setdefault(int(k), int(c)) == int(c)
            assert len(set(cols.values())) > 10
        assert len(np.unique(seen[11])) > 3 and len(np.unique(seen[12])) > 16
        d = A.DebugVizDesc(); vc2 = R.make_view_constants(view, W, H); d.view = C.pointer(vc2); d.viewmode = 4
        d.color_target = gpu_ctx.buffer(W * H * 4)
        assert gpu_ctx.lib.incr_debug_visualization(gpu_ctx.handle, C.byref(d), None) == -1 and b"source surface" in gpu_ctx.lib.incr_last_error()
        d.viewmode = 14
        assert gpu_ctx.lib.incr_debug_visualization(gpu_ctx.handle, C.byref(d), None) == -1
    finally:
        osc.close(); gscene.close()


def test_raster_crop_matches_committed_golden(rt, R, ob, A, sio, gpu_ctx, helmet):
    """tests/golden/helmet_raster_64.npz (tools/make_golden.py): the HIP frame reproduces the committed 64x64 ids, depth and
    G-buffers bit for bit, and the colour target within one storage step."""
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "helmet_raster_64.npz"))
    W = H = 64

    class NoTaa(A.FConfig):
        EnableTAA = False
    view = sio.FView().reset(helmet, aspect=1.0)
    gscene = gpu_ctx.create_scene(helmet)
    try:
        (vc, gs), = gpu_frame(rt, R, gpu_ctx, gscene, view, W, H, NoTaa, sh9=np.full((9, 3), 0.25, np.float32))
        np.testing.assert_array_equal(gs["counts"], g["counts"])
        np.testing.assert_array_equal(gs["vis"], g["vis"])
        np.testing.assert_array_equal(gs["depth"].view(np.uint32), g["depth"].view(np.uint32))
        for key, name in (("g0", "gbuffer_basecolor"), ("g1", "gbuffer_normals"), ("g2", "gbuffer_mra")):
            np.testing.assert_array_equal(gs[name], g[key], err_msg=key)
        assert colour_close(ob, gs["color"], g["color"], "colour vs golden") > 0.999
    finally:
        gscene.close()


@pytest.mark.parametrize("which,W,H", [("helmet", 160, 160), ("sponza_small", 256, 144)])
def test_forward_lighting_parity(request, rt, R, ob, A, sio, gpu_ctx, which, W, H):
    """FConfig::ForwardLighting (forward-pass.hlsl): depth bit-exact, colour within the storage-step tolerance of the deferred
    passes (same light functions), shadow-ray count equal; the shader's material view modes and the clear colour through the
    pass on its own; the whole frame through the C++ renderer."""
    class Forward(A.FConfig):
        ForwardLighting = True
    scene = request.getfixturevalue(which)
    view = sio.FView().reset(scene, aspect=W / H)
    gscene = gpu_ctx.create_scene(scene)
    osc = ob.OracleScene(scene)
    try:
        g = gpu_frame(rt, R, gpu_ctx, gscene, view, W, H, Forward, frames=2)
        hr = R.HostRenderer(gpu_ctx, gscene, W, H, cfg=Forward, sh9_rgb=DEFAULT_SH9)
        prev, hs = None, None
        for frame, (vc, gs) in enumerate(g):
            vco, hs = run_oracle_frame(ob, R, A, scene, osc, view, W, H, frame_index=frame, prev=prev, hs=hs, cfg=Forward)
            prev = R.matrix_to_numpy(vco.unjittered_view_proj)
            o = hs.arr
            np.testing.assert_array_equal(gs["depth"].view(np.uint32), o["depth"].view(np.uint32))
            assert colour_close(ob, gs["color"], o["color"], "forward colour") > 0.995
            assert colour_close(ob, gs["taa%d" % (frame & 1)], o["taa%d" % (frame & 1)], "forward taa") > 0.99
            assert gs["ray_stats"].light_shadow_rays == int(o["ray_stats"][3])
            hr.render(view); gpu_ctx.synchronize()
            np.testing.assert_array_equal(hr.download(A.SURFACE_HDR_RASTER), gs["color"].view(np.uint32), err_msg="C++ renderer frame %d" % frame)
            np.testing.assert_array_equal(hr.download(A.SURFACE_DEPTH), gs["depth"].view(np.uint32))
        hr.close()
        # the pass on its own: clear colour behind the geometry, material channels in the shader's view modes
        surf = R.RasterSurfaces(gpu_ctx, W, H, scene.num_meshlets)
        hs2 = ob.HostSurfaces(W, H, scene.num_meshlets, 16 * 9 * 24, 64)
        hs2.arr["sh9"][:, :3] = DEFAULT_SH9
        gpu_ctx.upload(surf.ptr["sh9"], hs2.arr["sh9"])
        vc = R.make_view_constants(view, W, H)
        for mode in (0, 2, 3, 4, 5):
            class M(Forward):
                Viewmode = mode
                EnableTAA = False
            for (name, dg), (_, do) in zip(R.build_descs(gscene.handle, scene, vc, surf.ptr, 0, M), R.build_descs(None, scene, vc, hs2.ptr, 0, M)):
                rt.check(getattr(gpu_ctx.lib, R.RasterRenderer.PASS_FN[name])(gpu_ctx.handle, C.byref(dg), None))
                ob.run_pass(osc, name, do)
                if name == "forward_lighting":
                    break
            gpu_ctx.synchronize()
            col = surf.download("color")
            if mode == 0:
                assert colour_close(ob, col, hs2.arr["color"], "forward pass") > 0.995
            else:
                np.testing.assert_array_equal(col, hs2.arr["color"], err_msg="forward viewmode %d" % mode)
            background = hs2.arr["depth"] == 0
            if background.any():
                assert (col[background] == ob.lib().orc_pack_r11g11b10((C.c_float * 3)(0.8, 0.8, 1.0), 0)).all()
        surf.close()
    finally:
        osc.close(); gscene.close()


def test_raster_tiny_and_empty_frames(rt, R, ob, A, sio, gpu_ctx, sponza_small):
    """Ragged and degenerate frames through every optional pass at once (primitive path off / on, HBAO, forward lighting):
    1x1, 3x2 and 65x1 targets, and a camera that looks away from the whole scene (every draw culled, all background)."""
    class Everything(A.FConfig):
        EnableHBAO = True
        UseBentNormals = True

    class EverythingPrims(Everything):
        UseMeshlets = False
        ForwardLighting = True
    gscene = gpu_ctx.create_scene(sponza_small)
    osc = ob.OracleScene(sponza_small)
    try:
        for cfg in (Everything, EverythingPrims):
            for W, H in ((1, 1), (3, 2), (65, 1), (2, 67)):
                view = sio.FView().reset(sponza_small, aspect=W / H)
                (vc, gs), = gpu_frame(rt, R, gpu_ctx, gscene, view, W, H, cfg)
                _, hs = run_oracle_frame(ob, R, A, sponza_small, osc, view, W, H, cfg=cfg)
                np.testing.assert_array_equal(gs["vis"], hs.arr["vis"])
                np.testing.assert_array_equal(gs["depth"].view(np.uint32), hs.arr["depth"].view(np.uint32))
                assert colour_close(ob, gs["color"], hs.arr["color"], "%dx%d colour" % (W, H)) >= 0.5
            # looking straight away from the atrium: nothing survives culling, nothing is rasterised
            W, H = 96, 54
            view = sio.FView().reset(sponza_small, aspect=W / H)
            away = sio.FView().reset(sponza_small, aspect=W / H)
            away.position = np.asarray(view.position, np.float32) - 500.0 * np.asarray(view.look, np.float32)
            away.look = -np.asarray(view.look, np.float32); away.right = -np.asarray(view.right, np.float32)
            away.update_view_transform()
            (vc, gs), = gpu_frame(rt, R, gpu_ctx, gscene, away, W, H, cfg)
            _, hs = run_oracle_frame(ob, R, A, sponza_small, osc, away, W, H, cfg=cfg)
            np.testing.assert_array_equal(gs["counts"], hs.arr["counts"])
            np.testing.assert_array_equal(gs["vis"], hs.arr["vis"])
            np.testing.assert_array_equal(gs["depth"].view(np.uint32), hs.arr["depth"].view(np.uint32))
            assert colour_close(ob, gs["color"], hs.arr["color"], "away colour") > 0.99
    finally:
        osc.close(); gscene.close()


def test_decal_gbuffer_pass_parity(rt, R, ob, A, sio, gpu_ctx, sponza_decals, monkeypatch):
    """incr_gbuffer_decals (gbuffer-raster-pass.inl): four overlapping decal primitives blended in draw order -- G-buffers
    bit-exact against the oracle at two resolutions, through the C++ renderer, in a row band, and with a forced tiny tile
    list the result must not change (list flushes keep the order)."""
    gscene = gpu_ctx.create_scene(sponza_decals)
    osc = ob.OracleScene(sponza_decals)
    try:
        for W, H in ((256, 144), (203, 117)):
            if W == 203:
                monkeypatch.setenv("INCR_DECAL_LIST_CAP", "256")          # 896 slots: the tile list is flushed between rounds
            view = sio.FView().reset(sponza_decals, aspect=W / H)
            g = gpu_frame(rt, R, gpu_ctx, gscene, view, W, H, A.FConfig, frames=2)
            hr = R.HostRenderer(gpu_ctx, gscene, W, H, sh9_rgb=DEFAULT_SH9)
            band = R.HostRenderer(gpu_ctx, gscene, W, H, sh9_rgb=DEFAULT_SH9)
            y0, y1 = H // 3, 2 * H // 3
            band.set_band(y0, y1)
            prev, hs = None, None
            for frame, (vc, gs) in enumerate(g):
                vco, hs = run_oracle_frame(ob, R, A, sponza_decals, osc, view, W, H, frame_index=frame, prev=prev, hs=hs)
                prev = R.matrix_to_numpy(vco.unjittered_view_proj)
                o = hs.arr
                np.testing.assert_array_equal(gs["vis"], o["vis"])
                for name in INT_SURFACES[1:]:
                    np.testing.assert_array_equal(gs[name], o[name], err_msg="%s frame %d" % (name, frame))
                assert ((gs["gbuffer_basecolor"] >> 24) > 0).mean() > 0.05            # decals really landed
                assert colour_close(ob, gs["color"], o["color"], "colour with decals") > 0.999
                hr.render(view); band.render(view); gpu_ctx.synchronize()
                for which_s, name in ((A.SURFACE_GBUFFER_BASECOLOR, "gbuffer_basecolor"), (A.SURFACE_GBUFFER_NORMALS, "gbuffer_normals"),
                                      (A.SURFACE_GBUFFER_MRA, "gbuffer_mra"), (A.SURFACE_HDR_RASTER, "color")):
                    np.testing.assert_array_equal(hr.download(which_s), gs[name].view(np.uint32), err_msg="C++ renderer %s" % name)
                    np.testing.assert_array_equal(band.download(which_s)[y0:y1], gs[name].view(np.uint32)[y0:y1], err_msg="band %s" % name)
            hr.close(); band.close()
        # a scene without decals: the pass is a no-op, not an error
        d = A.GBufferDecalsDesc()
        assert gpu_ctx.lib.incr_gbuffer_decals(gpu_ctx.handle, C.byref(d), None) == -1
    finally:
        osc.close(); gscene.close()


def test_raster_bad_arguments(rt, R, A, sio, gpu_ctx, helmet):
    gscene = gpu_ctx.create_scene(helmet)
    try:
        view = sio.FView().reset(helmet, aspect=1.0)
        vc = R.make_view_constants(view, 64, 64)
        lib = gpu_ctx.lib
        d = A.VisibilityDesc(); d.scene, d.view = gscene.handle, C.pointer(vc)
        assert lib.incr_visibility(gpu_ctx.handle, C.byref(d), None) == -1 and b"NULL" in lib.incr_last_error()
        d = A.LightCullDesc(); d.scene, d.view = gscene.handle, C.pointer(vc)
        buf = gpu_ctx.buffer(1 << 20)
        d.culled_light_lists = d.light_grid = d.culled_light_count = buf
        d.cluster_dim[:] = (16, 9, 100); d.max_lights_per_cluster = 64; d.cluster_depth_extent = 200.0
        assert lib.incr_light_cull(gpu_ctx.handle, C.byref(d), None) == -1 and b"cluster grid" in lib.incr_last_error()
        t = A.TaaResolveDesc(); t.view = C.pointer(vc)
        t.hdr_scene_color = t.history_in = t.history_out = t.depth_buffer = buf; t.history_index = 1
        assert lib.incr_taa_resolve(gpu_ctx.handle, C.byref(t), None) == -1 and b"alias" in lib.incr_last_error()
        dl = A.DirectLightingDesc(); dl.inp = R.lighting_inputs(gscene.handle, vc, dict(color=buf, depth=buf, gbuffer_basecolor=buf, gbuffer_normals=buf, gbuffer_mra=buf))
        dl.directional_light_index = -1
        assert lib.incr_direct_lighting(gpu_ctx.handle, C.byref(dl), None) == -1
        gpu_ctx.free(buf)
    finally:
        gscene.close()


def test_cpp_host_renderer_matches_pass_by_pass(rt, R, A, sio, gpu_ctx, sponza_small):
    """csrc/host/renderer.cpp (Renderer::Render in C++: RenderJob::<Pass>::Execute in the reference's order and with its
    frame-to-frame state) must produce exactly what the pass-by-pass sequence issued from Python produces."""
    W, H = 224, 126
    view = sio.FView().reset(sponza_small, aspect=W / H)
    gscene = gpu_ctx.create_scene(sponza_small)
    try:
        # raster branch, three frames (jitter sequence, TAA ping-pong, prevViewProj carried across frames)
        hr = R.HostRenderer(gpu_ctx, gscene, W, H, sh9_rgb=DEFAULT_SH9)
        rr = R.RasterRenderer(gpu_ctx, gscene, W, H); rr.set_sh9(DEFAULT_SH9)
        for frame in range(3):
            hr.render(view); rr.render(view)
            gpu_ctx.synchronize()
            for which, name in ((A.SURFACE_VISIBILITY, "vis"), (A.SURFACE_DEPTH, "depth"), (A.SURFACE_GBUFFER_BASECOLOR, "gbuffer_basecolor"),
                                (A.SURFACE_GBUFFER_NORMALS, "gbuffer_normals"), (A.SURFACE_GBUFFER_MRA, "gbuffer_mra"), (A.SURFACE_HDR_RASTER, "color"),
                                (A.SURFACE_TAA_ACCUMULATION, "taa%d" % (frame & 1)), (A.SURFACE_BACK_BUFFER, "ldr")):
                a = hr.download(which)
                b = rr.surf.download(name)
                np.testing.assert_array_equal(a, b.view(np.uint32), err_msg="frame %d %s" % (frame, name))
        (culled_objects, culled_lights), rays = hr.stats()                         # Renderer::GetRenderStats read-back
        assert [culled_objects, culled_lights] == list(rr.surf.download("render_stats")) and culled_objects > 0
        assert rays.light_shadow_rays == rr.surf.download("ray_stats").light_shadow_rays > 0
        hr.close(); rr.close()
        # path-trace branch: samples 1..MaxSampleCount-1, then the accumulation stops (renderer.cpp:1681)
        class Cfg(A.FConfig):
            MaxSampleCount = 4           # sqrtSampleCount = 2; samples 1, 2, 3
        hr = R.HostRenderer(gpu_ctx, gscene, W, H, cfg=Cfg, path_trace=True)
        pt = rt.PathTracer(gpu_ctx, gscene, W, H)
        for n in (1, 2, 3):
            hr.render(view); pt.render(view, sample_index=n, cfg=Cfg)
            np.testing.assert_array_equal(hr.download(A.SURFACE_PATHTRACE_HISTORY), pt.read()["history"], err_msg="sample %d" % n)
        assert hr.stats()[1].total_rays() == pt.read()["stats"].total_rays() > 0
        before = hr.download(A.SURFACE_PATHTRACE_HISTORY)
        hr.render(view)                                      # sample index 4 == MaxSampleCount: no further sample
        np.testing.assert_array_equal(hr.download(A.SURFACE_PATHTRACE_HISTORY), before)
        gpu_ctx.lib.incr_renderer_reset_accumulation(hr.handle)
        hr.render(view)
        pt2 = rt.PathTracer(gpu_ctx, gscene, W, H); pt2.render(view, sample_index=1, cfg=Cfg)
        np.testing.assert_array_equal(hr.download(A.SURFACE_PATHTRACE_HISTORY), pt2.read()["history"])
        hr.close(); pt.close(); pt2.close()
    finally:
        gscene.close()


def test_row_band_renderers_equal_full_frame(rt, R, A, sio, gpu_ctx, sponza_small):
    """Sort-first multi-GPU raster on one device: three C++ renderers stand for three ranks (ragged last band), each
    renders its row band; copying the TAA bands between them plays the RCCL all_gather.  Every band must be bit-identical
    to the same rows of the single-GPU frame, over three frames of TAA history."""
    W, H, N = 200, 113, 3
    rows = (H + N - 1) // N
    view = sio.FView().reset(sponza_small, aspect=W / H)
    gscene = gpu_ctx.create_scene(sponza_small)
    try:
        full = R.HostRenderer(gpu_ctx, gscene, W, H, sh9_rgb=DEFAULT_SH9)
        ranks = [R.HostRenderer(gpu_ctx, gscene, W, H, sh9_rgb=DEFAULT_SH9) for _ in range(N)]
        for r, hr in enumerate(ranks):
            hr.set_band(r * rows, min((r + 1) * rows, H))
        for frame in range(3):
            full.render(view)
            for hr in ranks:
                hr.render(view)
            gpu_ctx.synchronize()
            # "all_gather": every rank receives every other rank's band of the accumulation surface it just wrote
            bands = [hr.download(A.SURFACE_TAA_ACCUMULATION)[r * rows:min((r + 1) * rows, H)] for r, hr in enumerate(ranks)]
            gathered = np.concatenate(bands, axis=0)
            np.testing.assert_array_equal(gathered, full.download(A.SURFACE_TAA_ACCUMULATION), err_msg="frame %d taa" % frame)
            for hr in ranks:
                gpu_ctx.upload(hr.surface_ptr(A.SURFACE_TAA_ACCUMULATION), gathered)
            for which in (A.SURFACE_VISIBILITY, A.SURFACE_DEPTH, A.SURFACE_GBUFFER_NORMALS, A.SURFACE_HDR_RASTER):
                ref = full.download(which)
                for r, hr in enumerate(ranks):
                    y0, y1 = r * rows, min((r + 1) * rows, H)
                    np.testing.assert_array_equal(hr.download(which)[y0:y1], ref[y0:y1], err_msg="frame %d surface %d rank %d" % (frame, which, r))
        ranks[0].tonemap(); gpu_ctx.synchronize()
        np.testing.assert_array_equal(ranks[0].download(A.SURFACE_BACK_BUFFER), full.download(A.SURFACE_BACK_BUFFER))
        full.close()
        for hr in ranks:
            hr.close()
    finally:
        gscene.close()


def test_visibility_band_queue_overflow_path(rt, R, A, sio, gpu_ctx, sponza_small, monkeypatch):
    """Large triangles are cut into band items; when the band queue is full they are scanned whole by one wave instead.
    Forcing a tiny queue must not change a single id or depth bit."""
    import os
    W, H = 640, 360
    view = sio.FView().reset(sponza_small, aspect=W / H)
    gscene = gpu_ctx.create_scene(sponza_small)
    try:
        def frame():
            hr = R.HostRenderer(gpu_ctx, gscene, W, H)
            hr.render(view); gpu_ctx.synchronize()
            st = (C.c_uint32 * 4)()
            rt.check(gpu_ctx.lib.incr_visibility_get_stats(gpu_ctx.handle, st))
            out = hr.download(A.SURFACE_VISIBILITY), hr.download(A.SURFACE_DEPTH), list(st)
            hr.close()
            return out
        monkeypatch.setenv("INCR_VIS_BIN_CAP", "0")                            # no screen bins: every large triangle takes the band queue
        vis0, depth0, st0 = frame()
        assert st0[0] > 0 and st0[1] > 0 and st0[2] > 0 and st0[3] == 0        # triangles, band items, small items, no overflow
        monkeypatch.setenv("INCR_VIS_BAND_QUEUE_CAP", "16")
        vis1, depth1, st1 = frame()
        monkeypatch.delenv("INCR_VIS_BAND_QUEUE_CAP")
        assert st1[0] == st0[0] and st1[2] == st0[2] and st1[3] > 0            # same triangles, most large ones took the overflow path
        np.testing.assert_array_equal(vis1, vis0)
        np.testing.assert_array_equal(depth1, depth0)
    finally:
        gscene.close()


def test_sun_cover_map_changes_nothing(rt, R, A, sio, gpu_ctx, sponza_small, monkeypatch):
    """The sun's shadow rays skip the BVH walk where the sun cover map (the scene's greatest height toward the sun per cell of an orthographic grid)
    proves nothing can be hit.  With the map and without it the lit frame and the ray counters must agree bit for bit -- also after a mesh has
    moved (the map is rebuilt when incr_update_bvh changed something) and for a renderer on a second stream (which waits for the build)."""
    import torch
    W, H = 640, 360
    view = sio.FView().reset(sponza_small, aspect=W / H)

    def frames(cover):
        if cover: monkeypatch.delenv("INCR_SUN_COVER", raising=False)
        else: monkeypatch.setenv("INCR_SUN_COVER", "0")
        gscene = gpu_ctx.create_scene(sponza_small)
        out = []
        try:
            hr = R.HostRenderer(gpu_ctx, gscene, W, H, sh9_rgb=DEFAULT_SH9)
            hr.render(view); gpu_ctx.synchronize()
            out.append((hr.download(A.SURFACE_HDR_RASTER), hr.stats()[1].light_shadow_rays))
            other = torch.cuda.Stream(device=0)
            hr2 = R.HostRenderer(gpu_ctx, gscene, W, H, sh9_rgb=DEFAULT_SH9, stream=C.c_void_p(other.cuda_stream))
            hr2.render(view); other.synchronize(); gpu_ctx.synchronize()
            out.append((hr2.download(A.SURFACE_HDR_RASTER), hr2.stats()[1].light_shadow_rays))
            hr2.close()
            m = np.array(sponza_small.mesh_transforms[0], dtype=np.float32).reshape(4, 4).copy()
            m[3, 1] += 0.75                                        # the whole mesh three quarters of a unit up: every height on the map changes
            gscene.update_bvh(mesh_transforms=[m]); gpu_ctx.synchronize()
            hr.render(view); gpu_ctx.synchronize()
            out.append((hr.download(A.SURFACE_HDR_RASTER), hr.stats()[1].light_shadow_rays))
            hr.close()
        finally:
            gscene.close()
        return out
    with_map, without = frames(True), frames(False)
    assert with_map[0][1] > 0 and (with_map[0][0] != with_map[2][0]).any()          # the sun shines and the move changed the picture
    for k, ((a, ra), (b, rb)) in enumerate(zip(with_map, without)):
        np.testing.assert_array_equal(a, b, err_msg="frame %d" % k)
        assert ra == rb, k


@pytest.mark.parametrize("res", [(640, 360), (1000, 523)])
def test_visibility_screen_bins(rt, R, A, sio, gpu_ctx, sponza_small, monkeypatch, res):
    """Opaque large triangles are filed per screen bin and scanned into LDS by one workgroup per bin (scan_bin, the first phase of vis_scan_kernel); a triangle that finds a
    bin full takes the band queue instead.  With the bins off, on, and two entries small, ids and depths must agree bit for bit -- also at a
    resolution that is no multiple of the bin or the tile size."""
    W, H = res
    view = sio.FView().reset(sponza_small, aspect=W / H)
    gscene = gpu_ctx.create_scene(sponza_small)
    try:
        def frame():
            hr = R.HostRenderer(gpu_ctx, gscene, W, H)
            hr.render(view); gpu_ctx.synchronize()
            st = (C.c_uint32 * 4)()
            rt.check(gpu_ctx.lib.incr_visibility_get_stats(gpu_ctx.handle, st))
            out = hr.download(A.SURFACE_VISIBILITY), hr.download(A.SURFACE_DEPTH), list(st)
            hr.close()
            return out
        vis_on, depth_on, st_on = frame()
        monkeypatch.setenv("INCR_VIS_BIN_CAP", "0")
        vis_off, depth_off, st_off = frame()
        monkeypatch.setenv("INCR_VIS_BIN_CAP", "2")
        vis_two, depth_two, st_two = frame()
        monkeypatch.delenv("INCR_VIS_BIN_CAP")
        assert st_on[0] == st_off[0] == st_two[0] and st_on[1] < st_two[1] <= st_off[1]      # same triangles; band items: alpha-tested only < + overflowed <= all
        assert (vis_on != 0xFFFFF000).any()
        for v, d in ((vis_off, depth_off), (vis_two, depth_two)):
            np.testing.assert_array_equal(v, vis_on)
            np.testing.assert_array_equal(d, depth_on)
    finally:
        gscene.close()


def test_rccl_collective_on_library_surface(rt, R, A, sio, gpu_ctx, helmet):
    """bench.py gathers the TAA bands with torch.distributed directly on the renderer's surface (CUDA array interface over
    the raw pointer).  One-rank RCCL group: the aliasing tensor must see the surface and the collective must write it."""
    import torch
    import torch.distributed as dist
    import bench
    W, H = 64, 48
    view = sio.FView().reset(helmet, aspect=W / H)
    gscene = gpu_ctx.create_scene(helmet)
    hr = R.HostRenderer(gpu_ctx, gscene, W, H)
    try:
        hr.render(view); gpu_ctx.synchronize()
        ptr = hr.surface_ptr(A.SURFACE_TAA_ACCUMULATION)
        t = torch.as_tensor(bench._DeviceWords(ptr, W * H), device=torch.device("cuda", 0))
        ref = hr.download(A.SURFACE_TAA_ACCUMULATION)
        np.testing.assert_array_equal(t.cpu().numpy().view(np.uint32).reshape(H, W), ref)
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29655", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        try:
            band = (t + 1).clone()
            dist.all_gather_into_tensor(t, band)
            torch.cuda.synchronize()
        finally:
            dist.destroy_process_group()
        np.testing.assert_array_equal(hr.download(A.SURFACE_TAA_ACCUMULATION), (ref.view(np.int32) + 1).view(np.uint32))
    finally:
        hr.close(); gscene.close()


def test_raster_4k_properties(rt, R, ob, A, sio, gpu_ctx, synth):
    """Full-size frame (BASELINE config 3 shape: Sponza statistics at 3840x2160): size-independent properties."""
    scene = synth.sponza_stat(texture_size=256)
    scene = R.generate_meshlets(scene)
    W, H = 3840, 2160
    view = sio.FView().reset(scene, aspect=W / H)
    gscene = gpu_ctx.create_scene(scene)
    try:
        class NoCull(A.FConfig):
            FrustumCulling = False
        (vc, a), = gpu_frame(rt, R, gpu_ctx, gscene, view, W, H, A.FConfig)
        (_, b), = gpu_frame(rt, R, gpu_ctx, gscene, view, W, H, NoCull)
        # conservative culling: identical visibility with and without it; determinism run to run
        np.testing.assert_array_equal(a["vis"], b["vis"])
        (_, c), = gpu_frame(rt, R, gpu_ctx, gscene, view, W, H, A.FConfig)
        for n in ("vis", "depth", "color", "taa0", "ldr"):
            np.testing.assert_array_equal(a[n], c[n], err_msg=n)
        vis = a["vis"]
        covered = vis != 0xFFFFF000
        assert covered.mean() > 0.95
        ml, vi, tri = scene.meshlets
        mid = vis[covered] >> 7
        assert mid.max() < scene.num_meshlets
        tcount = np.array([m.triangleCount for m in ml[:scene.num_meshlets]])
        assert ((vis[covered] & 127) < tcount[mid]).all()
        # every id in the buffer belongs to a meshlet that survived culling
        n_def, n_ds = a["counts"]
        alive = np.zeros(scene.num_meshlets, bool)
        alive[a["default_list"][:n_def, 0]] = True; alive[a["double_sided_list"][:n_ds, 0]] = True
        assert alive[np.unique(mid)].all()
        assert (a["depth"][covered] > 0).all() and (a["depth"][~covered] == 0).all() and a["depth"].max() <= 1
    finally:
        gscene.close()
