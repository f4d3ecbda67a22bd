"""Raster half on the CPU: the host meshlet builder, the per-frame view constants, and the raster oracle checked against
the ray-casting oracle (two independent restatements must agree on what is visible).  No GPU needed."""
import ctypes as C

import numpy as np
import pytest

DEFAULT_SH9 = np.array([[1.0, 1.1, 1.3], [0.3, 0.3, 0.35], [0.1, 0.1, 0.2], [-0.05, -0.04, -0.02], [0.02, 0.02, 0.03],
                        [0.04, 0.05, 0.05], [-0.03, -0.03, -0.02], [0.01, 0.02, 0.02], [0.05, 0.04, 0.03]], dtype=np.float32)


def meshlet_triangles(scene):
    """Decode every meshlet triangle to its three ORIGINAL vertex indices, in the order the shaders read them
    (visibility-pass.hlsl:78: packed fields high to low)."""
    ml, vi, tri = scene.meshlets
    out = []
    for k in range(scene.num_meshlets):
        m = ml[k]
        p = tri[m.triangleBegin:m.triangleBegin + m.triangleCount]
        loc = np.stack([(p >> 20) & 0xff, (p >> 10) & 0xff, p & 0xff], axis=1)
        assert (loc < m.vertexCount).all()
        out.append(vi[m.vertexBegin + loc])
    return out


@pytest.mark.parametrize("which", ["helmet", "sponza_small"])
def test_meshletize_invariants(request, which):
    scene = request.getfixturevalue(which)
    ml, vi, tri = scene.meshlets
    per_meshlet = meshlet_triangles(scene)
    # every primitive triangle appears exactly once, with its original winding (rotation aside)
    k = 0
    for mesh_index, prims in enumerate(scene.mesh_primitives):
        for p in prims:
            stride = scene.accessors[p["index"]][2]
            idx = scene.accessor_array(p["index"], np.uint16 if stride == 2 else np.uint32, 1, p["index_count"]).astype(np.uint32).reshape(-1, 3)
            got = []
            while k < scene.num_meshlets and ml[k].meshIndex == mesh_index and ml[k].positionAccessor == p["position"] and ml[k].materialIndex == p["material"] \
                    and len(got) < len(idx):
                m = ml[k]
                assert 1 <= m.vertexCount <= 64 and 1 <= m.triangleCount <= 126            # scene.cpp:1545-1546
                assert len(np.unique(vi[m.vertexBegin:m.vertexBegin + m.vertexCount])) == m.vertexCount
                got.append(per_meshlet[k]); k += 1
            got = np.concatenate(got)
            assert got.shape == idx.shape

            def canon(t):      # rotate each triple so that its smallest index comes first (keeps the winding)
                r = np.argmin(t, axis=1)
                return np.stack([t[np.arange(len(t)), (r + j) % 3] for j in range(3)], axis=1)
            a, b = canon(got), canon(idx)
            a = a[np.lexsort(a.T[::-1])]; b = b[np.lexsort(b.T[::-1])]
            np.testing.assert_array_equal(a, b)
    assert k == scene.num_meshlets
    # bounding spheres: MinimumBoundingSphere (mesh-utils.cpp:114-180) compares against the INITIAL radius^2 in its growing pass
    # (radiusSq is computed once, :158), so a late point inside the grown sphere can pull the sphere back: about 6 % of the
    # meshlet vertices end up outside it, by up to 1.36 r on these scenes.  Restated as is (the culling of the reference, which
    # is what the visibility ids depend on); exact equality with the oracle is tests/test_meshletize.py.
    for k in range(0, scene.num_meshlets, max(1, scene.num_meshlets // 50)):
        m = ml[k]
        nverts = int(vi[m.vertexBegin:m.vertexBegin + m.vertexCount].max()) + 1
        pos = scene.accessor_array(m.positionAccessor, np.float32, 3, nverts)[vi[m.vertexBegin:m.vertexBegin + m.vertexCount]]
        c, r = np.array(m.boundingSphere[:3]), m.boundingSphere[3]
        assert (np.linalg.norm(pos - c, axis=1) <= r * 1.5 + 1e-6).all()
    # meshlets are reasonably full (greedy growth along adjacency, not one triangle each)
    assert tri.size / scene.num_meshlets > 20


def test_meshletize_rejects_bad_input(rt):
    lib = rt.load()
    h = C.c_void_p()
    idx = np.array([0, 1, 5], dtype=np.uint32); pos = np.zeros((3, 3), dtype=np.float32)
    assert lib.incr_meshletize(idx.ctypes.data, 3, pos.ctypes.data, 3, 12, 64, 126, C.byref(h)) == -1      # index out of range
    assert lib.incr_meshletize(idx.ctypes.data, 2, pos.ctypes.data, 3, 12, 64, 126, C.byref(h)) == -1      # not a multiple of 3
    assert lib.incr_meshletize(None, 0, None, 0, 12, 64, 126, C.byref(h)) == 0                              # empty input -> empty set
    assert lib.incr_meshlet_set_counts(h, None, None) == 0
    lib.incr_meshlet_set_free(h)
    # degenerate + duplicate triangles survive
    idx = np.array([0, 1, 2, 0, 1, 2, 1, 1, 2], dtype=np.uint32)
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], dtype=np.float32)
    assert lib.incr_meshletize(idx.ctypes.data, 9, pos.ctypes.data, 3, 12, 64, 126, C.byref(h)) == 0
    nv, nt = C.c_uint32(), C.c_uint32()
    assert lib.incr_meshlet_set_counts(h, C.byref(nv), C.byref(nt)) >= 1 and nt.value == 3
    lib.incr_meshlet_set_free(h)


def test_view_constants_host_matches_oracle(helmet, sio, R, ob, A):
    view = sio.FView().reset(helmet, aspect=16 / 9)
    W, H = 640, 360
    for frame in (0, 1, 7, 15, 16, 33):
        j, jo = R.pixel_jitter(frame, W, H), ob.pixel_jitter(frame, W, H)
        np.testing.assert_array_equal(j.view(np.uint32), jo.view(np.uint32))
        assert abs(j[0]) <= 1.0 / W and abs(j[1]) <= 1.0 / H
        prev = R.matrix_to_numpy(R.make_view_constants(view, W, H).unjittered_view_proj)
        vc, vo = R.make_view_constants(view, W, H, j, prev), ob.view_constants(view, W, H, j, prev)
        for name, _ in A.ViewConstants._fields_[:12]:
            a, b = R.matrix_to_numpy(getattr(vc, name)), R.matrix_to_numpy(getattr(vo, name))
            # products are bit-identical; inverses go through double and one rounding in both implementations
            if "inv" in name:
                np.testing.assert_allclose(a, b, rtol=2e-7, atol=1e-12)
            else:
                np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))
    assert R.pixel_jitter(0, W, H)[0] == np.float32(-1.0 / W) and (R.pixel_jitter(16, W, H) == R.pixel_jitter(0, W, H)).all()
    # V * (P * T) really is the jittered projection: NDC shifts by the jitter
    vc = R.make_view_constants(view, W, H, (0.25, -0.5))
    p0, p1 = R.matrix_to_numpy(R.make_view_constants(view, W, H).view_proj_transform), R.matrix_to_numpy(vc.view_proj_transform)
    pt = np.array([0.3, 0.2, 4.0, 1.0], dtype=np.float32) @ np.linalg.inv(R.matrix_to_numpy(vc.view_transform).astype(np.float64))
    a, b = pt @ p0.astype(np.float64), pt @ p1.astype(np.float64)
    np.testing.assert_allclose(b[:2] / b[3] - a[:2] / a[3], [0.25, -0.5], atol=1e-5)


def run_oracle_frame(ob, R, A, scene, oscene, view, W, H, frame_index=0, prev=None, cfg=None, hs=None, sh9=DEFAULT_SH9, jitter=None):
    cfg = cfg or A.FConfig
    if jitter is None:
        jitter = R.pixel_jitter(frame_index, W, H) if cfg.EnableTAA and cfg.Viewmode == 0 else (0.0, 0.0)
    vc = R.make_view_constants(view, W, H, jitter, prev, cfg.CameraNearPlane)
    if hs is None:
        hs = ob.HostSurfaces(W, H, scene.num_meshlets, cfg.LightClusterDimX * cfg.LightClusterDimY * cfg.LightClusterDimZ, cfg.MaxLightsPerCluster)
        hs.arr["sh9"][:, :3] = sh9
    for name, d in R.build_descs(None, scene, vc, hs.ptr, frame_index, cfg):
        ob.run_pass(oscene, name, d)
    return vc, hs


def pinhole_rays(R, vc, W, H):
    ivp = R.matrix_to_numpy(vc.inv_view_proj_transform).astype(np.float64)
    ys, xs = np.mgrid[0:H, 0:W]
    px = (xs.ravel() + 0.5) / W * 2 - 1; py = -((ys.ravel() + 0.5) / H * 2 - 1)
    wp = np.stack([px, py, np.full_like(px, 0.5), np.ones_like(px)], 1) @ ivp
    wp = wp[:, :3] / wp[:, 3:4]
    eye = np.array(list(vc.eye_pos), dtype=np.float64)
    d = wp - eye
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.tile(eye, (W * H, 1)).astype(np.float32), d.astype(np.float32)


def test_visibility_oracle_agrees_with_ray_casting(helmet, sio, R, ob, A):
    """The rasteriser restatement and the ray caster restatement are independent: same front-face rule, same triangle."""
    W = H = 160
    osc = ob.OracleScene(helmet)
    view = sio.FView().reset(helmet, aspect=1.0)
    vc, hs = run_oracle_frame(ob, R, A, helmet, osc, view, W, H, jitter=(0.0, 0.0))
    vis, depth = hs.arr["vis"], hs.arr["depth"]
    covered = vis != 0xFFFFF000
    assert hs.arr["counts"][0] == helmet.num_meshlets and hs.arr["counts"][1] == 0          # single-sided material
    o, d = pinhole_rays(R, vc, W, H)
    ids, tuv = osc.trace_rays(o, d, 0.001, 10000.0)
    hit = (ids[:, 0] != 0).reshape(H, W)
    assert (covered == hit).mean() > 0.999
    ml, vi, tri = helmet.meshlets
    prim = helmet.mesh_primitives[0][0]
    idx = helmet.accessor_array(prim["index"], np.uint16, 1, prim["index_count"]).astype(np.uint32).reshape(-1, 3)
    both = covered & hit
    m_id, t_id = (vis >> 7)[both], (vis & 127)[both]
    begin_v = np.array([ml[int(k)].vertexBegin for k in m_id]); begin_t = np.array([ml[int(k)].triangleBegin for k in m_id])
    p = tri[begin_t + t_id]
    got = vi[begin_v[:, None] + np.stack([(p >> 20) & 0xff, (p >> 10) & 0xff, p & 0xff], axis=1)]
    want = idx[ids[:, 3].reshape(H, W)[both]]
    assert (got == want).all(axis=1).mean() > 0.99
    # reverse-Z infinite projection: depth = near / view-space z
    look = np.array(list(view.to_ctypes().look), dtype=np.float64)
    vz = ((d.astype(np.float64) @ look) * tuv[:, 0]).reshape(H, W)
    assert np.abs(depth[both] - 1.0 / vz[both]).max() < 2e-3
    assert (depth[~covered] == 0).all() and depth.max() <= 1.0
    # G-buffer of covered pixels: unit normals after the octahedral round trip, emissive in the colour target
    g1 = hs.arr["gbuffer_normals"][covered]
    oct_ = np.stack([(g1 & 0xffff).astype(np.uint16).view(np.float16), (g1 >> 16).astype(np.uint16).view(np.float16)], axis=1).astype(np.float32)
    assert ((oct_ >= 0) & (oct_ <= 1)).all()
    assert (hs.arr["gbuffer_basecolor"][~covered] == 0).all() and (hs.arr["gbuffer_basecolor"][covered] >> 24 == 0).all()
    osc.close()


def test_oracle_frame_sponza_small(sponza_small, sio, R, ob, A):
    """Lights, alpha-masked double-sided geometry and near-plane clipping (the camera stands inside the atrium)."""
    W, H = 192, 108
    osc = ob.OracleScene(sponza_small)
    view = sio.FView().reset(sponza_small, aspect=W / H)
    vc, hs = run_oracle_frame(ob, R, A, sponza_small, osc, view, W, H)
    n_def, n_ds = hs.arr["counts"]
    culled = hs.arr["render_stats"][0]
    assert n_def + n_ds + culled == sponza_small.num_meshlets and n_ds > 0 and culled > 0
    for name, n in (("default_list", n_def), ("double_sided_list", n_ds)):
        ids = hs.arr[name][:n, 0]
        assert (np.diff(ids.astype(np.int64)) > 0).all()                       # ascending meshlet id
    vis = hs.arr["vis"]
    assert (vis != 0xFFFFF000).mean() > 0.9
    # culling is conservative: rendering with frustum culling off gives the same visibility buffer
    class NoCull(A.FConfig):
        FrustumCulling = False
    _, hs2 = run_oracle_frame(ob, R, A, sponza_small, osc, view, W, H, cfg=NoCull)
    assert hs2.arr["render_stats"][0] == 0
    np.testing.assert_array_equal(hs2.arr["vis"], vis)
    np.testing.assert_array_equal(hs2.arr["depth"].view(np.uint32), hs.arr["depth"].view(np.uint32))
    # light culling: 64 punctual lights, lists hold ascending global indices, grid offsets are c * MaxLightsPerCluster
    grid = hs.arr["light_grid"]
    assert (grid[:, 0] == np.arange(len(grid)) * 64).all() and grid[:, 1].max() <= 64 and grid[:, 1].sum() == hs.arr["light_count"][0] > 0
    for c in np.nonzero(grid[:, 1])[0][:50]:
        l = hs.arr["light_lists"][c, :grid[c, 1]]
        assert (np.diff(l.astype(np.int64)) > 0).all() and (l >= 1).all()     # light 0 is the sun: never in a cluster list
    # lighting happened: colour target is finite and mostly non-zero on covered pixels
    col = ob.unpack_r11g11b10(hs.arr["color"])
    assert np.isfinite(col[vis != 0xFFFFF000]).all() and (col.sum(axis=-1) > 0).mean() > 0.5
    assert hs.arr["ray_stats"][3] > 0                                            # the sun traced shadow rays
    # TAA frame 0 copies the colour target
    np.testing.assert_array_equal(hs.arr["taa0"], hs.arr["color"])
    # second frame: history is blended (not a copy) and stays finite
    prev = R.matrix_to_numpy(vc.unjittered_view_proj)
    vc1, hs = run_oracle_frame(ob, R, A, sponza_small, osc, view, W, H, frame_index=1, prev=prev, hs=hs)
    t1 = ob.unpack_r11g11b10(hs.arr["taa1"])
    assert np.isfinite(t1[vis != 0xFFFFF000]).all() and not np.array_equal(hs.arr["taa1"], hs.arr["color"])
    osc.close()


def primitive_triangle_keys(scene, vis):
    """Per covered pixel of a primitive-path visibility buffer: (primitive id, sorted vertex index triple)."""
    prims = [p for plist in scene.mesh_primitives for p in plist]
    covered = vis != 0xFFFFF000
    pid, tri = (vis >> 20)[covered], (vis & 0xFFFFF)[covered]
    out = np.zeros((covered.sum(), 4), dtype=np.int64)
    out[:, 0] = pid
    for k in np.unique(pid):
        p = prims[int(k)]
        stride = scene.accessors[p["index"]][2]
        idx = scene.accessor_array(p["index"], np.uint16 if stride == 2 else np.uint32, 1, p["index_count"]).astype(np.int64).reshape(-1, 3)
        sel = pid == k
        out[sel, 1:] = np.sort(idx[tri[sel]], axis=1)
    return out


def meshlet_triangle_keys(scene, vis):
    """Same for a meshlet-path buffer; the meshlets of one primitive are contiguous and in primitive order."""
    ml, vi, tri = scene.meshlets
    prims = [(mi, p) for mi, plist in enumerate(scene.mesh_primitives) for p in plist]
    prim_of = np.zeros(scene.num_meshlets, dtype=np.int64)
    k = 0
    for pid, (mesh_index, p) in enumerate(prims):
        while k < scene.num_meshlets and ml[k].meshIndex == mesh_index and ml[k].positionAccessor == p["position"] and ml[k].materialIndex == p["material"] \
                and ml[k].normalAccessor == p["normal"]:
            prim_of[k] = pid; k += 1
    covered = vis != 0xFFFFF000
    m_id, t_id = (vis >> 7)[covered].astype(np.int64), (vis & 127)[covered].astype(np.int64)
    vb = np.array([m.vertexBegin for m in ml[:scene.num_meshlets]], dtype=np.int64); tb = np.array([m.triangleBegin for m in ml[:scene.num_meshlets]], dtype=np.int64)
    p = tri[tb[m_id] + t_id].astype(np.int64)
    verts = vi[vb[m_id][:, None] + np.stack([(p >> 20) & 0xff, (p >> 10) & 0xff, p & 0xff], axis=1)].astype(np.int64)
    out = np.zeros((covered.sum(), 4), dtype=np.int64)
    out[:, 0] = prim_of[m_id]; out[:, 1:] = np.sort(verts, axis=1)
    return out


def test_oracle_primitive_path_matches_meshlet_path(sponza_small, helmet, sio, R, ob, A):
    """FConfig::UseMeshlets == false draws the same triangles through cs_primitive_cull_main / vs_primitive_main /
    GetPrimitiveTriangleData: same depth and G-buffer, ids name the same triangle in the other encoding."""
    class Prims(A.FConfig):
        UseMeshlets = False
        EnableTAA = False
    class Meshlets(A.FConfig):
        EnableTAA = False
    for scene, (W, H) in ((helmet, (128, 128)), (sponza_small, (160, 90))):
        osc = ob.OracleScene(scene)
        view = sio.FView().reset(scene, aspect=W / H)
        _, hm = run_oracle_frame(ob, R, A, scene, osc, view, W, H, cfg=Meshlets)
        _, hp = run_oracle_frame(ob, R, A, scene, osc, view, W, H, cfg=Prims)
        nprims = sum(len(p) for p in scene.mesh_primitives)
        n_def, n_ds = hp.arr["counts"]
        assert n_def + n_ds + hp.arr["render_stats"][0] == nprims
        prims = [p for plist in scene.mesh_primitives for p in plist]
        for name, n in (("default_list", n_def), ("double_sided_list", n_ds)):
            lst = hp.arr[name][:n]
            assert (np.diff(lst[:, 0].astype(np.int64)) > 0).all()
            assert all(int(c) == prims[int(i)]["index_count"] for i, c in lst)            # vertexCount = m_indexCount
        cov_m, cov_p = hm.arr["vis"] != 0xFFFFF000, hp.arr["vis"] != 0xFFFFF000
        np.testing.assert_array_equal(cov_m, cov_p)
        # per-meshlet bounds cull more than per-primitive bounds, never differently on screen; depth ties between
        # coplanar triangles may resolve to another id, so compare depth everywhere and ids almost everywhere
        np.testing.assert_array_equal(hm.arr["depth"].view(np.uint32), hp.arr["depth"].view(np.uint32))
        same = (meshlet_triangle_keys(scene, hm.arr["vis"]) == primitive_triangle_keys(scene, hp.arr["vis"])).all(axis=1)
        assert same.mean() > 0.999
        for name in ("gbuffer_basecolor", "gbuffer_normals", "gbuffer_mra"):
            assert (hm.arr[name][cov_m][same] == hp.arr[name][cov_p][same]).all(), name
        col_m, col_p = ob.unpack_r11g11b10(hm.arr["color"]), ob.unpack_r11g11b10(hp.arr["color"])
        ok = np.zeros(cov_m.shape, bool); ok[cov_m] = same
        np.testing.assert_array_equal(col_m[ok], col_p[ok])
        osc.close()


def test_hbao_fixed_trig_is_accurate(ob):
    """The pinned polynomial cos / sin / acos of the HBAO pass stay within 2.5e-7 of the real functions on the ranges
    the pass uses (angles in [-pi, pi], acos argument in [-1, 1])."""
    out = (C.c_float * 3)()
    xs = np.concatenate([np.linspace(-np.pi, np.pi, 4001), [0.0, np.pi / 4, np.pi / 2, 3 * np.pi / 4]]).astype(np.float32)
    err_s = err_c = 0.0
    for x in xs:
        ob.lib().orc_hb_math(float(x), out)
        err_s = max(err_s, abs(out[0] - np.sin(np.float64(x)))); err_c = max(err_c, abs(out[1] - np.cos(np.float64(x))))
    assert err_s < 2.5e-7 and err_c < 2.5e-7, (err_s, err_c)
    err_a = 0.0
    for x in np.linspace(-1, 1, 4001).astype(np.float32):
        ob.lib().orc_hb_math(float(x), out)
        err_a = max(err_a, abs(out[2] - np.arccos(np.float64(x))))
    assert err_a < 6e-7, err_a
    ob.lib().orc_hb_math(float("nan"), out)
    assert all(np.isnan(v) for v in out)


def test_oracle_hbao_and_bent_normals(sponza_small, sio, R, ob, A):
    """hbao.hlsl restated: unoccluded surfaces facing the eye integrate to 1, creases darken, the bent normal stays in the
    upper hemisphere of the surface, and sky lighting is scaled by the AO factor."""
    W, H = 128, 72
    class Hbao(A.FConfig):
        EnableHBAO = True
        UseBentNormals = True
        EnableTAA = False
        EnableDirectLighting = False
    class Plain(A.FConfig):
        EnableTAA = False
        EnableDirectLighting = False
    osc = ob.OracleScene(sponza_small)
    view = sio.FView().reset(sponza_small, aspect=W / H)
    _, h0 = run_oracle_frame(ob, R, A, sponza_small, osc, view, W, H, cfg=Plain)
    _, h1 = run_oracle_frame(ob, R, A, sponza_small, osc, view, W, H, cfg=Hbao)
    covered = h1.arr["vis"] != 0xFFFFF000
    ao = h1.arr["ao"][covered].astype(np.float64) / 255
    assert ao.max() == 1.0 and 0.5 < ao.mean() < 0.999 and (ao < 0.9).mean() > 0.02       # open floor vs creases
    rays = int(h1.arr["ray_stats"][3])
    assert 8 * covered.sum() <= rays <= 32 * W * H
    # bent normal: unit vector, within 90 degrees of the shading normal on nearly every covered pixel
    def oct_decode(words):
        f = np.stack([(words & 0xffff).astype(np.uint16).view(np.float16), (words >> 16).astype(np.uint16).view(np.float16)], -1).astype(np.float64) * 2 - 1
        n = np.concatenate([f, (1 - np.abs(f[..., :1]) - np.abs(f[..., 1:2]))], -1)
        t = np.clip(-n[..., 2], 0, 1)
        n[..., 0] += np.where(n[..., 0] >= 0, -t, t); n[..., 1] += np.where(n[..., 1] >= 0, -t, t)
        return n / np.linalg.norm(n, axis=-1, keepdims=True)
    bent, nrm = oct_decode(h1.arr["bent_normals"][covered]), oct_decode(h1.arr["gbuffer_normals"][covered])
    ok = np.isfinite(bent).all(axis=1)
    # (the reference sums x-weighted slice tangents over a HALF circle of azimuths, so its bent normal leans towards the
    # view's up-right even on open surfaces: "mostly the same hemisphere" is all the formula promises)
    assert ok.mean() > 0.99 and ((bent[ok] * nrm[ok]).sum(axis=1) > 0).mean() > 0.75
    # sky lighting: colour(with HBAO) <= colour(without) where the bent normal equals the normal is not guaranteed, so compare
    # luminance statistically and exactly on pixels with ao == 0
    c0, c1 = ob.unpack_r11g11b10(h0.arr["color"]), ob.unpack_r11g11b10(h1.arr["color"])
    lit = covered & np.isfinite(c1).all(axis=-1)        # a pixel whose bent normal is NaN (the slice sum cancels) shades to NaN, as in the shader
    assert lit.sum() >= covered.sum() - 2
    assert c1[lit].sum() < c0[lit].sum()
    dark = covered & (h1.arr["ao"] == 0)
    if dark.any():
        g = ob.unpack_r11g11b10(h1.arr["color"])[dark]
        e = ob.unpack_r11g11b10(h0.arr["color"])[dark]
        assert (g <= e).all()
    osc.close()


def test_oracle_debug_views(helmet, sio, R, ob, A):
    """debug-visualization.hlsl restated: the colour target is overwritten by the selected G-buffer channel / id hue."""
    W = H = 96
    osc = ob.OracleScene(helmet)
    view = sio.FView().reset(helmet, aspect=1.0)
    out = {}
    for mode in (3, 4, 7, 9, 10):
        class Cfg(A.FConfig):
            Viewmode = mode
            EnableTAA = False
        _, hs = run_oracle_frame(ob, R, A, helmet, osc, view, W, H, cfg=Cfg)
        out[mode] = (ob.unpack_r11g11b10(hs.arr["color"]), hs)
    covered = out[3][1].arr["vis"] != 0xFFFFF000
    mra = out[3][1].arr["gbuffer_mra"]
    np.testing.assert_allclose(out[3][0][..., 0], (mra & 255) / 255.0, atol=1 / 64)            # metallic, 6-bit mantissa store
    assert (out[3][0][..., 0] == out[3][0][..., 1]).all()
    base = out[4][1].arr["gbuffer_basecolor"]
    np.testing.assert_allclose(out[4][0][..., 2], ((base >> 16) & 255) / 255.0, atol=1 / 32)
    assert (out[7][1].arr["color"] == 0).all()                                                 # Reflections: not a raster view -> 0
    tri = (out[9][1].arr["vis"] & 127)[covered]
    cols = out[9][1].arr["color"][covered]
    assert len({(int(t), int(c)) for t, c in zip(tri, cols)}) == len(set(tri.tolist())) > 20     # one hue per triangle id
    g1 = out[10][1].arr["gbuffer_normals"]
    n10 = out[10][0][covered]
    assert (n10 >= 0).all() and (n10 <= 1.0).all() and n10.std() > 0.05 and g1[covered].any()
    osc.close()


def test_oracle_forward_lighting(helmet, sponza_small, sio, R, ob, A):
    """forward-pass.hlsl restated on top of the primitive visibility pass: same nearest fragments as the deferred frame,
    the clear colour behind them, emissive-only output when every light term is off, material channels in the debug modes."""
    W, H = 128, 72
    for scene in (helmet, sponza_small):
        osc = ob.OracleScene(scene)
        view = sio.FView().reset(scene, aspect=W / H)

        class Deferred(A.FConfig):
            EnableTAA = False
            EnableDirectLighting = False
            EnableSkyLighting = False

        class Forward(Deferred):
            ForwardLighting = True
            EnableDiffuseIBL = False
            EnableSpecularIBL = False

        class ForwardBase(Forward):
            Viewmode = 4
        _, hd = run_oracle_frame(ob, R, A, scene, osc, view, W, H, cfg=Deferred)
        _, hf = run_oracle_frame(ob, R, A, scene, osc, view, W, H, cfg=Forward)
        # (in a whole frame the debug-view pass overwrites these modes afterwards, renderer.cpp:1961: stop at the forward pass)
        hb = ob.HostSurfaces(W, H, scene.num_meshlets, 16 * 9 * 24, 64)
        vcb = R.make_view_constants(view, W, H)
        for name, d in R.build_descs(None, scene, vcb, hb.ptr, 0, ForwardBase):
            ob.run_pass(osc, name, d)
            if name == "forward_lighting":
                break
        covered = hd.arr["vis"] != 0xFFFFF000
        # no frustum culling and per-primitive draws change neither coverage nor depth
        np.testing.assert_array_equal(hf.arr["depth"].view(np.uint32), hd.arr["depth"].view(np.uint32))
        # nothing but emissive * 20000 left: exactly what the G-buffer pass put into the colour target -- wherever both paths
        # interpolate the same uv, i.e. everywhere except the dynamic-sky background
        # (the G-buffer pass projects with the combined localToClip matrix, the forward vertex shader in two steps: the
        # barycentrics differ in the last bits and a few emissive texels with them)
        same = hf.arr["color"][covered] == hd.arr["color"][covered]
        assert same.mean() > 0.98
        if (~covered).any():
            bg = ob.unpack_r11g11b10(hf.arr["color"][~covered])
            # the dynamic sky pass runs after the forward pass and fills the far plane: no clear colour survives in the frame
            assert np.isfinite(bg).all()
        # base colour view: the G-buffer's UNORM8 base colour, re-quantised to R11G11B10
        g0 = hd.arr["gbuffer_basecolor"][covered]
        want = np.stack([(g0 >> s) & 255 for s in (0, 8, 16)], -1) / 255.0
        got = ob.unpack_r11g11b10(hb.arr["color"][covered])
        assert np.abs(got - want).max() < 1 / 32 + 1 / 255
        osc.close()


def test_oracle_decals(sponza_small, sponza_decals, sio, R, ob, A, tmp_path):
    """gbuffer-raster-pass restated: decals never touch ids or depth, only blend into the G-buffers inside their footprints,
    replace normals / metallic-roughness only when their material has the map, and the loader files "decal*" meshes apart."""
    import json
    W, H = 192, 108

    class NoTaa(A.FConfig):
        EnableTAA = False
    frames = []
    for scene in (sponza_small, sponza_decals):
        osc = ob.OracleScene(scene)
        view = sio.FView().reset(scene, aspect=W / H)
        frames.append(run_oracle_frame(ob, R, A, scene, osc, view, W, H, cfg=NoTaa)[1])
        osc.close()
    a, b = (f.arr for f in frames)
    np.testing.assert_array_equal(a["vis"], b["vis"])
    np.testing.assert_array_equal(a["depth"].view(np.uint32), b["depth"].view(np.uint32))
    touched = a["gbuffer_basecolor"] != b["gbuffer_basecolor"]
    assert 0.05 < touched.mean() < 0.6
    # the blend writes alpha = a*a + dst.a*(1-a) > 0 exactly where a decal fragment passed; the scene's own alpha is 0
    assert ((b["gbuffer_basecolor"] >> 24) > 0)[touched].mean() > 0.99 and ((a["gbuffer_basecolor"] >> 24) == 0).all()
    for name in ("gbuffer_normals", "gbuffer_mra"):
        changed = a[name] != b[name]
        assert changed.any() and not (changed & ((b["gbuffer_basecolor"] >> 24) == 0)).any(), name
    # decal 0 and 3 have no normal map, 0 and 2 no metallic-roughness map: some touched pixels keep each of the two surfaces
    assert (touched & (a["gbuffer_normals"] == b["gbuffer_normals"])).any() and (touched & (a["gbuffer_mra"] == b["gbuffer_mra"])).any()
    assert (a["color"] != b["color"]).any()                    # and the lighting sees the new G-buffer
    # loader: meshes whose name starts with "decal" go to the decal list (scene.cpp:281), everything else to the scene meshes
    tri = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    blob = tri.tobytes() + np.array([0, 1, 2, 0], np.uint16).tobytes()
    (tmp_path / "t.bin").write_bytes(blob)
    acc = [dict(bufferView=0, componentType=5126, count=3, type="VEC3", min=[0, 0, 0], max=[1, 1, 0]), dict(bufferView=1, componentType=5123, count=3, type="SCALAR")]
    prim = dict(attributes=dict(POSITION=0), indices=1)
    gltf = dict(asset=dict(version="2.0"), buffers=[dict(uri="t.bin", byteLength=len(blob))],
                bufferViews=[dict(buffer=0, byteOffset=0, byteLength=36), dict(buffer=0, byteOffset=36, byteLength=6)], accessors=acc,
                meshes=[dict(name="body", primitives=[prim]), dict(name="decal_sticker", primitives=[prim, prim])],
                nodes=[dict(mesh=0), dict(mesh=1, translation=[0, 0, 1])], scenes=[dict(nodes=[0, 1])], scene=0)
    (tmp_path / "t.gltf").write_text(json.dumps(gltf))
    s = sio.load_gltf(str(tmp_path / "t.gltf"))
    assert s.mesh_names == ["body"] and len(s.mesh_primitives) == 1 and len(s.decal_primitives) == 2
    d = s.to_desc()
    assert d.num_primitives == 1 and d.num_decal_primitives == 2 and d.decal_transforms[1].m[14] != 0.0


def test_oct_encoding_round_trip(ob):
    rng = np.random.default_rng(3)
    n = rng.normal(size=(500, 3)).astype(np.float32)
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    for v in n:
        enc, dec = (C.c_float * 2)(), (C.c_float * 3)()
        ob.lib().orc_oct_roundtrip((C.c_float * 3)(*v), enc, dec)
        assert 0 <= enc[0] <= 1 and 0 <= enc[1] <= 1
        np.testing.assert_allclose(np.array(dec[:]), v, atol=2e-6)
