"""Tangent generation for primitives without a TANGENT accessor (MeshUtils::FixupMeshes, mesh-utils.cpp:560-638 -> MikkTSpace):
the product's implementation (csrc/mikktspace_gen.cpp, through the C ABI) against the oracle's step-by-step restatement of
mikktspace.c (oracle/oracle_mikkt.cpp), bit for bit, and both against the tangents the Damaged Helmet asset ships -- which were
written by MikkTSpace itself (Blender's exporter): the one place where real output of the third-party code pins the restatement."""
import ctypes as C

import numpy as np
import pytest


def product_tangents(rt, pos, nrm, uv, idx):
    L = rt.load()
    pos = np.ascontiguousarray(pos, np.float32); nrm = np.ascontiguousarray(nrm, np.float32); uv = np.ascontiguousarray(uv, np.float32); idx = np.ascontiguousarray(idx, np.uint32)
    out = np.full((pos.shape[0], 4), 7.0, np.float32)
    rc = L.incr_generate_tangents(pos.ctypes.data, nrm.ctypes.data, uv.ctypes.data, idx.ctypes.data, pos.shape[0], idx.shape[0], out.ctypes.data)
    assert rc == 0
    return out


def helmet_arrays(helmet):
    p = helmet.mesh_primitives[0][0]
    nv = 14890
    pos = helmet.accessor_array(p["position"], np.float32, 3, nv); nrm = helmet.accessor_array(p["normal"], np.float32, 3, nv)
    uv = helmet.accessor_array(p["uv"], np.float32, 2, nv); tan = helmet.accessor_array(p["tangent"], np.float32, 4, nv)
    idx = helmet.accessor_array(p["index"], np.uint16, 1, p["index_count"]).astype(np.uint32).reshape(-1, 3)
    return pos, nrm, uv, idx, tan


def test_helmet_with_tangents_stripped(rt, ob, helmet):
    pos, nrm, uv, idx, shipped = helmet_arrays(helmet)
    o = ob.generate_tangents(pos, nrm, uv, idx)
    g = product_tangents(rt, pos, nrm, uv, idx)
    np.testing.assert_array_equal(g.view(np.uint32), o.view(np.uint32))                 # product == oracle, bit for bit
    used = np.zeros(pos.shape[0], bool); used[idx.ravel()] = True
    # against the asset's own MikkTSpace tangents: the direction to float rounding; the sign is the opposite one because the
    # exporter wrote the tangents before it flipped V (glTF's texture origin), and the reference would see the flipped UVs too
    d = np.abs(o[used, :3] - shipped[used, :3]).max(axis=1)
    assert np.quantile(d, 0.999) < 3e-7 and (d < 1e-5).mean() > 0.9995
    assert (o[used, 3] == -shipped[used, 3]).mean() > 0.999
    assert (np.abs(np.linalg.norm(o[used, :3], axis=1) - 1) < 1e-5).mean() > 0.999
    assert (o[~used] == 0).all()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_meshes_with_degenerates_mirrored_uvs_and_shared_seams(rt, ob, seed):
    """A bumpy grid with: mirrored UV halves (both orientations), duplicated seam vertices (welded by value), zero-area UV triangles
    (group-with-anything), degenerate triangles (two corners at one position), a non-manifold fin, an unreferenced vertex."""
    rng = np.random.default_rng(seed)
    n = 24
    ys, xs = np.mgrid[0:n, 0:n].astype(np.float32)
    pos = np.stack([xs, rng.normal(0, 0.3, (n, n)).astype(np.float32), ys], -1).reshape(-1, 3)
    nrm = rng.normal(0, 0.2, (n * n, 3)).astype(np.float32) + np.array([0, 1, 0], np.float32); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    uv = np.stack([np.abs(xs - n / 2) / n, ys / n], -1).reshape(-1, 2).astype(np.float32)      # mirrored about the middle column
    tris = []
    for y in range(n - 1):
        for x in range(n - 1):
            a = y * n + x
            tris += [[a, a + 1, a + n], [a + 1, a + n + 1, a + n]]
    idx = np.array(tris, np.uint32)
    # duplicate a column of vertices (a seam): same position / normal / uv under new indices
    seam = np.arange(5, n * n, n)
    base = pos.shape[0]
    pos = np.concatenate([pos, pos[seam], [[100, 100, 100]]]); nrm = np.concatenate([nrm, nrm[seam], [[0, 1, 0]]]); uv = np.concatenate([uv, uv[seam], [[0, 0]]])
    remap = {int(v): base + k for k, v in enumerate(seam)}
    right = idx[:, :].copy()
    for t in range(idx.shape[0]):
        if (pos[idx[t], 0] >= 5).all():
            right[t] = [remap.get(int(v), int(v)) for v in idx[t]]
    idx = right
    # zero-area UVs on a patch, degenerate triangles, a fin sharing an edge with two faces
    uv[idx[40:46].ravel()] = uv[idx[40, 0]]
    idx = np.concatenate([idx, [[3, 3, 4], [10, 11, 10]], [[0, 1, n * n - 1]]]).astype(np.uint32)
    order = rng.permutation(idx.shape[0]); idx = idx[order]
    o = ob.generate_tangents(pos, nrm, uv, idx)
    g = product_tangents(rt, pos.astype(np.float32), nrm.astype(np.float32), uv.astype(np.float32), idx)
    np.testing.assert_array_equal(g.view(np.uint32), o.view(np.uint32))
    assert set(np.unique(o[:, 3])) <= {-1.0, 0.0, 1.0} and (o[:, 3] == 1).any() and (o[:, 3] == -1).any()
    assert (o[-1] == 0).all()                                                             # the unreferenced vertex


def test_argument_checks(rt):
    L = rt.load()
    z = np.zeros((3, 3), np.float32); uv = np.zeros((3, 2), np.float32); out = np.zeros((3, 4), np.float32)
    bad = np.array([[0, 1, 5]], np.uint32)
    assert L.incr_generate_tangents(z.ctypes.data, z.ctypes.data, uv.ctypes.data, bad.ctypes.data, 3, 1, out.ctypes.data) != 0
    assert L.incr_generate_tangents(None, z.ctypes.data, uv.ctypes.data, bad.ctypes.data, 3, 1, out.ctypes.data) != 0
    ok = np.array([[0, 1, 2]], np.uint32)
    assert L.incr_generate_tangents(z.ctypes.data, z.ctypes.data, uv.ctypes.data, ok.ctypes.data, 3, 0, out.ctypes.data) == 0


def test_loader_generates_tangents_where_the_asset_ships_none(rt, ob, sio, helmet):
    """scene_io.load_gltf = FScene::ReloadModel's FixupMeshes step: with the helmet's TANGENT accessor ignored, the primitive gets a
    generated one (new buffer / view / accessor, stride 16) holding exactly what the oracle's MikkTSpace restatement computes."""
    import os
    from conftest import ROOT
    s = sio.load_gltf(os.path.join(ROOT, "assets/damaged-helmet/DamagedHelmet.gltf"), load_images=False, ignore_tangents=True)
    p = s.mesh_primitives[0][0]
    assert p["tangent"] != helmet.mesh_primitives[0][0]["tangent"] and s.accessors[p["tangent"]][2] == 16
    pos, nrm, uv, idx, _ = helmet_arrays(helmet)
    got = s.accessor_array(p["tangent"], np.float32, 4, pos.shape[0])
    np.testing.assert_array_equal(got.view(np.uint32), ob.generate_tangents(pos, nrm, uv, idx).view(np.uint32))
    # the asset as shipped keeps its own accessor
    assert helmet.mesh_primitives[0][0]["tangent"] == sio.load_gltf(os.path.join(ROOT, "assets/damaged-helmet/DamagedHelmet.gltf"), load_images=False).mesh_primitives[0][0]["tangent"]
