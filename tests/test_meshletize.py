"""MeshUtils::Meshletize (mesh-utils.cpp:640-835): the product's host meshletizer (csrc/meshletize.cpp, through the C ABI)
against the step-for-step oracle restatement (oracle/oracle_meshlet.cpp) and the committed golden of the helmet -- meshlet
count, vertex lists, packed 10:10:10 triangles and bounding spheres bit for bit.  CPU only (host code on both sides)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN


def product_meshlets(rt, A, idx, pos, max_verts=64, max_prims=126):
    lib = rt.load()
    idx = np.ascontiguousarray(idx, dtype=np.uint32).ravel(); pos = np.ascontiguousarray(pos, dtype=np.float32).reshape(-1, 3)
    h = C.c_void_p()
    rt.check(lib.incr_meshletize(idx.ctypes.data, idx.size, pos.ctypes.data, pos.shape[0], 12, max_verts, max_prims, C.byref(h)))
    nvi, ntri = C.c_uint32(), C.c_uint32()
    nm = lib.incr_meshlet_set_counts(h, C.byref(nvi), C.byref(ntri))
    ml = (A.GpuMeshlet * max(nm, 1))(); vi = np.zeros(max(nvi.value, 1), np.uint32); tri = np.zeros(max(ntri.value, 1), np.uint32)
    rt.check(lib.incr_meshlet_set_copy(h, ml, vi.ctypes.data, tri.ctypes.data))
    lib.incr_meshlet_set_free(h)
    counts = np.array([[m.vertexBegin, m.vertexCount, m.triangleBegin, m.triangleCount] for m in ml[:nm]], dtype=np.uint32).reshape(-1, 4)
    spheres = np.array([list(m.boundingSphere) for m in ml[:nm]], dtype=np.float32).reshape(-1, 4)
    return dict(counts=counts, spheres=spheres, vertex_indices=vi[:nvi.value], packed_triangles=tri[:ntri.value])


def assert_same(a, b, what):
    np.testing.assert_array_equal(a["counts"], b["counts"], err_msg=what + ": meshlet ranges")
    np.testing.assert_array_equal(a["vertex_indices"], b["vertex_indices"], err_msg=what + ": vertex lists")
    np.testing.assert_array_equal(a["packed_triangles"], b["packed_triangles"], err_msg=what + ": packed triangles")
    np.testing.assert_array_equal(a["spheres"].view(np.uint32), b["spheres"].view(np.uint32), err_msg=what + ": bounding spheres")


def primitive_arrays(scene, p):
    n = p["index_count"]
    stride = scene.accessors[p["index"]][2]
    idx = scene.accessor_array(p["index"], np.uint16 if stride == 2 else np.uint32, 1, n).astype(np.uint32).ravel()
    nverts = max(int(p.get("vertex_count", 0)), int(idx.max()) + 1)
    return idx, np.ascontiguousarray(scene.accessor_array(p["position"], np.float32, 3, nverts))


def check_invariants(m, idx, max_verts, max_prims):
    ntri = idx.size // 3
    assert m["counts"][:, 3].sum() == ntri and (m["counts"][:, 3] <= max_prims).all() and (m["counts"][:, 3] > 0).all()
    # every input triangle appears exactly once, with reversed winding
    seen = np.zeros(ntri, dtype=np.int64)
    key = {}
    tri_sorted = np.sort(idx.reshape(-1, 3), axis=1)
    for t in range(ntri):
        key.setdefault(tuple(tri_sorted[t]), []).append(t)
    for vb, vc, tb, tc in m["counts"]:
        verts = m["vertex_indices"][vb:vb + vc]
        for p in m["packed_triangles"][tb:tb + tc]:
            loc = (p & 1023, (p >> 10) & 1023, (p >> 20) & 1023)
            assert max(loc) < vc
            g = tuple(sorted(int(verts[i]) for i in loc))
            assert g in key
    assert sum(len(v) for v in key.values()) == ntri


def test_helmet_matches_oracle_and_golden(rt, A, ob, helmet):
    p = helmet.mesh_primitives[0][0]
    idx, pos = primitive_arrays(helmet, p)
    assert pos.shape[0] == p["vertex_count"] == 14890 and idx.size == 46356
    prod = product_meshlets(rt, A, idx, pos)
    orc = ob.meshletize(idx, pos)
    assert_same(prod, orc, "helmet")
    g = np.load(os.path.join(GOLDEN, "helmet_meshlets.npz"))
    assert_same(prod, {k: g[k] for k in ("counts", "spheres", "vertex_indices", "packed_triangles")}, "helmet golden")
    assert (prod["counts"][:, 1] <= 64).all()
    check_invariants(prod, idx, 64, 126)
    # the scene-level packing uses the same call: FScene meshlets == the golden count
    assert helmet.num_meshlets == prod["counts"].shape[0]


def test_sponza_small_primitives_match_oracle(rt, A, ob, sponza_small):
    for k, p in enumerate(sponza_small.mesh_primitives[0]):
        idx, pos = primitive_arrays(sponza_small, p)
        assert_same(product_meshlets(rt, A, idx, pos), ob.meshletize(idx, pos), "sponza_small primitive %d" % k)


@pytest.mark.parametrize("kind,ntri", [("plain", 300), ("degenerate", 300), ("flat", 200), ("u32", 129), ("plain", 1)])
def test_soups_and_limits(rt, A, ob, synth, kind, ntri):
    s = synth.triangle_soup(0xC0FFEE + ntri, ntri, kind)
    for prims in s.mesh_primitives:
        for p in prims:
            idx, pos = primitive_arrays(s, p)
            for mv, mp in ((64, 126), (8, 4), (3, 1), (256, 128)):
                a, b = product_meshlets(rt, A, idx, pos, mv, mp), ob.meshletize(idx, pos, mv, mp)
                assert_same(a, b, "%s %d verts %d prims" % (kind, mv, mp))
                assert (a["counts"][:, 3] <= mp).all()


def test_grid_with_duplicate_positions_and_non_manifold_edges(rt, A, ob):
    """Welded seams (same position, different index), a fin sharing an edge with two faces (the best-normal pairing) and
    position-hash point reps: the adjacency corner cases of BuildAdjacencyList (:285-487)."""
    n = 12
    xs, ys = np.meshgrid(np.arange(n, dtype=np.float32), np.arange(n, dtype=np.float32))
    pos = np.stack([xs.ravel(), ys.ravel(), np.zeros(n * n, np.float32)], 1)
    tris = []
    for y in range(n - 1):
        for x in range(n - 1):
            a = y * n + x
            tris += [[a, a + 1, a + n], [a + 1, a + n + 1, a + n]]
    # duplicate a column of vertices (a UV seam) and re-point half the triangles at the copies
    dup = {y * n + 6: n * n + y for y in range(n)}
    pos = np.concatenate([pos, pos[[y * n + 6 for y in range(n)]]])
    tris = np.array(tris, dtype=np.uint32)
    right = (pos[tris[:, 0], 0] >= 6) & (pos[tris[:, 1], 0] >= 6) & (pos[tris[:, 2], 0] >= 6)
    for old, new in dup.items():
        sel = right[:, None] & (tris == old)
        tris[sel] = new
    # a fin standing on an interior edge: three faces share it
    base = pos.shape[0]
    pos = np.concatenate([pos, np.array([[3.5, 3.0, 2.0]], np.float32)])
    tris = np.concatenate([tris, np.array([[3 * n + 3, 3 * n + 4, base]], np.uint32)])
    idx = tris.ravel()
    for mv, mp in ((64, 126), (16, 10)):
        a, b = product_meshlets(rt, A, idx, pos, mv, mp), ob.meshletize(idx, pos, mv, mp)
        assert_same(a, b, "grid %d/%d" % (mv, mp))
        check_invariants(a, idx, mv, mp)
    assert a["counts"].shape[0] > 10


def test_rejects_bad_input(rt):
    lib = rt.load()
    h = C.c_void_p()
    idx = np.array([0, 1, 5], np.uint32); pos = np.zeros((3, 3), np.float32)
    assert lib.incr_meshletize(idx.ctypes.data, 3, pos.ctypes.data, 3, 12, 64, 126, C.byref(h)) != 0        # index out of range
    assert lib.incr_meshletize(idx.ctypes.data, 2, pos.ctypes.data, 3, 12, 64, 126, C.byref(h)) != 0        # not a multiple of 3
    idx[2] = 2
    assert lib.incr_meshletize(idx.ctypes.data, 3, pos.ctypes.data, 3, 12, 64, 126, C.byref(h)) == 0
    assert lib.incr_meshlet_set_counts(h, None, None) == 1
    lib.incr_meshlet_set_free(h)
