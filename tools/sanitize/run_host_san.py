"""Drives tools/sanitize/_build/libincr_host_san.so (ASan + UBSan build of the host-only C++) over the test scenes.
Run through tests/test_host_sanitizers.py, which sets LD_PRELOAD to libasan; exits non-zero on any finding."""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
PKG = "incarnation-renderer_amd"
A = importlib.import_module(PKG + ".ctypes_api")
sio = importlib.import_module(PKG + ".scene_io")
synth = importlib.import_module(PKG + ".synth")

L = C.CDLL(os.path.join(ROOT, "tools", "sanitize", "_build", "libincr_host_san.so"))
vp, u32 = C.c_void_p, C.c_uint32
L.san_build_scene.restype = C.c_int
L.san_build_scene.argtypes = [C.POINTER(A.SceneDesc), C.POINTER(C.c_uint64), C.c_char_p, u32]
L.incr_meshletize.restype = C.c_int
L.incr_meshletize.argtypes = [vp, u32, vp, u32, u32, u32, u32, C.POINTER(vp)]
L.incr_meshlet_set_counts.restype = u32; L.incr_meshlet_set_counts.argtypes = [vp, C.POINTER(u32), C.POINTER(u32)]
L.incr_meshlet_set_copy.restype = C.c_int; L.incr_meshlet_set_copy.argtypes = [vp, vp, vp, vp]
L.incr_meshlet_set_free.restype = None; L.incr_meshlet_set_free.argtypes = [vp]
L.incr_make_view_constants.restype = C.c_int
L.incr_make_view_constants.argtypes = [C.POINTER(A.View), C.POINTER(A.View), C.POINTER(C.c_float), C.POINTER(A.Matrix), C.c_float, u32, u32, C.POINTER(A.ViewConstants)]


def build(scene, name):
    d = scene.to_desc()
    counts = (C.c_uint64 * 4)(); why = C.create_string_buffer(256)
    rc = L.san_build_scene(C.byref(d), counts, why, 256)
    assert rc == 0, (name, rc, why.value)
    assert counts[0] == scene.num_triangles and counts[1] > 0, (name, list(counts))
    print("%s: %d triangles, %d nodes, %d grid offsets, %d grid lights" % ((name,) + tuple(counts)))


def meshletize(scene, name):
    total = 0
    for prims in scene.mesh_primitives:
        for p in prims:
            n = p["index_count"]
            stride = scene.accessors[p["index"]][2]
            idx = np.ascontiguousarray(scene.accessor_array(p["index"], np.uint16 if stride == 2 else np.uint32, 1, n).astype(np.uint32).ravel())
            nverts = int(idx.max()) + 1 if n else 0
            pos = np.ascontiguousarray(scene.accessor_array(p["position"], np.float32, 3, nverts))
            h = vp()
            rc = L.incr_meshletize(idx.ctypes.data, n, pos.ctypes.data, nverts, 12, 64, 126, C.byref(h))
            assert rc == 0, (name, rc)
            nv, nt = u32(), u32()
            nm = L.incr_meshlet_set_counts(h, C.byref(nv), C.byref(nt))
            ml = (A.GpuMeshlet * max(nm, 1))(); vi = np.zeros(max(nv.value, 1), np.uint32); tri = np.zeros(max(nt.value, 1), np.uint32)
            assert L.incr_meshlet_set_copy(h, C.cast(ml, vp), vi.ctypes.data, tri.ctypes.data) == 0
            assert nt.value == n // 3
            L.incr_meshlet_set_free(h)
            total += nm
    print("%s: %d meshlets" % (name, total))


helmet = sio.load_gltf(os.path.join(ROOT, "assets/damaged-helmet/DamagedHelmet.gltf"))
sponza = synth.sponza_stat(texture_size=8, tri_scale=1.0 / 16.0)
edge = synth.sponza_stat(texture_size=8, tri_scale=1.0 / 64.0, with_lights=False)
edge.mesh_visibility = [0] * len(edge.mesh_primitives)          # nothing visible: empty BVH
for scene, name in ((helmet, "helmet"), (sponza, "sponza_small")):
    build(scene, name)
    meshletize(scene, name)
d = edge.to_desc(); counts = (C.c_uint64 * 4)(); why = C.create_string_buffer(256)
assert L.san_build_scene(C.byref(d), counts, why, 256) == 0 and counts[0] == 0, (list(counts), why.value)
view = sio.FView().reset(helmet, aspect=1.0).to_ctypes()
vc = A.ViewConstants()
assert L.incr_make_view_constants(C.byref(view), None, (C.c_float * 2)(0.001, -0.002), None, 1.0, 640, 480, C.byref(vc)) == 0
assert L.incr_make_view_constants(None, None, None, None, 1.0, 640, 480, C.byref(vc)) != 0
print("host sanitizer run clean")


# ---- fuzz: random triangle soups with the degenerate cases a loader can hand over ----------------------------
def soup_scene(rng, ntri, kind):
    s = sio.PackedScene()
    nv = max(3, ntri // 2 + 3)
    P = rng.uniform(-5, 5, (nv, 3)).astype(np.float32)
    if kind == "degenerate":
        P[: nv // 2] = P[0]                                   # half the vertices coincide: zero-area triangles
    elif kind == "huge":
        P *= np.float32(1e18)
    elif kind == "flat":
        P[:, 1] = 0.0                                         # everything in one plane: zero-thickness boxes
    I = rng.integers(0, nv, (ntri, 3)).astype(np.uint32 if kind == "u32" else np.uint16)
    N = np.tile(np.array([0, 1, 0], np.float32), (nv, 1)); T4 = np.tile(np.array([1, 0, 0, 1], np.float32), (nv, 1))
    UV = rng.uniform(0, 1, (nv, 2)).astype(np.float32)
    prims = synth._pack_geometry(s, [(P, N, T4, UV, I)], lambda k: 0)
    s.mesh_names.append("soup"); s.mesh_primitives.append(prims); s.mesh_transforms.append(sio.mat_identity())
    m = A.Material()
    for n in A._MAT_TEX:
        setattr(m, n + "TextureIndex", -1); setattr(m, n + "SamplerIndex", -1)
    s.materials.append(m)
    s.finalize_sun_light()
    return s


rng = np.random.default_rng(0xF22)
for kind in ("plain", "degenerate", "huge", "flat", "u32"):
    for ntri in (1, 2, 63, 64, 65, 127, 500, 4097):
        sc = soup_scene(rng, ntri, kind)
        d = sc.to_desc(); counts = (C.c_uint64 * 4)(); why = C.create_string_buffer(256)
        rc = L.san_build_scene(C.byref(d), counts, why, 256)
        assert rc == 0 and counts[0] == ntri, (kind, ntri, rc, why.value, list(counts))
        meshletize(sc, "soup-%s-%d" % (kind, ntri))
print("host sanitizer fuzz clean")
