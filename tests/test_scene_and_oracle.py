"""Host logic (scene packing, FView, host BVH build) and oracle regression against tests/golden."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


def test_helmet_packing(helmet, A):
    # 1 mesh, 1 primitive, 14,890 verts, 46,356 u16 indices, strides 12/12/16/8/2, 1 material, 5 textures, 0 lights
    assert len(helmet.mesh_primitives) == 1 and len(helmet.mesh_primitives[0]) == 1
    p = helmet.mesh_primitives[0][0]
    assert p["index_count"] == 46356 and helmet.num_triangles == 15452
    strides = [helmet.accessors[p[k]][2] for k in ("position", "normal", "tangent", "uv", "index")]
    assert strides == [12, 12, 16, 8, 2]
    assert len(helmet.materials) == 1 and len(helmet.textures) == 5 and len(helmet.global_lights) == 0
    m = helmet.materials[0]
    assert m.alphaMode == A.ALPHA_OPAQUE and list(m.emissiveFactor) == [1, 1, 1]
    fmts = {helmet.textures[getattr(m, n + "TextureIndex")][1] for n in ("emissive", "baseColor")}
    assert fmts == {A.TEX_RGBA8_SRGB}
    assert helmet.textures[m.normalTextureIndex][1] == A.TEX_RG_SNORM
    assert helmet.textures[m.metallicRoughnessTextureIndex][1] == A.TEX_BGRA8_SWIZZLE_RG
    # node rotation (90 deg about X) * RH2LH: row-vector convention, determinant -1 (mirrored instance)
    t = helmet.mesh_transforms[0].astype(np.float64)
    assert np.linalg.det(t[:3, :3]) == pytest.approx(-1.0, abs=1e-5)
    np.testing.assert_allclose(t[:3, :3] @ t[:3, :3].T, np.eye(3), atol=1e-6)


def test_view_matches_golden_and_is_orthonormal(helmet, sio):
    g = np.load(os.path.join(GOLDEN, "helmet_64.npz"))
    v = sio.FView().reset(helmet, aspect=1.0)
    np.testing.assert_array_equal(v.view_transform, g["view"])
    np.testing.assert_array_equal(v.projection_transform, g["proj"])
    r = v.view_transform[:3, :3].astype(np.float64)
    np.testing.assert_allclose(r.T @ r, np.eye(3), atol=1e-6)
    # reverse-Z infinite projection (demo.cpp:94-102): depth = near / z_view, far plane at depth 0
    P = v.projection_transform
    assert P[2, 2] == 0 and P[2, 3] == 1 and P[3, 2] == 1.0 and P[3, 3] == 0
    assert P[1, 1] == pytest.approx(1.0 / np.tan(0.39959652046304894 / 2), rel=1e-6)
    # camera sits at the glTF camera node (z mirrored into the left-handed world)
    np.testing.assert_allclose(v.position, [-3.0414262, 0.24117944, -5.2125583], rtol=1e-6)


def test_host_bvh_invariants(rt, A, helmet, sponza_small, synth):
    lib = rt.load()
    lib.incr_scene_validate_host.restype = C.c_int
    lib.incr_scene_validate_host.argtypes = [C.POINTER(A.SceneDesc), C.POINTER(A.SceneInfo)]
    for scene, ntri in ((helmet, 15452), (sponza_small, sponza_small.num_triangles)):
        d, info = scene.to_desc(), A.SceneInfo()
        assert lib.incr_scene_validate_host(C.byref(d), C.byref(info)) == 0, lib.incr_last_error()
        assert info.num_triangles == ntri and ntri // 64 <= info.num_bvh_nodes <= ntri
        slots = info.triangle_bytes // 48                   # four record slots per leaf child: triangles + holes (device_types.h LEAF SLOTS)
        assert info.triangle_bytes == slots * 48 and info.shading_bytes == ntri * 128 and ntri <= slots <= 4 * ntri and slots % 4 == 0
        assert info.bvh_bytes == info.num_bvh_nodes * 80
        assert info.bvh_max_depth <= 40
    # a broken desc is rejected with a message instead of being rendered
    bad = synth.sponza_stat(texture_size=8, tri_scale=1 / 64.0, with_lights=False)
    bad.mesh_primitives[0][3]["material"] = 99
    d = bad.to_desc()
    assert lib.incr_scene_validate_host(C.byref(d), C.byref(A.SceneInfo())) == -1
    assert b"materialIndex" in lib.incr_last_error()


def test_sponza_stat_statistics(synth, A):
    s = synth.sponza_stat(texture_size=8, with_lights=True)
    counts = [p["index_count"] for p in s.mesh_primitives[0]]
    assert counts == synth.SPONZA_INDEX_COUNTS and sum(counts) // 3 == 262267
    assert len(s.materials) == 25 and len(s.textures) == 73
    masked = [i for i, m in enumerate(s.materials) if m.alphaMode == A.ALPHA_MASKED]
    assert masked == [0, 1, 20] and all(s.materials[i].doubleSided for i in masked)
    assert len(s.global_lights) == 65 and s.global_lights[0].type == A.LIGHT_DIRECTIONAL
    assert all(s.accessors[p["index"]][2] == 2 for p in s.mesh_primitives[0])          # u16 indices
    # determinism: the generator is a pure function of its seed
    s2 = synth.sponza_stat(texture_size=8, with_lights=True)
    assert np.array_equal(s.buffers[0], s2.buffers[0]) and all(np.array_equal(a[0], b[0]) for a, b in zip(s.textures, s2.textures))


def test_oracle_matches_golden_helmet(helmet_oracle, helmet, sio, rt):
    g = np.load(os.path.join(GOLDEN, "helmet_64.npz"))
    W = H = 64
    view = sio.FView().reset(helmet, aspect=1.0)
    d = rt.make_pathtrace_desc(view, W, H, 1)
    ys, xs = np.mgrid[0:H, 0:W]
    o, dr = helmet_oracle.camera_rays(d, xs.ravel(), ys.ravel())
    np.testing.assert_array_equal(o.view(np.uint32), g["origins"].view(np.uint32))
    np.testing.assert_array_equal(dr.view(np.uint32), g["dirs"].view(np.uint32))
    ids, tuv = helmet_oracle.trace_rays(o, dr, 0.001, 10000.0)
    np.testing.assert_array_equal(ids, g["ids"])
    np.testing.assert_array_equal(tuv.view(np.uint32), g["tuv"].view(np.uint32))
    assert 0.2 < ids[:, 0].mean() < 0.9                         # the helmet covers a good part of the frame
    r = helmet_oracle.pathtrace(d, num_threads=2)
    np.testing.assert_array_equal(r["target"], g["target1"])
    hist = r["history"]
    for n in (2, 3, 4):
        d.current_sample_index = n
        r = helmet_oracle.pathtrace(d, history=hist, num_threads=2)
    np.testing.assert_array_equal(hist, g["history4"])
    st = r["stats"]
    assert [st.camera_rays, st.bounce_rays, st.sky_shadow_rays, st.light_shadow_rays, st.shaded_hits] == list(g["stats"][3])
    assert st.camera_rays == W * H and st.light_shadow_rays == 0          # the helmet scene has no lights


def test_oracle_thread_count_does_not_change_bits(helmet_oracle, helmet, sio, rt):
    view = sio.FView().reset(helmet, aspect=1.0)
    d = rt.make_pathtrace_desc(view, 48, 48, 3)
    a = helmet_oracle.pathtrace(d, num_threads=1)
    b = helmet_oracle.pathtrace(d, num_threads=5)
    np.testing.assert_array_equal(a["target"], b["target"])
    assert a["stats"].total_rays() == b["stats"].total_rays()


def test_oracle_integrate_is_a_running_mean(helmet_oracle, helmet, sio, rt, ob):
    """pathtrace-integrate.hlsl:27: history_n = ((n-1) * history_{n-1} + current) / n on R11G11B10 storage."""
    view = sio.FView().reset(helmet, aspect=1.0)
    d = rt.make_pathtrace_desc(view, 32, 32, 1)
    r1 = helmet_oracle.pathtrace(d)
    h1 = r1["history"].copy()
    d.current_sample_index = 2
    r2 = helmet_oracle.pathtrace(d, history=r1["history"])
    cur = r2["target"].view(np.float16).astype(np.float32)[..., :3]
    want = (np.float32(1.0) * ob.unpack_r11g11b10(h1) + cur) / np.float32(2.0)
    got = ob.unpack_r11g11b10(r2["history"])
    # truncating store: got <= want < got + one step (6/5-bit mantissas)
    fin = np.isfinite(want) & (want < 60000) & (want >= 0)                   # R11G11B10 has no sign: negatives store 0
    assert (got[fin] <= want[fin]).all()
    assert (want[fin] - got[fin] <= np.maximum(want[fin] * 2.0 ** -5, 2e-6)).all()


def test_masked_geometry_and_lights_in_oracle(sponza_small_oracle, sponza_small, sio, rt):
    view = sio.FView().reset(sponza_small, aspect=16 / 9)
    d = rt.make_pathtrace_desc(view, 96, 54, 1)
    r = sponza_small_oracle.pathtrace(d, num_threads=4)
    st = r["stats"]
    assert st.camera_rays == 96 * 54 and st.light_shadow_rays > 0 and st.sky_shadow_rays > 0
    t = r["target"].view(np.float16).astype(np.float32)
    assert np.isfinite(t).mean() > 0.99 and t[..., :3].max() > 0


def test_glb_and_data_uri_containers_load_identically(sio, tmp_path):
    """The same asset as .gltf + files, as .gltf with data: URIs and as a binary .glb (JSON + BIN chunks, bufferView images)
    must pack to identical scene arrays -- the containers tinygltf accepts (scene.cpp:113-130)."""
    import base64
    import json
    import struct
    src_dir = os.path.join(ROOT, "assets", "damaged-helmet")
    want = sio.load_gltf(os.path.join(src_dir, "DamagedHelmet.gltf"), max_texture_size=64)
    g = json.load(open(os.path.join(src_dir, "DamagedHelmet.gltf")))
    read = lambda uri: open(os.path.join(src_dir, uri), "rb").read()

    # (1) data: URIs
    gd = json.loads(json.dumps(g))
    for b in gd["buffers"]:
        b["uri"] = "data:application/octet-stream;base64," + base64.b64encode(read(b["uri"])).decode()
    for im in gd["images"]:
        im["uri"] = "data:image/jpeg;base64," + base64.b64encode(read(im["uri"])).decode()
    p1 = tmp_path / "inline.gltf"; p1.write_text(json.dumps(gd))

    # (2) GLB: one BIN chunk = the geometry buffer followed by the encoded images, each behind a new bufferView
    gb = json.loads(json.dumps(g))
    assert len(gb["buffers"]) == 1
    binary = bytearray(read(gb["buffers"][0]["uri"]))
    for im in gb["images"]:
        data = read(im.pop("uri"))
        while len(binary) % 4: binary.append(0)
        gb["bufferViews"].append({"buffer": 0, "byteOffset": len(binary), "byteLength": len(data)})
        im["bufferView"] = len(gb["bufferViews"]) - 1; im["mimeType"] = "image/jpeg"
        binary += data
    del gb["buffers"][0]["uri"]; gb["buffers"][0]["byteLength"] = len(binary)
    js = json.dumps(gb).encode(); js += b" " * (-len(js) % 4)
    binary += b"\0" * (-len(binary) % 4)
    p2 = tmp_path / "packed.glb"
    p2.write_bytes(struct.pack("<4sII", b"glTF", 2, 12 + 8 + len(js) + 8 + len(binary)) + struct.pack("<II", len(js), 0x4E4F534A) + js
                   + struct.pack("<II", len(binary), 0x004E4942) + bytes(binary))

    for p in (p1, p2):
        got = sio.load_gltf(str(p), max_texture_size=64)
        assert len(got.textures) == len(want.textures) and len(got.materials) == len(want.materials)
        for (a, fa), (b, fb) in zip(got.textures, want.textures):
            assert fa == fb and np.array_equal(a, b)
        assert len(got.mesh_primitives) == len(want.mesh_primitives) and got.accessors == want.accessors and got.num_triangles == want.num_triangles
        n = len(want.buffers[0]) - 32                                   # the loader pads every buffer
        assert np.array_equal(got.buffers[0][:n], want.buffers[0][:n])
        assert bytes(got.materials[0]) == bytes(want.materials[0])


def test_oracle_hit_chain_agrees_with_its_ray_queries(helmet_oracle, helmet, sio, rt):
    """incr_pathtrace_desc::debug_hit_chain (test instrumentation): vertex 0 of every pixel's chain is the closest hit of its camera ray."""
    W = H = 48
    view = sio.FView().reset(helmet, aspect=1.0)
    d = rt.make_pathtrace_desc(view, W, H, 1)
    r = helmet_oracle.pathtrace(d, num_threads=2, want_chain=True)
    ys, xs = np.mgrid[0:H, 0:W]
    o, dr = helmet_oracle.camera_rays(d, xs.ravel(), ys.ravel())
    ids, _ = helmet_oracle.trace_rays(o, dr, 0.001, 10000.0)
    c0 = r["chain"][:, :, 0, :2].reshape(-1, 2)
    hit = ids[:, 0] == 1
    assert hit.any() and (~hit).any()
    np.testing.assert_array_equal(c0[hit, 0], ids[hit, 1] << 16 | ids[hit, 2])
    np.testing.assert_array_equal(c0[hit, 1], ids[hit, 3])
    assert (c0[~hit] == 0xffffffff).all()
    # vertices beyond the end of a path keep the caller's fill; a miss at vertex 0 ends the path
    assert (r["chain"][:, :, 1:, :].reshape(-1, 3, 3)[~hit] == 0xfffffffe).all()
    assert (r["chain"][:, :, 0, 2].reshape(-1)[~hit] == 0).all() and (r["chain"][:, :, 0, 2].reshape(-1)[hit] >> 31).any()       # shadow word: some sky samples unoccluded


def test_skewed_scene_builds_within_the_traversal_depth(rt, A, synth, monkeypatch, sponza_small):
    """A geometric progression of triangles drives the SAH into a 60-level binary tree; the 8-wide tree must still fit the traversal
    stack (kMaxBvhDepth = 25 levels).  And the builder's fallback -- rebuild with the SAH cut off at a shallower binary depth, down to
    balanced median splits -- is exercised on an ordinary scene through its test hook."""
    lib = rt.load()
    lib.incr_scene_validate_host.restype = C.c_int
    lib.incr_scene_validate_host.argtypes = [C.POINTER(A.SceneDesc), C.POINTER(A.SceneInfo)]
    s = synth.triangle_soup(7, 200, kind="progression")
    d, info = s.to_desc(), A.SceneInfo()
    assert lib.incr_scene_validate_host(C.byref(d), C.byref(info)) == 0, lib.incr_last_error()
    assert info.num_triangles == 200 and info.bvh_max_depth <= 25
    d, base = sponza_small.to_desc(), A.SceneInfo()
    assert lib.incr_scene_validate_host(C.byref(d), C.byref(base)) == 0
    monkeypatch.setenv("INCR_BVH_DEPTH_LIMIT", str(base.bvh_max_depth - 1))
    forced = A.SceneInfo()
    assert lib.incr_scene_validate_host(C.byref(d), C.byref(forced)) == 0, lib.incr_last_error()        # check_bvh runs inside: every triangle in exactly one leaf, boxes conservative
    assert forced.num_triangles == base.num_triangles and forced.num_bvh_nodes != base.num_bvh_nodes     # another tree came out


def test_content_cache_round_trip(sio, tmp_path):
    """FConfig::UseContentCache (scene.cpp:13-89): the second load of a model takes its images from <model dir>/.content-cache and
    packs the same bytes; another size limit is another cache entry."""
    import shutil
    src = os.path.join(ROOT, "assets", "damaged-helmet")
    dst = tmp_path / "helmet"
    shutil.copytree(src, dst)
    path = str(dst / "DamagedHelmet.gltf")
    a = sio.load_gltf(path, max_texture_size=64, use_content_cache=True)
    assert a.content_cache_stats == {"hits": 0, "misses": 5} and len(os.listdir(dst / ".content-cache")) == 5
    b = sio.load_gltf(path, max_texture_size=64, use_content_cache=True)
    assert b.content_cache_stats == {"hits": 5, "misses": 0}
    for (ia, fa), (ib, fb) in zip(a.textures, b.textures):
        assert fa == fb and np.array_equal(ia, ib)
    c = sio.load_gltf(path, max_texture_size=32, use_content_cache=True)
    assert c.content_cache_stats["misses"] == 5 and len(os.listdir(dst / ".content-cache")) == 10
    plain = sio.load_gltf(path, max_texture_size=64)
    assert all(np.array_equal(x[0], y[0]) for x, y in zip(a.textures, plain.textures))


def test_oracle_viewmode_7_mirror_bounce(ob, rt, sio, A, helmet, helmet_oracle):
    """VIEWMODE 7 (pathtracing.hlsl:212-224) in the oracle: the first hit traces no shadow ray (its shadow word is 0) and always continues
    with one mirror ray; later vertices are ordinary.  VIEWMODE 6 and 8-13 have no branch in the shader: the normal path."""
    W = H = 48
    view = sio.FView().reset(helmet, aspect=1.0)
    NOT_TRACED = 0xfffffffe

    def run(vm):
        cfg = type("Cfg", (A.FConfig,), {"Viewmode": vm})
        return helmet_oracle.pathtrace(rt.make_pathtrace_desc(view, W, H, 2, cfg=cfg), num_threads=os.cpu_count() or 1, want_chain=True)
    n, r = run(0), run(7)
    hit0 = (r["chain"][..., 0, 0] != 0xffffffff) & (r["chain"][..., 0, 0] != NOT_TRACED)
    assert hit0.sum() > 200
    np.testing.assert_array_equal(r["chain"][..., 0, :2], n["chain"][..., 0, :2])          # same camera rays, same first hits
    assert (r["chain"][..., 0, 2][hit0] == 0).all() and (n["chain"][..., 0, 2][hit0] != 0).any()
    assert (r["chain"][..., 1, 0][hit0] != NOT_TRACED).all()
    assert r["stats"].bounce_rays >= int(hit0.sum()) and r["stats"].sky_shadow_rays < n["stats"].sky_shadow_rays
    assert not np.array_equal(r["target"], n["target"])
    for vm in (6, 8, 13):
        o = run(vm)
        np.testing.assert_array_equal(o["target"], n["target"]); np.testing.assert_array_equal(o["chain"], n["chain"])
