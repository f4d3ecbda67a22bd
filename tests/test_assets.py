"""The reference's other complete assets (content/models/suzanne, content/models/transmission-test, copied under assets/ as data) through the
glTF loader: what the packed scene holds must be what the glTF JSON states (FScene::LoadNode / LoadMesh / LoadMaterial, scene.cpp:228-369, 850-1010)."""
import json
import os

import numpy as np

from conftest import ROOT


def gltf_json(rel):
    with open(os.path.join(ROOT, rel)) as f:
        return json.load(f)


def node_local(n, sio):
    if "matrix" in n:
        return np.array(n["matrix"], np.float32).reshape(4, 4)
    T = sio.mat_translation(*n["translation"]) if "translation" in n else sio.mat_identity()
    Rm = sio.mat_from_quaternion(*n["rotation"]) if "rotation" in n else sio.mat_identity()
    S = sio.mat_scale(*n["scale"]) if "scale" in n else sio.mat_identity()
    return sio.mat_mul(sio.mat_mul(S, Rm), T)


def test_suzanne_matches_its_json(sio, A, suzanne, rt):
    g = gltf_json("assets/suzanne/Suzanne.gltf")
    assert len(g["nodes"]) == 3 and len(suzanne.mesh_primitives) == 1 and len(suzanne.mesh_primitives[0]) == 1
    prim = suzanne.mesh_primitives[0][0]
    ia = g["accessors"][g["meshes"][0]["primitives"][0]["indices"]]
    assert ia["componentType"] == 5123 and prim["index_count"] == ia["count"] == 11808          # the 16-bit index path
    assert suzanne.num_triangles == ia["count"] // 3
    # shipped tangents are used as they are: the material has no normal map, and FixupMeshes (mesh-utils.cpp:576) only generates for one that has
    assert prim["tangent"] == g["meshes"][0]["primitives"][0]["attributes"]["TANGENT"]
    m = suzanne.materials[0]
    assert m.alphaMode == A.ALPHA_OPAQUE and m.doubleSided == 0 and m.normalTextureIndex == -1 and m.emissiveTextureIndex == -1
    assert m.baseColorTextureIndex >= 0 and m.metallicRoughnessTextureIndex >= 0 and m.baseColorTextureIndex != m.metallicRoughnessTextureIndex
    assert (m.metallicFactor, m.roughnessFactor, m.transmissionFactor) == (1.0, 1.0, 0.0)
    assert suzanne.textures[m.baseColorTextureIndex][1] == A.TEX_RGBA8_SRGB and suzanne.textures[m.metallicRoughnessTextureIndex][1] == A.TEX_BGRA8_SWIZZLE_RG
    # the camera hangs under a two-node chain: world = orientation x camera x RH2LH (row vectors, scene.cpp:250-256)
    RH2LH = sio.mat_identity(); RH2LH[2, 2] = -1
    cam_world = sio.mat_mul(node_local(g["nodes"][1], sio), sio.mat_mul(node_local(g["nodes"][2], sio), RH2LH))
    assert len(suzanne.cameras) == 1
    np.testing.assert_array_equal(suzanne.cameras[0]["transform"], cam_world)
    np.testing.assert_array_equal(suzanne.mesh_transforms[0], RH2LH)
    # MikkTSpace on this mesh against the tangents the asset ships (Blender's exporter = mikktspace.c): same test as the helmet's
    at = g["meshes"][0]["primitives"][0]["attributes"]
    nv = g["accessors"][at["POSITION"]]["count"]
    pos = np.ascontiguousarray(suzanne.accessor_array(at["POSITION"], np.float32, 3, nv))
    nrm = np.ascontiguousarray(suzanne.accessor_array(at["NORMAL"], np.float32, 3, nv))
    uv = np.ascontiguousarray(suzanne.accessor_array(at["TEXCOORD_0"], np.float32, 2, nv))
    idx = np.ascontiguousarray(suzanne.accessor_array(ia and g["meshes"][0]["primitives"][0]["indices"], np.uint16, 1, ia["count"]).astype(np.uint32).reshape(-1, 3))
    shipped = np.array(suzanne.accessor_array(at["TANGENT"], np.float32, 4, nv))
    out = np.zeros((nv, 4), np.float32)
    rt.check(rt.load().incr_generate_tangents(pos.ctypes.data, nrm.ctypes.data, uv.ctypes.data, idx.ctypes.data, nv, idx.shape[0], out.ctypes.data))
    d = np.abs((out[:, :3] * shipped[:, :3]).sum(axis=1))          # |cos| of the angle between generated and shipped tangent
    assert np.isfinite(out).all() and (d > 0.999).mean() > 0.97, float((d > 0.999).mean())
    # the sign is the opposite one: the reference (and this build) run MikkTSpace on glTF's V-down UVs (quirk Q17, tests/test_mikktspace.py)
    assert ((out[:, 3] * shipped[:, 3]) < 0).mean() > 0.97


def test_transmission_test_matches_its_json(sio, A, transmission_test):
    g = gltf_json("assets/transmission-test/TransmissionTest.gltf")
    s = transmission_test
    assert len(g["nodes"]) == 25 and len(g["meshes"]) == 22 and len(g["materials"]) == 14
    # every node with a mesh becomes one mesh entity, in depth-first node order, with world = local x parent x RH2LH
    RH2LH = sio.mat_identity(); RH2LH[2, 2] = -1
    expected = []
    def walk(ni, parent):
        n = g["nodes"][ni]
        world = sio.mat_mul(node_local(n, sio), parent)
        if "mesh" in n:
            expected.append((g["meshes"][n["mesh"]].get("name", ""), n["mesh"], world))
        for c in n.get("children", []):
            walk(c, world)
    for r in g["scenes"][0]["nodes"]:
        walk(r, RH2LH)
    assert len(expected) == len(s.mesh_primitives) == 22
    assert any("children" in n for n in g["nodes"])          # a real hierarchy: some meshes take their transform through a parent
    for (name, mi, world), got_name, got_world, prims in zip(expected, s.mesh_names, s.mesh_transforms, s.mesh_primitives):
        assert got_name == name
        np.testing.assert_array_equal(got_world, world)
        assert len(prims) == len(g["meshes"][mi]["primitives"])
        for p, jp in zip(prims, g["meshes"][mi]["primitives"]):
            ia = g["accessors"][jp["indices"]]
            assert ia["componentType"] == 5125 and p["index_count"] == ia["count"] and p["material"] == jp["material"]          # the 32-bit index path
            assert p["vertex_count"] == g["accessors"][jp["attributes"]["POSITION"]]["count"]
    assert s.num_triangles == sum(g["accessors"][p["indices"]]["count"] // 3 for n in g["nodes"] if "mesh" in n for p in g["meshes"][n["mesh"]]["primitives"])
    # materials: alpha mode, double-sidedness, transmission factor and map exactly as the JSON states them (scene.cpp:939-951, 1003-1011)
    modes = {"OPAQUE": A.ALPHA_OPAQUE, "MASK": A.ALPHA_MASKED, "BLEND": A.ALPHA_BLEND}
    n_mask = n_trans_tex = 0
    for jm, m in zip(g["materials"], s.materials):
        assert m.alphaMode == modes[jm.get("alphaMode", "OPAQUE")]
        assert m.doubleSided == int(jm.get("doubleSided", False))
        tr = jm.get("extensions", {}).get("KHR_materials_transmission", {})
        assert m.transmissionFactor == np.float32(tr.get("transmissionFactor", 0.0))
        assert (m.transmissionTextureIndex >= 0) == ("transmissionTexture" in tr)
        if "transmissionTexture" in tr:
            assert s.textures[m.transmissionTextureIndex][1] == A.TEX_RGBA8_UNORM; n_trans_tex += 1
        n_mask += m.alphaMode == A.ALPHA_MASKED
        pbr = jm.get("pbrMetallicRoughness", {})
        assert (m.baseColorTextureIndex >= 0) == ("baseColorTexture" in pbr) and (m.metallicRoughnessTextureIndex >= 0) == ("metallicRoughnessTexture" in pbr)
        assert (m.emissiveTextureIndex >= 0) == ("emissiveTexture" in jm)
        np.testing.assert_array_equal(np.array(m.baseColorFactor[:], np.float32), np.array(pbr.get("baseColorFactor", [1, 1, 1, 1])[:3], np.float32))
    assert n_mask == 6 and n_trans_tex == 6
    # images: three JPEGs and six PNGs, one texture slot per (image, format) use
    assert sum(im["uri"].endswith(".jpg") for im in g["images"]) == 3 and len(s.textures) >= 7
    assert len(s.cameras) == 1 and abs(s.cameras[0]["yfov"] - g["cameras"][0]["perspective"]["yfov"]) < 1e-7
