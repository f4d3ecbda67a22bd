import importlib
import os
import sys

import pytest

try:        # torch bundles its own HIP runtime: it has to be the first one loaded in a process that also loads libincr_hip.so
    import torch  # noqa: F401  (bench.py imports in the same order)
except Exception:   # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PKG = "incarnation-renderer_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def mod(name):
    return importlib.import_module(PKG + "." + name)


@pytest.fixture(scope="session")
def A():
    return mod("ctypes_api")


@pytest.fixture(scope="session")
def sio():
    return mod("scene_io")


@pytest.fixture(scope="session")
def synth():
    return mod("synth")


@pytest.fixture(scope="session")
def rt():
    return mod("runtime")


@pytest.fixture(scope="session")
def R():
    return mod("raster")


@pytest.fixture(scope="session")
def ob():
    from oracle import binding
    binding.lib()
    return binding


@pytest.fixture(scope="session")
def helmet(sio):
    s = sio.load_gltf(os.path.join(ROOT, "assets/damaged-helmet/DamagedHelmet.gltf"))
    return mod("raster").generate_meshlets(s)          # FScene::GenerateMeshlets at load time (scene.cpp:197)


@pytest.fixture(scope="session")
def suzanne(sio):
    """content/models/suzanne of the reference checkout: one mesh, 16-bit indices, shipped tangents, a camera under a two-node chain."""
    s = sio.load_gltf(os.path.join(ROOT, "assets/suzanne/Suzanne.gltf"), max_texture_size=512)
    return mod("raster").generate_meshlets(s)


@pytest.fixture(scope="session")
def transmission_test(sio):
    """content/models/transmission-test: 22 meshes under a node hierarchy, 32-bit indices, KHR_materials_transmission, MASK materials, JPEG + PNG images."""
    s = sio.load_gltf(os.path.join(ROOT, "assets/transmission-test/TransmissionTest.gltf"))
    return mod("raster").generate_meshlets(s)


@pytest.fixture(scope="session")
def helmet_oracle(ob, helmet):
    return ob.OracleScene(helmet)


@pytest.fixture(scope="session")
def sponza_small(synth):
    """Sponza-stat at 1/16 of the triangle budget with 64^2 textures (seconds on the CPU oracle)."""
    return mod("raster").generate_meshlets(synth.sponza_stat(texture_size=64, tri_scale=1.0 / 16.0))


@pytest.fixture(scope="session")
def sponza_small_oracle(ob, sponza_small):
    return ob.OracleScene(sponza_small)


@pytest.fixture(scope="session")
def gpu_ctx(rt):
    ctx = rt.Context(0)
    yield ctx
    ctx.close()


GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def sponza_decals(synth):
    """sponza_small plus four overlapping decal primitives (FScene::m_sceneMeshDecals) over the nave floor."""
    return mod("raster").generate_meshlets(synth.add_decals(synth.sponza_stat(texture_size=64, tri_scale=1.0 / 16.0)))
