"""The PNG writer of examples/render.py (the headless stand-in for the reference's window)."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_png_writer_round_trip(tmp_path):
    spec = importlib.util.spec_from_file_location("render_example", os.path.join(ROOT, "examples", "render.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    rng = np.random.default_rng(5)
    for shape in ((1, 1), (7, 13), (64, 48)):
        img = rng.integers(0, 256, size=shape + (4,), dtype=np.uint8)
        p = str(tmp_path / ("t%dx%d.png" % shape))
        m.write_png(p, img)
        np.testing.assert_array_equal(m.read_png(p), img)
