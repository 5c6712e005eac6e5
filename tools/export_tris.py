"""Developer tool: triangles of a stand-in scene as raw doubles [n][3][3] for the CPU simulators (tools/sim_oct8.cpp).
usage: python tools/export_tris.py cornell_box|veach_mis|bathroom out.f64"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pooraytracer_amd import scenes  # noqa: E402

data = getattr(scenes, sys.argv[1])()
v = np.ascontiguousarray(data.vertices, dtype=np.float64).reshape(-1, 3, 3)
v.tofile(sys.argv[2])
print(sys.argv[2], v.shape[0], "triangles")
