"""Regenerates tests/golden/*.npz from the CPU oracle (oracle/pt_oracle.cpp).

These fixtures pin the ORACLE against regressions and give the GPU tests committed expected values;
they are NOT reference outputs (the reference cannot be built or run here: PARITY UNPINNED).
Usage: python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from pooraytracer_amd import scenes  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def make(name, sc, spp, max_depth, seed):
    o = oracle.Oracle(sc)
    img, cnt = o.render(spp=spp, max_depth=max_depth, seed=seed)
    lo, hi = sc.bounds()
    rays = scenes.random_rays(4096, lo, hi, seed=77)
    h = o.trace_closest(rays)
    meta = dict(spp=spp, max_depth=max_depth, seed=seed, counters=cnt)
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), image=img, rays=rays.view(np.float64).reshape(-1, 8),
                        hit_t=h["t"], hit_prim=h["prim"], hit_alpha=h["alpha"], hit_beta=h["beta"],
                        hit_front=h["front"], light_order=o.light_order(), meta=json.dumps(meta))
    print(name, img.mean(axis=(0, 1)), cnt)


if __name__ == "__main__":
    make("tiny_cornell", scenes.tiny_scene(), spp=8, max_depth=10, seed=1)
    make("mixed", scenes.mixed_materials(), spp=8, max_depth=8, seed=5)
