"""CPU tests of the host-side I/O rows: the PNG texture reader against Pillow."""
import os
import subprocess

import numpy as np
import pytest
from PIL import Image

from pooraytracer_amd import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def png_check(tmp_path_factory):
    build.build_host_example()
    exe = str(tmp_path_factory.mktemp("png") / "png_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", os.path.join(ROOT, "tests", "cpp", "png_check.cpp"), "-L",
                           os.path.dirname(build.HOST_LIB), "-Wl,-rpath," + os.path.dirname(build.HOST_LIB),
                           "-lpooraytracer_host", "-lprt_hip", "-o", exe])
    return exe


@pytest.mark.parametrize("mode,level", [("L", 6), ("LA", 9), ("RGB", 0), ("RGB", 1), ("RGB", 9), ("RGBA", 6)])
def test_png_reader_matches_pillow(png_check, tmp_path, mode, level):
    rng = np.random.default_rng(5)
    h, w = 37, 53
    c = {"L": 1, "LA": 2, "RGB": 3, "RGBA": 4}[mode]
    # smooth gradients + noise so that every PNG filter type and both Huffman block kinds get used
    yy, xx = np.mgrid[0:h, 0:w]
    img = ((xx * 3 + yy * 5)[..., None] + rng.integers(0, 40, size=(h, w, c))).astype(np.uint8)
    path = str(tmp_path / f"t_{mode}_{level}.png")
    Image.fromarray(img.squeeze() if c == 1 else img, mode).save(path, compress_level=level)
    out = subprocess.run([png_check, path], capture_output=True, timeout=60)
    assert out.returncode == 0
    head, raw = out.stdout.split(b"\n", 1)
    assert [int(x) for x in head.split()] == [w, h, c]
    assert np.array_equal(np.frombuffer(raw, dtype=np.uint8).reshape(h, w, c), img)


def test_png_reader_rejects_unsupported(png_check, tmp_path):
    p = str(tmp_path / "pal.png")
    Image.fromarray(np.zeros((4, 4), dtype=np.uint8), "P").save(p)
    assert subprocess.run([png_check, p], capture_output=True).returncode != 0
    q = tmp_path / "junk.png"
    q.write_bytes(b"not a png at all")
    assert subprocess.run([png_check, str(q)], capture_output=True).returncode != 0
