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


def _decode(png_check, path):
    out = subprocess.run([png_check, path], capture_output=True, timeout=60)
    assert out.returncode == 0, path
    head, raw = out.stdout.split(b"\n", 1)
    w, h, c = (int(x) for x in head.split())
    return np.frombuffer(raw, dtype=np.uint8).reshape(h, w, c)


def _smooth_image(h, w, c, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([128 + 100 * np.sin(xx / 9.0 + k) * np.cos(yy / 7.0 - k) for k in range(c)], axis=-1)
    return np.clip(base + rng.normal(0, 6, size=(h, w, c)), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("mode,kw", [
    ("RGB", dict(subsampling=0)),                                   # 4:4:4 baseline
    ("RGB", dict(subsampling=1)),                                   # 4:2:2 (h2 filter)
    ("RGB", dict(subsampling=2)),                                   # 4:2:0 (h2v2 filter)
    ("RGB", dict(subsampling=2, progressive=True)),                 # progressive, spectral selection + refinement
    ("RGB", dict(subsampling=0, progressive=True, quality=95)),
    ("RGB", dict(subsampling=2, restart_marker_blocks=3)),          # DRI / RSTn
    ("RGB", dict(subsampling=0, optimize=True, quality=35)),        # optimised Huffman tables, coarse quantisation
    ("L", dict()),                                                  # greyscale -> 1 channel
    ("L", dict(progressive=True)),
])
def test_jpeg_reader_close_to_pillow(png_check, tmp_path, mode, kw):
    """JPEG decoders are only required to agree within a tolerance (IDCT rounding, chroma filter taps, colour
    matrix rounding differ between stb-style and libjpeg-turbo); entropy-decoding mistakes show up as errors of
    tens of grey levels or as a failed decode.  Odd sizes exercise partial MCUs."""
    c = 3 if mode == "RGB" else 1
    h, w = 45, 61
    img = _smooth_image(h, w, c, 11)
    path = str(tmp_path / "t.jpg")
    kw = dict(kw)
    kw.setdefault("quality", 85)
    Image.fromarray(img.squeeze() if c == 1 else img, mode).save(path, **kw)
    ref = np.asarray(Image.open(path)).reshape(h, w, c).astype(np.int32)
    got = _decode(png_check, path).astype(np.int32)
    assert got.shape == (h, w, c)
    d = np.abs(got - ref)
    sub = kw.get("subsampling", 0)
    # chroma upsampling filters differ at block edges when subsampled; luma/444 only differ by rounding
    # measured: 4:4:4 max 2 / mean 0.014, grey max 1, 4:2:0 max 2 / mean 0.07, 4:2:2 max 8 / mean 0.21
    assert d.max() <= {0: 2, 1: 10, 2: 4}[sub], (d.max(), d.mean())
    assert d.mean() <= {0: 0.1, 1: 0.4, 2: 0.2}[sub], d.mean()


def test_jpeg_reader_exact_on_flat_blocks(png_check, tmp_path):
    """DC-only 8x8 blocks: every conforming IDCT gives the same flat value, grey files have no colour
    transform => bit-exact against Pillow, which pins the entropy decoder + dequantisation + level shift."""
    img = np.kron(np.arange(30, 250, 220 // 24, dtype=np.uint8)[:24].reshape(4, 6), np.ones((8, 8), dtype=np.uint8))
    for prog in (False, True):
        path = str(tmp_path / f"flat{int(prog)}.jpg")
        Image.fromarray(img, "L").save(path, quality=100, progressive=prog)
        ref = np.asarray(Image.open(path))
        assert np.array_equal(_decode(png_check, path)[..., 0], ref)


def test_jpeg_reader_rejects_garbage(png_check, tmp_path):
    q = tmp_path / "junk.jpg"
    q.write_bytes(b"\xff\xd8\xff\xe0 definitely not a jpeg")
    assert subprocess.run([png_check, str(q)], capture_output=True).returncode != 0
