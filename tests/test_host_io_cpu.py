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


# ---------------------------------------------------------------------------------- OBJ / MTL / XML loader behaviour
@pytest.fixture(scope="module")
def model_dump(tmp_path_factory):
    build.build_host_example()
    exe = str(tmp_path_factory.mktemp("model") / "model_dump")
    lib_dir = os.path.dirname(build.HOST_LIB)
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "model_dump.cpp"),
                           "-L", lib_dir, "-Wl,-rpath," + lib_dir, "-lpooraytracer_host", "-lprt_hip", "-o", exe])
    return exe


def _write_scene(d, name, obj, mtl, xml):
    os.makedirs(d / name, exist_ok=True)
    (d / name / f"{name}.obj").write_text(obj)
    (d / name / f"{name}.mtl").write_text(mtl)
    (d / name / f"{name}.xml").write_text(xml)
    return str(d / name)


CAM_XML = ('<?xml version="1.0"?>\n<camera type="perspective" width="16" height="16" fovy="40">\n<eye x="0" y="0" z="5"/>'
           '<lookat x="0" y="0" z="0"/><up x="0" y="1" z="0"/></camera>\n')


def test_loader_triangulates_polygons_as_a_fan(model_dump, tmp_path):
    """The reference parses with tinyobjloader's default reader configuration (Source/Model.cpp:62-66: triangulate
    defaults to true), so `Model` never sees a face with more than three vertices and its `fv != 3` guard
    (Model.cpp:138-141) is not reached; this loader does the triangulation itself, as a fan around the first vertex
    (v0 v1 v2, v0 v2 v3, ...), and keeps triangles untouched.  Documented in INTEGRATION.md."""
    obj = ("mtllib poly.mtl\ng quad\nusemtl DiffuseWhite\n"
           "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 2 0 1\nv 3 0 1\nv 3.5 1 1\nv 2.5 2 1\nv 1.5 1 1\n"
           "vt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\n"
           "f 1/1 2/2 3/3 4/4\n"          # quad
           "g penta\nusemtl DiffuseWhite\n"
           "f 5 6 7 8 9\n"                # pentagon, no texture coordinates
           "f -5 -4 -3\n")                # relative indices: the triangle (5, 6, 7)
    mtl = "newmtl DiffuseWhite\nKd 0.5 0.5 0.5\n"
    d = _write_scene(tmp_path, "poly", obj, mtl, CAM_XML)
    r = subprocess.run([model_dump, d, "poly"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.splitlines()
    meshes = [x for x in lines if x.startswith("mesh")]
    assert [m.split()[1] for m in meshes] == ["quad", "penta"] and meshes[0].endswith("tris 2") and meshes[1].endswith("tris 4")
    tris = np.array([[float(v) for v in x.split()[1:]] for x in lines if x.startswith("t ")])
    V = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [2, 0, 1], [3, 0, 1], [3.5, 1, 1], [2.5, 2, 1], [1.5, 1, 1]], dtype=float)
    want = [(0, 1, 2), (0, 2, 3), (4, 5, 6), (4, 6, 7), (4, 7, 8), (4, 5, 6)]
    assert np.array_equal(tris[:, :9].reshape(-1, 3, 3), np.array([[V[i] for i in w] for w in want]))
    # texture coordinates follow their vertices through the fan; faces without them get the degenerate-UV fix-up (Model.cpp:170-175)
    assert np.array_equal(tris[0, 9:], [0, 0, 1, 0, 1, 1]) and np.array_equal(tris[1, 9:], [0, 0, 1, 1, 0, 1])
    assert np.array_equal(tris[2, 9:], [0, 0, 1, 0, 1, 1])


def test_loader_light_without_radiance_throws_like_the_reference(model_dump, tmp_path):
    """`lightRadianceMap.at(mtlname)` (Source/Model.cpp:294): a material whose NAME makes it a DiffuseLight
    (light1..4 / Light, Model.cpp:16-51) but that has no <light mtlname=...> entry in the XML makes the reference
    throw std::out_of_range out of the Model constructor; so does this loader.  Unknown names become Lambertian."""
    obj = "mtllib l.mtl\ng a\nusemtl light2\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\ng b\nusemtl SomethingElse\nf 1 2 3\n"
    mtl = "newmtl light2\nKd 0 0 0\nnewmtl SomethingElse\nKd 0.2 0.3 0.4\n"
    d = _write_scene(tmp_path, "l", obj, mtl, CAM_XML)   # no <light> element at all
    r = subprocess.run([model_dump, d, "l"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and r.stdout.startswith("refused:"), r.stdout
    d = _write_scene(tmp_path, "l", obj, mtl, CAM_XML + '<light mtlname="light2" radiance="3,2,1"/>\n')
    r = subprocess.run([model_dump, d, "l"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout
    meshes = [x for x in r.stdout.splitlines() if x.startswith("mesh")]
    assert "emission 1" in meshes[0] and "emission 0 skipNEE 0" in meshes[1]
    # a radiance for a name that is not in the light table is ignored, and a light in the table but with another
    # material's radiance only still throws
    d = _write_scene(tmp_path, "l", obj, mtl, CAM_XML + '<light mtlname="SomethingElse" radiance="3,2,1"/>\n')
    assert subprocess.run([model_dump, d, "l"], capture_output=True, text=True, timeout=60).returncode == 1
