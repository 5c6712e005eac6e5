"""Host-side code under AddressSanitizer + UndefinedBehaviorSanitizer (CPU builds only; SURVEY.md §5 "race
detection / sanitizers").  Covers the code that parses untrusted files — the PNG and JPEG texture readers, fed
hundreds of corrupted inputs — and the CPU oracle on a scene that exercises every code path it has."""
import os
import subprocess

import numpy as np
import pytest
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-O1", "-g", "-std=c++17"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=99", UBSAN_OPTIONS="halt_on_error=1:exitcode=98")


@pytest.fixture(scope="module")
def image_check_san(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("san") / "image_check_san")
    host = os.path.join(ROOT, "pooraytracer_amd", "host")
    subprocess.check_call(["g++"] + SAN + [os.path.join(ROOT, "tests", "cpp", "png_check.cpp"), os.path.join(host, "png_decode.cpp"),
                                         os.path.join(host, "jpeg_decode.cpp"), "-o", exe])
    return exe


def _corruptions(data, rng, n):
    for k in range(n):
        b = bytearray(data)
        mode = k % 4
        if mode == 0:    # truncate
            b = b[:rng.integers(1, len(b))]
        elif mode == 1:  # flip a few bytes
            for _ in range(int(rng.integers(1, 6))):
                b[rng.integers(0, len(b))] = rng.integers(0, 256)
        elif mode == 2:  # overwrite a run with one value (kills headers / tables / lengths)
            i = int(rng.integers(0, len(b) - 8))
            b[i:i + int(rng.integers(2, 40))] = bytes([int(rng.integers(0, 256))]) * 8
        else:            # duplicate a slice in the middle
            i, j = sorted(int(x) for x in rng.integers(0, len(b), size=2))
            b = b[:j] + b[i:j] + b[j:]
        yield bytes(b)


def test_texture_readers_survive_corrupted_files(image_check_san, tmp_path):
    rng = np.random.default_rng(2024)
    yy, xx = np.mgrid[0:40, 0:56]
    img = np.stack([(xx * 4) % 256, (yy * 6) % 256, (xx + yy) * 2 % 256], -1).astype(np.uint8)
    sources = {}
    Image.fromarray(img, "RGB").save(tmp_path / "a.png")
    Image.fromarray(img, "RGB").save(tmp_path / "b.jpg", quality=80)
    Image.fromarray(img, "RGB").save(tmp_path / "c.jpg", quality=80, progressive=True, subsampling=2)
    Image.fromarray(img[..., 0], "L").save(tmp_path / "d.jpg", quality=60, restart_marker_blocks=2)
    for f in ("a.png", "b.jpg", "c.jpg", "d.jpg"):
        sources[f] = (tmp_path / f).read_bytes()
        ok = subprocess.run([image_check_san, str(tmp_path / f)], capture_output=True, env=ENV, timeout=60)
        assert ok.returncode == 0, ok.stderr.decode()[-2000:]
    victim = tmp_path / "x.bin"
    decoded = rejected = 0
    for name, data in sources.items():
        for bad in _corruptions(data, rng, 120):
            victim.write_bytes(bad)
            r = subprocess.run([image_check_san, str(victim)], capture_output=True, env=ENV, timeout=60)
            # 0 = decoded something, 1 = rejected; anything else is a sanitizer report or a crash
            assert r.returncode in (0, 1), (name, r.returncode, r.stderr.decode()[-3000:])
            decoded += r.returncode == 0
            rejected += r.returncode == 1
    assert rejected > 50 and decoded > 10  # both outcomes were exercised


def test_oracle_clean_under_sanitizers(tmp_path):
    exe = str(tmp_path / "oracle_san")
    subprocess.check_call(["g++"] + SAN + ["-ffp-contract=off", "-pthread", "-I", os.path.join(ROOT, "oracle"),
                                         os.path.join(ROOT, "tests", "cpp", "oracle_sanitize.cpp"),
                                         os.path.join(ROOT, "oracle", "pt_oracle.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, env=ENV, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-3000:] + r.stdout.decode()
    assert r.stdout.decode().startswith("ok ")


def test_host_scene_preparation_under_sanitizers(tmp_path):
    """Triangle precompute, SAH BVH + 16-bit box quantisation and the light tree of libprt_hip are plain host C++:
    built with ASan/UBSan and run on random + degenerate input, with the tree invariants checked in the harness."""
    exe = str(tmp_path / "host_build_san")
    csrc = os.path.join(ROOT, "pooraytracer_amd", "csrc")
    subprocess.check_call(["g++"] + SAN + [os.path.join(ROOT, "tests", "cpp", "host_build_sanitize.cpp"),
                                         os.path.join(csrc, "bvh_build.cpp"), os.path.join(csrc, "scene_setup.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, env=ENV, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:] + r.stdout.decode()
    assert r.stdout.decode().count(" ok") == 7  # five sizes at the origin + two scenes far from it


def test_obj_mtl_xml_loader_survives_corrupted_files(tmp_path):
    """The OBJ/MTL/XML loader (pooraytracer_amd/host/model.cpp, Camera::SetViewParametersByXmlFile) parses text it
    did not write: under ASan/UBSan it must either load or throw on mangled input, never crash or read out of
    bounds.  Links the (uninstrumented) libprt_hip.so only to resolve symbols; no device call is made."""
    from pooraytracer_amd import build, scenes
    build.build_host_example()
    exe = str(tmp_path / "model_san")
    host = os.path.join(ROOT, "pooraytracer_amd", "host")
    lib_dir = os.path.dirname(build.LIB)
    subprocess.check_call(["g++"] + SAN + ["-pthread", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "model_sanitize.cpp")] +
                          [os.path.join(host, f) for f in ("model.cpp", "host_api.cpp", "png_decode.cpp", "jpeg_decode.cpp")] +
                          ["-L", lib_dir, "-Wl,-rpath," + lib_dir, "-lprt_hip", "-o", exe])
    data = scenes.mixed_materials(16, 16)
    res = tmp_path / "res"
    scenes.export_obj(data, str(res), texture_format="png")
    d = res / data.name
    ok = subprocess.run([exe, str(d), data.name], capture_output=True, env=ENV, timeout=120)
    assert ok.returncode == 0, ok.stderr.decode()[-3000:] + ok.stdout.decode()
    rng = np.random.default_rng(7)
    outcomes = {0: 0, 1: 0}
    for ext in ("obj", "mtl", "xml"):
        path = d / f"{data.name}.{ext}"
        good = path.read_bytes()
        for bad in _corruptions(good, rng, 60):
            path.write_bytes(bad)
            r = subprocess.run([exe, str(d), data.name], capture_output=True, env=ENV, timeout=120)
            assert r.returncode in (0, 1), (ext, r.returncode, r.stderr.decode()[-3000:])
            outcomes[r.returncode] += 1
        path.write_bytes(good)
    assert outcomes[0] > 10 and outcomes[1] > 10
