"""The output stage against the ONLY golden data the reference holds: the `.hdr` + `.png` pairs in its Results/ directory,
both written by `Camera::WriteColorAttachment` (reference Source/Camera.cpp:279-331; `LinearToSRGB` :206-221) from the
same `colorAttachment`.  tests/golden/results_pairs.npz holds crops and scanlines cut from them by
tests/golden/make_results_pairs.py (data: pixel bytes, encoded bytes, checksums).

What they pin (the hot path itself stays unpinned — the scenes that produced them are absent):
  * the .hdr stores every linear float v as a byte m = trunc(v * 2^(136-e)) under the pixel's shared exponent e, so
    v lies in [m q, (m+1) q), q = 2^(e-136); the .png stores trunc(clamp(sRGB(v), 0, 0.9999) * 255).  The product's stage
    (NaN scrub + sRGB + clamp + truncation; host `Camera::WriteColorAttachment` and the GPU's k_tonemap) is monotone, so
    fed the two ends of that interval it must BRACKET the reference's byte, and fed the lower end it must reproduce most
    bytes exactly (wherever the quantum does not straddle an 8-bit step);
  * the product's .hdr writer, fed the decoded floats, must write the reference's bytes back: same header, same
    per-component run-length coding (stb_image_write's) — scanline by scanline here, whole files where the reference is present;
  * the product's PNG reader must decode the reference's stb-written PNGs exactly as Pillow does (run where the reference
    is present; the checksums of Pillow's decodes travel in the fixture).
"""
import hashlib
import os
import subprocess
import zlib

import numpy as np
import pytest

from pooraytracer_amd import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RESULTS = "/root/reference/Results"
FIX = os.path.join(ROOT, "tests", "golden", "results_pairs.npz")


@pytest.fixture(scope="module")
def fx():
    return np.load(FIX)


@pytest.fixture(scope="module")
def tools(tmp_path_factory):
    build.build_host_example()
    d = tmp_path_factory.mktemp("outstage")
    libdir = os.path.dirname(build.HOST_LIB)
    exes = {}
    for name in ("output_check", "png_check"):
        exes[name] = str(d / name)
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", name + ".cpp"),
                               "-L", libdir, "-Wl,-rpath," + libdir, "-lpooraytracer_host", "-lprt_hip", "-o", exes[name]])
    exes["dir"] = d
    return exes


def rgbe_interval(rgbe):
    """(lo, hi) float32 ends of the interval of linear values a stored RGBE pixel stands for: [m q, (m+1) q)."""
    e = rgbe[..., 3].astype(np.int32)
    q = np.where(e > 0, np.ldexp(1.0, e - 136), 0.0)[..., None]
    m = rgbe[..., :3].astype(np.float64)
    lo = (m * q).astype(np.float32)
    hi = np.nextafter(((m + 1.0) * q).astype(np.float32), np.float32(0))  # the largest float below (m+1) q
    # an all-zero pixel: max component below 1e-32
    hi = np.where(e[..., None] > 0, hi, np.float32(1e-32))
    return lo, hi


def numpy_stage(x):
    """Camera.cpp:206-221,291-301 restated in numpy (doubles, like the reference)."""
    x = np.nan_to_num(np.asarray(x, dtype=np.float64), nan=0.0)
    s = np.where(x <= 0.0031308, 12.92 * x, 1.055 * np.power(np.maximum(x, 0), 1 / 2.4) - 0.055)
    return (np.clip(s, 0.0, 0.9999) * 255).astype(np.uint8)


def decode_hdr(data):
    from tests.golden.make_results_pairs import parse_hdr
    return parse_hdr(data)


def run_stage(tools, rgb_f32, tag):
    """Camera::WriteColorAttachment of the host library on these floats -> (png pixels via Pillow, .hdr file bytes)."""
    from PIL import Image
    h, w = rgb_f32.shape[:2]
    raw = tools["dir"] / f"{tag}.raw"
    np.ascontiguousarray(rgb_f32, dtype=np.float32).tofile(raw)
    png = tools["dir"] / f"{tag}.png"
    subprocess.check_call([tools["output_check"], str(w), str(h), str(raw), str(png)], timeout=120)
    return np.asarray(Image.open(png).convert("RGB")), open(str(png)[:-4] + ".hdr", "rb").read()


def check_bracket(got_lo, got_hi, ref_png, min_exact, max_width=12):
    assert (got_lo <= ref_png).all() and (ref_png <= got_hi).all(), \
        (int((got_lo > ref_png).sum()), int((ref_png > got_hi).sum()))
    exact = float((got_lo == ref_png).mean())
    assert exact >= min_exact, exact
    # the brackets are tight: a few 8-bit steps where a dark component shares a bright pixel's exponent, none or one elsewhere
    width = got_hi.astype(int) - got_lo.astype(int)
    assert width.max() <= max_width and (width <= 1).mean() > 0.90, (int(width.max()), float((width <= 1).mean()))
    return exact


def test_fixture_has_what_the_docstring_says(fx):
    assert fx["crop_rgbe"].shape == (12, 128, 128, 4) and fx["crop_png"].shape == (12, 128, 128, 3)
    assert len(fx["pairs"]) == 4 and all(str(p).startswith("bathroom2_") for p in fx["pairs"])
    assert bytes(fx["hdr_header"]).startswith(b"#?RADIANCE\n# Written by stb_image_write.h\n")
    # the crops cover dark and bright regions, clamped pixels included
    lo, _ = rgbe_interval(fx["crop_rgbe"])
    assert lo.max() > 1.0 and (fx["crop_png"] == 254).any() and (fx["crop_png"] < 30).any()


def test_numpy_restatement_of_the_stage_brackets_the_reference_png(fx):
    """The restatement the GPU tone-map test is checked against (tests/test_gpu_parity.py) is itself pinned here."""
    lo, hi = rgbe_interval(fx["crop_rgbe"])
    exact = check_bracket(numpy_stage(lo), numpy_stage(hi), fx["crop_png"], 0.70)
    print(f"numpy stage: {exact:.3f} of the bytes exact from the lower end of the RGBE interval")


def test_host_output_stage_brackets_the_reference_png_and_rewrites_its_hdr(fx, tools):
    """`Camera::WriteColorAttachment` of libpooraytracer_host on the decoded floats of all twelve crops (stacked into one
    128 x 1536 image): PNG bytes bracket the reference's, >= 70 % exact; the .hdr it writes decodes to the same RGBE bytes."""
    rgbe = fx["crop_rgbe"].reshape(-1, 128, 4)
    ref = fx["crop_png"].reshape(-1, 128, 3)
    lo, hi = rgbe_interval(rgbe)
    p_lo, hdr_lo = run_stage(tools, lo, "lo")
    p_hi, _ = run_stage(tools, hi, "hi")
    exact = check_bracket(p_lo, p_hi, ref, 0.70)
    assert np.array_equal(p_lo, numpy_stage(lo)) and np.array_equal(p_hi, numpy_stage(hi))
    _, back, _ = decode_hdr(hdr_lo)
    assert np.array_equal(back, rgbe)
    print(f"host stage: {exact:.3f} exact")


def test_host_hdr_writer_reproduces_the_reference_scanlines_byte_for_byte(fx, tools):
    """Sixteen whole scanlines of the reference's files with their ENCODED bytes: the writer, fed the decoded floats of a
    row, must emit exactly those bytes (header comment + EXPOSURE line + dimensions, then stb's per-component RLE)."""
    offs = fx["rle_row_offsets"]
    blob = fx["rle_row_bytes"].tobytes()
    header = bytes(fx["hdr_header"])
    for k, row in enumerate(fx["rle_row_rgbe"]):
        lo, _ = rgbe_interval(row[None])
        _, hdr = run_stage(tools, lo, f"row{k}")
        want_header = header.replace(b"-Y 720 +X 1280", b"-Y 1 +X 1280")
        assert hdr[:len(want_header)] == want_header
        assert hdr[len(want_header):] == blob[offs[k]:offs[k + 1]], k


@pytest.mark.skipif(not os.path.isdir(RESULTS), reason="the reference's Results/ exists in the build container only")
def test_whole_reference_hdr_files_round_trip_through_the_product_writer(fx, tools):
    """Where the reference is present: decode each .hdr, hand the floats to Camera::WriteColorAttachment, and the file it
    writes has the reference file's sha256; its PNG brackets / mostly equals the reference's PNG over the WHOLE frame."""
    from PIL import Image
    for name, sha in zip(fx["pairs"], fx["file_sha256"]):
        data = open(os.path.join(RESULTS, str(name) + ".hdr"), "rb").read()
        assert hashlib.sha256(data).hexdigest() == str(sha)
        _, px, _ = decode_hdr(data)
        lo, hi = rgbe_interval(px)
        p_lo, hdr = run_stage(tools, lo, "full_lo")
        assert hashlib.sha256(hdr).hexdigest() == str(sha), name
        p_hi, _ = run_stage(tools, hi, "full_hi")
        ref = np.asarray(Image.open(os.path.join(RESULTS, str(name) + ".png")).convert("RGB"))
        check_bracket(p_lo, p_hi, ref, 0.70, max_width=40)


@pytest.mark.skipif(not os.path.isdir(RESULTS), reason="the reference's Results/ exists in the build container only")
def test_png_reader_decodes_the_reference_pngs_like_pillow(fx, tools):
    """host/png_decode.cpp on the twelve stb-written PNGs of Results/ (8-bit RGB, stb's own deflate and filter choice):
    same size and the checksum of Pillow's decode (stored in the fixture; Pillow is not consulted here)."""
    for name, crc, size in zip(fx["png_names"], fx["png_crc32"], fx["png_size"]):
        out = subprocess.run([tools["png_check"], os.path.join(RESULTS, str(name))], capture_output=True, timeout=120)
        assert out.returncode == 0, name
        head, raw = out.stdout.split(b"\n", 1)
        w, h, c = (int(x) for x in head.split())
        assert (h, w, c) == tuple(int(v) for v in size), name
        assert zlib.crc32(raw) == int(crc), name


@pytest.mark.gpu
def test_gpu_tonemap_brackets_the_reference_png(fx, gpu):
    """k_tonemap (prt_tonemap_srgb8) on the decoded floats: the bytes it makes of the two ends of every RGBE interval
    bracket the reference PNG's byte, >= 70 % exact from the lower end, and equal the host stage's except where the device
    `pow` lands on the other side of a truncation (<= 1 level, < 0.1 % of the bytes)."""
    import torch
    from pooraytracer_amd import api, scenes
    sc = api.Scene(scenes.tiny_scene()).upload(gpu)
    rgbe = fx["crop_rgbe"].reshape(-1, 128, 4)
    ref = fx["crop_png"].reshape(-1, 128, 3)
    lo, hi = rgbe_interval(rgbe)
    got = []
    for x in (lo, hi):
        t = torch.from_numpy(np.ascontiguousarray(x)).cuda()
        u8 = torch.zeros(t.shape, dtype=torch.uint8, device="cuda")
        sc.tonemap_srgb8(t.data_ptr(), x.shape[1], x.shape[0], u8.data_ptr())
        torch.cuda.synchronize()
        got.append(u8.cpu().numpy())
    for g, x in zip(got, (lo, hi)):
        d = np.abs(g.astype(int) - numpy_stage(x).astype(int))
        assert d.max() <= 1 and (d > 0).mean() < 1e-3
    # bracket with that one level of slack on the device pow
    assert (got[0].astype(int) - 1 <= ref).all() and (ref <= got[1].astype(int) + 1).all()
    assert (got[0] == ref).mean() >= 0.70
    sc.close()
