#!/usr/bin/env python3
"""Build-container only: cut fixtures out of the ONE piece of golden data the reference holds for this path — the
`.hdr` + `.png` pairs that `Camera::WriteColorAttachment` (reference Source/Camera.cpp:279-331, `LinearToSRGB`
:206-221) wrote from the same `colorAttachment` into /root/reference/Results/.

They cannot pin the hot path (the scene files that produced them are absent), but they DO pin the output stage
 and exercise the image readers (f3): the .hdr holds the linear floats (as RGBE), the .png what
the reference's NaN scrub + sRGB + clamp + 8-bit truncation made of them.

Writes tests/golden/results_pairs.npz — DATA only (pixel bytes, encoded scanline bytes, sizes, checksums):
  pairs            names of the pairs
  crop_xy          (n_crops, 3): pair index, x0, y0 of each 128x128 crop
  crop_rgbe        (n_crops, 128, 128, 4) u8: RGBE bytes as stored in the .hdr
  crop_png         (n_crops, 128, 128, 3) u8: the PNG's pixels at the same place
  rle_rows         (n_rows, 2): pair index, row — whole scanlines kept with their ENCODED bytes
  rle_row_rgbe     (n_rows, W, 4) u8: the scanline's RGBE pixels
  rle_row_bytes    object-free: concatenated encoded bytes + offsets (stb_image_write's per-component RLE)
  hdr_header       the file header bytes (identical for the four files)
  file_sha256      sha256 of every .hdr file (so a full-file re-encode can be checked where the reference is present)
  png_crc32        zlib.crc32 of Pillow's decode of EVERY Results/*.png (full image), with the sizes
Usage: python tests/golden/make_results_pairs.py   (needs /root/reference; never runs on the GPU box)
"""
import glob
import hashlib
import os
import sys
import zlib

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
RESULTS = "/root/reference/Results"
CROP = 128


def parse_hdr(data):
    """Radiance RGBE with per-component RLE scanlines -> (header bytes, (H, W, 4) u8, [(start, end) of each scanline])."""
    pos = data.index(b"\n\n") + 2
    end = data.index(b"\n", pos)
    dims = data[pos:end].split()
    assert dims[0] == b"-Y" and dims[2] == b"+X", dims
    h, w = int(dims[1]), int(dims[3])
    pos = end + 1
    header = data[:pos]
    px = np.zeros((h, w, 4), dtype=np.uint8)
    spans = []
    for y in range(h):
        start = pos
        assert data[pos] == 2 and data[pos + 1] == 2 and (data[pos + 2] << 8 | data[pos + 3]) == w, (y, data[pos:pos + 4])
        pos += 4
        for c in range(4):
            x = 0
            while x < w:
                n = data[pos]
                if n > 128:  # run
                    n -= 128
                    px[y, x:x + n, c] = data[pos + 1]
                    pos += 2
                else:        # literal dump
                    px[y, x:x + n, c] = np.frombuffer(data, dtype=np.uint8, count=n, offset=pos + 1)
                    pos += 1 + n
                x += n
            assert x == w
        spans.append((start, pos))
    assert pos == len(data), (pos, len(data))
    return header, px, spans


def main():
    if not os.path.isdir(RESULTS):
        sys.exit("needs /root/reference/Results (build container only)")
    hdrs = sorted(glob.glob(os.path.join(RESULTS, "*.hdr")))
    pairs, crop_xy, crop_rgbe, crop_png, rle_rows, rle_row_rgbe, blobs, sha = [], [], [], [], [], [], [], []
    header0 = None
    for k, hp in enumerate(hdrs):
        pp = hp[:-4] + ".png"
        data = open(hp, "rb").read()
        header, px, spans = parse_hdr(data)
        header0 = header0 or header
        assert header == header0
        png = np.asarray(Image.open(pp).convert("RGB"))
        assert png.shape[:2] == px.shape[:2]
        h, w = px.shape[:2]
        pairs.append(os.path.basename(hp)[:-4])
        sha.append(hashlib.sha256(data).hexdigest())
        # crops: the brightest 128x128 block (window / lamp: clamped values, large exponents), the darkest, one in between
        e = px[..., 3].astype(np.int32)
        blocks = [(int(e[y:y + CROP, x:x + CROP].sum()), x, y) for y in range(0, h - CROP + 1, CROP) for x in range(0, w - CROP + 1, CROP)]
        blocks.sort()
        for _, x, y in (blocks[0], blocks[len(blocks) // 2], blocks[-1]):
            crop_xy.append((k, x, y))
            crop_rgbe.append(px[y:y + CROP, x:x + CROP].copy())
            crop_png.append(png[y:y + CROP, x:x + CROP].copy())
        for y in (0, h // 3, h // 2, h - 1):
            rle_rows.append((k, y))
            rle_row_rgbe.append(px[y].copy())
            blobs.append(data[spans[y][0]:spans[y][1]])
    offs = np.cumsum([0] + [len(b) for b in blobs]).astype(np.int64)
    png_names, png_crc, png_size = [], [], []
    for pp in sorted(glob.glob(os.path.join(RESULTS, "*.png"))):
        a = np.asarray(Image.open(pp))
        png_names.append(os.path.basename(pp))
        png_crc.append(zlib.crc32(np.ascontiguousarray(a).tobytes()))
        png_size.append(a.shape if a.ndim == 3 else a.shape + (1,))
    out = os.path.join(HERE, "results_pairs.npz")
    np.savez_compressed(out, pairs=np.array(pairs), crop_xy=np.array(crop_xy, dtype=np.int32),
                        crop_rgbe=np.array(crop_rgbe), crop_png=np.array(crop_png),
                        rle_rows=np.array(rle_rows, dtype=np.int32), rle_row_rgbe=np.array(rle_row_rgbe),
                        rle_row_bytes=np.frombuffer(b"".join(blobs), dtype=np.uint8), rle_row_offsets=offs,
                        hdr_header=np.frombuffer(header0, dtype=np.uint8), file_sha256=np.array(sha),
                        png_names=np.array(png_names), png_crc32=np.array(png_crc, dtype=np.uint32), png_size=np.array(png_size, dtype=np.int32))
    print(out, os.path.getsize(out), "bytes;", len(crop_xy), "crops,", len(rle_rows), "scanlines,", len(png_names), "png checksums")


if __name__ == "__main__":
    main()
