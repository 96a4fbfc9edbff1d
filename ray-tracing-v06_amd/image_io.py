"""Image output on the caller's side of the path — write_renderbuffer of the reference app
(main/src/FirstApp.cpp:108-122): 8-bit conversion `uint8(v * 255.999f)` of R, G, B (alpha dropped), rows flipped
(row 0 of the framebuffer is the BOTTOM of the picture, Renderer.cu:192), then an image file.  The reference
writes JPEG q95 through stb_image_write; here PNG (lossless, zlib from the standard library) and binary PPM.
A NaN channel (NaN pixels) is written as 0: the reference's static_cast of NaN is undefined
behaviour and yields 0 on x86.
"""
import struct
import zlib

import numpy as np


def to_rgb8(framebuffer):
    """float RGBA [H][W][4], row 0 = bottom  ->  uint8 RGB [H][W][3], row 0 = top."""
    fb = np.asarray(framebuffer, dtype=np.float32)
    if fb.ndim != 3 or fb.shape[2] != 4:
        raise ValueError("framebuffer must be [H][W][4] float32")
    rgb = fb[::-1, :, :3] * np.float32(255.999)
    rgb = np.where(np.isnan(rgb), np.float32(0.0), rgb)
    return np.clip(rgb, 0.0, 255.0).astype(np.uint8)  # truncation toward zero, like static_cast<uint8_t>


def write_ppm(path, framebuffer):
    img = to_rgb8(framebuffer)
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(img.tobytes())


def write_png(path, framebuffer):
    img = to_rgb8(framebuffer)
    h, w, _ = img.shape
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))

    def chunk(tag, data):
        body = tag + data
        return struct.pack(">I", len(data)) + body + struct.pack(">I", zlib.crc32(body) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


# ---------------------------------------------------------------------------------------------------------------------
# JPEG — the format the reference app writes (`stbi_write_jpg(path, w, h, 3, data, 95)`, FirstApp.cpp:120): a baseline
# (sequential, Huffman, 8-bit) JFIF encoder with the conventions of that call: the standard Annex-K quantisation tables
# scaled for the quality (q95 -> factor 10 %), no chroma subsampling above quality 90, the standard Annex-K Huffman
# tables, rows already flipped by to_rgb8.  A lossy format: pixel values match the reference writer's only up to the DCT's
# rounding, so parity tests use PNG / PPM; this exists so that a caller gets the same kind of file the reference produces.
# ---------------------------------------------------------------------------------------------------------------------
_ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63])
_Q_LUM = np.array([16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
                   18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99])
_Q_CHR = np.array([17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99]
                  + [99] * 32)
_DC_LUM_BITS = [0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0]
_DC_CHR_BITS = [0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0]
_DC_VALS = list(range(12))
_AC_LUM_BITS = [0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d]
_AC_LUM_VALS = [0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08,
                0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28,
                0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
                0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89,
                0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6,
                0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
                0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa]
_AC_CHR_BITS = [0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77]
_AC_CHR_VALS = [0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91,
                0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26,
                0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58,
                0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87,
                0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4,
                0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
                0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa]


def _huffman_codes(bits, vals):
    """canonical code of a (BITS, HUFFVAL) table (ITU T.81 Annex C): symbol -> (code, length)"""
    table, code, k = {}, 0, 0
    for length in range(1, 17):
        for _ in range(bits[length - 1]):
            table[vals[k]] = (code, length)
            code += 1
            k += 1
        code <<= 1
    return table


def _dct_blocks(plane):
    """8x8 forward DCT-II (orthonormal, the JPEG definition) of every block of a [H][W] float plane whose sides are multiples of 8"""
    k = np.arange(8)
    c = np.sqrt(2.0 / 8.0) * np.cos((2 * k[None, :] + 1) * k[:, None] * np.pi / 16.0)
    c[0, :] = np.sqrt(1.0 / 8.0)
    h, w = plane.shape
    b = plane.reshape(h // 8, 8, w // 8, 8).transpose(0, 2, 1, 3)           # [by][bx][8][8]
    return np.einsum("ij,yxjk,lk->yxil", c, b, c)


def write_jpg(path, framebuffer, quality=95):
    """Baseline JPEG of the framebuffer with stb_image_write's conventions for `stbi_write_jpg(..., quality)` (FirstApp.cpp:120)."""
    rgb = to_rgb8(framebuffer).astype(np.float64)
    h, w, _ = rgb.shape
    quality = min(100, max(1, int(quality)))
    scale = 5000 // quality if quality < 50 else 200 - quality * 2
    qt = [np.clip((t * scale + 50) // 100, 1, 255).astype(np.int64) for t in (_Q_LUM, _Q_CHR)]
    # JFIF colour transform, level shift, edge replication up to a multiple of 8 (no subsampling: stb subsamples only for quality <= 90)
    r, g, b = rgb[..., 0], rgb[..., 1], rgb[..., 2]
    planes = [0.29900 * r + 0.58700 * g + 0.11400 * b - 128.0, -0.16874 * r - 0.33126 * g + 0.50000 * b, 0.50000 * r - 0.41869 * g - 0.08131 * b]
    ph, pw = (h + 7) // 8 * 8, (w + 7) // 8 * 8
    coefs = []
    for ci, pl in enumerate(planes):
        pl = np.pad(pl, ((0, ph - h), (0, pw - w)), mode="edge")
        d = _dct_blocks(pl).reshape(ph // 8, pw // 8, 64) / qt[0 if ci == 0 else 1][None, None, :]
        q = np.where(d < 0, np.ceil(d - 0.5), np.floor(d + 0.5)).astype(np.int64)     # round half away from zero, as stb does
        coefs.append(q[..., _ZIGZAG])
    dc_tab = [_huffman_codes(_DC_LUM_BITS, _DC_VALS), _huffman_codes(_DC_CHR_BITS, _DC_VALS)]
    ac_tab = [_huffman_codes(_AC_LUM_BITS, _AC_LUM_VALS), _huffman_codes(_AC_CHR_BITS, _AC_CHR_VALS)]

    acc, nbits, out = 0, 0, bytearray()

    def put(code, length):
        nonlocal acc, nbits
        acc = (acc << length) | code
        nbits += length
        while nbits >= 8:
            byte = (acc >> (nbits - 8)) & 0xFF
            out.append(byte)
            if byte == 0xFF:
                out.append(0)          # byte stuffing
            nbits -= 8
        acc &= (1 << nbits) - 1

    def magnitude(v):                  # (category, extra bits) of a coefficient (T.81 F.1.2.1)
        a = -v if v < 0 else v
        cat = a.bit_length()
        return cat, (v if v >= 0 else v + (1 << cat) - 1)

    prev_dc = [0, 0, 0]
    for by in range(ph // 8):
        for bx in range(pw // 8):
            for ci in range(3):
                blk = coefs[ci][by, bx]
                t = 0 if ci == 0 else 1
                diff = int(blk[0]) - prev_dc[ci]
                prev_dc[ci] = int(blk[0])
                cat, extra = magnitude(diff)
                put(*dc_tab[t][cat])
                if cat:
                    put(extra, cat)
                nz = np.nonzero(blk[1:])[0]
                pos = 0
                for k in nz:
                    run = int(k) - pos
                    while run >= 16:
                        put(*ac_tab[t][0xF0])      # ZRL: sixteen zeros
                        run -= 16
                    cat, extra = magnitude(int(blk[1 + k]))
                    put(*ac_tab[t][(run << 4) | cat])
                    put(extra, cat)
                    pos = int(k) + 1
                if pos < 63:
                    put(*ac_tab[t][0x00])          # EOB
    if nbits:
        put((1 << (8 - nbits)) - 1, 8 - nbits)     # pad the last byte with ones

    def segment(marker, payload):
        return struct.pack(">BBH", 0xFF, marker, len(payload) + 2) + payload

    def dht(cls_id, bits, vals):
        return bytes([cls_id]) + bytes(bits) + bytes(vals)

    head = b"\xff\xd8" + segment(0xE0, b"JFIF\x00\x01\x01\x00\x00\x01\x00\x01\x00\x00")
    head += segment(0xDB, b"\x00" + bytes(qt[0][_ZIGZAG].astype(np.uint8)) + b"\x01" + bytes(qt[1][_ZIGZAG].astype(np.uint8)))
    head += segment(0xC0, struct.pack(">BHHB", 8, h, w, 3) + bytes([1, 0x11, 0, 2, 0x11, 1, 3, 0x11, 1]))
    head += segment(0xC4, dht(0x00, _DC_LUM_BITS, _DC_VALS) + dht(0x10, _AC_LUM_BITS, _AC_LUM_VALS)
                    + dht(0x01, _DC_CHR_BITS, _DC_VALS) + dht(0x11, _AC_CHR_BITS, _AC_CHR_VALS))
    head += segment(0xDA, bytes([3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 63, 0]))
    with open(path, "wb") as f:
        f.write(head + bytes(out) + b"\xff\xd9")
