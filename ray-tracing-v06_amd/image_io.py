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
