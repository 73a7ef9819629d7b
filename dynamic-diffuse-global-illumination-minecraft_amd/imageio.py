"""Raster export and debug views: dependency-free PNG writer, the probe
texture in the reference's raster layout, and the cage-index debug colouring of the reference's
README (README.md:89-91: colour = index of the cage's base probe)."""
import struct
import zlib

import numpy as np


def write_png(path, image):
    """image: uint8 [H, W, 3|4]."""
    img = np.ascontiguousarray(image, dtype=np.uint8)
    h, w, c = img.shape
    if c not in (3, 4):
        raise ValueError("expected RGB or RGBA")
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))

    def chunk(tag, data):
        body = tag + data
        return struct.pack(">I", len(data)) + body + struct.pack(">I", zlib.crc32(body) & 0xFFFFFFFF)

    with open(path, "wb") as fh:
        fh.write(b"\x89PNG\r\n\x1a\n")
        fh.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6 if c == 4 else 2, 0, 0, 0)))
        fh.write(chunk(b"IDAT", zlib.compress(raw, 6)))
        fh.write(chunk(b"IEND", b""))


def dump_probe_textures(engine, prefix):
    """REF mode: <prefix>_albedo.png and <prefix>_distance.png in the reference's W x H layout
    (the view compute_pass.comp:185-190 can blit for debugging)."""
    albedo, distance = engine.read_textures()
    write_png(prefix + "_albedo.png", albedo)
    write_png(prefix + "_distance.png", np.concatenate([distance[..., :3], np.full_like(distance[..., :1], 255)], axis=-1))
    return albedo.shape


def cage_debug_image(cage_idx8, height, width):
    """Colour each shading point by its cage's base probe index (corner 0); magenta outside the field.
    (Host-side twin of ddgi_render's render_mode 7, for cage indices that come from ddgi_sample.)"""
    base = np.asarray(cage_idx8, dtype=np.int64).reshape(height, width, 8)[..., 0]
    h = (base * 2654435761) & 0xFFFFFF
    img = np.stack([(h >> 16) & 255, (h >> 8) & 255, h & 255, np.full_like(h, 255)], axis=-1).astype(np.uint8)
    img[base < 0] = (255, 0, 255, 255)
    return img
