"""Image files and image differences for the tracer's two outputs.

The accumulator is float32 (H, W, 4) in linear radiance (sum over frames, alpha 1); the display image is
uint8 (H, W, 4) as the shader's epilogue stores it (shaders/raytrace.comp.glsl:337-342: normalise by the
sample count, c/(c+1), gamma 1/2.2, UNORM8).  Writers need nothing beyond numpy and zlib.
"""
import struct
import zlib

import numpy as np


def write_ppm(path, rgba8):
    """Binary PPM (P6) of the display image; alpha is dropped."""
    img = np.ascontiguousarray(rgba8, dtype=np.uint8)
    if img.ndim != 3 or img.shape[2] not in (3, 4):
        raise ValueError("expected (H, W, 3|4) uint8")
    h, w = img.shape[:2]
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (w, h))
        f.write(np.ascontiguousarray(img[:, :, :3]).tobytes())


def read_ppm(path):
    """Reads a binary PPM written by write_ppm -> (H, W, 3) uint8."""
    with open(path, "rb") as f:
        data = f.read()
    fields, pos = [], 0
    while len(fields) < 4:
        while data[pos:pos + 1].isspace():
            pos += 1
        if data[pos:pos + 1] == b"#":
            pos = data.index(b"\n", pos) + 1
            continue
        end = pos
        while not data[end:end + 1].isspace():
            end += 1
        fields.append(data[pos:end])
        pos = end
    if fields[0] != b"P6" or int(fields[3]) != 255:
        raise ValueError("%s is not an 8-bit binary PPM" % (path,))
    w, h = int(fields[1]), int(fields[2])
    pixels = np.frombuffer(data, dtype=np.uint8, count=w * h * 3, offset=pos + 1)
    return pixels.reshape(h, w, 3).copy()


def _png_chunk(tag, payload):
    body = tag + payload
    return struct.pack(">I", len(payload)) + body + struct.pack(">I", zlib.crc32(body) & 0xFFFFFFFF)


def write_png(path, rgba8, keep_alpha=False):
    """8-bit PNG (RGB, or RGBA with keep_alpha) of the display image, filter type 0 on every row."""
    img = np.ascontiguousarray(rgba8, dtype=np.uint8)
    if img.ndim != 3 or img.shape[2] not in (3, 4):
        raise ValueError("expected (H, W, 3|4) uint8")
    if not (keep_alpha and img.shape[2] == 4):
        img = np.ascontiguousarray(img[:, :, :3])
    h, w, c = img.shape
    rows = np.zeros((h, 1 + w * c), dtype=np.uint8)
    rows[:, 1:] = img.reshape(h, w * c)
    header = struct.pack(">IIBBBBB", w, h, 8, 6 if c == 4 else 2, 0, 0, 0)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(_png_chunk(b"IHDR", header))
        f.write(_png_chunk(b"IDAT", zlib.compress(rows.tobytes(), 6)))
        f.write(_png_chunk(b"IEND", b""))


def read_png(path):
    """Reads a PNG written by write_png (8-bit RGB/RGBA, filter 0, no interlace) -> (H, W, C) uint8."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("%s is not a PNG" % (path,))
    pos, idat, header = 8, [], None
    while pos < len(data):
        (n,) = struct.unpack(">I", data[pos:pos + 4])
        tag = data[pos + 4:pos + 8]
        payload = data[pos + 8:pos + 8 + n]
        (crc,) = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
        if zlib.crc32(tag + payload) & 0xFFFFFFFF != crc:
            raise ValueError("%s: bad CRC in %r chunk" % (path, tag))
        if tag == b"IHDR":
            header = struct.unpack(">IIBBBBB", payload)
        elif tag == b"IDAT":
            idat.append(payload)
        pos += 12 + n
    w, h, depth, colour, _, _, interlace = header
    if depth != 8 or colour not in (2, 6) or interlace != 0:
        raise ValueError("%s: only 8-bit RGB/RGBA non-interlaced PNGs are read" % (path,))
    c = 4 if colour == 6 else 3
    rows = np.frombuffer(zlib.decompress(b"".join(idat)), dtype=np.uint8).reshape(h, 1 + w * c)
    if np.any(rows[:, 0] != 0):
        raise ValueError("%s: filtered rows are not supported" % (path,))
    return rows[:, 1:].reshape(h, w, c).copy()


def save_accum(path, accum, frames, spp):
    """Stores the linear accumulator with its sample count (npz: accum float32 (H,W,4), frames, spp)."""
    np.savez_compressed(path, accum=np.asarray(accum, dtype=np.float32), frames=np.uint32(frames), spp=np.uint32(spp))


def load_accum(path):
    with np.load(path, allow_pickle=False) as doc:
        return doc["accum"].astype(np.float32), int(doc["frames"]), int(doc["spp"])


def display_from_accum(accum, sample_count):
    """The shader's epilogue (:337-342) in float64 numpy: a viewer-grade conversion for accumulators that
    never went through the device tail (e.g. an oracle render); within one code value of the exact fp32 path."""
    c = np.asarray(accum, dtype=np.float64)[..., :3] / float(sample_count)
    c = c / (c + 1.0)
    c = np.power(np.maximum(c, 0.0), 1.0 / 2.2)
    out = np.empty(c.shape[:2] + (4,), dtype=np.uint8)
    out[..., :3] = np.floor(np.clip(c, 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8)
    out[..., 3] = 255
    return out


def rmse_linear(accum_a, count_a, accum_b, count_b):
    """Root-mean-square difference of two accumulators after normalising each by its own sample count,
    over RGB, and the same number relative to the mean luminance of the second image
    (the acceptance bound for converged renders is relative RMSE <= 1e-3)."""
    a = np.asarray(accum_a, dtype=np.float64)[..., :3] / float(count_a)
    b = np.asarray(accum_b, dtype=np.float64)[..., :3] / float(count_b)
    if a.shape != b.shape:
        raise ValueError("shape mismatch %s vs %s" % (a.shape, b.shape))
    rmse = float(np.sqrt(np.mean((a - b) ** 2)))
    lum = float(np.mean(b @ np.array([0.2126, 0.7152, 0.0722])))
    return rmse, (rmse / lum if lum > 0.0 else float("inf"))


def diff_rgba8(img_a, img_b):
    """Differences of two display images over RGB: RMSE in code values, largest difference, and the
    fraction of channel values that differ at all."""
    a = np.asarray(img_a)[..., :3].astype(np.int32)
    b = np.asarray(img_b)[..., :3].astype(np.int32)
    if a.shape != b.shape:
        raise ValueError("shape mismatch %s vs %s" % (a.shape, b.shape))
    d = np.abs(a - b)
    return {"rmse": float(np.sqrt(np.mean(d.astype(np.float64) ** 2))), "max": int(d.max()) if d.size else 0,
            "differing": float(np.mean(d != 0)) if d.size else 0.0}
