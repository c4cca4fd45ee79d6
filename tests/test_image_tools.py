"""Image writers/readers and the image-difference measures (CPU only)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def image(pkg):
    return pkg.image


def _gradient(h=37, w=53):
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    img[..., 3] = 255
    return img


def test_png_round_trip(image, tmp_path):
    img = _gradient()
    image.write_png(tmp_path / "a.png", img)
    back = image.read_png(tmp_path / "a.png")
    assert back.shape == (37, 53, 3)
    assert np.array_equal(back, img[..., :3])
    image.write_png(tmp_path / "b.png", img, keep_alpha=True)
    assert np.array_equal(image.read_png(tmp_path / "b.png"), img)


def test_png_is_well_formed(image, tmp_path):
    """Signature, IHDR fields and chunk CRCs as a strict decoder expects them."""
    import struct
    import zlib
    img = _gradient(4, 5)
    path = tmp_path / "a.png"
    image.write_png(path, img)
    data = path.read_bytes()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, tags = 8, []
    while pos < len(data):
        (n,) = struct.unpack(">I", data[pos:pos + 4])
        tag = data[pos + 4:pos + 8]
        (crc,) = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
        assert crc == zlib.crc32(data[pos + 4:pos + 8 + n]) & 0xFFFFFFFF
        tags.append(tag)
        if tag == b"IHDR":
            assert struct.unpack(">IIBBBBB", data[pos + 8:pos + 8 + n]) == (5, 4, 8, 2, 0, 0, 0)
        pos += 12 + n
    assert tags == [b"IHDR", b"IDAT", b"IEND"]


def test_ppm_round_trip(image, tmp_path):
    img = _gradient()
    image.write_ppm(tmp_path / "a.ppm", img)
    assert np.array_equal(image.read_ppm(tmp_path / "a.ppm"), img[..., :3])
    assert (tmp_path / "a.ppm").read_bytes().startswith(b"P6\n53 37\n255\n")


def test_bad_inputs(image, tmp_path):
    with pytest.raises(ValueError):
        image.write_png(tmp_path / "x.png", np.zeros((4, 4), dtype=np.uint8))
    (tmp_path / "junk.png").write_bytes(b"not a png at all")
    with pytest.raises(ValueError):
        image.read_png(tmp_path / "junk.png")
    with pytest.raises(ValueError):
        image.diff_rgba8(np.zeros((2, 2, 4), np.uint8), np.zeros((3, 2, 4), np.uint8))


def test_difference_measures(image):
    a = np.zeros((8, 8, 4), dtype=np.uint8)
    b = a.copy()
    b[0, 0, 0] = 4
    d = image.diff_rgba8(a, b)
    assert d["max"] == 4 and d["differing"] == pytest.approx(1.0 / 192.0)
    assert d["rmse"] == pytest.approx(np.sqrt(16.0 / 192.0))
    acc = np.full((8, 8, 4), 2.0, dtype=np.float32)
    # the same radiance at different sample counts compares equal
    rmse, rel = image.rmse_linear(acc, 4, acc * 2.0, 8)
    assert rmse == 0.0 and rel == 0.0
    rmse, rel = image.rmse_linear(acc * 1.01, 4, acc, 4)
    assert rmse == pytest.approx(0.005, rel=1e-4)
    assert rel == pytest.approx(0.01, rel=1e-3)


def test_display_conversion_matches_oracle_tail(image, oracle):
    """The float64 viewer conversion stays within one code value of the oracle's exact fp32 tail."""
    direction, focus = oracle.default_aim()
    accum = None
    for frame in range(2):
        params = oracle.make_params((13.0, 2.0, 3.0), direction, 0.05, 20.0, focus, 64, 36, frame=frame, spp=2,
                                    max_depth=4, sphere_count=5)
        res = oracle.render(params, oracle.default_scene(), accum=accum, nthreads=2)
        accum = res["accum"]
    shown = image.display_from_accum(accum, 2 * 2)
    d = image.diff_rgba8(shown, res["output"])
    assert d["max"] <= 1
    assert np.all(shown[..., 3] == 255)


def test_accum_file_and_cli(image, tmp_path):
    acc = np.random.default_rng(3).random((6, 7, 4), dtype=np.float32)
    image.save_accum(tmp_path / "a.npz", acc, 3, 4)
    image.save_accum(tmp_path / "b.npz", acc * 2.0, 6, 4)
    back, frames, spp = image.load_accum(tmp_path / "a.npz")
    assert np.array_equal(back, acc) and (frames, spp) == (3, 4)
    tool = os.path.join(ROOT, "tools", "image_diff.py")
    out = subprocess.run([sys.executable, tool, str(tmp_path / "a.npz"), str(tmp_path / "b.npz"), "--max-rmse", "1e-6"],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert json.loads(out.stdout)["relative_rmse"] < 1e-6
    img = _gradient()
    image.write_png(tmp_path / "a.png", img)
    img[0, 0, 0] ^= 0x80
    image.write_ppm(tmp_path / "b.ppm", img)
    out = subprocess.run([sys.executable, tool, str(tmp_path / "a.png"), str(tmp_path / "b.ppm"), "--max-rmse", "0.01"],
                         capture_output=True, text=True)
    assert out.returncode == 1
    assert json.loads(out.stdout)["max"] == 128
