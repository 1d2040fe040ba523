"""Image::readJpeg of the host mirror (image/image.cpp:827-879 -> CImg::load_jpeg -> libjpeg): the decoder written for it
(mvskit_amd/host/jpeg_decode.cpp) against libjpeg-turbo's pixels -- committed golden files (tests/golden/jpeg, made by
tests/golden/make_jpeg_golden.py with PIL) and, where PIL is importable, freshly encoded files of many shapes.  Equality is
byte for byte: the decoder follows libjpeg's integer arithmetic (islow IDCT, triangle chroma up-sampling, fixed-point
YCbCr -> RGB)."""
import ctypes as C
import io
import os

import numpy as np
import pytest

from mvskit_amd import build, engine

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg")


@pytest.fixture(scope="module")
def host():
    build.build_engine()
    engine.load_library()
    L = C.CDLL(build.build_host())
    L.mvshost_jpeg_decode.argtypes = [C.c_char_p, C.c_longlong, C.c_void_p, C.c_longlong, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_int]
    L.mvshost_jpeg_probe.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_longlong]
    return L


def decode(L, data):
    w, h, c = C.c_int(), C.c_int(), C.c_int()
    err = C.create_string_buffer(256)
    if L.mvshost_jpeg_decode(data, len(data), None, 0, C.byref(w), C.byref(h), C.byref(c), err, 256) != 0:
        raise ValueError(err.value.decode())
    out = np.empty((h.value, w.value, c.value), np.uint8)
    assert L.mvshost_jpeg_decode(data, len(data), out.ctypes.data, out.size, C.byref(w), C.byref(h), C.byref(c), err, 256) == 0
    return out


def test_golden_files(host):
    expected = np.load(os.path.join(GOLDEN, "jpeg_expected.npz"))
    assert len(expected.files) >= 10
    for name in expected.files:
        with open(os.path.join(GOLDEN, name + ".jpg"), "rb") as f:
            data = f.read()
        got = decode(host, data)
        assert got.shape == expected[name].shape, name
        np.testing.assert_array_equal(got, expected[name], err_msg=name)
    # the files cover what their names say
    def markers(name):
        with open(os.path.join(GOLDEN, name + ".jpg"), "rb") as f:
            return f.read()
    assert b"\xff\xc2" in markers("progressive_420") and b"\xff\xc0" in markers("baseline_444")
    assert b"\xff\xdd" in markers("baseline_422_restart") and b"\xff\xdd" in markers("progressive_444_restart")


def test_read_jpeg_of_the_mirror(host, tmp_path):
    """Photo::readJpeg: interleaved RGB; a grey file becomes R = G = B (image.cpp:850-858)."""
    expected = np.load(os.path.join(GOLDEN, "jpeg_expected.npz"))
    for name in ("baseline_420_odd", "grey_baseline"):
        w, h = C.c_int(), C.c_int()
        path = os.path.join(GOLDEN, name + ".jpg").encode()
        assert host.mvshost_jpeg_probe(path, C.byref(w), C.byref(h), None, 0) == 0
        out = np.empty((h.value, w.value, 3), np.uint8)
        assert host.mvshost_jpeg_probe(path, C.byref(w), C.byref(h), out.ctypes.data, out.size) == 0
        ref = expected[name]
        np.testing.assert_array_equal(out, ref if ref.shape[2] == 3 else np.repeat(ref, 3, axis=2))
    w, h = C.c_int(), C.c_int()
    assert host.mvshost_jpeg_probe(str(tmp_path / "missing.jpg").encode(), C.byref(w), C.byref(h), None, 0) == -1


def test_errors_are_reported_not_guessed(host):
    with open(os.path.join(GOLDEN, "baseline_444.jpg"), "rb") as f:
        data = f.read()
    with pytest.raises(ValueError, match="SOI"):
        decode(host, b"P6\n4 4\n255\n" + bytes(48))
    with pytest.raises(ValueError, match="past the end"):
        decode(host, data[:40])
    sof = data.index(b"\xff\xc0")
    with pytest.raises(ValueError, match="8-bit"):
        decode(host, data[:sof + 4] + b"\x0c" + data[sof + 5:])       # 12-bit precision
    with pytest.raises(ValueError, match="arithmetic"):
        decode(host, data[:sof + 1] + b"\xc9" + data[sof + 2:])       # SOF9
    with pytest.raises(ValueError, match="component count"):
        decode(host, data[:sof + 9] + b"\x04" + data[sof + 10:])      # four components
    # a file cut inside the entropy-coded data still decodes (libjpeg warns and pads with zeros): same size, top rows intact
    full, cut = decode(host, data), decode(host, data[: len(data) * 2 // 3])
    assert cut.shape == full.shape and np.array_equal(cut[:8], full[:8]) and not np.array_equal(cut, full)


def test_against_pil_on_many_shapes(host):
    PIL = pytest.importorskip("PIL")
    from PIL import Image, features

    if not features.check_feature("libjpeg_turbo"):
        pytest.skip("PIL is not built on libjpeg-turbo: its pixels are not the ones the golden files pin")
    rng = np.random.RandomState(0)
    cases = 0
    for (w, h) in [(64, 48), (37, 29), (200, 131), (17, 9), (8, 8), (1, 1), (3, 5), (5, 2), (33, 16), (16, 33)]:
        y, x = np.mgrid[0:h, 0:w]
        a = np.stack([128 + 100 * np.sin(x / 7.0 + y / 13.0), 128 + 90 * np.cos(x / 5.0 - y / 9.0), (x * 3 + y * 5) % 256], -1)
        a = np.clip(a + rng.normal(0, 12, (h, w, 3)), 0, 255).astype(np.uint8)
        for q in (95, 60, 20):
            for prog in (False, True):
                for rst in (0, 2):
                    variants = [("L", {})] + [("RGB", dict(subsampling=ss)) for ss in (0, 1, 2)]
                    for mode, opts in variants:
                        opts = dict(opts, quality=q, progressive=prog)
                        if rst:
                            opts["restart_marker_blocks"] = rst
                        bio = io.BytesIO()
                        Image.fromarray(a[..., 0] if mode == "L" else a).save(bio, "JPEG", **opts)
                        data = bio.getvalue()
                        ref = np.asarray(Image.open(io.BytesIO(data)))
                        ref = ref if ref.ndim == 3 else ref[..., None]
                        got = decode(host, data)
                        assert got.shape == ref.shape
                        np.testing.assert_array_equal(got, ref, err_msg=f"{w}x{h} q{q} {mode} {opts}")
                        cases += 1
    assert cases == 480
