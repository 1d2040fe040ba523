"""Golden vectors for the host mirror's JPEG reader (mvskit_amd/host/jpeg_decode.cpp): small JPEG files of every kind it
supports, written by PIL's encoder, and the pixels PIL's decoder (libjpeg-turbo -- the library CImg::load_jpeg reaches
through libjpeg on current systems, image/image.cpp:837) gives for them.  The files and jpeg_expected.npz are committed;
tests/test_jpeg_decode.py checks the decoder against them without needing PIL.
    python tests/golden/make_jpeg_golden.py"""
import io
import os

import numpy as np
from PIL import Image

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "jpeg")


def picture(w, h, seed):
    rng = np.random.RandomState(seed)
    y, x = np.mgrid[0:h, 0:w]
    a = np.stack([128 + 100 * np.sin(x / 7.0 + y / 13.0), 128 + 90 * np.cos(x / 5.0 - y / 9.0), (x * 3 + y * 5) % 256], -1)
    return np.clip(a + rng.normal(0, 12, (h, w, 3)), 0, 255).astype(np.uint8)


CASES = {  # name -> (width, height, grey, save options)
    "baseline_444": (40, 30, False, dict(quality=90, subsampling=0)),
    "baseline_422_restart": (45, 31, False, dict(quality=75, subsampling=1, restart_marker_blocks=3)),
    "baseline_420_odd": (37, 29, False, dict(quality=80, subsampling=2)),
    "baseline_420_optimized_q30": (64, 48, False, dict(quality=30, subsampling=2, optimize=True)),
    "progressive_420": (50, 34, False, dict(quality=85, subsampling=2, progressive=True)),
    "progressive_444_restart": (33, 17, False, dict(quality=70, subsampling=0, progressive=True, restart_marker_rows=1)),
    "grey_baseline": (41, 23, True, dict(quality=88)),
    "grey_progressive": (24, 40, True, dict(quality=60, progressive=True)),
    "rgb_colourspace": (30, 20, False, dict(quality=92, keep_rgb=True)),
    "tiny_2x2_chroma": (3, 5, False, dict(quality=90, subsampling=2)),
}


def main():
    os.makedirs(HERE, exist_ok=True)
    expected = {}
    for k, (name, (w, h, grey, opts)) in enumerate(sorted(CASES.items())):
        a = picture(w, h, k)
        im = Image.fromarray(a[..., 0] if grey else a)
        bio = io.BytesIO()
        im.save(bio, "JPEG", **opts)
        data = bio.getvalue()
        with open(os.path.join(HERE, name + ".jpg"), "wb") as f:
            f.write(data)
        ref = np.asarray(Image.open(io.BytesIO(data)))
        expected[name] = ref if ref.ndim == 3 else ref[..., None]
    np.savez_compressed(os.path.join(HERE, "jpeg_expected.npz"), **expected)
    print({k: v.shape for k, v in expected.items()})


if __name__ == "__main__":
    main()
