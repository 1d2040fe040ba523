"""DepthNormInit's PLY branch (pmmvps/depth_normal_init.cpp:34-144 over io/io_file.c + RPly) in the host mirror: a data set
directory with CONTOUR2 cameras, JPEG images, PGM masks, a point cloud ply/00000000.ply and one normal map per view
(ASCII, little- and big-endian binary PLY with extra properties and a face element) -> seed patches, compared with a numpy
restatement of the same branch.  Host code only: no GPU."""
import ctypes as C
import math
import os
import shutil
import struct

import numpy as np
import pytest

from mvskit_amd import build, engine

F = np.float32
GOLDEN_JPG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg", "baseline_444.jpg")  # 40 x 30
W, H, NV = 40, 30, 4


@pytest.fixture(scope="module")
def host():
    build.build_engine()
    engine.load_library()
    L = C.CDLL(build.build_host())
    L.mvshost_camera_probe.argtypes = [C.c_char_p, C.c_void_p]
    L.mvshost_seeds_from_plys.argtypes = [C.c_char_p, C.c_longlong, C.c_void_p]
    L.mvshost_seeds_from_plys.restype = C.c_longlong
    L.mvshost_ply_probe.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_longlong, C.POINTER(C.c_int)]
    L.mvshost_ply_probe.restype = C.c_longlong
    return L


def write_ply(path, fmt, xyz, normals=None, extra_colour=False, faces=None):
    n = len(xyz)
    head = ["ply", f"format {fmt} 1.0", "comment written by tests/test_seed_plys.py", f"element vertex {n}",
            "property float x", "property float y", "property double z" if fmt != "ascii" else "property float z"]
    if extra_colour:
        head += ["property uchar red", "property uchar green", "property uchar blue"]
    if normals is not None:
        head += ["property float nx", "property float ny", "property float nz"]
    if faces is not None:
        head += [f"element face {len(faces)}", "property list uchar int vertex_indices"]
    head.append("end_header")
    with open(path, "wb") as f:
        f.write(("\n".join(head) + "\n").encode())
        e = "<" if fmt == "binary_little_endian" else ">"
        for i in range(n):
            if fmt == "ascii":
                vals = [repr(float(v)) for v in xyz[i]]
                if extra_colour:
                    vals += ["10", "20", "30"]
                if normals is not None:
                    vals += [repr(float(v)) for v in normals[i]]
                f.write((" ".join(vals) + "\n").encode())
            else:
                f.write(struct.pack(e + "ffd", *[float(v) for v in xyz[i]]))
                if extra_colour:
                    f.write(bytes([10, 20, 30]))
                if normals is not None:
                    f.write(struct.pack(e + "fff", *[float(v) for v in normals[i]]))
        for face in faces or []:
            if fmt == "ascii":
                f.write((f"{len(face)} " + " ".join(str(v) for v in face) + "\n").encode())
            else:
                f.write(struct.pack(e + "B" + "i" * len(face), len(face), *face))


def euler_camera(a, b, g, t, fx=60.0, fy=60.0, cx=W / 2.0, cy=H / 2.0):
    """CONTOUR2 line of a camera (camera.cpp:117-134, quat2proj 241-261) and its rotation."""
    s1, s2, s3, c1, c2, c3 = (math.sin(math.radians(a)), math.sin(math.radians(b)), math.sin(math.radians(g)),
                              math.cos(math.radians(a)), math.cos(math.radians(b)), math.cos(math.radians(g)))
    R = np.array([[c2 * c3, c3 * s2 * s1 - s3 * c1, c3 * s2 * c1 + s3 * s1], [s3 * c2, s3 * s2 * s1 + c3 * c1, s3 * s2 * c1 - c3 * s1],
                  [-s2, c2 * s1, c2 * c1]])
    text = "CONTOUR2\n%r %r 0 %r %r 0\n%r %r %r\n%r %r %r\n" % tuple(float(v) for v in (fx, fy, cx, cy, a, b, g, t[0], t[1], t[2]))
    return text, R


@pytest.fixture(scope="module")
def dataset(tmp_path_factory, host):
    root = tmp_path_factory.mktemp("plyset")
    for d in ("txt", "image", "mask", "ply"):
        os.makedirs(root / d)
    (root / "option").write_text(f"level 0\ncsize 2\nthreshold 0.7\nwsize 7\nminImageNum 2\nimages -1 0 {NV}\n")
    Rs, Ps = [], []
    for v in range(NV):
        ang = -15.0 + 10.0 * v  # cameras on an arc around the y axis, 4 units from a plane through the origin
        th = math.radians(ang)
        centre = np.array([4.0 * math.sin(th), 0.2 * v, -4.0 * math.cos(th)])
        text, R = euler_camera(3.0 * v - 4.0, ang, 2.0 * v, (0.0, 0.0, 0.0))
        t = -R @ centre
        text, R = euler_camera(3.0 * v - 4.0, ang, 2.0 * v, t)
        (root / "txt" / f"{v:08d}.txt").write_text(text)
        P = np.zeros(12, F)
        assert host.mvshost_camera_probe(str(root / "txt" / f"{v:08d}.txt").encode(), P.ctypes.data) == 0
        Rs.append(R.astype(F))
        Ps.append(P.reshape(3, 4))
        shutil.copy(GOLDEN_JPG, root / "image" / f"{v:04d}0000.jpg")  # no .ppm next to it: PhotoSet::init falls through to the JPEG
        m = np.full((H, W), 255, np.uint8)
        m[:, : 6 + 3 * v] = 0  # a background band per view
        m[10:14, 20:26] = 100  # below the 127 threshold: background
        with open(root / "mask" / f"{v:08d}.pgm", "wb") as f:
            f.write(b"P5\n%d %d\n255\n" % (W, H) + m.tobytes())
    rng = np.random.RandomState(5)
    pts = np.stack([rng.uniform(-1.4, 1.4, 300), rng.uniform(-0.9, 0.9, 300), 0.05 * rng.normal(size=300)], 1)
    pts[:5] += 50.0  # far outside every image
    write_ply(root / "ply" / "00000000.ply", "ascii", pts, faces=[[0, 1, 2], [2, 3, 4, 5]])
    n_world = np.array([0.1, -0.05, -1.0])
    n_world /= np.linalg.norm(n_world)
    maps = []
    for v in range(NV):
        ys, xs = np.mgrid[0:H, 0:W]
        keep = (xs + 2 * ys + v) % 7 != 0  # some pixels carry no normal: zero vectors in the map
        xy = np.stack([xs[keep], ys[keep], np.zeros(keep.sum())], 1).astype(np.float64)
        jitter = rng.normal(0, 0.05, (len(xy), 3))
        nrm = (Rs[v].astype(np.float64).T @ (n_world[None] + jitter).T).T  # camera-frame normals: R * n gives them back in world axes
        fmt = ["ascii", "binary_little_endian", "binary_big_endian", "binary_little_endian"][v]
        write_ply(root / "ply" / f"{v + 1:08d}.ply", fmt, xy, nrm, extra_colour=v != 0, faces=[[0, 1, 2]] if v == 1 else None)
        m = np.zeros((H, W, 3), F)
        # what the reader hands on: float32 of the stored value (float32 in the file), rotated in float32
        m[ys[keep], xs[keep]] = (Rs[v] @ nrm.astype(F).T).T
        maps.append(m)
    masks = []
    for v in range(NV):
        with open(root / "mask" / f"{v:08d}.pgm", "rb") as f:
            raw = f.read()
        masks.append(np.where(np.frombuffer(raw[-W * H:], np.uint8).reshape(H, W) > 127, 255, 0))
    return dict(root=root, P=Ps, R=Rs, pts=pts, maps=maps, masks=masks)


def test_ply_reader(host, dataset):
    root = dataset["root"]
    n = host.mvshost_ply_probe(str(root / "ply" / "00000000.ply").encode(), None, None, 0, None)
    assert n == 300
    p = np.zeros((n, 3))
    has = C.c_int(-1)
    assert host.mvshost_ply_probe(str(root / "ply" / "00000000.ply").encode(), p.ctypes.data, None, n, C.byref(has)) == n
    assert has.value == 0
    np.testing.assert_array_equal(p, dataset["pts"].astype(F).astype(np.float64))  # `property float`: float32 precision, exact round trip of repr()
    ref = None
    for v in range(NV):  # the four encodings carry the same kind of data; every one must give finite unit-ish normals at the right pixels
        path = str(root / "ply" / f"{v + 1:08d}.ply").encode()
        n = host.mvshost_ply_probe(path, None, None, 0, None)
        xy, nr = np.zeros((n, 3)), np.zeros((n, 3))
        assert host.mvshost_ply_probe(path, xy.ctypes.data, nr.ctypes.data, n, C.byref(has)) == n and has.value == 1
        assert xy[:, 0].min() == 0 and xy[:, 0].max() == W - 1 and xy[:, 1].max() == H - 1 and np.all(xy[:, 2] == 0)
        assert np.all(np.abs(np.linalg.norm(nr, axis=1) - 1.0) < 0.3)
        m = np.zeros((H, W, 3), F)
        m[xy[:, 1].astype(int), xy[:, 0].astype(int)] = (dataset["R"][v] @ nr.astype(F).T).T
        np.testing.assert_allclose(m, dataset["maps"][v], rtol=0, atol=1e-6)
    assert host.mvshost_ply_probe(str(root / "ply" / "nothing.ply").encode(), None, None, 0, None) == -1
    (root / "ply" / "short.ply").write_bytes(b"ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\nend_header\n0 0 0\n1 1\n")
    assert host.mvshost_ply_probe(str(root / "ply" / "short.ply").encode(), None, None, 0, None) == -1


def _camera(P):
    """Camera::updateCamera + Optim::setAxesScales (camera.cpp:65-89, optim.cpp:43-65)."""
    M = P[:, :3].astype(np.float64)
    centre = np.append((-np.linalg.inv(M) @ P[:, 3].astype(np.float64)).astype(F), F(1))
    z = (P[2, :3] / np.sqrt(F(np.dot(P[2, :3], P[2, :3])))).astype(F)
    y = np.cross(z, P[0, :3]).astype(F)
    y = (y / np.linalg.norm(y).astype(F)).astype(F)
    x = np.cross(y, z).astype(F)
    return centre, F(np.dot(P[0, :3], x)) + F(np.dot(P[1, :3], y))


def _sort_images(cams, coord, normal, images, level=0):
    """Optim::sortImages(patch, 0), optim.cpp:221-258."""
    thr = F(1.0) - F(math.cos(F(10.0) * F(math.pi) / F(180.0)))
    idx, units, rays = [], [], []
    for im in images:
        centre, ips = cams[im]
        ray = (centre - coord).astype(F)
        fz = np.sqrt(F(np.dot(ray, ray)))
        ray = (ray / fz).astype(F)
        d = F(np.dot(ray, normal))
        if d <= 0:
            continue
        idx.append(im)
        units.append(F(F(2.0 * float(fz) * (1 << level) / float(ips)) / d))
        rays.append(ray)
    out = []
    if len(idx) < 2:
        return out
    while idx:
        k = int(np.argmin(units))
        out.append(idx[k])
        nu = []
        for i in range(len(idx)):
            if i == k:
                continue
            f = min(thr, max(F(thr / F(2)), F(1) - F(np.dot(rays[k], rays[i]))))
            nu.append(F(units[i] * thr / f))
        idx = [v for i, v in enumerate(idx) if i != k]
        rays = [v for i, v in enumerate(rays) if i != k]
        units = nu
    return out


def test_seed_patches_from_plys(host, dataset):
    out = np.zeros(400, dtype=engine.PATCH_DTYPE)
    n = host.mvshost_seeds_from_plys(str(dataset["root"]).encode() + b"/", 400, out.ctypes.data)
    assert n > 100, n
    out = out[:n]
    cams = [_camera(P) for P in dataset["P"]]
    exp = []
    for X in dataset["pts"].astype(F):
        coord = np.append(X, F(1))
        images, n3 = [], np.zeros(3, F)
        for v in range(NV):
            ic = (dataset["P"][v] @ coord).astype(F)
            if ic[2] <= 0:
                continue
            x, y = int(math.floor(F(ic[0] / ic[2]) + F(0.5))), int(math.floor(F(ic[1] / ic[2]) + F(0.5)))
            if not (0 <= x < W and 0 <= y < H) or dataset["masks"][v][y, x] <= 0:
                continue
            n3 = (n3 + dataset["maps"][v][y, x]).astype(F)
            images.append(v)
        if len(images) < 2 or np.linalg.norm(n3) == 0:
            continue
        n3 = (n3 / F(len(images))).astype(F)
        n3 = (n3 / np.sqrt(F(np.dot(n3, n3)))).astype(F)
        normal = np.append(n3, -F(np.dot(X, n3)))
        order = _sort_images(cams, coord, normal, images)
        if order:
            exp.append((coord, normal, order))
    assert len(exp) == n
    same_order = 0
    for rec, (coord, normal, order) in zip(out, exp):
        np.testing.assert_array_equal(rec["coord"], coord)
        np.testing.assert_allclose(rec["normal"], normal, rtol=0, atol=2e-6)
        assert rec["nimages"] == len(order) and sorted(rec["images"][: len(order)]) == sorted(order)
        same_order += list(rec["images"][: len(order)]) == order
    assert same_order >= 0.98 * n  # the greedy order may differ where two penalised units tie to the last bit
    assert len({len(o) for _, _, o in exp}) > 1  # masks and map holes do cut view lists
