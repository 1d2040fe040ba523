"""Host-side mirror of the reference classes (mvskit_amd/host): option / camera / patch text formats on CPU,
and the whole Option -> PmMvps::init -> run sequence against the oracle on the GPU."""
import ctypes as C
import math
import os

import numpy as np
import pytest

from mvskit_amd import build, engine, synth


@pytest.fixture(scope="module")
def host():
    build.build_engine()
    engine.load_library()  # loads torch's HIP runtime first, then libmvskit_engine.so
    L = C.CDLL(build.build_host())
    L.mvshost_option_probe.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_void_p]
    L.mvshost_patch_roundtrip.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_void_p]
    L.mvshost_camera_probe.argtypes = [C.c_char_p, C.c_void_p]
    L.mvshost_run_dataset.argtypes = [C.c_char_p, C.c_int, C.c_uint, C.c_longlong, C.c_void_p, C.c_void_p]
    L.mvshost_set_ply_output.argtypes = [C.c_char_p]
    L.mvshost_set_ply_output.restype = None
    L.mvshost_pbm_probe.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p]
    L.mvshost_set_ranks.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_int]
    L.mvshost_set_ranks.restype = None
    L.mvshost_set_filter.argtypes = [C.c_int]
    L.mvshost_set_filter.restype = None
    L.mvshost_optim_chain.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_uint,
                                      C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.mvshost_run.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_uint,
                              C.c_int, C.c_longlong, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p]
    return L


def test_option_file(host, tmp_path):
    # keys and defaults of pmmvps/option.cpp:19-33,53-116
    (tmp_path / "option").write_text("# comment line\nlevel 0\ncsize 2\nthreshold 0.65\nwsize 7\nminImageNum 2\nCPU 8\nmaxAngle 12\nquad 2.0\nimages -1 0 5\n")
    oi, of = np.zeros(7, np.int32), np.zeros(3, np.float32)
    assert host.mvshost_option_probe(str(tmp_path).encode() + b"/", b"option", oi.ctypes.data, of.ctypes.data) == 0
    assert list(oi) == [5, 0, 2, 7, 2, -1, 5]
    np.testing.assert_allclose(of, [0.65, 12 * math.pi / 180, 2.0], rtol=1e-6)
    (tmp_path / "bad").write_text("level 1\nnosuchkey 3\nimages 2 0 1\n")
    assert host.mvshost_option_probe(str(tmp_path).encode() + b"/", b"bad", oi.ctypes.data, of.ctypes.data) == -1  # option.cpp:117-120
    (tmp_path / "noimages").write_text("level 1\n")
    assert host.mvshost_option_probe(str(tmp_path).encode() + b"/", b"noimages", oi.ctypes.data, of.ctypes.data) == -1  # option.cpp:125-128
    (tmp_path / "defaults").write_text("images 3 4 7 9\n")
    assert host.mvshost_option_probe(str(tmp_path).encode() + b"/", b"defaults", oi.ctypes.data, of.ctypes.data) == 0
    assert list(oi) == [3, 1, 2, 7, 3, 3, 3]
    np.testing.assert_allclose(of, [0.7, 10 * math.pi / 180, 2.5], rtol=1e-6)


def test_patch_text_roundtrip(host):
    # pmmvps/patch.cpp:31-79
    text = b"PATCHES\n0.5 -1.25 2 1\n0 0.6 0.8 0\n0.91 0.015 0.3\n3\n2 0 5 \n2\n1 4 \n"
    out = C.create_string_buffer(1024)
    rec = np.zeros(1, dtype=engine.PATCH_DTYPE)
    n = host.mvshost_patch_roundtrip(text, out, 1024, rec.ctypes.data)
    assert n > 0
    toks = out.value.decode().split()
    assert toks == text.decode().split()
    r = rec[0]
    np.testing.assert_allclose(r["coord"], [0.5, -1.25, 2, 1])
    np.testing.assert_allclose(r["normal"], [0, 0.6, 0.8, 0])
    assert (r["nimages"], r["nvimages"]) == (3, 2) and list(r["images"][:3]) == [2, 0, 5] and list(r["vimages"][:2]) == [1, 4]
    # PATCHA carries a type and a direction that are skipped (patch.cpp:37-41)
    texta = b"PATCHA\n1 2 3 1\n0 0 1 0\n0.5 0.1 0.2\n7 0 0 1 0\n1\n3 \n0\n\n"
    n = host.mvshost_patch_roundtrip(texta, out, 1024, rec.ctypes.data)
    assert n > 0 and rec[0]["nimages"] == 1 and rec[0]["images"][0] == 3


def test_camera_text(host, tmp_path):
    P = np.arange(12, dtype=np.float32) * 0.5 + 1
    (tmp_path / "c0.txt").write_text("CONTOUR\n" + " ".join(str(float(v)) for v in P) + "\n")
    got = np.zeros(12, np.float32)
    assert host.mvshost_camera_probe(str(tmp_path / "c0.txt").encode(), got.ctypes.data) == 0
    np.testing.assert_array_equal(got, P)  # camera.cpp:110-116
    # CONTOUR2 with the intrinsics of the sample camera in test/test.cpp:96-100 and Euler angles (degrees) + t
    fx = fy = 765.702941895
    skew, cx, cy = 0.0200504438012, 320.0, 240.0
    a, b, g = 10.0, -20.0, 30.0
    t = [-0.959477365017, 0.0510520711541, 0.251982748508]
    (tmp_path / "c2.txt").write_text(f"CONTOUR2\n{fx} {fy} {skew} {cx} {cy} 0\n{a} {b} {g} {t[0]} {t[1]} {t[2]}\n")
    assert host.mvshost_camera_probe(str(tmp_path / "c2.txt").encode(), got.ctypes.data) == 0
    ar, br, gr = map(math.radians, (a, b, g))
    s1, s2, s3, c1, c2, c3 = math.sin(ar), math.sin(br), math.sin(gr), math.cos(ar), math.cos(br), math.cos(gr)
    R = np.array([[c2 * c3, c3 * s2 * s1 - s3 * c1, c3 * s2 * c1 + s3 * s1], [s3 * c2, s3 * s2 * s1 + c3 * c1, s3 * s2 * c1 - c3 * s1],
                  [-s2, c2 * s1, c2 * c1]])  # Camera::quat2proj, camera.cpp:241-261
    K = np.array([[fx, skew, cx], [0, fy, cy], [0, 0, 1]])
    exp = K @ np.concatenate([R, np.array(t)[:, None]], 1)
    np.testing.assert_allclose(got.reshape(3, 4), exp, rtol=2e-5, atol=1e-4)
    (tmp_path / "bad.txt").write_text("CONTOUR9\n1 2 3\n")
    assert host.mvshost_camera_probe(str(tmp_path / "bad.txt").encode(), got.ctypes.data) == -1  # camera.cpp:45-48


def test_pbm_mask(host, tmp_path):
    """Image::readPBMImage (image.cpp:881-946): binary P4, set bit = background (0), clear bit = 255, the bits taken as one
    continuous stream (the reference does not skip the row padding of the PBM format)."""
    w, h = 13, 5  # not a multiple of 8: the continuous-stream reading shows
    rng = np.random.RandomState(4)
    bits = rng.randint(0, 2, size=w * h).astype(np.uint8)
    packed = np.packbits(bits)  # most significant bit first, as image.cpp:929-941 walks them
    f = tmp_path / "00000000.pbm"
    f.write_bytes(b"P4\n# made by the test\n%d %d\n" % (w, h) + packed.tobytes())
    out = np.zeros(w * h, np.uint8)
    assert host.mvshost_pbm_probe(str(f).encode(), w, h, out.ctypes.data) == 0
    np.testing.assert_array_equal(out, np.where(bits == 1, 0, 255).astype(np.uint8))
    assert host.mvshost_pbm_probe(str(f).encode(), w + 1, h, out.ctypes.data) == -1  # size must match the image
    g = tmp_path / "ascii.pbm"
    g.write_bytes(b"P1\n2 2\n0 1 1 0\n")
    assert host.mvshost_pbm_probe(str(g).encode(), 2, 2, out.ctypes.data) == -1  # "Only accept binary pbm format"


@pytest.mark.gpu
def test_optim_class_of_the_mirror(host, small_multi_scene):
    """The mirror's Optim (optim.hpp:24-112: preProcess / refinePatch / postProcess / computeNcc on one Patch) against the
    same single functions driven through the ctypes binding, patch by patch: identical records and flags."""
    sc = small_multi_scene
    seeds = synth.make_seeds(sc, stride=12, seed=8)[:48]
    n = seeds.shape[0]
    P = np.ascontiguousarray(sc.P, dtype=np.float32)
    img = np.ascontiguousarray(sc.images)
    sd = np.ascontiguousarray(seeds)
    out = np.zeros(n, dtype=engine.PATCH_DTYPE)
    flags = np.zeros(n, np.int32)
    ncc = np.zeros(n, np.float32)
    assert host.mvshost_optim_chain(sc.nviews, sc.W, sc.H, P.ctypes.data, img.ctypes.data, 0, 2, 7, 3, C.c_float(0.7), 9, n, sd.ctypes.data,
                                    out.ctypes.data, flags.ctypes.data, ncc.ctypes.data) == 0
    e = engine.Engine(sc.nviews, level=0, csize=2, wsize=7, minImageNum=3, nccThreshold=0.7, seed=9, depth=0)
    e.set_scene(sc)
    passed = 0
    for i in range(n):
        one = np.ascontiguousarray(seeds[i:i + 1])
        one["nvimages"] = 0
        _, v, _ = e.probe(engine.PROBE_NCC, one)
        assert v[0] == ncc[i]
        rec, _, f = e.probe(engine.PROBE_PREPROCESS, one)
        if f[0] == 0:
            rec, _, _ = e.probe(engine.PROBE_REFINE, rec)
            rec, _, f = e.probe(engine.PROBE_POSTPROCESS, rec)
        assert f[0] == flags[i], i
        for k in ("coord", "normal", "ncc", "dscale", "nimages", "images"):
            np.testing.assert_array_equal(rec[k][0], out[k][i], err_msg=f"{k} of patch {i}")
        passed += int(f[0] == 0)
    assert passed > 10
    e.close()


@pytest.mark.gpu
def test_pmmvps_run_with_ranks_world1(host, small_multi_scene, tmp_path):
    """The C++ host mirror with PmMvps::setRanks: communicator id through a file, engine created as shard 0 of 1,
    mvs_engine_comm_init, and Propagate::run exchanging inside the engine over RCCL -- the code path every rank of an
    N-GPU job runs -- must return the patches of the plain single-GPU run.  (RCCL refuses two ranks on one device, so one
    rank is what a one-GPU box can execute; the N-rank merge logic is covered by tests/test_dist_gloo.py and
    tests/test_gpu_dist.py.)"""
    sc = small_multi_scene
    seeds = synth.make_seeds(sc, stride=4, seed=21)
    iters = 3
    cap = 400000
    P = np.ascontiguousarray(sc.P, dtype=np.float32)
    img = np.ascontiguousarray(sc.images)
    sd = np.ascontiguousarray(seeds)

    def run():
        out = np.zeros(cap, dtype=engine.PATCH_DTYPE)
        n, ptot = C.c_longlong(), C.c_longlong()
        assert host.mvshost_run(sc.nviews, sc.W, sc.H, P.ctypes.data, img.ctypes.data, 0, 2, 7, 3, C.c_float(0.7), 9, iters, sd.shape[0], sd.ctypes.data,
                                cap, out.ctypes.data, C.byref(n), C.byref(ptot)) == 0
        return out[: n.value].copy(), ptot.value

    plain, total_plain = run()
    idf = tmp_path / "comm.id"
    idf.write_bytes(b"\0" * 160)  # what an earlier job that died may have left behind: replaced, never read
    host.mvshost_set_ranks(0, 1, str(idf).encode(), 0)
    try:
        ranked, total_ranked = run()
        assert not idf.exists()  # rank 0 removes the id file once its communicator stands: a later job cannot pick it up
        again, total_again = run()  # the same path a second time: a new id, a new file, the same patches
    finally:
        host.mvshost_set_ranks(0, 0, b"", 0)
    assert not idf.exists() and total_again == total_ranked and again.tobytes() == ranked.tobytes()
    assert total_plain == total_ranked and total_plain > 5000
    assert plain.shape == ranked.shape and plain.shape[0] > seeds.shape[0]
    assert plain.tobytes() == ranked.tobytes()


@pytest.mark.gpu
def test_pmmvps_run_matches_oracle(host, small_plane_scene, tmp_path):
    import oracle_binding as ob

    sc = small_plane_scene
    seeds = synth.make_seeds(sc, stride=4, seed=21)
    iters = 2
    o = ob.Oracle(sc.nviews, level=0, minImageNum=2, enable_check=1, seed=9, schedule=ob.SCHEDULE_ENGINE, sum_mode=ob.SUM_TREE64, nthreads=8,
                  depth=1)
    o.set_scene(sc)
    o.add_patches(seeds)
    total = 0
    for it in range(iters):
        total += o.propagate(it)["patches"]
        o.filter()            # Filter::run (pmmvps.cpp:101)
        o.update_threshold()  # PmMvps::updateThreshold + ++m_depth (pmmvps.cpp:103-105)
    po = o.patches()
    out = np.zeros(po.shape[0] + 1000, dtype=engine.PATCH_DTYPE)
    nout, ptot = C.c_longlong(), C.c_longlong()
    P = np.ascontiguousarray(sc.P, dtype=np.float32)
    img = np.ascontiguousarray(sc.images)
    sd = np.ascontiguousarray(seeds)
    ply = tmp_path / "out.ply"
    host.mvshost_set_ply_output(str(ply).encode())
    r = host.mvshost_run(sc.nviews, sc.W, sc.H, P.ctypes.data, img.ctypes.data, 0, 2, 7, 2, C.c_float(0.7), 9, iters, sd.shape[0], sd.ctypes.data,
                         out.shape[0], out.ctypes.data, C.byref(nout), C.byref(ptot))
    host.mvshost_set_ply_output(b"")
    assert r == 0
    assert ptot.value == total and nout.value == po.shape[0]
    pe = out[: nout.value]
    np.testing.assert_array_equal(pe["images"], po["images"])
    np.testing.assert_allclose(pe["coord"], po["coord"], rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(pe["normal"], po["normal"], rtol=0, atol=1e-3)

    # PatchManager::writePly (patch_manager.cpp:542-633): vertex colour = mean over m_images of the bilinear sample at the
    # patch's projection
    lines = ply.read_text().split("\n")
    body = lines[lines.index("end_header") + 1:]
    assert int(lines[2].split()[-1]) == pe.shape[0] and len([b for b in body if b]) == pe.shape[0]
    for k in range(0, pe.shape[0], max(1, pe.shape[0] // 50)):
        rec = pe[k]
        acc = np.zeros(3, np.float64)
        for v in rec["images"][: rec["nimages"]]:
            x = sc.P[v].astype(np.float64) @ rec["coord"].astype(np.float64)
            fx, fy = x[0] / x[2], x[1] / x[2]
            lx, ly = int(fx), int(fy)
            dx, dy = fx - lx, fy - ly
            im = sc.images[v].astype(np.float64)
            acc += (im[ly, lx] * (1 - dx) * (1 - dy) + im[ly + 1, lx] * (1 - dx) * dy + im[ly, lx + 1] * dx * (1 - dy) + im[ly + 1, lx + 1] * dx * dy)
        exp = np.minimum(255, np.floor(acc / rec["nimages"] + 0.5))
        got = np.array(body[k].split()[6:9], dtype=np.float64)
        assert np.all(np.abs(got - exp) <= 1), (k, got, exp)
        np.testing.assert_allclose(np.array(body[k].split()[:3], dtype=np.float64), rec["coord"][:3], rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
def test_dataset_on_disk_equals_in_memory_run(host, small_plane_scene, tmp_path):
    """The reference's driver sequence on files (test/test.cpp:155-161): option file (option.cpp:35-149), CONTOUR cameras
    (camera.cpp:27-63,102-141), binary PPM images, seeds in the .patch text format (patch.cpp:31-56,
    patch_manager.cpp:435-466), PLY / .patch output (patch_manager.cpp:499-633) -- same result as the in-memory run."""
    sc = small_plane_scene
    seeds = synth.make_seeds(sc, stride=4, seed=21)
    root = tmp_path / "data"
    for d in ("txt", "image", "ply"):
        (root / d).mkdir(parents=True)
    (root / "option").write_text(f"# synthetic plane\nlevel 0\ncsize 2\nthreshold 0.7\nwsize 7\nminImageNum 2\nimages -1 0 {sc.nviews}\n")
    for v in range(sc.nviews):
        (root / "txt" / f"{v:08d}.txt").write_text("CONTOUR\n" + "\n".join(" ".join(repr(float(x)) for x in row) for row in sc.P[v]) + "\n")
        with open(root / "image" / f"{v:04d}0000.ppm", "wb") as f:
            f.write(b"P6\n%d %d\n255\n" % (sc.W, sc.H))
            f.write(np.ascontiguousarray(sc.images[v]).tobytes())
    with open(root / "ply" / "00000000.patch", "w") as f:
        f.write(f"PATCHES\n{seeds.shape[0]}\n")
        for r in seeds:
            f.write("PATCHS\n" + " ".join(repr(float(x)) for x in r["coord"]) + "\n" + " ".join(repr(float(x)) for x in r["normal"]) + "\n")
            f.write(f"{float(r['ncc'])!r} {float(r['dscale'])!r} {float(r['ascale'])!r}\n{int(r['nimages'])}\n")
            f.write(" ".join(str(int(x)) for x in r["images"][: r["nimages"]]) + "\n0\n\n")
    iters = 2
    cap = 200000
    out_f = np.zeros(cap, dtype=engine.PATCH_DTYPE)
    nf = C.c_longlong()
    assert host.mvshost_run_dataset((str(root) + "/").encode(), iters, 9, cap, out_f.ctypes.data, C.byref(nf)) == 0
    out_m = np.zeros(cap, dtype=engine.PATCH_DTYPE)
    nm, ptot = C.c_longlong(), C.c_longlong()
    P = np.ascontiguousarray(sc.P, dtype=np.float32)
    img = np.ascontiguousarray(sc.images)
    sd = np.ascontiguousarray(seeds)
    assert host.mvshost_run(sc.nviews, sc.W, sc.H, P.ctypes.data, img.ctypes.data, 0, 2, 7, 2, C.c_float(0.7), 9, iters, sd.shape[0], sd.ctypes.data,
                            cap, out_m.ctypes.data, C.byref(nm), C.byref(ptot)) == 0
    assert nf.value == nm.value and nf.value > seeds.shape[0]
    a, b = out_f[: nf.value], out_m[: nm.value]
    for f in ("coord", "normal", "ncc", "nimages", "images"):
        np.testing.assert_array_equal(a[f], b[f], err_msg=f)
    # outputs of PmMvps::run: one PLY per iteration (before and after Filter::run) and the final .patch file
    for it in range(iters):
        for name in (f"refined_patches_before_refine_{it}.ply", f"refined_patches_{it}.ply"):
            head = (root / "ply" / name).read_text().split("\n")[:3]
            assert head[0] == "ply" and head[2].startswith("element vertex ")
    txt = (root / "ply" / "final.patch").read_text().split()
    assert txt[0] == "PATCHES" and int(txt[1]) == nf.value
    np.testing.assert_allclose([float(x) for x in txt[3:6]], a["coord"][0][:3], rtol=1e-5)


@pytest.mark.gpu
def test_jpeg_contour2_ply_seed_dataset(host, small_plane_scene, tmp_path):
    """The ingestion the reference's own data sets use, end to end: CONTOUR2 cameras (camera.cpp:117-134), JPEG images
    (image.cpp:827-879; written here by PIL, read by the mirror's decoder), PGM masks, seeds from a point-cloud PLY plus one
    normal-map PLY per view (depth_normal_init.cpp:34-144).  PmMvps::run on the directory must give exactly the patches of
    the in-memory run on PIL's pixels, the probed projections and the seeds DepthNormInit::buildPatches reports."""
    pytest.importorskip("PIL")
    import scipy.linalg
    from PIL import Image

    host.mvshost_set_seed_plys.argtypes = [C.c_int]
    host.mvshost_set_seed_plys.restype = None
    host.mvshost_seeds_from_plys.argtypes = [C.c_char_p, C.c_longlong, C.c_void_p]
    host.mvshost_seeds_from_plys.restype = C.c_longlong
    sc = small_plane_scene
    root = tmp_path / "data"
    for d in ("txt", "image", "mask", "ply"):
        (root / d).mkdir(parents=True)
    (root / "option").write_text(f"level 0\ncsize 2\nthreshold 0.7\nwsize 7\nminImageNum 2\nimages -1 0 {sc.nviews}\n")
    P_host = np.zeros((sc.nviews, 3, 4), np.float32)
    decoded = np.zeros_like(sc.images)
    n_world = np.array([0.0, 0.0, 1.0])
    for v in range(sc.nviews):
        K, R = scipy.linalg.rq(sc.P[v][:, :3].astype(np.float64))
        S = np.diag(np.sign(np.diag(K)))
        K, R = K @ S, S @ R
        t = np.linalg.solve(K, sc.P[v][:, 3].astype(np.float64))
        K = K / K[2, 2]
        b = -math.asin(R[2, 0])
        a, g = math.atan2(R[2, 1], R[2, 2]), math.atan2(R[1, 0], R[0, 0])
        vals = [K[0, 0], K[1, 1], K[0, 1], K[0, 2], K[1, 2], 0.0, math.degrees(a), math.degrees(b), math.degrees(g), t[0], t[1], t[2]]
        (root / "txt" / f"{v:08d}.txt").write_text("CONTOUR2\n" + " ".join(repr(float(x)) for x in vals) + "\n")
        assert host.mvshost_camera_probe(str(root / "txt" / f"{v:08d}.txt").encode(), P_host[v].ctypes.data) == 0
        np.testing.assert_allclose(P_host[v] / P_host[v][2, 3], sc.P[v] / sc.P[v][2, 3], rtol=0, atol=2e-3)
        Image.fromarray(sc.images[v]).save(root / "image" / f"{v:04d}0000.jpg", quality=95, subsampling=1)
        decoded[v] = np.asarray(Image.open(root / "image" / f"{v:04d}0000.jpg"))
        with open(root / "mask" / f"{v:08d}.pgm", "wb") as f:
            f.write(b"P5\n%d %d\n255\n" % (sc.W, sc.H) + bytes([255]) * (sc.W * sc.H))
        ys, xs = np.mgrid[0:sc.H, 0:sc.W]
        n_cam = R.T @ (n_world if (R @ n_world)[2] < 0 else -n_world)  # towards the camera; the mirror rotates it back with R
        with open(root / "ply" / f"{v + 1:08d}.ply", "wb") as f:
            f.write(f"ply\nformat binary_little_endian 1.0\nelement vertex {sc.W * sc.H}\nproperty float x\nproperty float y\nproperty float z\n"
                    "property float nx\nproperty float ny\nproperty float nz\nend_header\n".encode())
            rec = np.zeros((sc.H, sc.W, 6), "<f4")
            rec[..., 0], rec[..., 1] = xs, ys
            rec[..., 3:] = n_cam
            f.write(rec.tobytes())
    seeds0 = synth.make_seeds(sc, stride=4, seed=21)
    with open(root / "ply" / "00000000.ply", "w") as f:
        f.write(f"ply\nformat ascii 1.0\nelement vertex {seeds0.shape[0]}\nproperty float x\nproperty float y\nproperty float z\nend_header\n")
        for r in seeds0:
            f.write(" ".join(repr(float(x)) for x in r["coord"][:3]) + "\n")
    prefix = (str(root) + "/").encode()
    seeds = np.zeros(seeds0.shape[0], dtype=engine.PATCH_DTYPE)
    ns = host.mvshost_seeds_from_plys(prefix, seeds.shape[0], seeds.ctypes.data)
    assert 0.8 * seeds0.shape[0] <= ns <= seeds0.shape[0]
    seeds = np.ascontiguousarray(seeds[:ns])
    assert np.all(seeds["nimages"] >= 2)
    iters, cap = 2, 200000
    out_f = np.zeros(cap, dtype=engine.PATCH_DTYPE)
    nf = C.c_longlong()
    host.mvshost_set_seed_plys(1)
    try:
        assert host.mvshost_run_dataset(prefix, iters, 9, cap, out_f.ctypes.data, C.byref(nf)) == 0
    finally:
        host.mvshost_set_seed_plys(0)
    out_m = np.zeros(cap, dtype=engine.PATCH_DTYPE)
    nm, ptot = C.c_longlong(), C.c_longlong()
    img = np.ascontiguousarray(decoded)
    assert host.mvshost_run(sc.nviews, sc.W, sc.H, P_host.ctypes.data, img.ctypes.data, 0, 2, 7, 2, C.c_float(0.7), 9, iters, seeds.shape[0], seeds.ctypes.data,
                            cap, out_m.ctypes.data, C.byref(nm), C.byref(ptot)) == 0
    assert nf.value == nm.value and nf.value > 2 * ns
    a, b = out_f[: nf.value], out_m[: nm.value]
    for f in ("coord", "normal", "ncc", "nimages", "images"):
        np.testing.assert_array_equal(a[f], b[f], err_msg=f)
