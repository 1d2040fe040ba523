"""A second, independent reading of the reference's hot-path arithmetic -- written in numpy straight from the reference's
source text (file:line cited per function), with no code shared with oracle/pmmvs_oracle.cpp -- compared with the oracle in
its reference summation order (ORC_SUM_SEQ).  The reference ships no golden vectors and cannot be built here (Eigen, CImg,
NLopt absent), so this is what stands between "the oracle agrees with itself" and "two restatements of the same source
agree": pyramids byte for byte, projections / patch axes / sampling / normalisation / NCC to float rounding (the oracle's
documented deviations are reciprocal-multiplies and fused multiply-adds)."""
import numpy as np
import pytest

import oracle_binding as ob
from mvskit_amd import synth

F = np.float32


# ------------------------------------------------------------------ image/image.cpp
def ref_pyr_down(img):
    """Image::buildImagePyramid, image.cpp:245-315, filter 0: 4x4 [1 3 3 1]x[1 3 3 1]/64 at stride 2, taps outside the image
    dropped without renormalising, divided by mask.sum() (= 1) again, round half up."""
    H, W = img.shape[:2]
    h, w = H // 2, W // 2
    mask = np.array([[1, 3, 3, 1], [3, 9, 9, 3], [3, 9, 9, 3], [1, 3, 3, 1]], dtype=F)
    mask = (mask / mask.sum()).astype(F)
    out = np.zeros((h, w, 3), np.uint8)
    src = img.astype(F)
    for y in range(h):
        for x in range(w):
            color = np.zeros(3, F)
            for i in range(-1, 3):
                yt = 2 * y + i
                if yt < 0 or H - 1 < yt:
                    continue
                for j in range(-1, 3):
                    xt = 2 * x + j
                    if xt < 0 or W - 1 < xt:
                        continue
                    color = (color + mask[i + 1, j + 1] * src[yt, xt]).astype(F)
            color = (color / mask.sum(dtype=F)).astype(F)
            out[y, x] = np.floor(color + F(0.5)).astype(np.int32).astype(np.uint8)
    return out


def ref_get_color(img, x, y):
    """Image::getColor(float, float, level), bilinear branch, image.cpp:447-472."""
    lx, ly = int(x), int(y)
    dx1 = F(x) - F(lx)
    dx0 = F(1) - dx1
    dy1 = F(y) - F(ly)
    dy0 = F(1) - dy1
    f00, f01, f10, f11 = dx0 * dy0, dx0 * dy1, dx1 * dy0, dx1 * dy1
    p = img.astype(F)
    c = (p[ly, lx] * f00 + p[ly + 1, lx] * f01).astype(F)
    c = (c + (p[ly, lx + 1] * f10 + p[ly + 1, lx + 1] * f11).astype(F)).astype(F)
    return c


# ------------------------------------------------------------------ image/camera.cpp
def ref_project(P, X):
    """Camera::project, camera.cpp:310-326 (P = m_projections[level])."""
    ic = (P.astype(F) @ X.astype(F)).astype(F)
    if ic[2] <= 0:
        return np.array([-65535, -65535, -1], F)
    return (ic / ic[2]).astype(F)


def ref_unproject(P, ic):
    """Camera::unproject, camera.cpp:329-337."""
    M = P[:, :3].astype(np.float64)
    b = ic.astype(F) - P[:, 3].astype(F)
    return np.append((np.linalg.inv(M) @ b.astype(np.float64)).astype(F), F(1))


class RefCam:
    """Camera::updateCamera (camera.cpp:65-100) + Optim::setAxesScales (optim.cpp:43-65) for one view."""

    def __init__(self, P0, level):
        self.P = [P0.astype(F)]
        for _ in range(1, level + 3):
            q = self.P[-1].copy()
            q[:2] = (q[:2] / F(2)).astype(F)  # updateProjection, camera.cpp:91-100
            self.P.append(q)
        M = P0[:, :3].astype(np.float64)
        self.center = np.append((-np.linalg.inv(M) @ P0[:, 3].astype(np.float64)).astype(F), F(1))  # getCameraCenter, :295-308
        o = P0[2].astype(F)
        self.oaxis = (o / np.linalg.norm(o[:3]).astype(F)).astype(F)  # camera.cpp:76-80
        z = self.oaxis[:3]
        x = P0[0, :3].astype(F)
        y = np.cross(z, x).astype(F)
        y = (y / np.linalg.norm(y).astype(F)).astype(F)
        x = np.cross(y, z).astype(F)
        self.xaxis, self.yaxis, self.zaxis = x, y, z
        self.ipscale = F(np.dot(P0[0].astype(F), np.append(x, F(0)))) + F(np.dot(P0[1].astype(F), np.append(y, F(0))))


# ------------------------------------------------------------------ pmmvps/optim.cpp
def ref_get_unit(cam, coord, level):
    """Optim::getUnit, optim.cpp:34-41 (evaluated in double, returned as float)."""
    fz = np.linalg.norm((coord - cam.center).astype(F)).astype(F)
    if cam.ipscale == 0:
        return F(1)
    return F(2.0 * float(fz) * (1 << level) / float(cam.ipscale))


def ref_get_paxes(cam, coord, normal, level):
    """Optim::getPAxes, optim.cpp:67-84."""
    pscale = ref_get_unit(cam, coord, level)
    n3 = normal[:3].astype(F)
    y3 = np.cross(n3, cam.xaxis).astype(F)
    y3 = (y3 / np.linalg.norm(y3).astype(F)).astype(F)
    x3 = np.cross(y3, n3).astype(F)
    px = (np.append(x3, F(0)) * pscale).astype(F)
    py = (np.append(y3, F(0)) * pscale).astype(F)
    c0 = ref_project(cam.P[level], coord)
    xdis = np.linalg.norm((ref_project(cam.P[level], (coord + px).astype(F)) - c0).astype(F)).astype(F)
    ydis = np.linalg.norm((ref_project(cam.P[level], (coord + py).astype(F)) - c0).astype(F)).astype(F)
    return (px / xdis).astype(F), (py / ydis).astype(F)


def ref_get_tex(cam, pyr, coord, px, py, pz, level, wsize, cos_thr):
    """Optim::getTex (optim.cpp:790-844) with getTexSafe (895-915); pyr[l] = HxWx3 uint8."""
    ray = (cam.center - coord).astype(F)
    ray = (ray / np.linalg.norm(ray).astype(F)).astype(F)
    weight = max(F(0), F(np.dot(ray, pz)))
    if weight < cos_thr:
        return None
    margin = wsize // 2
    center = ref_project(cam.P[level], coord)
    dx = (ref_project(cam.P[level], (coord + px).astype(F)) - center).astype(F)
    dy = (ref_project(cam.P[level], (coord + py).astype(F)) - center).astype(F)
    ratio = (np.linalg.norm(dx).astype(F) + np.linalg.norm(dy).astype(F)) / F(2)
    ld = int(np.floor(np.log(float(ratio)) / np.log(2.0) + 0.5))
    ld = max(-level, min(2, ld))
    scale = F(2.0 ** ld)
    nl = level + ld
    center, dx, dy = (center / scale).astype(F), (dx / scale).astype(F), (dy / scale).astype(F)
    m = F(margin)
    corners = [center - dx * m - dy * m, center + dx * m - dy * m, center - dx * m + dy * m, center + dx * m + dy * m]
    xs, ys = [c[0] for c in corners], [c[1] for c in corners]
    H, W = pyr[nl].shape[:2]
    if min(xs) < 2 or W - 1 - 2 <= max(xs) or min(ys) < 2 or H - 1 - 2 <= max(ys):
        return None
    tl = (center - dx * m - dy * m).astype(F)
    tex = np.zeros((wsize * wsize, 3), F)
    for y in range(wsize):
        for x in range(wsize):
            samp = (tl + dx * F(x) + dy * F(y)).astype(F)
            tex[y * wsize + x] = ref_get_color(pyr[nl], samp[0], samp[1])
    return tex


def ref_normalize(tex):
    """Optim::normalize, optim.cpp:917-940 (sequential float sums)."""
    sz = tex.shape[0]
    ave = np.zeros(3, F)
    for t in tex:
        ave = (ave + t).astype(F)
    ave = (ave / F(sz)).astype(F)
    ssd = F(0)
    for t in tex:
        d = (t - ave).astype(F)
        ssd = F(ssd + F(np.dot(d, d)))
    msd = F(np.sqrt(ssd / F(3 * sz)))
    if msd == 0:
        msd = F(1)
    return ((tex - ave).astype(F) / msd).astype(F)


def ref_dot(t0, t1):
    """Optim::dot, optim.cpp:601-609."""
    s = F(0)
    for a, b in zip(t0, t1):
        s = F(s + F(np.dot(a, b)))
    return F(s / F(3 * t0.shape[0]))


def ref_robustincc(x):  # optim.cpp:622-624
    return F(x) / (F(1) + F(3) * F(x))


def ref_compute_incc(cams, pyrs, coord, normal, idx, level, wsize, tau, robust, cos_thr):
    """Optim::computeINCC (optim.cpp:630-706, non-PAIRNCC branch) with computeWeights (942-948) / computeUnits (109-132)."""
    if len(idx) < 2:
        return F(2)
    units = []
    for v in idx:
        u = ref_get_unit(cams[v], coord, level)
        ray = (cams[v].center - coord).astype(F)
        ray = (ray / np.linalg.norm(ray).astype(F)).astype(F)
        d = F(np.dot(ray, normal))
        units.append(F(u / d) if d > 0 else F(2 ** 31 // 2))
    w = [F(1)] + [min(F(1), F(units[0] / u)) for u in units[1:]]
    px, py = ref_get_paxes(cams[idx[0]], coord, normal, level)
    sz = min(tau, len(idx))
    texs = []
    for i in range(sz):
        t = ref_get_tex(cams[idx[i]], pyrs[idx[i]], coord, px, py, normal, level, wsize, cos_thr)
        texs.append(None if t is None else ref_normalize(t))
    if texs[0] is None:
        return F(2)
    score, total = F(0), F(0)
    for i in range(1, sz):
        if texs[i] is None:
            continue
        total = F(total + w[i])
        incc = F(1.0 - float(ref_dot(texs[0], texs[i])))
        score = F(score + (ref_robustincc(incc) if robust else incc) * w[i])
    return F(2) if total == 0 else F(score / total)


# ------------------------------------------------------------------ the comparison
@pytest.fixture(scope="module")
def setup():
    sc = synth.make_scene(nviews=4, W=160, H=120, arc_deg=45.0, radius=4.0, kind="multi")
    o = ob.Oracle(sc.nviews, level=0, csize=2, wsize=7, minImageNum=2, schedule=ob.SCHEDULE_ENGINE, sum_mode=ob.SUM_SEQ, enable_check=0, seed=1)
    o.set_scene(sc)
    cams = [RefCam(sc.P[v], 0) for v in range(sc.nviews)]
    pyrs = []
    for v in range(sc.nviews):
        levels = [sc.images[v]]
        for _ in range(2):
            levels.append(ref_pyr_down(levels[-1]))
        pyrs.append(levels)
    seeds = synth.make_seeds(sc, stride=9, seed=4)
    return sc, o, cams, pyrs, seeds


def test_pyramid_second_reading(setup):
    sc, o, cams, pyrs, _ = setup
    for v in (0, 3):
        for level in (1, 2):
            np.testing.assert_array_equal(o.pyramid(v, level), pyrs[v][level])


def test_camera_second_reading(setup):
    sc, o, cams, _, seeds = setup
    for v in range(sc.nviews):
        c = o.camera(v)
        np.testing.assert_allclose(c["center"], cams[v].center, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(c["oaxis"], cams[v].oaxis, rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(c["xaxis"], cams[v].xaxis, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(c["yaxis"], cams[v].yaxis, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(c["zaxis"], cams[v].zaxis, rtol=1e-5, atol=1e-6)
        assert abs(c["ipscale"] - cams[v].ipscale) <= 1e-5 * abs(cams[v].ipscale)
    for s in seeds[:40]:
        X = s["coord"].astype(F)
        for v in (0, 2):
            for level in (0, 1):
                P = cams[v].P[level]
                ic = ref_project(P, X)
                np.testing.assert_allclose(o.project(v, X, level), ic, rtol=2e-6, atol=1e-4)
                homog = (ic * F(np.dot(P[2], X))).astype(F)  # (u*w, v*w, w): unproject inverts P exactly
                back = ref_unproject(P, homog)
                np.testing.assert_allclose(o.unproject(v, np.append(homog, F(1)), level), back, rtol=1e-4, atol=1e-4)
                np.testing.assert_allclose(back[:3], X[:3], rtol=1e-3, atol=1e-3)


def test_units_axes_and_samples_second_reading(setup):
    sc, o, cams, pyrs, seeds = setup
    rng = np.random.RandomState(3)
    for s in seeds[:30]:
        X, N = s["coord"].astype(F), s["normal"].astype(F)
        v = int(s["images"][0])
        assert abs(o.get_unit(v, X) - ref_get_unit(cams[v], X, 0)) <= 2e-6 * ref_get_unit(cams[v], X, 0)
        px, py = o.get_paxes(v, X, N)
        rx, ry = ref_get_paxes(cams[v], X, N, 0)
        np.testing.assert_allclose(px, rx, rtol=2e-5, atol=1e-9)
        np.testing.assert_allclose(py, ry, rtol=2e-5, atol=1e-9)
    for _ in range(200):
        v, level = rng.randint(sc.nviews), rng.randint(3)
        H, W = pyrs[v][level].shape[:2]
        x, y = F(rng.uniform(1, W - 2)), F(rng.uniform(1, H - 2))
        np.testing.assert_allclose(o.get_color(v, x, y, level), ref_get_color(pyrs[v][level], x, y), rtol=1e-6, atol=1e-4)


def test_texture_and_incc_second_reading(setup):
    sc, o, cams, pyrs, seeds = setup
    cos_thr = F(np.cos(F(60.0 * np.pi / 180.0)))
    checked = 0
    for s in seeds[:60]:
        X, N = s["coord"].astype(F), s["normal"].astype(F)
        idx = [int(i) for i in s["images"][: s["nimages"]]]
        if len(idx) < 2:
            continue
        px, py = ref_get_paxes(cams[idx[0]], X, N, 0)
        t = ref_get_tex(cams[idx[1]], pyrs[idx[1]], X, px, py, N, 0, 7, cos_thr)
        flag, tex = o.get_tex(X, px, py, N, idx[1], normalize=False)
        assert (t is None) == (flag != 0)
        if t is not None:
            np.testing.assert_allclose(tex.reshape(49, 3), t, rtol=1e-5, atol=2e-3)  # positions differ by float rounding of the axes
            flag, texn = o.get_tex(X, px, py, N, idx[1], normalize=True)
            np.testing.assert_allclose(texn.reshape(49, 3), ref_normalize(t), rtol=1e-3, atol=2e-3)
        for robust in (0, 1):
            got = o.compute_incc(s, robust=robust)
            exp = ref_compute_incc(cams, pyrs, X, N, idx, 0, 7, min(2 * 2, sc.nviews), robust, cos_thr)  # m_tau, pmmvps.cpp:32
            assert abs(got - exp) <= 5e-5 * max(1.0, abs(exp)), (got, exp)  # measured worst case 6.4e-6
            checked += 1
    assert checked > 60
    for r in np.linspace(0, 0.33, 12):
        assert abs(ob.lib().orc_robustincc(float(r)) - ref_robustincc(r)) < 1e-7


# ------------------------------------------------------------------ refinement variables, cost, scales, neighbours
ASCALE = F(np.pi / F(48.0))  # Optim::refinePatch, optim.cpp:487


def ref_encode(cam, center, ray, dscale, coord, normal):
    """Optim::encode, optim.cpp:549-580 (libm calls; the oracle's own polynomials are within 3e-7 of them)."""
    x0 = F(np.dot((coord - center).astype(F), ray)) / F(dscale)
    n3 = normal[:3].astype(F)
    fx, fy, fz = F(np.dot(cam.xaxis, n3)), F(np.dot(cam.yaxis, n3)), F(np.dot(cam.zaxis, n3))
    a2 = F(np.arcsin(max(F(-1), min(F(1), fy))))
    cosb = F(np.cos(a2))
    if cosb == 0:
        a1 = F(0)
    else:
        sina, cosa = F(fx / cosb), F(-fz / cosb)
        a1 = F(np.arccos(max(F(-1), min(F(1), cosa))))
        if sina < 0:
            a1 = -a1
    return np.array([x0, a1 / ASCALE, a2 / ASCALE], F)


def ref_decode(cam, center, ray, dscale, x):
    """Optim::decode, optim.cpp:582-599."""
    coord = (center + F(dscale) * F(x[0]) * ray).astype(F)
    a1, a2 = F(x[1] * ASCALE), F(x[2] * ASCALE)
    fx, fy, fz = F(np.sin(a1) * np.cos(a2)), F(np.sin(a2)), F(-np.cos(a1) * np.cos(a2))
    n3 = (cam.xaxis * fx + cam.yaxis * fy + cam.zaxis * fz).astype(F)
    return coord, np.append(n3, F(0))


def ref_cost_func(cams, pyrs, center, ray, dscale, idx, x, level, wsize, tau, min_image_num, cos_thr):
    """Optim::cost_func, optim.cpp:401-468 (the non-pairwise branch)."""
    coord, normal = ref_decode(cams[idx[0]], center, ray, dscale, x)
    px, py = ref_get_paxes(cams[idx[0]], coord, normal, level)
    sz = min(tau, len(idx))
    minimum = min(min_image_num, sz)
    texs = []
    for i in range(sz):
        t = ref_get_tex(cams[idx[i]], pyrs[idx[i]], coord, px, py, normal, level, wsize, cos_thr)
        texs.append(None if t is None else ref_normalize(t))
    if texs[0] is None:
        return 2.0
    ans, denom = 0.0, 0
    for i in range(1, sz):
        if texs[i] is None:
            continue
        ans += float(ref_robustincc(F(1.0 - float(ref_dot(texs[0], texs[i])))))
        denom += 1
    return 2.0 if denom < minimum - 1 else ans / denom


def ref_set_scales(cams, coord, idx, level, wsize, tau):
    """PatchManager::setScales, patch_manager.cpp:378-399 (m_dscale starts at 0, patch.cpp:13-25)."""
    unit = ref_get_unit(cams[idx[0]], coord, level)
    unit2 = F(2) * unit
    ray = (coord - cams[idx[0]].center).astype(F)
    ray = (ray / np.linalg.norm(ray).astype(F)).astype(F)
    num = min(tau, len(idx))
    ds = F(0)
    for i in range(1, num):
        P = cams[idx[i]].P[level]
        diff = (ref_project(P, coord) - ref_project(P, (coord - unit2 * ray).astype(F))).astype(F)
        ds = F(ds + np.linalg.norm(diff).astype(F))
    ds = F(ds / F(num - 1))
    ds = F(unit2 / ds)
    return ds, F(np.arctan(ds / (unit * F(wsize) / F(2))))


def ref_is_neighbor(cams, a, b, csize, level, thr):
    """PmMvps::isNeighbor, pmmvps.cpp:117-147 -- including the reference's constant cosf(120 / pi * 180)."""
    hunit = F((ref_get_unit(cams[int(a["images"][0])], a["coord"].astype(F), level) + ref_get_unit(cams[int(b["images"][0])], b["coord"].astype(F), level)) / F(2) * F(csize))
    na, nb = a["normal"].astype(F), b["normal"].astype(F)
    if F(np.dot(na, nb)) < F(np.cos(F(F(120.0) / F(np.pi) * F(180.0)))):
        return 0
    diff = (a["coord"] - b["coord"]).astype(F)
    vunit = F(a["dscale"] + b["dscale"])
    f0, f1 = F(np.dot(na, diff)), F(np.dot(nb, diff))
    ftmp = F(F(abs(f0) + abs(f1)) / F(2) / vunit)
    hsize = F(np.linalg.norm((diff - f0 * na + diff - f1 * nb).astype(F)).astype(F) / F(2) / hunit)
    if 1.0 < hsize:
        ftmp = F(ftmp / min(F(2), hsize))
    return 1 if ftmp < thr else 0


def test_encode_decode_and_cost_second_reading(setup):
    sc, o, cams, pyrs, seeds = setup
    cos_thr = F(np.cos(F(60.0 * np.pi / 180.0)))
    rng = np.random.RandomState(7)
    checked = 0
    for s in seeds[:40]:
        idx = [int(i) for i in s["images"][: s["nimages"]]]
        if len(idx) < 2:
            continue
        X, N = s["coord"].astype(F), s["normal"].astype(F)
        rec = s.copy()
        rec["dscale"] = F(0.01)
        cam = cams[idx[0]]
        ray = (X - cam.center).astype(F)
        ray = (ray / np.linalg.norm(ray).astype(F)).astype(F)
        x = ref_encode(cam, X, ray, rec["dscale"], X, N)
        np.testing.assert_allclose(o.encode(rec), x, rtol=0, atol=2e-4)  # angles in units of pi/48: 2e-4 units = 1.3e-5 rad
        x2 = (x + rng.uniform(-1.5, 1.5, 3)).astype(F)
        c, n = o.decode(rec, x2)
        rc, rn = ref_decode(cam, X, ray, rec["dscale"], x2)
        np.testing.assert_allclose(c, rc, rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(n, rn, rtol=0, atol=2e-6)
        got = o.cost(rec, x2)
        exp = ref_cost_func(cams, pyrs, X, ray, rec["dscale"], idx, x2, 0, 7, min(4, sc.nviews), 2, cos_thr)
        assert abs(got - exp) <= 5e-5 * max(1.0, abs(exp)), (got, exp)
        checked += 1
    assert checked > 25


def test_scales_and_neighbours_second_reading(setup):
    sc, o, cams, pyrs, seeds = setup
    tau = min(4, sc.nviews)
    done = []
    for s in seeds[:60]:
        f, rec = o.preprocess(s)
        if f != 0:
            continue
        idx = [int(i) for i in rec["images"][: rec["nimages"]]]
        ds, asc = ref_set_scales(cams, rec["coord"].astype(F), idx, 0, 7, tau)
        # m_dscale is a quotient of ~1-pixel differences of projections ~100 pixels large: rounding of the projections (fused or not)
        # shows at 1e-5 relative
        assert abs(rec["dscale"] - ds) <= 2e-4 * ds and abs(rec["ascale"] - asc) <= 2e-4 * asc, (rec["dscale"], ds, rec["ascale"], asc)
        done.append(rec)
    assert len(done) > 30
    pairs = same = 0
    for i in range(0, len(done) - 1):
        for j in (i + 1, (i + 7) % len(done)):
            if i == j:
                continue
            for thr in (0.25, 1.0, 4.0):
                pairs += 1
                same += o.is_neighbor(done[i], done[j], thr) == ref_is_neighbor(cams, done[i], done[j], 2, 0, thr)
    assert pairs > 100 and same == pairs


def test_generate_patch_second_reading(setup):
    """Propagate::generatePatch, propagate.cpp:220-237: depth along the optical axis of the source's reference view, the pixel
    unprojected at that depth, the source's normal; views whose cell falls outside the grid are dropped (setGridsImages)."""
    sc, o, cams, pyrs, seeds = setup
    rng = np.random.RandomState(11)
    checked = 0
    for s in seeds[:60]:
        v = int(s["images"][0])
        X = s["coord"].astype(F)
        ic0 = ref_project(cams[v].P[0], X)
        ic = np.array([ic0[0] + rng.uniform(-3, 3), ic0[1] + rng.uniform(-3, 3), 1.0], F)
        f, rec = o.generate_patch(s, ic)
        if f != 0:
            continue
        depth = F(np.dot(cams[v].oaxis, X))
        exp = ref_unproject(cams[v].P[0], (depth * ic).astype(F))
        np.testing.assert_allclose(rec["coord"], exp, rtol=2e-5, atol=2e-5)
        np.testing.assert_array_equal(rec["normal"], s["normal"])
        kept = [int(i) for i in rec["images"][: rec["nimages"]]]
        assert kept and kept[0] == v and set(kept) <= {int(i) for i in s["images"][: s["nimages"]]}
        for i in kept:  # every kept view sees the new point inside its grid (cells of csize 2 at level 0)
            p = ref_project(cams[i].P[0], exp)
            gx, gy = int(np.floor(p[0] + F(0.5))) // 2, int(np.floor(p[1] + F(0.5))) // 2
            assert 0 <= gx < (sc.W + 1) // 2 and 0 <= gy < (sc.H + 1) // 2
        checked += 1
    assert checked > 40


def ref_quad_residual(cams, rec, coords, level, tau):
    """Filter::filterQuad / ortho / lls, filter.cpp:329-430: the least-squares quadric through the neighbours by SVD
    (Eigen's jacobiSvd().solve there, numpy's lstsq here), in float like the reference."""
    z = rec["normal"].astype(F)
    if abs(z[0]) > 0.5:
        x = np.array([z[1], -z[0], 0, 0], F)
    elif abs(z[1]) > 0.5:
        x = np.array([0, z[2], -z[1], 0], F)
    else:
        x = np.array([-z[2], 0, z[0], 0], F)
    x = (x / np.linalg.norm(x).astype(F)).astype(F)
    y = np.array([z[1] * x[2] - z[2] * x[1], z[2] * x[0] - z[0] * x[2], z[0] * x[1] - z[1] * x[0], 0], F)
    diff = (coords.astype(F) - rec["coord"].astype(F)).astype(F)
    h = F(0)
    for d in diff:
        h = F(h + np.linalg.norm(d).astype(F))
    h = F(h / F(len(diff)))
    fx, fy, fz = (diff @ x / h).astype(F), (diff @ y / h).astype(F), (diff @ z).astype(F)
    A = np.stack([fx * fx, fy * fy, fx * fy, fx, fy], 1).astype(F)
    sol = np.linalg.lstsq(A, fz, rcond=None)[0].astype(F)
    idx = [int(i) for i in rec["images"][: rec["nimages"]]]
    inum = min(tau, len(idx))
    unit = F(0)
    for i in range(inum):
        unit = F(unit + ref_get_unit(cams[idx[i]], rec["coord"].astype(F), level))
    unit = F(unit / F(inum))
    res = F(0)
    for n in range(len(diff)):
        r = sol[0] * (fx[n] * fx[n]) + sol[1] * (fy[n] * fy[n]) + sol[2] * (fx[n] * fy[n]) + sol[3] * fx[n] + sol[4] * fy[n] - fz[n]
        res = F(res + F(abs(r)) / unit)
    return F(res / F(len(diff) - 5))


def test_filter_quad_second_reading(setup):
    """The oracle solves the 5-parameter fit by normal equations in double, the reference by an SVD in float: the residual the
    decision rests on must agree (a curved, noisy neighbourhood and a flat one; the decision threshold is 2.5)."""
    sc, o, cams, pyrs, seeds = setup
    rng = np.random.RandomState(13)
    worst = 0.0
    for k, s in enumerate(seeds[:40]):
        n = int(rng.randint(7, 120))
        unit = float(ref_get_unit(cams[int(s["images"][0])], s["coord"].astype(F), 0))
        z = s["normal"][:3].astype(np.float64)
        t1 = np.cross(z, [1.0, 0.3, 0.2]); t1 /= np.linalg.norm(t1)
        t2 = np.cross(z, t1)
        uv = rng.uniform(-6, 6, (n, 2)) * unit
        curv = (0.0 if k % 3 == 0 else rng.uniform(-0.5, 0.5)) / unit
        noise = rng.normal(0, (0.1, 1.0, 4.0)[k % 3] * unit, n)
        hgt = curv * (uv[:, 0] ** 2 - 0.5 * uv[:, 1] ** 2 + 0.3 * uv[:, 0] * uv[:, 1]) + 0.2 * uv[:, 0] + noise
        pts = s["coord"][:3].astype(np.float64) + uv[:, :1] * t1 + uv[:, 1:] * t2 + hgt[:, None] * z
        coords = np.concatenate([pts, np.ones((n, 1))], 1).astype(F)
        got = float(o.quad_residual(s, coords))
        exp = float(ref_quad_residual(cams, s, coords, 0, min(4, sc.nviews)))
        assert abs(got - exp) <= 2e-3 * max(exp, 1e-3), (k, n, got, exp)
        assert (got < 2.5) == (exp < 2.5)
        worst = max(worst, abs(got - exp) / max(exp, 1e-3))
    assert worst < 2e-3


def test_set_inccs_matrix_second_reading(setup):
    """Optim::setINCCs (matrix form, optim.cpp:748-783), what Optim::setRefImage (348-383) sums by rows to pick the reference view."""
    sc, o, cams, pyrs, seeds = setup
    cos_thr = F(np.cos(F(60.0 * np.pi / 180.0)))
    checked = 0
    for s in seeds[:40]:
        idx = [int(i) for i in s["images"][: s["nimages"]]]
        if len(idx) < 3:
            continue
        X, N = s["coord"].astype(F), s["normal"].astype(F)
        px, py = ref_get_paxes(cams[idx[0]], X, N, 0)
        texs = []
        for v in idx:
            t = ref_get_tex(cams[v], pyrs[v], X, px, py, N, 0, 7, cos_thr)
            texs.append(None if t is None else ref_normalize(t))
        n = len(idx)
        exp = np.zeros((n, n), F)
        for i in range(n):
            for j in range(i + 1, n):
                exp[i, j] = exp[j, i] = F(2) if texs[i] is None or texs[j] is None else ref_robustincc(F(1) - ref_dot(texs[i], texs[j]))
        got = o.set_inccs_matrix(s, robust=1)
        np.testing.assert_allclose(got, exp, rtol=0, atol=5e-5)
        sums_e, sums_g = exp.sum(1), got.sum(1)
        if np.sort(sums_e)[1] - np.sort(sums_e)[0] > 1e-3:  # away from ties both readings choose the same reference view
            assert int(np.argmin(sums_e)) == int(np.argmin(sums_g))
        checked += 1
    assert checked > 15
