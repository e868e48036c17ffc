"""GPU (-m gpu): the optional grid search (pt_set_accel(PT_ACCEL_GRID)) against the oracle's
closest-hit search over every sphere (HittableList.cpp:21-36), to the same bar as the linear scan, and against
the linear scan of the same library on whole frames."""
import os

import numpy as np
import pytest

from test_gpu_parity import FAR_N_TOL, REL, check_scan, f32, spheres, tight_bar, vec_rel_err

pytestmark = pytest.mark.gpu

# TIGHT bars of the tests below (reported, not failing: see the header of test_gpu_parity.py): at most 2x the maxima the
# device measurably reaches (printed as "[measured] ..."). (mismatches, t, point, normal) / (rays, pixels off) / {search: (mismatches, t)}
CLOUD = (0, 1.6e-6, 3.6e-7, 1.4e-5)                      # measured 0, 8.0e-7, 1.8e-7, 7.0e-6
FRAME_A = (1e-9, 1e-9)                                  # measured: the same ray count, no pixel off (255 spheres, 96x54x3)
FRAME_B = (1e-9, 1e-9)                                  # the same at 101x57x6
RANDOM = {"grid": (2, 2.4e-6), "scan": (2, 2.4e-6)}     # measured (1, 1.2e-6) and (1, 1.2e-6)


@pytest.fixture(scope="module")
def gdev():
    from path_tracer_amd import Device
    d = Device(0)
    d.set_accel(1)
    yield d
    d.close()


def both(gdev, orc, sc, o, d, allow=2, t_min=0.001, far=False, name=None, tol=(1e-4, 1e-4, 1e-4)):
    gdev.set_scene(sc)
    hs = orc.scene_create(sc)
    ref = orc.scan(hs, o, d, t_min)
    out = gdev.scan(o, d, t_min)
    orc.scene_free(hs)
    # far: rays that start hundreds of units away. An fp32 direction places the hit point to ~|o| * 6e-8, which a
    # sphere of radius 0.1 turns into a few 1e-4 of normal. Index, t and point keep the 1e-4 bar.
    hit = check_scan(out, ref, allow_mismatch=allow, name=("grid: " + name) if name else None, t_tol=tol[0], p_tol=tol[1], n_tol=tol[2],
                     hard_n=1e-3 if far else REL)
    if far:
        assert hit.sum() > 0.5 * len(hit)
    return out, ref


def test_grid_scan_500_spheres(gdev, orc):
    from path_tracer_amd import host
    sc = host.generate_scene(500, 1, 2)
    rng = np.random.default_rng(7)
    n = 20000
    o = f32(np.stack([rng.uniform(-10, 10, n), rng.uniform(0.5, 6, n), rng.uniform(-10, 10, n)], axis=1))
    d = rng.normal(size=(n, 3))
    d[:, 1] = -np.abs(d[:, 1])
    d = f32(d * rng.uniform(0.05, 12, size=(n, 1)))
    # (bars: at most 2x the measured maxima, as in test_gpu_parity.py; the grid returns the very numbers of the scan here)
    out, ref = both(gdev, orc, sc, o, d, allow=0, name="505 spheres, 20k rays from above", tol=(3.4e-6, 4.3e-6, 8.8e-5))
    info = gdev.accel_info()
    assert 1 <= info["big"] <= 5 and info["entries"] >= 500 and min(info["dims"]) >= 1        # the ground sphere and the large fixed spheres are "big"
    # secondary rays: from just above the surfaces in every direction, and from far outside the grid box
    sel = ref["idx"] >= 0
    p0 = f32(ref["p"][sel] + 1e-3 * ref["n"][sel])
    d1 = f32(rng.normal(size=p0.shape))
    both(gdev, orc, sc, p0, d1, allow=0, name="505 spheres, secondary rays", tol=(1.6e-5, 1.4e-6, 7e-6))
    far = f32(np.stack([rng.uniform(-300, 300, 4000), rng.uniform(0.01, 40, 4000), rng.uniform(-300, 300, 4000)], axis=1))
    aim = f32(np.stack([rng.uniform(-10, 10, 4000), rng.uniform(0, 0.5, 4000), rng.uniform(-10, 10, 4000)], axis=1))
    both(gdev, orc, sc, far, f32(aim - far), allow=0, far=True, name="rays from 300 units away", tol=(9e-7, 2.9e-5, FAR_N_TOL))
    # axis-parallel and grazing directions (zero components in the walk)
    ax = np.array([[1, 0, 0], [-1, 0, 0], [0, 0, 1], [0, 0, -1], [0, -1, 0], [0, 1, 0], [1, 0, 1], [1, -1e-9, 0]], dtype=np.float64)
    oa = f32(np.stack([rng.uniform(-9, 9, 4000), rng.uniform(0.05, 0.6, 4000), rng.uniform(-9, 9, 4000)], axis=1))
    both(gdev, orc, sc, oa, ax[rng.integers(0, len(ax), 4000)], allow=0, name="axis-parallel rays", tol=(7e-8, 1e-10, 1e-10))


def test_grid_scan_and_frame_10k_spheres(gdev, dev, orc):
    """configs[4] through the grid: the scan against the oracle, a small frame against the oracle with the
    same philox numbers, and a larger frame against the linear scan of the same library."""
    from path_tracer_amd import host
    sc = host.generate_scene(10000, 1, 2, half_extent=45.0)
    cam = host.camera_setup()
    rng = np.random.default_rng(4)
    n = 8000
    o = f32(np.stack([rng.uniform(-40, 40, n), rng.uniform(0.5, 4, n), rng.uniform(-40, 40, n)], axis=1))
    d = rng.normal(size=(n, 3)); d[:, 1] = -np.abs(d[:, 1]) * 0.3
    both(gdev, orc, sc, o, f32(d), allow=0, name="9,912 spheres, 8k rays", tol=(1.2e-6, 4e-6, 3.4e-5))
    info = gdev.accel_info()
    assert info["big"] >= 1 and info["entries"] >= 9000 and info["dims"][0] > 50 and info["dims"][2] > 50
    gdev.set_camera(cam)
    hs = orc.scene_create(sc)
    img, rays = gdev.render_image(64, 36, 2, 50, seed=(2 << 32) | 1)
    ref, _, ref_rays = orc.render(hs, cam, 64, 36, 0, 2, 50, mode=1, seed_a=1, seed_b=2, threads=os.cpu_count() or 1)
    orc.scene_free(hs)
    diff = np.abs(img - ref).max(axis=2) / 2
    print("[measured] grid, 9,912-sphere frame 64x36x2 against the oracle: pixels off by > 2e-3: %.4f, rays %d vs %d" % ((diff > 2e-3).mean(), rays, ref_rays))
    assert (diff > 2e-3).mean() < 0.02 and abs(rays - ref_rays) < 0.01 * ref_rays
    tight_bar("grid 9,912 spheres vs oracle: pixels off", (diff > 2e-3).mean(), 0.0044); tight_bar("grid 9,912 spheres vs oracle: rays", abs(rays / ref_rays - 1), 3.5e-3)   # measured 0.0022, 22 of 12,604
    # grid against linear scan, same seed: the paths are the same until a grazing decision differs
    dev.set_scene(sc); dev.set_camera(cam)
    a, ra = gdev.render_image(320, 180, 4, 50, seed=11)
    b, rb = dev.render_image(320, 180, 4, 50, seed=11)
    diff = np.abs(a - b).max(axis=2) / 4
    print("[measured] grid against scan, 320x180x4 on 9,912 spheres: pixels off by > 2e-3: %.4f, rays %d vs %d" % ((diff > 2e-3).mean(), ra, rb))
    assert (diff > 2e-3).mean() < 0.005 and abs(ra - rb) < 0.005 * rb
    tight_bar("grid vs scan 9,912 spheres: pixels off", (diff > 2e-3).mean(), 0.0004); tight_bar("grid vs scan 9,912 spheres: rays", abs(ra / rb - 1), 1.2e-4)   # measured 0.0002, 36 of 629,293
    print("[measured] grid against scan: mean ratio %.7f" % (float(a.mean()) / float(b.mean())))
    assert abs(float(a.mean()) - float(b.mean())) < 0.001 * float(b.mean())
    a2, ra2 = gdev.render_image(320, 180, 4, 50, seed=11)
    assert ra2 == ra and np.array_equal(a, a2)                       # integer sums: repeatable bit for bit


def test_grid_edges(gdev, orc, dev):
    import json
    from conftest import GOLDEN
    from path_tracer_amd import PtError
    with open(os.path.join(GOLDEN, "kat.json")) as f:
        kat = json.load(f)
    for name in ("sphere_hit", "sphere_hit_inside"):
        k = kat[name]
        gdev.set_scene(spheres([k["sphere"]]))
        r = gdev.scan([k["ray"]["o"]], [k["ray"]["d"]])
        assert r["idx"][0] == 0 and r["t"][0] == pytest.approx(k["t"], rel=1e-6) and r["front"][0] == k["front"]
        assert np.allclose(r["p"][0], k["p"], atol=1e-5) and np.allclose(r["n"][0], k["n"], atol=1e-6)
    k = kat["tie_goes_to_later_index"]
    gdev.set_scene(spheres(k["spheres"]))
    r = gdev.scan([k["ray"]["o"]], [k["ray"]["d"]])
    assert r["idx"][0] == k["idx"] and r["t"][0] == pytest.approx(k["t"], rel=1e-6)
    # duplicates of one small sphere among others: the last copy wins, whatever cell order finds first
    lst = [{"c": [0.3 * i, 0, -6], "r": 0.2} for i in range(-8, 9)]
    lst += [{"c": [0, 0, -6], "r": 0.2}] * 3
    gdev.set_scene(spheres(lst))
    r = gdev.scan([[0, 0, 0]], [[0, 0, -1]])
    assert r["idx"][0] == len(lst) - 1
    # a miss, a sphere behind the ray, t_min cutting the near root, an empty batch
    gdev.set_scene(spheres([{"c": [0, 0, -5], "r": 1}]))
    r = gdev.scan([[0, 0, 0], [0, 0, 0], [0, 0, 0]], [[0, 1, 0], [0, 0, 1], [0, 0, -1]])
    assert list(r["idx"]) == [-1, -1, 0]
    r = gdev.scan([[0, 0, 0]], [[0, 0, -1]], t_min=5.0)
    assert r["idx"][0] == 0 and r["t"][0] == pytest.approx(6.0, rel=1e-6) and r["front"][0] == 0
    assert len(gdev.scan(np.zeros((0, 3)), np.zeros((0, 3)))["idx"]) == 0
    # small counts; mixed sizes (every sphere big, none big, half and half); a 3-D cloud rather than a plane
    rng = np.random.default_rng(3)
    for n, rmax in ((1, 1.0), (2, 1.0), (5, 1.0), (9, 1.0), (64, 0.3), (400, 0.25), (400, 3.0)):
        lst = [{"c": list(rng.uniform(-6, 6, 3) + np.array([0, 0, -14])), "r": float(rng.uniform(0.05, rmax))} for _ in range(n)]
        o = f32(rng.uniform(-1, 1, size=(2048, 3)))
        d = f32(np.stack([rng.uniform(-0.6, 0.6, 2048), rng.uniform(-0.6, 0.6, 2048), -np.ones(2048)], axis=1))
        both(gdev, orc, spheres(lst), o, d, allow=CLOUD[0], name="cloud of %d, r < %g, from outside" % (n, rmax), tol=CLOUD[1:])
        # rays that start inside the cloud
        o2 = f32(rng.uniform(-6, 6, size=(2048, 3)) + np.array([0, 0, -14]))
        both(gdev, orc, spheres(lst), o2, f32(rng.normal(size=(2048, 3))), allow=CLOUD[0], name="cloud of %d, r < %g, from inside" % (n, rmax), tol=CLOUD[1:])
    with pytest.raises(PtError):
        gdev.set_accel(7)
    # back to the full scan on the same handle: identical to a handle that never left it
    from path_tracer_amd import host
    sc = host.generate_scene(250, 1, 2)
    cam = host.camera_setup()
    gdev.set_scene(sc); gdev.set_camera(cam)
    dev.set_scene(sc); dev.set_camera(cam)
    gdev.set_accel(0)
    a, ra = gdev.render_image(96, 54, 3, 50, seed=5)
    b, rb = dev.render_image(96, 54, 3, 50, seed=5)
    gdev.set_accel(1)
    assert ra == rb and np.array_equal(a, b)
    c, rc = gdev.render_image(96, 54, 3, 50, seed=5)
    print("[measured] grid against scan, 96x54x3 on 255 spheres: pixels off by > 2e-3: %.4f, rays %d vs %d" % ((np.abs(c - b).max(axis=2) / 3 > 2e-3).mean(), rc, rb))
    assert abs(rc - rb) < 0.005 * rb and (np.abs(c - b).max(axis=2) / 3 > 2e-3).mean() < 0.005
    tight_bar("grid vs scan 96x54x3: rays", abs(rc / rb - 1), FRAME_A[0]); tight_bar("grid vs scan 96x54x3: pixels off", (np.abs(c - b).max(axis=2) / 3 > 2e-3).mean(), FRAME_A[1])
    # the corners of the sample loop in grid mode: depth 0 and 1, an empty sample range, partial tiles, sample ranges that add up
    img, rays = gdev.render_image(40, 23, 3, 0)
    assert rays == 0 and not img.any()
    img, rays = gdev.render_image(40, 23, 2, 1)
    assert rays == 40 * 23 * 2
    gdev.clear_accum(40, 23)
    assert gdev.render(40, 23, 5, 5, 50, 1, want_rays=True) == 0 and not gdev.read_accum(40, 23).any()
    a, ra = gdev.render_image(101, 57, 6, 50, seed=9)
    b, rb = dev.render_image(101, 57, 6, 50, seed=9)
    print("[measured] grid against scan, 101x57x6 on 255 spheres: pixels off by > 2e-3: %.4f, rays %d vs %d" % ((np.abs(a - b).max(axis=2) / 6 > 2e-3).mean(), ra, rb))
    assert abs(ra - rb) < 0.005 * rb and (np.abs(a - b).max(axis=2) / 6 > 2e-3).mean() < 0.005
    tight_bar("grid vs scan 101x57x6: rays", abs(ra / rb - 1), FRAME_B[0]); tight_bar("grid vs scan 101x57x6: pixels off", (np.abs(a - b).max(axis=2) / 6 > 2e-3).mean(), FRAME_B[1])
    gdev.clear_accum(101, 57)
    r1 = gdev.render(101, 57, 0, 2, 50, 9, want_rays=True)
    r2 = gdev.render(101, 57, 2, 6, 50, 9, want_rays=True)
    assert r1 + r2 == ra and np.abs(gdev.read_accum(101, 57) - a).max() <= 4e-6 * max(1.0, float(a.max()))


def test_random_scenes_both_searches(gdev, dev, orc):
    """Seeded random scenes the generator never makes: radii over three decades, spheres that overlap and contain
    one another (so rays start INSIDE spheres and the far root of Sphere.cpp:19 decides), flat, cubic and
    clustered layouts. Both searches against the oracle's HittableList::isHit."""
    rng = np.random.default_rng(2026)
    worst = {"grid": (0, 0.0), "scan": (0, 0.0)}
    for case in range(8):
        n = int(rng.integers(3, 900))
        layout = case % 4
        if layout == 0:      # cube
            c = rng.uniform(-8, 8, size=(n, 3))
        elif layout == 1:    # slab
            c = np.stack([rng.uniform(-30, 30, n), rng.uniform(-0.5, 0.5, n), rng.uniform(-30, 30, n)], axis=1)
        elif layout == 2:    # clusters
            centres = rng.uniform(-20, 20, size=(5, 3))
            c = centres[rng.integers(0, 5, n)] + rng.normal(scale=1.0, size=(n, 3))
        else:                # a line of spheres plus a few enclosing ones
            c = np.stack([np.linspace(-15, 15, n), np.zeros(n), np.zeros(n)], axis=1) + rng.normal(scale=0.05, size=(n, 3))
        r = 10.0 ** rng.uniform(-2.0, 0.7, n)
        r[rng.integers(0, n, 2)] = rng.uniform(10, 60, 2)          # spheres that contain much of the scene
        lst = [{"c": list(map(float, c[i])), "r": float(r[i])} for i in range(n)]
        sc = spheres(lst)
        m = 4096
        o = f32(c[rng.integers(0, n, m)] + rng.normal(scale=2.0, size=(m, 3)))
        d = f32(rng.normal(size=(m, 3)) * rng.uniform(0.1, 5, size=(m, 1)))
        hs = orc.scene_create(sc)
        ref = orc.scan(hs, o, d)
        orc.scene_free(hs)
        assert (ref["idx"] >= 0).mean() > 0.5
        for which, device in (("grid", gdev), ("scan", dev)):
            device.set_scene(sc)
            out = device.scan(o, d)
            mism = out["idx"] != ref["idx"]
            hit = ~mism & (ref["idx"] >= 0)
            terr = float((np.abs(out["t"][hit] - ref["t"][hit]) / ref["t"][hit]).max())
            print("[measured] random scene %d (%d spheres), %s: %d of %d picked another sphere, t %.2e" % (case, n, which, mism.sum(), m, terr))
            worst[which] = (max(worst[which][0], int(mism.sum())), max(worst[which][1], terr))
            assert mism.sum() <= 8 and terr < REL, (case, n, which, int(mism.sum()), terr)
            tight_bar("random scene %d, %s: other sphere" % (case, which), mism.sum(), RANDOM[which][0]); tight_bar("random scene %d, %s: t" % (case, which), terr, RANDOM[which][1])
            assert np.array_equal(out["front"][hit], ref["front"][hit])
            assert (ref["front"][hit] == 0).sum() > 0              # rays that start inside a sphere are among them
    print("[measured] random scenes, worst case per search: %s" % worst)
