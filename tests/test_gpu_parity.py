"""GPU (-m gpu): the HIP path, called through the C-ABI, against the oracle on the same inputs,
against the committed fixtures and against the hand-derived known answers.

Tolerances (BASELINE.json north_star):
  * scene buffers: bit-exact (CPU tests: test_host.py);
  * per-ray t, point, normal, scatter direction under injected numbers: north_star's bar is 1e-4 relative to the
    double path (REL below) and that is what FAILS a test. Beside it every test has a TIGHT bar, at most 2x what the
    device measurably achieves on that input (each test prints its maxima as "[measured] ..."; DESIGN.md section 5
    lists them). The kernels are deterministic, so the measured numbers repeat exactly on gfx950 with one compiler;
    a value above its tight bar (a compiler change, a codegen difference) is REPORTED as a pytest warning and in
    gpurun_out/parity_tight_bars.json, and does not redden the suite as long as 1e-4 holds;
  * frame under identical philox numbers: per-pixel agreement up to the few paths whose discrete
    decisions flip in fp32; frame under different numbers: per-pixel RMSE within the expected
    sigma at matched spp and mean luminance within 0.5 %.
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
REL = 1e-4


def f32(a):
    return np.asarray(a, dtype=np.float32).astype(np.float64)


TIGHT = {}          # name -> (value, tight bar, held): written to gpurun_out/parity_tight_bars.json as the tests run


def tight_bar(name, value, bar):
    """The reported bar: a measured maximum above `bar` (2x what this kernel reached when the bar was set) warns, it does not
    fail; the failing bar is north_star's, asserted by the caller."""
    import warnings
    held = bool(value <= bar)
    TIGHT[name] = (float(value), float(bar), held)
    if not held:
        warnings.warn("[tight bar] %s: measured %.3e is above the tight bar %.3e (north_star's bar is asserted separately)" % (name, value, bar))
    try:
        from conftest import ROOT
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "parity_tight_bars.json"), "w") as f:
            json.dump({k: {"measured": v[0], "tight_bar": v[1], "held": v[2]} for k, v in TIGHT.items()}, f, indent=1)
    except OSError:
        pass
    return held


def vec_rel_err(a, b):
    """|a - b| relative to the length of b, per row."""
    return np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(b, axis=1), 1e-30)


@pytest.fixture(scope="module")
def kat():
    with open(os.path.join(GOLDEN, "kat.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def scene250(orc, dev):
    from path_tracer_amd import host
    sc = host.generate_scene(250, 1, 2)
    cam = host.camera_setup()
    dev.set_scene(sc)
    dev.set_camera(cam)
    hs = orc.scene_create(sc)
    yield sc, cam, hs
    orc.scene_free(hs)


def use(dev, sc, cam=None):
    dev.set_scene(sc)
    if cam is not None:
        dev.set_camera(cam)


def spheres(lst):
    sc = {k: [] for k in ("cx", "cy", "cz", "r", "kind", "ar", "ag", "ab", "fuzz", "ior")}
    for s in lst:
        sc["cx"].append(s["c"][0]); sc["cy"].append(s["c"][1]); sc["cz"].append(s["c"][2]); sc["r"].append(s["r"])
        sc["kind"].append(s.get("kind", 0))
        a = s.get("albedo", (1, 1, 1))
        sc["ar"].append(a[0]); sc["ag"].append(a[1]); sc["ab"].append(a[2])
        sc["fuzz"].append(s.get("fuzz", 0.0)); sc["ior"].append(s.get("ior", 1.5))
    return sc


# ---------------------------------------------------------------------------------------------
def test_philox_known_answers_and_oracle(dev, orc, kat):
    ctr = np.array([v["ctr"] for v in kat["philox4x32_10"]], dtype=np.uint32)
    for v in kat["philox4x32_10"]:
        assert list(dev.philox([v["ctr"]], v["key"])[0]) == v["out"]
    rng = np.random.default_rng(1)
    ctr = rng.integers(0, 2**32, size=(4096, 4), dtype=np.uint64).astype(np.uint32)
    key = np.array([123456789, 987654321], dtype=np.uint32)
    out = dev.philox(ctr, key)
    ref = np.stack([orc.philox(c, key) for c in ctr[:512]])
    assert np.array_equal(out[:512], ref)


def test_gen_ray_T5(dev, orc, scene250):
    sc, cam, hs = scene250
    use(dev, sc, cam)
    g = np.load(os.path.join(GOLDEN, "t5_gen_ray.npz"))
    o, d = dev.gen_ray(int(g["W"]), int(g["H"]), g["pi"], g["pj"], g["u1"], g["u2"], g["diskx"], g["disky"])
    print("[measured] T5 ray generation fixture: o %.2e d %.2e" % (vec_rel_err(o, g["o"]).max(), vec_rel_err(d, g["d"]).max()))
    assert vec_rel_err(o, g["o"]).max() < REL and vec_rel_err(d, g["d"]).max() < REL
    tight_bar("T5 fixture origin", vec_rel_err(o, g["o"]).max(), 1.2e-7); tight_bar("T5 fixture direction", vec_rel_err(d, g["d"]).max(), 3.9e-7)
    # a non-default camera and a 4K frame, against the live oracle
    kw = dict(pos=(3, 5, 9), look=(1, 0.5, 0), up=(0, 1, 0), aspect=1.7777777777, focal=1.5, fov=40.0, aperture=0.3)
    cam2 = orc.camera_setup(**kw)
    dev.set_camera(cam2)
    rng = np.random.default_rng(5)
    n = 4096
    pi, pj = rng.integers(0, 3840, n), rng.integers(0, 2160, n)
    u1, u2 = f32(rng.random(n)), f32(rng.random(n))
    dx, dy = f32(rng.uniform(-0.7, 0.7, n)), f32(rng.uniform(-0.7, 0.7, n))
    o, d = dev.gen_ray(3840, 2160, pi, pj, u1, u2, dx, dy)
    oo, od = orc.gen_ray(cam2, 3840, 2160, pi, pj, u1, u2, dx, dy)
    print("[measured] T5 ray generation, 4K frame: o %.2e d %.2e" % (vec_rel_err(o, oo).max(), vec_rel_err(d, od).max()))
    assert vec_rel_err(o, oo).max() < REL and vec_rel_err(d, od).max() < REL
    tight_bar("T5 4K origin", vec_rel_err(o, oo).max(), 1e-7); tight_bar("T5 4K direction", vec_rel_err(d, od).max(), 3e-7)
    dev.set_camera(cam)


FAR_N_TOL = 6.2e-4  # normals of hits seen from 300 units away (measured 3.1e-4): see test_scan_rays_from_far_away
MEASURED = {}       # name -> measured maxima, printed by each test (pytest -s shows them; DESIGN.md section 5 records them)


def check_scan(g, r, allow_mismatch=0, name=None, t_tol=REL, p_tol=REL, n_tol=REL, hard_n=REL):
    """Device scan against the double path. FAILS on north_star's 1e-4 (hard_n for the normal, which one test must widen
    for a stated physical reason); t_tol / p_tol / n_tol are the tight bars (at most 2x what the device measurably
    achieves): above them the test warns and records, it does not fail."""
    mism = g["idx"] != r["idx"]
    hit = ~mism & (r["idx"] >= 0)
    assert hit.sum() > 0
    m = dict(mismatches=int(mism.sum()), n=len(mism),
             t=float((np.abs(g["t"][hit] - r["t"][hit]) / r["t"][hit]).max()),
             p=float(vec_rel_err(g["p"][hit], r["p"][hit]).max()),
             nrm=float(np.abs(g["n"][hit] - r["n"][hit]).max()))          # unit vectors: absolute = relative
    if name:
        MEASURED[name] = m
        print("[measured] %s: %s" % (name, m))
    # a ray within fp32 rounding of a silhouette may pick the other sphere: the failing bar leaves room for a few more than measured
    assert m["mismatches"] <= allow_mismatch + 3 + len(mism) // 5000, "index mismatches: %d" % m["mismatches"]
    tight_bar((name or "scan") + ": other sphere", m["mismatches"], allow_mismatch)
    assert m["t"] < REL and m["p"] < REL and m["nrm"] < hard_n, m
    label = name or "scan"
    tight_bar(label + ": t", m["t"], t_tol); tight_bar(label + ": point", m["p"], p_tol); tight_bar(label + ": normal", m["nrm"], n_tol)
    assert np.array_equal(g["front"][hit], r["front"][hit])
    return hit


def test_scan_T3_fixture(dev, scene250):
    sc, cam, hs = scene250
    use(dev, sc, cam)
    g = np.load(os.path.join(GOLDEN, "t3_scan.npz"))
    r = {k: g["r_" + k] for k in ("t", "idx", "p", "n", "front")}
    out = dev.scan(g["o"], g["d"])
    hit = check_scan(out, r, allow_mismatch=0, name="T3 fixture (255 spheres, 4096 rays)", t_tol=1.5e-6, p_tol=1.2e-6, n_tol=6e-6)
    assert (r["idx"][hit] == 1).sum() > 100          # plenty of ground hits (r = 1000), many of them grazing


def test_scan_T3_live_500_spheres(dev, orc):
    from path_tracer_amd import host
    sc = host.generate_scene(500, 1, 2)
    cam = host.camera_setup()
    use(dev, sc, cam)
    hs = orc.scene_create(sc)
    rng = np.random.default_rng(7)
    n = 20000
    # rays from above the scene in random downward directions, un-normalised
    o = f32(np.stack([rng.uniform(-10, 10, n), rng.uniform(0.5, 6, n), rng.uniform(-10, 10, n)], axis=1))
    d = rng.normal(size=(n, 3))
    d[:, 1] = -np.abs(d[:, 1])
    d = f32(d * rng.uniform(0.05, 12, size=(n, 1)))
    check_scan(dev.scan(o, d), orc.scan(hs, o, d), allow_mismatch=0, name="505 spheres, 20k rays from above", t_tol=3.4e-6, p_tol=4.3e-6, n_tol=8.8e-5)
    # rays leaving the surfaces (secondary rays), starting just above the hit point
    r0 = orc.scan(hs, o, d)
    sel = r0["idx"] >= 0
    p0 = f32(r0["p"][sel] + 1e-3 * r0["n"][sel])
    d1 = f32(rng.normal(size=p0.shape))
    check_scan(dev.scan(p0, d1), orc.scan(hs, p0, d1), allow_mismatch=2, name="505 spheres, secondary rays", t_tol=1.6e-5, p_tol=1.4e-6, n_tol=7e-6)
    orc.scene_free(hs)


def test_scan_rays_from_far_away(dev, orc):
    """Rays that start hundreds of units out (a bounce off the far ground) and aim at the small spheres: the scan
    works from the point of the ray's line nearest the world origin, so the distance of the start does not eat
    the digits of the discriminant: no sphere is mistaken for another (0 of 6,000; 349 with the textbook form).
    The normal cannot keep 1e-4 here whatever the kernel does: the ray's direction is an fp32 vector, so after
    |o| ~ 300-420 units the point where it arrives is uncertain by |o| x 2^-24 ~ 2.5e-5, and a sphere of radius 0.1
    turns that into 2.5e-4 of normal (measured 3.1e-4). t and the point keep north_star's 1e-4."""
    from path_tracer_amd import host
    sc = host.generate_scene(500, 1, 2)
    use(dev, sc, host.camera_setup())
    hs = orc.scene_create(sc)
    rng = np.random.default_rng(11)
    n = 6000
    far = f32(np.stack([rng.uniform(-300, 300, n), rng.uniform(0.01, 40, n), rng.uniform(-300, 300, n)], axis=1))
    aim = f32(np.stack([rng.uniform(-10, 10, n), rng.uniform(0, 0.5, n), rng.uniform(-10, 10, n)], axis=1))
    d = f32(aim - far)
    out, ref = dev.scan(far, d), orc.scan(hs, far, d)
    orc.scene_free(hs)
    hit = check_scan(out, ref, allow_mismatch=0, name="rays from 300 units away", t_tol=9e-7, p_tol=7.8e-5, n_tol=FAR_N_TOL, hard_n=1e-3)
    assert hit.sum() > 0.5 * n and (ref["idx"][hit] > 4).sum() > 0.2 * n        # plenty of small spheres among the hits


def test_scan_kats_and_edges(dev, orc, kat):
    for name in ("sphere_hit", "sphere_hit_inside"):
        k = kat[name]
        use(dev, spheres([k["sphere"]]))
        r = dev.scan([k["ray"]["o"]], [k["ray"]["d"]])
        assert r["idx"][0] == 0 and r["t"][0] == pytest.approx(k["t"], rel=1e-6)
        assert np.allclose(r["p"][0], k["p"], atol=1e-5) and np.allclose(r["n"][0], k["n"], atol=1e-6)
        assert r["front"][0] == k["front"]
    k = kat["tie_goes_to_later_index"]
    use(dev, spheres(k["spheres"]))
    r = dev.scan([k["ray"]["o"]], [k["ray"]["d"]])
    assert r["idx"][0] == k["idx"] and r["t"][0] == pytest.approx(k["t"], rel=1e-6)
    # a miss, a sphere behind the ray, t_min cutting the near root, an empty batch
    use(dev, spheres([{"c": [0, 0, -5], "r": 1}]))
    r = dev.scan([[0, 0, 0], [0, 0, 0], [0, 0, 0]], [[0, 1, 0], [0, 0, 1], [0, 0, -1]])
    assert list(r["idx"]) == [-1, -1, 0]
    r = dev.scan([[0, 0, 0]], [[0, 0, -1]], t_min=5.0)          # near root 4 < t_min: far root 6
    assert r["idx"][0] == 0 and r["t"][0] == pytest.approx(6.0, rel=1e-6) and r["front"][0] == 0
    r = dev.scan(np.zeros((0, 3)), np.zeros((0, 3)))
    assert len(r["idx"]) == 0
    # sphere counts around the pair/quad padding
    rng = np.random.default_rng(3)
    for n in (1, 2, 3, 4, 5, 7, 8, 9):
        lst = [{"c": list(rng.uniform(-3, 3, 3) + np.array([0, 0, -8])), "r": float(rng.uniform(0.3, 1.0))} for _ in range(n)]
        sc = spheres(lst)
        use(dev, sc)
        hs = orc.scene_create(sc)
        o = np.zeros((512, 3))
        d = f32(np.stack([rng.uniform(-0.6, 0.6, 512), rng.uniform(-0.6, 0.6, 512), -np.ones(512)], axis=1))
        check_scan(dev.scan(o, d), orc.scan(hs, o, d), allow_mismatch=0, name="%d spheres (padding)" % n, t_tol=5.4e-7, p_tol=6.8e-7, n_tol=4.1e-6)
        orc.scene_free(hs)


def test_scatter_T4_fixture(dev):
    g = np.load(os.path.join(GOLDEN, "t4_scatter.npz"))
    s = dev.scatter(g["kind"], g["albedo"], g["fuzz"], g["ior"], g["d_in"], g["p"], g["nrm"], g["front"], g["u"])
    assert np.array_equal(s["ok"], g["r_ok"])
    m = dict(d=float(vec_rel_err(s["d"], g["r_d"]).max()), atten=float(np.abs(s["atten"] - g["r_atten"]).max()))
    print("[measured] T4 scatter fixture: %s" % m)
    assert m["d"] < REL and m["atten"] == 0.0, m                  # the albedos of the fixture are fp32 values
    tight_bar("T4 scatter fixture", m["d"], 1e-6)                 # measured 4.8e-7
    for kind in (0, 1, 2):
        assert (g["kind"] == kind).sum() > 100


def test_scatter_kats(dev, kat):
    k = kat["lambertian"]
    s = dev.scatter([0], [k["albedo"]], [0.0], [0.0], [[0, -1, 0]], [[0, 0, 0]], [k["n"]], [1], [k["u"]])
    assert np.allclose(s["d"][0], k["d_out"], atol=1e-6) and np.allclose(s["atten"][0], k["atten"], atol=1e-7) and s["ok"][0] == 1
    k = kat["metal_mirror"]
    s = dev.scatter([1], [k["albedo"]], [k["fuzz"]], [0.0], [k["d_in"]], [[0, 0, 0]], [k["n"]], [1], [k["u"]])
    assert np.allclose(s["d"][0], k["d_out"], atol=1e-6) and s["ok"][0] == 1
    for name in ("dielectric_quirk", "dielectric_reflect_by_coin"):
        k = kat[name]
        s = dev.scatter([2], [[1, 1, 1]], [0.0], [k["ior"]], [k["d_in"]], [[0, 0, 0]], [k["n"]], [k["front"]], [[0, 0, k["u3"]]])
        assert np.allclose(s["d"][0], k["d_out"], atol=1e-6) and s["ok"][0] == 1 and np.allclose(s["atten"][0], [1, 1, 1])
    for v in kat["sphere_point"]:
        s = dev.scatter([1], [[1, 1, 1]], [1.0], [0.0], [[0, -1, 0]], [[0, 0, 0]], [[0, 1, 0]], [1], [[v["u1"], v["u2"], 0]])
        assert np.allclose(s["d"][0] - np.array([0, 1, 0]), v["v"], atol=2e-6)
    # a metal hit at exact grazing: refl.n = 0 is not > 0, the scatter is refused (Metal.cpp:14)
    s = dev.scatter([1], [[1, 1, 1]], [0.0], [0.0], [[1, 0, 0]], [[0, 0, 0]], [[0, 1, 0]], [1], [[0.1, 0.2, 0]])
    assert s["ok"][0] == 0


def test_round2_hand_derived_kats_on_the_device(dev, orc, kat):
    """The vectors added to kat.json in round 2 (TIR, oblique refraction, Schlick at cos 0.5, fuzzed metal, the
    Lambertian guard, pixel -> ray for a 2-pixel-wide image with an axis-aligned camera): the device answers them to
    fp32 accuracy. The CPU suite asserts the same vectors on the oracle (tests/test_oracle.py)."""
    import kat_cases
    worst = kat_cases.run_scatter_cases(dev.scatter, kat, atol=1e-7)        # measured 4.8e-8
    print("scatter KATs: largest deviation %.2e" % worst)
    k = kat["camera_axis_aligned"]
    cam = np.array(k["pos"] + k["lower_left"] + k["horizontal"] + k["vertical"] + k["u"] + k["v"] + [k["lens_radius"]], dtype=np.float64)
    dev.set_camera(cam)                                              # the hand-derived constants themselves
    p = kat["pixel_to_ray"]
    for r in p["rays"]:
        o, d = dev.gen_ray(p["width"], p["height"], [r["i"]], [r["j"]], [r["u1"]], [r["u2"]], [r["disk"][0]], [r["disk"][1]])
        assert np.abs(o[0] - r["o"]).max() < 1e-6 and np.abs(d[0] - r["d"]).max() < 1e-6, (r, o, d)
    # the first random sphere of seed 5489 (evaluated from the published mt19937 outputs) is hit where the arithmetic says
    from path_tracer_amd import host
    want = kat_cases.first_random_sphere(kat)
    sc = host.generate_scene(1, 5489, 5489)
    use(dev, sc, host.camera_setup())
    c = np.array([want["cx"], want["r"], want["cz"]])
    r = dev.scan([c + [0, 5, 0]], [[0, -2, 0]])                      # straight down onto its top: t = (5 - r) / 2
    assert r["idx"][0] == 5 and r["t"][0] == pytest.approx((5 - want["r"]) / 2, rel=1e-6)
    assert np.allclose(r["n"][0], [0, 1, 0], atol=1e-6) and np.allclose(r["p"][0], c + [0, want["r"], 0], atol=1e-5)


def test_round3_whole_path_kats_on_the_device(dev, kat):
    """Whole calcColour paths derived by hand (kat.json: whole_paths; arithmetic in each entry's `source`): miss -> sky,
    albedo x sky, product of two albedos, depth exhausted -> black, metal at exact grazing -> black, facing mirrors with
    depth 3 and 50, glass straight through with the un-normalised cosine, the short ray the quirk makes reflect, the tie
    that colours the path. pt_trace runs them on the device functions of the sample loop; colours AND ray counts."""
    import kat_cases
    from path_tracer_amd import host

    def trace_one(scene, o, d, max_depth, u):
        dev.set_scene(scene)
        return dev.trace(o, d, max_depth, u)

    worst = kat_cases.run_path_cases(trace_one, kat, atol=1e-4)         # north_star's bar; the tight one is reported
    print("[measured] whole-path KATs: largest colour deviation %.2e (tight bar 2e-6)" % worst)
    tight_bar("whole-path KATs", worst, 2e-6)
    dev.set_accel(1)                                                    # the same through the grid search
    try:
        worst_g = kat_cases.run_path_cases(trace_one, kat, atol=1e-4)
    finally:
        dev.set_accel(0)
    tight_bar("whole-path KATs, grid search", worst_g, 2e-6)
    use(dev, host.generate_scene(250, 1, 2), host.camera_setup())


def test_trace_edges(dev):
    """pt_trace: an empty batch, depth 0 without a uniform array, bad arguments, and the call before any scene is set."""
    from path_tracer_amd import Device, PtError, host
    use(dev, host.generate_scene(250, 1, 2), host.camera_setup())
    r = dev.trace(np.zeros((0, 3)), np.zeros((0, 3)), 5, np.zeros((0, 5, 3)))
    assert r["rgb"].shape == (0, 3) and len(r["rays"]) == 0
    r = dev.trace([[8, 2, 3]], [[-8, -2, -3]], 0, np.zeros((1, 0, 3)))
    assert list(r["rgb"][0]) == [0, 0, 0] and r["rays"][0] == 0                     # Path Tracer.cpp:86
    with pytest.raises(PtError, match="bad arguments"):
        dev.trace([[0, 0, 0]], [[0, 0, -1]], -1, np.zeros((1, 0, 3)))
    fresh = Device(0)
    try:
        with pytest.raises(PtError, match="set the scene first"):
            fresh.trace([[0, 0, 0]], [[0, 0, -1]], 3, np.zeros((1, 3, 3)))
    finally:
        fresh.close()
    # 1,000 copies of one ray give 1,000 identical answers (no cross-talk between the threads of a launch)
    u = np.tile(np.array([[0.3, 0.6, 0.2]]), (1000, 50, 1))
    r = dev.trace(np.tile([[8.0, 2.0, 3.0]], (1000, 1)), np.tile([[-8.0, -1.9, -3.0]], (1000, 1)), 50, u)
    assert np.all(r["rgb"] == r["rgb"][0]) and np.all(r["rays"] == r["rays"][0]) and r["rays"][0] >= 2


def test_set_scene_rejects_what_the_sums_cannot_hold(dev):
    """A negative or non-finite albedo is refused at pt_set_scene (the fixed-point sums hold non-negative radiance only)."""
    from path_tracer_amd import PtError, host
    sc = host.generate_scene(10, 1, 2)
    for bad in (-0.1, float("nan"), float("inf")):
        sc2 = {k: np.array(v, copy=True) for k, v in sc.items()}
        i = int(np.flatnonzero(sc2["kind"] != 2)[0])
        sc2["ag"][i] = bad
        with pytest.raises(PtError, match="albedo"):
            dev.set_scene(sc2)
    use(dev, host.generate_scene(250, 1, 2), host.camera_setup())


def test_trace_against_the_oracle_on_random_paths(dev, orc):
    """pt_trace against orc_trace on the 505-sphere scene: camera-like rays with random injected uniforms. Paths whose
    discrete decisions agree (same ray count) must agree in colour to fp32 accuracy; a few may flip a decision in fp32."""
    from path_tracer_amd import host
    sc = host.generate_scene(500, 1, 2)
    use(dev, sc, host.camera_setup())
    hs = orc.scene_create(sc)
    rng = np.random.default_rng(5)
    n, depth = 6000, 50
    o = np.tile(np.array([8.0, 2.0, 3.0]), (n, 1)) + f32(rng.normal(size=(n, 3)) * 0.05)
    tgt = np.stack([rng.uniform(-9, 9, n), rng.uniform(0, 1.5, n), rng.uniform(-9, 9, n)], axis=1)
    d = f32((tgt - o) * rng.uniform(0.5, 1.5, (n, 1)))
    u = f32(rng.integers(0, 1 << 24, size=(n, depth, 3)) / float(1 << 24))
    g = dev.trace(o, d, depth, u)
    r = orc.trace(hs, o, d, depth, u)
    orc.scene_free(hs)
    same = g["rays"] == r["rays"]
    err = np.abs(g["rgb"] - r["rgb"]).max(axis=1)
    agree = same & (err <= 1e-3)
    print("[measured] trace: %d of %d paths with another ray count, %d more off by > 1e-3; mean rays %.3f; median colour deviation %.2e, max among agreeing %.2e"
          % ((~same).sum(), n, (same & ~agree).sum(), r["rays"].mean(), np.median(err), err[agree].max()))
    assert r["rays"].mean() > 2.0 and r["rays"].max() >= 20
    assert (~agree).mean() < 0.01                                       # paths whose discrete decisions flipped in fp32
    assert np.median(err) < 1e-6 and err[agree].max() <= 1e-3
    assert abs(int(g["rays"].sum()) - int(r["rays"].sum())) < 0.005 * r["rays"].sum()
    assert abs(g["rgb"].mean() - r["rgb"].mean()) < 0.005 * r["rgb"].mean()
    use(dev, host.generate_scene(250, 1, 2), host.camera_setup())


def test_scatter_dielectric_regimes(dev, orc):
    """Total internal reflection, the un-normalised-cosine quirk and Schlick, against the oracle."""
    rng = np.random.default_rng(11)
    n = 6000
    nrm = rng.normal(size=(n, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    d = rng.normal(size=(n, 3))
    d -= nrm * np.maximum(0, (d * nrm).sum(1, keepdims=True)) * 2
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d *= rng.choice([0.05, 0.5, 1.0, 2.0, 9.0], size=(n, 1))
    d, nrm = f32(d), f32(nrm)
    keep = (d * nrm).sum(1) < -1e-3
    d, nrm = d[keep], nrm[keep]
    n = len(d)
    front = rng.integers(0, 2, n)
    u = f32(rng.random((n, 3)))
    kind = np.full(n, 2)
    args = (kind, np.ones((n, 3)), np.zeros(n), np.full(n, 1.5), d, np.zeros((n, 3)), nrm, front, u)
    g, r = dev.scatter(*args), orc.scatter(*args)
    # a coin that lands within fp32 rounding of the reflectance may flip: allow a handful
    err = vec_rel_err(g["d"], r["d"])
    same = err < REL
    print("[measured] dielectric regimes: %d of %d coins flipped, largest deviation of the rest %.2e" % ((~same).sum(), n, err[same].max()))
    assert (~same).sum() <= 3 and err[same].max() < REL, ((~same).sum(), err.max())    # a coin within fp32 rounding of the reflectance may flip
    tight_bar("dielectric regimes: coins flipped", (~same).sum(), 0); tight_bar("dielectric regimes: direction", err[same].max(), 2.3e-6)   # measured 0, 1.14e-6
    assert np.all(g["ok"] == 1)


def test_sky(dev, orc, kat):
    for v in kat["sky"]:
        assert np.allclose(dev.sky([v["d"]])[0], v["rgb"], atol=1e-6)
    rng = np.random.default_rng(2)
    d = f32(rng.normal(size=(2048, 3)) * 5)
    assert np.abs(dev.sky(d) - orc.sky(d)).max() < 1e-6


# ---------------------------------------------------------------------------------------------
def frame_vs_oracle(dev, orc, sc, cam, W, H, spp, depth, key0=1, key1=2, frac_tol=0.02):
    use(dev, sc, cam)
    hs = orc.scene_create(sc)
    img, rays = dev.render_image(W, H, spp, depth, seed=(key1 << 32) | key0)
    ref, _, ref_rays = orc.render(hs, cam, W, H, 0, spp, depth, mode=1, seed_a=key0, seed_b=key1, threads=os.cpu_count() or 1)
    orc.scene_free(hs)
    diff = np.abs(img.astype(np.float64) - ref).max(axis=2) / spp
    return img, ref, rays, ref_rays, float((diff > 2e-3).mean()), float(np.median(diff))


def test_frame_same_numbers_as_oracle(dev, orc, scene250):
    """T6, strongest form: the oracle draws the same philox numbers, so every path should match
    until an fp32 rounding flips a discrete decision."""
    sc, cam, hs = scene250
    img, ref, rays, ref_rays, frac_off, med = frame_vs_oracle(dev, orc, sc, cam, 96, 54, 8, 50)
    print("[measured] frame 96x54x8 under identical numbers: median %.2e, pixels off by > 2e-3: %.4f, rays %d vs %d, mean ratio %.6f"
          % (med, frac_off, rays, ref_rays, img.mean() / ref.mean()))
    # failing bars: fp32 accuracy of the typical pixel, at most 1 % of the pixels with a flipped decision, north_star's 0.5 % on the mean
    assert med < 1e-6 and frac_off < 0.01, (med, frac_off)
    assert abs(img.mean() - ref.mean()) < 0.005 * ref.mean() and abs(rays - ref_rays) < 0.005 * ref_rays
    tight_bar("frame 96x54x8: median", med, 5e-8); tight_bar("frame 96x54x8: pixels off", frac_off, 0.0028)          # measured 2.4e-8, 0.0014
    tight_bar("frame 96x54x8: mean", abs(img.mean() / ref.mean() - 1), 3e-5); tight_bar("frame 96x54x8: rays", abs(rays / ref_rays - 1), 2e-4)   # 1.5e-5, 8 of 99,119
    g = np.load(os.path.join(GOLDEN, "t6b_philox_96x54.npz"))
    d = np.abs(img.astype(np.float64) - g["sum"]).max(axis=2) / 8
    assert (d > 2e-3).mean() < 0.01 and abs(rays - int(g["rays"])) < 0.005 * int(g["rays"])
    tight_bar("frame 96x54x8 vs fixture: pixels off", (d > 2e-3).mean(), 0.0028)


def test_frame_odd_size_and_glass_heavy_scene(dev, orc):
    """400x225-like odd sizes (partial tiles) and the dielectric/metal-heavy scene of configs[3]."""
    from path_tracer_amd import host
    cam = host.camera_setup(aspect=101 / 57)
    sc = host.generate_scene(250, 3, 4, diffuse_below=0.1, metal_below=0.5)
    img, ref, rays, ref_rays, frac_off, med = frame_vs_oracle(dev, orc, sc, cam, 101, 57, 4, 100, key0=7, key1=9)
    print("[measured] glass-heavy frame 101x57x4: median %.2e, pixels off by > 2e-3: %.4f, rays %d vs %d, mean ratio %.6f"
          % (med, frac_off, rays, ref_rays, img.mean() / ref.mean()))
    assert med < 1e-6 and frac_off < 0.01, (med, frac_off)
    assert abs(img.mean() - ref.mean()) < 0.005 * ref.mean() and abs(rays - ref_rays) < 0.005 * ref_rays
    tight_bar("glass-heavy frame: median", med, 5e-8); tight_bar("glass-heavy frame: pixels off", frac_off, 0.0034)  # measured 2.4e-8, 0.0017
    tight_bar("glass-heavy frame: mean", abs(img.mean() / ref.mean() - 1), 2e-5); tight_bar("glass-heavy frame: rays", abs(rays / ref_rays - 1), 1e-3)   # 1.0e-5, 26 of 56,888


def test_glass_heavy_scene_statistics(dev, orc):
    """configs[3] (10 % diffuse, 40 % metal, 50 % glass, depth 100) against the oracle's render with the reference's
    Mersenne-Twister streams, made here at low resolution: the GPU frame at 64 times the samples has no noise of
    its own left, so chi^2/N is the oracle's noise (1.0) plus any bias of the device path."""
    from path_tracer_amd import host
    sc = host.generate_scene(250, 3, 4, diffuse_below=0.1, metal_below=0.5)
    cam = host.camera_setup(aspect=80 / 45)
    W, H, spp_ref, depth = 80, 45, 256, 100
    hs = orc.scene_create(sc)
    s, ss, ref_rays = orc.render(hs, cam, W, H, 0, spp_ref, depth, mode=0, seed_a=21, seed_b=22, threads=min(16, os.cpu_count() or 1),
                                 want_sumsq=True)
    orc.scene_free(hs)
    mean_ref = s / spp_ref
    var_ref = np.maximum(ss / spp_ref - mean_ref ** 2, 0.0) * spp_ref / (spp_ref - 1)      # per-sample variance
    use(dev, sc, cam)
    spp = 64 * spp_ref
    img, rays = dev.render_image(W, H, spp, depth, seed=31)
    mean = img.astype(np.float64) / spp
    var = var_ref * (1.0 / spp + 1.0 / spp_ref)
    chi = ((mean - mean_ref) ** 2).sum() / var.sum()
    assert 0.9 < chi < 1.1, chi
    assert abs(mean.mean() / mean_ref.mean() - 1) < 0.005
    assert abs(rays / (W * H * spp) / (ref_rays / (W * H * spp_ref)) - 1) < 0.01


def test_frame_statistics_T6(dev, scene250):
    """T6 as BASELINE states it: against the converged render made with the reference's Mersenne-Twister
    streams (fixture), per-pixel RMSE within the expected sigma at matched spp, mean luminance within 0.5 %."""
    sc, cam, hs = scene250
    use(dev, sc, cam)
    g = np.load(os.path.join(GOLDEN, "t6_image_160x90.npz"))
    spp = 256
    img, rays = dev.render_image(160, 90, spp, 50, seed=2026)
    mean = img.astype(np.float64) / spp
    lum_gpu, lum_ref = mean.mean(), g["mean"].astype(np.float64).mean()
    assert abs(lum_gpu - lum_ref) < 0.005 * lum_ref, (lum_gpu, lum_ref)
    var = g["var"].astype(np.float64) * (1.0 / spp + 1.0 / int(g["spp"]))     # expected variance of the difference
    chi = ((mean - g["mean"]) ** 2).sum() / var.sum()
    assert 0.85 < chi < 1.15, chi
    rmse = np.sqrt(((mean - g["mean"]) ** 2).mean())
    sigma = np.sqrt(var.mean())
    assert rmse <= 1.1 * sigma, (rmse, sigma)
    assert abs(rays / (160 * 90 * spp) - float(g["rays_per_sample"])) < 0.01 * float(g["rays_per_sample"])
    # Bias check: with 16 times the fixture's samples the GPU frame's own noise is negligible, so what is left of
    # chi^2 is the fixture's noise (1.0) plus any systematic difference; both closest-hit searches.
    from path_tracer_amd import Device
    for accel in (0, 1):
        d2 = Device(0)
        d2.set_accel(accel); d2.set_scene(sc); d2.set_camera(cam)
        big = 16 * int(g["spp"])
        img2, rays2 = d2.render_image(160, 90, big, 50, seed=777)
        d2.close()
        m2 = img2.astype(np.float64) / big
        v2 = g["var"].astype(np.float64) * (1.0 / big + 1.0 / int(g["spp"]))
        chi2 = ((m2 - g["mean"]) ** 2).sum() / v2.sum()
        assert 0.93 < chi2 < 1.07, (accel, chi2)
        assert abs(m2.mean() / lum_ref - 1) < 0.001
        assert abs(rays2 / (160 * 90 * big) - float(g["rays_per_sample"])) < 0.002 * float(g["rays_per_sample"])


def test_depth_zero_empty_range_and_errors(dev, scene250):
    from path_tracer_amd import Device, PtError
    sc, cam, hs = scene250
    use(dev, sc, cam)
    img, rays = dev.render_image(40, 23, 3, 0)            # Path Tracer.cpp:86
    assert rays == 0 and not img.any()
    dev.clear_accum(40, 23)
    assert dev.render(40, 23, 5, 5, 50, 1, want_rays=True) == 0
    assert not dev.read_accum(40, 23).any()
    img, rays = dev.render_image(40, 23, 2, 1)            # depth 1: exactly one ray per sample
    assert rays == 40 * 23 * 2
    with pytest.raises(PtError):
        dev.render(0, 10, 0, 1, 5, 1)
    with pytest.raises(PtError):
        dev.render(10, 10, 3, 1, 5, 1)
    with pytest.raises(PtError):
        dev.render(10, 10, 0, 1, -1, 1)
    fresh = Device(0)
    with pytest.raises(PtError, match="scene"):
        fresh.render(8, 8, 0, 1, 5, 1)
    with pytest.raises(PtError):
        fresh.set_scene(dict(cx=[0], cy=[0], cz=[0], r=[-1.0], kind=[0], ar=[1], ag=[1], ab=[1], fuzz=[0], ior=[1.5]))
    with pytest.raises(PtError):
        fresh.set_scene(dict(cx=[0], cy=[0], cz=[0], r=[1.0], kind=[7], ar=[1], ag=[1], ab=[1], fuzz=[0], ior=[1.5]))
    fresh.close()


def test_determinism_linearity_and_split(dev, scene250):
    """Size-independent properties: the same call twice is bit-identical (integer sums), a sample
    range is the sum of its parts, the job split does not change the frame."""
    sc, cam, hs = scene250
    use(dev, sc, cam)
    W, H = 200, 113
    a, ra = dev.render_image(W, H, 12, 50, seed=99)
    b, rb = dev.render_image(W, H, 12, 50, seed=99)
    assert ra == rb and np.array_equal(a, b)
    dev.clear_accum(W, H)
    r1 = dev.render(W, H, 0, 5, 50, 99, want_rays=True)
    r2 = dev.render(W, H, 5, 12, 50, 99, want_rays=True)
    c = dev.read_accum(W, H)
    assert r1 + r2 == ra
    assert np.abs(c - a).max() <= 4e-6 * max(1.0, float(a.max()))
    d, _ = dev.render_image(W, H, 12, 50, seed=100)
    assert not np.array_equal(a, d) and abs(float(a.mean()) - float(d.mean())) < 0.02 * float(a.mean())
    # samples [12, 24) are new samples, not a repeat of [0, 12)
    e, _ = dev.render_image(W, H, 12, 50, seed=99, spp_begin=12)
    assert not np.array_equal(a, e)


def test_frame_does_not_depend_on_the_job_split(dev, scene250, monkeypatch):
    """Jobs are an 8x8 tile x a chunk of samples, large ones first and small ones at the end of a launch, handed to
    whichever wave asks next at one of three priorities; the radiance sums are integers. So the frame and the ray count
    must be bit for bit the same whatever the job sizes: one chunk per tile, one sample per job, no small jobs, only
    small jobs; image sizes with partial tiles on both edges; sample ranges shorter than a chunk. (libpt_hip.so reads the
    settings in pt_create.)"""
    from path_tracer_amd import Device
    sc, cam, hs = scene250
    settings = [dict(),                                                                   # the defaults
                dict(PT_FINE_CHUNK="0"),                                                  # no small jobs
                dict(PT_FINE_CHUNK="1", PT_FINE_SHARE="400"),                             # one-sample jobs for half the call
                dict(PT_TARGET_JOBS="1"),                                                 # one large chunk per tile (+ small ones)
                dict(PT_TARGET_JOBS="100000000", PT_MIN_CHUNK="1", PT_FINE_CHUNK="0"),    # one sample per job throughout
                dict(PT_MIN_CHUNK="3", PT_FINE_CHUNK="2", PT_FINE_SHARE="37")]
    cases = [(200, 113, 0, 16), (67, 35, 3, 10), (8, 8, 0, 1), (33, 9, 5, 7)]
    want = None
    for env in settings:
        for k in ("PT_FINE_CHUNK", "PT_FINE_SHARE", "PT_TARGET_JOBS", "PT_MIN_CHUNK"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        d = Device(0)
        try:
            d.set_scene(sc); d.set_camera(cam)
            got = [d.render_image(W, H, s1 - s0, 50, seed=31, spp_begin=s0) for (W, H, s0, s1) in cases]
        finally:
            d.close()
        if want is None:
            want = got
        else:
            for (img, rays), (ref_img, ref_rays), case in zip(got, want, cases):
                assert rays == ref_rays and np.array_equal(img, ref_img), (env, case)


def test_external_accumulation_buffer(dev, scene250):
    import torch
    sc, cam, hs = scene250
    use(dev, sc, cam)
    W, H = 64, 36
    own, rays = dev.render_image(W, H, 4, 50, seed=5)
    buf = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    dev.set_stream(torch.cuda.current_stream().cuda_stream)
    dev.render(W, H, 0, 4, 50, 5, accum_ptr=buf.data_ptr())
    dev.render(W, H, 0, 4, 50, 5, accum_ptr=buf.data_ptr())      # adds: twice the frame
    torch.cuda.synchronize()
    dev.set_stream(0)
    assert np.allclose(buf.cpu().numpy(), 2 * own, rtol=1e-6, atol=1e-6)
    kt = dev.kernel_times()
    assert len(kt) >= 2 and np.all(kt > 0)


def test_full_size_properties(dev, orc):
    """configs[1] at full size: 1920x1080, 505 spheres, maxDepth 50 (fewer spp than the bench, the
    properties do not depend on it): ray count, luminance, black-free sky band, repeatability."""
    from path_tracer_amd import host
    sc = host.generate_scene(500, 1, 2)
    cam = host.camera_setup()
    use(dev, sc, cam)
    W, H, spp = 1920, 1080, 8
    img, rays = dev.render_image(W, H, spp, 50, seed=42)
    assert np.isfinite(img).all() and img.min() >= 0 and img.max() <= spp * 1.0001
    rps = rays / (W * H * spp)
    assert 2.3 < rps < 3.0
    # low-resolution oracle of the same scene and camera: same mean luminance and rays per sample
    hs = orc.scene_create(sc)
    ref, _, ref_rays = orc.render(hs, cam, 240, 135, 0, 16, 50, mode=1, seed_a=1, seed_b=2, threads=os.cpu_count() or 1)
    orc.scene_free(hs)
    assert abs(img.mean() / spp - ref.mean() / 16) < 0.01 * ref.mean() / 16
    assert abs(rps - ref_rays / (240 * 135 * 16)) < 0.02 * rps
    # block averages of the full frame against the oracle's low-resolution frame
    blk = img.reshape(135, 8, 240, 8, 3).mean(axis=(1, 3)) / spp
    assert np.abs(blk - ref / 16).mean() < 0.03
    img2, rays2 = dev.render_image(W, H, spp, 50, seed=42)
    assert rays2 == rays and np.array_equal(img, img2)


def test_reference_default_frame_400x225_100spp(dev, orc):
    """configs[0]'s frame on the device: the reference's own defaults (config.txt: 400 wide, aspect 16:9 -> 225 rows,
    100 rays per pixel, depth 50; Path Tracer.cpp:114-133, :199: 5 fixed + 250 random spheres). 225 = 28 x 8 + 1, so the
    bottom row of tiles holds one row of pixels (the compact job enumeration), and the frame is small enough for the
    end-of-frame hand-over to k_tail to matter. The first 8 samples per pixel against the oracle drawing the same
    philox numbers; all 100 against that oracle's rays per sample and luminance."""
    from path_tracer_amd import host
    sc = host.generate_scene(250, 1, 2)
    cam = host.camera_setup(aspect=1.7777777777)
    W, H, depth = 400, int(400 / 1.7777777777), 50
    assert H == 225 and len(sc["kind"]) == 255
    img8, ref8, rays8, ref_rays8, frac_off, med = frame_vs_oracle(dev, orc, sc, cam, W, H, 8, depth, key0=3, key1=5)
    print("[measured] 400x225x8 under identical numbers: median %.2e, pixels off by > 2e-3: %.4f, rays %d vs %d, mean ratio %.6f"
          % (med, frac_off, rays8, ref_rays8, img8.mean() / ref8.mean()))
    assert med < 1e-6 and frac_off < 0.01
    assert abs(rays8 - ref_rays8) < 0.005 * ref_rays8 and abs(img8.mean() / ref8.mean() - 1) < 0.005
    tight_bar("400x225x8: median", med, 5e-8); tight_bar("400x225x8: pixels off", frac_off, 0.0016)                 # measured 2.4e-8, 0.0008
    tight_bar("400x225x8: rays", abs(rays8 / ref_rays8 - 1), 1.3e-4); tight_bar("400x225x8: mean", abs(img8.mean() / ref8.mean() - 1), 3e-6)   # 6.5e-5, 1e-6
    assert np.abs(img8[H - 1] - ref8[H - 1]).max() / 8 < 0.2 and img8[H - 1].any()   # the one-row tiles at the bottom are filled
    img, rays = dev.render_image(W, H, 100, depth, seed=(5 << 32) | 3)
    assert np.isfinite(img).all() and img.min() >= 0 and img.max() <= 100 * 1.0001
    rps, rps_ref = rays / (W * H * 100), ref_rays8 / (W * H * 8)
    lum, lum_ref = img.mean() / 100, ref8.mean() / 8
    print("[measured] 400x225x100: rays per sample %.4f (oracle at 8 spp %.4f), luminance %.5f (%.5f)" % (rps, rps_ref, lum, lum_ref))
    assert abs(rps / rps_ref - 1) < 0.005 and abs(lum / lum_ref - 1) < 0.005  # north_star: mean luminance within 0.5 %
    tight_bar("400x225x100: rays per sample", abs(rps / rps_ref - 1), 1e-4); tight_bar("400x225x100: luminance", abs(lum / lum_ref - 1), 3e-4)   # measured 4e-5, 1.4e-4
    again, rays_again = dev.render_image(W, H, 100, depth, seed=(5 << 32) | 3)
    assert rays_again == rays and np.array_equal(again, img)                 # whatever the scheduling and the hand-over


def test_4k_frame_split_across_sample_ranges(dev):
    """configs[2] at full width and height (3840x2160; 32 spp instead of 1024, the properties do not depend on it):
    the frame rendered as 8 disjoint sample ranges (what 8 GPUs do) and summed equals the frame rendered in one
    call up to float addition order, ray counts add up exactly, and the image statistics are those of the scene."""
    from path_tracer_amd import host
    sc = host.generate_scene(500, 1, 2)
    use(dev, sc, host.camera_setup())
    W, H, spp = 3840, 2160, 32
    whole, rays = dev.render_image(W, H, spp, 50, seed=42)
    assert np.isfinite(whole).all() and whole.min() >= 0 and whole.max() <= spp * 1.0001
    assert 2.3 < rays / (W * H * spp) < 3.0
    parts = np.zeros_like(whole, dtype=np.float64)
    total = 0
    for g in range(8):
        img, r = dev.render_image(W, H, spp // 8, 50, seed=42, spp_begin=g * (spp // 8))
        parts += img
        total += r
    assert total == rays
    assert np.abs(parts - whole).max() <= 1e-5 * spp
    # a 1080p frame of the same scene and seed has the same mean luminance (different pixel grid, same integrand)
    small, _ = dev.render_image(1920, 1080, spp, 50, seed=42)
    assert abs(float(whole.mean()) - float(small.mean())) < 0.003 * float(small.mean())


def test_large_scene_10k_spheres(dev, orc):
    """configs[4]: 10,000 random spheres on a half extent of 45; scan and a small frame."""
    from path_tracer_amd import host
    sc = host.generate_scene(10000, 1, 2, half_extent=45.0)
    assert len(sc["kind"]) > 9000
    cam = host.camera_setup()
    use(dev, sc, cam)
    hs = orc.scene_create(sc)
    rng = np.random.default_rng(4)
    n = 4000
    o = f32(np.stack([rng.uniform(-40, 40, n), rng.uniform(0.5, 4, n), rng.uniform(-40, 40, n)], axis=1))
    d = rng.normal(size=(n, 3)); d[:, 1] = -np.abs(d[:, 1]) * 0.3
    d = f32(d)
    check_scan(dev.scan(o, d), orc.scan(hs, o, d), allow_mismatch=2, name="9,912 spheres, 4k rays", t_tol=8.4e-7, p_tol=8.8e-7, n_tol=2e-5)
    img, rays = dev.render_image(64, 36, 2, 50, seed=(2 << 32) | 1)
    ref, _, ref_rays = orc.render(hs, cam, 64, 36, 0, 2, 50, mode=1, seed_a=1, seed_b=2, threads=os.cpu_count() or 1)
    orc.scene_free(hs)
    diff = np.abs(img - ref).max(axis=2) / 2
    print("[measured] 9,912-sphere frame 64x36x2: pixels off by > 2e-3: %.4f, rays %d vs %d" % ((diff > 2e-3).mean(), rays, ref_rays))
    assert (diff > 2e-3).mean() < 0.02 and abs(rays - ref_rays) < 0.01 * ref_rays
    tight_bar("9,912-sphere frame: pixels off", (diff > 2e-3).mean(), 0.0044); tight_bar("9,912-sphere frame: rays", abs(rays / ref_rays - 1), 3.5e-3)   # measured 0.0022, 22 of 12,604


def test_output_stage_on_device_is_byte_exact(dev, orc, scene250):
    """gamma + clamp + quantise on the device give the bytes the reference's writeColour
    prints (Path Tracer.cpp:62-74), checked against the ORACLE's writer (orc_format_ppm) and the host-side writer on
    the same sums."""
    from path_tracer_amd import host
    sc, cam, hs = scene250
    use(dev, sc, cam)
    W, H, spp = 200, 113, 6
    img, _ = dev.render_image(W, H, spp, 50, seed=3)
    q = dev.quantise_rgb8(W, H, spp)
    lines = host.format_ppm(img, spp).decode().split("\n")
    ref = np.array([[int(v) for v in ln.split()] for ln in lines[3:3 + W * H]], dtype=np.uint8).reshape(H, W, 3)
    assert np.array_equal(q, ref)
    o_lines = orc.format_ppm(img.astype(np.float64), spp).decode().split("\n")
    assert o_lines[:3] == ["P3", "%d %d" % (W, H), "255"] and o_lines[:3] == lines[:3]
    o_ref = np.array([[int(v) for v in ln.split()] for ln in o_lines[3:3 + W * H]], dtype=np.uint8).reshape(H, W, 3)
    assert np.array_equal(q, o_ref)                                   # device bytes = oracle bytes
    assert 40 < q.mean() < 250 and q.std() > 10                       # a real picture, not a constant
    # extremes: zero, exactly one per sample, far above one
    sums = np.zeros((4, 8, 3), dtype=np.float32)
    sums[0, 1] = spp
    sums[0, 2] = 100 * spp
    sums[0, 3] = 0.25 * spp
    dev.write_accum(sums)
    q = dev.quantise_rgb8(8, 4, spp)
    assert list(q[0, :4, 0]) == [0, 255, 255, 128]


def test_resume_from_saved_sums(dev, orc, scene250):
    """a frame continued from a saved buffer equals the frame rendered in one go, and both are the
    oracle's frame for the same philox numbers (samples 0..8)."""
    from path_tracer_amd import Device
    sc, cam, hs = scene250
    use(dev, sc, cam)
    W, H = 96, 54
    dev.clear_accum(W, H)
    dev.render(W, H, 0, 5, 50, 11)
    dev.render(W, H, 5, 9, 50, 11)
    whole = dev.read_accum(W, H)
    dev.clear_accum(W, H)
    dev.render(W, H, 0, 5, 50, 11)
    saved = dev.read_accum(W, H)
    other = Device(0)
    other.set_scene(sc)
    other.set_camera(cam)
    other.write_accum(saved)
    other.render(W, H, 5, 9, 50, 11)
    resumed = other.read_accum(W, H)
    other.close()
    assert np.array_equal(resumed, whole)
    ref, _, _ = orc.render(hs, cam, W, H, 0, 9, 50, mode=1, seed_a=11, seed_b=0, threads=4)
    diff = np.abs(resumed.astype(np.float64) - ref).max(axis=2) / 9
    assert np.median(diff) < 1e-6 and (diff > 2e-3).mean() < 0.01 and abs(resumed.mean() / ref.mean() - 1) < 0.005
