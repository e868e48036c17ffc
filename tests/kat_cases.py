"""The hand-derived known answers of tests/golden/kat.json that were added in round 2, as inputs for the scatter /
ray-generation / camera / scene entry points. Shared by the CPU tests (the oracle) and the GPU tests (the device):
the same vectors must hold on both."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load():
    with open(os.path.join(GOLDEN, "kat.json")) as f:
        return json.load(f)


def scatter_cases(kat):
    """[(name, kind, albedo, fuzz, ior, d_in, n, front, u(3), expected d_out, expected ok)]"""
    out = []
    k = kat["dielectric_total_internal_reflection"]
    out.append(("tir", 2, [1, 1, 1], 0.0, k["ior"], k["d_in"], k["n"], k["front"], [0, 0, k["u3"]], k["d_out"], k["ok"]))
    k = kat["dielectric_oblique_refraction"]
    out.append(("oblique refraction", 2, [1, 1, 1], 0.0, k["ior"], k["d_in"], k["n"], k["front"], [0, 0, k["u3"]], k["d_out"], k["ok"]))
    k = kat["schlick_at_cos_half"]
    for which in ("coin_reflects", "coin_refracts"):
        out.append(("schlick 0.07, " + which, 2, [1, 1, 1], 0.0, k["ior"], k["d_in"], k["n"], k["front"], [0, 0, k[which]["u3"]],
                    k[which]["d_out"], 1))
    for i, k in enumerate(kat["metal_fuzz"]):
        out.append(("metal fuzz %d" % i, 1, k["albedo"], k["fuzz"], 0.0, k["d_in"], k["n"], 1, k["u"], k["d_out"], k["ok"]))
    k = kat["lambertian_degenerate_guard"]
    out.append(("lambertian guard", 0, k["albedo"], 0.0, 0.0, [0, 0, -1], k["n"], 1, k["u"], k["d_out"], k["ok"]))
    return out


def run_scatter_cases(scatter, kat, atol):
    """scatter: callable with the signature of Oracle.scatter / Device.scatter. Returns the largest deviation seen."""
    worst = 0.0
    for name, kind, albedo, fuzz, ior, d_in, n, front, u, d_out, ok in scatter_cases(kat):
        s = scatter([kind], [albedo], [fuzz], [ior], [d_in], [[0, 0, 0]], [n], [front], [u])
        err = float(np.abs(s["d"][0] - np.array(d_out, dtype=np.float64)).max())
        assert err <= atol, (name, s["d"][0], d_out)
        assert s["ok"][0] == ok, name
        want = [1, 1, 1] if kind == 2 else albedo                    # Dielectric.cpp:7, Lambertian.cpp:14, Metal.cpp:11
        assert np.allclose(s["atten"][0], want, atol=1e-7), name
        worst = max(worst, err)
    return worst


def canon(a, b):
    """libstdc++ generate_canonical<double, 53> from two mt19937 outputs: one rounding of a + b 2^32, then / 2^64."""
    return float(a + b * 2 ** 32) / 2.0 ** 64


def first_random_sphere(kat):
    """The first random sphere for seedA = seedB = 5489, evaluated from the published mt19937 outputs (kat.json)."""
    x = kat["first_random_sphere_seed_5489"]["mt19937_first_outputs"]
    u = [canon(x[2 * i], x[2 * i + 1]) for i in range(5)]
    uniform = lambda c, lo, hi: c * (hi - lo) + lo                   # libstdc++ uniform_real_distribution::operator()
    return dict(selector=uniform(u[0], 0.0, 1.0), r=uniform(u[1], 0.1, 0.4), cx=uniform(u[2], -10.0, 10.0), cz=uniform(u[3], -10.0, 10.0),
                albedo=[uniform(u[0], 0.0, 1.0), uniform(u[1], 0.0, 1.0), uniform(u[2], 0.0, 1.0)])


def path_scene(case):
    """The spheres of a whole-path case as the scene dict Device.set_scene / Oracle.scene_create take."""
    sp = case["spheres"]
    sc = {k: np.zeros(len(sp)) for k in ("cx", "cy", "cz", "r", "ar", "ag", "ab", "fuzz", "ior")}
    sc["kind"] = np.zeros(len(sp), dtype=np.int32)
    for i, s in enumerate(sp):
        sc["cx"][i], sc["cy"][i], sc["cz"][i] = s["c"]
        sc["r"][i] = s["r"]
        sc["kind"][i] = s["kind"]
        a = s.get("albedo", [1, 1, 1])
        sc["ar"][i], sc["ag"][i], sc["ab"][i] = a
        sc["fuzz"][i] = s.get("fuzz", 0.0)
        sc["ior"][i] = s.get("ior", 0.0)
    return sc


def run_path_cases(trace_one, kat, atol):
    """trace_one(scene dict, o[1][3], d[1][3], max_depth, u[1][max_depth][3]) -> dict(rgb=[1][3], rays=[1]).
    Asserts colour and ray count of every hand-derived whole path (kat.json: whole_paths). Returns the largest deviation."""
    worst = 0.0
    for case in kat["whole_paths"]:
        md = case["max_depth"]
        u = np.zeros((1, max(md, 0), 3))
        for k, v in enumerate(case["u"]):
            u[0, k] = v
        got = trace_one(path_scene(case), [case["o"]], [case["d"]], md, u)
        err = float(np.abs(np.asarray(got["rgb"][0], dtype=np.float64) - np.array(case["rgb"], dtype=np.float64)).max())
        assert err <= atol, (case["name"], got["rgb"][0], case["rgb"])
        assert int(got["rays"][0]) == case["rays"], (case["name"], int(got["rays"][0]), case["rays"])
        worst = max(worst, err)
    return worst
