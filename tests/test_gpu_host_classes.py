"""GPU (-m gpu): the reference's class surface (Hittable::isHit, Material::isScattered,
Camera::getCurrentRay, and the whole frame) as kept by the host-side C++, answered by the device."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def kat():
    with open(os.path.join(GOLDEN, "kat.json")) as f:
        return json.load(f)


def test_sphere_is_hit(kat):
    from path_tracer_amd import host
    for name in ("sphere_hit", "sphere_hit_inside"):
        k = kat[name]
        r = host.sphere_is_hit(k["sphere"]["c"], k["sphere"]["r"], k["ray"]["o"], k["ray"]["d"])
        assert r is not None and r["t"] == pytest.approx(k["t"], rel=1e-6) and r["front"] == k["front"]
        assert np.allclose(r["p"], k["p"], atol=1e-5) and np.allclose(r["n"], k["n"], atol=1e-6)
    assert host.sphere_is_hit((0, 0, -5), 1.0, (0, 0, 0), (0, 1, 0)) is None
    # t_max below the hit distance: no hit (Sphere.cpp:18-21)
    assert host.sphere_is_hit((0, 0, -5), 1.0, (0, 0, 0), (0, 0, -1), 0.001, 3.0) is None
    assert host.sphere_is_hit((0, 0, -5), 1.0, (0, 0, 0), (0, 0, -1), 0.001, 4.5) is not None


def test_world_is_hit_matches_oracle(orc):
    from path_tracer_amd import host
    sc = host.generate_scene(60, 1, 2)
    hs = orc.scene_create(sc)
    rng = np.random.default_rng(0)
    for _ in range(24):
        o = np.array([8.0, 2.0, 3.0])
        d = np.float32(rng.normal(size=3) * 2).astype(np.float64)
        ref = orc.scan(hs, [o], [d])
        got = host.world_is_hit(sc, o, d)
        if ref["idx"][0] < 0:
            assert got is None
        else:
            assert got is not None and got["t"] == pytest.approx(ref["t"][0], rel=1e-4)
            assert np.allclose(got["n"], ref["n"][0], atol=1e-4) and got["front"] == ref["front"][0]
            assert got["kind"] == sc["kind"][ref["idx"][0]]
    orc.scene_free(hs)


def test_material_scatter_and_camera_ray(orc, kat):
    from path_tracer_amd import host
    k = kat["lambertian"]
    s = host.material_scatter(0, k["albedo"], 0.0, 0.0, (0, -1, 0), (1, 2, 3), k["n"], 1, k["u"])
    assert s["ok"] == 1 and np.allclose(s["d"], k["d_out"], atol=1e-6) and np.allclose(s["atten"], k["atten"], atol=1e-7)
    k = kat["dielectric_quirk"]
    s = host.material_scatter(2, (1, 1, 1), 0.0, k["ior"], k["d_in"], (0, 0, 0), k["n"], k["front"], (0, 0, k["u3"]))
    assert s["ok"] == 1 and np.allclose(s["d"], k["d_out"], atol=1e-6)
    cam = orc.camera_setup()
    o, d = host.camera_ray(0.25, 0.75, 0.3, -0.2)
    # normalised coordinates (x, y): pixel 0 of a 2x2 image with jitter (x, y), (0 + x) / (2 - 1) = x
    oo, od = orc.gen_ray(cam, 2, 2, [0], [0], [0.25], [0.75], [np.float32(0.3)], [np.float32(-0.2)])
    assert np.allclose(o, oo[0], rtol=1e-5) and np.allclose(d, od[0], rtol=1e-5, atol=1e-5)


def test_whole_frame_through_the_host_classes(orc):
    """config text -> Settings -> Camera + scene -> DeviceRenderer -> P3 bytes, compared with the oracle's
    PPM of the same frame (same philox numbers): nearly every channel value identical."""
    from path_tracer_amd import host
    cfg = ("imageAspectRatio = 1.7777777777\nimageWidth = 64\nraysPerPixel = 16\nmaterialMaxDepth = 50\n"
           "cameraPosition = (8,2,3)\ncameraLookingAt = (0,0,0)\ncameraUpOrient = (0,1,0)\nfocalLength = 1\n"
           "verticalFOV = 60\napertureSize = 0.1\nnumberOfSpheres = 250\nsceneSeedA = 1\nsceneSeedB = 2\nrenderSeed = 8589934593\n")
    ppm, rate = host.render_ppm(cfg)
    lines = ppm.decode().split("\n")
    assert lines[0] == "P3" and lines[1] == "64 36" and lines[2] == "255" and rate > 0
    got = np.array([[int(v) for v in ln.split()] for ln in lines[3:3 + 64 * 36]])
    sc = orc.generate_scene(250, 1, 2)
    hs = orc.scene_create(sc)
    s, _, _ = orc.render(hs, orc.camera_setup(aspect=1.7777777777), 64, 36, 0, 16, 50, mode=1, seed_a=1, seed_b=2, threads=4)
    orc.scene_free(hs)
    ref_lines = orc.format_ppm(s, 16).decode().split("\n")
    ref = np.array([[int(v) for v in ln.split()] for ln in ref_lines[3:3 + 64 * 36]])
    assert got.shape == ref.shape
    assert (np.abs(got - ref) > 1).mean() < 0.03
    assert abs(got.mean() - ref.mean()) < 0.5
    # the optional key of this build: the same frame with the grid search
    ppm2, _ = host.render_ppm(cfg + "closestHitSearch = grid\n")
    got2 = np.array([[int(v) for v in ln.split()] for ln in ppm2.decode().split("\n")[3:3 + 64 * 36]])
    assert (np.abs(got2 - got) > 1).mean() < 0.01
