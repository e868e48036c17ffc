"""Generates the fixtures in tests/golden/ from the CPU oracle (oracle/oracle.cpp).

PARITY UNPINNED: the reference holds no fixtures and cannot be built here, so these vectors are
outputs of the oracle RESTATEMENT, not of the reference. They freeze the oracle's behaviour at
the time of writing (g++ 11.4, libstdc++), so that a later change to the oracle or to the HIP
path shows up as a diff. kat.json is different: it is hand-derived from the reference's source
 and was not produced by any program.

Run: python3 tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle_lib import Oracle, build  # noqa: E402


def f32(a):
    """Round to float32 and back, so that the fp32 device path and the oracle see identical inputs."""
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def main():
    build()
    orc = Oracle()
    rng = np.random.default_rng(20261111)

    # T1: scenes at fixed seeds
    for count, a, b in [(250, 1, 2), (500, 1, 2), (250, 20260101, 77)]:
        sc = orc.generate_scene(count, a, b)
        np.savez_compressed(os.path.join(HERE, "scene_%d_%d_%d.npz" % (count, a, b)), **sc)

    # T2: camera constants, default and one non-default camera
    cams = np.stack([orc.camera_setup(),
                     orc.camera_setup(pos=(3, 5, 9), look=(1, 0.5, 0), up=(0, 1, 0), aspect=1.7777777777, focal=1.5,
                                      fov=40.0, aperture=0.3)])
    np.save(os.path.join(HERE, "camera.npy"), cams)

    sc = orc.generate_scene(250, 1, 2)
    hs = orc.scene_create(sc)
    cam = cams[0]

    # T5: ray generation
    n = 512
    W, H = 400, 225
    pi = rng.integers(0, W, n).astype(np.int32)
    pj = rng.integers(0, H, n).astype(np.int32)
    u1, u2 = f32(rng.random(n)), f32(rng.random(n))
    ang, rad = rng.random(n), np.sqrt(rng.random(n))
    dx, dy = f32(rad * np.cos(2 * np.pi * ang)), f32(rad * np.sin(2 * np.pi * ang))
    o, d = orc.gen_ray(cam, W, H, pi, pj, u1, u2, dx, dy)
    np.savez_compressed(os.path.join(HERE, "t5_gen_ray.npz"), W=W, H=H, pi=pi, pj=pj, u1=u1, u2=u2, diskx=dx, disky=dy, o=o, d=d)

    # T3: camera rays, rays leaving hit points, rays grazing the ground sphere
    ro, rd = f32(o), f32(d)
    hit0 = orc.scan(hs, ro, rd)
    sel = hit0["idx"] >= 0
    p0 = f32(hit0["p"][sel] + 0.01 * hit0["n"][sel])
    d2 = rng.normal(size=p0.shape) * rng.uniform(0.05, 2.0, size=(len(p0), 1))
    d2 = f32(d2)
    m = 256
    gx, gz = rng.uniform(-9, 9, m), rng.uniform(-9, 9, m)
    gy = rng.uniform(0.001, 0.5, m)
    gaz = rng.uniform(0, 2 * np.pi, m)
    slope = rng.uniform(-0.05, 0.0, m)                    # shallow, downward: grazing hits on r = 1000
    go = f32(np.stack([gx, gy, gz], axis=1))
    gd = f32(np.stack([np.cos(gaz), slope, np.sin(gaz)], axis=1) * rng.uniform(0.5, 9.0, size=(m, 1)))
    ao = np.concatenate([ro, p0, go])
    ad = np.concatenate([rd, d2, gd])
    r = orc.scan(hs, ao, ad)
    np.savez_compressed(os.path.join(HERE, "t3_scan.npz"), o=ao, d=ad, **{"r_" + k: v for k, v in r.items()})

    # T4: scatter, every material, injected uniforms
    n = 768
    kind = (np.arange(n) % 3).astype(np.int32)
    albedo = rng.random((n, 3))
    fuzz = np.where(kind == 1, rng.random(n) * 0.5, 0.0)
    fuzz[kind == 1][:8] = 0.0
    ior = np.where(kind == 2, 1.5, 0.0)
    nrm = rng.normal(size=(n, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    d_in = rng.normal(size=(n, 3))
    d_in -= nrm * np.maximum(0, (d_in * nrm).sum(1, keepdims=True)) * 2     # make the normal oppose the ray
    d_in *= rng.choice([0.3, 1.0, 1.7, 9.0], size=(n, 1))                    # |d| matters (Dielectric.cpp:19)
    d_in, nrm = f32(d_in), f32(nrm)
    keep = (d_in * nrm).sum(1) < -1e-3
    front = rng.integers(0, 2, n).astype(np.int32)
    p = f32(rng.normal(size=(n, 3)))
    u = f32(rng.random((n, 3)))
    albedo, fuzz = f32(albedo), f32(fuzz)
    sel = np.nonzero(keep)[0]
    args = [a[sel] for a in (kind, albedo, fuzz, ior, d_in, p, nrm, front, u)]
    s = orc.scatter(*args)
    np.savez_compressed(os.path.join(HERE, "t4_scatter.npz"), kind=args[0], albedo=args[1], fuzz=args[2], ior=args[3],
                        d_in=args[4], p=args[5], nrm=args[6], front=args[7], u=args[8],
                        **{"r_" + k: v for k, v in s.items()})

    # T6: converged low-resolution render with the reference's Mersenne-Twister streams + per-pixel variance
    W, H, spp = 160, 90, 2048
    ssum, ssq, rays = orc.render(hs, cam, W, H, 0, spp, 50, mode=0, seed_a=101, seed_b=202, threads=os.cpu_count() or 1,
                                 shared_b=0, want_sumsq=True)
    mean = ssum / spp
    var = np.maximum(ssq / spp - mean * mean, 0.0)
    np.savez_compressed(os.path.join(HERE, "t6_image_160x90.npz"), mean=mean.astype(np.float32), var=var.astype(np.float32),
                        spp=spp, rays_per_sample=rays / (W * H * spp))

    # T6b: the same frame with philox numbering (what the GPU draws), 8 spp, for a near-exact comparison
    W, H, spp = 96, 54, 8
    psum, _, prays = orc.render(hs, cam, W, H, 0, spp, 50, mode=1, seed_a=1, seed_b=2, threads=os.cpu_count() or 1)
    np.savez_compressed(os.path.join(HERE, "t6b_philox_96x54.npz"), sum=psum, spp=spp, rays=prays, key0=1, key1=2)

    # T7: PPM bytes of a small accumulation buffer
    W, H, spp = 16, 9, 8
    s7, _, _ = orc.render(hs, cam, W, H, 0, spp, 50, mode=1, seed_a=5, seed_b=6, threads=1)
    s7 = s7.astype(np.float32).astype(np.float64)
    ppm = orc.format_ppm(s7, spp)
    np.savez_compressed(os.path.join(HERE, "t7_ppm.npz"), sum=s7.astype(np.float32), spp=spp, ppm=np.frombuffer(ppm, dtype=np.uint8))
    orc.scene_free(hs)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
