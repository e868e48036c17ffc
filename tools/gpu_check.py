"""Scratch GPU check: HIP path vs oracle on scan / scatter / gen_ray / image. Not a test."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from oracle_lib import Oracle
from path_tracer_amd import Device

orc = Oracle(); dev = Device(0)
sc = orc.generate_scene(500, 1, 2); cam = orc.camera_setup()
dev.set_scene(sc); dev.set_camera(cam); hs = orc.scene_create(sc)
print("philox", dev.philox([[0,0,0,0],[0x243f6a88,0x85a308d3,0x13198a2e,0x03707344]], [0,0]))
rng = np.random.default_rng(0)
n = 20000
W, H = 1920, 1080
pi = rng.integers(0, W, n); pj = rng.integers(0, H, n)
u1 = rng.random(n).astype(np.float32); u2 = rng.random(n).astype(np.float32)
ang = rng.random(n); rad = np.sqrt(rng.random(n))
dx = (rad*np.cos(2*np.pi*ang)).astype(np.float32); dy = (rad*np.sin(2*np.pi*ang)).astype(np.float32)
og, dg = dev.gen_ray(W, H, pi, pj, u1, u2, dx, dy)
oo, do = orc.gen_ray(cam, W, H, pi, pj, u1, u2, dx, dy)
print("gen_ray max rel err o", np.abs(og-oo).max()/np.abs(oo).max(), "d", np.abs(dg-do).max()/np.abs(do).max())
# scan: camera rays + random rays from points above ground
o = og.astype(np.float32).astype(np.float64); d = dg.astype(np.float32).astype(np.float64)
g = dev.scan(o, d); r = orc.scan(hs, o, d)
same = g["idx"] == r["idx"]
print("scan camera rays: idx mismatches", (~same).sum(), "of", n, "hits", (r["idx"]>=0).sum())
hit = same & (r["idx"]>=0)
print("  t rel err max", (np.abs(g["t"]-r["t"])[hit]/r["t"][hit]).max(), "n err", np.abs(g["n"]-r["n"])[hit].max(), "p err", np.abs(g["p"]-r["p"])[hit].max(), "front mism", (g["front"]!=r["front"])[hit].sum())
# secondary rays: start at hit points, random directions
p0 = r["p"][hit][:8000].astype(np.float32).astype(np.float64)
d2 = rng.normal(size=p0.shape); d2 *= rng.uniform(0.1, 2.0, size=(len(p0),1)); d2 = d2.astype(np.float32).astype(np.float64)
p0 = p0 + 0.01*r["n"][hit][:8000]; p0 = p0.astype(np.float32).astype(np.float64)
g = dev.scan(p0, d2); r2 = orc.scan(hs, p0, d2)
same = g["idx"] == r2["idx"]; hit2 = same & (r2["idx"]>=0)
print("scan secondary: idx mismatches", (~same).sum(), "of", len(p0), "hits", (r2["idx"]>=0).sum())
print("  t rel err max", (np.abs(g["t"]-r2["t"])[hit2]/r2["t"][hit2]).max(), "n err", np.abs(g["n"]-r2["n"])[hit2].max())
# image parity with identical philox uniforms
W2, H2, spp = 160, 90, 8
t0 = time.time(); img, rays = dev.render_image(W2, H2, spp, 50, seed=(2 << 32) | 1); print("gpu render", time.time()-t0, "rays", rays)
t0 = time.time(); ref, _, rrays = orc.render(hs, cam, W2, H2, 0, spp, 50, mode=1, seed_a=1, seed_b=2, threads=8); print("oracle render", time.time()-t0, "rays", rrays)
diff = np.abs(img/spp - ref/spp)
print("image: mean", img.mean()/spp, ref.mean()/spp, "max diff", diff.max(), "mean diff", diff.mean(), "frac pixels > 1e-3", (diff.max(axis=2) > 1e-3).mean())
# bigger timing
for (w, h, s) in [(1920, 1080, 10), (1920, 1080, 100)]:
    dev.clear_accum(w, h); dev.synchronize()
    t0 = time.time(); rays = dev.render(w, h, 0, s, 50, 42, want_rays=True); dt = time.time()-t0
    kt = dev.kernel_times()
    fl = rays*(17*505+60)
    print(f"{w}x{h}x{s}: {dt*1e3:.1f} ms kernel {kt} ms  {w*h*s/dt/1e6:.1f} Msamples/s rays {rays} rays/sample {rays/(w*h*s):.3f}  TFLOP/s {fl/ (kt[-1]*1e-3)/1e12:.2f} frac {fl/(kt[-1]*1e-3)/157.3e12:.3f}")
