"""Diagnostic: kernel time of configs[1] and a hash of a small frame (frames are bit-reproducible, so the
hash must not move when only scheduling changes)."""
import hashlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from path_tracer_amd import Device, host
n = int(os.environ.get("SPHERES", "500")); spp = int(os.environ.get("SPP", "100")); reps = int(os.environ.get("REPS", "3"))
W, H = int(os.environ.get("W", "1920")), int(os.environ.get("H", "1080"))
sc = host.generate_scene(n, 1, 2, half_extent=float(os.environ.get("EXTENT", "10")), diffuse_below=float(os.environ.get("DIFFUSE", "0.6")), metal_below=float(os.environ.get("METAL", "0.9"))); cam = host.camera_setup(aspect=W / H)
dev = Device(0); dev.set_accel(int(os.environ.get("ACCEL", "0"))); dev.set_scene(sc); dev.set_camera(cam)
if os.environ.get("ACCEL", "0") != "0": print("grid", dev.accel_info())
img, rays = dev.render_image(200, 113, 16, 50, seed=7)
print("hash", hashlib.sha1(img.tobytes()).hexdigest()[:16], "rays", rays, "mean", float(img.mean()) / 16)
dev.kernel_times()
dev.clear_accum(W, H)
for _ in range(reps):
    rays = dev.render(W, H, 0, spp, int(os.environ.get("DEPTH", "50")), 42, want_rays=True)
kt = dev.kernel_times()
ns = len(sc["kind"])
fl = rays * (17 * ns + 60)
print("spheres %d  %dx%d x %d spp: kernel ms %s  best %.2f  Msamples/s %.0f  TFLOP/s %.1f  frac %.3f  rays/sample %.3f"
      % (ns, W, H, spp, np.round(kt, 2), kt.min(), W * H * spp / kt.min() / 1e3, fl / kt.min() / 1e9, fl / kt.min() / 1e9 / 157.3, rays / (W * H * spp)))
