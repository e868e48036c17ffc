"""Diagnostic: run one configs[1] frame with the PT_STATS build (PT_HIP_LIB=tools/build/libpt_hip_stats.so)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from path_tracer_amd import Device, host
sc = host.generate_scene(int(os.environ.get("SPHERES", "500")), 1, 2, half_extent=float(os.environ.get("EXTENT", "10"))); cam = host.camera_setup()
dev = Device(0); dev.set_accel(int(os.environ.get("ACCEL", "0"))); dev.set_scene(sc); dev.set_camera(cam)
W, H, spp = 1920, 1080, int(os.environ.get("SPP", "100"))
dev.clear_accum(W, H)
rays = dev.render(W, H, 0, spp, 50, 42, want_rays=True)
print("rays", rays, "rays/64", rays / 64, "kernel ms", dev.kernel_times())
