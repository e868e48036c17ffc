"""Diagnostic: the split between the sample-loop kernel and the end-of-frame kernel (pt_kernel_times_split) for the
400x225 frame, configs[1] and configs[3], best and median of REPS launches."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from path_tracer_amd import Device, host
reps = int(os.environ.get("REPS", "30"))
dev = Device(0)
for name, W, H, n, spp, depth, dif, met in (("400x225", 400, 225, 250, 100, 50, 0.6, 0.9), ("configs1", 1920, 1080, 500, 100, 50, 0.6, 0.9),
                                            ("configs3", 1920, 1080, 500, 100, 100, 0.1, 0.5)):
    dev.set_scene(host.generate_scene(n, 1, 2, diffuse_below=dif, metal_below=met)); dev.set_camera(host.camera_setup(aspect=W / H))
    dev.clear_accum(W, H)
    r = reps if W < 1000 else max(3, reps // 8)
    for _ in range(r): dev.render(W, H, 0, spp, depth, 42)
    dev.kernel_times()
    for _ in range(r): dev.render(W, H, 0, spp, depth, 42)
    whole, tail = dev.kernel_times_split()
    print("%-9s whole best %.3f median %.3f ms | k_tail best %.3f median %.3f ms | k_render median %.3f" % (name, whole.min(), np.median(whole), tail.min(), np.median(tail), np.median(whole - tail)))
