#!/usr/bin/env python3
"""bench.py — the sample loop on N MI355X GPUs of one node.

Workload (BASELINE.json configs[1]): 1920x1080, 100 spp, maxDepth 50, 5 fixed + 500 random
spheres (scene seeds 1/2, default camera, render seed 42), synthetic. One step = one frame:
every pixel gets its samples, the accumulation buffer is cleared first. With N > 1 every rank
renders its own disjoint 100-spp range of the same frame (weak scaling: N*100 spp in all) and
the float buffers are summed to rank 0 with an RCCL reduce inside the timed step.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the fields).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_TFLOPS = 157.3          # MI355X FP32 vector peak (MI355X_MICROARCH.md): 256 CU x 2.4 GHz x 256 flop/clk
FLOP_PER_TEST = 17                # one ray-sphere test through the discriminant
FLOP_PER_RAY_FIXED = 60


def cpu_baseline(n_spheres_random, threads):
    """The oracle's faithful Mersenne-Twister render (shared RNG-B under a mutex, one task per row,
    like the reference's ThreadPool path) on a bounded sample of the same scene and camera."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import Oracle
    orc = Oracle()
    sc = orc.generate_scene(n_spheres_random, 1, 2)
    cam = orc.camera_setup()
    hs = orc.scene_create(sc)
    w, h = 960, 540
    spp = 1
    t0 = time.time()
    orc.render(hs, cam, w, h, 0, spp, 50, mode=0, seed_a=11, seed_b=12, threads=threads, shared_b=1)
    dt = time.time() - t0
    total = w * h * spp
    # extend towards ~12 s of CPU work, at most 16 spp
    more = int(min(15, max(0, 12.0 / max(dt, 1e-3) - 1)))
    if more > 0:
        t1 = time.time()
        orc.render(hs, cam, w, h, spp, spp + more, 50, mode=0, seed_a=11, seed_b=12, threads=threads, shared_b=1)
        dt += time.time() - t1
        total += w * h * more
        spp += more
    orc.scene_free(hs)
    return {"value": total / dt / 1e6, "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": "%dx%d, %d spp, maxDepth 50, same %d-sphere scene and camera; oracle restatement with the "
                      "reference's shared mutex-guarded RNG-B and one task per row, %d threads, %.1f s"
                      % (w, h, spp, n_spheres_random + 5, threads, dt)}


def measured_traffic(default_workload):
    """HBM bytes per k_render launch from the newest committed rocprofv3 counter summary (profiles/*_summary.json:
    FETCH_SIZE and WRITE_SIZE collected in separate --pmc passes, FETCH doubled as the gfx950 correction). Only
    valid for the default workload, on which it was measured; the kernel is VALU-bound and these bytes are the
    fixed-point job sums (atomics) plus a few KB of scene data."""
    if not default_workload:
        return None
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_summary.json")))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            return float(json.load(f)["k_render_hbm_bytes_per_launch"]["total"])
    except (KeyError, ValueError, OSError):
        return None


def extras(dev, host, accum, W, H, spp, depth, scene, cam):
    """Outside the timed region and outside the roofline figure: the same workload with the optional grid search
    (pt_set_accel), and configs[4] (10k spheres, 256 spp) once with each search."""
    import torch

    def kernel_ms(reps, s):
        dev.kernel_times()
        for _ in range(reps):
            accum.zero_()
            dev.render(W, H, 0, s, depth, 42, accum_ptr=accum.data_ptr())
        torch.cuda.synchronize()
        return float(np.min(dev.kernel_times()))

    res = {"note": "kernel times (HIP events), best of the runs; not part of `value` or `roofline`"}
    dev.set_accel(1)
    kernel_ms(1, spp)
    ms = kernel_ms(3, spp)
    res["grid_search_same_workload"] = {"kernel_ms": ms, "Msamples_per_s": W * H * spp / ms / 1e3, "grid": dev.accel_info()}
    big = host.generate_scene(10000, 1, 2, half_extent=45.0)
    dev.set_scene(big)
    ms_grid = kernel_ms(2, 256)
    info = dev.accel_info()
    dev.set_accel(0)
    ms_scan = kernel_ms(1, 256)
    res["configs4_10k_spheres_256spp"] = {"spheres": len(big["kind"]), "full_scan_kernel_ms": ms_scan, "grid_kernel_ms": ms_grid,
                                          "full_scan_Msamples_per_s": W * H * 256 / ms_scan / 1e3,
                                          "grid_Msamples_per_s": W * H * 256 / ms_grid / 1e3, "grid": info}
    dev.set_scene(scene)
    dev.set_camera(cam)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=100, help="samples per pixel per GPU per step")
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--spheres", type=int, default=500)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the grid-search and 10k-sphere lines (N=1 only)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="multi-rank control flow on ONE device: every rank renders on cuda:0 and the reduce goes "
                         "through gloo on host copies (RCCL refuses two ranks on one GPU); not a measurement")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from path_tracer_amd import distributed as ptd
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the sample loop has no CPU path", file=sys.stderr)
        sys.exit(2)
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    ptd.init("gloo" if args.rehearse_on_one_gpu else "nccl")   # "nccl" is RCCL; a single process needs no group

    from path_tracer_amd import Device, host

    W, H, spp, depth = args.width, args.height, args.spp, args.depth
    scene = host.generate_scene(args.spheres, 1, 2)
    n_spheres = len(scene["kind"])
    cam = host.camera_setup(aspect=W / H)
    dev = Device(local_rank)
    dev.set_scene(scene)
    dev.set_camera(cam)
    dev.set_stream(torch.cuda.current_stream().cuda_stream)
    accum = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    spp_begin, spp_end = ptd.weak_range(rank, spp)        # disjoint ranges of one frame

    def step():
        accum.zero_()
        dev.render(W, H, spp_begin, spp_end, depth, 42, accum_ptr=accum.data_ptr())
        if args.rehearse_on_one_gpu and world > 1:
            host_copy = accum.cpu()
            ptd.reduce_frame(host_copy, dst=0)
            accum.copy_(host_copy)
        else:
            ptd.reduce_frame(accum, dst=0)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    dev.kernel_times()                       # drop the warm-up records
    rays0 = dev.ray_count()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    rays = dev.ray_count() - rays0
    ktimes = dev.kernel_times()

    if world > 1:
        stat_dev = "cpu" if args.rehearse_on_one_gpu else "cuda"
        t = torch.tensor([dt], dtype=torch.float64, device=stat_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        r = torch.tensor([rays], dtype=torch.int64, device=stat_dev)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        total_rays = int(r.item())
    else:
        total_rays = rays

    if rank == 0:
        samples = float(W) * H * spp * world * args.steps
        k_ms = float(np.mean(ktimes)) if len(ktimes) else float("nan")
        rays_per_launch = rays / max(args.steps, 1)                       # this rank's kernel
        flops_per_launch = rays_per_launch * (FLOP_PER_TEST * n_spheres + FLOP_PER_RAY_FIXED)
        achieved = flops_per_launch / (k_ms * 1e-3) / 1e12
        out = {
            "metric": "Msamples/s at 1080p/500 spheres",
            "value": samples / dt / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "configs[1]: %dx%d, %d spp per GPU, maxDepth %d, %d spheres (5 fixed + %d random, "
                                   "scene seeds 1/2), default camera, render seed 42" % (W, H, spp, depth, n_spheres, args.spheres),
                       "width": W, "height": H, "spp_per_gpu": spp, "max_depth": depth, "spheres": n_spheres,
                       "parallelism": "spp-split x%d + RCCL reduce" % world if world > 1 else "single GPU"},
            "rays_per_s": total_rays / dt,
            "rays_per_sample": total_rays / samples,
            "roofline": {"bound": "valu_fp32", "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_TFLOPS,
                         "traffic": measured_traffic((W, H, spp, depth, args.spheres) == (1920, 1080, 100, 50, 500)),
                         "kernel": "k_render<0>", "kernel_ms": k_ms,
                         "flops_per_launch": flops_per_launch,
                         "note": "algorithmic flops = rays x (17 x spheres + 60); rays counted by the kernel"},
        }
        if world == 1 and not args.no_extras:
            try:
                out["extras"] = extras(dev, host, accum, W, H, spp, depth, scene, cam)
            except Exception as e:                                   # never lose the headline line to an extra
                out["extras"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args.spheres, min(16, os.cpu_count() or 1))
            except Exception as e:                                   # e.g. oracle/_build missing: the headline line still prints
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
