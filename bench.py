#!/usr/bin/env python3
"""bench.py — the sample loop on N MI355X GPUs of one node.

Default workload (BASELINE.json configs[1]): 1920x1080, 100 spp, maxDepth 50, 5 fixed + 500 random
spheres (scene seeds 1/2, default camera, render seed 42), synthetic. One step = one frame:
every pixel gets its samples, the accumulation buffer is cleared first. With N > 1 every rank
renders a disjoint sample range of the SAME frame and the float buffers are summed to rank 0 with
an RCCL reduce inside the timed step:
  --scaling weak   (default) every rank renders `spp` samples per pixel: N x spp in all;
  --scaling strong the frame's `spp` samples per pixel are cut into N near-equal ranges.
--workload configs1 (default) | configs2 (3840x2160, 1024 spp, 505 spheres) | configs4 (1920x1080,
256 spp, 10,000 random spheres on a half extent of 45) sets the sizes BASELINE.json names;
--width/--height/--spp/--depth/--spheres override single values.

Launching N ranks: either let this script start them,
    python3 bench.py --gpus 8 [--scaling strong --workload configs2]
(the parent process starts N children, one per GPU, BEFORE it touches the GPU, and prints rank 0's line),
or use an outside launcher that sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*:
    python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \\
        bench.py --gpus 8

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the fields).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_TFLOPS = 157.3          # MI355X FP32 vector peak (MI355X_MICROARCH.md): 256 CU x 2.4 GHz x 256 flop/clk
FLOP_PER_TEST = 17                # one ray-sphere test through the discriminant
FLOP_PER_RAY_FIXED = 60

WORKLOADS = {
    # name: (label, width, height, spp, depth, random spheres, half extent)
    "configs1": ("configs[1]", 1920, 1080, 100, 50, 500, 10.0),
    "configs2": ("configs[2]", 3840, 2160, 1024, 50, 500, 10.0),
    "configs4": ("configs[4]", 1920, 1080, 256, 50, 10000, 45.0),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="configs1")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: --spp samples per pixel PER GPU; strong: --spp samples per pixel in all, split over the GPUs")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None, help="samples per pixel (per GPU when weak, per frame when strong)")
    ap.add_argument("--depth", type=int, default=None)
    ap.add_argument("--spheres", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the grid-search and 10k-sphere lines (N=1 only)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="multi-rank control flow on ONE device: every rank renders on cuda:0 and the reduce goes "
                         "through gloo on host copies (RCCL refuses two ranks on one GPU); not a measurement")
    ap.add_argument("--rehearse-without-gpu", action="store_true",
                    help="launch, rendezvous, reduce, timing and reporting only: no rank renders anything (zero-filled "
                         "host buffers, gloo), `value` is null; for checking the launch path on a machine without a GPU")
    args = ap.parse_args(argv)
    label, w, h, spp, depth, spheres, half = WORKLOADS[args.workload]
    args.label = label
    args.custom = any(v is not None for v in (args.width, args.height, args.spp, args.depth, args.spheres))
    args.width = w if args.width is None else args.width
    args.height = h if args.height is None else args.height
    args.spp = spp if args.spp is None else args.spp
    args.depth = depth if args.depth is None else args.depth
    args.spheres = spheres if args.spheres is None else args.spheres
    args.half_extent = half
    if args.gpus < 1 or args.steps < 1 or args.warmup < 0 or args.spp < 1:
        ap.error("--gpus, --steps, --spp must be positive and --warmup non-negative")
    return args


# ---------------------------------------------------------------------------------------------
# Parent: start one child per GPU. Nothing here may touch the GPU (no torch import, no Device).
# ---------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n, argv):
    """Starts n fresh copies of this script, one rank each, and relays rank 0's output. A process that has used
    the GPU is never replaced or re-executed: the parent only waits."""
    port = _free_port()
    procs = []
    for rank in range(n):
        env = dict(os.environ)
        env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PT_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL between processes needs it on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rc = procs[0].returncode
    deadline = time.time() + 120
    for p in procs[1:]:
        try:
            p.wait(timeout=max(1.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            p.kill()                                             # the exact child started above
            p.wait()
        rc = rc or p.returncode
    # rank 0's JSON line goes to stdout; anything else a library printed there (gloo's connection notes) to stderr
    for line in out.decode(errors="replace").splitlines():
        is_json = line.startswith("{") and line.rstrip().endswith("}")
        print(line, file=sys.stdout if is_json else sys.stderr, flush=True)
    return rc


# ---------------------------------------------------------------------------------------------
def cpu_baseline(threads):
    """The oracle's faithful Mersenne-Twister render (shared RNG-B under a mutex, one task per row, like the
    reference's ThreadPool path), timed on this machine's host cores: configs[0] in full (the reference's default
    400x225 x 100 spp frame of 5 + up to 250 spheres) and a bounded sample of configs[1]'s scene."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import Oracle
    orc = Oracle()
    cam = orc.camera_setup()
    res = {"unit": "Msamples/s", "cores": threads, "kind": "port",
           "note": "the port has the reference's locks (shared mutex-guarded RNG-B) and task shape (one per row) but not its "
                   "shared_ptr copies and virtual calls, so it is somewhat cheaper per ray than the reference itself"}

    # configs[1]'s scene, bounded: 960x540 at as many spp as fit in about 12 s
    sc = orc.generate_scene(500, 1, 2)
    hs = orc.scene_create(sc)
    w, h, spp = 960, 540, 1
    t0 = time.time()
    orc.render(hs, cam, w, h, 0, spp, 50, mode=0, seed_a=11, seed_b=12, threads=threads, shared_b=1)
    dt = time.time() - t0
    total = w * h * spp
    more = int(min(15, max(0, 12.0 / max(dt, 1e-3) - 1)))
    if more > 0:
        t1 = time.time()
        orc.render(hs, cam, w, h, spp, spp + more, 50, mode=0, seed_a=11, seed_b=12, threads=threads, shared_b=1)
        dt += time.time() - t1
        total += w * h * more
        spp += more
    orc.scene_free(hs)
    res["value"] = total / dt / 1e6
    res["sample"] = ("%dx%d, %d spp, maxDepth 50, same %d-sphere scene and camera; oracle restatement with the "
                     "reference's shared mutex-guarded RNG-B and one task per row, %d threads, %.1f s"
                     % (w, h, spp, len(sc["kind"]), threads, dt))

    # configs[0] in full: 400x225, 100 spp, 5 fixed + 250 random spheres (fewer if the placement stops early)
    sc0 = orc.generate_scene(250, 1, 2)
    hs0 = orc.scene_create(sc0)
    t0 = time.time()
    orc.render(hs0, cam, 400, 225, 0, 100, 50, mode=0, seed_a=11, seed_b=12, threads=threads, shared_b=1)
    dt0 = time.time() - t0
    orc.scene_free(hs0)
    res["configs0"] = {"value": 400 * 225 * 100 / dt0 / 1e6, "unit": "Msamples/s", "seconds": dt0,
                       "sample": "configs[0] in full: 400x225, 100 spp, maxDepth 50, %d spheres, %d threads" % (len(sc0["kind"]), threads)}
    return res


def measured_traffic(default_workload):
    """HBM bytes per k_render launch from the newest committed rocprofv3 counter summary (profiles/*_summary.json:
    FETCH_SIZE and WRITE_SIZE collected in separate --pmc passes, FETCH doubled as the gfx950 correction). Only
    valid for the default workload, on which it was measured; the kernel is VALU-bound and these bytes are the
    fixed-point job sums (atomics) plus a few KB of scene data. Returns (bytes, source file) or (None, None)."""
    if not default_workload:
        return None, None
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_summary.json")), reverse=True):
        try:
            with open(path) as f:
                return float(json.load(f)["k_render_hbm_bytes_per_launch"]["total"]), os.path.relpath(path, ROOT)
        except (KeyError, ValueError, OSError, TypeError):
            continue
    return None, None


def extras(dev, host, accum, W, H, spp, depth, scene, cam):
    """Outside the timed region and outside the roofline figure: the same workload with the optional grid search
    (pt_set_accel), configs[4] (10k spheres, 256 spp) once with each search, the reference's
    default 400x225 frame, and the output stage at 4K."""
    import numpy as np
    import torch

    def kernel_ms(reps, s, w=W, h=H, buf=accum):
        dev.kernel_times()
        for _ in range(reps):
            buf.zero_()
            dev.render(w, h, 0, s, depth, 42, accum_ptr=buf.data_ptr())
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
    r0 = dev.ray_count()
    ms_scan = kernel_ms(1, 256)
    rays_scan = dev.ray_count() - r0
    n_big = len(big["kind"])
    res["configs4_10k_spheres_256spp"] = {"spheres": n_big, "full_scan_kernel_ms": ms_scan, "grid_kernel_ms": ms_grid,
                                          "full_scan_Msamples_per_s": W * H * 256 / ms_scan / 1e3,
                                          "full_scan_roofline_frac": rays_scan * (FLOP_PER_TEST * n_big + FLOP_PER_RAY_FIXED) / (ms_scan * 1e-3) / 1e12 / PEAK_FP32_TFLOPS,
                                          "grid_Msamples_per_s": W * H * 256 / ms_grid / 1e3, "grid": info}
    # the reference's own default frame (config.txt: 400x225, 100 spp, 5 + 250 spheres): small enough to under-fill the chip
    small = host.generate_scene(250, 1, 2)
    dev.set_scene(small)
    dev.set_camera(host.camera_setup(aspect=400 / 225))
    sbuf = torch.zeros((225, 400, 3), dtype=torch.float32, device="cuda")
    kernel_ms(20, 100, 400, 225, sbuf)          # a 1.5 ms kernel: the clocks need some tens of launches to settle
    r0 = dev.ray_count()
    ms_small = kernel_ms(20, 100, 400, 225, sbuf)
    rays_small = (dev.ray_count() - r0) / 20
    n_small = len(small["kind"])
    res["configs0_scene_on_gpu_400x225_100spp"] = {
        "spheres": n_small, "kernel_ms": ms_small, "Msamples_per_s": 400 * 225 * 100 / ms_small / 1e3,
        "roofline_frac": rays_small * (FLOP_PER_TEST * n_small + FLOP_PER_RAY_FIXED) / (ms_small * 1e-3) / 1e12 / PEAK_FP32_TFLOPS}
    dev.set_scene(scene)
    dev.set_camera(cam)
    return res


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    import numpy as np
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
    no_gpu = args.rehearse_without_gpu
    if not no_gpu and not torch.cuda.is_available():
        print("bench.py: no GPU visible; the sample loop has no CPU path", file=sys.stderr)
        sys.exit(2)
    host_reduce = args.rehearse_on_one_gpu or no_gpu
    if host_reduce:
        local_rank = 0
    if not no_gpu:
        # a launcher that gives every rank its own visible device (HIP_VISIBLE_DEVICES per rank) leaves one device, index 0
        local_rank %= max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
    ptd.init("gloo" if host_reduce else "nccl")   # "nccl" is RCCL; a single process needs no group

    W, H, spp, depth = args.width, args.height, args.spp, args.depth
    if args.scaling == "weak":
        spp_begin, spp_end = ptd.weak_range(rank, spp)          # disjoint ranges of one frame, `spp` each
        frame_spp = spp * world
    else:
        spp_begin, spp_end = ptd.split_range(spp, world, rank)  # the frame's `spp` cut into `world` ranges
        frame_spp = spp

    if no_gpu:
        dev = None
        n_spheres = args.spheres + 5
        accum = torch.zeros((H, W, 3), dtype=torch.float32)

        def step():
            accum.zero_()
            ptd.reduce_frame(accum, dst=0)

        def fence():
            if world > 1:
                dist.barrier()
    else:
        from path_tracer_amd import Device, host
        scene = host.generate_scene(args.spheres, 1, 2, half_extent=args.half_extent)
        n_spheres = len(scene["kind"])
        cam = host.camera_setup(aspect=W / H)
        dev = Device(local_rank)
        dev.set_scene(scene)
        dev.set_camera(cam)
        dev.set_stream(torch.cuda.current_stream().cuda_stream)
        accum = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")

        def step():
            accum.zero_()
            dev.render(W, H, spp_begin, spp_end, depth, 42, accum_ptr=accum.data_ptr())
            if host_reduce and world > 1:
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
    rays0 = 0
    if dev:
        dev.kernel_times()                       # drop the warm-up records
        rays0 = dev.ray_count()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    rays = dev.ray_count() - rays0 if dev else 0
    ktimes = dev.kernel_times() if dev else []

    if world > 1:
        stat_dev = "cpu" if host_reduce else "cuda"
        t = torch.tensor([dt], dtype=torch.float64, device=stat_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        r = torch.tensor([rays], dtype=torch.int64, device=stat_dev)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        total_rays = int(r.item())
    else:
        total_rays = rays

    if rank == 0:
        samples = float(W) * H * frame_spp * args.steps
        k_ms = float(np.mean(ktimes)) if len(ktimes) else float("nan")
        rays_per_launch = rays / max(args.steps, 1)                       # this rank's kernel
        flops_per_launch = rays_per_launch * (FLOP_PER_TEST * n_spheres + FLOP_PER_RAY_FIXED)
        achieved = flops_per_launch / (k_ms * 1e-3) / 1e12
        default_workload = (args.workload == "configs1" and not args.custom)
        traffic, traffic_source = measured_traffic(default_workload)
        per_gpu = "%d spp per GPU" % spp if args.scaling == "weak" else "%d spp per frame" % spp
        out = {
            "metric": "Msamples/s at 1080p/500 spheres",
            "value": None if no_gpu else samples / dt / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%s: %dx%d, %s, maxDepth %d, %d spheres (5 fixed + %d random, "
                                   "scene seeds 1/2), default camera, render seed 42" % (args.label, W, H, per_gpu, depth, n_spheres, args.spheres),
                       "width": W, "height": H, "max_depth": depth, "spheres": n_spheres,
                       "parallelism": "spp-split x%d + RCCL reduce" % world if world > 1 else "single GPU"},
            "rays_per_s": total_rays / dt,
            "rays_per_sample": total_rays / samples,
            "roofline": {"bound": "valu_fp32", "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_TFLOPS,
                         "traffic": traffic,
                         "traffic_source": ("replayed from %s (rocprofv3 --pmc passes of this workload, taken in a separate run: "
                                            "counters cannot be collected inside the timed run)" % traffic_source) if traffic_source else None,
                         "kernel": "k_render<0>", "kernel_ms": k_ms,
                         "flops_per_launch": flops_per_launch,
                         "note": "algorithmic flops = rays x (17 x spheres + 60); rays counted by the kernel"},
        }
        if args.scaling == "weak":
            out["config"]["spp_per_gpu"] = spp
        else:
            out["config"]["spp_per_frame"] = spp
            out["config"]["spp_this_rank"] = spp_end - spp_begin
        if no_gpu:
            out["data"] = "rehearsal without a GPU: nothing was rendered"
            out["roofline"] = None
        elif host_reduce and world > 1:
            out["data"] = "synthetic; REHEARSAL on one GPU (all ranks on cuda:0, reduce through gloo on host copies): not a measurement"
        if world == 1 and not args.no_extras and not no_gpu:
            try:
                out["extras"] = extras(dev, host, accum, W, H, spp, depth, scene, cam)
            except Exception as e:                                   # never lose the headline line to an extra
                out["extras"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline and not no_gpu:
            try:
                out["cpu_baseline"] = cpu_baseline(min(16, os.cpu_count() or 1))
            except Exception as e:                                   # e.g. oracle/_build missing: the headline line still prints
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
