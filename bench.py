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
    ap.add_argument("--launch-timeout", type=float, default=None,
                    help="seconds the parent waits for its ranks before it terminates them (default PT_BENCH_TIMEOUT or 1500)")
    ap.add_argument("--one-rank-group", action="store_true",
                    help="with --gpus 1: join a ONE-rank RCCL process group and run the reduce, the barriers, the event timing and "
                         "the gathers of the N > 1 path on the device (RCCL takes one rank per GPU, so a one-GPU box cannot run two); "
                         "the line then carries `phases` like an N > 1 line")
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


def _tail(path, n_bytes=3000):
    try:
        with open(path, "rb") as f:
            f.seek(0, os.SEEK_END)
            size = f.tell()
            f.seek(max(0, size - n_bytes))
            return f.read().decode(errors="replace")
    except OSError:
        return ""


def spawn_ranks(n, argv, timeout_s=None):
    """Starts n fresh copies of this script, one rank each, and relays rank 0's output. A process that has used
    the GPU is never replaced or re-executed: the parent only watches its children.

    Fail-fast: ALL children are polled. The moment one exits non-zero (a bad device index, an RCCL set-up error) the
    others would wait in the rendezvous or in a collective for the backend's multi-minute timeout, so they are
    terminated (exactly the children started here, by PID), the failed rank is named with the tail of its stderr and
    its exit code is returned. The whole launch has a time limit (--launch-timeout / PT_BENCH_TIMEOUT, seconds)."""
    import tempfile
    if timeout_s is None:
        timeout_s = float(os.environ.get("PT_BENCH_TIMEOUT", "1500"))
    port = _free_port()
    tmp = tempfile.mkdtemp(prefix="pt_bench_")
    procs, logs = [], []
    for rank in range(n):
        env = dict(os.environ)
        env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PT_BENCH_CHILD="1")
        if "HSA_ENABLE_IPC_MODE_LEGACY" not in env and not os.environ.get("PT_BENCH_KEEP_ENV"):
            # The pool's image exports HSA_ENABLE_IPC_MODE_LEGACY=0 (its host driver only does dmabuf IPC; without it
            # RCCL between processes fails with `hipIpcGetMemHandle: invalid argument`, per the pool's documentation),
            # so on the pool this branch does not run. Elsewhere the variable is added and SAID so; PT_BENCH_KEEP_ENV=1
            # leaves the environment alone. DESIGN.md section 4.
            env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
            if rank == 0:
                print("bench.py: HSA_ENABLE_IPC_MODE_LEGACY was not set; children get =0 (PT_BENCH_KEEP_ENV=1 to leave it unset)",
                      file=sys.stderr, flush=True)
        so = open(os.path.join(tmp, "rank%d.out" % rank), "wb")
        se = open(os.path.join(tmp, "rank%d.err" % rank), "wb")
        logs.append((so, se))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, stdout=so, stderr=se))

    def stop_all():
        for p in procs:
            if p.poll() is None:
                p.terminate()                                    # the exact children started above
        t_end = time.time() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()

    t_start = last_note = time.time()
    rc, failed = 0, None
    while True:
        states = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(states) if c not in (None, 0)]
        if bad:
            failed, rc = bad[0]
            stop_all()
            break
        if all(c == 0 for c in states):
            break
        if time.time() - t_start > timeout_s:
            running = [r for r, c in enumerate(states) if c is None]
            print("bench.py: launch timed out after %.0f s; ranks still running: %s" % (timeout_s, running), file=sys.stderr, flush=True)
            failed, rc = running[0], 124
            stop_all()
            break
        if time.time() - last_note > 60:
            last_note = time.time()
            print("bench.py: %d ranks running, %.0f s" % (n, last_note - t_start), file=sys.stderr, flush=True)
        time.sleep(0.05)
    for so, se in logs:
        so.close()
        se.close()
    # rank 0's stderr is relayed whole (warnings of the libraries); a failed rank's tail is shown with its number
    sys.stderr.write(_tail(os.path.join(tmp, "rank0.err"), 1 << 20))
    if failed is not None:
        print("bench.py: rank %d of %d exited with code %d after %.1f s; the other ranks were terminated. Its stderr ends:\n%s"
              % (failed, n, rc, time.time() - t_start, _tail(os.path.join(tmp, "rank%d.err" % failed))), file=sys.stderr, flush=True)
    # rank 0's JSON line goes to stdout; anything else a library printed there (gloo's connection notes) to stderr
    try:
        with open(os.path.join(tmp, "rank0.out"), "rb") as f:
            out = f.read()
    except OSError:
        out = b""
    for line in out.decode(errors="replace").splitlines():
        is_json = line.startswith("{") and line.rstrip().endswith("}")
        if is_json and failed is not None:
            continue                                             # a failed launch reports nothing as a result
        print(line, file=sys.stdout if is_json else sys.stderr, flush=True)
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
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
    import re

    def newest_first(path):
        # r03_summary.json (the round's main collection) before r03c_, r03b_ (other boxes), before r02_ ...; the grid's are not these
        m = re.match(r"r(\d+)([a-z]?)_summary\.json$", os.path.basename(path))
        return (-int(m.group(1)), m.group(2)) if m else (1, path)

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_summary.json")), key=newest_first):
        try:
            with open(path) as f:
                return float(json.load(f)["k_render_hbm_bytes_per_launch"]["total"]), os.path.relpath(path, ROOT)
        except (KeyError, ValueError, OSError, TypeError):
            continue
    return None, None


def n1_record(args, ms_per_step=None):
    """The 1-GPU step time of the same workload arguments, for `scaling_efficiency_vs`: PT_BENCH_N1_MS when given, else what
    an earlier `--gpus 1` run of the same arguments on this machine left in the temporary directory. With ms_per_step
    (a 1-GPU run) the record is written."""
    import tempfile
    key = "%s_%dx%d_%dspp_d%d_%dsph_%s" % (args.workload, args.width, args.height, args.spp, args.depth, args.spheres, args.scaling)
    path = os.path.join(tempfile.gettempdir(), "pt_bench_n1_%s.json" % key)
    if ms_per_step is not None:
        try:
            with open(path, "w") as f:
                json.dump({"ms_per_step": ms_per_step, "time": time.time()}, f)
        except OSError:
            pass
        return None
    if os.environ.get("PT_BENCH_N1_MS"):
        try:
            return {"ms_per_step": float(os.environ["PT_BENCH_N1_MS"]), "source": "PT_BENCH_N1_MS"}
        except ValueError:
            return None
    try:
        with open(path) as f:
            return {"ms_per_step": float(json.load(f)["ms_per_step"]), "source": path}
    except (OSError, ValueError, KeyError):
        return None


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

    def frac(rays, n_sph, ms):
        return rays * (FLOP_PER_TEST * n_sph + FLOP_PER_RAY_FIXED) / (ms * 1e-3) / 1e12 / PEAK_FP32_TFLOPS

    def scan_line(reps, s, w=W, h=H, buf=accum, n_sph=None, d=depth):
        """Full-scan workload: kernel ms (k_render<0> + k_tail) with the k_tail part apart, rays per launch, roofline fraction."""
        dev.kernel_times()
        r0 = dev.ray_count()
        for _ in range(reps):
            buf.zero_()
            dev.render(w, h, 0, s, d, 42, accum_ptr=buf.data_ptr())
        torch.cuda.synchronize()
        whole, tail = dev.kernel_times_split()
        rays = (dev.ray_count() - r0) / reps
        ms = float(np.min(whole))
        return {"kernel_ms": ms, "k_tail_ms": float(np.min(tail)), "k_render_ms": ms - float(tail[int(np.argmin(whole))]),
                "Msamples_per_s": w * h * s / ms / 1e3, "rays_per_sample": rays / (w * h * s),
                "roofline_frac": frac(rays, n_sph if n_sph else len(scene["kind"]), ms)}

    # the headline workload once more, with the split between the sample-loop kernel and the end-of-frame kernel
    res["configs1_kernel_split"] = scan_line(3, spp)
    # the north-star frame: the same scene at 1,024 spp (DEFAULT workload only: other sizes say so in their own line)
    res["north_star_1080p_1024spp"] = scan_line(1, 1024)
    # configs[3]: 10 % diffuse / 40 % metal / 50 % glass, maxDepth 100 (stress for divergence and path regeneration)
    glassy = host.generate_scene(500, 1, 2, diffuse_below=0.1, metal_below=0.5)
    dev.set_scene(glassy)
    line = scan_line(3, spp, n_sph=len(glassy["kind"]), d=100)
    line["spheres"] = len(glassy["kind"])
    res["configs3_glass_metal_heavy_depth100"] = line
    dev.set_scene(scene)
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
    n_small = len(small["kind"])
    line = scan_line(20, 100, 400, 225, sbuf, n_sph=n_small, d=50)
    line["spheres"] = n_small
    res["configs0_scene_on_gpu_400x225_100spp"] = line
    dev.set_scene(scene)
    dev.set_camera(cam)
    return res


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:], args.launch_timeout))
    # fault injection for tests/test_bench_launch.py: the named rank dies before it joins the rendezvous
    if os.environ.get("PT_BENCH_FAIL_RANK") == os.environ.get("RANK", "0") and os.environ.get("PT_BENCH_CHILD"):
        print("bench.py: PT_BENCH_FAIL_RANK: rank %s exits with code 3 on purpose" % os.environ.get("RANK"), file=sys.stderr, flush=True)
        sys.exit(3)
    if os.environ.get("PT_BENCH_HANG_RANK") == os.environ.get("RANK", "0") and os.environ.get("PT_BENCH_CHILD"):
        time.sleep(3600)                          # fault injection: a rank that never reaches the rendezvous

    # Rank 0's stdout carries ONE line, the result. Libraries print there too (RCCL its version banner at set-up, gloo a
    # connection note): from here on file descriptor 1 is stderr's, and the result goes to the descriptor saved now.
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

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
    grouped = world > 1 or args.one_rank_group      # the collectives run (a one-rank group exercises the same calls)
    if args.one_rank_group:
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
    ptd.init("gloo" if host_reduce else "nccl", force=args.one_rank_group)   # "nccl" is RCCL; a single process needs no group

    W, H, spp, depth = args.width, args.height, args.spp, args.depth
    if args.scaling == "weak":
        spp_begin, spp_end = ptd.weak_range(rank, spp)          # disjoint ranges of one frame, `spp` each
        frame_spp = spp * world
    else:
        spp_begin, spp_end = ptd.split_range(spp, world, rank)  # the frame's `spp` cut into `world` ranges
        frame_spp = spp

    reduce_marks = []                                # per step: (event, event) around the reduce, or seconds on the host path
    if no_gpu:
        dev = None
        n_spheres = args.spheres + 5
        accum = torch.zeros((H, W, 3), dtype=torch.float32)

        def step():
            accum.zero_()
            t_r = time.perf_counter()
            ptd.reduce_frame(accum, dst=0)
            reduce_marks.append(time.perf_counter() - t_r)

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
            if not grouped:
                return
            if host_reduce:
                t_r = time.perf_counter()
                host_copy = accum.cpu()
                ptd.reduce_frame(host_copy, dst=0)
                accum.copy_(host_copy)
                reduce_marks.append(time.perf_counter() - t_r)
            else:
                # Events on torch's current stream, the stream the kernels run on (set_stream above). dist.reduce runs on
                # RCCL's own stream, which first waits for this one and which this one then waits for, so the pair brackets
                # "wait for the slowest rank + the transfer" as this rank sees it.
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ptd.reduce_frame(accum, dst=0, force=args.one_rank_group)
                e1.record()
                reduce_marks.append((e0, e1))

        def fence():
            torch.cuda.synchronize()
            if grouped:
                dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    rays0 = 0
    if dev:
        dev.kernel_times()                       # drop the warm-up records
        rays0 = dev.ray_count()
    del reduce_marks[:]
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    rays = dev.ray_count() - rays0 if dev else 0
    ktimes = dev.kernel_times() if dev else []

    phases = None
    if grouped:
        stat_dev = "cpu" if host_reduce else "cuda"
        own_dt = dt
        t = torch.tensor([dt], dtype=torch.float64, device=stat_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        r = torch.tensor([rays], dtype=torch.int64, device=stat_dev)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        total_rays = int(r.item())
        # where each rank's step went: its sample-loop kernels, its reduce (wait for the slowest rank + transfer), its wall time
        if reduce_marks and not isinstance(reduce_marks[0], float):
            red = [a.elapsed_time(b) for a, b in reduce_marks]
        else:
            red = [1e3 * v for v in reduce_marks]
        mine = torch.tensor([float(np.mean(ktimes)) if len(ktimes) else float("nan"),
                             float(np.max(ktimes)) if len(ktimes) else float("nan"),
                             float(np.mean(red)) if red else float("nan"),
                             own_dt / args.steps * 1e3, float(rays)], dtype=torch.float64, device=stat_dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        tab = np.array([e.cpu().numpy() for e in every])
        num = lambda v: None if v != v else float(v)                     # NaN (nothing rendered: no kernel time) -> null
        have_k = bool(np.isfinite(tab[:, 0]).any())
        phases = {"kernel_ms_per_rank": [num(v) for v in tab[:, 0]],
                  "kernel_ms_max_per_rank": [num(v) for v in tab[:, 1]],
                  "reduce_ms_per_rank": [num(v) for v in tab[:, 2]],
                  "step_ms_per_rank": [num(v) for v in tab[:, 3]],
                  "rays_per_rank": [int(v) for v in tab[:, 4]],
                  "reduce_ms": num(tab[0, 2]),
                  "max_rank_skew_ms": float(np.nanmax(tab[:, 0]) - np.nanmin(tab[:, 0])) if have_k else None,
                  "note": "means over the timed steps; kernel = k_render + k_tail (HIP events in the library); reduce = events on "
                          "the kernels' stream around dist.reduce (host path: wall clock incl. the copies), i.e. waiting for the "
                          "slowest rank plus the transfer; reduce_ms is rank 0's (the root); skew = max - min kernel_ms over ranks"}
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
        if phases:
            out["phases"] = phases
        n1 = n1_record(args, out["ms_per_step"] if world == 1 and not no_gpu else None)
        if world > 1 and n1:
            ideal = n1["ms_per_step"] if args.scaling == "weak" else n1["ms_per_step"] / world
            out["scaling_efficiency_vs"] = {"n1_ms_per_step": n1["ms_per_step"], "source": n1["source"],
                                            "efficiency": ideal / out["ms_per_step"],
                                            "note": "ideal step = the 1-GPU step of the same arguments (weak) or that / N (strong); "
                                                    "the driver computes its own from the per-N lines"}
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
        if args.one_rank_group and not no_gpu:
            out["data"] = "synthetic; one-rank RCCL group (--one-rank-group): the collectives of the N > 1 path on one device"
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
        print(json.dumps(out), file=result_out, flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
