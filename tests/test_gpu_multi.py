"""GPU (-m gpu): more than one rank / more than one handle.

  * two PROCESSES that render disjoint sample ranges with the HIP path and sum them (gloo on host copies, because
    this box has one GPU and RCCL takes one rank per device): equals the one-process frame;
  * bench.py --gpus 2 through its own launcher on one GPU (--rehearse-on-one-gpu);
  * pt_reduce (RCCL, one process, one handle per GPU): needs two devices, skipped otherwise; its argument checks and
    the one-handle case run on any box."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _n_devices():
    import torch
    return torch.cuda.device_count()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, %(root)r)
import torch
import torch.distributed as dist
from path_tracer_amd import Device, host, distributed as ptd
rank, _, world = ptd.init("gloo")
W, H, total, depth, seed = 160, 90, 13, 50, 42
dev = Device(0)
dev.set_scene(host.generate_scene(250, 1, 2))
dev.set_camera(host.camera_setup())
b, e = ptd.split_range(total, world, rank)
dev.clear_accum(W, H)
rays = dev.render(W, H, b, e, depth, seed, want_rays=True)
part = torch.from_numpy(dev.read_accum(W, H))
ptd.reduce_frame(part, dst=0)
r = torch.tensor([rays], dtype=torch.int64)
dist.all_reduce(r)
if rank == 0:
    whole, whole_rays = dev.render_image(W, H, total, depth, seed=seed)
    np.savez(%(out)r, reduced=part.numpy(), whole=whole, rays=int(r.item()), whole_rays=whole_rays)
dist.barrier()
dist.destroy_process_group()
dev.close()
'''


def test_two_processes_render_with_hip_and_reduce(tmp_path):
    out = str(tmp_path / "two.npz")
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % {"root": ROOT, "out": out})
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ)
        env.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    for p in procs:
        so, se = p.communicate(timeout=300)
        assert p.returncode == 0, se.decode()[-2000:]
    g = np.load(out)
    assert int(g["rays"]) == int(g["whole_rays"])                      # ray counts add exactly
    # the two halves were rounded to float separately before the sum: equal up to float addition order
    assert np.abs(g["reduced"] - g["whole"]).max() <= 4e-6 * max(1.0, float(g["whole"].max()))
    assert g["whole"].any()


def test_bench_two_ranks_through_its_own_launcher():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    results = {}
    for scaling in ("weak", "strong"):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--width", "320",
                            "--height", "180", "--spp", "8", "--scaling", scaling, "--rehearse-on-one-gpu"], env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
        assert len(lines) == 1, lines
        results[scaling] = json.loads(lines[0])
    w, s = results["weak"], results["strong"]
    assert w["n_gpus"] == s["n_gpus"] == 2 and w["value"] > 0 and s["value"] > 0 and "REHEARSAL" in w["data"]
    # weak: 8 spp per rank = 16 per frame; strong: 8 per frame = 4 per rank; both count every sample of the frame once
    assert w["config"]["spp_per_gpu"] == 8 and s["config"]["spp_per_frame"] == 8 and s["config"]["spp_this_rank"] == 4
    assert abs(w["rays_per_sample"] - s["rays_per_sample"]) < 0.05 * s["rays_per_sample"]
    assert 2.0 < s["rays_per_sample"] < 3.2


def test_bench_one_rank_rccl_group_on_the_device():
    """bench.py's own N > 1 code (RCCL reduce of the device buffer between HIP events on the kernels' stream, barriers,
    gathers of device tensors) with a one-rank group: the exact statements an 8-GPU run executes, on the one GPU there is."""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--one-rank-group", "--steps", "3", "--warmup", "1", "--width", "640",
                        "--height", "360", "--spp", "16", "--no-extras", "--no-cpu-baseline"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines                                   # RCCL's version banner goes to stderr, the result alone to stdout
    out = json.loads(lines[0])
    ph = out["phases"]
    assert out["n_gpus"] == 1 and out["value"] > 0 and "one-rank RCCL group" in out["data"]
    assert len(ph["kernel_ms_per_rank"]) == 1 and ph["kernel_ms_per_rank"][0] > 0
    assert 0 <= ph["reduce_ms"] < ph["step_ms_per_rank"][0] and ph["rays_per_rank"][0] > 640 * 360 * 16 * 3
    assert 0.2 < out["roofline"]["frac"] < 1.0


_RCCL_ONE_RANK = r'''
import os, sys
sys.path.insert(0, %(root)r)
import torch
import torch.distributed as dist
from path_tracer_amd import Device, host, distributed as ptd
torch.cuda.set_device(0)
rank, _, world = ptd.init("nccl", force=True)          # "nccl" is RCCL on ROCm
assert dist.get_backend() == "nccl" and (rank, world) == (0, 1)
W, H = 160, 90
dev = Device(0)
dev.set_scene(host.generate_scene(250, 1, 2)); dev.set_camera(host.camera_setup())
dev.set_stream(torch.cuda.current_stream().cuda_stream)
accum = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
dev.render(W, H, 0, 4, 50, 42, accum_ptr=accum.data_ptr())
before = accum.clone()
dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM)        # what ptd.reduce_frame issues with more than one rank
torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
t = torch.tensor([1.5], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX)
r = torch.tensor([dev.ray_count()], dtype=torch.int64, device="cuda"); dist.all_reduce(r, op=dist.ReduceOp.SUM)
assert torch.equal(accum, before) and before.any() and float(t.item()) == 1.5 and int(r.item()) == dev.ray_count()
dist.barrier(); dist.destroy_process_group(); dev.close()
print("rccl one-rank ok")
'''


def test_one_rank_rccl_group_runs_the_collectives_of_the_bench(tmp_path):
    """RCCL takes one rank per GPU, so a one-GPU box cannot run two ranks; a one-rank group still goes through the same
    init (device_id), reduce, barrier and all_reduce calls on device tensors that bench.py makes with N ranks."""
    script = tmp_path / "one_rank.py"
    script.write_text(_RCCL_ONE_RANK % {"root": ROOT})
    env = dict(os.environ)
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    p = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0 and b"rccl one-rank ok" in p.stdout, p.stderr.decode()[-3000:]


def test_pt_reduce_argument_checks_on_one_gpu():
    from path_tracer_amd import Device, PtError, host
    from path_tracer_amd._capi import reduce, reduce_stats
    a, b = Device(0), Device(0)
    try:
        for d in (a, b):
            d.set_scene(host.generate_scene(20, 1, 2))
            d.set_camera(host.camera_setup())
            d.clear_accum(32, 18)
        inits0, _ = reduce_stats()
        reduce([a])                                           # one handle: nothing to sum
        with pytest.raises(PtError, match="one handle per GPU"):
            reduce([a, b])                                    # both on device 0
        b.clear_accum(16, 9)
        with pytest.raises(PtError):
            reduce([a, b])
        assert reduce_stats()[0] == inits0                    # no communicator was set up for any of these
    finally:
        a.close(); b.close()


def test_pt_reduce_through_rccl_with_one_rank(monkeypatch):
    """pt_reduce's RCCL calls (librccl through dlopen, ncclCommInitAll, ncclReduce with the float32 / sum codes, group
    start / end, the cached communicator) on the one GPU there is: PT_REDUCE_FORCE_RCCL=1 makes a one-handle reduce go
    through a one-rank communicator instead of returning at once. The buffer must come back unchanged."""
    from path_tracer_amd import Device, host
    from path_tracer_amd._capi import reduce, reduce_stats
    monkeypatch.setenv("PT_REDUCE_FORCE_RCCL", "1")
    d = Device(0)
    try:
        d.set_scene(host.generate_scene(250, 1, 2)); d.set_camera(host.camera_setup())
        W, H = 160, 90
        before, _ = d.render_image(W, H, 3, 50, seed=5)
        inits0, _ = reduce_stats()
        reduce([d]); reduce([d])
        inits, ranks = reduce_stats()
        assert inits == inits0 + 1 and ranks == 1                  # set up once, re-used by the second call
        after = d.read_accum(W, H)
        assert before.any() and np.array_equal(after, before)
    finally:
        d.close()


@pytest.mark.skipif(_n_devices() < 2, reason="pt_reduce sums over RCCL between two GPUs of this process")
def test_pt_reduce_two_gpus_and_cached_communicators():
    from path_tracer_amd import Device, host
    from path_tracer_amd._capi import reduce, reduce_stats
    sc, cam = host.generate_scene(250, 1, 2), host.camera_setup()
    W, H, total = 200, 113, 12
    devs = [Device(0), Device(1)]
    try:
        for d in devs:
            d.set_scene(sc); d.set_camera(cam)
        inits0, _ = reduce_stats()
        for rep in range(2):
            rays = 0
            for g, d in enumerate(devs):
                d.clear_accum(W, H)
                rays += d.render(W, H, total * g // 2, total * (g + 1) // 2, 50, 42, want_rays=True)
            reduce(devs, root=0)
            summed = devs[0].read_accum(W, H)
            whole, whole_rays = devs[1].render_image(W, H, total, 50, seed=42)
            assert rays == whole_rays
            assert np.abs(summed - whole).max() <= 4e-6 * max(1.0, float(whole.max()))
            inits, ranks = reduce_stats()
            assert inits == inits0 + 1 and ranks == 2          # the second reduce re-used the first one's communicators
    finally:
        for d in devs:
            d.close()
