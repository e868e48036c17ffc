"""CPU: bench.py starts its own ranks. `--gpus 2` with no WORLD_SIZE in the environment must spawn two fresh
processes (one per GPU), rendezvous, reduce, time and print ONE JSON line from rank 0; the outside-launcher path
(torch.distributed.run) must keep working. Without a GPU nothing can be rendered (the sample loop has no CPU
path), so these runs use --rehearse-without-gpu: launch and collective control flow only, `value` is null."""
import json
import os
import socket
import subprocess
import sys

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")
SMALL = ["--steps", "2", "--warmup", "1", "--width", "64", "--height", "36", "--rehearse-without-gpu"]


def _clean_env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    return env


def _one_json_line(stdout):
    lines = [ln for ln in stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_bench_spawns_its_own_ranks():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--scaling", "strong", "--spp", "7"] + SMALL, env=_clean_env(),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    out = _one_json_line(p.stdout)
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1
    assert out["scaling"] == "strong" and out["config"]["spp_per_frame"] == 7 and out["config"]["spp_this_rank"] == 3
    assert out["value"] is None and "rehearsal" in out["data"]          # nothing was rendered: not a measurement
    assert out["ms_per_step"] > 0


def test_phases_and_efficiency_fields_of_a_two_rank_line():
    env = _clean_env()
    env["PT_BENCH_N1_MS"] = "1000"
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2"] + SMALL, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    out = _one_json_line(p.stdout)
    ph = out["phases"]
    for k in ("kernel_ms_per_rank", "reduce_ms_per_rank", "step_ms_per_rank", "rays_per_rank"):
        assert len(ph[k]) == 2, k
    assert ph["reduce_ms"] == ph["reduce_ms_per_rank"][0] and ph["reduce_ms"] > 0 and "max_rank_skew_ms" in ph
    assert all(abs(v - out["ms_per_step"]) < 0.5 * out["ms_per_step"] + 5 for v in ph["step_ms_per_rank"])
    eff = out["scaling_efficiency_vs"]
    assert eff["source"] == "PT_BENCH_N1_MS" and abs(eff["efficiency"] - 1000.0 / out["ms_per_step"]) < 1e-6   # weak: ideal = the 1-GPU step


def test_one_rank_group_runs_the_collective_code_of_the_n_rank_path():
    p = subprocess.run([sys.executable, BENCH, "--one-rank-group"] + SMALL, env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    out = _one_json_line(p.stdout)              # gloo's connection note (and RCCL's banner on a GPU) must not reach stdout
    ph = out["phases"]
    assert out["n_gpus"] == 1 and len(ph["step_ms_per_rank"]) == 1 and ph["reduce_ms"] >= 0
    assert ph["kernel_ms_per_rank"] == [None] and ph["max_rank_skew_ms"] is None        # nothing was rendered: no kernel time


def test_a_rank_that_dies_at_start_ends_the_launch_quickly():
    """Rank 1 exits with code 3 before the rendezvous: rank 0 would wait in init_process_group for minutes. The parent
    must notice, terminate rank 0, name the failed rank and return non-zero, in well under the backend's timeout."""
    import time
    env = _clean_env()
    env["PT_BENCH_FAIL_RANK"] = "1"
    t0 = time.time()
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2"] + SMALL, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    took = time.time() - t0
    assert p.returncode == 3, (p.returncode, p.stderr.decode()[-2000:])
    assert took < 30, took
    err = p.stderr.decode()
    assert "rank 1 of 2 exited with code 3" in err and "PT_BENCH_FAIL_RANK" in err
    assert not [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]      # no result line from a failed launch


def test_launch_timeout_terminates_the_ranks():
    import time
    env = _clean_env()
    env["PT_BENCH_HANG_RANK"] = "0"
    t0 = time.time()
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-timeout", "8"] + SMALL, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=120)
    assert p.returncode == 124 and time.time() - t0 < 40, (p.returncode, p.stderr.decode()[-2000:])
    assert b"launch timed out" in p.stderr


def test_bench_under_an_outside_launcher():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), BENCH, "--gpus", "2"] + SMALL, env=_clean_env(),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    out = _one_json_line(p.stdout)              # under an outside launcher too: nothing but the result on stdout
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["spp_per_gpu"] == 100


def test_workload_table_and_rank_mismatch():
    sys.path.insert(0, ROOT)
    import bench
    a = bench.parse_args(["--workload", "configs2", "--scaling", "strong", "--gpus", "8"])
    assert (a.width, a.height, a.spp, a.depth, a.spheres) == (3840, 2160, 1024, 50, 500) and not a.custom
    a = bench.parse_args(["--workload", "configs4"])
    assert (a.width, a.height, a.spp, a.spheres, a.half_extent) == (1920, 1080, 256, 10000, 45.0)
    a = bench.parse_args([])
    assert (a.width, a.height, a.spp, a.depth, a.spheres, a.scaling, a.gpus) == (1920, 1080, 100, 50, 500, "weak", 1)
    # an outside launcher that started a different number of ranks than --gpus says is an error, not a guess
    env = _clean_env()
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2"] + SMALL, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode == 2 and b"WORLD_SIZE" in p.stderr
