"""CPU, world_size 2, gloo: the spp split and the reduce compose to the single-process frame.
The per-rank "renderer" here is the oracle in philox mode (the same numbering the GPU draws), so
this checks the partition + collective logic that bench.py and the driver use, without a GPU."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from path_tracer_amd import distributed as ptd


def test_ranges_are_disjoint_and_cover():
    for total, world in [(100, 1), (100, 2), (100, 8), (1024, 8), (7, 4), (3, 8)]:
        r = [ptd.split_range(total, world, k) for k in range(world)]
        assert r[0][0] == 0 and r[-1][1] == total
        assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
        assert max(e - b for b, e in r) - min(e - b for b, e in r) <= 1
    assert [ptd.weak_range(k, 100) for k in range(3)] == [(0, 100), (100, 200), (200, 300)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from oracle_lib import Oracle
    r, _, w = ptd.init("gloo")
    assert (r, w) == (rank, world)
    orc = Oracle()
    sc = orc.generate_scene(60, 1, 2)
    hs = orc.scene_create(sc)
    cam = orc.camera_setup()
    W, H, total = 48, 27, 6
    b, e = ptd.split_range(total, world, rank)
    part, _, rays = orc.render(hs, cam, W, H, b, e, 50, mode=1, seed_a=42, seed_b=0, threads=1)
    accum = torch.from_numpy(part.astype(np.float32))
    ptd.reduce_frame(accum, dst=0)
    rays_t = torch.tensor([rays], dtype=torch.int64)
    dist.all_reduce(rays_t)
    if rank == 0:
        full, _, full_rays = orc.render(hs, cam, W, H, 0, total, 50, mode=1, seed_a=42, seed_b=0, threads=1)
        np.savez(out_path, reduced=accum.numpy(), full=full, rays=int(rays_t.item()), full_rays=full_rays)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_split_and_reduce(tmp_path):
    out = str(tmp_path / "r.npz")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    g = np.load(out)
    assert int(g["rays"]) == int(g["full_rays"])
    assert np.allclose(g["reduced"], g["full"], rtol=1e-6, atol=1e-6)


def test_single_process_needs_no_group():
    os.environ.pop("WORLD_SIZE", None)
    os.environ.pop("RANK", None)
    t = torch.ones(4)
    assert ptd.reduce_frame(t) is t
