"""N > 1 path on CPU: two gloo ranks shard a batch, each produces the control records of its
block (here with the oracle standing in for the GPU solve -- tests may use it), and one
all-gather reassembles the global result identically on both ranks."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mpc_local_planner_amd import scenarios as sc, sharding
from oracle import oracle as orc


def test_shard_ranges_cover_the_batch():
    for world in (1, 2, 3, 8):
        for total in (0, 1, 7, 64, 4096, 1048576):
            edges = [sharding.shard_range(r, world, total) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            for a, b in zip(edges[:-1], edges[1:]):
                assert a[1] == b[0]
            sizes = sharding.shard_sizes(world, total)
            assert max(sizes) - min(sizes) <= 1 and sum(sizes) == total


def _records(x0, p, N):
    cfg = orc.default_config(N, 0.35)
    xo, uo, so, io = orc.rti_batch(cfg, x0, p)
    rec = np.zeros((x0.shape[0], 6))
    rec[:, 0:2] = uo[:, 0, :]
    rec[:, 2:4] = xo[:, 1, 4:6]
    rec[:, 5] = so + 16 * io
    return rec


def _worker(rank, world, port, total, N, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        start, stop = sharding.shard_range(rank, world, total)
        x0, p = sc.config2(stop - start, start=start)          # this rank's block, generated locally
        local = torch.from_numpy(_records(x0, p, N))
        allrec = sharding.gather_controls(local, total=total)
        np.save(os.path.join(out_dir, f"rank{rank}.npy"), allrec.numpy())
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_solve_and_gather(tmp_path):
    world, total, N = 2, 13, 12                 # ragged on purpose: 7 + 6
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, total, N, str(tmp_path)), nprocs=world, join=True)
    a = np.load(tmp_path / "rank0.npy")
    b = np.load(tmp_path / "rank1.npy")
    assert a.shape == (total, 6) and np.array_equal(a, b)
    x0, p = sc.config2(total)
    assert np.array_equal(a, _records(x0, p, N))   # same as solving the whole batch on one rank


def _pipeline_worker(rank, world, port, steps, B, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        records = [torch.zeros(B * 6, dtype=torch.float64) for _ in range(2)]
        gathered = [torch.full((world * B * 6,), -1.0, dtype=torch.float64) for _ in range(2)]
        pipe = sharding.ControlGatherPipeline(records, gathered, sharding.HostSync())
        seen = []
        for s in range(steps):
            i = pipe.begin_step()
            assert i == (s & 1)
            # nothing may still be reading the buffer the "solve" is about to overwrite
            assert pipe.tokens[i] is None
            if s >= 2:   # ... and the gather of step s-2, which used this pair of buffers, is complete
                want = torch.cat([_step_records(r, s - 2, B) for r in range(world)])
                assert torch.equal(gathered[i], want)
            records[i].copy_(_step_records(rank, s, B))         # stands in for the solve of step s
            pipe.end_step()
            seen.append(i)
        pipe.finish()
        i, last = pipe.last()
        want = torch.cat([_step_records(r, steps - 1, B) for r in range(world)])
        assert i == ((steps - 1) & 1) and torch.equal(last, want)
        assert torch.equal(last.reshape(world, B * 6)[rank], records[i])
        np.save(os.path.join(out_dir, f"pipe{rank}.npy"), last.numpy())
    finally:
        dist.destroy_process_group()


def _step_records(rank, step, B):
    return (torch.arange(B * 6, dtype=torch.float64) + 1000.0 * step + 1e6 * rank)


def test_two_rank_gather_pipeline_double_buffering(tmp_path):
    """The overlap logic bench.py uses on GPUs (two record buffers, the gather of step s in flight
    during the solve of step s+1, a solve waiting only for the collective that last read its buffer)
    with two gloo ranks on host tensors: every gathered block is the right step's, on both ranks."""
    world, steps, B = 2, 7, 5
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_pipeline_worker, args=(world, port, steps, B, str(tmp_path)), nprocs=world, join=True)
    a = np.load(tmp_path / "pipe0.npy")
    b = np.load(tmp_path / "pipe1.npy")
    assert np.array_equal(a, b) and a.shape == (world * B * 6,)
