"""Multi-GPU sharding of a batch of OCP instances.

Instances are independent, so the batch is cut into contiguous blocks, one per rank, with no
collective on the data path.  The only exchange is one all-gather of the fixed-size control
records (u0[2], x1.delta, x1.throttle, cost, status+16*iters -- include/nmpc.h:
nmpc_get_controls) at the end of a step; on GPUs it runs over RCCL/xGMI straight from the
handle's device buffer, on CPU (tests) over gloo.
"""
import numpy as np

RECORD = 6   # doubles per control record


def shard_range(rank, world, total):
    """Contiguous block [start, stop) of `total` instances owned by `rank` (blocks differ by at
    most one instance, lower ranks take the remainder)."""
    if not (0 <= rank < world) or total < 0:
        raise ValueError("bad rank/world/total")
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_sizes(world, total):
    return [shard_range(r, world, total)[1] - shard_range(r, world, total)[0] for r in range(world)]


def gather_controls(ctrl_local, total=None, group=None):
    """All-gather the [B_local, 6] control records of every rank into [total, 6] on every rank.
    `ctrl_local` is a torch tensor (CPU with gloo, CUDA with nccl = RCCL).  Equal shard sizes use
    one all_gather_into_tensor; ragged shards are padded to the largest one."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_local = ctrl_local.shape[0]
    if total is None:
        total = n_local * world
    sizes = shard_sizes(world, total)
    if sizes[rank] != n_local:
        raise ValueError(f"rank {rank} holds {n_local} records, expected {sizes[rank]}")
    nmax = max(sizes)
    flat = ctrl_local.reshape(-1)
    if n_local < nmax:
        flat = torch.cat([flat, flat.new_zeros((nmax - n_local) * RECORD)])
    out = flat.new_empty(world * nmax * RECORD)
    dist.all_gather_into_tensor(out, flat.contiguous(), group=group)
    out = out.reshape(world, nmax, RECORD)
    if min(sizes) == nmax:
        return out.reshape(total, RECORD)
    return torch.cat([out[r, :sizes[r]] for r in range(world)], dim=0)


class HostSync:
    """Ordering primitives of ControlGatherPipeline for host tensors (gloo, tests): the producer runs
    synchronously, a collective is an async work handle."""

    def wait_collective(self, token):
        if token is not None:
            token.wait()

    def produced(self):
        return None

    def launch(self, produced_token, fn):
        return fn(True)          # fn(async_op) -> work handle

    def finish(self, token):
        """Block the host until the collective behind `token` has completed."""
        token.wait()


class CudaSync:
    """Ordering primitives for device tensors: the producer (the solver) runs on `solve_stream`, the
    collectives on a side stream; events order them, the host never waits."""

    def __init__(self, solve_stream, device):
        import torch
        self.torch = torch
        self.solve_stream = solve_stream
        self.comm_stream = torch.cuda.Stream(device=device)

    def wait_collective(self, token):
        if token is not None:
            self.solve_stream.wait_event(token)

    def produced(self):
        ev = self.torch.cuda.Event()
        ev.record(self.solve_stream)
        return ev

    def launch(self, produced_token, fn):
        self.comm_stream.wait_event(produced_token)
        with self.torch.cuda.stream(self.comm_stream):
            fn(False)
            done = self.torch.cuda.Event()
            done.record(self.comm_stream)
        return done

    def finish(self, token):
        """Block the host until the collective behind `token` has completed (Event.wait() would only make
        the current stream wait: the host must not read the gathered block before this returns)."""
        token.synchronize()


class ControlGatherPipeline:
    """All-gather of the control records of step s overlapped with the solve of step s+1.

    The solves write their records alternately into two buffers (`records[0/1]`, on the GPU handed to
    the handle with nmpc_set_controls_device); the collective of a step runs as soon as that step's
    solve has filled its buffer, and a solve only waits for the collective that last READ the buffer it
    is about to overwrite (two steps back).  No host synchronisation and no copy per step.

        i = pipe.begin_step()        # buffer index the next solve must write into (waits if needed)
        ... enqueue the solve writing pipe.records[i] ...
        pipe.end_step()              # enqueue the all-gather of pipe.records[i] into pipe.gathered[i]
        pipe.finish()                # wait for everything in flight; pipe.last() = newest gathered block
    """

    def __init__(self, records, gathered, sync, group=None):
        assert len(records) == 2 and len(gathered) == 2
        self.records, self.gathered, self.sync, self.group = records, gathered, sync, group
        self.tokens = [None, None]        # collective still reading records[i] / writing gathered[i]
        self.steps = 0
        self._open = None

    def begin_step(self):
        assert self._open is None, "begin_step twice without end_step"
        i = self.steps & 1
        self.sync.wait_collective(self.tokens[i])      # the gather of two steps ago has read records[i]
        self.tokens[i] = None
        self._open = i
        return i

    def end_step(self):
        import torch.distributed as dist
        i = self._open
        assert i is not None, "end_step without begin_step"
        produced = self.sync.produced()
        self.tokens[i] = self.sync.launch(
            produced, lambda async_op: dist.all_gather_into_tensor(self.gathered[i], self.records[i],
                                                                   group=self.group, async_op=async_op))
        self._open = None
        self.steps += 1

    def finish(self):
        for i in (0, 1):
            tok = self.tokens[i]
            if tok is None:
                continue
            self.sync.finish(tok)
            self.tokens[i] = None

    def last(self):
        """(buffer index, gathered tensor) of the newest completed step; call finish() first."""
        assert self.steps > 0
        i = (self.steps - 1) & 1
        return i, self.gathered[i]
