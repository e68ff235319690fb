#!/usr/bin/env python3
"""bench.py -- batched SQP-RTI solves/s of the path-tracking OCP on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched
under torch.distributed.run, one rank per GPU.  A *step* is one SQP-RTI solve of one batch of
B = 4096 independent OCP instances per GPU (BASELINE.json configs[1]: N = 40, dt = 0.35,
randomised states / cubic path coefficients, fp64), started from the cold initial guess
x_k = x0, u_k = 0 (src/AcadosSolver.cpp:78-92) so that every step does identical work.
Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

Extra keys: `roofline` (dominant kernel vs the fp64 vector peak; HBM view alongside),
`cpu_baseline` (the CPU oracle on the GPU box's host cores: a restatement, NOT acados), `warm`
(same measurement for warm-started RTI calls), `multi_gpu` (with/without the RCCL all-gather).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_VECTOR_PEAK_TFLOPS = 78.6   # = 1/2 of the 157.3 TF fp32 vector spec (MI355X_MICROARCH.md); also the fp64 MFMA peak
FP64_FMA_MEASURED_TFLOPS = 67.3  # chip-wide v_fma_f64 loop on this pool (tools/micro/fma64_peak.hip)
HBM_PEAK_GBPS = 8000.0
# HBM bytes per 4096-instance launch come from the PMC passes committed under profiles/ (rocprofv3
# cannot be read from inside this process): 2 x FETCH_SIZE (gfx950 tallies 128-B requests as 64 B;
# the factor is calibrated for wide streaming reads only, so this is an upper bound) + WRITE_SIZE,
# both in KiB as rocprofv3 reports them.
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r1e", "pmc_summary.json")


def pmc_traffic_bytes_per_launch():
    """Measured HBM bytes of one B = 4096, N = 40 launch of nmpc_solve_kernel, or None."""
    try:
        with open(PMC_SUMMARY) as f:
            d = json.load(f)
        return int((2.0 * d["FETCH_SIZE"]["mean_per_launch"] + d["WRITE_SIZE"]["mean_per_launch"]) * 1024)
    except (OSError, KeyError, ValueError):
        return None


def flops_per_solve(N, n_ipm):
    # F(N, n_ipm) = N (1400 + 2200 n_ipm + 16)
    return N * (1400.0 + 2200.0 * n_ipm + 16.0)


def bytes_per_solve(N, warm):
    cold = 8 * (6 + 6) + 8 * (8 * N + 6) + 16
    return cold + (8 * (8 * N + 6) if warm else 0)


class Hip:
    """Minimal ctypes view of the HIP runtime for event timing on the solver's stream."""

    def __init__(self):
        self.l = C.CDLL("libamdhip64.so")
        self.l.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
        self.l.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
        self.l.hipEventSynchronize.argtypes = [C.c_void_p]
        self.l.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
        self.l.hipEventDestroy.argtypes = [C.c_void_p]

    def event(self):
        e = C.c_void_p()
        assert self.l.hipEventCreate(C.byref(e)) == 0
        return e

    def record(self, e, stream):
        assert self.l.hipEventRecord(e, stream) == 0

    def elapsed_ms(self, a, b):
        assert self.l.hipEventSynchronize(b) == 0
        ms = C.c_float()
        assert self.l.hipEventElapsedTime(C.byref(ms), a, b) == 0
        return float(ms.value)


def cpu_baseline(N, seconds_target=12.0):
    """Time the CPU oracle (oracle/nmpc_oracle.c, 'port': a restatement, not acados/HPIPM) on a
    bounded sample of the same workload, one instance per thread on the host cores we may use."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as orc
    from mpc_local_planner_amd import capi, scenarios as sc
    c = capi.default_config(N, 0.35)
    ocfg = orc.config_like(c)
    x0, p = sc.config2(64)
    t = time.perf_counter()
    orc.rti_batch(ocfg, x0, p)
    per = (time.perf_counter() - t) / 64
    cores = max(1, min(int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or (os.cpu_count() or 1),
                       os.cpu_count() or 1, 16))
    chunk = max(8, int(seconds_target / per / cores / 4))
    nchunks = cores * 4
    total = chunk * nchunks
    x0, p = sc.config2(total)
    # single core
    n1 = max(64, int(min(total, 3.0 / per)))
    t = time.perf_counter()
    orc.rti_batch(ocfg, x0[:n1], p[:n1])
    single = n1 / (time.perf_counter() - t)

    def work(i):
        orc.rti_batch(ocfg, x0[i * chunk:(i + 1) * chunk], p[i * chunk:(i + 1) * chunk])
    t = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        list(ex.map(work, range(nchunks)))
    multi = total / (time.perf_counter() - t)
    return {"value": multi, "unit": "solves/s", "cores": cores, "kind": "port",
            "single_core_value": single,
            "sample": f"{total} cold SQP-RTI solves (N={N}) of the config-2 distribution, "
                      f"one instance per thread; CPU restatement (oracle/), not acados/HPIPM"}


def extras(capi, sc, hip, N):
    """Secondary measurements on one GPU (never `value`): the other BASELINE configs in short form."""
    ex = {}
    # config 1: ONE solve (the reference's use: one vehicle, 33 ms loop budget, src/main.cpp:14), cold RTI
    x1, p1 = sc.config2(1)
    lat = {}
    for n1 in (30, N):
        s = capi.BatchSolver(N=n1, dt=0.35, capacity=1)
        s.set_x0(x1); s.set_params(p1)
        s.solve(1); s.sync()
        ms = []
        for _ in range(20):
            s.reset(None, 1)
            e0, e1 = hip.event(), hip.event()
            hip.record(e0, s.stream()); s.solve(1); hip.record(e1, s.stream())
            ms.append(hip.elapsed_ms(e0, e1))
        t0 = time.perf_counter()
        for _ in range(20):
            s.reset(None, 1); s.solve(1); s.sync()
        lat[f"N{n1}"] = {"kernel_ms": float(np.median(ms)), "call_to_sync_ms": (time.perf_counter() - t0) / 20 * 1e3}
        s.close()
    ex["config1_single_solve"] = dict(lat, note="one instance, cold SQP-RTI: device time of the launch and host time "
                                                "from nmpc_reset+nmpc_solve to nmpc_sync")
    # config 3, one GPU's share: 131,072 instances in one launch (no tail effect of a small batch)
    Bl = 131072
    s = capi.BatchSolver(N=N, dt=0.35, capacity=Bl)
    x0, p = sc.config2(Bl)
    s.set_x0(x0); s.set_params(p)
    s.solve(Bl); s.sync()
    ms = []
    for _ in range(3):
        s.reset(None, Bl)
        e0, e1 = hip.event(), hip.event()
        hip.record(e0, s.stream()); s.solve(Bl); hip.record(e1, s.stream())
        ms.append(hip.elapsed_ms(e0, e1))
    ex["large_batch"] = {"instances": Bl, "kernel_ms": float(np.median(ms)), "solves_per_s": Bl / (np.median(ms) * 1e-3),
                         "note": "cold SQP-RTI, config-2 distribution, per-GPU share of config 3"}
    s.close()
    # config 4: full SQP (<= 20 iterations, step tolerance 1e-6), N = 100, bounds active
    B4 = 4096
    s = capi.BatchSolver(N=100, dt=0.35, capacity=B4)
    x0, p = sc.config4(B4)
    s.set_x0(x0); s.set_params(p)
    e0, e1 = hip.event(), hip.event()
    hip.record(e0, s.stream()); s.solve(B4, mode=capi.MODE_SQP, max_iter=20, tol=1e-6); hip.record(e1, s.stream())
    t4 = hip.elapsed_ms(e0, e1)
    g = s.get(B4)
    hist = np.bincount(g["sqp_iters"], minlength=21)[:21].tolist()
    ex["config4_sqp_N100"] = {"instances": B4, "kernel_ms": t4, "converged_frac": float((g["status"] == 0).mean()),
                              "sqp_iters_hist": hist, "mean_ipm_iters": float(g["ipm_iters"].mean()),
                              "qp_solves_per_s": float(g["sqp_iters"].sum() / (t4 * 1e-3))}
    s.close()
    # config 5: closed loop, 65,536 vehicles x 200 ticks, per-instance weights / v_ref
    B5, ticks = 65536, 200
    track = sc.test_track()
    ws, pv = sc.cars_on_track(B5, track)
    rng = np.random.default_rng(5)
    s = capi.BatchSolver(N=N, dt=0.35, capacity=B5)
    s.set_track(track)
    s.set_weights(np.array(s.cfg.W[:])[None, :] * rng.uniform(0.5, 2.0, size=(B5, 7)))
    s.closed_loop(ws, pv, 2)
    t = time.perf_counter()
    s.closed_loop(ws, pv, ticks)
    dt5 = time.perf_counter() - t
    last = s.get_controls(B5)[:, 5].astype(np.int64)      # status + 16 * iterations of the last tick
    ex["config5_closed_loop"] = {"vehicles": B5, "ticks": ticks, "wall_s": dt5,
                                 "vehicle_ticks_per_s": B5 * ticks / dt5,
                                 "last_tick_ok_frac": float((last % 16 == 0).mean()),
                                 "last_tick_mean_ipm_iters": float((last // 16).mean()),
                                 "note": "front-end + SQP-RTI + plant per tick on the device, host transfers of the "
                                         "states at both ends included"}
    s.close()
    return ex


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4096, help="instances per GPU")
    ap.add_argument("--horizon", type=int, default=40)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from mpc_local_planner_amd import capi, scenarios as sc, sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    # NMPC_BENCH_FORCE_DIST=1 runs the collective path with a single rank too (rehearsal on a one-GPU box)
    dist_on = world > 1 or os.environ.get("NMPC_BENCH_FORCE_DIST") == "1"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    N, B = args.horizon, args.batch
    solver = capi.BatchSolver(N=N, dt=0.35, capacity=B, device=local_rank)
    start, stop = sharding.shard_range(rank, world, world * B)   # this rank's block of the global batch
    x0, p = sc.config2(stop - start, start=start)
    solver.set_x0(x0)
    solver.set_params(p)                         # inputs now resident in HBM
    hip = Hip()
    stream = solver.stream()

    # control records of all ranks (RCCL all-gather over xGMI), zero-copy view of the handle's buffer
    class _Arr:
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False),
                                             "version": 2}
    dev = torch.device("cuda", local_rank)
    # The gather of step s overlaps the solve of step s+1: the solves write their control records
    # alternately into two device buffers (nmpc_set_controls_device), a side stream runs the
    # collective on the buffer a solve has just filled, and the solver's stream only waits for the
    # collective that last read the buffer it is about to overwrite.  No host synchronisation and
    # no copy per step.
    solve_stream = torch.cuda.ExternalStream(int(stream), device=dev)
    comm_stream = torch.cuda.Stream(device=dev)
    records = [torch.zeros(B * 6, dtype=torch.float64, device=dev) for _ in range(2)]
    gathered = [torch.empty(world * B * 6, dtype=torch.float64, device=dev) for _ in range(2)]
    solved = [torch.cuda.Event() for _ in range(2)]
    gather_done = [torch.cuda.Event() for _ in range(2)]
    n_gathers = [0]

    def before_gathered_solve():
        i = n_gathers[0] & 1
        if n_gathers[0] >= 2:
            solve_stream.wait_event(gather_done[i])         # the collective of two steps ago has read records[i]
        solver.set_controls_device(records[i].data_ptr())

    def enqueue_gather():
        i = n_gathers[0] & 1
        solved[i].record(solve_stream)
        comm_stream.wait_event(solved[i])
        with torch.cuda.stream(comm_stream):
            dist.all_gather_into_tensor(gathered[i], records[i])
            gather_done[i].record(comm_stream)
        n_gathers[0] += 1

    def barrier():
        solver.sync()
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()

    def run(steps, cold, gather, events=None):
        for s in range(steps):
            if cold:
                solver.reset(None, B)
            if gather and dist_on:
                before_gathered_solve()
            if events is not None:
                hip.record(events[2 * s], stream)
            solver.solve(B)
            if events is not None:
                hip.record(events[2 * s + 1], stream)
            if gather and dist_on:
                enqueue_gather()

    def timed(steps, cold, gather, with_events=False):
        events = [hip.event() for _ in range(2 * steps)] if with_events else None
        barrier()
        t0 = time.perf_counter()
        run(steps, cold, gather, events)
        barrier()
        dt = time.perf_counter() - t0
        kms = None
        if events:
            kms = float(np.mean([hip.elapsed_ms(events[2 * s], events[2 * s + 1]) for s in range(steps)]))
        if dist_on:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, kms

    gather = dist_on
    run(args.warmup, True, gather)
    dt_main, kernel_ms = timed(args.steps, True, gather, with_events=True)
    if dist_on:
        # the last gather must hold this rank's records of the last gathered step in this rank's block
        last = gathered[(n_gathers[0] - 1) & 1].reshape(world, B, 6)
        if not torch.equal(last[rank], records[(n_gathers[0] - 1) & 1].reshape(B, 6)):
            raise SystemExit("all-gather returned records that differ from this rank's own")
        if not np.array_equal(solver.get_controls(B), records[(n_gathers[0] - 1) & 1].reshape(B, 6).cpu().numpy()):
            raise SystemExit("control records did not land in the caller's device buffer")
        solver.set_controls_device(None)
    g = solver.get(B)
    n_ipm_cold = float(g["ipm_iters"].mean())
    ok_cold = float((g["status"] == 0).mean())

    # secondary measurements (not `value`): warm-started RTI; multi-GPU without the gather
    run(2, False, False)
    dt_warm, kernel_ms_warm = timed(args.steps, False, False, with_events=True)
    gw = solver.get(B)
    n_ipm_warm = float(gw["ipm_iters"].mean())
    dt_nogather = None
    if dist_on:
        run(2, True, False)
        dt_nogather, _ = timed(args.steps, True, False)

    total_solves = world * B * args.steps
    value = total_solves / dt_main
    out = {
        "metric": "batched NMPC SQP-RTI solves/sec (N=40, fp64)",
        "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt_main / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"config 2: {B} path-tracking OCP instances per GPU, N={N}, dt=0.35, "
                               "cold SQP-RTI (one QP) per step, seed 0x5EED0001",
                   "batch_per_gpu": B, "horizon": N, "parallelism": f"dp{world} (instances sharded, "
                   "RCCL all-gather of control records)" if world > 1 else "single GPU"},
        "mean_ipm_iters": n_ipm_cold, "converged_frac": ok_cold,
    }
    if kernel_ms:
        fl = flops_per_solve(N, n_ipm_cold) * B
        by = bytes_per_solve(N, False) * B
        tf = fl / (kernel_ms * 1e-3) / 1e12
        out["roofline"] = {
            "bound": "mfma", "kernel": "nmpc_solve_kernel", "achieved": tf, "peak": FP64_VECTOR_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": tf / FP64_VECTOR_PEAK_TFLOPS,
            "traffic": pmc_traffic_bytes_per_launch() if (B == 4096 and N == 40) else None,
            "note": "compute roof for f64: the dense fp64 MFMA peak and the fp64 vector-FMA peak are the same "
                    "78.6 TFLOP/s on MI355X; this kernel's flops are vector FMAs (per-instance matrices are 6x6..9x9, "
                    "no MFMA-shaped work).  Measured binding resources are the LDS pipe and serial-recursion "
                    "latency, see DESIGN.md section 5; the HBM view is in 'hbm'",
            "kernel_ms": kernel_ms, "flops_per_solve": flops_per_solve(N, n_ipm_cold),
            "peak_measured_fma64": FP64_FMA_MEASURED_TFLOPS,
            "hbm": {"bound": "hbm", "achieved": by / (kernel_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s", "frac": by / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                    "bytes_per_solve": bytes_per_solve(N, False)},
        }
    out["warm"] = {"value": total_solves / dt_warm, "ms_per_step": 1e3 * dt_warm / args.steps,
                   "kernel_ms": kernel_ms_warm, "mean_ipm_iters": n_ipm_warm}
    if dt_nogather:
        out["multi_gpu"] = {"with_gather": value, "without_gather": total_solves / dt_nogather}
    if rank == 0 and world == 1 and not args.no_extras:
        out["extras"] = extras(capi, sc, hip, N)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(N)
    if rank == 0:
        print(json.dumps(out))
    solver.close()
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
