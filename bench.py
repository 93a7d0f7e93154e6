#!/usr/bin/env python3
"""bench.py — headline benchmark: simulated msgs/sec (+ histories/sec passing the checker), broadcast n=25.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one pass of the hot path over one batch: every rank simulates `--instances` (default 4096,
BASELINE.json configs[1]) independent broadcast test instances (25 nodes, grid topology, --rate 100,
--time-limit 20 + 10 s quiesce + 25 final reads — the invocation of doc/03-broadcast/02-performance.md:87)
to completion and runs the set-full checker over all emitted histories, everything resident in HBM.
Scaling is weak (fixed instances per GPU, distinct seeds per rank): the data path has no
collective; per step only the per-rank verdict/message totals (16 B) are all-reduced over RCCL.  After the
timed region the variable-length history gather to rank 0 (RCCL all_gather over xGMI) is measured once
and reported as `history_gather` (north_star asks for it; it is not part of `value`).

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def headline_config(E, seed):
    return E.test_config("broadcast", bin="broadcast-ff", node_count=25, rate=100, time_limit=20, latency=0,
                         latency_dist="constant", topology="grid", seed=seed, inbox_capacity=6)


def host_cores():
    """Threads worth starting: the CPUs this process may run on, capped by the container's CPU quota (cgroup v2 cpu.max /
    v1 cfs quota) — a 256-thread box that grants a pod ten cores' worth of time is a ten-core baseline."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: (t.strip(), open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()))):
        try:
            q, per = parse(open(path).read())
            if q != "max" and int(q) > 0:
                n = min(n, max(1, -(-int(q) // int(per))))
            break
        except (OSError, ValueError):
            continue
    return max(1, n)


def cpu_baseline(cfg, seconds):
    """The CPU oracle (a port with identical semantics) on the host cores, for ~`seconds` of wall time on all
    cores and on one core.  Every thread owns its output buffers and calls the C entry point directly in a
    loop (ctypes releases the GIL), so the threads do not serialise on Python allocations."""
    import concurrent.futures as cf
    import numpy as np
    import oracle_lib as O
    lib = O.load()
    cores = host_cores()
    chunk = 16

    def worker(tid, budget):
        rows = np.zeros((chunk, cfg.max_rows), dtype=O.OP_DT)
        pay = np.zeros((chunk, cfg.max_payload_words), dtype=np.uint32)
        stats = np.zeros(chunk, dtype=O.STATS_DT)
        meta = np.zeros(chunk, dtype=O.META_DT)
        msgs = inst = 0
        t_end = time.perf_counter() + budget
        while time.perf_counter() < t_end:
            rc = lib.oracle_run(C.byref(cfg), 20_000_000 + tid * 1_000_000 + inst, chunk, rows.ctypes.data, pay.ctypes.data,
                                stats.ctypes.data, meta.ctypes.data, None)
            assert rc == 0
            msgs += int(stats["all_send"].sum())
            inst += chunk
        return msgs, inst

    t0 = time.perf_counter()
    m1, i1 = worker(0, min(3.0, seconds))
    single = m1 / (time.perf_counter() - t0)
    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(cores) as ex:
        outs = list(ex.map(lambda k: worker(k + 1, seconds), range(cores)))
    dt = time.perf_counter() - t0
    msgs = sum(o[0] for o in outs)
    insts = sum(o[1] for o in outs)
    return {"value": msgs / dt, "unit": "msgs/s", "cores": cores, "kind": "port",
            "sample": f"{insts} instances of the same workload on {cores} threads for {dt:.1f} s; single core: {single:.3g} msgs/s over {i1} instances",
            "single_core_value": single}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--instances", type=int, default=4096, help="test instances per GPU per step")
    ap.add_argument("--cpu-sample", type=float, default=10.0, help="seconds of CPU-oracle work for cpu_baseline (0 = skip)")
    ap.add_argument("--seed", type=int, default=2026)
    ap.add_argument("--no-gather", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    from maelstrom_amd import build, engine as E
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if rank == 0:
        build.build(verbose=False)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # developer knobs for a dry run of the multi-rank path on a one-GPU box: every rank on device 0, gloo instead of RCCL
        if os.environ.get("MSIM_BENCH_ONE_DEVICE"):
            local_rank = 0
        torch.cuda.set_device(local_rank)
        backend = os.environ.get("MSIM_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        dist.barrier()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    cfg = headline_config(E, args.seed)
    eng = E.Engine(cfg, device=local_rank)
    n = args.instances

    def torch_view(ptr, nbytes, dtype, device):
        class _W:  # zero-copy view of engine-owned HBM through __cuda_array_interface__
            pass
        w = _W()
        w.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}
        return torch.as_tensor(w, device=device).view(dtype)

    acc = torch.zeros(5, dtype=torch.int64, device=dev)  # msgs, valid histories, flagged, rows, payload words

    from maelstrom_amd import ensemble as EN

    def step(k):
        first = k * world * n + EN.shard(world * n, rank, world)[0]  # distinct instances for every (step, rank)
        eng.run(first, n)     # simulate (blocking; kernel time from HIP events inside the library)
        eng.check()           # set-full over the HBM-resident histories
        db = eng.device_buffers()
        stats = torch_view(db.stats, db.stats_bytes, torch.int64, dev).view(-1, 6)
        meta = torch_view(db.meta, db.meta_bytes, torch.int32, dev).view(-1, 8)
        chk = torch_view(db.check, db.check_bytes, torch.int32, dev).view(-1, 17)
        acc.add_(torch.stack([stats[:, 0].sum(), (chk[:, 0] == 1).sum(), (meta[:, 2] != 0).sum(),
                              meta[:, 0].sum(dtype=torch.int64), meta[:, 1].sum(dtype=torch.int64)]))
        return eng.kernel_ms()

    for k in range(args.warmup):
        step(k)
    acc.zero_()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    sim_ms, chk_ms = [], []
    t0 = time.perf_counter()
    for k in range(args.steps):
        a, b = step(args.warmup + k)
        sim_ms.append(a)
        chk_ms.append(b)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    rows_tot, words_tot = int(acc[3]), int(acc[4])
    agg = acc[:3].clone()
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist:
        dist.all_reduce(agg)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    msgs_all, valid_all, flagged_all = int(agg[0]), int(agg[1]), int(agg[2])

    gather = None
    if not args.no_gather:
        gather = history_gather(eng, torch, dist, dev, world, rank, torch_view)

    if rank == 0:
        k = args.steps
        sim_avg = sum(sim_ms) / k
        chk_avg = sum(chk_ms) / k
        # algorithmic bytes of one sim launch on this rank: 16 B/row + 4 B/payload word + 48 B stats
        b_alg = (16.0 * rows_tot + 4.0 * words_tot) / k + 48.0 * n
        achieved = b_alg / (sim_avg * 1e-3) / 1e9
        out = {
            "metric": "simulated_msgs_per_sec (histories/sec passing checker in histories_per_sec)",
            "value": msgs_all / elapsed,
            "unit": "msgs/s",
            "n_gpus": world, "steps": k, "warmup": args.warmup,
            "ms_per_step": elapsed / k * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": "broadcast n=25 x %d instances/GPU (grid, rate 100/s, time-limit 20 s + 10 s quiesce + final reads, latency 0, fire-and-forget gossip)" % n,
                       "instances_per_gpu": n, "parallelism": "ensemble-dp%d" % world},
            "histories_per_sec": valid_all / elapsed,
            "histories_checked": n * k * world, "histories_valid": valid_all, "instances_flagged": flagged_all,
            "msgs_per_instance": msgs_all / (n * k * world),
            "kernel_ms": {"sim": sim_avg, "check": chk_avg},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                         "kernel": "sim_kernel_colo<BCAST_FF>", "algorithmic_bytes_per_launch": b_alg},
        }
        # HBM bytes per launch from the PMC passes of the committed profile (counters cannot be read inside this process):
        # FETCH_SIZE x 2 + WRITE_SIZE, KiB -> bytes (tools/rocpd_summary.py --traffic); null if the profile is absent or was
        # taken with a different batch size
        tj = os.path.join(ROOT, "profiles", "r01_headline_traffic.json")
        if os.path.exists(tj) and n == 4096:
            try:
                kern = json.load(open(tj))["kernels"]
                b = [v["hbm_bytes_per_dispatch"] for kname, v in kern.items() if "sim_kernel_colo" in kname and "hbm_bytes_per_dispatch" in v]
                if b:
                    out["roofline"]["traffic"] = b[0]
                    out["roofline"]["traffic_source"] = "profiles/r01_headline_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"
            except Exception:
                pass
        # the reference's only published figure for this path (README.md:39-42; other hardware, not reproduced here: no JVM)
        out["reference_published"] = {"value": 6.0e4, "unit": "msgs/s", "hardware": "48-way Xeon", "source": "jepsen-io/maelstrom README.md:39-42",
                                      "note": "quoted for scale only; vs_baseline stays null because BASELINE.json publishes no number for this metric"}
        if gather:
            out["history_gather"] = gather
        if args.cpu_sample > 0 and world == 1:   # the CPU leg is measured once, at N=1
            out["cpu_baseline"] = cpu_baseline(cfg, args.cpu_sample)
        print(json.dumps(out), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


def history_gather(eng, torch, dist, dev, world, rank, torch_view):
    """Variable-length history gather of the last batch over RCCL (maelstrom_amd.ensemble); at world=1 this
    is the on-device compaction only."""
    from maelstrom_amd import ensemble as EN
    db = eng.device_buffers()
    meta = torch_view(db.meta, db.meta_bytes, torch.int32, dev).view(-1, 8)
    rows = torch_view(db.rows, db.rows_bytes, torch.int32, dev).view(db.n_instances, db.max_rows, 4)
    pay = torch_view(db.payload, db.payload_bytes, torch.int32, dev).view(db.n_instances, db.max_payload_words)
    best = None
    for _ in range(2):  # first pass warms torch's kernels / RCCL channels
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        t0 = time.perf_counter()
        crow, cpay, nr, nw = EN.compact(rows, pay, meta)
        parts, nbytes = EN.gather_histories(crow, cpay, nr, nw, dist, world)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return {"bytes": nbytes, "ms": best * 1e3, "GB_per_s": nbytes / best / 1e9, "ranks": world}


if __name__ == "__main__":
    main()
