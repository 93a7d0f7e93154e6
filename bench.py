#!/usr/bin/env python3
"""bench.py — headline benchmark: simulated msgs/sec (+ histories/sec passing the checker), broadcast n=25.

    python bench.py --gpus N --steps K --warmup W [--config cfg2|cfg4]
    (N>1: either under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...` — the driver's
    way — or plainly as `python bench.py --gpus N`, which starts those N ranks itself; it refuses to run on fewer devices than N
    and never reports an n_gpus other than --gpus.)

One "step" = one pass of the hot path over one batch (steps are pipelined: `--in-flight` of them, default 3 for cfg2, are on the GPU
together, each on its own engine context and HIP stream; all K start and end inside the timed region): every rank simulates `--instances` (default 4096,
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


def cfg4_config(E, seed):
    """BASELINE.json configs[3]: lin-kv over 5-node Raft, concurrency 10 (core.clj:111), rate 30/s, 60 s (doc/06-raft/04-committing.md:418)."""
    return E.test_config("lin-kv", bin="raft", node_count=5, rate=30, time_limit=60, seed=seed)


CONFIGS = {
    # name: (config builder, instances over the whole job or None = --instances per GPU, description, dominant kernel)
    "cfg2": (headline_config, None, "broadcast n=25 x %d instances/GPU (grid, rate 100/s, time-limit 20 s + 10 s quiesce + final reads, latency 0, fire-and-forget gossip)",
             "sim_kernel_duo<LAT0, DEG4> (duo.hip: two clusters per wavefront)", "sim_kernel_duo", "r06f_headline_counters.json"),
    "cfg4": (cfg4_config, 65536, "lin-kv over 5-node Raft x %d instances/GPU (65536 over the job; concurrency 10, rate 30/s, time-limit 60 s, latency 0), histories gathered to rank 0 over RCCL",
             "raft4_kernel<> (raft4.hip: four clusters per wavefront)", "raft4_kernel", "r06f_cfg4_counters.json"),
}


def respawn_if_needed(args):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks (one per GPU) under it and become that job."""
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None:
        if int(env_world) != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={env_world} ranks; n_gpus must be what was asked for")
        return
    if args.gpus <= 1:
        return
    import socket
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus and not os.environ.get("MSIM_BENCH_ONE_DEVICE"):
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} HIP device(s) visible; one rank per GPU is the only layout (SURVEY.md §8e)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush(); sys.stderr.flush()
    os.execv(sys.executable, cmd)


def supervise(argv):
    """N = 1 outside a launcher: the measurement runs in a child process (`--worker`) and this process only forwards its one JSON line.
    A GPU memory-access fault aborts the process it happens in (the round-4 driver run ended that way 2 s in — BENCH_r04.json holds no
    stage marker, so where is not known; hundreds of repeats of the same command did not fault, and the hunt under fenced slabs is
    tools/guard_sweep.sh / DESIGN.md §7), and only a parent can still report what had been measured by then.  The child notes its progress in a status file (MSIM_BENCH_STATUS): the headline once the timed
    region is over, then every secondary leg it enters.  If it dies, it is started again — without the leg it died in — at most
    twice; the line that is finally printed says so in `attempts`.  If no attempt gets through its legs, the headline of the last
    attempt that measured one is printed with the legs marked absent.  Nothing here touches the timed region."""
    import subprocess
    import tempfile
    skip, failures, partial = [], [], None
    for attempt in range(3):
        fd, status = tempfile.mkstemp(prefix="msim_bench_", suffix=".status")
        os.close(fd)
        env = dict(os.environ, MSIM_BENCH_STATUS=status)
        cmd = [sys.executable, os.path.abspath(__file__)] + list(argv) + ["--worker"] + skip
        try:   # a child hung in the timed region or in teardown (a GPU fault can hang as well as abort) must not hold the parent for ever
            r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, timeout=float(os.environ.get("MSIM_BENCH_CHILD_TIMEOUT", "900")))
        except subprocess.TimeoutExpired as ex:   # (subprocess.run has killed the child by now)
            r = subprocess.CompletedProcess(cmd, -9, stdout=ex.stdout or b"")
            sys.stderr.write(f"[bench] attempt {attempt + 1}: the measuring process did not finish in {ex.timeout:.0f} s and was killed\n")
        lines = [ln for ln in r.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
        notes = []
        try:
            with open(status) as f:
                notes = [json.loads(ln) for ln in f if ln.strip()]
        except (OSError, ValueError):
            pass
        finally:
            try:
                os.unlink(status)
            except OSError:
                pass
        if lines:   # the line is out: the measurement is complete whatever happened to the process afterwards (a teardown that dies is noted, not repeated)
            out = json.loads(lines[-1])
            if failures or r.returncode != 0:
                out["attempts"] = {"n": attempt + 1, "failed": failures}
                if r.returncode != 0:
                    out["attempts"]["exit_after_the_line"] = r.returncode
            print(json.dumps(out), flush=True)
            return 0
        leg = next((x["name"] for x in reversed(notes) if x.get("stage") == "leg"), None)
        head = next((x["line"] for x in reversed(notes) if x.get("stage") == "headline"), None)
        failures.append({"attempt": attempt + 1, "returncode": r.returncode, "died_in": leg or ("timed region" if head is None else "after the timed region")})
        sys.stderr.write(f"[bench] attempt {attempt + 1} ended with return code {r.returncode} ({failures[-1]['died_in']})\n")
        if head is not None:
            partial = head
        if r.returncode >= 0 and r.returncode not in (134, 139):   # an ordinary error exit (bad flags, no device): the same again would end the same
            break
        flag = {"history_gather": ["--no-gather"], "incl_fetch": ["--no-fetch"], "cpu_baseline": ["--cpu-sample", "0"]}.get(leg)
        if flag and flag[0] not in skip:
            skip += flag
    if partial is not None:
        partial["attempts"] = {"n": len(failures), "failed": failures, "note": "no attempt finished its secondary legs; this is the headline of the last attempt that measured one"}
        print(json.dumps(partial), flush=True)
        return 0
    return failures[-1]["returncode"] if failures and failures[-1]["returncode"] > 0 else 1


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
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--instances", type=int, default=0, help="test instances per GPU per step (default: 4096 for cfg2; 65536 / --gpus for cfg4)")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="cfg2", help="BASELINE.json config: cfg2 = the headline (broadcast n=25), cfg4 = lin-kv over Raft, 65536 instances over the job + RCCL history gather")
    ap.add_argument("--cpu-sample", type=float, default=10.0, help="seconds of CPU-oracle work for cpu_baseline (0 = skip)")
    ap.add_argument("--seed", type=int, default=2026)
    ap.add_argument("--in-flight", type=int, default=0, help="steps in flight together, one engine context and HIP stream each (default: 3 for cfg2, 1 for cfg4)")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--no-fetch", action="store_true", help="skip the PCIe-inclusive leg (value_incl_fetch)")
    ap.add_argument("--worker", action="store_true", help=argparse.SUPPRESS)   # the measuring process under supervise()
    args = ap.parse_args()
    respawn_if_needed(args)
    if args.gpus == 1 and not args.worker and os.environ.get("WORLD_SIZE") is None and not os.environ.get("MSIM_BENCH_INPROCESS"):
        sys.exit(supervise(sys.argv[1:]))

    # ONE JSON line on stdout: libraries underneath (RCCL's version banner at communicator set-up, gloo's connection notes) write
    # to file descriptor 1 — for the duration of the run it points at stderr; the result goes to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    from maelstrom_amd import build, engine as E
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, (world, args.gpus)
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

    make_cfg, job_instances, workload_text, kernel_text, kernel_key, counters_file = CONFIGS[args.config]
    cfg = make_cfg(E, args.seed)
    eng = E.Engine(cfg, device=local_rank)
    n = args.instances or (job_instances // world if job_instances else 4096)

    def torch_view(ptr, nbytes, dtype, device):
        class _W:  # zero-copy view of engine-owned HBM through __cuda_array_interface__
            pass
        w = _W()
        w.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}
        return torch.as_tensor(w, device=device).view(dtype)

    acc = torch.zeros(5, dtype=torch.int64, device=dev)  # msgs, valid histories, flagged, rows, payload words

    from maelstrom_amd import ensemble as EN

    # Steps are pipelined: D engine contexts, each with its own HIP stream, hold the batches of steps k, k+1, .. k+D-1 in flight together.
    # One launch of 4096 clusters is 2048 wavefronts — two per SIMD — and every one of them is a chain of dependent LDS round trips that
    # nothing hides; with the next steps' wavefronts resident beside them the SIMDs have something to issue while a wavefront waits, and
    # the tail of a launch (the slowest wavefront sets its duration) is filled by the head of the next (tools/cfg2_overlap.py:
    # profiles/r06_cfg2_overlap.jsonl).  Every step is still one whole pass — simulate n instances, check every history — and all K of
    # them start and end inside the timed region; --in-flight 1 is the one-batch-at-a-time run of the earlier rounds.
    depth = max(1, min(args.in_flight or (3 if args.config == "cfg2" else 1), args.steps))
    engs = [eng] + [E.Engine(cfg, device=local_rank) for _ in range(depth - 1)]
    pending = [None] * depth      # per context: the step whose batch it holds
    sim_ms, chk_ms = [], []

    def retire(j):
        e = engs[j]
        e.check()             # waits for this context's simulation (same stream), then the workload checker over the HBM-resident histories (cfg2: set-full; cfg4: per-key linearizability)
        db = e.device_buffers()
        stats = torch_view(db.stats, db.stats_bytes, torch.int64, dev).view(-1, 6)
        meta = torch_view(db.meta, db.meta_bytes, torch.int32, dev).view(-1, 8)
        chk = torch_view(db.check, db.check_bytes, torch.int32, dev).view(-1, 17)
        acc.add_(torch.stack([stats[:, 0].sum(), (chk[:, 0] == 1).sum(), (meta[:, 2] != 0).sum(),
                              meta[:, 0].sum(dtype=torch.int64), meta[:, 1].sum(dtype=torch.int64)]))
        a, b = e.kernel_ms()  # this launch's duration from the HIP events on its own stream (other launches were in flight beside it)
        sim_ms.append(a)
        chk_ms.append(b)
        pending[j] = None

    def step(k):
        j = k % depth
        if pending[j] is not None:
            retire(j)
        first = k * world * n + EN.shard(world * n, rank, world)[0]  # distinct instances for every (step, rank)
        engs[j].run_async(first, n)   # simulate, on the context's own stream
        pending[j] = k

    def drain():
        for k, j in sorted((k, j) for j, k in enumerate(pending) if k is not None):
            retire(j)

    for k in range(args.warmup):
        step(k)
    drain()
    acc.zero_()
    del sim_ms[:], chk_ms[:]
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
    drain()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    rows_tot, words_tot = int(acc[3]), int(acc[4])
    # the workload checker alone on the chip (outside the timed region): five launches over the batch the first context still holds —
    # check_kernel / lin_check_kernel is the one kernel of a step that IS bandwidth work (it reads every history once), so its own
    # roofline fraction is reported beside the simulation kernel's (roofline.checker)
    chk_solo = []
    try:
        for _ in range(5):
            engs[0].check()
            chk_solo.append(engs[0].kernel_ms()[1])
    except Exception:
        chk_solo = []
    agg = acc[:3].clone()
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist:
        dist.all_reduce(agg)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    msgs_all, valid_all, flagged_all = int(agg[0]), int(agg[1]), int(agg[2])

    # The two secondary measurements involve collectives of their own (the RCCL history gather) and a second engine context; they run
    # under a watchdog so that whatever happens to them the headline line above them is printed: a leg that does not finish is
    # reported as such, and the process then leaves without the final barrier (the other ranks are in the same position).
    import threading
    stuck = []
    status_path = os.environ.get("MSIM_BENCH_STATUS")

    def note(rec):   # progress notes for supervise(): which stage / leg this process is in, and the headline once it is measured
        if status_path and rank == 0:
            with open(status_path, "a") as f:
                f.write(json.dumps(rec) + "\n")
        if rec.get("name") and os.environ.get("MSIM_BENCH_TEST_ABORT_IN") == rec["name"]:   # test hook: die like a GPU fault does (tests/test_bench_line_gpu.py)
            os.abort()

    def guarded(label, seconds, fn):
        box = {}

        def run():
            try:
                torch.cuda.set_device(local_rank)   # the current device is per thread
                box["v"] = fn()
            except Exception as ex:   # reported, not raised: the headline measurement stands
                box["e"] = f"{type(ex).__name__}: {ex}"
        th = threading.Thread(target=run, daemon=True)
        th.start()
        th.join(seconds)
        if th.is_alive():
            stuck.append(label)
            return None, f"{label}: not finished after {seconds:.0f} s"
        return box.get("v"), box.get("e")

    if rank == 0:
        k = args.steps
        sim_avg = sum(sim_ms) / k
        chk_avg = sum(chk_ms) / k
        # algorithmic bytes of one sim launch on this rank: 16 B/row + 4 B/payload word + 48 B stats
        b_alg = (16.0 * rows_tot + 4.0 * words_tot) / k + 48.0 * n
        # `depth` launches of the kernel share the chip: a launch takes sim_avg from start to end while, on average, in_flight launches
        # (= the launches' summed durations / the timed region) run beside each other — HBM sees in_flight launches' bytes per sim_avg
        in_flight = min(float(depth), max(1.0, sum(sim_ms) * 1e-3 / elapsed)) if depth > 1 else 1.0
        achieved = in_flight * b_alg / (sim_avg * 1e-3) / 1e9
        out = {
            "metric": "simulated_msgs_per_sec (histories/sec passing checker in histories_per_sec)",
            "value": msgs_all / elapsed,
            "unit": "msgs/s",
            "n_gpus": world, "steps": k, "warmup": args.warmup,
            "ms_per_step": elapsed / k * 1e3,
            "higher_is_better": True, "scaling": "strong" if job_instances and not args.instances else "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": workload_text % n, "name": args.config,
                       "instances_per_gpu": n, "parallelism": "ensemble-dp%d" % world},
            # the metric as SURVEY.md §8(d)(i) words it: simulate + check + the histories' way to host memory (PCIe) in the timed region
            "value_incl_fetch": None, "incl_fetch": None,   # filled in by the PCIe-inclusive leg below
            "histories_per_sec": valid_all / elapsed,
            # the checkers behind histories_per_sec restate [upstream] Jepsen / Knossos / Elle from their published descriptions: no JVM
            # here to pin them against.  Reference-held vectors: pn_counter_test.clj (the counter checker) and the
            # anomalies doc/05-datomic prints (the list-append checker, tests/test_elle_reference_vectors.py); set-full (this line's
            # checker) and the linearizability search are held to the runs of the real checkers the tutorial prints — closing reads with
            # the stable / stale / lost elements of doc/03-broadcast/01-broadcast.md:388-430 and 02-performance.md:282-301, the
            # linearizable run and the write-2-read-4 pair of doc/06-raft/01-key-value.md:131-195 (tests/test_checker_reference_vectors.py)
            "checker_parity": "partial (doc vectors) for set-full, linearizability and list-append; pinned (pn_counter_test.clj) for the counters",
            "histories_checked": n * k * world, "histories_valid": valid_all, "instances_flagged": flagged_all,
            "msgs_per_instance": msgs_all / (n * k * world),
            "kernel_ms": {"sim": sim_avg, "check": chk_avg, "steps_in_flight": depth,
                          "note": "per launch, HIP events on the launch's own stream; with steps_in_flight > 1 a launch shares the chip with the next steps' launches, so sim exceeds ms_per_step"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                         "kernel": kernel_text, "algorithmic_bytes_per_launch": b_alg,
                         "launches_in_flight": in_flight, "achieved_per_launch": b_alg / (sim_avg * 1e-3) / 1e9,
                         "how": "achieved = launches_in_flight x algorithmic_bytes_per_launch / the kernel's average launch duration (kernel_ms.sim); launches_in_flight = summed launch durations / timed region"},
        }
        if chk_solo:
            cs = sorted(chk_solo)[len(chk_solo) // 2]
            out["roofline"]["checker"] = {"kernel": "check_kernel (csrc/checker.hip: set-full)" if args.config == "cfg2" else "lin_check_kernel (csrc/lin_check_dev.hip)",
                                          "bound": "hbm", "ms": cs, "achieved": b_alg / (cs * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                          "frac": b_alg / (cs * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                          "how": "the same algorithmic bytes (the checker reads every history row and payload word once) / the median of five launches alone on the chip, after the timed region"}
        # HBM bytes per launch and the instruction-issue picture from the PMC passes of the committed profile (counters cannot be
        # read inside this process): FETCH_SIZE x 2 + WRITE_SIZE, KiB -> bytes; SQ_* per launch (tools/profile_headline.sh ->
        # tools/rocpd_summary.py --counters).  null if the profile is absent or was taken with a different batch size.
        cj = os.path.join(ROOT, "profiles", counters_file)
        if os.path.exists(cj) and n == (4096 if args.config == "cfg2" else 8192):
            try:
                kern = json.load(open(cj))["kernels"]
                kd = [v for kname, v in kern.items() if kernel_key in kname]
                if kd:
                    kd, c = kd[0], kd[0]["counters_per_dispatch"]
                    if "hbm_bytes_per_dispatch" in kd:
                        out["roofline"]["traffic"] = kd["hbm_bytes_per_dispatch"]
                        out["roofline"]["traffic_source"] = f"profiles/{counters_file} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"
                    # The kernel is not bound by HBM (frac above): what bounds it is VALU issue.  A VALU instruction of a 64-lane wavefront
                    # occupies its SIMD's 16-lane vector pipe for 4 cycles, so the pipes of the chip are busy for 4 x SQ_INSTS_VALU cycles out
                    # of n_SIMD x kernel cycles (SQ_BUSY_CYCLES counts every cycle once per shader engine: 32 of them on MI355X).
                    insts = sum(c.get(k, 0.0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH", "SQ_INSTS_VMEM_WR", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SMEM"))
                    n_simd, n_se = 1024.0, 32.0
                    kcycles = c.get("SQ_BUSY_CYCLES", 0.0) / n_se
                    live_lanes = 2 * cfg.n_nodes if args.config == "cfg2" else 4 * (cfg.n_nodes + cfg.concurrency)
                    out["roofline"]["secondary"] = {
                        "bound": "VALU issue: in an undisturbed launch (what the PMC passes see: counters serialise launches) the vector pipes are busy valu_issue_frac of the kernel's cycles "
                                 "with lane_utilisation of the lanes carrying a cluster's endpoints, and one batch (%d wavefronts on %d SIMDs) gives every SIMD %.1f wavefronts to hide LDS / ds_bpermute "
                                 "round trips with; the timed region keeps %d steps in flight, so that the next launches' wavefronts fill those stalls (valu_issue_frac_in_flight: the same instruction "
                                 "count over the measured ms_per_step)" % (int(kd.get("wavefronts") or 0), int(n_simd), (kd.get("wavefronts") or 0) / n_simd, depth),
                        "source": f"profiles/{counters_file} (rocprofv3 --pmc SQ_* passes of the same kernel at the same batch)",
                        "valu_issue_frac": 4.0 * c.get("SQ_INSTS_VALU", 0.0) / (n_simd * kcycles) if kcycles else None,
                        # the same instructions over the time a step takes with the pipeline full (clock from the profile: kernel cycles / profiled kernel time)
                        "valu_issue_frac_in_flight": (4.0 * c.get("SQ_INSTS_VALU", 0.0) / (n_simd * (kcycles / kd["avg_ms"]) * (elapsed / k * 1e3))) if kcycles and kd.get("avg_ms") else None,
                        "lane_utilisation": live_lanes / 64.0,
                        "kernel_cycles": kcycles,
                        "wavefronts": kd.get("wavefronts"), "lds_bytes_per_wavefront": kd.get("lds_bytes"),
                        "insts_per_launch": {k[9:].lower(): c[k] for k in sorted(c) if k.startswith("SQ_INSTS_")},
                        "insts_per_message": insts / (msgs_all / (k * world)) if msgs_all else None,
                        "wave_cycles_per_launch": c.get("SQ_WAVE_CYCLES"),
                        "frac_of_wave_cycles": kd.get("derived"),
                        "profiled_kernel_ms": kd.get("avg_ms"),
                        "batch_sweep": "profiles/r03q_headline_batch_sweep.jsonl + r03q_bench.json (4096 / 8192 / 16384 instances: 8.9 / 20.1 / 35.7 ms of simulation, 2.6e10 msgs/s incl. the checker at 16384)" if args.config == "cfg2" else None,
                    }
            except Exception:
                pass
        # the reference's only published figure for this path (README.md:39-42; other hardware, not reproduced here: no JVM)
        out["reference_published"] = {"value": 6.0e4, "unit": "msgs/s", "hardware": "48-way Xeon", "source": "jepsen-io/maelstrom README.md:39-42",
                                      "note": "quoted for scale only; vs_baseline stays null because BASELINE.json publishes no number for this metric"}
        note({"stage": "headline", "line": out})

    # ---- the secondary legs (every rank: they hold collectives at N > 1); the headline above is already on record ----
    gather = gather_err = None
    if not args.no_gather:
        note({"stage": "leg", "name": "history_gather"})
        gather, gather_err = guarded("history_gather", 180.0, lambda: history_gather(eng, torch, dist, dev, world, rank, torch_view))
    incl = incl_err = None
    if not args.no_fetch and not stuck and args.config == "cfg2":   # (the PCIe-inclusive leg is defined for the headline)
        def incl_leg():
            first0 = (args.warmup + args.steps + 2) * world * n + EN.shard(world * n, rank, world)[0] * 4
            isteps = max(args.steps, 100)   # a pipeline's rate is its steady state: enough batches that filling and draining it do not show
            im, idt, ibytes = fetch_inclusive(E, cfg, torch, dev, local_rank, n, isteps, first0, torch_view)
            it = torch.tensor([float(im), idt], dtype=torch.float64, device=dev)
            if dist:
                tm = it[1:].clone()
                dist.all_reduce(it[:1])
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                it[1] = tm[0]
            return {"value": float(it[0]) / float(it[1]), "ms_per_step": float(it[1]) / isteps * 1e3, "steps": isteps, "bytes_fetched_per_batch": ibytes,
                    "how": "two engine contexts alternate: after batch k is simulated and checked its histories are compacted on the device and their PCIe copies queued (msim_fetch_begin); they cross while batch k+1 runs on the other context; msim_fetch waits for them"}
        note({"stage": "leg", "name": "incl_fetch"})
        incl, incl_err = guarded("incl_fetch", 240.0, incl_leg)

    if rank == 0:
        out["value_incl_fetch"] = incl["value"] if incl else None
        out["incl_fetch"] = incl
        if gather:
            out["history_gather"] = gather
        if gather_err:
            out["history_gather_error"] = gather_err
        if incl_err:
            out["incl_fetch_error"] = incl_err
        if args.cpu_sample > 0 and world == 1:   # the CPU leg is measured once, at N=1
            note({"stage": "leg", "name": "cpu_baseline"})
            out["cpu_baseline"] = cpu_baseline(cfg, args.cpu_sample)
            out["cpu_baseline"]["process_harness"] = process_harness(min(20.0, 2 * args.cpu_sample))
        note({"stage": "final"})
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if stuck:   # a leg is still inside a collective: no barrier, no teardown that could wait for it
        sys.stderr.write(f"[bench] rank {rank}: {', '.join(stuck)} did not finish; leaving without the final barrier\n")
        sys.stderr.flush()
        os._exit(0)
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    for e in engs:
        e.close()


def fetch_inclusive(E, cfg, torch, dev, local_rank, n, steps, first0, torch_view):
    """SURVEY.md §8(d)(i) counts msgs/s over kernel + gather + D2H: the same step with the histories' way to pinned host memory
    inside the timed region.  Two engine contexts alternate: right after batch k is simulated and checked, `msim_fetch_begin`
    compacts the used prefix of its slabs on the device and queues the two PCIe copies; they run while the other context simulates
    batch k+1, and `msim_fetch` waits for them before the context is used again.  Returns (msgs, seconds, bytes fetched per batch)."""
    engs = [E.Engine(cfg, device=local_rank) for _ in range(2)]
    msgs = torch.zeros(1, dtype=torch.int64, device=dev)

    sent = {}   # per context: the all_send column of its msim_net_stats slab, where it lies in HBM (stable once the slabs exist)

    def one(e, first):
        e.run(first, n)
        e.check()
        if id(e) in sent:
            msgs.add_(sent[id(e)].sum())

    try:
        for k, e in enumerate(engs):   # warm-up: code load, slabs, pinned mirrors
            one(e, first0 + k * n)
            e.fetch()
            db = e.device_buffers()
            sent[id(e)] = torch_view(db.stats, db.stats_bytes, torch.int64, dev).view(-1, 6)[:, 0]
            msgs.add_(sent[id(e)].sum())
        torch.cuda.synchronize()
        msgs.zero_()
        began = [False, False]
        t0 = time.perf_counter()
        for k in range(steps):
            e = engs[k % 2]
            if began[k % 2]:
                e.fetch()                      # batch k-2 of this context is on the host before its buffers are reused
            one(e, first0 + (2 + k) * n)
            e.fetch_begin()
            began[k % 2] = True
        for j, e in enumerate(engs):
            if began[j]:
                e.fetch()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rows, pay = engs[(steps - 1) % 2].raw_history(0)   # fetched views are valid: touch one
        assert len(rows) > 0
        m = engs[(steps - 1) % 2]
        db = m.device_buffers()
        meta = torch_view(db.meta, db.meta_bytes, torch.int32, dev).view(-1, 8)
        nbytes = int(meta[:, 0].sum(dtype=torch.int64)) * 16 + int(meta[:, 1].sum(dtype=torch.int64)) * 4 + n * (48 + 32)
        return int(msgs.item()), dt, nbytes
    finally:
        for e in engs:
            e.close()


def process_harness(seconds_budget=20.0):
    """the process-faithful stand-in for the reference's cost structure — one OS process per node, JSON
    lines over pipes, routed in memory as fast as it goes (tools/process_harness_rate.py).  The reference's own demo/js/gossip.js
    when node.js and the reference tree are present (build container), this repository's python node (tools/harness_node.py)
    otherwise (the GPU box)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import process_harness_rate as PH
        k = 400
        r = PH.measure(25, k)
        if r["seconds"] < seconds_budget / 8:   # a second, longer sample within the budget
            r = PH.measure(25, min(4000, int(k * seconds_budget / 4 / max(r["seconds"], 0.05))))
        return {"value": float(r["msgs_per_s"]), "unit": "msgs/s", "processes": 25, "harness": r["harness"], "host_cpus": r["host_cpus"],
                "sample": "%d broadcasts, %d messages in %.1f s" % (r["broadcasts"], r["messages"], r["seconds"])}
    except Exception as ex:   # no interpreter for the nodes, no pipes: say why
        return {"value": None, "reason": "%s: %s" % (type(ex).__name__, ex)}


def history_gather(eng, torch, dist, dev, world, rank, torch_view):
    """Variable-length history gather of the last batch to rank 0.  Transport "cabi" = msim_gather behind the C-ABI
    (csrc/gather.cpp: device-side compaction, ncclAllGather of the byte counts, one grouped ncclSend/ncclRecv per slab kind over
    RCCL/xGMI); the RCCL id reaches the ranks through torch.distributed.  If RCCL cannot be bound behind the library the same
    exchange runs on torch tensors (maelstrom_amd.ensemble.gather_to_root).  At world=1 this is the on-device compaction only."""
    from maelstrom_amd import ensemble as EN
    from maelstrom_amd import engine as E
    transport, note = "cabi", None
    try:
        if dist:
            idt = torch.zeros(128, dtype=torch.uint8, device=dev)
            if rank == 0:
                idt = torch.frombuffer(bytearray(E.Engine.comm_unique_id()), dtype=torch.uint8).to(dev)
            dist.broadcast(idt, 0)
            eng.comm_init(bytes(idt.cpu().numpy().tobytes()), rank, world)
    except E.EngineError as ex:
        transport, note = "torch", str(ex)
    agree = torch.tensor([1 if transport == "cabi" else 0], dtype=torch.int64, device=dev)
    if dist:
        dist.all_reduce(agree, op=dist.ReduceOp.MIN)   # every rank takes the same transport
    if int(agree.item()) == 0:
        transport = "torch"
    best, nbytes, recv = None, 0, 0
    for _ in range(2):  # first pass warms RCCL channels / allocations
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        t0 = time.perf_counter()
        if transport == "cabi":
            g = eng.gather(0)
            if rank == 0:
                nbytes, recv = int(g.rows_bytes + g.payload_bytes + g.meta_bytes + g.stats_bytes), int(g.bytes_received)
        else:
            db = eng.device_buffers()
            meta = torch_view(db.meta, db.meta_bytes, torch.int32, dev).view(-1, 8)
            rows = torch_view(db.rows, db.rows_bytes, torch.int32, dev).view(db.n_instances, db.max_rows, 4)
            pay = torch_view(db.payload, db.payload_bytes, torch.int32, dev).view(db.n_instances, db.max_payload_words)
            crow, cpay, nr, nw = EN.compact(rows, pay, meta)
            u8 = lambda t: t.contiguous().view(torch.uint8).reshape(-1)
            got, recv = EN.gather_to_root([u8(crow), u8(cpay), u8(meta), torch_view(db.stats, db.stats_bytes, torch.uint8, dev)], dist, world, rank, 0)
            if rank == 0:
                nbytes = sum(int(t.numel()) for t in got)
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    out = {"bytes": nbytes, "bytes_over_links": recv, "ms": best * 1e3, "GB_per_s": nbytes / best / 1e9, "ranks": world, "transport": transport,
           "pattern": "device compaction -> all-gather of byte counts -> one send per slab kind per peer to the root (bytes moved = sum of history bytes)"}
    if note:
        out["cabi_unavailable"] = note
    return out


if __name__ == "__main__":
    main()
