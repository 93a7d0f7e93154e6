"""Device checker (set-full / echo) against the pure-Python restatement of jepsen's algorithm."""
import numpy as np
import pytest

from maelstrom_amd import engine as E
import setfull_ref as R

pytestmark = pytest.mark.gpu

VALID = {1: True, 0: False, 2: "unknown"}


def _check(cfg, n):
    with E.Engine(cfg) as eng:
        eng.run(0, n)
        eng.check()
        eng.fetch()
        res = eng.check_results()
        for i in range(n):
            h = eng.history(i)
            ref = R.set_full(h)
            g = res[i]
            assert VALID[int(g["valid"])] == ref["valid?"], (i, g, ref)
            for k, rk in (("attempt_count", "attempt-count"), ("stable_count", "stable-count"), ("lost_count", "lost-count"),
                          ("never_read_count", "never-read-count"), ("stale_count", "stale-count"), ("duplicated_count", "duplicated-count")):
                assert int(g[k]) == ref[rk], (i, k, int(g[k]), ref[rk])
            if ref["stable-latencies"]:
                assert [int(x) for x in g["stable_latency_ms"]] == [ref["stable-latencies"][q] for q in (0, 0.5, 0.95, 0.99, 1)], (i, g, ref)
            inv = sum(1 for op in h if op["type"] == ":invoke" and op["process"] != ":nemesis")
            assert int(g["op_count"]) == inv
        return res


def test_set_full_healthy_broadcast(lib):
    res = _check(E.test_config("broadcast", node_count=5, rate=20, time_limit=5, latency=50, seed=4), 8)
    assert (res["valid"] == 1).all() and (res["lost_count"] == 0).all()


def test_set_full_stale_latencies_n25(lib):
    res = _check(E.test_config("broadcast", node_count=25, rate=50, time_limit=5, latency=100, latency_dist="exponential", seed=8), 4)
    assert (res["stale_count"] > 0).all()  # 100 ms hops are visible as stale reads (02-performance.md:205-211)


def test_set_full_partitioned_fire_and_forget_loses_or_stales(lib):
    # fire-and-forget gossip under partitions: messages dropped for good => elements missing on some nodes
    res = _check(E.test_config("broadcast", bin="broadcast-ff", node_count=5, rate=20, time_limit=20, nemesis=["partition"],
                               nemesis_interval=3, latency=10, seed=6), 16)
    assert ((res["lost_count"] > 0) | (res["stale_count"] > 0)).any()


def test_set_full_partitioned_retry_is_valid(lib):
    res = _check(E.test_config("broadcast", bin="broadcast-ack-retry", node_count=5, rate=10, time_limit=20, topology="tree4",
                               nemesis=["partition"], nemesis_interval=5, latency=10, seed=6), 16)
    assert (res["valid"] == 1).all() and (res["lost_count"] == 0).all()  # 02-performance.md:519-541


def test_g_set_checker(lib):
    _check(E.test_config("g-set", node_count=5, rate=10, time_limit=10, seed=2), 4)


def test_wide_g_set_checker(lib):
    """100 worker threads: the checker keeps two pending invocations per lane."""
    _check(E.test_config("g-set", node_count=100, rate=100, time_limit=10, latency=100, latency_dist="exponential", p_loss=0.05, seed=4), 3)


def test_wide_broadcast_checker(lib):
    """set-full over 100 nodes' reads (broadcast.clj:216-228)."""
    res = _check(E.test_config("broadcast", node_count=100, rate=100, time_limit=10, latency=20, seed=5), 3)
    assert (res["valid"] == 1).all() and (res["lost_count"] == 0).all()


def test_wide_pn_counter_checker(lib):
    """100 nodes' final reads against the acceptable sums (pn_counter.clj:84-123): device verdict, and the host checker on one history."""
    cfg = E.test_config("pn-counter", node_count=100, rate=100, time_limit=10, latency=20, seed=6)
    with E.Engine(cfg) as eng:
        eng.run(0, 4)
        eng.check()
        eng.fetch()
        res = eng.check_results()
        assert (res["valid"] == 1).all() and (res["error_count"] == 0).all() and (res["attempt_count"] == 100).all()
        rows, _ = eng.raw_history(2)
        one = E.check_pn_history(rows)
        assert one["valid?"] is True and len(one["final-reads"]) == 100


def test_echo_checker(lib):
    cfg = E.test_config("echo", node_count=3, rate=10, time_limit=5, seed=2)
    with E.Engine(cfg) as eng:
        eng.run(0, 8)
        eng.check()
        eng.fetch()
        res = eng.check_results()
        for i in range(8):
            assert R.echo_check(eng.history(i))["valid?"] and int(res[i]["valid"]) == 1 and int(res[i]["error_count"]) == 0


def test_echo_checker_counts_timeouts_and_unfinished_invocations(lib):
    """workload/echo.clj:44-63 pairs every :invoke with its completion and requires (:echo (:value complete)) to equal the
    request: a timed-out echo (:info, value = the request string) is an error, and so is an invocation with no completion."""
    cfg = E.test_config("echo", node_count=3, rate=20, time_limit=8, p_loss=0.2, seed=5)
    with E.Engine(cfg) as eng:
        eng.run(0, 8)
        eng.check()
        eng.fetch()
        res = eng.check_results()
        lossy = 0
        for i in range(8):
            h = eng.history(i)
            ref = R.echo_check(h)
            infos = sum(1 for op in h if op["type"] == ":info" and op["process"] != ":nemesis")
            assert len(ref["errors"]) >= infos
            assert int(res[i]["error_count"]) == len(ref["errors"]) and (int(res[i]["valid"]) == 1) == ref["valid?"]
            lossy += infos > 0
        assert lossy >= 4   # 20 % loss over ~160 echoes: timeouts in (nearly) every instance


def test_lin_kv_checker_on_raft_histories(lib):
    """msim_check for lin-kv = per-key linearizability (host side, csrc/lin_check.cpp) over the fetched histories."""
    import linearizable_ref as L
    from maelstrom_amd import _abi as A
    cfg = E.test_config("lin-kv", bin="raft", node_count=5, rate=30, time_limit=30, nemesis=["partition"], nemesis_interval=6,
                        latency=10, seed=29)
    with E.Engine(cfg) as eng:
        eng.run(0, 16)
        eng.check()
        eng.fetch()
        res = eng.check_results()
        assert (res["valid"] == 1).all()
        for i in range(4):
            h = eng.history(i)
            ref = L.check(h)
            assert all(ref.values()) and int(res[i]["attempt_count"]) == len(ref)
            assert int(res[i]["op_count"]) == sum(1 for op in h if op["type"] == ":invoke" and op["process"] != ":nemesis")


def test_txn_list_append_checker_on_engine_histories(lib):
    """msim_check for txn-list-append = the host list-append checker over the fetched histories: every history the
    single-root node produces is strict-serializable; conflicts show up as :fail, never as anomalies."""
    cfg = E.test_config("txn-list-append", node_count=5, rate=100, time_limit=15, latency=5, nemesis=["partition"], nemesis_interval=5, seed=9)
    with E.Engine(cfg) as eng:
        eng.run(0, 16)
        eng.check()
        eng.fetch()
        res = eng.check_results()
        assert (res["valid"] == 1).all() and (res["error_count"] == 0).all()
        assert (res["fail_count"] > 0).all() and (res["ok_count"] > res["fail_count"]).all()
        rows, pay = eng.raw_history(3)
        one = E.check_txn_history(rows, pay)
        assert one["valid?"] is True and one["txn-count"] == int(res["attempt_count"][3])


def test_txn_rw_register_checker_on_engine_histories(lib):
    """msim_check for txn-rw-register judges by the configured consistency model (core.clj:118,160-165): the HAT node's
    histories are read-committed, and not serializable."""
    kw = dict(node_count=2, rate=100, time_limit=15, nemesis=["partition"], nemesis_interval=5, seed=10)
    cfg = E.test_config("txn-rw-register", **kw)
    with E.Engine(cfg) as eng:
        eng.run(0, 16)
        eng.check()
        eng.fetch()
        res = eng.check_results()
        assert (res["valid"] == 1).all() and (res["info_count"] == 0).all() and (res["ok_count"] > 1000).all()
        assert ((res["error_count"] & ~np.uint32(16 | 32 | 512)) == 0).all()   # nothing but G-single / G2 (/ -realtime)
        rows, pay = eng.raw_history(3)
        one = E.check_rw_history(rows, pay, "read-committed")
        assert one["valid?"] is True and one["txn-count"] == int(res["attempt_count"][3])
    with E.Engine(E.test_config("txn-rw-register", consistency_model="serializable", **kw)) as eng:
        eng.run(0, 16)
        eng.check()
        assert (eng.check_results()["valid"] == 0).sum() >= 12


def test_pn_counter_checker_on_engine_histories(lib):
    cfg = E.test_config("pn-counter", node_count=5, rate=50, time_limit=10, latency=20, latency_dist="exponential", p_loss=0.05, seed=12)
    with E.Engine(cfg) as eng:
        eng.run(0, 16)
        eng.check()
        eng.fetch()
        res = eng.check_results()
        assert (res["valid"] == 1).all() and (res["error_count"] == 0).all() and (res["attempt_count"] >= 1).all()
        rows, _ = eng.raw_history(5)
        one = E.check_pn_history(rows)
        assert one["valid?"] is True and len(one["final-reads"]) == int(res["attempt_count"][5])


def test_services_under_the_linearizability_checker(lib):
    """lin_kv_proxy.rb:33-35: lin-kv passes, seq-kv / lww-kv show linearization failures (engine histories, msim_check)."""
    out = {}
    for service in ("lin-kv", "seq-kv", "lww-kv"):
        cfg = E.test_config("lin-kv", bin="lin-kv-proxy", proxy_service=service, node_count=5, rate=100, time_limit=30, latency=5, seed=14)
        with E.Engine(cfg) as eng:
            eng.run(0, 32)
            eng.check()
            out[service] = int((eng.check_results()["valid"] == 1).sum())
    assert out["lin-kv"] == 32 and out["seq-kv"] < 16 and out["lww-kv"] < 16, out


def test_availability_checker_device_equals_host(lib):
    """checker.clj:6-39 for a whole run on the device (msim_check_availability) = the host entry point history by history"""
    for cfg in (E.test_config("echo", node_count=3, rate=20, time_limit=8, p_loss=0.2, seed=5),
                E.test_config("lin-kv", bin="raft", node_count=5, rate=30, time_limit=10, latency=10, nemesis=["partition"], nemesis_interval=3, seed=6)):
        with E.Engine(cfg) as eng:
            eng.run(0, 12)
            eng.fetch()
            for a in (None, "total", 0.9):
                dev = eng.check_availability(a)
                for i in range(12):
                    assert dev[i] == E.check_availability_rows(eng.raw_history(i)[0], a)
            assert not all(r["valid?"] for r in eng.check_availability("total"))   # loss / partitions cost some operations
            assert all(r["valid?"] for r in eng.check_availability(None))


def test_unique_ids_device_checker_equals_host(lib):
    """[upstream] jepsen.checker/unique-ids on the device (csrc/unique_check_dev.hip) against the host checker (csrc/pn_check.cpp):
    engine histories (flake ids under partitions: never duplicated), then the same histories with ids copied over other ids."""
    import ctypes as C
    from maelstrom_amd import _abi as A

    def host(rows):
        res = A.CheckResult()
        rows = np.ascontiguousarray(rows)
        assert A.load().msim_check_unique_rows(rows.ctypes.data_as(C.c_void_p), len(rows), C.byref(res)) == 0
        return res

    fields = ("valid", "attempt_count", "duplicated_count", "op_count", "ok_count", "fail_count", "info_count")
    cfg = E.test_config("unique-ids", node_count=3, rate=500, time_limit=6, latency=5, nemesis=["partition"], nemesis_interval=2, seed=12)
    n = 12
    for flags in (0, 0x2000):   # the table of a history in LDS (a workgroup per history; the default where it fits) / in HBM workspace
        with E.Engine(cfg) as eng:
            if flags:
                eng.set_dev_flags(flags)
            eng.run(0, n)
            eng.check()
            res = eng.check_results().copy()
            eng.fetch()
            hs = [eng.raw_history(i)[0].copy() for i in range(n)]
        for i in range(n):
            h = host(hs[i])
            for f in fields:
                assert int(res[i][f]) == int(getattr(h, f)), (i, f)
            assert [int(x) for x in res[i]["stable_latency_ms"][:2]] == [int(h.stable_latency_ms[0]), int(h.stable_latency_ms[1])]
        assert (res["valid"] == 1).all() and (res["ok_count"] > 100).all()
    rng = np.random.default_rng(2)
    bad = []
    for k, rows in enumerate(hs):
        rows = rows.copy()
        oks = np.flatnonzero(((rows["packed"] & 3) == A.T_OK) & (((rows["packed"] >> 2) & 31) == A.F_GENERATE))
        for _ in range(k):   # k duplicated acknowledgements (some of the same value)
            a, b = rng.choice(oks, size=2, replace=False)
            rows["value"][a] = rows["value"][b]
        bad.append(rows)
    bad.append(np.zeros(0, dtype=E.OP_DT))
    dev = E.check_unique_batch(bad)
    for i, rows in enumerate(bad):
        h = host(rows)
        for f in fields:
            assert int(dev[i][f]) == int(getattr(h, f)), (i, f, int(dev[i][f]), int(getattr(h, f)))
    assert int(dev[0]["valid"]) == 1 and (dev["valid"][1:n] == 0).all()


def test_unique_ids_batch_of_completions_only_history_terminates(lib):
    """A caller's history may be ALL acknowledgements (filtered to its completions): 36000 distinct :ok ids in 36000 rows are more than the
    LDS table's 32768 slots — the batch entry sizes by the rows, not by rows / 2 (the LDS kernel's probe loop spun for ever
    on a full table), and the verdict is the host checker's."""
    import ctypes as C
    from maelstrom_amd import _abi as A
    for n_ids, dup in ((36000, False), (36000, True), (19000, False), (30000, True)):
        rows = np.zeros(n_ids, dtype=E.OP_DT)
        rows["time_len"] = np.arange(n_ids, dtype=np.uint64) * 1000
        rows["packed"] = A.T_OK | (A.F_GENERATE << 2) | (1 << 12)
        rows["value"] = (np.arange(n_ids, dtype=np.uint64) * 2654435761 % (1 << 31)).astype(np.uint32)
        if dup:
            rows["value"][n_ids - 1] = rows["value"][7]
        dev = E.check_unique_batch([rows])[0]
        res = A.CheckResult()
        assert A.load().msim_check_unique_rows(rows.ctypes.data_as(C.c_void_p), len(rows), C.byref(res)) == 0
        assert int(dev["valid"]) == int(res.valid) == (0 if dup else 1) and int(dev["duplicated_count"]) == int(res.duplicated_count) == (1 if dup else 0)
        assert int(dev["ok_count"]) == n_ids and int(dev["error_count"]) == 0


def test_pn_counter_device_checker_equals_host(lib):
    """The counter checker on the device (csrc/pn_check_dev.hip: a 4096-value bitmap of acceptable sums) against the host checker
    (csrc/pn_check.cpp) — on the reference's own vectors (pn_counter_test.clj:10-36, KAT-9), on engine histories with lost acks
    (indeterminate adds), on corrupted final reads and on a window wider than the bitmap (host fallback)."""
    import ctypes as C
    from maelstrom_amd import _abi as A

    def host(rows):
        res = A.CheckResult()
        rows = np.ascontiguousarray(rows)
        assert A.load().msim_check_pn_rows(rows.ctypes.data_as(C.c_void_p), len(rows), C.byref(res), None, 0, None) == 0
        return res

    fields = ("valid", "attempt_count", "error_count", "stable_count", "op_count", "ok_count", "fail_count", "info_count")
    kat = [
        [],
        [{"type": ":ok", "f": ":add", "value": 2}, {"type": ":ok", "f": ":add", "value": 3}, {"type": ":ok", "f": ":read", "final?": True, "value": 5},
         {"type": ":ok", "f": ":read", "final?": True, "value": 4}],
        [{"type": ":ok", "f": ":add", "value": 10}, {"type": ":info", "f": ":add", "value": 5}, {"type": ":info", "f": ":add", "value": -1},
         {"type": ":info", "f": ":add", "value": -1}, {"type": ":ok", "f": ":read", "final?": True, "value": 11}, {"type": ":ok", "f": ":read", "final?": True, "value": 15}],
        [{"type": ":ok", "f": ":add", "value": 4}, {"type": ":fail", "f": ":add", "value": 100}, {"type": ":ok", "f": ":read", "value": 0},
         {"type": ":ok", "f": ":read", "final?": True, "value": 4}],
        [{"type": ":info", "f": ":add", "value": 3000}, {"type": ":info", "f": ":add", "value": -2000}, {"type": ":info", "f": ":add", "value": 70},
         {"type": ":ok", "f": ":read", "final?": True, "value": 1070}, {"type": ":ok", "f": ":read", "final?": True, "value": 1069}],   # window 5071 > 4096: host
        [{"type": ":info", "f": ":add", "value": 64}, {"type": ":info", "f": ":add", "value": 128}, {"type": ":info", "f": ":add", "value": -65},
         {"type": ":ok", "f": ":add", "value": -7}] + [{"type": ":ok", "f": ":read", "final?": True, "value": v} for v in (-72, -8, -7, 57, 121, 120, 185, 186, -73)],
    ]
    hs = [E.encode_pn_history(ops) for ops in kat]
    cfg = E.test_config("pn-counter", node_count=5, rate=100, time_limit=10, latency=50, latency_dist="exponential", p_loss=0.15, seed=17)
    n = 16
    with E.Engine(cfg) as eng:
        eng.run(0, n)
        eng.check()
        res = eng.check_results()
        eng.fetch()
        eh = [eng.raw_history(i)[0].copy() for i in range(n)]
    for i in range(n):
        h = host(eh[i])
        for f in fields:
            assert int(res[i][f]) == int(getattr(h, f)), (i, f, int(res[i][f]), int(getattr(h, f)))
    assert (res["info_count"] > 0).any() and (res["valid"] == 1).all()
    rng = np.random.default_rng(4)
    for rows in eh[:8]:   # corrupt a final read
        rows = rows.copy()
        fin = np.flatnonzero(((rows["packed"] >> 11) & 1 == 1) & ((rows["packed"] & 3) == A.T_OK))
        j = rng.choice(fin)
        rows["value"][j] = np.uint32((int(np.int32(rows["value"][j])) + int(rng.choice([-40, -3, 7, 1000]))) & 0xFFFFFFFF)
        hs.append(rows)
    dev = E.check_pn_batch(hs)
    for i, rows in enumerate(hs):
        h = host(rows)
        for f in fields:
            assert int(dev[i][f]) == int(getattr(h, f)), (i, f, int(dev[i][f]), int(getattr(h, f)))
    assert [int(v) for v in dev["valid"][:5]] == [1, 0, 0, 1, 0] and int(dev[2]["stable_count"]) == 2


@pytest.mark.parametrize("wl,kw", [
    ("lin-kv", dict(bin="raft", node_count=5, rate=30, time_limit=20, latency=10, nemesis=["partition"], nemesis_interval=5)),
    ("txn-list-append", dict(node_count=5, rate=100, time_limit=10, latency=5, nemesis=["partition"], nemesis_interval=3)),
    ("unique-ids", dict(node_count=3, rate=300, time_limit=5, latency=5)),
    ("pn-counter", dict(node_count=5, rate=50, time_limit=8, latency=30, p_loss=0.1)),
])
def test_device_and_host_checkers_give_the_same_results_through_msim_check(lib, wl, kw):
    """msim_check with the checkers where the histories are (default) and on the host cores (msim_set_dev_flags 0x800): every field
    of every result record equal — and the one-cluster-per-wavefront kernels (0x200) emit the same histories as the packed ones."""
    cfg = E.test_config(wl, seed=31, **kw)
    n = 24
    out = []
    for flags in (0, 0x800, 0x200):
        with E.Engine(cfg) as eng:
            eng.set_dev_flags(flags)
            eng.run(0, n)
            eng.check()
            res = eng.check_results()
            eng.fetch()
            out.append((res, [eng.raw_history(i)[0].tobytes() for i in range(n)]))
    assert out[0][0].tobytes() == out[1][0].tobytes() == out[2][0].tobytes()
    assert out[0][1] == out[2][1]


@pytest.mark.parametrize("kw", [
    dict(workload="g-set", node_count=5, concurrency=15, rate=1000, time_limit=2, latency=20, latency_dist="exponential", seed=31),
    dict(workload="broadcast", bin="broadcast-ack-retry", node_count=5, concurrency=20, rate=800, time_limit=3, latency=10, p_loss=0.1, seed=32),
    dict(workload="broadcast", bin="broadcast-ff", node_count=7, concurrency=21, rate=700, time_limit=3, latency=30, latency_dist="uniform",
         nemesis=["partition"], nemesis_interval=1, seed=33),
    dict(workload="g-set", node_count=3, concurrency=30, rate=1500, time_limit=1, seed=34),
])
def test_set_full_when_reads_overtake_each_other(lib, kw):
    """Many workers per node at a high rate: reads queue behind one another at the nodes (one input per node and round), so reads
    invoked later complete earlier — the case the single sweep of check_kernel corrects `known` for — and elements come and go between
    reads at different nodes (many last-present / last-absent transitions)."""
    res = _check(E.test_config(**kw), 6)
    assert (res["attempt_count"] > 20).all()
