"""Parity at the shapes the benchmarks time: the exact `bench.headline_config` (n=25, inbox_capacity=6,
rate 100, 20 s) and the cfg3 / cfg4 / cfg5 shapes of tools/bench_configs.py, bit-compared with the CPU oracle through the
C-ABI — rows, payload, net stats, flags and round counts of every instance."""
import hashlib

import numpy as np
import pytest

from maelstrom_amd import _abi as A
from maelstrom_amd import engine as E
import oracle_lib as O

pytestmark = pytest.mark.gpu


def _digest_engine(eng, i):
    rows, pay = eng.raw_history(i)
    st = eng.net_stats_raw(i)
    m = eng.meta(i)
    h = hashlib.sha256(rows.tobytes())
    h.update(pay.tobytes())
    h.update(np.array([getattr(st, f) for f, _ in A.NetStats._fields_], dtype=np.uint64).tobytes())
    h.update(np.array([m.n_rows, m.n_payload_words, m.flags, m.n_rounds], dtype=np.uint32).tobytes())
    return h.hexdigest()


def _digest_oracle(ora, i):
    rows, pay = ora.history(i)
    m = ora.meta[i]
    h = hashlib.sha256(rows.tobytes())
    h.update(pay.tobytes())
    h.update(np.array([int(x) for x in ora.stats[i]], dtype=np.uint64).tobytes())
    h.update(np.array([m["n_rows"], m["n_payload_words"], m["flags"], m["n_rounds"]], dtype=np.uint32).tobytes())
    return h.hexdigest()


def _compare_digests(cfg, first, n, batch=None):
    """Engine runs `batch` instances (default n) starting at `first`; the first n are compared with the oracle."""
    ora = O.run(cfg, first, n)
    with E.Engine(cfg) as eng:
        eng.run(first, batch or n)
        eng.fetch()
        bad = [first + i for i in range(n) if _digest_engine(eng, i) != _digest_oracle(ora, i)]
        assert not bad, f"{len(bad)} of {n} instances differ from the oracle, first: {bad[:8]}"
        assert all(eng.meta(i).flags == 0 for i in range(n))
    return ora


def test_headline_config_verbatim(lib):
    """bench.py's own config object, 256 instances, and the same instances as part of a full 4096-instance batch (odd and
    even wave halves, the tail of the grid)."""
    import bench
    cfg = bench.headline_config(E, 2026)
    assert cfg.inbox_capacity == 6 and cfg.n_nodes == 25
    ora = _compare_digests(cfg, 0, 256)
    assert (ora.stats["all_send"] > 50000).all()
    _compare_digests(cfg, 4096 * 7, 64, batch=4096)
    _compare_digests(cfg, 5, 33, batch=33)   # odd batch: the last wavefront holds one cluster


@pytest.mark.parametrize("latency,dist", [(10, "constant"), (100, "constant"), (100, "exponential")])
def test_headline_latency_sweep(lib, latency, dist):
    """the latency sweep of the headline shape (doc/03-broadcast/02-performance.md:140-205)."""
    cfg = E.test_config("broadcast", bin="broadcast-ff", node_count=25, rate=100, time_limit=20, latency=latency, latency_dist=dist, seed=99)
    _compare_digests(cfg, 0, 48)


@pytest.mark.parametrize("p_loss", [0.05, 0.5])
def test_cfg3_gset_n100_shape(lib, p_loss):
    cfg = E.test_config("g-set", node_count=100, rate=100, time_limit=20, latency=100, latency_dist="exponential", p_loss=p_loss, seed=99)
    _compare_digests(cfg, 0, 24)


@pytest.mark.parametrize("kw", [dict(), dict(latency=10, nemesis=["partition"], nemesis_interval=10)])
def test_cfg4_raft_shape(lib, kw):
    cfg = E.test_config("lin-kv", bin="raft", node_count=5, rate=30, time_limit=60, seed=99, **kw)
    assert cfg.concurrency == 10
    _compare_digests(cfg, 0, 64)


def test_cfg5_txn_list_append_shape(lib):
    cfg = E.test_config("txn-list-append", node_count=5, rate=100, time_limit=30, latency=5, nemesis=["partition"], nemesis_interval=10, seed=99)
    _compare_digests(cfg, 0, 64)


def test_cfg5_multi_key_txn_shape(lib):
    """cfg5 over the multi-key node (multi_key_txn.js: same architecture as core.clj:113-114's datomic_list_append.rb, a different program)."""
    cfg = E.test_config("txn-list-append", bin="multi-key-txn", node_count=5, rate=100, time_limit=30, latency=5, nemesis=["partition"], nemesis_interval=10, seed=99)
    _compare_digests(cfg, 0, 64)


def test_cfg5_datomic_shape(lib):
    """cfg5 over the node core.clj:113-114 runs (demo/ruby/datomic_list_append.rb)."""
    cfg = E.test_config("txn-list-append", bin="datomic", node_count=5, rate=100, time_limit=30, latency=5, nemesis=["partition"], nemesis_interval=10, seed=99)
    _compare_digests(cfg, 0, 64)
    _compare_digests(cfg, 0, 64, batch=12288)   # the batch size from which eight clusters per wavefront are taken (csrc/dt8.hip)


@pytest.mark.parametrize("service", ["lin-kv", "lww-kv"])
def test_the_reference_demo_invocation_of_the_lin_kv_proxy(lib, service):
    """core.clj:112: `{:workload :lin-kv :bin "demo/ruby/lin_kv_proxy.rb" :concurrency 10}` at tools/bench_configs.py's shape — one cluster per
    wavefront (a small launch) and, as part of a launch of 4096 (the size from which four clusters per wavefront are taken: csrc/svc4.hip), the
    same instances again; every history of the full launch passes (lin-kv) or is judged by (lww-kv) the device's linearizability search."""
    cfg = E.test_config("lin-kv", bin="lin-kv-proxy", node_count=5, concurrency=10, rate=30, time_limit=60, latency=5, proxy_service=service, seed=99)
    _compare_digests(cfg, 0, 48)
    _compare_digests(cfg, 0, 48, batch=4096)
    _compare_digests(cfg, 4096 * 3 + 1, 31, batch=4097)   # a last wavefront with one cluster
    if service == "lin-kv":
        with E.Engine(cfg) as eng:
            eng.run(0, 4096)
            eng.check()
            assert (eng.check_results()["valid"] == 1).all()


def test_unique_ids_over_lin_tso_at_the_bench_shape(lib):
    cfg = E.test_config("unique-ids", bin="tso-ids", node_count=3, rate=1000, time_limit=10, latency=5, nemesis=["partition"], nemesis_interval=3, seed=99)
    _compare_digests(cfg, 0, 32)
    _compare_digests(cfg, 7, 32, batch=4096)   # four clusters per wavefront (svc4_kernel<.., TSO>)


def test_the_reference_demo_invocation_for_txn_list_append(lib):
    """core.clj:113-114: `{:workload :txn-list-append :bin "demo/ruby/datomic_list_append.rb"}` with the defaults of core.clj:136-229 — five nodes,
    one worker per node, rate 5, 60 s, latency 0."""
    cfg = E.test_config("txn-list-append", bin="datomic", seed=99)
    assert (cfg.n_nodes, cfg.concurrency, cfg.rate_mhz, cfg.time_limit_ms, cfg.latency_mean_ms) == (5, 5, 5000, 60000, 0)
    ora = _compare_digests(cfg, 0, 32)
    assert (ora.meta["n_rows"] > 400).all()


@pytest.mark.parametrize("kw", [
    dict(node_count=25, topology="line", latency=10),
    dict(node_count=25, topology="tree4", latency=0),
    dict(node_count=25, topology="total", latency=20, rate=20),
    dict(node_count=32, topology="grid", latency=5),
    dict(node_count=31, topology="tree2", latency=100, rate=200),
    dict(node_count=1, latency=0),
    dict(node_count=2, latency=3, rate=500),
    dict(node_count=9, bin="broadcast-ff-echoback", latency=0),
    dict(node_count=16, bin="broadcast-ff-echoback", latency=50, topology="tree3"),
    dict(node_count=25, latency=1000, rate=50),
    dict(node_count=5, rate=0.0, time_limit=3),
    dict(node_count=12, latency=30, rate=300, inbox_capacity=2, spill_capacity=64),
])
def test_two_clusters_per_wavefront_layout(lib, kw):
    """The layouts of duo.hip (two clusters per wavefront): topologies with more than four neighbours, 32-node clusters, deep
    queues that spill to HBM, both node programs, long latencies, an idle generator."""
    base = dict(workload="broadcast", rate=100, time_limit=10, seed=123)
    base.update(kw)
    cfg = E.test_config(**base)
    _compare_digests(cfg, 1000, 21)
