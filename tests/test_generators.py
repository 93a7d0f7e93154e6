"""The generator interpreter (core.clj:67-80 over [upstream] jepsen.generator: stagger, mix, each-thread, time-limit, phases —
restated, parity unpinned): what the oracle's histories show of it has the published shape."""
import collections

import numpy as np
import pytest

from maelstrom_amd import _abi as A
from maelstrom_amd import engine as E
import oracle_lib as O


def _ops(workload, inst=0, **kw):
    cfg = E.test_config(workload, **kw)
    r = O.run(cfg, inst, 1)
    assert r.meta["flags"][0] == 0
    return cfg, [o for o in E.decode_history(*r.history(0), cfg.n_nodes, cfg.workload) if o["process"] != ":nemesis"]


def test_stagger_mix_and_final_reads_of_broadcast():
    rate, tl, n = 50.0, 40.0, 5
    cfg, ops = _ops("broadcast", node_count=n, rate=rate, time_limit=tl, seed=5)
    inv = [o for o in ops if o["type"] == ":invoke" and not o.get("final?")]
    t = np.array([o["time"] for o in inv]) / 1e9
    assert abs(len(inv) - rate * tl) < 4 * np.sqrt(rate * tl) and t.max() < tl          # (gen/time-limit): nothing is invoked after it
    gaps = np.diff(t)
    assert abs(gaps.mean() - 1 / rate) < 0.1 / rate and gaps.max() < 2 / rate + 1e-6    # (gen/stagger (/ rate)): uniform on [0, 2/rate)
    assert abs(np.quantile(gaps, 0.5) - 1 / rate) < 0.15 / rate
    f = collections.Counter(o["f"] for o in inv)
    assert abs(f[":broadcast"] - f[":read"]) < 4 * np.sqrt(len(inv))                    # (gen/mix [broadcasts reads]): a fair coin
    vals = [o["value"] for o in inv if o["f"] == ":broadcast"]
    assert vals == list(range(len(vals)))                                               # (map (fn [x] {:f :broadcast :value x}) (range)), broadcast.clj:237-238
    procs = collections.Counter(o["process"] for o in inv)
    assert set(procs) == set(range(n)) and max(procs.values()) < 1.3 * min(procs.values())   # any free worker, uniformly
    fin = [o for o in ops if o.get("final?") and o["type"] == ":invoke"]
    assert sorted(o["process"] for o in fin) == list(range(n)) and all(o["f"] == ":read" for o in fin)   # (gen/each-thread {:f :read :final? true})
    last = max(o["time"] for o in ops if not o.get("final?"))
    assert all(o["time"] >= last + 10e9 for o in fin)                                  # after (gen/sleep 10), core.clj:78


def test_linearizable_register_generator_shape():
    """[upstream] jepsen.tests.linearizable-register as lin_kv.clj:84 uses it: per key 2n threads, the first n only read, the
    others mix writes and compare-and-sets, values 0..4, a key is retired once 20 processes have touched it."""
    n = 3
    cfg, ops = _ops("lin-kv", bin="lin-kv-proxy", proxy_service="lin-kv", node_count=n, rate=100.0, time_limit=30.0, seed=6)
    inv = [o for o in ops if o["type"] == ":invoke"]
    assert cfg.concurrency == 2 * n
    for o in inv:
        reader = (o["process"] % (2 * n)) < n
        assert (o["f"] == ":read") == reader
        k, v = o["value"]
        if o["f"] == ":write":
            assert 0 <= v <= 4
        if o["f"] == ":cas":
            assert all(0 <= x <= 4 for x in v)
    f = collections.Counter(o["f"] for o in inv)
    assert 1.4 < f[":cas"] / f[":write"] < 2.8                                          # (gen/mix [w cas cas])
    keys = [o["value"][0] for o in inv]
    assert keys == sorted(keys) and len(set(keys)) >= 1                                 # one key at a time, in order
    for key in set(keys):
        assert len({o["process"] for o in inv if o["value"][0] == key}) <= 20          # (gen/process-limit 20)


def test_echo_and_unique_ids_and_counter_generators():
    _, ops = _ops("echo", node_count=2, rate=100.0, time_limit=10.0, seed=7)
    inv = [o for o in ops if o["type"] == ":invoke"]
    ks = [int(o["value"].split()[-1]) for o in inv]
    assert all(o["value"].startswith("Please echo ") for o in inv) and 0 <= min(ks) and max(ks) < 128 and len(set(ks)) > 100   # echo.clj:72-75
    _, ops = _ops("unique-ids", node_count=3, rate=100.0, time_limit=5.0, seed=7)
    assert {o["f"] for o in ops} == {":generate"}                                       # (gen/repeat {:f :generate}), unique_ids.clj:66
    _, ops = _ops("pn-counter", node_count=3, rate=100.0, time_limit=20.0, seed=7)
    adds = [o["value"] for o in ops if o["type"] == ":invoke" and o["f"] == ":add"]
    assert set(adds) == set(range(-5, 5))                                               # (- (rand-int 10) 5), pn_counter.clj:134-135
    assert abs(np.mean(adds) + 0.5) < 0.5


def test_txn_generator_shape():
    """[upstream] elle's wr-txns as both transactional workloads use it: 1..max-txn-length micro-ops, key i of the active pool
    twice as likely as key i-1, writes unique and ascending per key, a key retired after max-writes-per-key writes."""
    for wl in ("txn-list-append", "txn-rw-register"):
        cfg, ops = _ops(wl, node_count=3, rate=200.0, time_limit=20.0, seed=8, key_count=4, max_txn_length=4, max_writes_per_key=8)
        inv = [o["value"] for o in ops if o["type"] == ":invoke"]
        lens = collections.Counter(len(t) for t in inv)
        assert set(lens) == {1, 2, 3, 4} and max(lens.values()) < 1.25 * min(lens.values())
        writes = collections.defaultdict(list)
        for t in inv:
            for f, k, v in t:
                if f != ":r":
                    writes[k].append(v)
                else:
                    assert v is None                                                    # reads are submitted empty
        assert all(v == list(range(1, len(v) + 1)) and len(v) <= 8 for v in writes.values())
        assert len(writes) > 20                                                         # keys do get retired and replaced
        rw = collections.Counter(f for t in inv for f, _k, _v in t)
        assert 0.8 < rw[":r"] / (sum(rw.values()) - rw[":r"]) < 1.25
        # exponential key choice: among the first txns (pool = keys 0..3) key 3 is ~8x as likely as key 0
        first = collections.Counter(k for t in inv[:60] for _f, k, _v in t if k < 4)
        assert first[3] > first[0]
