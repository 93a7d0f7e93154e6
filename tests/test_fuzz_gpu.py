"""Seeded differential sweep: random test options (workload, node program, cluster size, topology, rate, latency model,
loss, nemesis, concurrency) through the HIP engine and the CPU oracle, bit-compared.  The short paths of the headline
kernel (lone operation, cascade_lean<>, quiet_run<>) only fire under particular conditions; this sweep crosses their
boundaries (latency 0 / 1 ms, high rates that overlap operations with cascades, tiny queues that spill, journal on/off)."""
import os
import random

import pytest

from maelstrom_amd import engine as E
from test_parity_gpu import _compare

pytestmark = pytest.mark.gpu
N_INST = int(os.environ.get("MSIM_FUZZ_INSTANCES", "3"))   # (9 or 17 put several clusters into the wavefronts of the multi-cluster layouts)


def _random_case(rng):
    wl = rng.choice(["broadcast"] * 6 + ["g-set", "pn-counter", "g-counter", "unique-ids", "echo"])
    kw = dict(node_count=rng.choice([1, 2, 3, 5, 7, 9, 16, 25, 32]), rate=rng.choice([5, 20, 100, 400, 2000]),
              time_limit=rng.choice([2, 3, 5]), seed=rng.randrange(1 << 40))
    lat = rng.choice([0, 0, 0, 1, 2, 10, 50])
    dist = rng.choice(["constant", "constant", "uniform", "exponential"]) if lat else "constant"
    kw.update(latency=lat, latency_dist=dist)
    if rng.random() < 0.2:
        kw["p_loss"] = rng.choice([0.02, 0.2])
    if rng.random() < 0.25 and kw["node_count"] >= 3:
        kw.update(nemesis=["partition"], nemesis_interval=rng.choice([1, 2]))
    if wl == "broadcast":
        kw["bin"] = rng.choice(["broadcast-ff"] * 4 + ["broadcast-ff-echoback", "broadcast-ack-retry", "broadcast-rpc-all"])
        kw["topology"] = rng.choice(["grid", "line", "total", "tree2", "tree3", "tree4"])
        if kw["topology"] == "total" and kw["node_count"] > 16:
            kw["rate"] = min(kw["rate"], 100)
    if wl in ("g-set", "pn-counter", "g-counter"):
        kw["time_limit"] = 6
    if rng.random() < 0.25 and kw["node_count"] <= 16:
        kw["concurrency"] = rng.choice([1, 2, 3]) * kw["node_count"] + rng.choice([0, 0, 1])
    if rng.random() < 0.2:
        kw["journal_capacity"] = 2000000
    if rng.random() < 0.2:
        kw["inbox_capacity"] = rng.choice([1, 2, 3])
    return wl, kw


@pytest.mark.parametrize("case", range(int(os.environ.get("MSIM_FUZZ_CASES", "48"))))   # MSIM_FUZZ_CASES=N widens the sweep
def test_random_options_engine_equals_oracle(lib, case):
    rng = random.Random(0xC0FFEE + case)
    wl, kw = _random_case(rng)
    try:
        cfg = E.test_config(wl, **kw)
    except E.EngineError as e:   # an option combination the reference (or this build) rejects: not a parity case
        pytest.skip(str(e))
    first = rng.randrange(1 << 20)
    try:
        E.Engine(cfg).close()
    except E.EngineError as e:   # e.g. more endpoints than one wavefront has lanes
        pytest.skip(str(e))
    _compare(cfg, first, N_INST)
    if cfg.n_nodes <= 8 and cfg.concurrency == cfg.n_nodes and not cfg.journal_capacity:
        # eight clusters per wavefront (csrc/uid8.hip, csrc/crdt8.hip, csrc/bcast8.hip — bit 15: also where the headline layout applies)
        _compare(cfg, first, N_INST, dev_flags=0x8400 if wl == "broadcast" else 0x400)


def _random_kv_case(rng):
    kind = rng.choice(["raft", "raft", "proxy", "proxy", "txn", "txn"])
    if os.environ.get("MSIM_FUZZ_KIND"):   # focus a sweep on one program
        kind = os.environ["MSIM_FUZZ_KIND"]
    n = rng.choice([1, 3, 5, 7])
    kw = dict(node_count=n, rate=rng.choice([10, 30, 100, 300]), time_limit=rng.choice([4, 8, 12]), seed=rng.randrange(1 << 40))
    lat = rng.choice([0, 1, 5, 20])
    kw.update(latency=lat, latency_dist=rng.choice(["constant", "uniform", "exponential"]) if lat else "constant")
    if rng.random() < 0.25:
        kw["p_loss"] = rng.choice([0.02, 0.1])
    if rng.random() < 0.3 and n >= 3:
        kw.update(nemesis=["partition"], nemesis_interval=rng.choice([1, 3]))
    if rng.random() < 0.2:
        kw["journal_capacity"] = 1000000
    if kind == "raft":
        return "lin-kv", dict(kw, bin="raft")
    if kind == "proxy":
        return "lin-kv", dict(kw, bin="lin-kv-proxy", proxy_service=rng.choice(["lin-kv", "seq-kv", "lww-kv"]))
    kw.update(key_count=rng.choice([1, 3, 10]), max_txn_length=rng.choice([1, 4, 8]), max_writes_per_key=rng.choice([2, 16, 40]))
    if kind == "hat":
        kw["node_count"] = rng.choice([2, 2, 3, 5, 8])
        if rng.random() < 0.5:
            kw.update(nemesis=["partition"], nemesis_interval=rng.choice([1, 3]))
        return "txn-rw-register", kw
    if kind == "mk":
        kw["bin"] = "multi-key-txn"
        kw["node_count"] = rng.choice([1, 2, 3, 5, 6, 7, 12])
    if kind == "dt":
        kw["bin"] = "datomic"
        kw["node_count"] = rng.choice([1, 2, 3, 5, 6, 7, 12])
    return "txn-list-append", kw


@pytest.mark.parametrize("case", range(int(os.environ.get("MSIM_FUZZ_CASES", "24"))))
def test_random_multi_key_txn_options_engine_equals_oracle(lib, case):
    """The same sweep for the canonical txn-list-append node (multi_key_txn: thunks in lww-kv, the root map in lin-kv)."""
    rng = random.Random(0xD47A + case)
    os.environ["MSIM_FUZZ_KIND"] = "mk"
    try:
        wl, kw = _random_kv_case(rng)
    finally:
        del os.environ["MSIM_FUZZ_KIND"]
    try:
        cfg = E.test_config(wl, **kw)
        E.Engine(cfg).close()
    except E.EngineError as e:
        pytest.skip(str(e))
    _compare(cfg, rng.randrange(1 << 20), N_INST)


@pytest.mark.parametrize("case", range(int(os.environ.get("MSIM_FUZZ_CASES", "24"))))
def test_random_datomic_txn_options_engine_equals_oracle(lib, case):
    """The same sweep for the Datomic-style transactor node (datomic_list_append.rb: the hash tree in lww-kv, the root pointer in lin-kv)."""
    rng = random.Random(0xDA70 + case)
    os.environ["MSIM_FUZZ_KIND"] = "dt"
    try:
        wl, kw = _random_kv_case(rng)
    finally:
        del os.environ["MSIM_FUZZ_KIND"]
    try:
        cfg = E.test_config(wl, **kw)
        E.Engine(cfg).close()
    except E.EngineError as e:
        pytest.skip(str(e))
    first = rng.randrange(1 << 20)
    _compare(cfg, first, N_INST, dev_flags=0x400)   # eight clusters per wavefront where csrc/dt8.hip applies (else the same kernel again)
    _compare(cfg, first, 3)                          # one cluster per wavefront
    # ... and the same options with several workers per node (dtg_kernel<>: a lane per endpoint), where nodes + workers + 2 services fit a wavefront
    k = rng.choice([2, 3, 10])
    if kw["node_count"] * (k + 1) + 2 <= 64:
        many = E.test_config(wl, concurrency=k * kw["node_count"], **kw)
        _compare(many, first, 3)


@pytest.mark.parametrize("case", range(int(os.environ.get("MSIM_FUZZ_CASES", "24"))))
def test_random_rw_register_options_engine_equals_oracle(lib, case):
    """The same sweep for txn-rw-register over the highly-available-transactions node."""
    rng = random.Random(0xFEED + case)
    os.environ["MSIM_FUZZ_KIND"] = "hat"
    try:
        wl, kw = _random_kv_case(rng)
    finally:
        del os.environ["MSIM_FUZZ_KIND"]
    try:
        cfg = E.test_config(wl, **kw)
        E.Engine(cfg).close()
    except E.EngineError as e:
        pytest.skip(str(e))
    first = rng.randrange(1 << 20)
    _compare(cfg, first, N_INST)
    _compare(cfg, first, N_INST, dev_flags=0x400)       # eight clusters per wavefront where csrc/hat8.hip applies (else the same kernel again)
    if cfg.n_nodes <= 4:
        _compare(cfg, first, N_INST, dev_flags=0x8400)  # its 4-lane groups
    k = rng.choice([2, 3, 7])                           # ... and the same options with several workers per node (hatg_kernel<>: a lane per endpoint)
    if kw["node_count"] * (k + 1) + 2 <= 64:
        _compare(E.test_config(wl, concurrency=k * kw["node_count"], **kw), first, 3)


@pytest.mark.parametrize("case", range(int(os.environ.get("MSIM_FUZZ_CASES", "16"))))
def test_random_kafka_options_engine_equals_oracle(lib, case):
    """The same sweep for the kafka workload (logs in lin-kv chunks, committed offsets: csrc/sim_kernel_kafka.inc and csrc/kafka8.hip against oracle/kafka_nodes.inc)."""
    rng = random.Random(0xCAFCA + case)
    n = rng.choice([1, 2, 3, 5, 7])
    kw = dict(node_count=n, rate=rng.choice([20, 60, 150, 400]), time_limit=rng.choice([3, 6, 10]), seed=rng.randrange(1 << 40),
              key_count=rng.choice([1, 2, 4, 8]), max_writes_per_key=rng.choice([8, 40, 200, 1024]))
    # a test has at most 8 keys: a key retires after max-writes-per-key sends, so keep retirements below 8 - key-count
    sends = kw["rate"] * kw["time_limit"]
    kw["max_writes_per_key"] = min(2046, max(kw["max_writes_per_key"], sends // max(1, 8 - kw["key_count"]) + 8 if kw["key_count"] < 8 else sends + 8))
    lat = rng.choice([0, 1, 5, 20, 120])
    kw.update(latency=lat, latency_dist=rng.choice(["constant", "uniform", "exponential"]) if lat else "constant")
    if rng.random() < 0.25:
        kw["p_loss"] = rng.choice([0.02, 0.1])
    if rng.random() < 0.4 and n >= 3:
        kw.update(nemesis=["partition"], nemesis_interval=rng.choice([1, 3]))
    if rng.random() < 0.2:
        kw["journal_capacity"] = 1000000
    try:
        cfg = E.test_config("kafka", **kw)
        E.Engine(cfg).close()
    except E.EngineError as e:
        pytest.skip(str(e))
    first = rng.randrange(1 << 20)
    _compare(cfg, first, N_INST)
    if n <= 7 and not kw.get("journal_capacity"):
        _compare(cfg, first, N_INST + 5, dev_flags=0x400)   # eight clusters per wavefront (csrc/kafka8.hip; large batches take it unasked)
    k = rng.choice([2, 3, 8])                                  # ... and the same options with several workers per node (kafkag_kernel<>: a lane per endpoint)
    if n * (k + 1) + 1 <= 64:
        _compare(E.test_config("kafka", concurrency=k * n, **kw), first, 3)


@pytest.mark.parametrize("case", range(int(os.environ.get("MSIM_FUZZ_CASES", "36"))))
def test_random_kv_options_engine_equals_oracle(lib, case):
    """The same sweep for the key-value programs: Raft, the proxy over the three services, single-root transactions."""
    rng = random.Random(0xBADC0DE + case)
    wl, kw = _random_kv_case(rng)
    try:
        cfg = E.test_config(wl, **kw)
        E.Engine(cfg).close()
    except E.EngineError as e:
        pytest.skip(str(e))
    first = rng.randrange(1 << 20)
    _compare(cfg, first, N_INST)
    if kw.get("bin") == "lin-kv-proxy":   # four clusters per wavefront (svc4_kernel<>) where it applies
        _compare(cfg, first, N_INST, dev_flags=0x400)
    if wl == "txn-list-append" and kw.get("bin") in (None, "multi-key-txn"):   # the single-root and the multi-key node: the same options with several workers per node (txng_kernel<> / mkg_kernel<>)
        k = rng.choice([2, 3, 10])
        if kw["node_count"] * (k + 1) + 2 <= 64:
            _compare(E.test_config(wl, concurrency=k * kw["node_count"], **kw), first, 3)
            if kw.get("bin") is None:   # single-root node: four clusters per wavefront (txng4_kernel<>) where it applies
                _compare(E.test_config(wl, concurrency=k * kw["node_count"], **kw), first, 5, dev_flags=0x400)
    if wl == "txn-list-append" and kw.get("bin") == "datomic":   # the Datomic-style node with several workers per node: one cluster per wavefront (dtg_kernel<>) and four (dtg4_kernel<>)
        k = rng.choice([2, 3, 10])
        if kw["node_count"] * (k + 1) + 2 <= 16:
            _compare(E.test_config(wl, concurrency=k * kw["node_count"], **kw), first, 3)
            _compare(E.test_config(wl, concurrency=k * kw["node_count"], **kw), first, 5, dev_flags=0x400)


def _random_wide_case(rng):
    wl = rng.choice(["broadcast"] * 5 + ["g-set", "g-set", "pn-counter", "g-counter"])
    n = rng.choice([33, 34, 48, 63, 64, 65, 96, 97, 100, 127])
    kw = dict(node_count=n, rate=rng.choice([5, 20, 60, 200]), time_limit=rng.choice([2, 3, 5]), seed=rng.randrange(1 << 40))
    lat = rng.choice([0, 0, 1, 5, 20, 80])
    kw.update(latency=lat, latency_dist=rng.choice(["constant", "uniform", "exponential"]) if lat else "constant")
    if rng.random() < 0.3:
        kw["p_loss"] = rng.choice([0.02, 0.2])
    if rng.random() < 0.35:
        kw.update(nemesis=["partition"], nemesis_interval=rng.choice([1, 2]))
    if wl == "broadcast":
        kw["bin"] = rng.choice(["broadcast-ff", "broadcast-ff", "broadcast-ff-echoback", "broadcast-ack-retry", "broadcast-ack-retry", "broadcast-rpc-all"])
        kw["topology"] = rng.choice(["grid", "grid", "line", "total", "tree2", "tree3", "tree4"])
        if kw["topology"] == "total" or kw["bin"] == "broadcast-rpc-all":
            kw["rate"] = min(kw["rate"], 20)
    else:
        kw["time_limit"] = 6
    if rng.random() < 0.2:
        kw["journal_capacity"] = 4000000
    return wl, kw


@pytest.mark.parametrize("case", range(int(os.environ.get("MSIM_FUZZ_CASES", "24"))))
def test_random_wide_options_engine_equals_oracle(lib, case):
    """The same sweep over clusters of 33..127 nodes (sim_kernel_wide<>): g-set, the counters and the four broadcast programs."""
    rng = random.Random(0x51DE + case)
    wl, kw = _random_wide_case(rng)
    try:
        cfg = E.test_config(wl, **kw)
        E.Engine(cfg).close()
    except E.EngineError as e:
        pytest.skip(str(e))
    _compare(cfg, rng.randrange(1 << 20), 2)
