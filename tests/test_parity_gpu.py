"""GPU parity tests proper: the HIP engine (through the C-ABI) against the CPU oracle, bit-exact.

T0 parity: history rows, payload words, net stats and meta of every instance must be
byte-identical between libmaelsim.so (MI355X) and oracle/libmaelsim_oracle.so on the same (seed, instance)."""
import ctypes as C

import numpy as np
import pytest

from maelstrom_amd import _abi as A
from maelstrom_amd import engine as E
import oracle_lib as O

pytestmark = pytest.mark.gpu


def _compare(cfg, first, n, dev_flags=0):
    ora = O.run(cfg, first, n)
    with E.Engine(cfg) as eng:
        if dev_flags:
            eng.set_dev_flags(dev_flags)
        eng.run(first, n)
        eng.fetch()
        for i in range(n):
            m = eng.meta(i)
            om = ora.meta[i]
            assert (m.n_rows, m.n_payload_words, m.flags, m.n_rounds) == (om["n_rows"], om["n_payload_words"], om["flags"], om["n_rounds"]), \
                f"meta differs for instance {first + i}: gpu {(m.n_rows, m.n_payload_words, m.flags, m.n_rounds)} oracle {tuple(om)}"
            assert m.flags == 0
            rows, pay = eng.raw_history(i)
            orows, opay = ora.history(i)
            assert rows.tobytes() == orows.tobytes(), f"history rows differ for instance {first + i}"
            assert pay.tobytes() == opay.tobytes(), f"payload differs for instance {first + i}"
            assert m.n_events == om["n_events"]
            if cfg.journal_capacity:
                assert eng.raw_journal(i).tobytes() == ora.events(i).tobytes(), f"net journal differs for instance {first + i}"
            st = eng.net_stats_raw(i)
            assert tuple(getattr(st, f) for f, _ in A.NetStats._fields_) == tuple(int(x) for x in ora.stats[i]), \
                f"net stats differ for instance {first + i}"
    return ora


def test_wave_primitives_selftest(lib):
    """DPP min / prefix-sum encodings agree with shuffle-based references on the device."""
    lib.msim_selftest_wave.argtypes = [C.c_int]
    lib.msim_selftest_wave.restype = C.c_int
    assert lib.msim_selftest_wave(0) == 0


def test_echo_parity(lib):
    cfg = E.test_config("echo", node_count=3, rate=5, time_limit=10, seed=1)
    _compare(cfg, 0, 8)
    _compare(cfg, 0, 11, dev_flags=0x400)   # eight clusters per wavefront (csrc/uid8.hip; large batches take it unasked)


@pytest.mark.parametrize("wl,kw", [
    ("echo", dict(node_count=5, rate=400, time_limit=4, latency=3, p_loss=0.1)),                                                  # lost requests / replies: timeouts, fresh clients, stale replies
    ("echo", dict(node_count=8, rate=1000, time_limit=3, latency=0)),                                                             # a full 8-lane group
    ("echo", dict(node_count=1, rate=100, time_limit=3, latency=1, latency_dist="uniform", p_loss=0.3)),
    ("unique-ids", dict(node_count=3, rate=1000, time_limit=5, latency=5, nemesis=["partition"], nemesis_interval=2)),            # the reference's demo shape (core.clj:122-126)
    ("unique-ids", dict(node_count=7, rate=600, time_limit=4, latency=30, latency_dist="exponential", p_loss=0.05)),             # Reusable clients across timeouts
    ("unique-ids", dict(node_count=2, rate=300, time_limit=4, latency=1, inbox_capacity=1, p_loss=0.2)),
])
def test_client_only_programs_packed_layout_parity(lib, wl, kw):
    """echo and unique-ids (flake ids) — programs whose nodes talk to their clients only — eight clusters per wavefront (uid8_kernel<>,
    csrc/uid8.hip) against the oracle, 17 clusters (two full wavefronts and a partial one)."""
    cfg = E.test_config(wl, seed=37, **kw)
    _compare(cfg, 0, 17, dev_flags=0x400)


@pytest.mark.parametrize("wl,kw", [
    ("g-set", dict(node_count=5, rate=200, time_limit=12, latency=20, latency_dist="exponential", p_loss=0.1, nemesis=["partition"], nemesis_interval=3)),   # lost replicates, timeouts, the heal before the final reads
    ("g-set", dict(node_count=8, rate=400, time_limit=7, latency=5)),                                                                                     # a full 8-lane group; several readers per round
    ("g-set", dict(node_count=3, rate=100, time_limit=6, latency=300, inbox_capacity=2)),                                                                 # replicates older than the next tick; spilled queues
    ("pn-counter", dict(node_count=5, rate=100, time_limit=12, latency=100, latency_dist="uniform", nemesis=["partition"], nemesis_interval=4)),
    ("g-counter", dict(node_count=7, rate=150, time_limit=11, latency=30, latency_dist="exponential", p_loss=0.05)),
    ("pn-counter", dict(node_count=1, rate=50, time_limit=6)),
])
def test_crdt_programs_packed_layout_parity(lib, wl, kw):
    """g-set, pn-counter and g-counter eight clusters per wavefront (crdt8_kernel<>, csrc/crdt8.hip) against the oracle, 17 clusters."""
    cfg = E.test_config(wl, seed=53, **kw)
    _compare(cfg, 0, 17, dev_flags=0x400)


@pytest.mark.parametrize("bin,kw", [
    ("broadcast-ff", dict(node_count=5, rate=100, time_limit=8, latency=10, nemesis=["partition"], nemesis_interval=2)),                                       # what duo.hip does not take: partitions
    ("broadcast-ff", dict(node_count=8, rate=200, time_limit=5, latency=20, latency_dist="exponential", p_loss=0.1, topology="tree2")),                        # loss: client timeouts, :fail reads
    ("broadcast-ff-echoback", dict(node_count=6, rate=100, time_limit=5, latency=5, topology="line")),                                                          # (healthy network: taken from duo.hip with bit 15)
    ("broadcast-ack-retry", dict(node_count=5, rate=60, time_limit=10, latency=10, nemesis=["partition"], nemesis_interval=2, p_loss=0.1)),                    # retries pile up behind the partitions
    ("broadcast-ack-retry", dict(node_count=3, rate=100, time_limit=5, latency=200, latency_dist="uniform", inbox_capacity=2)),                                # retries before the first ack; spilled queues
    ("broadcast-rpc-all", dict(node_count=8, rate=60, time_limit=5, latency=20, latency_dist="exponential", topology="total")),                                # a full 8-lane group, 7 acks per value and node
    ("broadcast-rpc-all", dict(node_count=1, rate=50, time_limit=4)),
])
def test_broadcast_programs_packed_layout_parity(lib, bin, kw):
    """The four broadcast programs eight clusters per wavefront (bcast8_kernel<>, csrc/bcast8.hip) against the oracle, 17 clusters."""
    cfg = E.test_config("broadcast", bin=bin, seed=61, **kw)
    _compare(cfg, 0, 17, dev_flags=0x8400)


def test_broadcast_ack_retry_large_batch_takes_the_packed_layout(lib):
    """12288 clusters and more run eight per wavefront without being asked to (msim_launch_bcast8): every one of 12300 identical to the oracle."""
    cfg = E.test_config("broadcast", bin="broadcast-ack-retry", node_count=5, rate=20, time_limit=4, latency=10, nemesis=["partition"], nemesis_interval=2, seed=67)
    _compare(cfg, 0, 12300)


def test_pn_counter_large_batch_takes_the_packed_layout(lib):
    """4096 clusters and more run eight per wavefront without being asked to (msim_launch_crdt8): every one of 4100 identical to the oracle."""
    cfg = E.test_config("pn-counter", node_count=5, rate=50, time_limit=6, latency=10, seed=59)
    _compare(cfg, 0, 4100)


def test_unique_ids_large_batch_takes_the_packed_layout(lib):
    """8192 clusters and more run eight per wavefront without being asked to (msim_launch_uid8): every one of 8200 identical to the oracle."""
    cfg = E.test_config("unique-ids", node_count=3, rate=200, time_limit=2, latency=5, nemesis=["partition"], nemesis_interval=1, seed=43)
    _compare(cfg, 0, 8200)


@pytest.mark.parametrize("topology", ["grid", "line", "total", "tree4"])
def test_broadcast_ff_parity_small(lib, topology):
    cfg = E.test_config("broadcast", node_count=5, rate=10, time_limit=5, topology=topology, seed=7)
    _compare(cfg, 0, 16)


@pytest.mark.parametrize("latency,dist", [(0, "constant"), (10, "constant"), (100, "constant"), (100, "exponential"), (50, "uniform")])
def test_broadcast_n25_parity(lib, latency, dist):
    cfg = E.test_config("broadcast", node_count=25, rate=100, time_limit=20, latency=latency, latency_dist=dist, seed=42)
    _compare(cfg, 100, 4)


@pytest.mark.parametrize("prog", ["broadcast-ff-echoback", "broadcast-rpc-all", "broadcast-ack-retry"])
def test_broadcast_variants_parity(lib, prog):
    cfg = E.test_config("broadcast", bin=prog, node_count=5, rate=20, time_limit=5, latency=10, seed=3)
    _compare(cfg, 0, 8)


def test_broadcast_partition_parity(lib):
    cfg = E.test_config("broadcast", bin="broadcast-ack-retry", node_count=5, rate=10, time_limit=20, topology="tree4",
                        nemesis=["partition"], nemesis_interval=5, latency=5, seed=11)
    _compare(cfg, 0, 16)


def test_broadcast_loss_parity(lib):
    cfg = E.test_config("broadcast", bin="broadcast-ack-retry", node_count=9, rate=20, time_limit=10, latency=20,
                        latency_dist="exponential", p_loss=0.05, seed=5)
    _compare(cfg, 0, 8)


def test_g_set_parity(lib):
    cfg = E.test_config("g-set", node_count=5, rate=10, time_limit=10, seed=9)
    _compare(cfg, 0, 8)
    _compare(cfg, 0, 11, dev_flags=0x400)   # eight clusters per wavefront (csrc/crdt8.hip; large batches take it unasked)
    cfg = E.test_config("g-set", node_count=25, rate=100, time_limit=10, latency=100, latency_dist="exponential", seed=9)
    _compare(cfg, 0, 4)


@pytest.mark.parametrize("n,kw", [
    (40, dict(latency=0)),
    (64, dict(latency=20, latency_dist="uniform")),
    (65, dict(latency=50, latency_dist="exponential", p_loss=0.05)),
    (100, dict(latency=100, latency_dist="exponential")),
    (100, dict(latency=100, latency_dist="exponential", p_loss=0.5)),
    (127, dict(latency=10)),
])
def test_wide_g_set_parity(lib, n, kw):
    """BASELINE cfg3: clusters wider than 32 nodes (two node/client pairs per lane, sim_kernel_wide<>), with
    randomized latency and message loss; same oracle code path as narrow clusters.  Both layouts of the nodes' sets: in HBM scratch
    (the default since replicate deliveries are merged lazily) and in LDS (MSIM_DEV_FLAGS bit 14: sim_kernel_wide<.., SETL = true>);
    with the lone-operation path (default) and with every operation on the general path (bit 1), in either layout."""
    cfg = E.test_config("g-set", node_count=n, rate=100, time_limit=12, seed=77, **kw)
    _compare(cfg, 0, 3)
    _compare(cfg, 0, 3, dev_flags=0x4000)
    _compare(cfg, 0, 2, dev_flags=0x2)
    _compare(cfg, 0, 2, dev_flags=0x4002)


@pytest.mark.parametrize("wl,n,kw", [
    ("g-set", 100, dict(rate=100, time_limit=11, latency=100, latency_dist="exponential")),                 # cfg3's shape: bursts of ~1400 rounds
    ("g-set", 45, dict(rate=50, time_limit=12, latency=3000, latency_dist="exponential")),                  # ticks overlap (stale slots stop a node in the middle of a time); in flight at the END of the run
    ("g-set", 60, dict(rate=100, time_limit=11, latency=400, latency_dist="uniform", p_loss=0.2)),          # lost requests: client timeouts bound the windows
    ("g-set", 70, dict(rate=100, time_limit=11, latency=7)),                                                # constant latency: every replicate of a tick due at one time
    ("pn-counter", 50, dict(rate=100, time_limit=11, latency=200, latency_dist="exponential")),             # time jumps of more than the table's 1024 ms
    ("g-counter", 70, dict(rate=100, time_limit=16, latency=1000, latency_dist="exponential")),
    ("g-set", 127, dict(rate=200, time_limit=11, latency=30, latency_dist="uniform")),
])
def test_wide_quiet_windows_keep_the_rounds(lib, wl, n, kw):
    """sim_kernel_wide<>'s quiet windows (replicate deliveries made ahead of the rounds, each node at its own times; sim_kernel_wide.inc R0)
    leave every word of the result as it was — the round count included, which they reconstruct from per-millisecond maxima — with the windows
    (default) and without (MSIM_DEV_FLAGS bit 2), against the oracle, which knows no windows."""
    cfg = E.test_config(wl, node_count=n, seed=123, **kw)
    _compare(cfg, 0, 4)
    _compare(cfg, 0, 2, dev_flags=0x4)


@pytest.mark.parametrize("n,kw", [
    (33, dict(latency=0)),
    (36, dict(latency=10, topology="line")),
    (64, dict(latency=20, latency_dist="uniform", topology="tree4")),
    (65, dict(latency=50, latency_dist="exponential", p_loss=0.05)),
    (100, dict(latency=100, latency_dist="exponential")),
    (100, dict(latency=0, bin="broadcast-ff-echoback")),
    (50, dict(latency=5, topology="total", rate=20, time_limit=4)),
    (127, dict(latency=10, topology="tree2")),
])
def test_wide_broadcast_parity(lib, n, kw):
    """Fire-and-forget broadcast on clusters wider than 32 nodes (the reference advertises "25+ nodes", README.md:40): every
    topology of broadcast.clj:40-185, both gossip variants, randomized latency and loss — sim_kernel_wide<NET_RANDOM, BCAST>."""
    kw = dict(dict(rate=100, time_limit=8), **kw)
    cfg = E.test_config("broadcast", node_count=n, seed=79, **kw)
    _compare(cfg, 0, 3)


@pytest.mark.parametrize("wl,n,kw", [
    ("g-set", 33, dict(latency=0)),
    ("g-set", 70, dict(latency=30, latency_dist="exponential", p_loss=0.05)),
    ("g-set", 127, dict(latency=10, nemesis_interval=1)),
    ("broadcast", 40, dict(latency=5)),
    ("broadcast", 64, dict(latency=20, latency_dist="uniform", topology="tree3", nemesis_interval=1)),
    ("broadcast", 100, dict(latency=50, latency_dist="exponential", p_loss=0.02)),
    ("broadcast", 127, dict(latency=0, topology="line", bin="broadcast-ff-echoback")),
])
def test_wide_partition_parity(lib, wl, n, kw):
    """The partition nemesis on clusters wider than 32 nodes: 128-bit grudges (all four specs come up over 16 seeds x several
    partitions), drops at poll time, the heal before the final reads — sim_kernel_wide<.., NEM = true>."""
    kw = dict(dict(rate=60, time_limit=10, nemesis=["partition"], nemesis_interval=2), **kw)
    cfg = E.test_config(wl, node_count=n, seed=81, **kw)
    ora = _compare(cfg, 0, 4)
    specs = set()
    for i in range(4):
        rows, _ = ora.history(i)
        nem = rows[(rows["packed"] >> 12) == A.PROCESS_NEMESIS]
        assert len(nem) >= 4
        specs |= {int(v) for v, pk in zip(nem["value"][::2], nem["packed"][::2]) if ((pk >> 2) & 31) == A.F_START_PARTITION}
    assert len(specs) >= 2


@pytest.mark.parametrize("prog,n,kw", [
    ("broadcast-ack-retry", 33, dict(latency=0)),
    ("broadcast-ack-retry", 50, dict(latency=20, latency_dist="exponential", p_loss=0.1)),
    ("broadcast-ack-retry", 64, dict(latency=10, topology="tree4", nemesis=["partition"], nemesis_interval=2)),
    ("broadcast-ack-retry", 100, dict(latency=50, latency_dist="uniform", p_loss=0.05, nemesis=["partition"], nemesis_interval=3)),
    ("broadcast-ack-retry", 127, dict(latency=5, topology="line", rate=20)),
    ("broadcast-rpc-all", 40, dict(latency=10, rate=20)),
    ("broadcast-rpc-all", 100, dict(latency=20, latency_dist="exponential", p_loss=0.05, rate=10, time_limit=5)),
    ("broadcast-rpc-all", 70, dict(latency=5, rate=10, time_limit=6, nemesis=["partition"], nemesis_interval=2)),
])
def test_wide_acknowledged_broadcast_parity(lib, prog, n, kw):
    """The acknowledged variants on clusters wider than 32 nodes: 128-bit unacked sets per (node, value), the 1 s retry FIFO, node
    msg_ids, acknowledgements crossing lanes (02-performance.md:406-441; demo/ruby/broadcast.rb:29-47) — sim_kernel_wide<.., WP 2 / 3, ..>."""
    kw = dict(dict(rate=40, time_limit=8), **kw)
    cfg = E.test_config("broadcast", bin=prog, node_count=n, seed=83, **kw)
    _compare(cfg, 0, 3)


@pytest.mark.parametrize("wl,n,kw", [
    ("pn-counter", 33, dict(latency=0)),
    ("pn-counter", 64, dict(latency=20, latency_dist="exponential", p_loss=0.05)),
    ("pn-counter", 100, dict(latency=50, latency_dist="uniform", nemesis=["partition"], nemesis_interval=3)),
    ("g-counter", 40, dict(latency=10)),
    ("g-counter", 127, dict(latency=30, latency_dist="exponential", p_loss=0.02, nemesis=["partition"], nemesis_interval=4)),
])
def test_wide_counter_parity(lib, wl, n, kw):
    """The PN / G counter CRDT (demo/ruby/pn_counter.rb) on clusters wider than 32 nodes: 2 N counters per node in HBM scratch, merged by
    maximum by the whole wavefront, reads summed by it — sim_kernel_wide<.., WP 4, ..>."""
    kw = dict(dict(rate=60, time_limit=12), **kw)
    cfg = E.test_config(wl, node_count=n, seed=85, **kw)
    _compare(cfg, 0, 3)


def test_wide_broadcast_journal_parity(lib):
    cfg = E.test_config("broadcast", node_count=70, rate=50, time_limit=6, latency=30, latency_dist="exponential", p_loss=0.05,
                        seed=80, journal_capacity=400000)
    ora = _compare(cfg, 0, 2)
    assert (ora.meta["n_events"] > 10000).all()


def test_wide_g_set_journal_parity(lib):
    cfg = E.test_config("g-set", node_count=70, rate=50, time_limit=6, latency=30, latency_dist="exponential", p_loss=0.05,
                        seed=78, journal_capacity=200000)
    ora = _compare(cfg, 0, 2)
    assert (ora.meta["n_events"] > 10000).all()


@pytest.mark.parametrize("conc", [3, 10, 7])
def test_general_layout_parity_when_concurrency_differs_from_nodes(lib, conc):
    """concurrency != n_nodes uses the general (endpoint-per-lane) kernel instead of the colocated one."""
    cfg = E.test_config("broadcast", node_count=5, concurrency=conc, rate=20, time_limit=5, latency=10, seed=21)
    _compare(cfg, 0, 8)
    cfg = E.test_config("broadcast", bin="broadcast-ack-retry", node_count=5, concurrency=conc, rate=20, time_limit=8,
                        nemesis=["partition"], nemesis_interval=3, latency=5, latency_dist="uniform", seed=22)
    _compare(cfg, 0, 8)


@pytest.mark.parametrize("kw", [
    dict(node_count=2, rate=20, time_limit=2, latency=2, key_count=3),
    dict(node_count=3, rate=50, time_limit=4, latency=5),
    dict(node_count=5, rate=100, time_limit=8, latency=10, latency_dist="exponential", nemesis=["partition"], nemesis_interval=2),
    dict(node_count=3, rate=200, time_limit=3, latency=3, latency_dist="uniform", key_count=2, max_txn_length=8, max_writes_per_key=32),
    dict(node_count=7, rate=100, time_limit=4, latency=0),
    dict(node_count=5, rate=100, time_limit=6, latency=5, p_loss=0.05, journal_capacity=400000),
])
def test_multi_key_txn_parity(lib, kw):
    """The canonical txn-list-append node (thunks in lww-kv, root map in lin-kv) against oracle/mk_nodes.inc, which the reference's own
    multi_key_txn.js pins on the process bridge (tests/test_process_bridge.py).  Both layouts: eight clusters per wavefront (mk8_kernel<>,
    csrc/mk8.hip: what the engine picks for <= 6 nodes, --max-txn-length <= 4, journal off) and one cluster per wavefront (mk_kernel<>:
    every other shape, and every shape under MSIM_DEV_FLAGS bit 9)."""
    cfg = E.test_config("txn-list-append", bin="multi-key-txn", seed=91, **kw)
    _compare(cfg, 0, 11)               # (11 clusters: a full group of eight and a partial one)
    _compare(cfg, 0, 4, dev_flags=0x200)


@pytest.mark.parametrize("kw", [
    dict(node_count=5, rate=100, time_limit=20, latency=100, latency_dist="exponential", p_loss=0.1, nemesis=["partition"], nemesis_interval=4),   # timeouts: several transactions in flight per node (the slots in HBM scratch), deep service queues
    dict(node_count=6, rate=300, time_limit=6, latency=20, latency_dist="uniform", key_count=2),                                                   # the widest group: 6 nodes + 2 services; contended keys
    dict(node_count=1, rate=50, time_limit=5, latency=1),
])
def test_multi_key_txn_packed_layout_parity(lib, kw):
    """Shapes that stress what is specific to mk8_kernel<>: transaction slots beyond the LDS one, spilled queues, a full 8-lane group."""
    cfg = E.test_config("txn-list-append", bin="multi-key-txn", seed=17, **kw)
    _compare(cfg, 0, 16)


@pytest.mark.parametrize("kw", [
    dict(node_count=2, rate=20, time_limit=2, latency=2, key_count=3),
    dict(node_count=3, rate=50, time_limit=4, latency=5),
    dict(node_count=5, rate=100, time_limit=8, latency=10, latency_dist="exponential", nemesis=["partition"], nemesis_interval=2),
    dict(node_count=3, rate=200, time_limit=3, latency=3, latency_dist="uniform", key_count=2, max_txn_length=8, max_writes_per_key=32),
    dict(node_count=7, rate=100, time_limit=4, latency=0),
    dict(node_count=5, rate=100, time_limit=6, latency=5, p_loss=0.05, journal_capacity=400000),                 # lost messages: clients time out, transactions queue behind the node's lock
    dict(node_count=3, rate=150, time_limit=10, latency=0, key_count=16, max_writes_per_key=2),                 # ~1000 keys: splits at every level, chains in the two-wide ranges
    dict(node_count=5, rate=60, time_limit=20, latency=10, p_loss=0.02),                                        # long enough for Promise#await's 5 s: the lock holder gives up (error 0), the node recovers
    dict(node_count=3, rate=80, time_limit=25, latency=30, latency_dist="exponential", p_loss=0.05),
    dict(node_count=1, rate=50, time_limit=5, latency=1),
    dict(node_count=12, rate=200, time_limit=5, latency=20, latency_dist="exponential"),
])
def test_datomic_txn_parity(lib, kw):
    """The node core.clj:113-114 runs for txn-list-append (demo/ruby/datomic_list_append.rb: a persistent hash tree in lww-kv, the root pointer
    in lin-kv, lazily loaded paths, a lock per node) — dt8_kernel<> (csrc/dt8.hip) and dt_kernel<> (csrc/sim_kernel_dt.inc) against oracle/dt_nodes.inc, which
    tests/test_datomic_tree.py holds to the Ruby classes written out in Python.  Parity with the reference itself is unpinned (no Ruby)."""
    cfg = E.test_config("txn-list-append", bin="datomic", seed=91, **kw)
    _compare(cfg, 0, 11, dev_flags=0x400)   # eight clusters per wavefront (dt8_kernel<>, csrc/dt8.hip: up to 6 nodes with the journal off; large launches take it unasked); 11 = a full group of eight and a partial one
    _compare(cfg, 0, 4)                     # one cluster per wavefront (dt_kernel<>)


@pytest.mark.parametrize("kw,first", [
    (dict(node_count=3, rate=20, time_limit=60, latency=1200, latency_dist="exponential", key_count=3), 4),   # instance 4: a cas served 5 s after it was sent, cas_ok
    (dict(node_count=2, rate=15, time_limit=60, latency=1300, latency_dist="exponential"), 10),                # instances 14 and 18 likewise
])
def test_datomic_late_cas_parity(lib, kw, first):
    """Latencies of seconds: a cas reaches lin-kv after its sender's Promise#await gave up and the sender holds the next transaction.  The request
    is self-contained (datomic_list_append.rb:376-388: from / to travel in the message; here in the sender's table of cas requests under the msg_id),
    so the service commits — or refuses — the transaction that SENT it, in the oracle (held to the reference classes on such runs by
    tests/test_datomic_tree.py::test_a_cas_served_after_its_sender_gave_up_is_still_its_own) and in both kernels."""
    cfg = E.test_config("txn-list-append", bin="datomic", seed=91, **kw)
    _compare(cfg, first, 11, dev_flags=0x400)
    _compare(cfg, first, 4)


@pytest.mark.parametrize("kw", [
    dict(node_count=1, concurrency=10, rate=100, time_limit=8, latency=0),                                     # doc/05-datomic/01-single-node.md:257: --node-count 1 --concurrency 10n --rate 100
    dict(node_count=2, concurrency=20, rate=300, time_limit=5, latency=3, latency_dist="uniform"),
    dict(node_count=5, concurrency=10, rate=200, time_limit=6, latency=5, nemesis=["partition"], nemesis_interval=2),
    dict(node_count=3, concurrency=9, rate=150, time_limit=20, latency=10, p_loss=0.03, journal_capacity=400000),   # lost messages: awaits give up while others wait for the lock
    dict(node_count=6, concurrency=48, rate=400, time_limit=4, latency=20, latency_dist="exponential"),       # 6 + 48 + 2 = 56 lanes
    dict(node_count=1, concurrency=61, rate=1000, time_limit=3, latency=1),                                    # a full wavefront: 1 node, 61 workers, lin-kv, lww-kv
    dict(node_count=1, concurrency=10, rate=100, time_limit=30, latency=5),                                    # the reference's invocation at bench length
    dict(node_count=2, concurrency=12, rate=300, time_limit=6, latency=10, latency_dist="exponential", p_loss=0.05, nemesis=["partition"], nemesis_interval=2),   # a full 16-lane group; lost messages: awaits give up
    dict(node_count=1, concurrency=13, rate=400, time_limit=5, latency=3, latency_dist="uniform", key_count=16, max_writes_per_key=2),   # 16 lanes; many keys: splits, chains
    dict(node_count=2, concurrency=4, rate=15, time_limit=60, latency=1300, latency_dist="exponential"),       # latencies of seconds: a cas served after its sender's await gave up
])
def test_datomic_many_workers_parity(lib, kw):
    """Several workers per node (`--concurrency k n`): dtg_kernel<> (csrc/sim_kernel_dtg.inc: a lane per endpoint) and dtg4_kernel<> (csrc/dtg4.hip:
    four clusters per wavefront where nodes + workers + lin-kv + lww-kv <= 16 and the journal is off; large launches take it unasked, here
    MSIM_DEV_FLAGS bit 10 asks for it) against oracle/dt_nodes.inc, which
    tests/test_datomic_tree.py::test_several_workers_per_node_queue_behind_the_lock_in_arrival_order holds to the reference classes."""
    cfg = E.test_config("txn-list-append", bin="datomic", seed=23, **kw)
    _compare(cfg, 0, 5)
    _compare(cfg, 2, 6, dev_flags=0x400)


@pytest.mark.parametrize("kw", [
    dict(node_count=1, concurrency=10, rate=100, time_limit=8, latency=0),
    dict(node_count=2, concurrency=20, rate=300, time_limit=5, latency=3, latency_dist="uniform"),
    dict(node_count=5, concurrency=10, rate=200, time_limit=6, latency=5, nemesis=["partition"], nemesis_interval=2),
    dict(node_count=3, concurrency=9, rate=150, time_limit=10, latency=10, p_loss=0.03, journal_capacity=400000),
    dict(node_count=1, concurrency=61, rate=1000, time_limit=3, latency=1),                                    # 1 node, 61 workers, lin-kv
    dict(node_count=1, concurrency=10, rate=100, time_limit=30, latency=5),                                    # the shape of the reference's own runs (doc/05-datomic/01-single-node.md:257)
    dict(node_count=5, concurrency=10, rate=400, time_limit=6, latency=20, latency_dist="exponential", p_loss=0.1, nemesis=["partition"], nemesis_interval=2),   # a full 16-lane group; loss: clients time out, queues beyond their LDS slots
    dict(node_count=2, concurrency=8, rate=300, time_limit=5, latency=3, latency_dist="uniform", key_count=3, max_txn_length=8, max_writes_per_key=40),
    dict(node_count=3, concurrency=12, rate=200, time_limit=5, latency=5, inbox_capacity=2, spill_capacity=40),  # 16 lanes; the servers' queues live in the spill
    dict(node_count=1, concurrency=14, rate=500, time_limit=4, latency=8),                                      # 16 lanes: 1 node, 14 workers, lin-kv
])
def test_single_key_txn_many_workers_parity(lib, kw):
    """Several workers per node for the single-root node (demo/clojure/single_key_txn.clj): txng_kernel<> (csrc/sim_kernel_txng.inc: a lane per
    endpoint, up to 64 transactions in flight per node) and txng4_kernel<> (csrc/txng4.hip: four clusters per wavefront where nodes + workers +
    lin-kv <= 16 and the journal is off; large launches take it unasked, here MSIM_DEV_FLAGS bit 10 asks for it; 6 clusters = a full wavefront
    and a partial one) against oracle/txn_nodes.inc."""
    cfg = E.test_config("txn-list-append", seed=29, **kw)
    _compare(cfg, 0, 5)
    _compare(cfg, 2, 6, dev_flags=0x400)


@pytest.mark.parametrize("kw", [
    dict(node_count=1, concurrency=10, rate=100, time_limit=8, latency=0),
    dict(node_count=2, concurrency=20, rate=300, time_limit=5, latency=3, latency_dist="uniform"),
    dict(node_count=5, concurrency=10, rate=200, time_limit=6, latency=5, nemesis=["partition"], nemesis_interval=2),
    dict(node_count=3, concurrency=9, rate=150, time_limit=8, latency=10, latency_dist="exponential", journal_capacity=600000),
    dict(node_count=2, concurrency=8, rate=200, time_limit=4, latency=2, key_count=3, max_txn_length=8, max_writes_per_key=32),   # eight keys per transaction slot
    dict(node_count=1, concurrency=61, rate=1000, time_limit=3, latency=1),                                    # a full wavefront: 1 node, 61 workers, lin-kv, lww-kv
])
def test_multi_key_txn_many_workers_parity(lib, kw):
    """Several workers per node for the multi-key node (demo/js/multi_key_txn.js): mkg_kernel<> (csrc/sim_kernel_mkg.inc: a lane per endpoint, up
    to 64 transactions in flight per node) against oracle/mk_nodes.inc, which tests/test_process_bridge.py holds to the real program with
    several workers per node."""
    cfg = E.test_config("txn-list-append", bin="multi-key-txn", seed=31, **kw)
    _compare(cfg, 0, 5)


@pytest.mark.parametrize("kw", [
    dict(node_count=2, concurrency=10, rate=100, time_limit=8, nemesis=["partition"], nemesis_interval=2),   # the reference's demo shape (core.clj:115-121) with --concurrency 5n
    dict(node_count=2, concurrency=20, rate=300, time_limit=5, latency=3, latency_dist="uniform"),
    dict(node_count=5, concurrency=10, rate=200, time_limit=6, latency=5, nemesis=["partition"], nemesis_interval=2),
    dict(node_count=3, concurrency=9, rate=150, time_limit=8, latency=10, latency_dist="exponential", p_loss=0.03, journal_capacity=600000),
    dict(node_count=8, concurrency=56, rate=500, time_limit=3, latency=2),                                    # a full wavefront: 8 nodes, 56 workers
])
def test_rw_register_many_workers_parity(lib, kw):
    """Several workers per node for txn-rw-register's node (demo/clojure/txn_rw_register_hat.clj): hatg_kernel<> (csrc/sim_kernel_hatg.inc: a lane
    per endpoint) against oracle/hat_nodes.inc."""
    cfg = E.test_config("txn-rw-register", seed=37, **kw)
    _compare(cfg, 0, 5)


@pytest.mark.parametrize("kw", [
    dict(node_count=1, concurrency=10, rate=100, time_limit=8, latency=0),
    dict(node_count=2, concurrency=20, rate=300, time_limit=5, latency=3, latency_dist="uniform"),
    dict(node_count=5, concurrency=10, rate=200, time_limit=8, latency=5, nemesis=["partition"], nemesis_interval=3),
    dict(node_count=3, concurrency=9, rate=150, time_limit=8, latency=10, latency_dist="exponential", p_loss=0.03, journal_capacity=600000),
    dict(node_count=1, concurrency=12, rate=150, time_limit=6, latency=2, key_count=2, max_writes_per_key=200),  # full chunks (32 messages): the send path recurs
    dict(node_count=3, concurrency=60, rate=1000, time_limit=3, latency=1),                                    # a full wavefront: 3 nodes, 60 workers, lin-kv
])
def test_kafka_many_workers_parity(lib, kw):
    """Several workers per node for the kafka workload (demo/clojure/kafka.clj): kafkag_kernel<> (csrc/sim_kernel_kafkag.inc: a lane per endpoint,
    up to 64 request handlers in flight per node, a client's offsets and assign / poll / commit state in its own lane) against oracle/kafka_nodes.inc."""
    cfg = E.test_config("kafka", seed=41, **kw)
    _compare(cfg, 0, 5)


def test_deep_queues_spill_to_hbm(lib):
    """Exponential latency => long head-of-line sleeps => queues far deeper than the LDS part: the HBM spill area
    behind each node's queue keeps the result bit-identical (and unflagged)."""
    cfg = E.test_config("broadcast", node_count=25, rate=100, time_limit=20, latency=100, latency_dist="exponential", seed=8)
    assert cfg.spill_capacity > 100
    _compare(cfg, 0, 4)
    cfg = E.test_config("broadcast", node_count=5, rate=50, time_limit=5, latency=200, latency_dist="exponential", seed=9, inbox_capacity=2)
    _compare(cfg, 0, 8)


@pytest.mark.parametrize("bin,conc,kw", [
    ("broadcast-ff", 5, dict(latency=0)),
    ("broadcast-ff", 5, dict(latency=20, latency_dist="exponential", p_loss=0.05)),
    ("broadcast-ack-retry", 5, dict(latency=10, nemesis=["partition"], nemesis_interval=3)),
    ("broadcast-rpc-all", 7, dict(latency=5)),
    ("broadcast-ff", 3, dict(latency=30, latency_dist="uniform")),
])
def test_net_journal_parity(lib, bin, conc, kw):
    """journal.clj:53,220-239: every send!/recv! as an event, bit-identical (ids, order, times) to the oracle,
    in both kernel layouts (concurrency == / != node count)."""
    cfg = E.test_config("broadcast", bin=bin, node_count=5, concurrency=conc, rate=20, time_limit=6, seed=31,
                        journal_capacity=60000, **kw)
    ora = _compare(cfg, 0, 8)
    assert (ora.meta["n_events"] > 100).all()
    cfg = E.test_config("g-set", node_count=5, rate=10, time_limit=10, seed=32, journal_capacity=20000)
    _compare(cfg, 0, 4)


@pytest.mark.parametrize("kw", [
    dict(),
    dict(latency=10),
    dict(latency=20, latency_dist="exponential", p_loss=0.02),
    dict(nemesis=["partition"], nemesis_interval=5, latency=5),
    dict(node_count=3, concurrency=6, nemesis=["partition"], nemesis_interval=4, time_limit=30, latency=10, latency_dist="uniform"),
    dict(journal_capacity=200000, latency=5),
])
def test_raft_lin_kv_parity(lib, kw):
    """Raft lin-kv (raft.rb / raft.py restated): elections, replication, commit/apply, proxying, timeouts, key rotation —
    history, stats, round count (and journal) bit-identical to the oracle."""
    kw = dict(dict(node_count=5, rate=30, time_limit=20, seed=17), **kw)
    cfg = E.test_config("lin-kv", bin="raft", **kw)
    _compare(cfg, 0, 6)


@pytest.mark.parametrize("kw", [
    dict(),
    dict(latency=5),
    dict(latency=20, latency_dist="exponential"),
    dict(latency=5, nemesis=["partition"], nemesis_interval=3),
    dict(latency=10, latency_dist="uniform", p_loss=0.02),
    dict(node_count=3, rate=200, latency=2),
    dict(node_count=9, rate=100, latency=50),
])
def test_txn_list_append_parity(lib, kw):
    """BASELINE configs[4]: single-root transactional nodes over the lin-kv service (txn_kernel<>)."""
    base = dict(node_count=5, rate=50, time_limit=10, seed=61)
    base.update(kw)
    cfg = E.test_config("txn-list-append", **base)
    _compare(cfg, 0, 6)


def test_txn_list_append_journal_parity(lib):
    cfg = E.test_config("txn-list-append", node_count=5, rate=50, time_limit=6, latency=5, nemesis=["partition"], nemesis_interval=2,
                        seed=62, journal_capacity=40000)
    ora = _compare(cfg, 0, 4)
    assert (ora.meta["n_events"] > 2000).all()


@pytest.mark.parametrize("kw", [
    dict(),
    dict(latency=5),
    dict(nemesis=["partition"], nemesis_interval=2),                                   # the reference's demo shape (core.clj:115-121)
    dict(latency=20, latency_dist="exponential", p_loss=0.05),
    dict(node_count=3, latency=10, latency_dist="uniform", nemesis=["partition"], nemesis_interval=2),
    dict(node_count=5, rate=200, latency=5, nemesis=["partition"], nemesis_interval=3),  # third parties relay (hat :143-150)
    dict(node_count=8, rate=50, latency=30),
    dict(journal_capacity=60000, latency=5, nemesis=["partition"], nemesis_interval=2),
    dict(key_count=3, max_txn_length=8, max_writes_per_key=40, rate=300),
])
def test_txn_rw_register_parity(lib, kw):
    """workload/txn_rw_register.clj over demo/clojure/txn_rw_register_hat.clj.  Both layouts: eight clusters per wavefront (hat8_kernel<>,
    csrc/hat8.hip: what the engine picks for batches of 8192 clusters and more with --max-txn-length <= 4 and the journal off — asked for
    here with MSIM_DEV_FLAGS bit 10 —, also in its 4-lane groups, bit 15) and one cluster per wavefront (hat_kernel<>: everything else)."""
    base = dict(node_count=2, rate=100, time_limit=8, seed=101)
    base.update(kw)
    cfg = E.test_config("txn-rw-register", **base)
    ora = _compare(cfg, 0, 11, dev_flags=0x400)   # (11 clusters: a full wavefront of eight and a partial one)
    assert (ora.stats["servers_send"] > 0).all()   # replicate / replicate_ack traffic did flow
    if cfg.n_nodes <= 4:
        _compare(cfg, 0, 19, dev_flags=0x8400)
    _compare(cfg, 0, 4)


@pytest.mark.parametrize("kw", [
    dict(node_count=4, rate=200, time_limit=8, latency=150, latency_dist="exponential", p_loss=0.1, nemesis=["partition"], nemesis_interval=3),   # a full 4-lane group; timeouts, spilled queues, stale replies
    dict(node_count=8, rate=300, time_limit=6, latency=5, latency_dist="uniform", nemesis=["partition"], nemesis_interval=2),                      # a full 8-lane group; relays; long lists behind the partitions
    dict(node_count=2, rate=400, time_limit=10, latency=2, nemesis=["partition"], nemesis_interval=5, key_count=2),                                 # hundreds of unreplicated txns per tick
    dict(node_count=3, rate=100, time_limit=5, latency=20, inbox_capacity=2),                                                                      # every queue spills
])
def test_txn_rw_register_packed_layout_parity(lib, kw):
    """Shapes that stress what is specific to hat8_kernel<>: both group sizes filled, queues beyond the LDS slots, the tick's list built
    by the lanes of a group."""
    cfg = E.test_config("txn-rw-register", seed=23, **kw)
    _compare(cfg, 0, 17, dev_flags=0x400)
    if cfg.n_nodes <= 4:
        _compare(cfg, 0, 17, dev_flags=0x8400)


def test_txn_rw_register_large_batch_takes_the_packed_layout(lib):
    """8192 clusters and more run eight per wavefront without being asked to (msim_launch_hat8): every one of 8200 identical to the oracle."""
    cfg = E.test_config("txn-rw-register", node_count=2, rate=100, time_limit=2, nemesis=["partition"], nemesis_interval=1, seed=29)
    _compare(cfg, 0, 8200)


@pytest.mark.parametrize("kw", [
    dict(),
    dict(latency=20),
    dict(latency=50, latency_dist="exponential", p_loss=0.1),
    dict(latency=10, nemesis=["partition"], nemesis_interval=4, time_limit=20),
    dict(node_count=25, rate=100, latency=100, latency_dist="uniform"),
    dict(concurrency=10, latency=5),
    dict(journal_capacity=60000, latency=5),
])
def test_pn_counter_parity(lib, kw):
    """workload/pn_counter.clj over the CRDT node of demo/ruby/pn_counter.rb, both kernel layouts."""
    base = dict(node_count=5, rate=20, time_limit=12, seed=71)
    base.update(kw)
    cfg = E.test_config("pn-counter", **base)
    _compare(cfg, 0, 6)
    if cfg.concurrency == cfg.n_nodes and cfg.n_nodes <= 8 and not cfg.journal_capacity:
        _compare(cfg, 0, 9, dev_flags=0x400)   # eight clusters per wavefront (csrc/crdt8.hip)


def test_g_counter_parity(lib):
    cfg = E.test_config("g-counter", node_count=5, rate=50, time_limit=10, latency=10, seed=72)
    _compare(cfg, 0, 6)
    cfg = E.test_config("g-counter", node_count=7, concurrency=14, rate=50, time_limit=8, latency=30, latency_dist="exponential", p_loss=0.05, seed=73)
    _compare(cfg, 0, 4)


@pytest.mark.parametrize("kw", [dict(), dict(concurrency=9, latency=20), dict(latency=30, latency_dist="exponential", p_loss=0.05, nemesis=["partition"], nemesis_interval=3)])
def test_unique_ids_parity(lib, kw):
    cfg = E.test_config("unique-ids", node_count=3, rate=200, time_limit=5, seed=81, **kw)
    _compare(cfg, 0, 6)
    if cfg.concurrency == cfg.n_nodes:
        _compare(cfg, 0, 9, dev_flags=0x400)   # eight clusters per wavefront (csrc/uid8.hip)
    with E.Engine(cfg) as eng:
        eng.run(0, 6)
        eng.check()
        res = eng.check_results()
        assert (res["valid"] == 1).all() and (res["duplicated_count"] == 0).all()


@pytest.mark.parametrize("service,kw", [
    ("lin-kv", dict()),
    ("lin-kv", dict(latency=20, latency_dist="exponential", p_loss=0.05)),
    ("lin-kv", dict(nemesis=["partition"], nemesis_interval=4, journal_capacity=60000)),
    ("seq-kv", dict(rate=100)),
    ("seq-kv", dict(latency=30, latency_dist="uniform", node_count=3)),
    ("lww-kv", dict(rate=100)),
    ("lww-kv", dict(node_count=7, concurrency=28, latency=10)),
    ("lin-kv", dict(concurrency=10, rate=30, time_limit=20)),    # the reference's demo invocation (core.clj:112): a full 16-lane group
    ("lin-kv", dict(concurrency=10, rate=400, latency=40, latency_dist="exponential", p_loss=0.1, nemesis=["partition"], nemesis_interval=3)),   # queues beyond their LDS slots, timeouts, late replies
    ("lww-kv", dict(concurrency=10, rate=200, latency=20, latency_dist="uniform", p_loss=0.02)),
    ("lin-kv", dict(node_count=1, concurrency=4, rate=100)),
    ("lww-kv", dict(node_count=3, concurrency=12, rate=100, inbox_capacity=2, spill_capacity=3)),   # 16 lanes; inbox overflow flagged alike
])
def test_lin_kv_proxy_and_services_parity(lib, service, kw):
    """demo/ruby/lin_kv_proxy.rb over lin-kv / seq-kv / lww-kv (service.clj): one cluster per wavefront (svc_kernel<>) and four (svc4_kernel<>,
    csrc/svc4.hip: lin-kv / lww-kv, clusters of <= 15 endpoints + the service, journal off; large launches take it unasked, here MSIM_DEV_FLAGS
    bit 10 asks for it; 6 clusters = a full wavefront of four and a partial one)."""
    base = dict(node_count=5, rate=60, time_limit=12, latency=5, seed=91)
    base.update(kw)
    cfg = E.test_config("lin-kv", bin="lin-kv-proxy", proxy_service=service, **base)
    _compare(cfg, 0, 6)
    _compare(cfg, 3, 6, dev_flags=0x400)


def test_raft_runs_match_the_recorded_reference_replays(lib):
    """The engine against the reference's own code, without oracle or reference tree at hand: for these runs every message was
    reproduced by demo/python/raft.py driven with the same schedule (tests/test_raft_reference_replay.py), and a digest of all
    :send events was recorded (tests/golden/raft_replay_digests.json).  The engine's journal must hash to the same digests."""
    import hashlib
    import json
    import os
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "raft_replay_digests.json")))
    for case in gold.values():
        base = dict(bin="raft", node_count=5, rate=30, time_limit=20, seed=57, journal_capacity=600000)
        base.update(case["options"])
        cfg = E.test_config("lin-kv", **base)
        with E.Engine(cfg) as eng:
            eng.run(0, len(case["digests"]))
            eng.fetch()
            for i, want in enumerate(case["digests"]):
                ev = eng.raw_journal(i)
                assert eng.meta(i).flags == 0
                sends = ev[((ev["msg"] >> 7) & 1) == 0]
                h = hashlib.sha256(np.stack([sends["time_us"], sends["msg"], sends["a"], sends["route"]], axis=1).astype(np.uint32).tobytes())
                assert h.hexdigest() == want


@pytest.mark.parametrize("kw", [
    dict(node_count=3, rate=100, time_limit=3, latency=5),
    dict(node_count=5, concurrency=10, rate=300, time_limit=3, latency=10, latency_dist="exponential"),
    dict(node_count=3, rate=200, time_limit=6, latency=3, nemesis=["partition"], nemesis_interval=2, journal_capacity=200000),
    dict(node_count=4, concurrency=12, rate=500, time_limit=4, latency=20, latency_dist="uniform", p_loss=0.05),
    dict(node_count=1, rate=50, time_limit=2),
    dict(node_count=5, concurrency=10, rate=1000, time_limit=3, latency=8, latency_dist="exponential", p_loss=0.05, nemesis=["partition"], nemesis_interval=1),   # a full 16-lane group
])
def test_unique_ids_over_lin_tso_parity(lib, kw):
    """unique-ids served through the lin-tso timestamp oracle (service.clj:116-132): svc_kernel<.., TSO> (one cluster per wavefront) and
    svc4_kernel<.., TSO> (four: nodes + workers + the service <= 16, journal off; MSIM_DEV_FLAGS bit 10 asks for it here) against
    oracle/svc_nodes.inc, which the process bridge pins with a real node process (tests/test_process_bridge.py)."""
    cfg = E.test_config("unique-ids", bin="tso-ids", seed=17, **kw)
    ora = _compare(cfg, 0, 6)
    _compare(cfg, 0, 6, dev_flags=0x400)
    for i in range(6):
        ops = E.decode_history(*ora.history(i), cfg.n_nodes, cfg.workload, cfg.node_program)
        ids = sorted(o["value"] for o in ops if o["type"] == ":ok")
        assert len(set(ids)) == len(ids) > 20


def test_unique_ids_over_lin_tso_pass_the_device_checker(lib):
    cfg = E.test_config("unique-ids", bin="tso-ids", node_count=3, rate=300, time_limit=5, latency=5, nemesis=["partition"], nemesis_interval=2, seed=18)
    with E.Engine(cfg) as eng:
        eng.run(0, 32)
        eng.check()
        res = eng.check_results()
        assert (res["valid"] == 1).all() and (res["duplicated_count"] == 0).all() and (res["ok_count"] > 100).all()
