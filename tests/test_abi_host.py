"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/maelsim.h
declares, validates options like the reference's CLI does, and fails loudly without a GPU."""
import ctypes as C
import os
import re

import pytest

from maelstrom_amd import _abi as A
from maelstrom_amd import engine as E

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "maelsim.h")).read()
    declared = set(re.findall(r"^(?:int|uint32_t|void|const char \*)\s*(msim_[a-z0-9_]+)\(", hdr, re.M))
    assert declared == set(A.EXPORTS), declared ^ set(A.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.msim_abi_version() == A.ABI_VERSION


def test_struct_layouts_match_header():
    assert C.sizeof(A.Config) == 120 and A.Config.seed.offset == 64 and A.Config.max_values.offset == 72
    assert C.sizeof(A.Op) == 16 and C.sizeof(A.NetStats) == 48 and C.sizeof(A.InstMeta) == 32 and C.sizeof(A.Event) == 16
    assert C.sizeof(A.CheckResult) == 68 and C.sizeof(A.DeviceBuffers) == 112 and A.Config.journal_capacity.offset == 92 and A.Config.key_count.offset == 96


def test_defaults_mirror_reference_cli(lib):
    cfg = A.Config()
    assert lib.msim_config_defaults(C.byref(cfg), A.WL_BROADCAST, 5) == 0
    # core.clj:136-229 + jepsen.cli: rate 5/s, latency 0 constant, grid, nemesis interval 10 s, concurrency 1n, 60 s
    assert (cfg.rate_mhz, cfg.latency_mean_ms, cfg.latency_dist, cfg.topology) == (5000, 0, A.LAT_CONSTANT, A.TOPO_GRID)
    assert (cfg.nemesis_mask, cfg.nemesis_interval_ms, cfg.concurrency, cfg.time_limit_ms) == (0, 10000, 5, 60000)
    assert (cfg.client_timeout_ms, cfg.quiesce_ms, cfg.p_loss_q32) == (5000, 10000, 0)  # client.clj:18-20, core.clj:78, net.clj:100


def test_finalize_derives_capacities_and_rejects_bad_options(lib):
    cfg = E.test_config("broadcast", node_count=25, rate=100, time_limit=20)
    assert cfg.max_values % 32 == 0 and cfg.max_values >= 1100 and cfg.max_rows >= 2 * 2300 and cfg.inbox_capacity >= 8
    with pytest.raises(E.EngineError, match="divides by zero"):   # net.clj:77 with --latency 0
        E.test_config("broadcast", node_count=5, latency=0, latency_dist="exponential")
    with pytest.raises(E.EngineError, match="2000 s of virtual time"):   # 2 x time must fit in u32 microseconds
        E.test_config("echo", node_count=1, time_limit=2100)
    with pytest.raises(E.EngineError, match="node_program"):
        E.test_config("echo", bin="g-set", node_count=3)
    with pytest.raises(E.EngineError, match="n_nodes"):
        E.test_config("broadcast", node_count=0)
    cfg = E.test_config("txn-list-append", node_count=5)
    assert (cfg.node_program, cfg.key_count, cfg.max_txn_length, cfg.max_writes_per_key) == (A.NODE_TXN_SINGLE_KEY, 10, 4, 16)  # core.clj:191-199
    assert E.test_config("txn-list-append", node_count=5, concurrency=10).concurrency == 10   # several workers per node: the single-root and the Datomic-style node (round 6)
    with pytest.raises(E.EngineError, match="one worker per node"):
        E.test_config("txn-list-append", node_count=5, concurrency=7)                          # ... in multiples of the node count
    assert E.test_config("txn-list-append", bin="multi-key-txn", node_count=5, concurrency=10).concurrency == 10
    assert E.test_config("kafka", node_count=5, concurrency=10).concurrency == 10
    with pytest.raises(E.EngineError, match="one worker per node"):
        E.test_config("kafka", node_count=5, concurrency=12)
    with pytest.raises(E.EngineError, match="max-txn-length"):
        E.test_config("txn-list-append", node_count=5, max_txn_length=9)
    with pytest.raises(E.EngineError, match="multiple of 2 x node-count"):
        E.test_config("lin-kv", bin="raft", node_count=5, concurrency=5)
    cfg = E.test_config("lin-kv", bin="raft", node_count=5)
    assert cfg.concurrency == 10
    assert cfg.spill_capacity >= 256
    with pytest.raises(KeyError):
        E.test_config("broadcast", topology="hypercube")


def test_engine_fails_loudly_without_a_gpu(lib):
    if lib.msim_device_count() > 0:
        pytest.skip("a HIP device is visible")
    with pytest.raises(E.EngineError, match="no HIP device"):
        E.Engine(E.test_config("echo", node_count=3))


def test_batch_checkers_validate_their_arguments_before_touching_a_device(lib):
    """msim_check_kafka_batch: payload words announced but no payload given is MSIM_E_INVALID (like msim_check_kafka_rows), not a kernel
    reading uninitialised memory."""
    import numpy as np
    rows = np.zeros(2, dtype=E.OP_DT)
    ro = np.array([0, 2], dtype=np.uint64)
    po = np.array([0, 5], dtype=np.uint64)
    out = np.zeros(1, dtype=E.CHECK_DT)
    nh = C.c_uint32(0)
    rc = lib.msim_check_kafka_batch(0, rows.ctypes.data_as(C.c_void_p), ro.ctypes.data_as(C.c_void_p), None, po.ctypes.data_as(C.c_void_p), 1, 3, out.ctypes.data_as(C.c_void_p), C.byref(nh))
    assert rc == A.E_INVALID


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under maelstrom_amd/ or include/ may import, include, load or link it."""
    bad = re.compile(r"^\s*(?:import|from)\s+\S*oracle|#\s*include\s+\S*oracle|oracle_lib|libmaelsim_oracle|oracle_run|CDLL\([^)]*oracle", re.M)
    for base in ("maelstrom_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".cpp", ".h")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    assert not bad.search(txt), f


def test_history_decoder_roundtrip():
    import numpy as np
    rows = np.zeros(4, dtype=E.OP_DT)
    rows["time_len"] = [1000, 2000 | (1 << 48), 3000, 4000]
    rows["packed"] = [A.T_INVOKE | (A.F_READ << 2) | (3 << 12), A.T_OK | (A.F_READ << 2) | (1 << 11) | (3 << 12),
                      A.T_INVOKE | (A.F_BROADCAST << 2) | (7 << 12), A.T_INFO | (A.F_BROADCAST << 2) | (A.ERR_NET_TIMEOUT << 7) | (7 << 12)]
    rows["value"] = [A.NO_VALUE, 0, 5, 5]
    payload = np.array([0b101001], dtype=np.uint32)
    h = E.decode_history(rows, payload, 5)
    assert h[1] == {"index": 1, "time": 2000, "type": ":ok", "f": ":read", "process": 3, "value": [0, 3, 5], "final?": True}
    assert h[3]["error"] == ":net-timeout" and h[3]["type"] == ":info" and h[2]["value"] == 5
    st = {"all_send": 10, "all_recv": 9, "clients_send": 4, "clients_recv": 4, "servers_send": 6, "servers_recv": 5}
    m = E.net_stats_map(st, rows)
    assert m["all"] == {"send-count": 10, "recv-count": 9, "msg-count": 10, "msgs-per-op": 5.0} and m["servers"]["msgs-per-op"] == 3.0
