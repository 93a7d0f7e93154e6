"""msim_history_edn_rows (csrc/edn.cpp) writes the same history.edn text as the Python mirror's decoder + printer, for
every workload's :value shape, and that text has the shape of the reference's own sample
(doc/05-datomic/02-shared-state.md:384-386)."""
import ctypes as C
import re

import pytest

from maelstrom_amd import _abi as A
from maelstrom_amd import engine as E
import oracle_lib as O

CASES = [
    ("echo", dict(node_count=2, rate=30, time_limit=4, p_loss=0.2)),
    ("broadcast", dict(node_count=5, rate=30, time_limit=6, latency=20, p_loss=0.1, nemesis=["partition"], nemesis_interval=1)),
    ("g-set", dict(node_count=5, rate=30, time_limit=6, latency=20)),
    ("lin-kv", dict(bin="raft", node_count=5, rate=30, time_limit=15, latency=10, nemesis=["partition"], nemesis_interval=3)),
    ("lin-kv", dict(bin="lin-kv-proxy", proxy_service="lww-kv", node_count=3, rate=40, time_limit=6)),
    ("txn-list-append", dict(node_count=5, rate=60, time_limit=6, latency=5, p_loss=0.05)),
    ("txn-rw-register", dict(node_count=2, rate=60, time_limit=6, nemesis=["partition"], nemesis_interval=2)),
    ("pn-counter", dict(node_count=3, rate=30, time_limit=8, latency=50, p_loss=0.1)),
    ("g-counter", dict(node_count=3, rate=30, time_limit=8)),
    ("unique-ids", dict(node_count=3, rate=60, time_limit=4, p_loss=0.1)),
]


@pytest.mark.parametrize("workload,kw", CASES)
def test_native_writer_equals_python_printer(workload, kw):
    cfg = E.test_config(workload, seed=33, **kw)
    r = O.run(cfg, 0, 2)
    for i in range(2):
        rows, pay = r.history(i)
        want = E.history_edn(E.decode_history(rows, pay, cfg.n_nodes, cfg.workload))
        got = E.history_edn_native(cfg, rows, pay)
        assert got == want
        assert got.count("\n") == len(rows)
        # the reference's sample line: {:type :invoke, :f :txn, :value [[:append 9 1]], :time 2883850541, :process 0, :index 0}
        for line in got.splitlines()[:50]:
            assert re.fullmatch(r"\{:type :(invoke|ok|fail|info), :f :[a-z-]+, :value .+, :time \d+, :process (\d+|:nemesis), :index \d+(, :error .+)?(, :final\? true)?\}", line), line


def test_size_query_and_small_buffer():
    cfg = E.test_config("broadcast", node_count=3, rate=10, time_limit=2, seed=1)
    r = O.run(cfg, 0, 1)
    rows, pay = r.history(0)
    lib = A.load()
    need = C.c_size_t()
    args = (C.byref(cfg), rows.ctypes.data_as(C.c_void_p), len(rows), pay.ctypes.data_as(C.c_void_p), len(pay))
    assert lib.msim_history_edn_rows(*args, None, 0, C.byref(need)) == 0 and need.value > 100
    small = C.create_string_buffer(16)
    assert lib.msim_history_edn_rows(*args, small, 16, None) == A.E_RANGE
    bad = rows.copy()
    bad["value"][:] = 0xFFFFFF0
    bad["time_len"][:] |= 5 << 48      # payload references outside the area are refused, not read
    assert lib.msim_history_edn_rows(C.byref(cfg), bad.ctypes.data_as(C.c_void_p), len(bad), pay.ctypes.data_as(C.c_void_p), len(pay), None, 0, C.byref(need)) == A.E_RANGE
