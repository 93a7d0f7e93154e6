"""txn-list-append's analysis on the device (csrc/txn_check_dev.hip, behind msim_check) against the host analysis (csrc/txn_check.cpp,
msim_check_txn_rows — itself held against tests/elle_ref.py and hand-made anomalies in tests/test_txn_list_append.py): every field of
every history's result, on engine histories (partitions, loss, timeouts), on corrupted ones and on the hand-made anomalous ones."""
import copy
import ctypes as C
import random

import numpy as np
import pytest

from maelstrom_amd import _abi as A
from maelstrom_amd import engine as E
import test_txn_list_append as T

pytestmark = pytest.mark.gpu

FIELDS = ("valid", "attempt_count", "stable_count", "lost_count", "stale_count", "error_count", "op_count", "ok_count", "fail_count", "info_count")


def _host(rows, pay):
    res = A.CheckResult()
    rows = np.ascontiguousarray(rows); pay = np.ascontiguousarray(pay, dtype=np.uint32)
    assert A.load().msim_check_txn_rows(rows.ctypes.data_as(C.c_void_p), len(rows), pay.ctypes.data_as(C.c_void_p), len(pay), C.byref(res)) == 0
    return res


def _same(dev, host, ctx):
    for f in FIELDS:
        assert int(dev[f]) == int(getattr(host, f)), (ctx, f, int(dev[f]), int(getattr(host, f)))


@pytest.mark.parametrize("kw", [
    dict(node_count=5, rate=100, time_limit=20, latency=5),
    dict(node_count=5, rate=100, time_limit=30, latency=5, nemesis=["partition"], nemesis_interval=10),          # BASELINE configs[4]
    dict(node_count=3, rate=200, time_limit=15, latency=20, latency_dist="exponential", p_loss=0.05),            # timeouts: :info transactions
    dict(node_count=5, rate=60, time_limit=20, latency=10, key_count=3, max_txn_length=6, nemesis=["partition"], nemesis_interval=4),
])
@pytest.mark.parametrize("flags", [0, 0x2000])   # the LDS kernel, a workgroup per history (default) / the HBM-table kernel
def test_device_pass_equals_host_analysis_on_engine_histories(lib, kw, flags):
    cfg = E.test_config("txn-list-append", seed=23, **kw)
    n = 64
    with E.Engine(cfg) as eng:
        if flags:
            eng.set_dev_flags(flags)
        eng.run(0, n)
        eng.check()
        res = eng.check_results()
        handed = eng.check_host_rechecks()
        eng.fetch()
        for i in range(n):
            rows, pay = eng.raw_history(i)
            if eng.meta(i).flags:
                assert int(res[i]["valid"]) == 0
                continue
            _same(res[i], _host(rows, pay), (kw, i))
        assert (res["valid"] == 1).all()
        assert handed == 0, handed   # strict-serializable histories are clean: the device decides all of them


def test_hand_made_anomalies_and_corrupted_histories(lib):
    A_, R_ = T.A_, T.R_
    cases = [
        [(":invoke", 0, [[A_, 1, 1], [R_, 1, None]]), (":ok", 0, [[A_, 1, 1], [R_, 1, [1]]]), (":invoke", 1, [[R_, 1, None], [A_, 1, 2]]), (":ok", 1, [[R_, 1, [1]], [A_, 1, 2]]),
         (":invoke", 0, [[R_, 1, None]]), (":ok", 0, [[R_, 1, [1, 2]]])],
        [(":invoke", 0, [[A_, 1, 1], [A_, 2, 1]]), (":invoke", 1, [[A_, 1, 2], [A_, 2, 2]]), (":ok", 0, [[A_, 1, 1], [A_, 2, 1]]), (":ok", 1, [[A_, 1, 2], [A_, 2, 2]]),
         (":invoke", 2, [[R_, 1, None], [R_, 2, None]]), (":ok", 2, [[R_, 1, [1, 2]], [R_, 2, [2, 1]]])],                                                     # G0
        [(":invoke", 0, [[A_, 1, 1]]), (":fail", 0, [[A_, 1, 1]]), (":invoke", 1, [[R_, 1, None]]), (":ok", 1, [[R_, 1, [1]]])],                               # G1a
        [(":invoke", 0, [[A_, 1, 1], [A_, 1, 2]]), (":invoke", 1, [[R_, 1, None]]), (":ok", 1, [[R_, 1, [1]]]), (":ok", 0, [[A_, 1, 1], [A_, 1, 2]])],          # G1b
        [(":invoke", 0, [[A_, 1, 1], [R_, 2, None]]), (":invoke", 1, [[A_, 2, 1], [R_, 1, None]]), (":ok", 0, [[A_, 1, 1], [R_, 2, [1]]]), (":ok", 1, [[A_, 2, 1], [R_, 1, [1]]])],   # G1c
        [(":invoke", 0, [[A_, 1, 1], [A_, 2, 1]]), (":invoke", 1, [[R_, 1, None], [R_, 2, None]]), (":ok", 0, [[A_, 1, 1], [A_, 2, 1]]), (":ok", 1, [[R_, 1, [1]], [R_, 2, None]]),
         (":invoke", 2, [[R_, 2, None]]), (":ok", 2, [[R_, 2, [1]]])],                                                                                          # G-single
        [(":invoke", 0, [[R_, 1, None], [A_, 2, 1]]), (":invoke", 1, [[R_, 2, None], [A_, 1, 1]]), (":ok", 0, [[R_, 1, None], [A_, 2, 1]]), (":ok", 1, [[R_, 2, None], [A_, 1, 1]]),
         (":invoke", 2, [[R_, 1, None], [R_, 2, None]]), (":ok", 2, [[R_, 1, [1]], [R_, 2, [1]]])],                                                             # G2
        [(":invoke", 0, [[A_, 1, 1], [R_, 1, None]]), (":ok", 0, [[A_, 1, 1], [R_, 1, None]])],                                                                  # internal
        [(":invoke", 0, [[R_, 1, None], [A_, 1, 1], [R_, 1, None]]), (":ok", 0, [[R_, 1, None], [A_, 1, 1], [R_, 1, [1]]])],
        [(":invoke", 0, [[A_, 1, 1]]), (":ok", 0, [[A_, 1, 1]]), (":invoke", 1, [[A_, 1, 2]]), (":ok", 1, [[A_, 1, 2]]), (":invoke", 2, [[R_, 1, None]]), (":ok", 2, [[R_, 1, [1, 2]]]),
         (":invoke", 3, [[R_, 1, None]]), (":ok", 3, [[R_, 1, [2, 1]]])],                                                                                       # incompatible order
        [(":invoke", 0, [[A_, 1, 1]]), (":ok", 0, [[A_, 1, 1]]), (":invoke", 1, [[R_, 1, None]]), (":ok", 1, [[R_, 1, [1, 1]]])],                                # duplicates
        [(":invoke", 0, [[A_, 1, 1]]), (":ok", 0, [[A_, 1, 1]]), (":invoke", 1, [[R_, 1, None]]), (":ok", 1, [[R_, 1, None]]), (":invoke", 2, [[R_, 1, None]]), (":ok", 2, [[R_, 1, [1]]])],   # realtime
        [(":invoke", 0, [[A_, 1, 1]]), (":invoke", 1, [[R_, 1, None]]), (":ok", 0, [[A_, 1, 1]]), (":ok", 1, [[R_, 1, None]]), (":invoke", 2, [[R_, 1, None]]), (":ok", 2, [[R_, 1, [1]]])],
        [(":invoke", 0, [[A_, 1, 1]]), (":info", 0, [[A_, 1, 1]]), (":invoke", 1, [[R_, 1, None]]), (":ok", 1, [[R_, 1, [1]]])],                               # indeterminate append observed
        [(":invoke", 0, [[A_, 1, 1]]), (":invoke", 0, [[A_, 1, 2]]), (":ok", 0, [[A_, 1, 2]]), (":ok", 0, [[A_, 1, 9]])],                                       # pairing: double invoke, stray completion
        [],
    ]
    hs = [T._h(*ops) for ops in cases]
    # engine histories corrupted the five ways of tests/test_txn_list_append.py
    cfg = E.test_config("txn-list-append", node_count=5, rate=60, time_limit=8, latency=5, key_count=3, seed=5)
    rng = random.Random(7)
    with E.Engine(cfg) as eng:
        eng.run(0, 3)
        eng.fetch()
        for i in range(3):
            rows, pay = eng.raw_history(i)
            hs.append((rows.copy(), pay.copy()))
            ops = [o for o in E.decode_history(rows, pay, cfg.n_nodes, A.WL_TXN_LIST_APPEND) if o["process"] != ":nemesis"]
            reads = [(oi, mi) for oi, o in enumerate(ops) if o["type"] == ":ok" for mi, m in enumerate(o["value"]) if m[0] == ":r" and m[2] and len(m[2]) >= 2]
            for trial in range(25):
                oi, mi = rng.choice(reads)
                mut = copy.deepcopy(ops)
                lst = mut[oi]["value"][mi][2]
                how = trial % 5
                if how == 0:
                    lst.pop()
                elif how == 1:
                    lst[0], lst[-1] = lst[-1], lst[0]
                elif how == 2:
                    lst.append(lst[0])
                elif how == 3:
                    lst.append(61)
                else:
                    mut[oi]["value"][mi][2] = None
                hs.append(E.encode_txn_history(mut))
    dev = E.check_txn_batch(hs)
    n_bad = 0
    for i, (rows, pay) in enumerate(hs):
        _same(dev[i], _host(rows, pay), i)
        n_bad += int(dev[i]["valid"]) == 0
    assert [int(v) for v in dev["valid"][:3]] == [1, 0, 0]
    assert n_bad > 40
