"""The lin-kv linearizability search on the device (csrc/lin_check_dev.hip) where it is hard: histories of a Raft cluster under
partitions, whose timed-out writes and cas stay pending for the rest of their key — more than 64 configurations while one call returns,
the keys pass 1's registers give up on.  Passes 2 and 3 (a workgroup per history, the configurations in an LDS hash table) and the host
search (csrc/lin_check.cpp) for what is left must report, field by field, what the host search alone reports — on the valid histories
and on copies with one read changed (mostly not linearizable).  With MSIM_DEV_FLAGS bit 13 the pools are small enough that every level
is reached, the host included.  tests/linearizable_ref.py (an independent search without the dominance / symmetry pruning) confirms
the verdict on the changed keys.  tests/test_hipemu_parity.py runs this file on the host wavefront emulator in the CPU suite."""
import ctypes as C

import numpy as np
import pytest

from maelstrom_amd import _abi as A
from maelstrom_amd import engine as E
import oracle_lib as O

pytestmark = pytest.mark.gpu
FIELDS = ("valid", "attempt_count", "error_count", "op_count", "ok_count", "fail_count", "info_count", "stable_count", "lost_count", "stale_count",
          "never_read_count", "duplicated_count")


def _host(rows):
    res = A.CheckResult()
    rows = np.ascontiguousarray(rows)
    assert A.load().msim_check_lin_kv_rows(rows.ctypes.data_as(C.c_void_p), len(rows), C.byref(res)) == 0
    return res


def _histories(n, time_limit=40, payloads=False):
    cfg = E.test_config("lin-kv", bin="raft", node_count=5, rate=30, time_limit=time_limit, latency=10, nemesis=["partition"], nemesis_interval=10, seed=99)
    ora = O.run(cfg, 0, n)
    assert (ora.meta["flags"] == 0).all()
    if payloads:
        return [(ora.history(i)[0].copy(), ora.history(i)[1].copy()) for i in range(n)]
    return [ora.history(i)[0].copy() for i in range(n)]


def _with_one_read_changed(rows, rng):
    """a copy with the value of one :ok read (of the busier second half, where the partitions are) replaced"""
    rows = rows.copy()
    packed = rows["packed"]
    idx = np.nonzero(((packed & 3) == A.T_OK) & (((packed >> 2) & 31) == A.F_READ))[0]
    idx = idx[idx > len(rows) // 3]
    i = int(idx[rng.integers(len(idx))])
    v = int(rows["value"][i])
    old = (v >> 8) & 0xFF
    new = (old + 1 + int(rng.integers(3))) % 5 if old != 0xFF else int(rng.integers(5))
    rows["value"][i] = (v & ~0xFF00) | (new << 8)
    return rows


@pytest.mark.parametrize("flags", [0, 0x2000])
def test_wide_keys_device_equals_host(lib, flags, monkeypatch):
    rng = np.random.default_rng(5)
    good = _histories(40)
    hs = good + [_with_one_read_changed(h, rng) for h in good]
    if flags:
        monkeypatch.setenv("MSIM_DEV_FLAGS", hex(flags))
    dev = E.check_lin_kv_batch(hs)
    n_invalid = 0
    for i, rows in enumerate(hs):
        h = _host(rows)
        for f in FIELDS:
            assert int(dev[i][f]) == int(getattr(h, f)), (i, f, int(dev[i][f]), int(getattr(h, f)))
        n_invalid += int(dev[i]["valid"]) == 0
    assert all(int(dev[i]["valid"]) == 1 for i in range(len(good)))
    assert n_invalid >= len(good) // 2   # a changed read is almost always a register value nobody wrote in time


def test_changed_reads_agree_with_the_independent_search(lib):
    import linearizable_ref as L
    rng = np.random.default_rng(6)
    good = _histories(6, time_limit=20, payloads=True)
    hs = [_with_one_read_changed(h, rng) for h, _ in good]
    dev = E.check_lin_kv_batch(hs)
    for rows, (_, pay), d in zip(hs, good, dev):
        ref = L.check(E.decode_history(rows, pay, 5, A.WL_LIN_KV))
        assert int(d["attempt_count"]) == len(ref) and int(d["error_count"]) == sum(1 for ok in ref.values() if not ok)


def _synthetic(seed, n_ops, keys, workers, p_overlap, p_wrong, p_fail, p_info, odd=0.0):
    """A lin-kv history built by hand: mostly operations that complete before the next begins (the runs the device's pair paths of
    round 6 take in one step: pair_rows / search_key in csrc/lin_check_dev.hip), with — at the given rates — operations that overlap,
    results no register ever held, :fail and :info completions, and (`odd`) rows no client produces: a completion without an invocation,
    a process that invokes twice."""
    import random
    rnd = random.Random(seed)
    ops, open_ops = [], []     # open: (process, key, f, v1, v2)
    reg = {}
    t = 0
    free = list(range(workers))
    done = 0
    while done < n_ops or open_ops:
        t += rnd.randrange(1, 2_000_000)
        start = done < n_ops and free and (not open_ops or rnd.random() < p_overlap)
        if start:
            p = free.pop(rnd.randrange(len(free)))
            k = rnd.choice(keys)
            f = rnd.choice([":read", ":write", ":cas"])
            v1, v2 = rnd.randrange(5), rnd.randrange(5)
            val = [k, None] if f == ":read" else ([k, v1] if f == ":write" else [k, [v1, v2]])
            ops.append({"type": ":invoke", "f": f, "process": p, "value": val, "time": t})
            open_ops.append((p, k, f, v1, v2)); done += 1
            if rnd.random() < odd:   # the same process again, before its completion
                ops.append({"type": ":invoke", "f": f, "process": p, "value": val, "time": t + 1})
            continue
        if not open_ops:
            continue
        p, k, f, v1, v2 = open_ops.pop(rnd.randrange(len(open_ops)))
        x = rnd.random()
        cur = reg.get(k)
        if x < p_fail or (f == ":cas" and cur != v1 and rnd.random() >= p_wrong):
            typ = ":fail"
        elif x < p_fail + p_info:
            typ = ":info"
        else:
            typ = ":ok"
        if typ == ":info" and f != ":read" and rnd.random() < 0.5:   # an indeterminate write / cas may have happened
            if f == ":write": reg[k] = v1
            elif cur == v1: reg[k] = v2
        if typ == ":ok":
            if f == ":write": reg[k] = v1
            elif f == ":cas": reg[k] = v2
            else:
                v1 = cur if rnd.random() >= p_wrong else rnd.randrange(5)
        val = [k, v1] if f != ":cas" else [k, [v1, v2]]
        ops.append({"type": typ, "f": f, "process": p, "value": val, "time": t})
        if rnd.random() < odd:   # a completion nobody invoked
            ops.append({"type": ":ok", "f": ":read", "process": 90 + rnd.randrange(5), "value": [k, rnd.randrange(5)], "time": t + 1})
        free.append(p)
    return ops


def test_synthetic_histories_device_equals_host(lib):
    """Sequential runs (the device's pair paths), overlaps, wrong results, :fail / :info, several keys interleaved, rows no client produces:
    field by field what the host search reports."""
    import random
    rnd = random.Random(77)
    hs = []
    for i in range(60):
        hs.append(_synthetic(1000 + i, n_ops=rnd.randrange(20, 400), keys=list(range(rnd.choice([1, 1, 2, 5]))), workers=rnd.choice([1, 2, 4, 10]),
                             p_overlap=rnd.choice([0.0, 0.05, 0.3]), p_wrong=rnd.choice([0.0, 0.0, 0.01, 0.1]), p_fail=rnd.choice([0.0, 0.1]),
                             p_info=rnd.choice([0.0, 0.0, 0.02]), odd=rnd.choice([0.0, 0.0, 0.02])))
    rows = [E.encode_lin_kv_history(h) for h in hs]
    dev = E.check_lin_kv_batch(rows)
    verdicts = set()
    for i, r in enumerate(rows):
        h = _host(r)
        for f in FIELDS:
            assert int(dev[i][f]) == int(getattr(h, f)), (i, f, int(dev[i][f]), int(getattr(h, f)))
        verdicts.add(int(dev[i]["valid"]))
    assert {0, 1} <= verdicts
