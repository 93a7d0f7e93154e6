"""msim_check_set_full_batch: the device checker of broadcast / g-set / echo over histories handed in by the caller — the oracle's
histories and mutations of them (reads that lose elements, completions turned into :fail / :info, echoes answered wrongly) —
against the pure-Python restatement of jepsen's set-full (tests/setfull_ref.py; parity unpinned: [upstream] jepsen is not vendored)."""
import numpy as np
import pytest

from maelstrom_amd import engine as E, _abi as A
import oracle_lib as O
import setfull_ref as R

pytestmark = pytest.mark.gpu

VALID = {1: True, 0: False, 2: "unknown"}
KEYS = (("attempt_count", "attempt-count"), ("stable_count", "stable-count"), ("lost_count", "lost-count"),
        ("never_read_count", "never-read-count"), ("stale_count", "stale-count"))


def _compare(cfg, hs, wl, res):
    for i, (rows, pay) in enumerate(hs):
        h = E.decode_history(rows, pay, cfg.n_nodes, wl)
        g = res[i]
        ref = R.set_full(h)
        assert VALID[int(g["valid"])] == ref["valid?"], (i, g, ref)
        for k, rk in KEYS:
            assert int(g[k]) == ref[rk], (i, k, int(g[k]), ref[rk])
        if ref["stable-latencies"]:
            assert [int(x) for x in g["stable_latency_ms"]] == [ref["stable-latencies"][q] for q in (0, 0.5, 0.95, 0.99, 1)], (i, g, ref)
        for k, t in (("op_count", ":invoke"), ("ok_count", ":ok"), ("fail_count", ":fail"), ("info_count", ":info")):
            assert int(g[k]) == sum(1 for op in h if op["type"] == t and op["process"] != ":nemesis"), (i, k)


def _mutate(rows, pay, rng, drop_bits=0.02, fail=0.03):
    """Reads lose elements (a lost / stale element for set-full), some read completions become :fail (their invocation pairs with
    nothing), some add completions become :info."""
    rows, pay = rows.copy(), pay.copy()
    typ, f = rows["packed"] & 3, (rows["packed"] >> 2) & 31
    for i in np.nonzero((f == A.F_READ) & (typ == A.T_OK))[0]:
        off, n = int(rows["value"][i]), int(rows["time_len"][i] >> 48)
        for w in range(n):
            if rng.random() < drop_bits * 8:
                pay[off + w] &= ~np.uint32(1 << int(rng.integers(32)))
        if rng.random() < fail:
            rows["packed"][i] = (rows["packed"][i] & ~np.uint32(3)) | A.T_FAIL
    for i in np.nonzero(((f == A.F_ADD) | (f == A.F_BROADCAST)) & (typ == A.T_OK))[0]:
        if rng.random() < fail:
            rows["packed"][i] = (rows["packed"][i] & ~np.uint32(3)) | A.T_INFO
    return rows, pay


CASES = [
    ("broadcast", dict(node_count=5, rate=20, time_limit=5, latency=50, seed=4), 4),
    ("broadcast", dict(node_count=25, rate=50, time_limit=5, latency=100, latency_dist="exponential", seed=8), 2),
    ("broadcast", dict(bin="broadcast-ff", node_count=5, rate=20, time_limit=20, nemesis=["partition"], nemesis_interval=3, latency=10, seed=6), 6),
    ("broadcast", dict(node_count=5, concurrency=10, rate=40, time_limit=8, latency=200, seed=5), 3),
    ("broadcast", dict(node_count=3, concurrency=1 * 3, rate=200, time_limit=3, latency=0, seed=9), 3),          # many rows of one thread per chunk
    ("g-set", dict(node_count=7, rate=30, time_limit=10, latency=30, latency_dist="exponential", p_loss=0.5, seed=3), 3),
    ("g-set", dict(node_count=40, concurrency=80, rate=60, time_limit=5, latency=20, seed=11), 2),               # more than 64 worker threads
]


@pytest.mark.parametrize("wl,kw,n", CASES)
def test_set_full_batch_matches_restatement(lib, wl, kw, n):
    cfg = E.test_config(wl, **kw)
    code = A.WL_BROADCAST if wl == "broadcast" else A.WL_G_SET
    o = O.run(cfg, 0, n)
    hs = [o.history(i) for i in range(n)]
    _compare(cfg, hs, code, E.check_set_full_batch(hs, cfg.concurrency, code, max_values=cfg.max_values))
    rng = np.random.default_rng(7)
    ms = [_mutate(r, p, rng) for r, p in hs]
    res = E.check_set_full_batch(ms, cfg.concurrency, code, max_values=cfg.max_values)
    _compare(cfg, ms, code, res)


def test_set_full_batch_equals_engine_check(lib):
    """The same kernel behind Engine.check() and behind the batch entry."""
    cfg = E.test_config("broadcast", node_count=25, rate=100, time_limit=4, latency=10, seed=2)
    with E.Engine(cfg) as eng:
        eng.run(0, 6)
        eng.check()
        eng.fetch()
        a = eng.check_results().copy()
        hs = [tuple(x.copy() for x in eng.raw_history(i)) for i in range(6)]   # (views of the engine's buffers: copied before it closes)
    b = E.check_set_full_batch(hs, cfg.concurrency, A.WL_BROADCAST, max_values=cfg.max_values)
    assert a.tobytes() == b.tobytes()


def test_echo_batch_counts_wrong_and_missing_replies(lib):
    cfg = E.test_config("echo", node_count=3, rate=20, time_limit=5, seed=1)
    o = O.run(cfg, 0, 3)
    hs = [o.history(i) for i in range(3)]
    res = E.check_set_full_batch(hs, cfg.concurrency, A.WL_ECHO, max_values=32)
    assert (res["valid"] == 1).all() and (res["error_count"] == 0).all()
    rng = np.random.default_rng(3)
    want = []
    ms = []
    for rows, pay in hs:
        rows = rows.copy()
        typ = rows["packed"] & 3
        oks = np.nonzero(typ == A.T_OK)[0]
        bad = np.append(rng.choice(oks[:-1], size=4, replace=False), oks[-1])   # (the last :ok: its worker thread invokes nothing after it)
        rows["value"][bad[:2]] ^= 1                                                   # a reply carrying another string
        rows["packed"][bad[2:4]] = (rows["packed"][bad[2:4]] & ~np.uint32(3)) | A.T_INFO   # indeterminate
        rows = np.delete(rows, bad[4])                                                 # an invocation that never completes
        ms.append((rows, pay)); want.append(5)
    res = E.check_set_full_batch(ms, cfg.concurrency, A.WL_ECHO, max_values=32)
    assert list(res["error_count"]) == want and (res["valid"] == 0).all()
