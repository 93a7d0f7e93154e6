"""The reference's own test of the sequential key-value service, test/maelstrom/service_test.clj:6-53, run against the oracle's
seq-kv (svc_nodes.inc, the restatement of service.clj:161-210 that the engine's svc_kernel<> is bit-compared with in
tests/test_parity_gpu.py::test_lin_kv_proxy_and_services_parity).  The reference builds `(s/sequential 16 (s/persistent-kv))`;
the engine's ring holds 32 states (service.clj:207 default), so `prep` writes buf/2 = 16 values here.  Same three assertions:
fresh clients may read OLD states (more than one distinct value over 64 clients), a client's own write pins it to the newest
state, and a client that keeps reading converges to the newest value."""
import ctypes as C

import numpy as np

from maelstrom_amd import _abi as A
from maelstrom_amd import engine as E
import oracle_lib as O

M_READ, M_READ_OK, M_WRITE, M_WRITE_OK, M_CAS, M_CAS_OK, M_ERROR = 9, 10, 14, 15, 16, 17, 18
BUF = 32
WRITE_COUNT = BUF // 2
KEY_X, KEY_ENSURE = 1, 2


def _service(service, requests, seed):
    """requests: [(client id, type, key, v1, v2)] -> [(reply type, value)]"""
    lib = O.load()
    lib.oracle_svc_trace.argtypes = [C.POINTER(A.Config), C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p]
    lib.oracle_svc_trace.restype = C.c_int
    cfg = E.test_config("lin-kv", bin="lin-kv-proxy", proxy_service=service, node_count=5, rate=10, time_limit=5)
    inp = np.array([[c, t, k | (v1 << 8) | (v2 << 16)] for c, t, k, v1, v2 in requests], dtype=np.uint32)
    out = np.zeros((len(requests), 2), dtype=np.uint32)
    assert lib.oracle_svc_trace(C.byref(cfg), seed, inp.ctypes.data, len(requests), out.ctypes.data) == len(requests)
    return [(int(t), int(a)) for t, a in out]


def _prep():
    """service_test.clj:10-19: client c0 writes 0, 1, ..., write-count - 1 to :x"""
    return [(0, M_WRITE, KEY_X, i, 0xFF) for i in range(WRITE_COUNT)]


def test_prep_writes_are_acknowledged():
    assert all(t == M_WRITE_OK for t, _ in _service("seq-kv", _prep(), 1))


def test_fresh_client_reads_return_old_state():
    """service_test.clj:20-28: 64 clients that never talked to the service read :x — they may be served from any of the states
    still in the ring, so more than one distinct value comes back."""
    for seed in range(8):
        reqs = _prep() + [(10 + i, M_READ, KEY_X, 0xFF, 0xFF) for i in range(64)]
        res = _service("seq-kv", reqs, seed)[WRITE_COUNT:]
        vals = {a if t == M_READ_OK else None for t, a in res}   # the oldest state has no :x yet: error 20
        assert len(vals) > 1
        assert all((t == M_READ_OK and a < WRITE_COUNT) or (t == M_ERROR and a == 20) for t, a in res)


def test_ensuring_recent_read_by_writing_something_unique():
    """service_test.clj:30-38: a write changes the state, so it runs on the newest one and moves the client there; the client's
    next read of :x returns the last value written."""
    for i in range(8):
        client = 100 + i
        reqs = _prep() + [(client, M_WRITE, KEY_ENSURE, i, 0xFF), (client, M_READ, KEY_X, 0xFF, 0xFF)]
        res = _service("seq-kv", reqs, 50 + i)
        assert res[-2][0] == M_WRITE_OK
        assert res[-1] == (M_READ_OK, WRITE_COUNT - 1)


def test_ensuring_recent_reads_by_reading_a_ton():
    """service_test.clj:40-53: every read moves the client's index forward at random and never back: it reaches the newest
    state within buf * 10 tries."""
    for i in range(8):
        tries = BUF * 10
        reqs = _prep() + [(200, M_READ, KEY_X, 0xFF, 0xFF)] * tries
        res = _service("seq-kv", reqs, 90 + i)[WRITE_COUNT:]
        vals = [a if t == M_READ_OK else -1 for t, a in res]
        assert WRITE_COUNT - 1 in vals, "never converged"
        assert all(b >= a for a, b in zip(vals, vals[1:])), "a client's reads went back in time"
        first = vals.index(WRITE_COUNT - 1)
        assert all(v == WRITE_COUNT - 1 for v in vals[first:])


def test_linearizable_service_always_reads_the_newest_state():
    """the same script against lin-kv (service.clj:141-155): no stale reads at all"""
    reqs = _prep() + [(10 + i, M_READ, KEY_X, 0xFF, 0xFF) for i in range(64)]
    res = _service("lin-kv", reqs, 3)[WRITE_COUNT:]
    assert all(r == (M_READ_OK, WRITE_COUNT - 1) for r in res)
