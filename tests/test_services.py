"""The lin-kv workload over demo/ruby/lin_kv_proxy.rb and the Maelstrom-provided key-value services (service.clj;
SURVEY.md §8a row a17), CPU side: message accounting, and what the reference's comment promises (lin_kv_proxy.rb:33-35
"Replace with seq-kv or lww-kv to see linearization failures!") — the linearizability checker passes every lin-kv
history and catches the weaker services."""
import ctypes as C

import numpy as np
import pytest

from maelstrom_amd import _abi as A
from maelstrom_amd import engine as E
import oracle_lib as O


def _lin(rows):
    res = A.CheckResult()
    rows = np.ascontiguousarray(rows)
    assert A.load().msim_check_lin_kv_rows(rows.ctypes.data_as(C.c_void_p), len(rows), C.byref(res)) == 0
    return res


def _cfg(service, **kw):
    base = dict(node_count=5, rate=60, time_limit=20, latency=5, seed=23)
    base.update(kw)
    return E.test_config("lin-kv", bin="lin-kv-proxy", proxy_service=service, **base)


@pytest.mark.parametrize("kw", [dict(), dict(latency=20, latency_dist="exponential"), dict(nemesis=["partition"], nemesis_interval=5)])
def test_proxy_to_lin_kv_is_linearizable_and_costs_four_messages_per_op(kw):
    cfg = _cfg("lin-kv", **kw)
    assert cfg.concurrency == 10 and cfg.node_program == A.NODE_LIN_KV_PROXY
    r = O.run(cfg, 0, 4)
    for i in range(4):
        assert r.meta["flags"][i] == 0
        rows, pay = r.history(i)
        ops = E.decode_history(rows, pay, 5, A.WL_LIN_KV)
        inv = [o for o in ops if o["type"] == ":invoke" and o["process"] != ":nemesis"]
        assert len(inv) > 500
        st = r.stats[i]
        assert int(st["clients_send"]) == 2 * (len(inv) + 5)      # request + relayed reply (+ init)
        assert int(st["servers_send"]) == 2 * len(inv)            # node -> service, service -> node
        res = _lin(rows)
        assert res.valid == 1 and res.error_count == 0
        kinds = {(o["f"], o["type"]) for o in ops}
        assert (":cas", ":fail") in kinds and (":read", ":ok") in kinds and (":write", ":ok") in kinds   # codes 20 / 22 relayed


@pytest.mark.parametrize("service", ["seq-kv", "lww-kv"])
def test_weaker_services_are_caught_by_the_checker(service):
    cfg = _cfg(service, rate=100, time_limit=30)
    r = O.run(cfg, 0, 6)
    bad = 0
    for i in range(6):
        assert r.meta["flags"][i] == 0
        rows, _ = r.history(i)
        bad += _lin(rows).valid == 0
    assert bad >= 4, bad     # stale reads from old states (seq-kv) / from the other replica (lww-kv)
