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


# ---- the oracle's proxy node and services against transliterations of service.clj / lin_kv_proxy.rb ----
def _replay(cfg, service, inst):
    import collections
    import services_ref as R
    lib = O.load()
    r = O.run(cfg, inst, 1)
    assert r.meta["flags"][0] == 0 and r.meta["n_events"][0] <= cfg.journal_capacity
    N, SVC = cfg.n_nodes, cfg.n_nodes + max(cfg.concurrency, cfg.n_nodes)
    T = {name: i for i, name in enumerate(A.MSG_TYPES)}
    ctr = [0]

    def rand_int(n):    # the service's rand-int draws: stream 12 of the instance, in request order
        x = lib.oracle_draw32(cfg.seed, inst, 12, ctr[0]); ctr[0] += 1
        return (x * n) >> 32
    svc = {"lin-kv": R.Linearizable, "seq-kv": R.Sequential, "lww-kv": R.Eventual}[service]()
    nodes = [R.ProxyNode(SVC) for _ in range(N)]
    out = collections.defaultdict(collections.deque)   # endpoint -> bodies it still has to send, in order
    content, n_req = {}, 0

    def body_of(typ, a, b):    # envelope -> message body (lin-kv encoding: key | v << 8 | v' << 16, 0xFF = absent)
        name = A.MSG_TYPES[typ]
        k, v1, v2 = a & 0xFF, (a >> 8) & 0xFF, (a >> 16) & 0xFF
        if name == "read":
            return {"type": "read", "key": k, "msg_id": b}
        if name == "write":
            return {"type": "write", "key": k, "value": v1, "msg_id": b}
        if name == "cas":
            return {"type": "cas", "key": k, "from": v1, "to": v2, "msg_id": b}
        return {"type": name, "msg_id": b}

    def same(body, typ, a, b):  # does a body the model wants to send equal the envelope the oracle sent?
        name = A.MSG_TYPES[typ]
        if body["type"] != name:
            return False
        if name in ("read", "write", "cas"):
            want = body["key"] | (body.get("value", body.get("from", 0xFF)) << 8) | (body.get("to", 0xFF) << 16)
            return want == a and (body["msg_id"] & 0xFFFF) == b
        val = {"read_ok": body.get("value"), "error": body.get("code")}.get(name, 0)
        return val == a and (body.get("in_reply_to", 0) & 0xFFFF) == b
    for ev in r.events(0):
        msg, a, route = int(ev["msg"]), int(ev["a"]), int(ev["route"])
        mid, recv, typ = msg >> 8, (msg >> 7) & 1, msg & 0x7F
        src, dest, b = route & 0xFF, (route >> 8) & 0xFF, route >> 16
        if not recv:
            if N <= src < SVC:
                content[mid] = body_of(typ, a, b)    # a client's request (or init)
                continue
            assert out[src], (src, A.MSG_TYPES[typ])
            to, body = out[src].popleft()
            assert to == dest and same(body, typ, a, b), (src, dest, to, body, A.MSG_TYPES[typ], a, b)
            content[mid] = body
        elif dest < N:
            body = content[mid]
            if body["type"] == "init":
                out[dest].append((src, {"type": "init_ok", "in_reply_to": body["msg_id"]}))
            elif "in_reply_to" in body:
                fwd = nodes[dest].on_reply(body)
                if fwd:
                    out[dest].append(fwd)
            else:
                out[dest].append(nodes[dest].on_request(src, body)); n_req += 1   # noqa: E702
        elif dest == SVC:
            body = content[mid]
            try:
                res = svc.handle(src, body, rand_int)
            except IndexError:
                continue    # "Error in service worker!" (service.clj:259-260): no reply
            out[SVC].append((src, dict(res, in_reply_to=body["msg_id"])))
    assert not any(out.values())
    return n_req


@pytest.mark.parametrize("service,kw", [("lin-kv", dict()), ("lin-kv", dict(latency=20, latency_dist="exponential", p_loss=0.05)),
                                        ("seq-kv", dict(rate=100)), ("seq-kv", dict(node_count=3, latency=30, latency_dist="uniform", nemesis=["partition"], nemesis_interval=3)),
                                        ("lww-kv", dict(rate=100)), ("lww-kv", dict(node_count=7, concurrency=28, latency=10))])
def test_oracle_proxy_and_services_equal_transliterated_reference(service, kw):
    cfg = _cfg(service, time_limit=10, journal_capacity=200000, **kw)
    for inst in range(3):
        assert _replay(cfg, service, inst) > 100
