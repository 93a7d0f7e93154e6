"""Direct Python transliterations of the reference's key-value services (src/maelstrom/service.clj) and of the proxying
node demo/ruby/lin_kv_proxy.rb (+ the RPC part of demo/ruby/node.rb) — written from those sources, independently of
oracle/svc_nodes.inc (immutable maps copied on write, a deque as ring buffer, no packing) — to cross-check the oracle's
restatement by replaying its network schedule.  Test infrastructure only."""
import collections


def assoc(m, k, v):
    """(assoc m k v): a new map"""
    m2 = dict(m)
    m2[k] = v
    return m2


def persistent_kv_handle(m, body):               # PersistentKV/handle, service.clj:31-56 -> (m', response body)
    k = body["key"]
    t = body["type"]
    if t == "read":
        return (m, {"type": "read_ok", "value": m[k]}) if k in m else (m, {"type": "error", "code": 20})
    if t == "write":
        return assoc(m, k, body["value"]), {"type": "write_ok"}
    if k in m:                                   # cas
        if body["from"] == m[k]:
            return assoc(m, k, body["to"]), {"type": "cas_ok"}
        return m, {"type": "error", "code": 22}
    if body.get("create_if_not_exists"):
        return assoc(m, k, body["to"]), {"type": "cas_ok"}
    return m, {"type": "error", "code": 20}


class LWWKV:                                     # service.clj:65-114: clock, m = {k: {ts, value}}
    def __init__(self, clock=0, m=None):
        self.clock, self.m = clock, m or {}

    def __eq__(self, other):
        return (self.clock, self.m) == (other.clock, other.m)

    def handle(self, body):
        k, t, m = body["key"], body["type"], self.m
        if t == "read":
            return (self, {"type": "read_ok", "value": m[k]["value"]}) if k in m else (self, {"type": "error", "code": 20})
        if t == "write":
            return LWWKV(self.clock + 1, assoc(m, k, {"ts": self.clock, "value": body["value"]})), {"type": "write_ok"}
        if k in m:
            if body["from"] == m[k]["value"]:
                return LWWKV(self.clock + 1, assoc(m, k, {"ts": self.clock, "value": body["to"]})), {"type": "cas_ok"}
            return self, {"type": "error", "code": 22}
        return self, {"type": "error", "code": 20}

    def merge(self, other):                      # merge-services, :99-114
        m = dict(self.m)
        for k, v2 in other.m.items():
            v1 = m.get(k)
            m[k] = v2 if v1 is None or v1["ts"] < v2["ts"] else v1
        return LWWKV(max(self.clock, other.clock), m)


class Linearizable:                              # :141-155
    def __init__(self):
        self.state = {}

    def handle(self, src, body, rand_int):
        self.state, res = persistent_kv_handle(self.state, body)
        return res


class Sequential:                                # :161-210, (sequential 32 (persistent-kv))
    def __init__(self, size=32):
        self.buffer = collections.deque([{}], maxlen=size)
        self.last_index, self.clients = 0, {}

    def handle(self, src, body, rand_int):
        client_index = self.clients.get(src, 0)
        index = rand_int(self.last_index - client_index + 1) + client_index
        assert client_index <= index <= self.last_index
        service = self.buffer[index - self.last_index - 1]    # (nth buffer (dec (- index last-index))): raises beyond the ring
        service2, res = persistent_kv_handle(service, body)
        if service == service2:
            self.clients[src] = index
            return res
        service2, res = persistent_kv_handle(self.buffer[-1], body)
        self.last_index += 1
        self.clients[src] = self.last_index
        self.buffer.append(service2)
        return res


class Eventual:                                  # :214-243, (eventual (lww-kv)), 2 replicas
    def __init__(self, n=2):
        self.replicas = [LWWKV() for _ in range(n)]

    def handle(self, src, body, rand_int):
        replicas = self.replicas
        n = len(replicas)
        merge_source, merge_dest = rand_int(n), rand_int(n)
        merged = replicas[merge_source].merge(replicas[merge_dest])
        replicas2 = list(replicas); replicas2[merge_dest] = merged        # noqa: E702  (the first replicas' of the let)
        i = rand_int(n)
        replica2, res = replicas[i].handle(body)
        replicas2 = list(replicas); replicas2[i] = replica2               # noqa: E702  (the second one starts from `replicas` again)
        self.replicas = replicas2
        return res


class ProxyNode:                                 # lin_kv_proxy.rb:8-43 over node.rb:95-102,148-182
    def __init__(self, service_name):
        self.service, self.next_msg_id, self.callbacks = service_name, 0, {}

    def on_request(self, src, body):             # proxy!: strip msg_id, rpc! the service -> (dest, body)
        proxy_body = {k: v for k, v in body.items() if k != "msg_id"}
        self.next_msg_id += 1
        self.callbacks[self.next_msg_id] = (src, body["msg_id"])
        return self.service, dict(proxy_body, msg_id=self.next_msg_id)

    def on_reply(self, body):                    # the callback: strip msg_id, reply! to the client -> (dest, body) or None
        cb = self.callbacks.pop(body["in_reply_to"], None)
        if cb is None:
            return None                          # "Ignoring reply ... with no callback"
        res = {k: v for k, v in body.items() if k not in ("msg_id", "in_reply_to")}
        return cb[0], dict(res, in_reply_to=cb[1])


class SingleKeyTxnNode:                          # demo/clojure/single_key_txn.clj:116-180 (one future per request)
    def __init__(self, service):
        self.service, self.next_msg_id, self.pending = service, 0, {}

    @staticmethod
    def apply_txn(state, txn):                   # apply-txn, :118-131
        state, out = dict(state), []
        for f, k, v in txn:
            if f == "r":
                out.append([f, k, state.get(k)])
            else:
                state[k] = list(state.get(k, [])) + [v]
                out.append([f, k, v])
        return state, out

    def rpc(self, body, cont):
        self.next_msg_id += 1
        self.pending[self.next_msg_id] = cont
        return self.service, dict(body, msg_id=self.next_msg_id)

    def on_txn(self, src, body):                 # handle-txn!: first the read of the root (read-service, :150-157)
        return self.rpc({"type": "read", "key": "root"}, ("read", src, body))

    def on_reply(self, body):                    # -> (dest, body) of the next message
        cont = self.pending.pop(body["in_reply_to"], None)
        if cont is None:
            return None
        if cont[0] == "read":
            _, src, req = cont
            if body["type"] == "read_ok":
                state = body["value"]
            elif body["type"] == "error" and body["code"] == 20:
                state = None                     # not found => nil
            else:
                return src, {"type": "error", "code": body["code"], "in_reply_to": req["msg_id"]}
            pairs = state or []
            m = {pairs[i]: pairs[i + 1] for i in range(0, len(pairs), 2)}          # pairs->map
            m2, txn2 = self.apply_txn(m, req["txn"])
            to = [x for kv in m2.items() for x in kv]                               # map->pairs
            return self.rpc({"type": "cas", "key": "root", "from": state, "to": to, "create_if_not_exists": True}, ("cas", src, req, txn2))
        _, src, req, txn2 = cont
        if body["type"] == "cas_ok":
            return src, {"type": "txn_ok", "txn": txn2, "in_reply_to": req["msg_id"]}
        return src, {"type": "error", "code": 30 if body["code"] == 22 else body["code"], "in_reply_to": req["msg_id"]}   # "root altered", :176-178
