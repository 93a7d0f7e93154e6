"""A direct Python transliteration of demo/clojure/kafka.clj (the node) and of the lin-kv service under it (service.clj:31-61) with REAL
values — chunks are lists of messages, the committed offsets a dict — written from the Clojure source, independently of
oracle/kafka_nodes.inc (which keeps a count per chunk and a version per offsets map).  tests/test_kafka.py replays the oracle's own network
schedule through it: every message the model emits must be the one the oracle emitted.  Test infrastructure only.

Each request runs in its own future in the reference (node.clj:173-195) and blocks on the RPCs it makes; here a handler is a generator that
yields (service, body) for every RPC and is resumed with the reply body (an {"type": "error", "code": ..} reply raises, node.clj:131-139)."""

CHUNK_SIZE = 32          # kafka.clj:19-21


class RpcError(Exception):
    def __init__(self, code):
        super().__init__(code)
        self.code = code


def chunk_key(log_key, offset):              # kafka.clj:44-51
    return f"log-{log_key}-{(offset - offset % CHUNK_SIZE) // CHUNK_SIZE}"


class KafkaNode:
    def __init__(self):
        self.offset_cache = {}               # :28-30
        self.next_msg_id = 0
        self.rpcs = {}                       # msg_id -> generator waiting for the reply (node.clj:121-139)
        self.out = []                        # (dest, body) in the order they are printed

    def bump_offset_cache(self, log_key, offset):   # :32-42
        self.offset_cache[log_key] = max(self.offset_cache.get(log_key, 0), offset)

    # -- futures as generators -----------------------------------------------------------------------------------------
    def _rpc(self, gen, dest, body):
        self.next_msg_id += 1
        self.rpcs[self.next_msg_id] = gen
        self.out.append((dest, dict(body, msg_id=self.next_msg_id)))

    def _step(self, gen, value=None, error=None):
        try:
            dest, body = gen.throw(error) if error is not None else gen.send(value)
        except StopIteration:
            return
        self._rpc(gen, dest, body)

    def handle(self, src, body):
        """one line of stdin (node.clj:173-195); returns the messages printed while handling it"""
        self.out = []
        if "in_reply_to" in body:
            gen = self.rpcs.pop(body["in_reply_to"], None)
            if gen is not None:
                if body["type"] == "error":
                    self._step(gen, error=RpcError(body["code"]))
                else:
                    self._step(gen, value=body)
        else:
            handler = {"send": self.h_send, "poll": self.h_poll, "list_committed_offsets": self.h_list, "commit_offsets": self.h_commit}.get(body["type"])
            if body["type"] == "init":
                self.out.append((src, {"type": "init_ok", "in_reply_to": body["msg_id"]}))
            elif handler:
                self._step(self._guard(handler, src, body))
        return self.out

    def _guard(self, handler, src, body):    # node.clj:180-194: an ex-info becomes an error reply
        try:
            yield from handler(src, body)
        except RpcError as e:
            self.out.append((src, {"type": "error", "code": e.code, "in_reply_to": body["msg_id"]}))

    def reply(self, src, req, body):         # node.clj:116-119
        self.out.append((src, dict(body, in_reply_to=req["msg_id"])))

    # -- kafka.clj ------------------------------------------------------------------------------------------------------
    def get_chunk(self, log_key, offset):    # :53-65 (the exceptionally of the callers is applied here: any error => [])
        try:
            res = yield ("lin-kv", {"type": "read", "key": chunk_key(log_key, offset)})
        except RpcError:
            return []
        chunk = res["value"]
        self.bump_offset_cache(log_key, offset - offset % CHUNK_SIZE + len(chunk))
        return chunk

    def try_append(self, log_key, msg):      # :67-96
        while True:
            offset = self.offset_cache.get(log_key, 0)
            chunk = yield from self.get_chunk(log_key, offset)
            i = len(chunk)
            if CHUNK_SIZE <= i:              # chunk full: bump our offset and retry
                self.offset_cache[log_key] = max(self.offset_cache.get(log_key, 0), offset - offset % CHUNK_SIZE + CHUNK_SIZE)
                continue
            yield ("lin-kv", {"type": "cas", "key": chunk_key(log_key, offset), "from": list(chunk), "to": list(chunk) + [msg], "create_if_not_exists": True})
            offset2 = offset - offset % CHUNK_SIZE + i
            self.bump_offset_cache(log_key, offset2 + 1)
            return offset2

    def h_send(self, src, req):              # :98-110
        try:
            offset = yield from self.try_append(req["key"], req["msg"])
        except RpcError as e:
            if e.code == 22:
                raise RpcError(30)           # "cas conflict"
            raise
        self.reply(src, req, {"type": "send_ok", "offset": offset})

    def h_poll(self, src, req):              # :112-139: lazy seqs — one chunk is read and dereferenced before the next read is sent
        msgs = {}
        for k, offset in req["offsets"].items():
            chunk = yield from self.get_chunk(k, offset)
            i0 = offset % CHUNK_SIZE
            msgs[k] = [[offset + j, m] for j, m in enumerate(chunk[i0:])]
        self.reply(src, req, {"type": "poll_ok", "msgs": msgs})

    def get_offsets(self):                   # :141-147
        try:
            res = yield ("lin-kv", {"type": "read", "key": "offsets"})
        except RpcError:
            return {}
        return res["value"]

    def h_list(self, src, req):              # :149-159
        offsets = yield from self.get_offsets()
        self.reply(src, req, {"type": "list_committed_offsets_ok", "offsets": {k: offsets[k] for k in req["keys"] if k in offsets}})

    def h_commit(self, src, req):            # :161-178
        offsets = yield from self.get_offsets()
        merged = dict(offsets)
        for k, o in req["offsets"].items():
            merged[k] = max(merged.get(k, o), o)
        try:
            yield ("lin-kv", {"type": "cas", "key": "offsets", "from": offsets, "to": merged, "create_if_not_exists": True})
        except RpcError as e:
            if e.code == 22:
                raise RpcError(30)
            raise
        self.reply(src, req, {"type": "commit_offsets_ok"})


class LinKV:
    """service.clj:31-61 (PersistentKV) behind the Linearizable wrapper (:141-155)"""

    def __init__(self):
        self.m = {}

    def handle(self, body):
        k = body["key"]
        if body["type"] == "read":
            if k in self.m:
                return {"type": "read_ok", "value": self.m[k]}
            return {"type": "error", "code": 20}
        if body["type"] == "cas":
            if k in self.m:
                if self.m[k] == body["from"]:
                    self.m[k] = body["to"]
                    return {"type": "cas_ok"}
                return {"type": "error", "code": 22}
            if body.get("create_if_not_exists"):
                self.m[k] = body["to"]
                return {"type": "cas_ok"}
            return {"type": "error", "code": 20}
        raise ValueError(body)
