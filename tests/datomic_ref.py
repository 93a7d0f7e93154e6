"""demo/ruby/datomic_list_append.rb (+ node.rb, promise.rb) in Python, class by class, with the tree nodes, their maps and the lists
MATERIALISED — the opposite of oracle/dt_nodes.inc, which keeps counts and versions only.  A blocking call of the Ruby program
(`sync_rpc!`, `Promise#await`, `Mutex#synchronize`) is a `yield` of a coroutine here, so that the same code can be driven message
by message.  TEST INFRASTRUCTURE: a cross-check written by the same author as the oracle (there is no Ruby to run the original),
it pins nothing against the reference."""
import zlib

RING_SIZE = 128       # datomic_list_append.rb:53
BRANCH_FACTOR = 8     # :56
VALUE_SVC, ROOT_SVC, ROOT = "lww-kv", "lin-kv", "root"


class RPCError(Exception):
    def __init__(self, code, text):
        super().__init__(text)
        self.code, self.text = code, text


class Promise:
    """promise.rb without the clock: `yield ("await", p)` parks the coroutine until deliver()"""
    WAITING = object()

    def __init__(self):
        self.value, self.waiters = Promise.WAITING, []

    def deliver(self, v):
        self.value = v
        ws, self.waiters = self.waiters, []
        for w in ws:
            w(v)
        return self


def tree_hash(k):  # :60-64
    return zlib.crc32(str(k).encode()) % RING_SIZE


class Tree:
    def __init__(self, node, ptr, rng, saved):
        self.node, self.ptr, self.range, self.saved = node, ptr, list(rng), saved

    @staticmethod
    def empty(node):  # :67-69
        return Leaf(node, "empty", [0, RING_SIZE], False, {})

    @staticmethod
    def from_json(node, ptr, json):  # :72-80
        if json["type"] == "branch":
            return Branch(node, ptr, json["range"], True, [list(p) for p in json["branches"]])
        return Leaf(node, ptr, json["range"], True, {k: list(v) for k, v in json["pairs"]})

    @staticmethod
    def load(node, ptr):  # :83-101 (a coroutine: returns the tree)
        while True:
            if ptr in node.cache:
                return node.cache[ptr]
            res = yield ("sync_rpc", VALUE_SVC, {"type": "read", "key": ptr})
            body = res["body"]
            if body["type"] == "read_ok":
                tree = Tree.from_json(node, ptr, body["value"])
                node.cache[ptr] = tree
                return tree

    def bounds_check(self, k):  # :117-123
        h = tree_hash(k)
        assert self.range[0] <= h < self.range[1], (k, h, self.range)

    def save_this(self):  # :127-143
        p = Promise()
        self.node.rpc(VALUE_SVC, {"type": "write", "key": self.ptr, "value": self.to_json()},
                      lambda res: p.deliver(res["body"]["type"] == "write_ok"))
        return p


class Leaf(Tree):
    def __init__(self, node, ptr, rng, saved, m):
        super().__init__(node, ptr, rng, saved)
        self.map = m

    def get(self, k):  # :156-159 (a coroutine like Branch#[])
        self.bounds_check(k)
        return self.map.get(k)
        yield  # pragma: no cover

    def assoc(self, k, v):  # :161-197
        self.bounds_check(k)
        if k in self.map or len(self.map) < BRANCH_FACTOR:
            return Leaf(self.node, self.node.new_ptr(), self.range, False, {**self.map, k: v})
        lower, upper = self.range
        branch_size = (upper - lower) // BRANCH_FACTOR
        merged = {**self.map, k: v}
        branches = []
        for i in range(BRANCH_FACTOR):
            b_lower = lower + i * branch_size
            b_upper = upper if i == BRANCH_FACTOR - 1 else b_lower + branch_size
            b_map = {kk: vv for kk, vv in merged.items() if b_lower <= tree_hash(kk) < b_upper}
            branches.append([b_upper, Leaf(self.node, self.node.new_ptr(), [b_lower, b_upper], False, b_map)])
        return Branch(self.node, self.node.new_ptr(), [lower, upper], False, branches)

    def to_json(self):
        return {"type": "leaf", "range": list(self.range), "pairs": [[k, list(v)] for k, v in self.map.items()]}

    def save(self):  # :212-224
        if self.saved:
            return Promise().deliver(True)
        p = self.save_this()
        p.waiters.append(lambda ok: setattr(self, "saved", True) if ok else None)
        return p


class Branch(Tree):
    def __init__(self, node, ptr, rng, saved, branches):
        super().__init__(node, ptr, rng, saved)
        self.branches = branches

    def branch_index(self, k):  # :234-250 (a coroutine: loads the branch it returns)
        self.bounds_check(k)
        h = tree_hash(k)
        for i, pair in enumerate(self.branches):
            upper, branch = pair
            if h < upper:
                if not isinstance(branch, Tree):
                    self.branches[i][1] = yield from Tree.load(self.node, branch)
                return i
        raise AssertionError("no branch for %r" % (k,))

    def get(self, k):  # :252-255
        i = yield from self.branch_index(k)
        return (yield from self.branches[i][1].get(k))

    def assoc(self, k, v):  # :257-268 (a coroutine because of branch_index)
        i = yield from self.branch_index(k)
        branches = list(self.branches)          # @branches.clone: the inner pairs stay shared
        upper, branch = branches[i]
        sub = branch.assoc(k, v)
        if not isinstance(sub, Tree):
            sub = yield from sub
        branches[i] = [upper, sub]
        return Branch(self.node, self.node.new_ptr(), self.range, False, branches)

    def to_json(self):  # :278-288
        return {"type": "branch", "range": list(self.range),
                "branches": [[u, b.ptr if isinstance(b, Tree) else b] for u, b in self.branches]}

    def save(self):  # :291-320
        if self.saved:
            return Promise().deliver(True)
        tasks = [b.save() for _, b in self.branches if isinstance(b, Tree)]
        tasks.append(self.save_this())
        p = Promise()
        left = [len(tasks)]
        oks = []

        def one(ok):
            oks.append(ok)
            left[0] -= 1
            if left[0] == 0:
                if all(oks):
                    self.saved = True
                p.deliver(all(oks))
        for t in tasks:
            if t.value is not Promise.WAITING:
                one(t.value)
            else:
                t.waiters.append(one)
        return p


class DatomicListAppendNode:
    """:323-424 over the message loop of node.rb:147-183"""

    AWAIT_US = 5_000_000          # Promise::TIMEOUT (promise.rb:5), in the replay's virtual microseconds

    def __init__(self, send, clock=lambda: 0):
        self.send = send              # send(dest, body)
        self.clock = clock            # now, in microseconds
        self.waits = {}               # thread -> (deadline, token, step) while it sits in Promise#await
        self.node_id, self.node_ids = None, None
        self.next_msg_id, self.callbacks = 0, {}
        self.ptr, self.cache = 0, {}
        self.lock_holder, self.lock_waiters = None, []

    # ---- node.rb ----
    def rpc(self, dest, body, handler):  # node.rb:91-98
        self.next_msg_id += 1
        self.callbacks[self.next_msg_id] = handler
        self.send(dest, {**body, "msg_id": self.next_msg_id})

    def reply(self, req, body):
        self.send(req["src"], {**body, "in_reply_to": req["body"]["msg_id"]})

    def handle(self, msg):  # node.rb:147-183: every message in its own thread
        body = msg["body"]
        if "in_reply_to" in body:
            h = self.callbacks.pop(body["in_reply_to"], None)
            if h:
                h(msg)
            return
        co = self.on_init(msg) if body["type"] == "init" else self.on_txn(msg)
        self._thread(co, msg)

    def _thread(self, co, msg):
        tokens = [0]

        def step(value=None, exc=None):
            self.waits.pop(id(co), None)
            try:
                req = co.throw(exc) if exc is not None else co.send(value)
            except StopIteration:
                return
            except RPCError as e:
                # The exception leaves the synchronize block (the lock is free), then node.rb:172-173 answers.  Whether the woken
                # waiter's first message or this answer is written first is the Ruby scheduler's choice; the thread that raised keeps
                # the interpreter until it blocks, so the answer goes first — the order DESIGN.md §2.4 fixes.
                self.reply(msg, {"type": "error", "code": e.code, "text": e.text})
                self._unlock(co)
                return
            tokens[0] += 1
            tok = tokens[0]

            def resume(v):   # a late delivery finds its promise abandoned (the await has given up)
                w = self.waits.get(id(co))
                if w is not None and w[1] == tok:
                    step(v)
            if req[0] == "sync_rpc":       # node.rb:117-123: rpc! + Promise#await
                self.waits[id(co)] = (self.clock() + self.AWAIT_US, tok, step)
                self.rpc(req[1], req[2], resume)
            elif req[0] == "await":
                if req[1].value is not Promise.WAITING:
                    step(req[1].value)
                else:
                    self.waits[id(co)] = (self.clock() + self.AWAIT_US, tok, step)
                    req[1].waiters.append(resume)
            elif req[0] == "lock":
                if self.lock_holder is None:
                    self.lock_holder = co
                    step(None)
                else:
                    self.lock_waiters.append((co, step))
            elif req[0] == "unlock":
                self._unlock(co)
                step(None)
        step(None)

    def fire_due(self, now):
        """Promise#await's timeout (promise.rb:17-30) for every thread whose wait began 5 s ago or more: RPCError.timeout"""
        fired = False
        for key, (deadline, _, step) in list(self.waits.items()):
            if deadline <= now and key in self.waits:
                fired = True
                step(exc=RPCError(0, "promise timed out"))
        return fired

    def _unlock(self, co):
        if self.lock_holder is co:
            self.lock_holder = None
            if self.lock_waiters:
                nxt, step = self.lock_waiters.pop(0)
                self.lock_holder = nxt
                step(None)

    # ---- datomic_list_append.rb ----
    def new_ptr(self):  # :352-355
        self.ptr += 1
        return "%s-%d" % (self.node_id, self.ptr)

    def on_init(self, msg):  # node.rb:23-38 + :337-345
        self.node_id, self.node_ids = msg["body"]["node_id"], msg["body"]["node_ids"]
        if self.node_ids[0] == self.node_id:
            t = Tree.empty(self)
            ok = yield ("await", t.save())
            if not ok:
                raise RPCError(14, "Couldn't write initial state")
            yield ("sync_rpc", ROOT_SVC, {"type": "write", "key": ROOT, "value": t.ptr})
        self.reply(msg, {"type": "init_ok"})

    def on_txn(self, msg):  # :347-372
        yield ("lock",)
        txn = msg["body"]["txn"]
        tree1 = yield from self.current_tree()
        tree2, txn2 = yield from self.apply_txn(tree1, txn)
        if tree1.ptr != tree2.ptr:
            ok = yield ("await", tree2.save())
            if not ok:
                raise RPCError(14, "Couldn't save new tree")
            yield from self.advance_root(tree1.ptr, tree2.ptr)
        # `@node.reply!` is the last statement inside the synchronize block: the answer leaves, then the lock
        self.reply(msg, {"type": "txn_ok", "txn": txn2})
        yield ("unlock",)

    def current_tree(self):  # :358-365
        res = yield ("sync_rpc", ROOT_SVC, {"type": "read", "key": ROOT})
        if res["body"]["type"] == "read_ok":
            return (yield from Tree.load(self, res["body"]["value"]))
        raise RPCError(14, "Unsure how to handle %r" % (res["body"],))

    def advance_root(self, p1, p2):  # :376-388
        res = yield ("sync_rpc", ROOT_SVC, {"type": "cas", "key": ROOT, "from": p1, "to": p2})
        if res["body"]["type"] != "cas_ok":
            raise RPCError(30, "pointer no longer %s" % p1)

    def apply_txn(self, tree, txn):  # :391-415
        txn2, t = [], tree
        for f, k, v in txn:
            if f == "r":
                txn2.append([f, k, (yield from t.get(k))])
            else:
                txn2.append([f, k, v])
                cur = yield from t.get(k)
                lst = list(cur) if cur is not None else []
                lst.append(v)
                nt = t.assoc(k, lst)
                t = nt if isinstance(nt, Tree) else (yield from nt)
        return t, txn2


class LinKV:
    """service.clj:31-61 behind :141-155"""

    def __init__(self):
        self.m = {}

    def handle(self, body):
        k = body["key"]
        if body["type"] == "read":
            return {"type": "read_ok", "value": self.m[k]} if k in self.m else {"type": "error", "code": 20}
        if body["type"] == "write":
            self.m[k] = body["value"]
            return {"type": "write_ok"}
        if k not in self.m:
            if body.get("create_if_not_exists"):
                self.m[k] = body["to"]
                return {"type": "cas_ok"}
            return {"type": "error", "code": 20}
        if self.m[k] != body["from"]:
            return {"type": "error", "code": 22}
        self.m[k] = body["to"]
        return {"type": "cas_ok"}


class LwwKV:
    """service.clj:214-243 as written over :65-114: two replicas that never exchange state; `rand_int(2)` is called three times per
    request (merge source, merge destination — computed and dropped — and the replica that serves it)"""

    def __init__(self, rand_int):
        self.rand_int, self.replicas = rand_int, [{}, {}]

    def handle(self, body):
        self.rand_int(2)
        self.rand_int(2)
        r = self.replicas[self.rand_int(2)]
        k = body["key"]
        if body["type"] == "write":
            r[k] = body["value"]
            return {"type": "write_ok"}
        return {"type": "read_ok", "value": r[k]} if k in r else {"type": "error", "code": 20}
