"""Host-side mirror of the reference's test-map -> history surface, above the C-ABI (include/maelsim.h).

`test_config(...)` plays the role of `maelstrom.core/maelstrom-test` (core.clj:53-102): it takes the CLI
option names of core.clj:136-229 and returns the finalized engine config.  `Engine.run` replaces
`jepsen.core/run!` for an ensemble of instances; `Engine.history(i)` yields the op maps that
`(checker/check (:checker test) test history opts)` consumes (core.clj:91-100), `Engine.net_stats(i)`
the map of net/checker.clj:28-41.  Everything here calls libmaelsim.so; there is no CPU fallback.
"""
import ctypes as C

import numpy as np

from . import _abi as A

OP_DT = np.dtype([("time_len", "<u8"), ("packed", "<u4"), ("value", "<u4")])
STATS_DT = np.dtype([(n, "<u8") for n in ("all_send", "all_recv", "clients_send", "clients_recv", "servers_send", "servers_recv")])
META_DT = np.dtype([(n, "<u4") for n in ("n_rows", "n_payload_words", "flags", "n_rounds", "n_events", "r0", "r1", "r2")])
EVENT_DT = np.dtype([(n, "<u4") for n in ("time_us", "msg", "a", "route")])
CHECK_DT = np.dtype([("valid", "<u4"), ("attempt_count", "<u4"), ("stable_count", "<u4"), ("lost_count", "<u4"),
                     ("never_read_count", "<u4"), ("stale_count", "<u4"), ("duplicated_count", "<u4"),
                     ("error_count", "<u4"), ("stable_latency_ms", "<u4", (5,)), ("op_count", "<u4"),
                     ("ok_count", "<u4"), ("fail_count", "<u4"), ("info_count", "<u4")])

WORKLOADS = {"echo": A.WL_ECHO, "broadcast": A.WL_BROADCAST, "g-set": A.WL_G_SET, "lin-kv": A.WL_LIN_KV,
             "txn-list-append": A.WL_TXN_LIST_APPEND, "pn-counter": A.WL_PN_COUNTER, "g-counter": A.WL_G_COUNTER, "unique-ids": A.WL_UNIQUE_IDS,
             "txn-rw-register": A.WL_TXN_RW_REGISTER, "kafka": A.WL_KAFKA}
NODE_PROGRAMS = {"echo": A.NODE_ECHO, "broadcast-ff": A.NODE_BCAST_FF, "broadcast-ff-echoback": A.NODE_BCAST_FF_ECHOBACK,
                 "broadcast-ack-retry": A.NODE_BCAST_ACK_RETRY, "broadcast-rpc-all": A.NODE_BCAST_RPC_ALL,
                 "g-set": A.NODE_G_SET, "raft": A.NODE_RAFT, "single-key-txn": A.NODE_TXN_SINGLE_KEY,
                 "pn-counter": A.NODE_PN_COUNTER, "flake-ids": A.NODE_FLAKE_IDS, "lin-kv-proxy": A.NODE_LIN_KV_PROXY,
                 "txn-rw-register-hat": A.NODE_TXN_RW_HAT,
                 "multi-key-txn": A.NODE_TXN_MULTI_KEY, "datomic": A.NODE_TXN_DATOMIC, "tso-ids": A.NODE_TSO_IDS, "kafka": A.NODE_KAFKA}
SERVICES = {"lin-kv": A.SVC_LIN_KV, "seq-kv": A.SVC_SEQ_KV, "lww-kv": A.SVC_LWW_KV}
CONSISTENCY_MODELS = {"strict-serializable": A.CM_STRICT_SERIALIZABLE, "serializable": A.CM_SERIALIZABLE,
                      "snapshot-isolation": A.CM_SNAPSHOT_ISOLATION, "read-committed": A.CM_READ_COMMITTED,
                      "read-uncommitted": A.CM_READ_UNCOMMITTED}
TOPOLOGIES = {"grid": A.TOPO_GRID, "line": A.TOPO_LINE, "total": A.TOPO_TOTAL, "tree": A.TOPO_TREE2,
              "tree2": A.TOPO_TREE2, "tree3": A.TOPO_TREE3, "tree4": A.TOPO_TREE4}
LATENCY_DISTS = {"constant": A.LAT_CONSTANT, "uniform": A.LAT_UNIFORM, "exponential": A.LAT_EXPONENTIAL}
TYPE_KW = {A.T_INVOKE: ":invoke", A.T_OK: ":ok", A.T_FAIL: ":fail", A.T_INFO: ":info"}
F_KW = {A.F_ECHO: ":echo", A.F_BROADCAST: ":broadcast", A.F_READ: ":read", A.F_ADD: ":add",
        A.F_START_PARTITION: ":start-partition", A.F_STOP_PARTITION: ":stop-partition", A.F_WRITE: ":write", A.F_CAS: ":cas",
        A.F_TXN: ":txn", A.F_GENERATE: ":generate", A.F_SEND: ":send", A.F_POLL: ":poll", A.F_ASSIGN: ":assign", A.F_CRASH: ":crash"}
ERR_KW = {A.ERR_NET_TIMEOUT: ":net-timeout", A.ERR_TEMPORARILY_UNAVAILABLE: [":temporarily-unavailable", "not a leader"],
          A.ERR_KEY_DOES_NOT_EXIST: [":key-does-not-exist", "not found"], A.ERR_PRECONDITION_FAILED: [":precondition-failed", "cas mismatch"],
          A.ERR_TXN_CONFLICT: [":txn-conflict", "root altered"], A.ERR_TIMEOUT: [":timeout", "promise timed out"], A.ERR_ABORT: [":abort", "aborted"]}
SPEC_KW = {A.SPEC_ONE: ":one", A.SPEC_MAJORITY: ":majority", A.SPEC_MAJORITIES_RING: ":majorities-ring",
           A.SPEC_MINORITY_THIRD: ":minority-third"}


class EngineError(RuntimeError):
    pass


def test_config(workload="broadcast", bin=None, node_count=5, concurrency=None, rate=5.0, time_limit=60.0,
                latency=0, latency_dist="constant", topology="grid", nemesis=(), nemesis_interval=10.0,
                p_loss=0.0, seed=0, **capacities):
    """Option map -> finalized msim_config.  Names follow core.clj:136-229; `bin` names a built-in node
    program (NODE_PROGRAMS) instead of an executable; `p_loss` is the net's :p-loss (net.clj:100,122)."""
    lib = A.load()
    cfg = A.Config()
    rc = lib.msim_config_defaults(C.byref(cfg), WORKLOADS[workload], int(node_count))
    if rc:
        raise EngineError(f"msim_config_defaults: {rc}")
    if bin is not None:
        cfg.node_program = NODE_PROGRAMS[bin]
    if concurrency is not None:
        cfg.concurrency = int(concurrency)
    cfg.rate_mhz = int(round(rate * 1000))
    cfg.time_limit_ms = int(round(time_limit * 1000))
    cfg.latency_mean_ms = int(latency)
    cfg.latency_dist = LATENCY_DISTS[latency_dist]
    cfg.topology = TOPOLOGIES[topology]
    cfg.nemesis_mask = A.NEMESIS_PARTITION if "partition" in set(nemesis) else 0
    cfg.nemesis_interval_ms = int(round(nemesis_interval * 1000))
    cfg.p_loss_q32 = min(int(p_loss * 2**32), 2**32 - 1)
    cfg.seed = int(seed)
    for k, v in capacities.items():
        if k == "proxy_service" and isinstance(v, str):
            v = SERVICES[v]
        if k == "consistency_model" and isinstance(v, str):
            v = CONSISTENCY_MODELS[v]
        if not hasattr(cfg, k):
            raise EngineError(f"unknown option {k}")
        setattr(cfg, k, int(v))
    err = C.create_string_buffer(256)
    rc = lib.msim_config_finalize(C.byref(cfg), err, 256)
    if rc:
        raise EngineError(f"invalid test options ({rc}): {err.value.decode()}")
    return cfg


class Engine:
    """One engine context on one HIP device (msim_create ... msim_destroy)."""

    def __init__(self, cfg, device=0):
        self.lib = A.load()
        self._ctx = C.c_void_p()
        err = C.create_string_buffer(256)
        rc = self.lib.msim_create(C.byref(cfg), device, C.byref(self._ctx), err, 256)
        if rc:
            raise EngineError(f"msim_create failed ({rc}): {err.value.decode()}")
        self.cfg = A.Config()
        self.lib.msim_get_config(self._ctx, C.byref(self.cfg))
        self.n = 0

    def _chk(self, rc, what):
        if rc:
            raise EngineError(f"{what} failed ({rc}): {self.lib.msim_last_error(self._ctx).decode()}")

    def close(self):
        if self._ctx:
            self.lib.msim_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, first_instance, n_instances):
        self._chk(self.lib.msim_run(self._ctx, first_instance, n_instances), "msim_run")
        self.n = n_instances

    def run_async(self, first_instance, n_instances, stream=None):
        self._chk(self.lib.msim_run_async(self._ctx, first_instance, n_instances, stream), "msim_run_async")
        self.n = n_instances

    def check(self):
        self._chk(self.lib.msim_check(self._ctx), "msim_check")

    def set_dev_flags(self, flags):
        """Developer switches of this context (msim_set_dev_flags): e.g. 0x200 one cluster per wavefront, 0x800 checkers on the host."""
        self._chk(self.lib.msim_set_dev_flags(self._ctx, int(flags)), "msim_set_dev_flags")

    def check_host_rechecks(self):
        """lin-kv: how many histories of the last check() the device handed to the host search."""
        return int(self.lib.msim_check_host_rechecks(self._ctx))

    def fetch(self):
        self._chk(self.lib.msim_fetch(self._ctx), "msim_fetch")

    def fetch_begin(self):
        """Compacts the histories on the device and queues their copies to the host; fetch() waits for them (msim_fetch_begin)."""
        self._chk(self.lib.msim_fetch_begin(self._ctx), "msim_fetch_begin")

    def kernel_ms(self):
        a, b = C.c_float(), C.c_float()
        self._chk(self.lib.msim_last_kernel_ms(self._ctx, C.byref(a), C.byref(b)), "msim_last_kernel_ms")
        return a.value, b.value

    def check_availability(self, availability=None):
        """maelstrom.checker/availability-checker (checker.clj:6-39) over every history of the last run: `availability` is None,
        "total" or a number from 0 to 1 (--availability, core.clj:149).  Returns [{:valid? :ok-fraction}] (+ the two counts)."""
        mode, a = _availability_mode(availability)
        res = (A.Availability * self.n)()
        self._chk(self.lib.msim_check_availability(self._ctx, mode, a, res, self.n), "msim_check_availability")
        return [{"valid?": bool(r.valid), "ok-fraction": float(r.ok_fraction), "ok-count": r.ok_count, "invoke-count": r.invoke_count} for r in res]

    # ---- multi-GPU ensemble (include/maelsim.h "multi-GPU ensemble"; maelstrom_amd/ensemble.py) ----
    @staticmethod
    def comm_unique_id():
        """rank 0: the 128-byte RCCL id every rank passes to comm_init (the host moves it: torch.distributed, files, sockets)"""
        buf = C.create_string_buffer(A.COMM_ID_BYTES)
        rc = A.load().msim_comm_unique_id(buf)
        if rc:
            raise EngineError(f"msim_comm_unique_id failed ({rc}): RCCL not available")
        return buf.raw

    def comm_init(self, comm_id, rank, world):
        self._chk(self.lib.msim_comm_init(self._ctx, bytes(comm_id), rank, world), "msim_comm_init")

    def gather(self, root=0):
        """Variable-length history gather of the last run to `root` (device-side compaction + RCCL send/recv); returns the
        msim_gathered record (device pointers are valid on the root until the next gather)."""
        g = A.Gathered()
        self._chk(self.lib.msim_gather(self._ctx, root, C.byref(g)), "msim_gather")
        return g

    def device_buffers(self):
        db = A.DeviceBuffers()
        self._chk(self.lib.msim_device_buffers_get(self._ctx, C.byref(db)), "msim_device_buffers_get")
        return db

    def raw_history(self, i):
        """(rows, payload) numpy views of instance i (after fetch)."""
        ops, n_ops = C.POINTER(A.Op)(), C.c_uint32()
        pay, n_words = C.POINTER(C.c_uint32)(), C.c_uint32()
        self._chk(self.lib.msim_history(self._ctx, i, C.byref(ops), C.byref(n_ops), C.byref(pay), C.byref(n_words)), "msim_history")
        rows = np.ctypeslib.as_array(C.cast(ops, C.POINTER(C.c_uint8)), shape=(max(n_ops.value, 1) * 16,))[: n_ops.value * 16].view(OP_DT)
        payload = np.ctypeslib.as_array(pay, shape=(max(n_words.value, 1),))[: n_words.value]
        return rows, payload

    def raw_journal(self, i):
        """numpy view of instance i's net journal (journal_capacity > 0, after fetch)."""
        ev, n = C.POINTER(A.Event)(), C.c_uint32()
        self._chk(self.lib.msim_journal(self._ctx, i, C.byref(ev), C.byref(n)), "msim_journal")
        return np.ctypeslib.as_array(C.cast(ev, C.POINTER(C.c_uint8)), shape=(max(n.value, 1) * 16,))[: n.value * 16].view(EVENT_DT)

    def journal(self, i):
        return decode_journal(self.raw_journal(i), self.cfg.n_nodes)

    def net_stats_raw(self, i):
        st = A.NetStats()
        self._chk(self.lib.msim_net_stats_get(self._ctx, i, C.byref(st)), "msim_net_stats_get")
        return st

    def meta(self, i):
        m = A.InstMeta()
        self._chk(self.lib.msim_meta(self._ctx, i, C.byref(m)), "msim_meta")
        return m

    def check_results(self):
        res, n = C.POINTER(A.CheckResult)(), C.c_uint32()
        self._chk(self.lib.msim_check_results(self._ctx, C.byref(res), C.byref(n)), "msim_check_results")
        arr = np.ctypeslib.as_array(C.cast(res, C.POINTER(C.c_uint8)), shape=(n.value * C.sizeof(A.CheckResult),))
        return arr.view(CHECK_DT).copy()  # small; a copy outlives the ctx

    # ---- Jepsen-shaped views ---------------------------------------------------------------------
    def history(self, i):
        rows, payload = self.raw_history(i)
        return decode_history(rows, payload, self.cfg.n_nodes, self.cfg.workload)

    def net_stats(self, i):
        """The :net :stats map of net/checker.clj:28-41,55-67."""
        st = self.net_stats_raw(i)
        rows, _ = self.raw_history(i)
        return net_stats_map(st, rows)


def net_stats_map(st, rows):
    typ = rows["packed"] & 3
    proc = rows["packed"] >> 12
    op_count = int(((typ == A.T_INVOKE) & (proc != A.PROCESS_NEMESIS)).sum())
    get = (lambda k: int(st[k])) if isinstance(st, (np.void, dict)) else (lambda k: int(getattr(st, k)))
    m = {k: {"send-count": get(f"{k}_send"), "recv-count": get(f"{k}_recv"), "msg-count": get(f"{k}_send")}
         for k in ("all", "clients", "servers")}
    if op_count:
        m["all"]["msgs-per-op"] = m["all"]["msg-count"] / op_count
        m["servers"]["msgs-per-op"] = m["servers"]["msg-count"] / op_count
    m["valid?"] = True
    return m


def endpoint_name(ep, n_nodes):
    return f"n{ep}" if ep < n_nodes else f"c{ep - n_nodes}"


def decode_journal(events, n_nodes):
    """Binary events -> the Event maps of net/journal.clj:53 ({:id :time :type :message {:id :src :dest :body}})."""
    out = []
    for i in range(len(events)):
        msg, route = int(events["msg"][i]), int(events["route"][i])
        body = {"type": A.MSG_TYPES[msg & 0x7F], "a": int(events["a"][i])}
        if route >> 16:
            body["msg_id/in_reply_to"] = route >> 16
        out.append({"id": i, "time": int(events["time_us"][i]) * 1000, "type": ":recv" if msg & 0x80 else ":send",
                    "message": {"id": msg >> 8, "src": endpoint_name(route & 0xFF, n_nodes),
                                "dest": endpoint_name((route >> 8) & 0xFF, n_nodes), "body": body}})
    return out


def journal_stats(events, n_nodes):
    """maelstrom.net.checker's fold over the journal (net/checker.clj:28-41): send/recv/msg counts for all,
    clients (a client endpoint involved, util.clj:12-16) and servers."""
    msg, route = events["msg"], events["route"]
    recv = (msg & 0x80) != 0
    cl = ((route & 0xFF) >= n_nodes) | (((route >> 8) & 0xFF) >= n_nodes)
    ids = msg >> 8
    res = {}
    for name, sel in (("all", np.ones(len(events), bool)), ("clients", cl), ("servers", ~cl)):
        res[name] = {"send-count": int((sel & ~recv).sum()), "recv-count": int((sel & recv).sum()),
                     "msg-count": int(len(np.unique(ids[sel])))}
    return res


def bitmap_to_list(words):
    out = []
    for wi, w in enumerate(np.asarray(words, dtype=np.uint32)):
        w = int(w)
        while w:
            b = w & -w
            out.append(wi * 32 + b.bit_length() - 1)
            w ^= b
    return out


def _nil(x):
    return None if x == 0xFF else x


def decode_rw_txn(words):
    """Payload words of one rw-register transaction -> [[f k v] ...] (txn_rw_register.clj:82-84): one word per micro-op."""
    return [[":w" if w & 1 else ":r", (w >> 1) & 0x7FFF, None if (w >> 16) & 0xFF == 0xFF else (w >> 16) & 0xFF] for w in (int(x) for x in words)]


def encode_rw_txn(txn):
    """[[f k v] ...] -> payload words (inverse of decode_rw_txn)."""
    return [(1 if f == ":w" else 0) | (k << 1) | ((0xFF if v is None else v) << 16) for f, k, v in txn]


def decode_poll(words):
    """The poll_ok block of a kafka :poll -> {key [[offset msg] ...]} (include/maelsim.h msim_op)."""
    out, i, words = {}, 0, [int(w) for w in words]
    while i < len(words):
        h = words[i]; i += 1
        key, n, o = h & 7, (h >> 8) & 0xFF, h >> 16
        msgs = []
        for e in range(n):
            msgs.append([o + e, (words[i + e // 2] >> (16 * (e % 2))) & 0xFFFF])
        i += (n + 1) // 2
        out.setdefault(str(key), []).extend(msgs)   # (a key's pairs that do not continue a run are a block of their own)
    return out


def _kafka_range(i, key, msg, offset):
    """The binary layout's fields (include/maelsim.h: keys 0..7, message values and offsets below 2047): out-of-range values are an
    error, never masked — a masked key or a count that spills into the offset bits would be CHECKED as a different history."""
    if not (0 <= int(key) < 8 and 0 <= int(msg) < 2047 and 0 <= int(offset) < 2047):
        raise EngineError(f"kafka op {i}: key {key} / message {msg} / offset {offset} outside the binary layout (keys 0..7, messages and offsets below 2047)")


def encode_kafka_history(ops):
    """Jepsen-shaped kafka ops ({type, process, f, value[, time]}; values as decode_history gives them) -> (rows, payload) in the engine's
    binary layout, so that externally produced histories can be fed to msim_check_kafka_rows."""
    tkw = {v: k for k, v in TYPE_KW.items()}
    fkw = {":send": A.F_SEND, ":poll": A.F_POLL, ":assign": A.F_ASSIGN, ":crash": A.F_CRASH}
    rows = np.zeros(len(ops), dtype=OP_DT)
    pay = []
    for i, op in enumerate(ops):
        typ, f, process = tkw[op["type"]], fkw[op["f"]], A.PROCESS_NEMESIS if op["process"] == ":nemesis" else int(op["process"])
        value, ln = A.NO_VALUE, 0
        if f == A.F_SEND:
            _, k, v = op["value"][0]
            msg, off = (v[1], v[0]) if isinstance(v, (list, tuple)) else (v, 0x7FF)
            _kafka_range(i, k, msg, off if isinstance(v, (list, tuple)) else 0)
            value = int(k) | (msg << 6) | (off << 17)
        elif f == A.F_POLL and typ == A.T_OK and len(op["value"][0]) > 1:
            value = len(pay)
            for k, pairs in op["value"][0][1].items():
                runs = []
                for o, m in pairs:   # runs of consecutive offsets, at most 255 messages each (the header's count has 8 bits)
                    _kafka_range(i, k, m, o)
                    if runs and runs[-1][0] + len(runs[-1][1]) == o and len(runs[-1][1]) < 255:
                        runs[-1][1].append(m)
                    else:
                        runs.append((o, [m]))
                if not pairs:
                    _kafka_range(i, k, 0, 0)
                for o, ms in runs or [(0, [])]:
                    pay.append(int(k) | (len(ms) << 8) | (o << 16))
                    for e in range(0, len(ms), 2):
                        pay.append(ms[e] | ((ms[e + 1] << 16) if e + 1 < len(ms) else 0))
            ln = len(pay) - value
            if ln == 0:
                value = A.NO_VALUE
        elif f == A.F_ASSIGN:
            value, ln = len(pay), len(op["value"])
            for k in op["value"]:
                _kafka_range(i, k, 0, 0)
            pay.extend(int(k) | (0x80000000 if op.get("seek-to-beginning?") else 0) for k in op["value"])
        rows[i] = (int(op.get("time", i * 1000)) | (ln << 48), typ | (f << 2) | (process << 12), value)
    return rows, np.asarray(pay if pay else [0], dtype=np.uint32)


def check_kafka_history(rows, payload):
    """The kafka checker (msim_check_kafka_rows) on one history -> dict."""
    lib = A.load()
    res = A.CheckResult()
    rows = np.ascontiguousarray(rows); payload = np.ascontiguousarray(payload, dtype=np.uint32)
    rc = lib.msim_check_kafka_rows(rows.ctypes.data_as(C.c_void_p), len(rows), payload.ctypes.data_as(C.c_void_p), len(payload), C.byref(res))
    if rc:
        raise EngineError(f"msim_check_kafka_rows: {rc}")
    return {"valid?": {1: True, 0: False, 2: "unknown"}[res.valid], "anomalies": sorted(n for b, n in A.KAFKA_ANOMALIES.items() if res.error_count & b),
            "send-count": res.attempt_count, "acked-count": res.stable_count, "lost-count": res.lost_count, "unobserved-count": res.never_read_count,
            "duplicate-count": res.duplicated_count, "ok-count": res.ok_count, "fail-count": res.fail_count, "info-count": res.info_count}


def decode_txn(words):
    """Payload words of one transaction -> [[f k v] ...] (txn_list_append.clj:27-39; encoding: include/maelsim.h msim_op)."""
    out, i, words = [], 0, [int(w) for w in words]
    while i < len(words):
        w = words[i]; i += 1
        key, x = (w >> 1) & 0x7FFF, (w >> 16) & 0xFF
        if w & 1:
            out.append([":append", key, x])
        elif x == 0xFF:
            out.append([":r", key, None])
        else:
            nw = (x + 3) // 4
            out.append([":r", key, [(words[i + e // 4] >> (8 * (e % 4))) & 0xFF for e in range(x)]])
            i += nw
    return out


def encode_txn(txn):
    """[[f k v] ...] -> payload words (inverse of decode_txn)."""
    words = []
    for f, k, v in txn:
        if f == ":append":
            words.append(1 | (k << 1) | (v << 16))
        elif v is None:
            words.append((k << 1) | (0xFF << 16))
        else:
            words.append((k << 1) | (len(v) << 16))
            for i in range(0, len(v), 4):
                words.append(sum(x << (8 * j) for j, x in enumerate(v[i:i + 4])))
    return words


def encode_txn_history(ops, rw=False):
    """Jepsen-shaped txn ops ({type, process, value[, time]}) -> (rows, payload) in the engine's binary layout, so that
    externally produced list-append (rw=True: rw-register) histories can be fed to msim_check_txn_rows / msim_check_rw_rows."""
    tkw = {v: k for k, v in TYPE_KW.items()}
    rows = np.zeros(len(ops), dtype=OP_DT)
    payload = []
    for i, op in enumerate(ops):
        w = encode_rw_txn(op["value"]) if rw else encode_txn(op["value"])
        rows["time_len"][i] = int(op.get("time", i * 1000)) | (len(w) << 48)
        rows["packed"][i] = tkw[op["type"]] | (A.F_TXN << 2) | (int(op["process"]) << 12)
        rows["value"][i] = len(payload)
        payload.extend(w)
    return rows, np.asarray(payload, dtype=np.uint32)


def check_txn_history(rows, payload):
    """The host list-append checker (msim_check_txn_rows) on one history -> dict with :valid? and the anomaly names."""
    lib = A.load()
    res = A.CheckResult()
    rows = np.ascontiguousarray(rows); payload = np.ascontiguousarray(payload, dtype=np.uint32)
    rc = lib.msim_check_txn_rows(rows.ctypes.data_as(C.c_void_p), len(rows), payload.ctypes.data_as(C.c_void_p), len(payload), C.byref(res))
    if rc:
        raise EngineError(f"msim_check_txn_rows: {rc}")
    return {"valid?": {1: True, 0: False, 2: "unknown"}[res.valid], "anomalies": sorted(n for b, n in A.ANOMALIES.items() if res.error_count & b),
            "txn-count": res.attempt_count, "ok-count": res.ok_count, "fail-count": res.fail_count, "info-count": res.info_count,
            "edge-count": res.lost_count, "cycle-txns": res.stale_count, "anomaly-bits": res.error_count}


def check_rw_history(rows, payload, consistency_model="strict-serializable"):
    """The host rw-register checker (msim_check_rw_rows) on one history -> dict like check_txn_history."""
    lib = A.load()
    res = A.CheckResult()
    rows = np.ascontiguousarray(rows); payload = np.ascontiguousarray(payload, dtype=np.uint32)
    rc = lib.msim_check_rw_rows(rows.ctypes.data_as(C.c_void_p), len(rows), payload.ctypes.data_as(C.c_void_p), len(payload),
                                CONSISTENCY_MODELS[consistency_model], C.byref(res))
    if rc:
        raise EngineError(f"msim_check_rw_rows: {rc}")
    return {"valid?": {1: True, 0: False, 2: "unknown"}[res.valid], "anomalies": sorted(n for b, n in A.ANOMALIES.items() if res.error_count & b),
            "txn-count": res.attempt_count, "ok-count": res.ok_count, "fail-count": res.fail_count, "info-count": res.info_count,
            "edge-count": res.lost_count, "cycle-txns": res.stale_count, "anomaly-bits": res.error_count}


def check_pn_history(rows):
    """The host pn-counter checker (msim_check_pn_rows) on one history -> the reference's result map (pn_counter.clj:120-123
    minus the op maps of :errors)."""
    lib = A.load()
    res = A.CheckResult()
    rows = np.ascontiguousarray(rows)
    ranges = (C.c_int64 * 512)()
    n = C.c_uint32()
    rc = lib.msim_check_pn_rows(rows.ctypes.data_as(C.c_void_p), len(rows), C.byref(res), ranges, 256, C.byref(n))
    if rc:
        raise EngineError(f"msim_check_pn_rows: {rc}")
    finals = [int(np.int32(np.uint32(v))) for v, p in zip(rows["value"], rows["packed"]) if (p >> 11) & 1 and (p & 3) == A.T_OK and (p >> 12) != A.PROCESS_NEMESIS]
    return {"valid?": bool(res.valid), "error-count": res.error_count, "final-reads": finals,
            "acceptable": [[ranges[2 * i], ranges[2 * i + 1]] for i in range(min(n.value, 256))]}


def encode_pn_history(ops):
    """Jepsen-shaped pn-counter ops ({type, f, value[, final?, process]}) -> rows, for msim_check_pn_rows."""
    tkw = {v: k for k, v in TYPE_KW.items()}
    rows = np.zeros(len(ops), dtype=OP_DT)
    for i, op in enumerate(ops):
        rows["time_len"][i] = i * 1000
        v = op.get("value")
        rows["value"][i] = 0xFFFFFFFF if v is None else (int(v) & 0xFFFFFFFF)
        f = A.F_ADD if op["f"] == ":add" else A.F_READ
        rows["packed"][i] = tkw[op["type"]] | (f << 2) | ((1 if op.get("final?") else 0) << 11) | (int(op.get("process", 0)) << 12)
    return rows


def encode_set_history(ops):
    """Jepsen-shaped broadcast / g-set ops ({type, process, f, value[, time, final?]}: :broadcast / :add carry the element, a read's :ok
    the list of elements) -> (rows, payload) in the engine's binary layout (include/maelsim.h msim_op: a read result is a bitmap in the
    payload area), so that externally produced histories can be fed to msim_check_set_full_batch.  Inverse of decode_history."""
    tkw = {v: k for k, v in TYPE_KW.items()}
    fkw = {":broadcast": A.F_BROADCAST, ":add": A.F_ADD, ":read": A.F_READ}
    rows = np.zeros(len(ops), dtype=OP_DT)
    pay = []
    for i, op in enumerate(ops):
        typ, f, process = tkw[op["type"]], fkw[op["f"]], int(op["process"])
        value, ln = A.NO_VALUE, 0
        if f != A.F_READ:
            value = int(op["value"])
            if not 0 <= value < 65536:
                raise EngineError(f"op {i}: element {value} outside the binary layout (0..65535)")
        elif typ == A.T_OK:
            els = [int(x) for x in op["value"]]
            if len(set(els)) != len(els):
                raise EngineError(f"op {i}: a read holding an element twice has no bitmap form")
            ln = (max(els) // 32 + 1) if els else 0
            value = len(pay)
            words = [0] * ln
            for x in els:
                words[x // 32] |= 1 << (x % 32)
            pay.extend(words)
        rows[i] = (int(op.get("time", i * 1000)) | (ln << 48), typ | (f << 2) | ((1 if op.get("final?") else 0) << 11) | (process << 12), value)
    return rows, np.asarray(pay if pay else [0], dtype=np.uint32)


def encode_lin_kv_history(ops):
    """Jepsen-shaped lin-kv ops ({type, process, f, value = [k v] / [k [v v']][, time]}; lin_kv.clj:53-67) -> rows for
    msim_check_lin_kv_rows / msim_check_lin_kv_batch.  Keys and values below 255 (nil = 255 in the binary layout)."""
    tkw = {v: k for k, v in TYPE_KW.items()}
    fkw = {":read": A.F_READ, ":write": A.F_WRITE, ":cas": A.F_CAS}
    rows = np.zeros(len(ops), dtype=OP_DT)
    for i, op in enumerate(ops):
        k, v = op["value"]
        v1, v2 = (v if op["f"] == ":cas" else (v, None))
        for x in (k, v1, v2):
            if x is not None and not 0 <= int(x) < 255:
                raise EngineError(f"op {i}: key / value {x} outside the binary layout (0..254)")
        value = int(k) | ((0xFF if v1 is None else int(v1)) << 8) | ((0xFF if v2 is None else int(v2)) << 16)
        rows[i] = (int(op.get("time", i * 1000)), tkw[op["type"]] | (fkw[op["f"]] << 2) | (int(op["process"]) << 12), value)
    return rows


def check_lin_kv_history(rows):
    """The per-key linearizability check (msim_check_lin_kv_rows, host search) on one history -> dict."""
    res = A.CheckResult()
    rows = np.ascontiguousarray(rows)
    rc = A.load().msim_check_lin_kv_rows(rows.ctypes.data_as(C.c_void_p), len(rows), C.byref(res))
    if rc:
        raise EngineError(f"msim_check_lin_kv_rows: {rc}")
    return {"valid?": {1: True, 0: False, 2: "unknown"}[res.valid], "key-count": res.attempt_count, "invalid-keys": res.error_count}


def decode_history(rows, payload, n_nodes, workload=A.WL_BROADCAST, node_program=None):
    """Binary rows -> list of Jepsen op maps (SURVEY.md §8b 'History surface').  `node_program` matters for unique-ids only (what an id is)."""
    ops = []
    for idx in range(len(rows)):
        tl, packed, value = int(rows["time_len"][idx]), int(rows["packed"][idx]), int(rows["value"][idx])
        typ, f, err = packed & 3, (packed >> 2) & 31, (packed >> 7) & 15
        final, process, ln = (packed >> 11) & 1, packed >> 12, tl >> 48
        op = {"index": idx, "time": tl & 0xFFFFFFFFFFFF, "type": TYPE_KW[typ], "f": F_KW.get(f, f),
              "process": ":nemesis" if process == A.PROCESS_NEMESIS else process}
        if workload == A.WL_LIN_KV and f in (A.F_READ, A.F_WRITE, A.F_CAS):  # independent tuples, lin_kv.clj:53-67
            k, v1, v2 = value & 0xFF, _nil((value >> 8) & 0xFF), _nil((value >> 16) & 0xFF)
            op["value"] = [k, [v1, v2]] if f == A.F_CAS else [k, v1]
        elif f == A.F_GENERATE and node_program == A.NODE_TSO_IDS:   # a lin-tso timestamp (service.clj:121-123)
            op["value"] = value if typ == A.T_OK else None
        elif f == A.F_GENERATE:   # flake id [time count node-id], flake_ids.clj:30-31
            op["value"] = [value >> 20, (value >> 5) & 0x7FFF, f"n{value & 31}"] if typ == A.T_OK else None
        elif f == A.F_SEND:   # [[:send k msg]] / [[:send k [offset msg]]], workload/kafka.clj:188-190
            k, msg, off = value & 63, (value >> 6) & 0x7FF, value >> 17
            op["value"] = [[":send", str(k), msg if off == 0x7FF else [off, msg]]]
        elif f == A.F_POLL:   # [[:poll]] / [[:poll {k [[offset msg] ...]}]], :171-186 (keys are strings, :247-283)
            op["value"] = [[":poll", decode_poll(payload[value:value + ln])]] if typ == A.T_OK else [[":poll"]]
            if typ != A.T_OK and ln:
                op["offsets"] = {str(int(w) & 7): int(w) >> 8 for w in payload[value:value + ln]}   # engine abstraction: what the client asked with
        elif f == A.F_ASSIGN:
            ws = [int(w) for w in payload[value:value + ln]]
            op["value"] = [str(w & 7) for w in ws]
            if ws and ws[0] >> 31:
                op["seek-to-beginning?"] = True
        elif f == A.F_CRASH:
            op["value"] = None
        elif f == A.F_TXN:
            op["value"] = (decode_rw_txn if workload == A.WL_TXN_RW_REGISTER else decode_txn)(payload[value:value + ln])
        elif workload in (A.WL_PN_COUNTER, A.WL_G_COUNTER) and f in (A.F_ADD, A.F_READ):  # pn_counter.clj:22-58: signed delta / counter value
            signed = value - (1 << 32) if value & 0x80000000 else value
            op["value"] = signed if (f == A.F_ADD or typ == A.T_OK) else None
        elif f == A.F_READ:
            op["value"] = bitmap_to_list(payload[value:value + ln]) if typ == A.T_OK else None
        elif f == A.F_ECHO:
            v = f"Please echo {value}"
            op["value"] = {"type": "echo_ok", "echo": v} if typ == A.T_OK else v
        elif f == A.F_START_PARTITION:
            if ln:
                g = payload[value:value + ln].reshape(n_nodes, A.MASK_WORDS)
                op["value"] = [":isolated", {f"n{d}": [f"n{s}" for s in bitmap_to_list(g[d])] for d in range(n_nodes) if g[d].any()}]
            else:
                op["value"] = SPEC_KW[value]
        elif f == A.F_STOP_PARTITION:
            op["value"] = None if idx == 0 or int(rows["packed"][idx - 1]) != packed or False else ":network-healed"
        else:
            op["value"] = None if value == A.NO_VALUE else value
        if err in ERR_KW:
            op["error"] = ERR_KW[err]
        if final:
            op["final?"] = True
        ops.append(op)
    return ops


def _availability_mode(availability):
    if availability is None:
        return A.AVAIL_NIL, 0.0
    if availability in ("total", ":total"):
        return A.AVAIL_TOTAL, 0.0
    if isinstance(availability, (int, float)) and not isinstance(availability, bool):
        return A.AVAIL_FRACTION, float(availability)
    raise EngineError(f"Don't know how to handle :availability {availability!r}")   # checker.clj:36-39


def check_availability_rows(rows, availability=None):
    """The availability checker for one history on the host (msim_check_availability_rows)."""
    mode, a = _availability_mode(availability)
    r = A.Availability()
    rows = np.ascontiguousarray(rows)
    rc = A.load().msim_check_availability_rows(rows.ctypes.data, len(rows), mode, a, C.byref(r))
    if rc:
        raise EngineError(f"msim_check_availability_rows: {rc}")
    return {"valid?": bool(r.valid), "ok-fraction": float(r.ok_fraction), "ok-count": r.ok_count, "invoke-count": r.invoke_count}


def check_lin_kv_batch(histories, device=0):
    """lin-kv: per-key linearizability of several histories (each an array of rows) with the device search behind Engine.check()
    (msim_check_lin_kv_batch).  Returns the CHECK_DT records, one per history."""
    hs = [np.ascontiguousarray(h) for h in histories]
    off = np.zeros(len(hs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(h) for h in hs])
    rows = np.concatenate(hs) if hs else np.zeros(0, dtype=hs[0].dtype if hs else np.uint8)
    out = np.zeros(len(hs), dtype=CHECK_DT)
    rc = A.load().msim_check_lin_kv_batch(device, rows.ctypes.data, off.ctypes.data, len(hs), out.ctypes.data)
    if rc:
        raise EngineError(f"msim_check_lin_kv_batch: {rc}")
    return out


def check_unique_batch(histories, device=0, fn="msim_check_unique_batch"):
    """unique-ids (or, with fn="msim_check_pn_batch", pn-counter / g-counter): several histories (arrays of rows) through the device
    checker behind Engine.check()."""
    hs = [np.ascontiguousarray(h) for h in histories]
    mr = max(1, max(len(h) for h in hs))
    slab = np.zeros((len(hs), mr), dtype=OP_DT)
    for i, h in enumerate(hs):
        slab[i, :len(h)] = h
    nr = np.asarray([len(h) for h in hs], dtype=np.uint32)
    out = np.zeros(len(hs), dtype=CHECK_DT)
    rc = getattr(A.load(), fn)(device, slab.ctypes.data, nr.ctypes.data, mr, len(hs), out.ctypes.data)
    if rc:
        raise EngineError(f"{fn}: {rc}")
    return out


def check_set_full_batch(histories, concurrency, workload=A.WL_BROADCAST, max_values=None, device=0):
    """broadcast / g-set (set-full) or echo: several histories — (rows, payload words) pairs — through the device checker behind
    Engine.check() (msim_check_set_full_batch)."""
    hs = [(np.ascontiguousarray(r), np.ascontiguousarray(p, dtype=np.uint32)) for r, p in histories]
    mr = max(1, max(len(r) for r, _ in hs))
    mp = max(1, max(len(p) for _, p in hs))
    rows = np.zeros((len(hs), mr), dtype=OP_DT)
    pay = np.zeros((len(hs), mp), dtype=np.uint32)
    for i, (r, p) in enumerate(hs):
        rows[i, :len(r)] = r
        pay[i, :len(p)] = p
    nr = np.asarray([len(r) for r, _ in hs], dtype=np.uint32)
    if max_values is None:   # every add / broadcast invocation creates one element
        f = (rows["packed"] >> 2) & 31
        adds = (((f == A.F_ADD) | (f == A.F_BROADCAST)) & ((rows["packed"] & 3) == 0)).sum(axis=1).max() if len(hs) else 0
        max_values = max(32, (int(adds) + 31) // 32 * 32)
    out = np.zeros(len(hs), dtype=CHECK_DT)
    rc = A.load().msim_check_set_full_batch(device, workload, concurrency, rows.ctypes.data, nr.ctypes.data, mr, pay.ctypes.data, mp, max_values, len(hs), out.ctypes.data)
    if rc:
        raise EngineError(f"msim_check_set_full_batch: {rc}")
    return out


def check_pn_batch(histories, device=0):
    """pn-counter / g-counter: several histories through the device checker behind Engine.check() (msim_check_pn_batch)."""
    return check_unique_batch(histories, device, fn="msim_check_pn_batch")


def check_txn_batch(histories, device=0):
    """txn-list-append: several histories, each (rows, payload), through the device pass behind Engine.check() with the host analysis
    for what it cannot prove clean (msim_check_txn_batch).  Returns the CHECK_DT records, one per history."""
    rs = [np.ascontiguousarray(h[0]) for h in histories]
    ps = [np.ascontiguousarray(h[1], dtype=np.uint32) for h in histories]
    ro = np.zeros(len(rs) + 1, dtype=np.uint64); ro[1:] = np.cumsum([len(x) for x in rs])
    po = np.zeros(len(ps) + 1, dtype=np.uint64); po[1:] = np.cumsum([len(x) for x in ps])
    rows = np.concatenate(rs) if rs else np.zeros(0, dtype=OP_DT)
    pay = np.concatenate(ps) if ps else np.zeros(0, dtype=np.uint32)
    if len(pay) == 0:
        pay = np.zeros(1, dtype=np.uint32)
    out = np.zeros(len(rs), dtype=CHECK_DT)
    rc = A.load().msim_check_txn_batch(device, rows.ctypes.data, ro.ctypes.data, pay.ctypes.data, po.ctypes.data, len(rs), out.ctypes.data)
    if rc:
        raise EngineError(f"msim_check_txn_batch: {rc}")
    return out


def check_rw_batch(histories, consistency_model="read-committed", device=0):
    """txn-rw-register: several histories, each (rows, payload), through the device pass behind Engine.check() with the host analysis for
    what it cannot prove valid (msim_check_rw_batch).  Returns (CHECK_DT records, how many histories went to the host)."""
    rs = [np.ascontiguousarray(h[0]) for h in histories]
    ps = [np.ascontiguousarray(h[1], dtype=np.uint32) for h in histories]
    ro = np.zeros(len(rs) + 1, dtype=np.uint64); ro[1:] = np.cumsum([len(x) for x in rs])
    po = np.zeros(len(ps) + 1, dtype=np.uint64); po[1:] = np.cumsum([len(x) for x in ps])
    rows = np.concatenate(rs) if rs else np.zeros(0, dtype=OP_DT)
    pay = np.concatenate(ps) if ps else np.zeros(0, dtype=np.uint32)
    if len(pay) == 0:
        pay = np.zeros(1, dtype=np.uint32)
    out = np.zeros(len(rs), dtype=CHECK_DT)
    n_host = C.c_uint32()
    rc = A.load().msim_check_rw_batch(device, rows.ctypes.data, ro.ctypes.data, pay.ctypes.data, po.ctypes.data, len(rs),
                                      CONSISTENCY_MODELS[consistency_model], out.ctypes.data, C.byref(n_host))
    if rc:
        raise EngineError(f"msim_check_rw_batch: {rc}")
    return out, n_host.value



def check_kafka_batch(histories, concurrency, device=0):
    """kafka: several histories, each (rows, payload), through the device pass behind Engine.check() (csrc/kafka_check_dev.hip) with the
    host checker for what it cannot prove clean (msim_check_kafka_batch); `concurrency` = the worker threads of the test.  Returns
    (CHECK_DT records, how many histories went to the host)."""
    rs = [np.ascontiguousarray(h[0]) for h in histories]
    ps = [np.ascontiguousarray(h[1], dtype=np.uint32) for h in histories]
    ro = np.zeros(len(rs) + 1, dtype=np.uint64); ro[1:] = np.cumsum([len(x) for x in rs])
    po = np.zeros(len(ps) + 1, dtype=np.uint64); po[1:] = np.cumsum([len(x) for x in ps])
    rows = np.concatenate(rs) if rs else np.zeros(0, dtype=OP_DT)
    pay = np.concatenate(ps) if ps else np.zeros(0, dtype=np.uint32)
    if len(pay) == 0:
        pay = np.zeros(1, dtype=np.uint32)
    out = np.zeros(len(rs), dtype=CHECK_DT)
    n_host = C.c_uint32()
    rc = A.load().msim_check_kafka_batch(device, rows.ctypes.data, ro.ctypes.data, pay.ctypes.data, po.ctypes.data, len(rs), concurrency,
                                         out.ctypes.data, C.byref(n_host))
    if rc:
        raise EngineError(f"msim_check_kafka_batch: {rc}")
    return out, n_host.value


def journal_fressian(cfg, events, payload):
    """One instance's net journal as the bytes of a net-journal/<stripe>.fressian file (msim_journal_fressian_rows, csrc/fressian.cpp):
    what maelstrom.net.journal writes (journal.clj:55-141) and maelstrom.net.checker / net.viz read."""
    lib = A.load()
    ev = np.ascontiguousarray(events)
    pay = np.ascontiguousarray(payload, dtype=np.uint32)
    args = (C.byref(cfg), ev.ctypes.data, len(ev), pay.ctypes.data, len(pay))
    need = C.c_size_t()
    rc = lib.msim_journal_fressian_rows(*args, None, 0, C.byref(need))
    if rc:
        raise EngineError(f"msim_journal_fressian_rows: {rc}")
    buf = (C.c_ubyte * max(need.value, 1))()
    rc = lib.msim_journal_fressian_rows(*args, buf, need.value, None)
    if rc:
        raise EngineError(f"msim_journal_fressian_rows: {rc}")
    return bytes(buf[: need.value])


def history_edn_native(cfg, rows, payload):
    """history.edn text of one history straight from the binary rows (msim_history_edn_rows, csrc/edn.cpp)."""
    lib = A.load()
    rows = np.ascontiguousarray(rows); payload = np.ascontiguousarray(payload, dtype=np.uint32)
    need = C.c_size_t()
    args = (C.byref(cfg), rows.ctypes.data_as(C.c_void_p), len(rows), payload.ctypes.data_as(C.c_void_p), len(payload))
    rc = lib.msim_history_edn_rows(*args, None, 0, C.byref(need))
    if rc:
        raise EngineError(f"msim_history_edn_rows: {rc}")
    buf = C.create_string_buffer(need.value)
    rc = lib.msim_history_edn_rows(*args, buf, need.value, None)
    if rc:
        raise EngineError(f"msim_history_edn_rows: {rc}")
    return buf.value.decode()


def history_edn(ops):
    """Jepsen history.edn text (one op map per line)."""
    def edn(v):
        if v is None:
            return "nil"
        if v is True:
            return "true"
        if isinstance(v, str):
            return v if v.startswith(":") else '"' + v + '"'
        if isinstance(v, dict):
            return "{" + ", ".join(f"{edn(k if str(k).startswith(':') else ':' + str(k)) if not (str(k).startswith('n') or str(k).isdigit()) else edn(str(k))} {edn(x)}" for k, x in v.items()) + "}"
        if isinstance(v, (list, tuple)):
            return "[" + " ".join(edn(x) for x in v) + "]"
        return str(v)
    lines = []
    for op in ops:
        keys = ["type", "f", "value", "time", "process", "index"] + [k for k in ("error", "final?", "seek-to-beginning?") if k in op]
        lines.append("{" + ", ".join(f":{k} {edn(op[k])}" for k in keys) + "}")
    return "\n".join(lines) + "\n"
