"""ctypes mirror of include/maelsim.h — the C-ABI boundary of libmaelsim.so.

This is the binding a reference maintainer would write on the JVM side as JNI (INTEGRATION.md shows that
stub); here it is ctypes because the host language available in this image is Python.  Nothing in this
module touches the oracle: if libmaelsim.so is missing or there is no HIP device the engine raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MSIM_LIB") or os.path.join(HERE, "libmaelsim.so")   # MSIM_LIB: developer knob for A/B builds

ABI_VERSION = 1

# enums (include/maelsim.h)
OK, E_INVALID, E_NO_DEVICE, E_HIP, E_NOMEM, E_RANGE, E_UNSUPPORTED, E_OVERFLOW = 0, -1, -2, -3, -4, -5, -6, -7
WL_ECHO, WL_BROADCAST, WL_G_SET, WL_LIN_KV, WL_TXN_LIST_APPEND, WL_PN_COUNTER, WL_G_COUNTER, WL_UNIQUE_IDS, WL_TXN_RW_REGISTER, WL_KAFKA = range(10)
NODE_ECHO, NODE_BCAST_FF, NODE_BCAST_FF_ECHOBACK, NODE_BCAST_ACK_RETRY, NODE_BCAST_RPC_ALL, NODE_G_SET, NODE_RAFT, NODE_TXN_SINGLE_KEY, NODE_PN_COUNTER, NODE_FLAKE_IDS, NODE_LIN_KV_PROXY, NODE_TXN_RW_HAT, NODE_TXN_MULTI_KEY, NODE_TSO_IDS, NODE_KAFKA, NODE_TXN_DATOMIC = range(16)
SVC_LIN_KV, SVC_SEQ_KV, SVC_LWW_KV = range(3)
CM_STRICT_SERIALIZABLE, CM_SERIALIZABLE, CM_SNAPSHOT_ISOLATION, CM_READ_COMMITTED, CM_READ_UNCOMMITTED = range(5)
LAT_CONSTANT, LAT_UNIFORM, LAT_EXPONENTIAL = range(3)
TOPO_GRID, TOPO_LINE, TOPO_TOTAL, TOPO_TREE2, TOPO_TREE3, TOPO_TREE4 = range(6)
NEMESIS_PARTITION = 1
T_INVOKE, T_OK, T_FAIL, T_INFO = range(4)
F_ECHO, F_BROADCAST, F_READ, F_ADD, F_START_PARTITION, F_STOP_PARTITION, F_WRITE, F_CAS, F_TXN, F_GENERATE, F_SEND, F_POLL, F_ASSIGN, F_CRASH = range(14)
ERR_NONE, ERR_NET_TIMEOUT, ERR_RPC, ERR_TEMPORARILY_UNAVAILABLE, ERR_KEY_DOES_NOT_EXIST, ERR_PRECONDITION_FAILED, ERR_TXN_CONFLICT, ERR_TIMEOUT, ERR_ABORT = range(9)
SPEC_ONE, SPEC_MAJORITY, SPEC_MAJORITIES_RING, SPEC_MINORITY_THIRD = range(4)
PROCESS_NEMESIS = 0xFFFFF
NO_VALUE = 0xFFFFFFFF
FLAG_ROWS_OVERFLOW, FLAG_PAYLOAD_OVERFLOW, FLAG_INBOX_OVERFLOW, FLAG_VALUES_OVERFLOW, FLAG_ROUND_LIMIT, FLAG_JOURNAL_OVERFLOW, FLAG_ARENA_OVERRUN = 1, 2, 4, 8, 16, 32, 64
MSG_TYPES = ["", "init", "init_ok", "topology", "topology_ok", "echo", "echo_ok", "broadcast", "broadcast_ok", "read", "read_ok",
             "add", "add_ok", "replicate", "write", "write_ok", "cas", "cas_ok", "error", "request_vote", "request_vote_res",
             "append_entries", "append_entries_res", "txn", "txn_ok", "generate", "generate_ok", "replicate_ack", "ts", "ts_ok",
             "send", "send_ok", "poll", "poll_ok", "list_committed_offsets", "list_committed_offsets_ok", "commit_offsets", "commit_offsets_ok"]
KAFKA_ANOMALIES = {1: "lost-write", 2: "nonmonotonic-poll", 4: "nonmonotonic-send", 8: "poll-skip", 16: "int-nonmonotonic-poll", 32: "int-poll-skip",
                   64: "inconsistent-offsets", 128: "duplicate", 256: "aborted-read", 512: "malformed"}
ANOMALIES = {1: "G0", 2: "G1a", 4: "G1b", 8: "G1c", 16: "G-single", 32: "G2", 64: "internal", 128: "duplicate-elements",
             256: "incompatible-order", 512: "realtime", 1024: "dirty-update", 2048: "cyclic-versions"}
MASK_WORDS = 4

EXPORTS = [
    "msim_abi_version", "msim_device_count", "msim_config_defaults", "msim_config_finalize", "msim_create",
    "msim_run", "msim_run_async", "msim_check", "msim_check_host_rechecks", "msim_set_dev_flags", "msim_check_lin_kv_batch", "msim_check_txn_batch", "msim_check_unique_batch", "msim_check_pn_batch", "msim_check_set_full_batch", "msim_check_rw_batch", "msim_check_kafka_rows", "msim_check_kafka_batch", "msim_check_lin_kv_rows", "msim_check_txn_rows", "msim_check_rw_rows", "msim_proscribed_anomalies", "msim_violated_anomalies", "msim_check_pn_rows", "msim_check_unique_rows", "msim_history_edn_rows", "msim_fetch", "msim_fetch_begin", "msim_history", "msim_net_stats_get", "msim_journal", "msim_meta",
    "msim_check_results", "msim_device_buffers_get", "msim_last_kernel_ms", "msim_get_config",
    "msim_selftest_wave", "msim_last_error", "msim_destroy",
    "msim_comm_unique_id", "msim_comm_init", "msim_gather", "msim_journal_fressian_rows",
    "msim_check_availability", "msim_check_availability_rows",
]
AVAIL_NIL, AVAIL_TOTAL, AVAIL_FRACTION = range(3)
COMM_ID_BYTES = 128


class Config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("abi_version", C.c_uint32), ("workload", C.c_uint32),
        ("node_program", C.c_uint32), ("n_nodes", C.c_uint32), ("concurrency", C.c_uint32),
        ("rate_mhz", C.c_uint32), ("time_limit_ms", C.c_uint32), ("latency_mean_ms", C.c_uint32),
        ("latency_dist", C.c_uint32), ("p_loss_q32", C.c_uint32), ("topology", C.c_uint32),
        ("nemesis_mask", C.c_uint32), ("nemesis_interval_ms", C.c_uint32), ("client_timeout_ms", C.c_uint32),
        ("quiesce_ms", C.c_uint32), ("seed", C.c_uint64), ("max_values", C.c_uint32), ("max_rows", C.c_uint32),
        ("max_payload_words", C.c_uint32), ("inbox_capacity", C.c_uint32), ("spill_capacity", C.c_uint32),
        ("journal_capacity", C.c_uint32), ("key_count", C.c_uint32), ("max_txn_length", C.c_uint32),
        ("max_writes_per_key", C.c_uint32), ("proxy_service", C.c_uint32), ("consistency_model", C.c_uint32), ("replication_words", C.c_uint32),
    ]


class Op(C.Structure):
    _fields_ = [("time_len", C.c_uint64), ("packed", C.c_uint32), ("value", C.c_uint32)]


class NetStats(C.Structure):
    _fields_ = [("all_send", C.c_uint64), ("all_recv", C.c_uint64), ("clients_send", C.c_uint64),
                ("clients_recv", C.c_uint64), ("servers_send", C.c_uint64), ("servers_recv", C.c_uint64)]


class InstMeta(C.Structure):
    _fields_ = [("n_rows", C.c_uint32), ("n_payload_words", C.c_uint32), ("flags", C.c_uint32), ("n_rounds", C.c_uint32),
                ("n_events", C.c_uint32), ("reserved", C.c_uint32 * 3)]


class Event(C.Structure):
    _fields_ = [("time_us", C.c_uint32), ("msg", C.c_uint32), ("a", C.c_uint32), ("route", C.c_uint32)]


class CheckResult(C.Structure):
    _fields_ = [("valid", C.c_uint32), ("attempt_count", C.c_uint32), ("stable_count", C.c_uint32),
                ("lost_count", C.c_uint32), ("never_read_count", C.c_uint32), ("stale_count", C.c_uint32),
                ("duplicated_count", C.c_uint32), ("error_count", C.c_uint32), ("stable_latency_ms", C.c_uint32 * 5),
                ("op_count", C.c_uint32), ("ok_count", C.c_uint32), ("fail_count", C.c_uint32), ("info_count", C.c_uint32)]


class DeviceBuffers(C.Structure):
    _fields_ = [("rows", C.c_void_p), ("payload", C.c_void_p), ("stats", C.c_void_p), ("meta", C.c_void_p),
                ("check", C.c_void_p), ("journal", C.c_void_p),
                ("rows_bytes", C.c_uint64), ("payload_bytes", C.c_uint64), ("stats_bytes", C.c_uint64),
                ("meta_bytes", C.c_uint64), ("check_bytes", C.c_uint64), ("journal_bytes", C.c_uint64),
                ("n_instances", C.c_uint32), ("max_rows", C.c_uint32),
                ("max_payload_words", C.c_uint32), ("journal_capacity", C.c_uint32)]


class Availability(C.Structure):
    _fields_ = [("valid", C.c_uint32), ("ok_fraction", C.c_float), ("ok_count", C.c_uint32), ("invoke_count", C.c_uint32)]


class Gathered(C.Structure):
    _fields_ = [("rows", C.c_void_p), ("payload", C.c_void_p), ("meta", C.c_void_p), ("stats", C.c_void_p),
                ("rows_bytes", C.c_uint64), ("payload_bytes", C.c_uint64), ("meta_bytes", C.c_uint64), ("stats_bytes", C.c_uint64),
                ("bytes_received", C.c_uint64), ("n_instances", C.c_uint32), ("world", C.c_uint32), ("rank", C.c_uint32), ("ms", C.c_float)]


assert C.sizeof(Config) == 120, C.sizeof(Config)
assert C.sizeof(Op) == 16 and C.sizeof(NetStats) == 48 and C.sizeof(InstMeta) == 32 and C.sizeof(CheckResult) == 68 and C.sizeof(Event) == 16

_lib = None


def load():
    """Loads libmaelsim.so (built in-tree by maelstrom_amd.build / __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `python -m maelstrom_amd.build` (hipcc, gfx950). "
                           "There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    P = C.POINTER
    lib.msim_abi_version.restype = C.c_uint32
    lib.msim_device_count.restype = C.c_int
    lib.msim_config_defaults.argtypes = [P(Config), C.c_uint32, C.c_uint32]
    lib.msim_config_finalize.argtypes = [P(Config), C.c_char_p, C.c_size_t]
    lib.msim_create.argtypes = [P(Config), C.c_int, P(C.c_void_p), C.c_char_p, C.c_size_t]
    lib.msim_run.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32]
    lib.msim_run_async.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]
    lib.msim_check.argtypes = [C.c_void_p]
    lib.msim_fetch.argtypes = [C.c_void_p]
    lib.msim_fetch_begin.argtypes = [C.c_void_p]
    lib.msim_fetch_begin.restype = C.c_int
    lib.msim_check_lin_kv_rows.argtypes = [C.c_void_p, C.c_uint32, P(CheckResult)]
    lib.msim_check_lin_kv_rows.restype = C.c_int
    lib.msim_set_dev_flags.argtypes = [C.c_void_p, C.c_uint32]
    lib.msim_set_dev_flags.restype = C.c_int
    lib.msim_check_host_rechecks.argtypes = [C.c_void_p]
    lib.msim_check_host_rechecks.restype = C.c_uint32
    lib.msim_check_lin_kv_batch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    lib.msim_check_lin_kv_batch.restype = C.c_int
    lib.msim_check_txn_batch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    lib.msim_check_txn_batch.restype = C.c_int
    lib.msim_check_unique_batch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    lib.msim_check_unique_batch.restype = C.c_int
    lib.msim_check_pn_batch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    lib.msim_check_pn_batch.restype = C.c_int
    lib.msim_check_set_full_batch.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    lib.msim_check_set_full_batch.restype = C.c_int
    lib.msim_check_rw_batch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    lib.msim_check_rw_batch.restype = C.c_int
    lib.msim_check_kafka_batch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    lib.msim_check_kafka_batch.restype = C.c_int
    lib.msim_check_kafka_rows.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, P(CheckResult)]
    lib.msim_check_kafka_rows.restype = C.c_int
    lib.msim_history_edn_rows.argtypes = [P(Config), C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_char_p, C.c_size_t, P(C.c_size_t)]
    lib.msim_history_edn_rows.restype = C.c_int
    lib.msim_history.argtypes = [C.c_void_p, C.c_uint32, P(P(Op)), P(C.c_uint32), P(P(C.c_uint32)), P(C.c_uint32)]
    lib.msim_net_stats_get.argtypes = [C.c_void_p, C.c_uint32, P(NetStats)]
    lib.msim_meta.argtypes = [C.c_void_p, C.c_uint32, P(InstMeta)]
    lib.msim_journal.argtypes = [C.c_void_p, C.c_uint32, P(P(Event)), P(C.c_uint32)]
    lib.msim_check_results.argtypes = [C.c_void_p, P(P(CheckResult)), P(C.c_uint32)]
    lib.msim_device_buffers_get.argtypes = [C.c_void_p, P(DeviceBuffers)]
    lib.msim_last_kernel_ms.argtypes = [C.c_void_p, P(C.c_float), P(C.c_float)]
    lib.msim_get_config.argtypes = [C.c_void_p, P(Config)]
    lib.msim_last_error.argtypes = [C.c_void_p]
    lib.msim_last_error.restype = C.c_char_p
    lib.msim_selftest_wave.argtypes = [C.c_int]
    lib.msim_selftest_wave.restype = C.c_int
    lib.msim_destroy.argtypes = [C.c_void_p]
    lib.msim_destroy.restype = None
    lib.msim_journal_fressian_rows.argtypes = [P(Config), C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, P(C.c_size_t)]
    lib.msim_journal_fressian_rows.restype = C.c_int
    lib.msim_check_availability.argtypes = [C.c_void_p, C.c_uint32, C.c_double, P(Availability), C.c_uint32]
    lib.msim_check_availability_rows.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_double, P(Availability)]
    lib.msim_check_availability.restype = lib.msim_check_availability_rows.restype = C.c_int
    lib.msim_comm_unique_id.argtypes = [C.c_char_p]
    lib.msim_comm_init.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int]
    lib.msim_gather.argtypes = [C.c_void_p, C.c_int, P(Gathered)]
    for name in ("msim_comm_unique_id", "msim_comm_init", "msim_gather"):
        getattr(lib, name).restype = C.c_int
    for name in ("msim_config_defaults", "msim_config_finalize", "msim_create", "msim_run", "msim_run_async",
                 "msim_check", "msim_fetch", "msim_fetch_begin", "msim_history", "msim_net_stats_get", "msim_journal", "msim_meta", "msim_check_results",
                 "msim_device_buffers_get", "msim_last_kernel_ms", "msim_get_config"):
        getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib
