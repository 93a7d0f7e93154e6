"""Test-side loader for the CPU oracle (oracle/libmaelsim_oracle.so).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

from maelstrom_amd import _abi as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libmaelsim_oracle.so")

OP_DT = np.dtype([("time_len", "<u8"), ("packed", "<u4"), ("value", "<u4")])
STATS_DT = np.dtype([(n, "<u8") for n in ("all_send", "all_recv", "clients_send", "clients_recv", "servers_send", "servers_recv")])
META_DT = np.dtype([(n, "<u4") for n in ("n_rows", "n_payload_words", "flags", "n_rounds", "n_events", "r0", "r1", "r2")])
EVENT_DT = np.dtype([(n, "<u4") for n in ("time_us", "msg", "a", "route")])

_lib = None


def load():
    global _lib
    if _lib is None:
        src = os.path.join(ORACLE_DIR, "maelsim_oracle.c")
        if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(src):
            subprocess.run(["make", "-C", ORACLE_DIR], check=True, capture_output=True)
        lib = C.CDLL(ORACLE_SO)
        lib.oracle_run.argtypes = [C.POINTER(A.Config), C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.oracle_run.restype = C.c_int
        lib.oracle_neg_ln_q16.argtypes = [C.c_uint32]
        lib.oracle_neg_ln_q16.restype = C.c_uint32
        lib.oracle_draw32.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64]
        lib.oracle_draw32.restype = C.c_uint32
        lib.oracle_topology.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p]
        lib.oracle_topology.restype = C.c_int
        _lib = lib
    return _lib


class OracleRun:
    """Outputs of oracle_run in the engine's slab layout."""

    def __init__(self, cfg, first, n):
        self.cfg, self.first, self.n = cfg, first, n
        self.rows = np.zeros((n, cfg.max_rows), dtype=OP_DT)
        self.payload = np.zeros((n, cfg.max_payload_words), dtype=np.uint32)
        self.stats = np.zeros(n, dtype=STATS_DT)
        self.meta = np.zeros(n, dtype=META_DT)
        self.journal = np.zeros((n, max(cfg.journal_capacity, 1)), dtype=EVENT_DT)

    def history(self, i):
        m = self.meta[i]
        return self.rows[i, : m["n_rows"]], self.payload[i, : m["n_payload_words"]]

    def events(self, i):
        return self.journal[i, : min(int(self.meta[i]["n_events"]), self.cfg.journal_capacity)]


def run(cfg, first=0, n=1):
    lib = load()
    out = OracleRun(cfg, first, n)
    rc = lib.oracle_run(C.byref(cfg), first, n, out.rows.ctypes.data, out.payload.ctypes.data,
                        out.stats.ctypes.data, out.meta.ctypes.data, out.journal.ctypes.data if cfg.journal_capacity else None)
    if rc != 0:
        raise RuntimeError(f"oracle_run failed: {rc}")
    return out
