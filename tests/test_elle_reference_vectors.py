"""REFERENCE-HELD vectors for the list-append checker: the transactions the reference's documentation prints with the anomalies the
real Elle found in them (doc/05-datomic/01-single-node.md:117-141,260-287,354-366; 02-shared-state.md:198,226-233;
04-optimization.md:22-56,311-345), as minimal histories in tests/golden/elle_doc_vectors.json (tests/golden/make_elle_doc_vectors.py
wrote it and checked every quoted fragment against the doc files).  Together with pn_counter_test.clj these are all the checker
results the reference tree holds; everything else about the transactional checkers rests on restatements.

CPU: msim_check_txn_rows (the host analysis) and the independent Python restatement must report what the docs report.
GPU (`-m gpu`): the device pass behind msim_check (txn_check_dev.hip -> msim_check_txn_batch) must give the host's result for them."""
import json
import os

import pytest

from maelstrom_amd import _abi as A
from maelstrom_amd import engine as E

REFERENCE_HELD = True   # marker: these expectations come from the reference tree, not from this repository's own restatements
HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "elle_doc_vectors.json")) as f:
    VECTORS = json.load(f)["vectors"]
CYCLES = {"G0", "G1c", "G-single", "G2"}


def _violates(anomaly_bits, model):
    lib = A.load()
    lib.msim_violated_anomalies.restype = A.C.c_uint32
    return lib.msim_violated_anomalies(A.C.c_uint32(anomaly_bits), A.C.c_uint32(E.CONSISTENCY_MODELS[model]))


def _bits(names):
    inv = {n: b for b, n in A.ANOMALIES.items()}
    out = 0
    for n in names:
        out |= inv[n]
    return out


@pytest.mark.parametrize("v", VECTORS, ids=[v["doc"] for v in VECTORS])
def test_host_checker_reports_what_the_reference_docs_report(v):
    rows, pay = E.encode_txn_history(v["history"])
    got = E.check_txn_history(rows, pay)
    assert got["valid?"] is v["valid"], (v["name"], got)
    for a in v["has"]:
        assert a in got["anomalies"], (v["name"], a, got)
    for a in v.get("has_not", []):
        assert a not in got["anomalies"], (v["name"], a, got)
    if v["exact"]:
        assert sorted(got["anomalies"]) == sorted(v["has"]), (v["name"], got)
    # --consistency-models: what the doc's :not / :also-not sets (or a passing re-run under another model) say
    bits = _bits(got["anomalies"])
    assert _violates(bits, "strict-serializable") != 0
    for m in v.get("valid_under", []):
        assert _violates(bits, m) == 0, (v["name"], m, got)
    if "internal" in v["has"]:   # :not #{:read-atomic ...}: internal anomalies are proscribed from snapshot isolation upwards
        assert _violates(bits, "snapshot-isolation") != 0 and _violates(_bits(["internal"]), "read-committed") == 0


@pytest.mark.parametrize("v", VECTORS, ids=[v["doc"] for v in VECTORS])
def test_python_restatement_reports_the_same(v):
    import elle_ref
    ref = elle_ref.analyse(v["history"])
    assert ref["valid?"] is v["valid"]
    for a in v["has"]:
        assert ("cycle" if a in CYCLES else a) in ref["anomalies"], (v["name"], a, ref)
    for a in v.get("has_not", []):
        assert a not in ref["anomalies"]


def test_incompatible_orders_name_the_docs_keys():
    """doc/05-datomic/01-single-node.md:358-366 lists the keys (7, 9, 10, 8) with the two irreconcilable values of each: every one of
    them alone is an incompatible order, and a history with the two reads of just one key is flagged for that key."""
    v = [x for x in VECTORS if "incompatible_keys" in x][0]
    for k in v["incompatible_keys"]:
        sub = [o for o in v["history"] if all(m[1] == k for m in o["value"])]
        got = E.check_txn_history(*E.encode_txn_history(sub))
        assert "incompatible-order" in got["anomalies"] and got["valid?"] is False, (k, got)


@pytest.mark.gpu
def test_device_pass_gives_the_hosts_result_on_the_doc_vectors(lib):
    import numpy as np
    import ctypes as C
    hs = [E.encode_txn_history(v["history"]) for v in VECTORS]
    dev = E.check_txn_batch(hs)
    for v, (rows, pay), d in zip(VECTORS, hs, dev):
        res = A.CheckResult()
        rows = np.ascontiguousarray(rows); pay = np.ascontiguousarray(pay, dtype=np.uint32)
        assert A.load().msim_check_txn_rows(rows.ctypes.data_as(C.c_void_p), len(rows), pay.ctypes.data_as(C.c_void_p), len(pay), C.byref(res)) == 0
        for f in ("valid", "attempt_count", "stable_count", "lost_count", "stale_count", "error_count", "op_count", "ok_count", "fail_count", "info_count"):
            assert int(d[f]) == int(getattr(res, f)), (v["name"], f, int(d[f]), int(getattr(res, f)))
        assert int(d["valid"]) == 0
