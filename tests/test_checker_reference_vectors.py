"""REFERENCE-HELD vectors for the two checkers the BASELINE metrics rest on: set-full (broadcast / g-set: `histories_per_sec` of the
headline) and the per-key linearizability search (lin-kv).  [upstream] jepsen.checker/set-full and Knossos are not vendored, but the
reference's tutorial prints runs of the real checkers: the closing reads of a broadcast run in their invocation / completion order with
the result map (doc/03-broadcast/01-broadcast.md:388-430: 21 messages, stable 1, stale (8), lost 20 with the list;
02-performance.md:282-301: lost (0 1 2 17 27 30 37 39), stable 37, stale 32), a linearizable single-node run with its final register
value and a write of 2 followed by a read of 4 that Knossos rejects (doc/06-raft/01-key-value.md:131-195).
tests/golden/checker_doc_vectors.json holds them as minimal histories (tests/golden/make_checker_doc_vectors.py wrote it and checked
every quoted fragment against the doc files).

CPU: the Python restatements (tests/setfull_ref.py, tests/linearizable_ref.py) and the host search (msim_check_lin_kv_rows) must
report what the docs report.  GPU (`-m gpu`): the device checkers behind msim_check — check_kernel through msim_check_set_full_batch,
lin_check_kernel through msim_check_lin_kv_batch — must too; tests/test_hipemu_parity.py runs the same GPU tests on the host wavefront
emulator in the CPU suite."""
import json
import os

import pytest

from maelstrom_amd import _abi as A
from maelstrom_amd import engine as E

REFERENCE_HELD = True   # marker: these expectations come from the reference tree, not from this repository's own restatements
HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "checker_doc_vectors.json")) as f:
    VECTORS = json.load(f)["vectors"]
SET = [v for v in VECTORS if v["checker"] == "set-full"]
LIN = [v for v in VECTORS if v["checker"] == "linearizable"]
VALID = {1: True, 0: False, 2: "unknown"}
COUNTS = (("attempt_count", "attempt-count"), ("stable_count", "stable-count"), ("lost_count", "lost-count"), ("stale_count", "stale-count"), ("never_read_count", "never-read-count"))


def test_fixture_is_what_the_generator_writes():
    import subprocess
    import sys
    if not os.path.isdir("/root/reference"):
        pytest.skip("the reference tree is not on this machine (the fixture travels, the docs do not)")
    before = open(os.path.join(HERE, "golden", "checker_doc_vectors.json")).read()
    r = subprocess.run([sys.executable, os.path.join(HERE, "golden", "make_checker_doc_vectors.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert open(os.path.join(HERE, "golden", "checker_doc_vectors.json")).read() == before
    assert "38 quoted fragments verified" in r.stdout


@pytest.mark.parametrize("v", SET, ids=[v["doc"] for v in SET])
def test_set_full_restatement_reports_what_the_reference_docs_report(v):
    import setfull_ref as R
    got = R.set_full(v["history"])
    for k, want in v["expect"].items():
        assert got[k] == want, (v["name"], k, got[k], want)
    # the encoding the device checker reads loses nothing
    rows, pay = E.encode_set_history(v["history"])
    back = E.decode_history(rows, pay, 5, A.WL_BROADCAST)
    for a, b in zip(back, v["history"]):
        assert (a["type"], a["f"], a["process"], a["value"], a["time"]) == (b["type"], b["f"], b["process"], b["value"], b["time"])


@pytest.mark.parametrize("v", LIN, ids=[v["doc"] + " " + v["name"][:24] for v in LIN])
def test_linearizability_search_reports_what_the_reference_docs_report(v):
    import linearizable_ref as L
    rows = E.encode_lin_kv_history(v["history"])
    got = E.check_lin_kv_history(rows)
    assert got["valid?"] is v["expect"]["valid?"] and got["key-count"] == 1 and got["invalid-keys"] == (0 if v["expect"]["valid?"] else 1), (v["name"], got)
    back = E.decode_history(rows, [], 1, A.WL_LIN_KV)
    ok, finals = L.check_key_configs(back)
    assert ok is v["expect"]["valid?"], (v["name"], ok)
    if "final-value" in v["expect"]:   # ":configs ({:model #knossos.model.CASRegister{:value 3}": the register the search ends with
        assert finals == [v["expect"]["final-value"]], finals


@pytest.mark.gpu
def test_device_set_full_checker_reports_what_the_reference_docs_report(lib):
    hs = [E.encode_set_history(v["history"]) for v in SET]
    res = E.check_set_full_batch(hs, 5, A.WL_BROADCAST, max_values=64)
    for v, g in zip(SET, res):
        assert VALID[int(g["valid"])] is v["expect"]["valid?"], (v["name"], g)
        for k, dk in COUNTS:
            if dk in v["expect"]:
                assert int(g[k]) == v["expect"][dk], (v["name"], dk, int(g[k]))


@pytest.mark.gpu
def test_device_linearizability_search_reports_what_the_reference_docs_report(lib):
    res = E.check_lin_kv_batch([E.encode_lin_kv_history(v["history"]) for v in LIN])
    for v, g in zip(LIN, res):
        assert VALID[int(g["valid"])] is v["expect"]["valid?"], (v["name"], g)
        assert int(g["attempt_count"]) == 1 and int(g["error_count"]) == (0 if v["expect"]["valid?"] else 1)
