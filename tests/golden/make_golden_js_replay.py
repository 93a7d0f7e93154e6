#!/usr/bin/env python3
"""Records tests/golden/js_crdt_replay_digests.json: for every case of tests/test_js_reference_replay.py the real
demo/js/crdt_gset.js / crdt_pn_counter.js processes must first reproduce the oracle's run message by message (needs node.js and
/root/reference); then a digest of that run is stored so that the pinned behaviour travels.
Run from the repository root: python tests/golden/make_golden_js_replay.py"""
import json
import os
import pathlib
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import test_js_reference_replay as T  # noqa: E402

if __name__ == "__main__":
    out = {}
    fn = T.test_reference_js_crdt_processes_print_what_the_oracle_sends
    for i, (workload, script, kw) in enumerate(T.CASES):
        with tempfile.TemporaryDirectory() as d:
            fn(workload, script, kw, pathlib.Path(d))      # raises if a process prints anything else than the oracle sent
        out[str(i)] = {"workload": workload, "script": script, "options": kw, "digest": T.run_digest(workload, kw)}
        print(i, workload, kw, out[str(i)]["digest"])
    for j, kw in enumerate(T.TXN_CASES):
        with tempfile.TemporaryDirectory() as d:
            T.test_reference_single_key_txn_js_processes_print_what_the_oracle_sends(kw, pathlib.Path(d))
        out[str(len(T.CASES) + j)] = {"workload": "txn-list-append", "script": "single_key_txn.js", "options": kw, "digest": T.run_digest("txn-list-append", kw)}
        print("txn", j, kw, out[str(len(T.CASES) + j)]["digest"])
    base = len(T.CASES) + len(T.TXN_CASES)
    for j, kw in enumerate(T.GOSSIP_CASES):
        with tempfile.TemporaryDirectory() as d:
            T.test_reference_gossip_js_processes_print_what_the_oracle_sends(kw, pathlib.Path(d))
        out[str(base + j)] = {"workload": "broadcast", "script": "gossip.js", "options": kw, "digest": T.run_digest("broadcast", kw)}
        print("gossip", j, kw, out[str(base + j)]["digest"])
    with open(os.path.join(HERE, "js_crdt_replay_digests.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
