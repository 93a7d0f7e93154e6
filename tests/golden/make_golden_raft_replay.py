#!/usr/bin/env python3
"""Records tests/golden/raft_replay_digests.json: for every case of tests/test_raft_reference_replay.py, first the replay of the
oracle's schedule through the reference's own demo/python/raft.py must pass (that is the check against the reference; it needs
/root/reference), then a digest of everything the nodes sent in that run is stored.  tests/test_raft_reference_replay.py::
test_runs_still_match_the_recorded_reference_replays recomputes the digests from the oracle alone, so the pinned behaviour
travels to machines without the reference tree.  Run from the repository root: python tests/golden/make_golden_raft_replay.py"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import test_raft_reference_replay as T  # noqa: E402

if __name__ == "__main__":
    out = {}
    for i, kw in enumerate(T.CASES):
        T.test_reference_raft_py_emits_what_the_oracle_emits.__wrapped__(kw) if hasattr(T.test_reference_raft_py_emits_what_the_oracle_emits, "__wrapped__") else T.test_reference_raft_py_emits_what_the_oracle_emits(kw)   # raises if raft.py disagrees with the oracle
        out[str(i)] = {"options": kw, "digests": T.run_digests(kw)}
        print(i, kw, out[str(i)]["digests"])
    with open(os.path.join(HERE, "raft_replay_digests.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
