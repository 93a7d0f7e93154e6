#!/usr/bin/env python3
"""Reference-held vectors for the set-full checker (broadcast / g-set) and the linearizability search (lin-kv): the reads and verdicts
the reference's own documentation prints, turned into minimal histories -> tests/golden/checker_doc_vectors.json.

[upstream] jepsen.checker/set-full and Knossos are not vendored in the reference tree, so the device checkers behind
`histories_per_sec` were so far pinned only by this repository's own restatements (tests/setfull_ref.py, tests/linearizable_ref.py).
The tutorial chapters DO print runs of the real checkers: the final reads of a run in their invocation / completion order together
with the result map (which elements were stable, stale, lost), and a non-linearizable pair of operations.  Every vector quotes the
lines it is taken from; when /root/reference is present the script checks that each quoted fragment really occurs in the named file
(whitespace-insensitive), so the fixture cannot drift from the docs.  What a vector asserts is what the doc shows and nothing more;
what the doc elides ("...": the main phase's operations) is filled in minimally and marked `filler`.

    python tests/golden/make_checker_doc_vectors.py        (run in the build container; needs /root/reference for the quote check)"""
import json
import os
import re

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
MS = 1_000_000   # history :time is in nanoseconds


def adds(f, owner, t0_ms, gap_ms):
    """Every element acknowledged once: {element: process}; element v is invoked at t0 + v * gap and acknowledged 1 ms later."""
    ops = []
    for v in sorted(owner):
        t = (t0_ms + v * gap_ms) * MS
        ops.append({"type": ":invoke", "f": f, "process": owner[v], "value": v, "time": t})
        ops.append({"type": ":ok", "f": f, "process": owner[v], "value": v, "time": t + MS})
    return ops


def reads(invokes, completions, final=True):
    """invokes: [(process, t_ms)] in history order; completions: [(process, t_ms, elements)] in history order."""
    ops = [dict({"type": ":invoke", "f": ":read", "process": p, "value": None, "time": t * MS}, **({"final?": True} if final else {})) for p, t in invokes]
    ops += [dict({"type": ":ok", "f": ":read", "process": p, "value": list(v), "time": t * MS}, **({"final?": True} if final else {})) for p, t, v in completions]
    return ops


# ---- doc/03-broadcast/01-broadcast.md:388-430: five nodes that do not gossip yet ----
B1_FINAL = {3: [1, 3, 6, 12], 2: [4, 16, 20], 0: [0, 9, 10, 11, 15, 18], 1: [2, 5, 7, 13, 14, 17, 19], 4: [8]}
B1_OWNER = {v: p for p, vs in B1_FINAL.items() for v in vs}   # a node only ever has its own client's messages
B1 = (adds(":broadcast", {v: p for v, p in B1_OWNER.items() if v <= 11}, 1000, 500)
      + reads([(0, 7456)], [(0, 7458, [0, 9, 10, 11])], final=False)          # "0 :invoke :read nil / 0 :ok :read [0 9 10 11]" right after broadcast 11
      + adds(":broadcast", {v: p for v, p in B1_OWNER.items() if v > 11}, 2000, 500)
      + reads([(2, 30042), (3, 30043), (0, 30043), (1, 30043), (4, 30043)],
              [(3, 30044, B1_FINAL[3]), (2, 30044, B1_FINAL[2]), (0, 30044, B1_FINAL[0]), (1, 30044, B1_FINAL[1]), (4, 30044, B1_FINAL[4])]))

# ---- doc/03-broadcast/02-performance.md:282-301: the grid under partitions, no retries yet ----
P_ALL = [3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 19, 20, 21, 22, 23, 24, 25, 26, 28, 29, 31, 32, 33, 34, 35, 36]
B2_FINAL = {2: P_ALL + [37, 39], 4: P_ALL, 3: [0, 1, 2, 17, 27, 30, 32, 33, 34, 35, 36, 38, 40, 41, 42, 43, 44], 1: P_ALL + [37, 39], 0: P_ALL + [38, 40, 41, 42, 43, 44]}
B2 = (adds(":broadcast", {v: v % 5 for v in range(45)}, 1000, 400)
      + reads([(4, 38996), (2, 38996), (1, 38997), (3, 38997), (0, 38997)],
              [(2, 38998, B2_FINAL[2]), (4, 38998, B2_FINAL[4]), (3, 38998, B2_FINAL[3]), (1, 38999, B2_FINAL[1]), (0, 38999, B2_FINAL[0])]))

VECTORS = [
    {
        "checker": "set-full", "name": "broadcast without gossip: every node only has its own client's messages",
        "doc": "doc/03-broadcast/01-broadcast.md:388-430",
        "quotes": ["jepsen.util 0	:invoke	:broadcast	11", "jepsen.util 0	:ok	:broadcast	11", "jepsen.util 0	:ok	:read	[0 9 10 11]",
                   "jepsen.util 2	:invoke	:read	nil\nINFO [2021-02-23 10:31:11,243] jepsen worker 3 - jepsen.util 3	:invoke	:read	nil\nINFO [2021-02-23 10:31:11,243] jepsen worker 0 - jepsen.util 0	:invoke	:read	nil\nINFO [2021-02-23 10:31:11,243] jepsen worker 1 - jepsen.util 1	:invoke	:read	nil\nINFO [2021-02-23 10:31:11,243] jepsen worker 4 - jepsen.util 4	:invoke	:read	nil",
                   "jepsen.util 3	:ok	:read	[1 3 6 12]", "jepsen.util 2	:ok	:read	[4 16 20]", "jepsen.util 0	:ok	:read	[0 9 10 11 15 18]",
                   "jepsen.util 1	:ok	:read	[2 5 7 13 14 17 19]", "jepsen.util 4	:ok	:read	[8]",
                   ":attempt-count 21,", ":stable-count 1,", ":stale-count 1,", ":stale (8),", ":lost-count 20,",
                   ":lost (0 1 2 3 4 5 6 7 9 10 11 12 13 14 15 16 17 18 19 20),",
                   "only one was *stable*: present durably\nin all reads after some time *t*", "worker 4 saw the set of\nmessages as just `[8]`. All other messages were considered *lost*."],
        "filler": "the broadcasts' times and issuing workers (a node holds what its own client sent, so the final reads name the worker of each message); the doc elides the main phase with '...'",
        "workload": "broadcast", "concurrency": 5, "history": B1,
        "expect": {"valid?": False, "attempt-count": 21, "stable-count": 1, "stale-count": 1, "stale": [8], "lost-count": 20,
                   "lost": [0, 1, 2, 3, 4, 5, 6, 7, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20], "never-read-count": 0},
    },
    {
        "checker": "set-full", "name": "grid broadcast under partitions without retries: messages lost on some nodes",
        "doc": "doc/03-broadcast/02-performance.md:282-301",
        "quotes": [":valid? false,\n            :lost-count 8,\n            :lost (0 1 2 17 27 30 37 39),\n            :stable-count 37,\n            :stale-count 32,",
                   "jepsen.util 4	:invoke	:read	nil\nINFO [2021-02-24 16:56:58,996] jepsen worker 2 - jepsen.util 2	:invoke	:read	nil\nINFO [2021-02-24 16:56:58,997] jepsen worker 1 - jepsen.util 1	:invoke	:read	nil\nINFO [2021-02-24 16:56:58,997] jepsen worker 3 - jepsen.util 3	:invoke	:read	nil\nINFO [2021-02-24 16:56:58,997] jepsen worker 0 - jepsen.util 0	:invoke	:read	nil",
                   "jepsen.util 2	:ok	:read	[3 4 5 6 7 8 9 10 11 12 13 14 15 16 18 19 20 21 22 23 24 25 26 28 29 31 32 33 34 35 36 37 39]",
                   "jepsen.util 4	:ok	:read	[3 4 5 6 7 8 9 10 11 12 13 14 15 16 18 19 20 21 22 23 24 25 26 28 29 31 32 33 34 35 36]",
                   "jepsen.util 3	:ok	:read	[0 1 2 17 27 30 32 33 34 35 36 38 40 41 42 43 44]",
                   "jepsen.util 1	:ok	:read	[3 4 5 6 7 8 9 10 11 12 13 14 15 16 18 19 20 21 22 23 24 25 26 28 29 31 32 33 34 35 36 37 39]",
                   "jepsen.util 0	:ok	:read	[3 4 5 6 7 8 9 10 11 12 13 14 15 16 18 19 20 21 22 23 24 25 26 28 29 31 32 33 34 35 36 38 40 41 42 43 44]",
                   "Some messages, like 0, are present on some nodes, but not\nothers."],
        "filler": "the 45 acknowledged broadcasts (times, issuing workers); the doc prints the closing reads and the result only",
        "workload": "broadcast", "concurrency": 5, "history": B2,
        "expect": {"valid?": False, "lost-count": 8, "lost": [0, 1, 2, 17, 27, 30, 37, 39], "stable-count": 37, "stale-count": 32},
    },
    {
        "checker": "linearizable", "name": "one Raft node: a write, the read that sees it, the closing cas",
        "doc": "doc/06-raft/01-key-value.md:131-157",
        "quotes": ["jepsen.util: 1	:invoke	:write	[0 2]", "jepsen.util: 1	:ok	:write	[0 2]", "jepsen.util: 0	:invoke	:read	[0 nil]", "jepsen.util: 0	:ok	:read	[0 2]",
                   ":results {0 {:linearizable {:valid? true,", ":configs ({:model #knossos.model.CASRegister{:value 3},", ":f :cas,", ":value [2\n                                                                     3],",
                   "the last operation to execute was\na `cas` of 2 to 3, and the resulting value was `3`"],
        "filler": "nothing: the four printed rows and the run's last operation",
        "history": [{"type": ":invoke", "f": ":write", "process": 1, "value": [0, 2], "time": 1300 * MS}, {"type": ":ok", "f": ":write", "process": 1, "value": [0, 2], "time": 1303 * MS},
                    {"type": ":invoke", "f": ":read", "process": 0, "value": [0, None], "time": 1526 * MS}, {"type": ":ok", "f": ":read", "process": 0, "value": [0, 2], "time": 1527 * MS},
                    {"type": ":invoke", "f": ":cas", "process": 1, "value": [0, [2, 3]], "time": 9786 * MS}, {"type": ":ok", "f": ":cas", "process": 1, "value": [0, [2, 3]], "time": 9787 * MS}],
        "expect": {"valid?": True, "final-value": 3},
    },
    {
        "checker": "linearizable", "name": "two independent copies: a write of 2 followed by a read of 4",
        "doc": "doc/06-raft/01-key-value.md:172-195",
        "quotes": ["Analysis invalid!", "this test run produced a write of 2 followed by\na read of 4--clearly impossible without an intervening write of 4.",
                   "we cannot\nexecute a read of 4 if the current state is 2."],
        "filler": "processes and times (the doc shows the pair in a plot)",
        "history": [{"type": ":invoke", "f": ":write", "process": 1, "value": [0, 2], "time": 1000 * MS}, {"type": ":ok", "f": ":write", "process": 1, "value": [0, 2], "time": 1002 * MS},
                    {"type": ":invoke", "f": ":read", "process": 0, "value": [0, None], "time": 1100 * MS}, {"type": ":ok", "f": ":read", "process": 0, "value": [0, 4], "time": 1101 * MS}],
        "expect": {"valid?": False},
    },
    {
        "checker": "linearizable", "name": "the same pair with the intervening write of 4 the doc names",
        "doc": "doc/06-raft/01-key-value.md:184-186",
        "quotes": ["clearly impossible without an intervening write of 4"],
        "filler": "the intervening write itself: the doc's sentence read the other way round",
        "history": [{"type": ":invoke", "f": ":write", "process": 1, "value": [0, 2], "time": 1000 * MS}, {"type": ":ok", "f": ":write", "process": 1, "value": [0, 2], "time": 1002 * MS},
                    {"type": ":invoke", "f": ":write", "process": 2, "value": [0, 4], "time": 1050 * MS}, {"type": ":ok", "f": ":write", "process": 2, "value": [0, 4], "time": 1052 * MS},
                    {"type": ":invoke", "f": ":read", "process": 0, "value": [0, None], "time": 1100 * MS}, {"type": ":ok", "f": ":read", "process": 0, "value": [0, 4], "time": 1101 * MS}],
        "expect": {"valid?": True},
    },
]


def squash(s):
    return re.sub(r"\s+", " ", s).strip()


def main():
    checked = 0
    if os.path.isdir(REF):
        for v in VECTORS:
            path = os.path.join(REF, v["doc"].split(":")[0])
            txt = squash(open(path).read())
            for q in v["quotes"]:
                assert squash(q) in txt, (v["doc"], q)
                checked += 1
    else:
        print("warning: /root/reference absent — quotes NOT checked")
    for v in VECTORS:
        for i, op in enumerate(v["history"]):
            op["index"] = i
    out = os.path.join(HERE, "checker_doc_vectors.json")
    with open(out, "w") as f:
        json.dump({"source": "tests/golden/make_checker_doc_vectors.py", "quotes_checked": checked, "vectors": VECTORS}, f, indent=1)
    print(f"wrote {out}: {len(VECTORS)} vectors, {checked} quoted fragments verified against {REF}")


if __name__ == "__main__":
    main()
