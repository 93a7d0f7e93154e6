#!/usr/bin/env python3
"""Reference-held vectors for the list-append checker: the transactions the reference's own documentation prints together with the
anomaly the real Elle assigned to them (doc/05-datomic/*.md), turned into minimal histories -> tests/golden/elle_doc_vectors.json.

Besides pn_counter_test.clj these are the only checker results the reference tree holds ("Missing" #7).  Every vector
quotes the lines it is taken from; when /root/reference is present the script checks that each quoted fragment really occurs in the
named file (whitespace-insensitive), so the fixture cannot drift from the docs.  What a vector asserts is what the doc shows and
nothing more: `has` = anomaly classes Elle reported for those very transactions, `has_not` = classes the doc says were absent,
`exact` = the doc prints the run's whole `:anomaly-types` and the minimal history contains nothing else, `valid_under` = models the
doc's `:not` / `:also-not` sets (or a passing run) admit.  The surrounding transactions (the appends a read needs so that it does not
look like garbage) are ours and marked `filler`.

    python tests/golden/make_elle_doc_vectors.py        (run in the build container; needs /root/reference for the quote check)"""
import json
import os
import re

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
A, R = ":append", ":r"


def seq(*txns):
    """Sequential history: every transaction completes before the next is invoked; txn = (process, value, [type])."""
    ops = []
    for t in txns:
        p, v = t[0], t[1]
        typ = t[2] if len(t) > 2 else ":ok"
        inv = [[f, k, None] if f == R else [f, k, x] for f, k, x in v]
        ops.append({"type": ":invoke", "f": ":txn", "process": p, "value": inv})
        ops.append({"type": typ, "f": ":txn", "process": p, "value": [list(m) for m in v]})
    return ops


def appends(key, values, process=90):
    return [(process, [[A, key, v]]) for v in values]


L7 = [1, 2, 3, 4, 5, 6, 7]
VECTORS = [
    {
        "name": "internal: a transaction does not see its own append",
        "doc": "doc/05-datomic/01-single-node.md:117-141",
        "quotes": [":anomaly-types (:internal),", ":value [[:append 9 6] [:r 9 nil]],", ":mop [:r 9 nil],", ":expected [... 6]}",
                   ":value [[:append 9 12] [:r 9 nil]],", ":value [[:append 9 16] [:r 9 nil]],", ":not #{:read-atomic},"],
        # the three transactions the doc lists, as processes 3, 0, 4; nothing else (the doc's run had one node that forgot every append)
        "history": seq((3, [[A, 9, 6], [R, 9, None]]), (0, [[A, 9, 12], [R, 9, None]]), (4, [[A, 9, 16], [R, 9, None]])),
        "has": ["internal"], "exact": True, "valid": False, "internal_ops": 3,
    },
    {
        "name": "internal: reads that saw appends of their own future (the mutated list)",
        "doc": "doc/05-datomic/01-single-node.md:260-287",
        "quotes": [":anomaly-types (:internal),", ":mop [:r 9 [1 2 3 4 5 6 7]],",
                   "8	:ok	:txn	[[:r 6 nil] [:append 9 1] [:append 9 2]]", "0	:ok	:txn	[[:append 7 1] [:r 8 nil] [:r 9 [1 2]]]",
                   "1	:ok	:txn	[[:r 7 [1]] [:append 9 3]]", "9	:ok	:txn	[[:append 9 4] [:r 9 [1 2 3 4]]]",
                   "3	:ok	:txn	[[:r 9 [1 2 3 4]] [:r 7 [1]] [:r 7 [1]] [:r 8 nil]]", "0	:ok	:txn	[[:r 9 [1 2 3 4 5]] [:r 6 nil] [:append 9 5]]",
                   "8	:ok	:txn	[[:r 9 [1 2 3 4 5 6 7]] [:append 9 6] [:append 9 7] [:r 9 [1 2 3 4 5 6 7]]]"],
        # the doc's own grep of the history for key 9, in order (one node, so in completion order)
        "history": seq((8, [[R, 6, None], [A, 9, 1], [A, 9, 2]]), (0, [[A, 7, 1], [R, 8, None], [R, 9, [1, 2]]]), (1, [[R, 7, [1]], [A, 9, 3]]),
                       (9, [[A, 9, 4], [R, 9, [1, 2, 3, 4]]]), (3, [[R, 9, [1, 2, 3, 4]], [R, 7, [1]], [R, 7, [1]], [R, 8, None]]),
                       (0, [[R, 9, [1, 2, 3, 4, 5]], [R, 6, None], [A, 9, 5]]), (8, [[R, 9, L7], [A, 9, 6], [A, 9, 7], [R, 9, L7]])),
        "has": ["internal"], "exact": False, "valid": False,
    },
    {
        "name": "incompatible-order: two nodes, two lists per key",
        "doc": "doc/05-datomic/01-single-node.md:354-366",
        "quotes": ["Well at least we don't have any *internal* consistency anomalies", ":incompatible-order ({:key 7,", ":values [[3 4] [1 2]]}",
                   "{:key 9,", ":values [[4 5 7]", "[1 2 3]]}", "{:key 10,", ":values [[8] [1]]}", "{:key 8,", ":values [[3 4] [1 2 5]]})},"],
        # for every key the two reads the doc pairs; filler: one append per element so that no read is of an element nobody wrote
        "history": seq(*(appends(7, [1, 2, 3, 4]) + appends(9, [1, 2, 3, 4, 5, 7]) + appends(10, [1, 8]) + appends(8, [1, 2, 3, 4, 5]) + [
            (0, [[R, 7, [3, 4]]]), (1, [[R, 7, [1, 2]]]), (0, [[R, 9, [4, 5, 7]]]), (1, [[R, 9, [1, 2, 3]]]),
            (0, [[R, 10, [8]]]), (1, [[R, 10, [1]]]), (0, [[R, 8, [3, 4]]]), (1, [[R, 8, [1, 2, 5]]])])),
        "filler": "the single-append transactions of process 90",
        "has": ["incompatible-order"], "has_not": ["internal"], "exact": False, "valid": False, "incompatible_keys": [7, 8, 9, 10],
    },
    {
        "name": "internal: two reads of one key inside one transaction differ",
        "doc": "doc/05-datomic/02-shared-state.md:198,226-233",
        "quotes": [":anomaly-types (:G-single :G1b :internal),", ":value [[:r 51 [1 2 3 4 5 6 7 8 9 10 11 12 13 14 15]]",
                   "[:r 51 [1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16]]],", ":process 2,"],
        "history": seq(*(appends(51, list(range(1, 17))) + [(2, [[R, 51, list(range(1, 16))], [R, 51, list(range(1, 17))]])])),
        "filler": "the single-append transactions of process 90",
        "has": ["internal"], "exact": False, "valid": False, "internal_ops": 1,
    },
    {
        "name": "G-single-realtime: a read strictly after an append does not see it (stale lww-kv read)",
        "doc": "doc/05-datomic/04-optimization.md:22-24,52-56",
        "quotes": [":anomaly-types (:G-single-realtime),", ":not #{:strict-serializable},", ":also-not #{}},",
                   "one appends 7 to key", "9, and strictly later, in real time, another reads key 9, and sees `nil`"],
        # filler: a later read that shows 7 in key 9's list — without one nobody knows the append took effect (in the doc's run later
        # transactions did; Elle orders versions by what reads observed)
        "history": seq((0, [[A, 9, 7]]), (1, [[R, 9, None]]), (2, [[R, 9, [7]]])),
        "filler": "the last read (process 2)",
        "has": ["G-single", "realtime"], "exact": True, "valid": False, "valid_under": ["serializable"],
    },
    {
        "name": "G-single-realtime: a read-only transaction runs on a stale root; serializable all the same",
        "doc": "doc/05-datomic/04-optimization.md:311-345",
        "quotes": [":anomaly-types (:G-single-realtime),", ":not #{:strict-serializable},", "failed to observe its append of 15 to key 44!",
                   "--consistency-models serializable", "Everything looks good!"],
        # T1 appends 15 to key 44 (after 1..14 are there); T2, strictly later, reads key 44 without the 15
        "history": seq(*(appends(44, list(range(1, 15))) + [(0, [[A, 44, 15]]), (1, [[R, 44, list(range(1, 15))]]), (2, [[R, 44, list(range(1, 16))]])])),
        "filler": "the single-append transactions of process 90 and the last read (process 2), which shows that the append of 15 took effect",
        "has": ["G-single", "realtime"], "exact": True, "valid": False, "valid_under": ["serializable"],
    },
]


def check_quotes():
    squash = lambda s: re.sub(r"\s+", " ", s)
    for v in VECTORS:
        path = os.path.join(REF, v["doc"].split(":")[0])
        text = squash(open(path).read())
        for q in v["quotes"]:
            assert squash(q) in text, f"{v['doc']}: quote not found: {q!r}"


def main():
    if os.path.isdir(REF):
        check_quotes()
        print("quotes found in the reference docs")
    out = os.path.join(HERE, "elle_doc_vectors.json")
    with open(out, "w") as f:
        json.dump({"source": "jepsen-io/maelstrom doc/05-datomic (the anomalies are the real Elle's, printed in the docs)", "vectors": VECTORS}, f, indent=1)
    print("wrote", out, len(VECTORS), "vectors")


if __name__ == "__main__":
    main()
