"""check_kernel (csrc/checker.hip, through msim_check_set_full_batch) on SYNTHETIC histories no run of the engine produces, against the
pure-Python restatement of [upstream] jepsen.checker/set-full (tests/setfull_ref.py): the shapes the round-6 sweep (a lane per read,
64 reads at a time, 16-byte loads of each lane's own bitmap) has code for and the engine's own histories seldom reach —
  * reads that complete in any order relative to their invocations (known = the smallest :ok index among the containing reads),
  * reads that fail or never complete in the middle of a chunk of 64 (ranks without a record),
  * elements that come and go between reads (last-present / last-absent far behind the frontier: no word is "settled"),
  * more than 1024 elements (bitmaps of 33 .. 47 words: the words beyond the eight 16-byte loads of the first pass over a chunk),
  * a bitmap that ends in the payload slab's last three words (the longest history of a batch fills its slab exactly), a slab of
    fewer than four words, read counts of exactly 64 / 128 (full last chunk) and 65.
tests/test_hipemu_parity.py runs this file on the host wavefront emulator in the CPU suite."""
import random

import numpy as np
import pytest

from maelstrom_amd import _abi as A
from maelstrom_amd import engine as E
import setfull_ref as R

pytestmark = pytest.mark.gpu
VALID = {1: True, 0: False, 2: "unknown"}


def synth(seed, n_reads, n_adds, workers, p_fail=0.05, p_info=0.03, flicker=0.0, lag=8, overtake=True):
    """A history of `n_adds` adds (elements 0, 1, 2 .. in invocation order) and `n_reads` reads by `workers` worker threads.  A read
    sees every element acknowledged more than `lag` elements ago (minus a random `flicker` share: elements that come and go) and a
    random subset of the newer ones; completions are drawn from the open operations at random (`overtake`) or oldest first."""
    rnd = random.Random(seed)
    ops, open_ops, free = [], [], list(range(workers))
    proc_of = {w: w for w in range(workers)}
    t = 0
    next_el, acked = 0, []
    reads_left, adds_left = n_reads, n_adds
    while reads_left or adds_left or open_ops:
        t += rnd.randrange(1, 3_000_000)
        can_invoke = free and (reads_left or adds_left)
        if can_invoke and (not open_ops or rnd.random() < 0.55):
            w = free.pop(rnd.randrange(len(free)))
            if adds_left and (not reads_left or rnd.random() < adds_left / (adds_left + reads_left)):
                ops.append({"type": ":invoke", "f": ":add", "process": proc_of[w], "value": next_el, "time": t})
                open_ops.append((w, ":add", next_el)); next_el += 1; adds_left -= 1
            else:
                ops.append({"type": ":invoke", "f": ":read", "process": proc_of[w], "value": None, "time": t})
                open_ops.append((w, ":read", None)); reads_left -= 1
            continue
        w, f, el = open_ops.pop(rnd.randrange(len(open_ops)) if overtake else 0)
        x = rnd.random()
        typ = ":fail" if x < p_fail else (":info" if x < p_fail + p_info else ":ok")
        if f == ":add":
            ops.append({"type": typ, "f": ":add", "process": proc_of[w], "value": el, "time": t})
            if typ != ":fail":
                acked.append(el)
        else:
            val = None
            if typ == ":ok":
                old = acked[:-lag] if len(acked) > lag else []
                new = acked[len(old):]
                val = [e for e in old if rnd.random() >= flicker] + [e for e in new if rnd.random() < 0.5]
                rnd.shuffle(val)
            ops.append({"type": typ, "f": ":read", "process": proc_of[w], "value": val, "time": t})
        if typ == ":info":
            proc_of[w] += workers   # [upstream] a crashed process is replaced by process + concurrency on the same thread
        free.append(w)
    for i, op in enumerate(ops):
        op["index"] = i
    return ops


def _compare(histories, workers, max_values=None):
    res = E.check_set_full_batch([E.encode_set_history(h) for h in histories], workers, A.WL_G_SET, max_values=max_values)
    for i, (h, g) in enumerate(zip(histories, res)):
        ref = R.set_full(h)
        assert VALID[int(g["valid"])] == ref["valid?"], (i, g, {k: ref[k] for k in ref if "count" in k})
        for k, rk in (("attempt_count", "attempt-count"), ("stable_count", "stable-count"), ("lost_count", "lost-count"),
                      ("never_read_count", "never-read-count"), ("stale_count", "stale-count")):
            assert int(g[k]) == ref[rk], (i, k, int(g[k]), ref[rk])
        if ref["stable-latencies"]:
            assert [int(x) for x in g["stable_latency_ms"]] == [ref["stable-latencies"][q] for q in (0, 0.5, 0.95, 0.99, 1)], (i, g, ref["stable-latencies"])
    return res


def test_reads_in_any_order_with_failures_and_elements_that_come_and_go(lib):
    for s in range(6):   # (one call per worker count: process mod concurrency is the worker thread)
        res = _compare([synth(100 + 10 * s + k, n_reads=150 + 37 * s, n_adds=90 + 11 * s, workers=3 + 4 * s, flicker=0.02 * s) for k in range(3)], workers=3 + 4 * s)
        assert (res["attempt_count"] > 80).all()


def test_more_than_1024_elements_and_a_bitmap_at_the_end_of_the_slab(lib):
    # 1100 .. 1500 elements: the last reads' bitmaps are 35 .. 47 words; the longest history's payload fills the batch's slab to the last word
    hs = [synth(200 + s, n_reads=70 + 30 * s, n_adds=1100 + 200 * s, workers=8, p_fail=0.02, p_info=0.0, flicker=0.001 * s, lag=40) for s in range(3)]
    _compare(hs, workers=8)
    # ... and the slab's end at every alignment: a history checked alone fills its slab exactly; bitmaps of 2 .. 5 words
    seen = set()
    for extra in range(8):
        h = synth(300 + extra, n_reads=40, n_adds=37 + 17 * extra, workers=5, p_fail=0.0, p_info=0.0, lag=3)
        seen.add(len(E.encode_set_history(h)[1]) % 4)
        _compare([h], workers=5)
    assert len(seen) >= 3, seen


def test_tiny_payload_slabs_and_full_chunks(lib):
    tiny = [
        [{"type": ":invoke", "f": ":add", "process": 0, "value": 0, "time": 1}, {"type": ":ok", "f": ":add", "process": 0, "value": 0, "time": 2},
         {"type": ":invoke", "f": ":read", "process": 1, "value": None, "time": 3}, {"type": ":ok", "f": ":read", "process": 1, "value": [0], "time": 4_000_000},
         {"type": ":invoke", "f": ":read", "process": 0, "value": None, "time": 5_000_000}, {"type": ":ok", "f": ":read", "process": 0, "value": [], "time": 6_000_000},
         {"type": ":invoke", "f": ":read", "process": 1, "value": None, "time": 7_000_000}, {"type": ":ok", "f": ":read", "process": 1, "value": [0], "time": 9_000_000}],
    ]
    for h in tiny:
        for i, op in enumerate(h):
            op["index"] = i
    assert len(E.encode_set_history(tiny[0])[1]) < 4   # a slab of fewer than four words
    _compare(tiny, workers=2)
    for n in (64, 65, 128):   # the reads fill the last chunk of 64 exactly / leave one read for the next
        _compare([synth(400 + n, n_reads=n, n_adds=50, workers=6, p_fail=0.0, p_info=0.0)], workers=6)


def test_a_read_that_completes_before_every_earlier_one(lib):
    # 70 reads invoked one after another, completed in REVERSE order: every read but the last-invoked is overtaken by all later ones
    h = []
    t = 0
    for e in range(3):
        t += 1000; h.append({"type": ":invoke", "f": ":add", "process": 100, "value": e, "time": t})
        t += 1000; h.append({"type": ":ok", "f": ":add", "process": 100, "value": e, "time": t})
    for p in range(70):
        t += 1000; h.append({"type": ":invoke", "f": ":read", "process": p, "value": None, "time": t})
    for p in reversed(range(70)):
        t += 2_000_000; h.append({"type": ":ok", "f": ":read", "process": p, "value": [0, 2] if p % 3 else [1], "time": t})
    for i, op in enumerate(h):
        op["index"] = i
    _compare([h], workers=101)


def test_random_shapes(lib):
    """Forty histories with every parameter drawn at random: worker counts, failure rates, how far the reads lag behind the adds, how
    many elements come and go — between 30 and 1400 elements, between 5 and 300 reads."""
    rnd = random.Random(2026)
    for k in range(40):
        workers = rnd.choice([1, 2, 3, 5, 8, 13, 25, 60])
        h = synth(5000 + k, n_reads=rnd.randrange(5, 300), n_adds=rnd.choice([30, 64, 100, 333, 700, 1030, 1400]), workers=workers,
                  p_fail=rnd.choice([0.0, 0.02, 0.2]), p_info=rnd.choice([0.0, 0.05]), flicker=rnd.choice([0.0, 0.0, 0.003, 0.05]),
                  lag=rnd.choice([0, 2, 10, 60]), overtake=rnd.random() < 0.7)
        _compare([h], workers=workers)
