"""The partition nemesis (nemesis.clj:10-16 over [upstream] jepsen.nemesis.combined/partition-package, restated — parity
unpinned): the grudges it produces have the shapes the published Jepsen documentation describes, the
schedule flip-flops start / stop with the configured mean interval, and the network obeys them (no :recv across a cut)."""
import collections

import numpy as np
import pytest

from maelstrom_amd import _abi as A
from maelstrom_amd import engine as E
import oracle_lib as O


def _nemesis_ops(cfg, inst):
    r = O.run(cfg, inst, 1)
    assert r.meta["flags"][0] == 0
    ops = E.decode_history(*r.history(0), cfg.n_nodes, cfg.workload)
    return r, [o for o in ops if o["process"] == ":nemesis"]


@pytest.mark.parametrize("n", [3, 5, 7, 12])
def test_grudges_have_the_documented_shapes(n):
    cfg = E.test_config("broadcast", bin="broadcast-ack-retry", node_count=n, rate=5, time_limit=120, latency=5, nemesis=["partition"], nemesis_interval=1, seed=41)
    seen = collections.Counter()
    for inst in range(4):
        _, nem = _nemesis_ops(cfg, inst)
        assert len(nem) % 2 == 0 and len(nem) >= 80
        for i in range(0, len(nem), 2):
            a, b = nem[i], nem[i + 1]
            assert a["time"] == b["time"] and a["type"] == b["type"] == ":info"      # invocation and completion of the nemesis op
            last = i + 2 == len(nem)               # the final phase heals the network whatever came last (core.clj:74-77)
            assert a["f"] == b["f"] == (":stop-partition" if last or (i // 2) % 2 else ":start-partition")   # flip-flop
            if a["f"] == ":stop-partition":
                assert b["value"] == ":network-healed"
                continue
            spec, (tag, grudge) = a["value"], b["value"]
            assert tag == ":isolated"
            drops = {int(d[1:]): {int(s[1:]) for s in srcs} for d, srcs in grudge.items()}
            sees = {d: set(range(n)) - drops.get(d, set()) for d in range(n)}
            assert all(d in sees[d] for d in range(n))
            seen[spec] += 1
            if spec == ":one":                     # one node cut off from all others, both directions
                lone = [d for d in range(n) if len(sees[d]) == 1]
                assert len(lone) == 1 and all(sees[d] == set(range(n)) - {lone[0]} for d in range(n) if d != lone[0])
            elif spec in (":majority", ":minority-third"):   # complete grudge over two components of a shuffled node list
                comps = {frozenset(v) for v in sees.values()}
                small = n // 2 if spec == ":majority" else (n - 1) // 3
                if small == 0:
                    assert comps == {frozenset(range(n))}     # nothing to cut (3 nodes, minority third)
                else:
                    assert sorted(len(c) for c in comps) == [small, n - small]
                    assert all(sees[x] == c for c in comps for x in c)   # symmetric and transitive
            else:                                  # majorities-ring: everyone sees a majority, nobody the same one
                assert spec == ":majorities-ring"
                assert all(len(v) == n // 2 + 1 for v in sees.values())
                assert len({frozenset(v) for v in sees.values()}) == n
    assert set(seen) == {":one", ":majority", ":majorities-ring", ":minority-third"} and min(seen.values()) >= 10


def test_schedule_interval_and_the_cut_is_obeyed():
    cfg = E.test_config("broadcast", bin="broadcast-ack-retry", node_count=5, rate=30, time_limit=200, latency=5, nemesis=["partition"], nemesis_interval=2,
                        seed=42, journal_capacity=1500000)
    gaps, crossed = [], 0
    for inst in range(2):
        r, nem = _nemesis_ops(cfg, inst)
        times = [o["time"] / 1e9 for o in nem[::2]]
        gaps += list(np.diff(times))
        # replay the cuts over the journal: a node never receives from a source it is currently dropping
        marks = [(o["time"] // 1000, {int(d[1:]): {int(s[1:]) for s in v} for d, v in o["value"][1].items()} if o["f"] == ":start-partition" else {})
                 for o in nem[1::2]]
        k, cur = 0, {}
        for ev in r.events(0):
            t, msg, route = int(ev["time_us"]), int(ev["msg"]), int(ev["route"])
            while k < len(marks) and marks[k][0] <= t:
                cur = marks[k][1]; k += 1
            if (msg >> 7) & 1:
                src, dest = route & 0xFF, (route >> 8) & 0xFF
                # a message taken off the queue before the cut may still be delivered after it (it is already past the check)
                crossed += src in cur.get(dest, ()) and t > marks[k - 1][0] + 5000
    assert crossed == 0
    assert 1.8 < np.mean(gaps) < 2.2 and max(gaps) < 4.0 and min(gaps) >= 0.0    # (gen/stagger interval): uniform on [0, 2 x interval)
