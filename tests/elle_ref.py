"""Pure-Python restatement of the list-append analysis (elle, Kingsbury & Alvaro VLDB 2020 §4-§5) for the default
strict-serializable model — written independently of csrc/txn_check.cpp (dict/set based, recursive cycle search) to
cross-check it on engine histories and on mutated ones.  Test infrastructure only.

history: list of op maps {type, process, value=[[f, k, v], ...]} in history order (as engine.decode_history yields)."""


def analyse(ops):
    txns, open_by_proc = [], {}
    frontier, rt_pred = [], {}
    for i, op in enumerate(ops):
        if op.get("process") == ":nemesis" or op.get("f") != ":txn":
            continue
        p = op["process"]
        if op["type"] == ":invoke":
            t = {"id": len(txns), "type": ":info", "mops": op["value"]}
            open_by_proc[p] = t
            rt_pred[t["id"]] = list(frontier)
            txns.append(t)
        elif p in open_by_proc:
            t = open_by_proc.pop(p)
            t["type"] = op["type"]
            if op["type"] == ":ok":
                t["mops"] = op["value"]
                frontier = [f for f in frontier if f not in rt_pred[t["id"]]] + [t["id"]]
    anomalies = set()
    writer, final = {}, {}
    for t in txns:
        for f, k, v in t["mops"]:
            if f == ":append":
                if (k, v) in writer:
                    anomalies.add("duplicate-elements")
                writer[(k, v)] = t["id"]
                final[(t["id"], k)] = v
    longest = {}
    for t in txns:
        if t["type"] != ":ok":
            continue
        seen_read, own = {}, {}
        for f, k, v in t["mops"]:
            if f == ":append":
                if k in seen_read:
                    seen_read[k] = seen_read[k] + [v]
                else:
                    own.setdefault(k, []).append(v)
                continue
            lst = v or []
            if len(set(lst)) != len(lst):
                anomalies.add("duplicate-elements")
            if k in seen_read:
                if seen_read[k] != lst:
                    anomalies.add("internal")
            else:
                o = own.get(k, [])
                if o and lst[len(lst) - len(o):] != o:
                    anomalies.add("internal")
            seen_read[k] = list(lst)
            ext = list(lst)
            while ext and writer.get((k, ext[-1])) == t["id"]:
                ext.pop()
            for e in ext:
                w = writer.get((k, e))
                if w is None or txns[w]["type"] == ":fail":
                    anomalies.add("G1a")
            if ext:
                w = writer.get((k, ext[-1]))
                if w is not None and w != t["id"] and final[(w, k)] != ext[-1]:
                    anomalies.add("G1b")
            if k not in longest or len(lst) > len(longest[k]):
                longest[k] = list(lst)
    edges = {}  # (a, b) -> set of kinds

    def add(a, b, kind):
        if a != b:
            edges.setdefault((a, b), set()).add(kind)
    live = lambda w: w is not None and txns[w]["type"] != ":fail"
    for k, order in longest.items():
        for x, y in zip(order, order[1:]):
            a, b = writer.get((k, x)), writer.get((k, y))
            if a is None or b is None:
                continue
            if txns[a]["type"] == ":fail" and txns[b]["type"] != ":fail":
                anomalies.add("dirty-update")
            if live(a) and live(b):
                add(a, b, "ww")
    for t in txns:
        if t["type"] != ":ok":
            continue
        for f, k, v in t["mops"]:
            if f != ":r" or k not in longest:
                continue
            lst, order = v or [], longest[k]
            if lst != order[:len(lst)]:
                anomalies.add("incompatible-order")
                continue
            ext = list(lst)
            while ext and writer.get((k, ext[-1])) == t["id"]:
                ext.pop()
            if ext and live(writer.get((k, ext[-1]))):
                add(writer[(k, ext[-1])], t["id"], "wr")
            if len(lst) < len(order) and live(writer.get((k, order[len(lst)]))):
                add(t["id"], writer[(k, order[len(lst)])], "rw")
    for t in txns:
        if t["type"] != ":fail":
            for f in rt_pred[t["id"]]:
                add(f, t["id"], "rt")

    def has_cycle(kinds):
        adj = {}
        for (a, b), ks in edges.items():
            if ks & kinds:
                adj.setdefault(a, []).append(b)
        color = {}
        for root in list(adj):
            if color.get(root):
                continue
            stack = [(root, iter(adj.get(root, [])))]
            color[root] = 1
            while stack:
                v, it = stack[-1]
                for w in it:
                    if color.get(w) == 1:
                        return True
                    if not color.get(w):
                        color[w] = 1
                        stack.append((w, iter(adj.get(w, []))))
                        break
                else:
                    color[v] = 2
                    stack.pop()
        return False
    dep = {"ww", "wr", "rw"}
    if has_cycle(dep):
        anomalies.add("cycle")
    elif has_cycle(dep | {"rt"}):
        anomalies.update({"cycle", "realtime"})
    ok = sum(1 for t in txns if t["type"] == ":ok")
    return {"valid?": "unknown" if (not anomalies and ok == 0) else not anomalies, "anomalies": anomalies, "txn-count": len(txns), "ok-count": ok}


def analyse_rw(ops):
    """rw-register analysis (elle.rw-register with writes-follow-reads version inference, as jepsen.tests.cycle.wr is
    wired by workload/txn_rw_register.clj:162-166) — independent of csrc/txn_check.cpp: dicts, sets, naive reachability.
    Returns the anomaly names with the cycle class (G0 / G1c / G-single / G2, + realtime when a realtime edge is needed)."""
    txns, open_by_proc = [], {}
    frontier, rt_pred = [], {}
    for op in ops:
        if op.get("process") == ":nemesis" or op.get("f") != ":txn":
            continue
        p = op["process"]
        if op["type"] == ":invoke":
            t = {"id": len(txns), "type": ":info", "mops": op["value"]}
            open_by_proc[p] = t
            rt_pred[t["id"]] = list(frontier)
            txns.append(t)
        elif p in open_by_proc:
            t = open_by_proc.pop(p)
            t["type"] = op["type"]
            if op["type"] == ":ok":
                t["mops"] = op["value"]
                frontier = [f for f in frontier if f not in rt_pred[t["id"]]] + [t["id"]]
    anomalies = set()
    writer, final = {}, {}
    for t in txns:
        for f, k, v in t["mops"]:
            if f == ":w":
                if (k, v) in writer:
                    anomalies.add("duplicate-elements")
                writer[(k, v)] = t["id"]
                final[(t["id"], k)] = v
    live = lambda w: w is not None and txns[w]["type"] != ":fail"
    edges = {}

    def add(a, b, kind):
        if a != b:
            edges.setdefault((a, b), set()).add(kind)
    succ = {}      # key -> {v1: set(v2)}
    versions = {}  # key -> set of non-nil versions known
    ext_reads = []  # (txn, key, value-or-None)
    for t in txns:
        if t["type"] != ":fail":
            for f, k, v in t["mops"]:
                if f == ":w":
                    versions.setdefault(k, set()).add(v)
        if t["type"] != ":ok":
            continue
        state = {}
        for f, k, v in t["mops"]:
            if f == ":w":
                state[k] = ("w", v)
                continue
            if k in state:
                if state[k][1] != v:
                    anomalies.add("internal")
                continue
            state[k] = ("r", v)
            ext_reads.append((t["id"], k, v))
            if v is None:
                continue
            versions.setdefault(k, set()).add(v)
            w = writer.get((k, v))
            if not live(w):
                anomalies.add("G1a")
                continue
            if w != t["id"]:
                if final[(w, k)] != v:
                    anomalies.add("G1b")
                add(w, t["id"], "wr")
            fw = final.get((t["id"], k))
            if fw is not None and fw != v:
                succ.setdefault(k, {}).setdefault(v, set()).add(fw)
    for k, vs in versions.items():
        succ.setdefault(k, {}).setdefault(None, set()).update(vs)
    for k in list(succ):
        g = succ[k]

        def reach(a, seen=None):
            seen = set() if seen is None else seen
            for b in g.get(a, ()):
                if b not in seen:
                    seen.add(b)
                    reach(b, seen)
            return seen
        if any(v in reach(v) for v in list(g)):
            anomalies.add("cyclic-versions")
            succ[k] = {}
            continue
        for v1, v2s in g.items():
            for v2 in v2s:
                if v1 is not None and live(writer.get((k, v1))) and live(writer.get((k, v2))):
                    add(writer[(k, v1)], writer[(k, v2)], "ww")
    for tid, k, v in ext_reads:
        for v2 in succ.get(k, {}).get(v, ()):
            if live(writer.get((k, v2))):
                add(tid, writer[(k, v2)], "rw")
    for t in txns:
        if t["type"] != ":fail":
            for f in rt_pred[t["id"]]:
                add(f, t["id"], "rt")

    def reachable(src, kinds):
        adj = {}
        for (a, b), ks in edges.items():
            if ks & kinds:
                adj.setdefault(a, []).append(b)
        seen, stack = set(), [src]
        while stack:
            v = stack.pop()
            for w in adj.get(v, ()):
                if w not in seen:
                    seen.add(w)
                    stack.append(w)
        return seen

    def cyclic(kinds):
        return any(a in reachable(a, kinds) for a in {a for (a, _b), ks in edges.items() if ks & kinds})

    def classify(extra):
        if cyclic({"ww"} | extra):
            return {"G0"}
        if cyclic({"ww", "wr"} | extra):
            return {"G1c"}
        if cyclic({"ww", "wr", "rw"} | extra):
            single = any("rw" in ks and a in reachable(b, {"ww", "wr"} | extra) for (a, b), ks in edges.items())
            return {"G-single"} if single else {"G2"}
        return set()
    c = classify(set())
    if not c:
        c = classify({"rt"})
        if c:
            c.add("realtime")
    anomalies |= c
    ok = sum(1 for t in txns if t["type"] == ":ok")
    return {"anomalies": anomalies, "txn-count": len(txns), "ok-count": ok}
