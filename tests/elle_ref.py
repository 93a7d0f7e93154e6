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
