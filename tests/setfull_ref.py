"""Pure-Python restatement of [upstream] jepsen.checker/set-full (+ the echo checker of
workload/echo.clj:44-63) over decoded op maps.  TEST INFRASTRUCTURE: the checker of record is the
device kernel in maelstrom_amd/csrc/checker.hip; this slow version cross-checks it on small histories.

Result-map shape: doc/03-broadcast/01-broadcast.md:564-577 (KAT-7)."""
import math


def set_full(history, linearizable=False):
    elements = {}  # value -> dict(known, last_present, last_absent)
    reads = {}
    dups = {}
    for op in history:
        if not isinstance(op["process"], int):
            continue
        f, typ, v, p = op["f"], op["type"], op.get("value"), op["process"]
        if f in (":add", ":broadcast"):
            if typ == ":invoke":
                elements[v] = {"element": v, "known": None, "last_present": None, "last_absent": None}
            elif typ == ":ok":
                e = elements[v]
                e["known"] = e["known"] or op
        elif f == ":read":
            if typ == ":invoke":
                reads[p] = op
            elif typ == ":fail":
                reads.pop(p, None)
            elif typ == ":ok":
                inv = reads[p]
                seen = {}
                for x in v:
                    seen[x] = seen.get(x, 0) + 1
                for x, c in seen.items():
                    if c > 1:
                        dups[x] = max(dups.get(x, 0), c)
                vs = set(v)
                for el, e in elements.items():
                    if el in vs:
                        e["known"] = e["known"] or op
                        if e["last_present"] is None or e["last_present"]["index"] < inv["index"]:
                            e["last_present"] = inv
                    else:
                        if e["last_absent"] is None or e["last_absent"]["index"] < inv["index"]:
                            e["last_absent"] = inv
    results = []
    for e in elements.values():
        k, lp, la = e["known"], e["last_present"], e["last_absent"]
        stable = bool(lp and (la["index"] if la else -1) < lp["index"])
        lost = bool(k and la and (lp["index"] if lp else -1) < la["index"] and k["index"] < la["index"])
        r = {"element": e["element"], "outcome": "stable" if stable else "lost" if lost else "never-read"}
        if stable:
            stable_time = la["time"] + 1 if la else 0
            r["stable_latency"] = int(max(0, stable_time - k["time"]) / 1e6)
        results.append(r)
    stable = [r for r in results if r["outcome"] == "stable"]
    lost = [r for r in results if r["outcome"] == "lost"]
    never = [r for r in results if r["outcome"] == "never-read"]
    stale = [r for r in stable if r["stable_latency"] > 0]
    lats = sorted(r["stable_latency"] for r in stable)
    pts = [0, 0.5, 0.95, 0.99, 1]
    dist = {q: lats[min(len(lats) - 1, int(math.floor(len(lats) * q)))] for q in pts} if lats else None
    valid = False if lost else ("unknown" if not stable else (False if (linearizable and stale) else True))
    return {"valid?": valid, "attempt-count": len(results), "stable-count": len(stable), "lost-count": len(lost),
            "lost": sorted(r["element"] for r in lost), "never-read-count": len(never),
            "never-read": sorted(r["element"] for r in never), "stale-count": len(stale),
            "stale": sorted(r["element"] for r in stale), "stable-latencies": dist,
            "duplicated-count": len(dups), "duplicated": dups}


def echo_check(history):
    pending, errs = {}, []
    for op in history:
        if op["type"] == ":invoke":
            pending[op["process"]] = op
        elif op["process"] in pending:
            inv = pending.pop(op["process"])
            # echo.clj:52-60: (not= (:value invoke) (:echo (:value complete))) — an :info / :fail completion carries the request
            # string as :value, so (:echo ...) is nil and the pair is an error
            got = op["value"].get("echo") if (op["type"] == ":ok" and isinstance(op["value"], dict)) else None
            if inv["value"] != got:
                errs.append((inv["value"], op["value"]))
    for inv in pending.values():   # pair-index maps an invocation without completion to nil
        errs.append((inv["value"], None))
    return {"valid?": not errs, "errors": errs}
