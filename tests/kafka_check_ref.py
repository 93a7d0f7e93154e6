"""An independent restatement of the kafka checker's anomalies (workload/kafka.clj:21-70) over decoded op maps — the test-side twin of
csrc/kafka_check.cpp (both by the same author; [upstream] jepsen.tests.kafka is not vendored: PARITY UNPINNED)."""


def check(ops):
    ops = [op for op in ops if op["process"] != ":nemesis"]
    log = {}          # key -> {offset: set of messages}
    where = {}        # key -> {message: set of offsets}
    acked, polled, failed = {}, {}, {}
    anomalies = set()
    for op in ops:
        if op["f"] == ":send":
            _, k, v = op["value"][0]
            if op["type"] == ":ok":
                off, msg = v
                log.setdefault(k, {}).setdefault(off, set()).add(msg)
                where.setdefault(k, {}).setdefault(msg, set()).add(off)
                acked.setdefault(k, set()).add(off)
            elif op["type"] == ":fail":
                failed.setdefault(k, set()).add(v)
        elif op["f"] == ":poll" and op["type"] == ":ok":
            for k, pairs in (op["value"][0][1] if len(op["value"][0]) > 1 else {}).items():
                for off, msg in pairs:
                    log.setdefault(k, {}).setdefault(off, set()).add(msg)
                    where.setdefault(k, {}).setdefault(msg, set()).add(off)
                    polled.setdefault(k, set()).add(off)
    if any(len(ms) > 1 for k in log for ms in log[k].values()):
        anomalies.add("inconsistent-offsets")
    dups = sum(len(offs) - 1 for k in where for offs in where[k].values() if len(offs) > 1)
    if dups:
        anomalies.add("duplicate")
    lost = unobserved = 0
    for k, offs in acked.items():
        top = max(polled.get(k, {-1}))
        for o in offs - polled.get(k, set()):
            if o < top:
                lost += 1
            else:
                unobserved += 1
    if lost:
        anomalies.add("lost-write")
    for k, offs in polled.items():
        if any(m in failed.get(k, set()) for o in offs for m in log[k][o]):
            anomalies.add("aborted-read")
    last_poll, last_send = {}, {}
    for op in ops:
        p = op["process"]
        if (op["f"] == ":assign" and op["type"] == ":invoke") or (op["f"] == ":poll" and op["type"] in (":fail", ":info")):
            for key in [kk for kk in last_poll if kk[0] == p]:
                del last_poll[key]
        if op["type"] != ":ok":
            continue
        if op["f"] == ":send":
            _, k, (off, _msg) = op["value"][0]
            if (p, k) in last_send and off <= last_send[(p, k)]:
                anomalies.add("nonmonotonic-send")
            last_send[(p, k)] = off
        elif op["f"] == ":poll" and len(op["value"][0]) > 1:
            for k, pairs in op["value"][0][1].items():
                seen_in_this_poll = False
                for off, _msg in pairs:
                    prev = last_poll.get((p, k))
                    if prev is not None and off != prev + 1:
                        pre = "int-" if seen_in_this_poll else ""
                        if off <= prev:
                            anomalies.add(pre + "nonmonotonic-poll")
                        elif any(o in log.get(k, {}) for o in range(prev + 1, off)):
                            anomalies.add(pre + "poll-skip")
                    last_poll[(p, k)] = off
                    seen_in_this_poll = True
    n_acked = sum(1 for op in ops if op["f"] == ":send" and op["type"] == ":ok")
    n_ok = sum(1 for op in ops if op["type"] == ":ok")
    return {"valid?": False if anomalies else ("unknown" if n_acked == 0 and n_ok == 0 else True), "anomalies": sorted(anomalies),
            "lost-count": lost, "unobserved-count": unobserved, "duplicate-count": dups, "acked-count": n_acked}
