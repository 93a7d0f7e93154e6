"""Pure-Python linearizability checker for a CAS register, per key (what [upstream] Knossos does for
jepsen.tests.linearizable-register via `independent/checker`).  TEST INFRASTRUCTURE.

Just-in-time linearization (Lowe / Knossos "linear"): walk the history; keep the set of configurations
(register value, frozenset of pending ops already linearized).  :fail ops never happened; :info ops stay
pending forever (they may take effect at any later time, or never)."""


def _step(state, op):
    """Apply op to the register; returns (ok, new_state)."""
    f, v = op["f"], op["value"][1]
    if f == ":read":
        return (v is None and op["type"] != ":ok") or state == v, state   # an :ok read must see the current value
    if f == ":write":
        return True, v
    a, b = v
    return state == a, b if state == a else state


def check_key(ops):
    """ops: history of one key, in order (maps with type/f/process/value; read :ok carries the value read)."""
    return check_key_configs(ops)[0]


def check_key_configs(ops):
    """-> (linearizable?, the register values of the configurations the search ends with: Knossos' :configs, e.g.
    `#knossos.model.CASRegister{:value 3}` at doc/06-raft/01-key-value.md:145)."""
    # pair invokes with completions; drop failed ops entirely
    comp = {}
    open_by_proc = {}
    for i, op in enumerate(ops):
        if op["type"] == ":invoke":
            open_by_proc[op["process"]] = i
        else:
            comp[open_by_proc.pop(op["process"])] = i
    events = []   # (kind, id) kind: 'call' / 'ret'
    eff = {}      # id -> op as it should be applied
    for i, op in enumerate(ops):
        if op["type"] == ":invoke":
            c = comp.get(i)
            if c is not None and ops[c]["type"] == ":fail":
                continue
            events.append(("call", i))
            eff[i] = dict(ops[c]) if c is not None and ops[c]["type"] == ":ok" else dict(op, type=":info")
            eff[i]["f"] = op["f"]
            if eff[i]["type"] != ":ok" and op["f"] == ":read":
                eff[i]["skip"] = True      # an unfinished read constrains nothing
        elif op["type"] == ":ok":
            inv = [k for k, c in comp.items() if c == i][0]
            events.append(("ret", inv))
    configs = {(None, frozenset())}
    pending = set()
    for kind, i in events:
        if kind == "call":
            pending.add(i)
            continue
        # op i returns: every surviving configuration must have linearized it by now
        out = set()
        seen = set(configs)
        stack = list(configs)
        while stack:
            state, lin = stack.pop()
            if i in lin:
                out.add((state, lin - {i}))
                continue
            for j in pending - lin:
                if eff[j].get("skip"):
                    continue
                ok, st2 = _step(state, eff[j])
                if ok:
                    c2 = (st2, lin | {j})
                    if c2 not in seen:
                        seen.add(c2)
                        stack.append(c2)
        pending.discard(i)
        configs = out
        if not configs:
            return False, []
    return True, sorted({st for st, _ in configs}, key=lambda x: (x is None, x))


def check(history):
    """history: decoded lin-kv ops ({:f :read/:write/:cas :value [k ...]}); returns {key: bool}."""
    by_key = {}
    for op in history:
        if op["process"] == ":nemesis" or op["f"] not in (":read", ":write", ":cas"):
            continue
        by_key.setdefault(op["value"][0], []).append(op)
    return {k: check_key(v) for k, v in by_key.items()}
