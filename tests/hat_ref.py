"""A direct Python transliteration of demo/clojure/txn_rw_register_hat.clj — written from the Clojure source, independently
of oracle/hat_nodes.inc (dicts and sets as in the original; no txn table, no bit masks, no lazy timers) — to cross-check the
oracle's restatement of that node by replaying the oracle's own network schedule through it.  Test infrastructure only.

Where the reference leaves a choice to Clojure's hash order the engine's canonical choice is followed:
`(first uts)` = the unreplicated txn that was created first, `(first (:nodes txn+))` = the lowest node id."""


class HatNode:
    def __init__(self, node_id, node_ids, created):
        self.id, self.node_ids = node_id, list(node_ids)
        self.lamport, self.kv = 0, {}            # state atom, txn_rw_register_hat.clj:22-28
        self.unreplicated = {}                   # :30-33  ts -> {"ts", "txn", "nodes"}
        self.created = created                   # shared: ts -> creation rank over the whole cluster (canonical `first`)

    def other_node_ids(self):                    # node.clj:87-90
        return set(self.node_ids) - {self.id}

    def apply_txn(self, txn_plus):               # apply-txn+, :34-74
        ts = txn_plus.get("ts")
        if ts is not None:
            lamport2 = max(self.lamport, ts[0] + 1)
        else:
            ts = (self.lamport, self.id)
            lamport2 = self.lamport + 1
        out = []
        for f, k, v in txn_plus["txn"]:
            cur = self.kv.get(k)
            if f == "r":
                out.append(["r", k, cur["value"] if cur else None])
            else:
                if not (cur and tuple(cur["ts"]) > tuple(ts)):   # (pos? (compare (:ts current-value) ts)) => leave in place
                    self.kv[k] = {"ts": ts, "value": v}
                out.append([f, k, v])
        self.lamport = lamport2
        return dict(txn_plus, ts=ts, txn=out)

    def on_txn(self, txn):                       # :120-130
        tp = self.apply_txn({"txn": txn})
        if tp["ts"] not in self.created:
            self.created[tp["ts"]] = len(self.created)
        self.unreplicated[tp["ts"]] = dict(tp, nodes=self.other_node_ids())   # later-replicate!, :85-90
        return tp["txn"]

    def replicate_step(self):                    # :92-105; returns (dest, [txn+ ...]) or None
        if not self.unreplicated:
            return None
        first = min(self.unreplicated.values(), key=lambda t: self.created[t["ts"]])
        node = min(first["nodes"])
        return node, [dict(t, nodes=set(t["nodes"])) for t in self.unreplicated.values() if node in t["nodes"]]

    def on_replicate(self, txns):                # :132-150; returns [(dest, tss) ...]
        for tp in txns:
            self.apply_txn(tp)
            rest = set(tp["nodes"]) - {self.id}
            if rest:
                self.unreplicated[tp["ts"]] = dict(tp, nodes=rest)
        tss = [tp["ts"] for tp in txns]
        return [(n, tss) for n in sorted(self.other_node_ids())]

    def on_replicate_ack(self, node, tss):       # :152-172
        for ts in tss:
            tp = self.unreplicated.get(ts)
            if tp is None:
                continue
            rest = tp["nodes"] - {node}
            if rest:
                self.unreplicated[ts] = dict(tp, nodes=rest)
            else:
                del self.unreplicated[ts]
