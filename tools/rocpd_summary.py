#!/usr/bin/env python3
"""Summarises a rocprofv3 rocpd (sqlite) result: per-kernel dispatch statistics (what `--stats` prints)
and per-kernel PMC counter sums.  Usage: tools/rocpd_summary.py <results.db> [more.db ...]"""
import sqlite3
import sys


def table(con, prefix):
    return [r[0] for r in con.execute("select name from sqlite_master where type='table'") if r[0].startswith(prefix)][0]


def summarise(path):
    con = sqlite3.connect(path)
    kd, ks = table(con, "rocpd_kernel_dispatch"), table(con, "rocpd_info_kernel_symbol")
    print(f"== {path}")
    print(f"{'kernel':60s} {'calls':>6s} {'total_ms':>10s} {'avg_ms':>10s} {'min_ms':>10s} {'max_ms':>10s} {'lds':>6s} {'vgpr':>5s} {'sgpr':>5s}")
    q = (f"select s.display_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), "
         f"max(d.group_segment_size), max(s.arch_vgpr_count), max(s.sgpr_count) from '{kd}' d join '{ks}' s on d.kernel_id = s.id "
         f"group by s.display_name order by 3 desc")
    for name, n, tot, avg, mn, mx, lds, vg, sg in con.execute(q):
        print(f"{name[:60]:60s} {n:6d} {tot/1e6:10.3f} {avg/1e6:10.4f} {mn/1e6:10.4f} {mx/1e6:10.4f} {lds:6d} {vg:5d} {sg:5d}")
    try:  # private (scratch) memory per work-item, where the schema has it
        for name, sc in con.execute(f"select s.display_name, max(d.private_segment_size) from '{kd}' d join '{ks}' s on d.kernel_id = s.id group by 1"):
            if sc:
                print(f"   scratch bytes/work-item: {name[:60]:60s} {sc}")
    except Exception:
        pass
    try:
        pe, pi = table(con, "rocpd_pmc_event"), table(con, "rocpd_info_pmc")
        q = (f"select s.display_name, i.name, count(*), sum(e.value), avg(e.value) from '{pe}' e join '{pi}' i on e.pmc_id = i.id "
             f"join '{kd}' d on e.event_id = d.event_id join '{ks}' s on d.kernel_id = s.id group by 1, 2 order by 1, 2")
        rows = list(con.execute(q))
        if rows:
            print("-- PMC counters (sum over dispatches / per-dispatch average)")
            for name, ctr, n, tot, avg in rows:
                print(f"{name[:60]:60s} {ctr:24s} n={n:4d} sum={tot:.6g} avg={avg:.6g}")
    except Exception as e:  # no counters collected in this pass
        print(f"-- no PMC data ({e.__class__.__name__}: {e})")


OURS = ("sim_kernel", "check_kernel", "raft_kernel", "raft4_kernel", "txn_kernel", "txn8_kernel", "mk_kernel", "mk8_kernel", "hat_kernel", "svc_kernel", "svc4_kernel", "compact_", "availability_kernel",
        "txn_check_kernel", "rw_check_kernel", "lin_check_kernel", "lin_check_wg_kernel", "unique_check_kernel", "pn_check_kernel", "kafka_kernel", "kafka8_kernel", "kafka_check_kernel",
        "hat8_kernel", "uid8_kernel", "crdt8_kernel", "bcast8_kernel", "dt_kernel", "dt8_kernel", "txn_check_lds_kernel")


def counters_json(paths, out):
    """--counters OUT.json: per-kernel averages per dispatch of every PMC counter collected (all passes), plus the dispatch
    statistics of the kernel trace, for the engine's own kernels — what bench.py's roofline.secondary quotes."""
    import json
    acc = {}
    for path in paths:
        con = sqlite3.connect(path)
        try:
            kd, ks = table(con, "rocpd_kernel_dispatch"), table(con, "rocpd_info_kernel_symbol")
        except IndexError:
            continue
        q = (f"select s.display_name, count(*), avg(d.end-d.start), max(d.group_segment_size), max(s.arch_vgpr_count), max(s.sgpr_count), max(d.grid_size_x), max(d.workgroup_size_x) "
             f"from '{kd}' d join '{ks}' s on d.kernel_id = s.id group by 1")
        try:
            rows = list(con.execute(q))
        except sqlite3.OperationalError:
            q = q.replace(", max(d.grid_size_x), max(d.workgroup_size_x)", ", 0, 0")
            rows = list(con.execute(q))
        for name, nd, avg, lds, vg, sg, gx, wx in rows:
            if any(k in name for k in OURS):
                a = acc.setdefault(name, {"counters_per_dispatch": {}})
                a.setdefault("dispatches_seen", []).append(nd)
                a["avg_ms"] = avg / 1e6 if "avg_ms" not in a else min(a["avg_ms"], avg / 1e6)   # the trace pass is the fastest (no counters)
                a["lds_bytes"], a["arch_vgpr"], a["sgpr"] = lds, vg, sg
                if gx and wx:
                    a["wavefronts"] = gx // 64
        try:
            pe, pi = table(con, "rocpd_pmc_event"), table(con, "rocpd_info_pmc")
        except IndexError:
            continue
        q = (f"select s.display_name, i.name, sum(e.value), count(distinct d.id) from '{pe}' e join '{pi}' i on e.pmc_id = i.id "
             f"join '{kd}' d on e.event_id = d.event_id join '{ks}' s on d.kernel_id = s.id group by 1, 2")
        for name, ctr, tot, nd in con.execute(q):
            if any(k in name for k in OURS):
                acc.setdefault(name, {"counters_per_dispatch": {}})["counters_per_dispatch"][ctr] = tot / max(nd, 1)
    for name, a in acc.items():
        c = a["counters_per_dispatch"]
        if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
            wc = c["SQ_WAVE_CYCLES"]
            a["derived"] = {k2: c[k1] / wc for k1, k2 in (("SQ_ACTIVE_INST_VALU", "valu_active_frac_of_wave_cycles"), ("SQ_INST_CYCLES_SALU", "salu_frac_of_wave_cycles"),
                                                           ("SQ_WAIT_ANY", "wait_any_frac_of_wave_cycles"), ("SQ_WAIT_INST_ANY", "wait_inst_frac_of_wave_cycles"),
                                                           ("SQ_ACTIVE_INST_ANY", "any_inst_active_frac_of_wave_cycles")) if k1 in c}
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            a["hbm_bytes_per_dispatch"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0   # see traffic_json
    json.dump({"source": "rocprofv3 --kernel-trace --stats + separate --pmc passes (tools/profile_headline.sh / tools/profile_config.sh)", "kernels": acc}, open(out, "w"), indent=1)


def traffic_json(paths, out):
    """--traffic OUT.json: per-kernel HBM traffic per dispatch from the FETCH_SIZE / WRITE_SIZE passes, corrected as
    /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for gfx950: both counters are in KiB; FETCH_SIZE counts
    128-B read requests as 64 B, so it is doubled; WRITE_SIZE is taken as is (calibrated here against the kernel's known
    write volume: WRITE_SIZE x 1024 = 1.01 x the algorithmic bytes)."""
    import json
    acc = {}
    for path in paths:
        con = sqlite3.connect(path)
        try:
            kd, ks = table(con, "rocpd_kernel_dispatch"), table(con, "rocpd_info_kernel_symbol")
            pe, pi = table(con, "rocpd_pmc_event"), table(con, "rocpd_info_pmc")
        except IndexError:
            continue
        q = (f"select s.display_name, i.name, sum(e.value), count(distinct d.id) from '{pe}' e join '{pi}' i on e.pmc_id = i.id "
             f"join '{kd}' d on e.event_id = d.event_id join '{ks}' s on d.kernel_id = s.id where i.name in ('FETCH_SIZE', 'WRITE_SIZE') group by 1, 2")
        for name, ctr, tot, nd in con.execute(q):
            if not any(k in name for k in OURS):
                continue  # the engine's own kernels only (torch's reductions in bench.py are not the path)
            acc.setdefault(name, {})[ctr + "_KiB_per_dispatch"] = tot / max(nd, 1)
    for name, d in acc.items():
        f, w = d.get("FETCH_SIZE_KiB_per_dispatch"), d.get("WRITE_SIZE_KiB_per_dispatch")
        if f is not None and w is not None:
            d["hbm_bytes_per_dispatch"] = (2.0 * f + w) * 1024.0
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), tools/profile_headline.sh", "kernels": acc}, open(out, "w"), indent=1)


if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0] == "--traffic":
        traffic_json(args[2:], args[1])
    elif args and args[0] == "--counters":
        counters_json(args[2:], args[1])
    else:
        for p in args:
            summarise(p)
