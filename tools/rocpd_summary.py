#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database: per-kernel count / total / average duration (the same columns as
`rocprofv3 --stats`' kernel table) and, when PMC counters were collected, their per-dispatch averages.
usage: rocpd_summary.py out_results.db [name-filter]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    T = lambda p: [t for t in tabs if t.startswith(p)][0]
    kd, ks = T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol")
    rows = list(c.execute(f"select s.display_name, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start), "
                          f"max(s.arch_vgpr_count), max(s.sgpr_count), max(d.group_segment_size), max(d.workgroup_size_x), max(d.grid_size_x) "
                          f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.display_name order by 3 desc"))
    tot = sum(r[2] for r in rows) or 1
    print(f"{'kernel':96s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'%':>6s} vgpr sgpr  lds  wg    grid")
    for name, n, t, mn, mx, vg, sg, lds, wg, grid in rows:
        if flt and flt not in name:
            continue
        short = name.split("(")[0]   # (whole: the template arguments tell the instantiations apart, and tools/traffic_from_pmc.py matches on "vnm::")
        print(f"{short:96s} {n:6d} {t / 1e6:10.3f} {t / n / 1e3:10.1f} {mn / 1e3:9.1f} {mx / 1e3:9.1f} {100.0 * t / tot:6.2f} {vg:4d} {sg:4d} {lds:5d} {wg:4d} {grid:7d}")
    try:
        pe, pi = T("rocpd_pmc_event"), T("rocpd_info_pmc")
        q = (f"select s.display_name, p.name, count(*), avg(e.value) from {pe} e join {pi} p on e.pmc_id = p.id "
             f"join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id group by s.display_name, p.name order by 1")
        pm = list(c.execute(q))
        if pm:
            print("\nPMC (average per dispatch)")
            for name, ctr, n, v in pm:
                if flt and flt not in name:
                    continue
                print(f"{name.split('(')[0]:96s} {ctr:14s} n={n:4d} avg={v:.6g}")
    except Exception as e:  # no counters in this run
        pass


if __name__ == "__main__":
    main()
