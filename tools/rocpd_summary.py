#!/usr/bin/env python3
"""Condenses a rocprofv3 results database into the small text summaries kept under profiles/.

    rocpd_summary.py kernels  RESULTS.db  > kernel_stats.csv     (--kernel-trace runs)
    rocpd_summary.py counters RESULTS.db  > pmc_summary.csv      (--pmc runs)

kernels : per kernel name: calls, total / average / min / max duration in ns, share of GPU time.
counters: per (kernel, counter): dispatches and the mean / min / max of the counter value.
"""
import csv
import re
import sqlite3
import sys


def short(name):
    """Kernel name without the argument list (template arguments stay)."""
    depth = 0
    for i, ch in enumerate(name):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            name = name[:i]
            break
    name = re.sub(r"^void ", "", name)
    return name if len(name) <= 120 else name[:117] + "..."


def main():
    mode, db = sys.argv[1], sys.argv[2]
    con = sqlite3.connect(db)
    out = csv.writer(sys.stdout)
    if mode == "kernels":
        rows = con.execute(
            "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
            "from kernels group by name order by 3 desc").fetchall()
        total = sum(r[2] for r in rows) or 1
        out.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
        for name, calls, tot, avg, mn, mx in rows:
            out.writerow([short(name), calls, tot, f"{avg:.1f}", mn, mx, f"{100.0 * tot / total:.2f}"])
    elif mode == "counters":
        rows = con.execute(
            "select kernel_name, counter_name, count(*), avg(value), min(value), max(value) "
            "from counters_collection group by 1, 2 order by 1, 2").fetchall()
        out.writerow(["Kernel", "Counter", "Dispatches", "Mean", "Min", "Max"])
        for name, ctr, n, avg, mn, mx in rows:
            out.writerow([short(name), ctr, n, f"{avg:.6g}", f"{mn:.6g}", f"{mx:.6g}"])
    else:
        sys.exit(__doc__)


if __name__ == "__main__":
    main()
