"""Reads a rocprofv3 kernel trace (CSV) of a `gcn` run and reports whether the test-cube gather,
issued on the side stream, ran while training kernels ran on the training stream.
    trace_overlap.py TRACE_DIR > summary"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
t0 = rows[0][0]
short = lambda n: n.replace("void ccnn::", "").split("<")[0]
gathers = [r for r in rows if "gather_tile" in r[2]]
train = [r for r in rows if "sample_fwd_bwd" in r[2] or "grad_reduce" in r[2]]
print(f"{len(rows)} kernel dispatches; queues used: {sorted({r[3] for r in rows})}; streams: {sorted({r[4] for r in rows})}")
for g in gathers:
    over = [t for t in train if t[0] < g[1] and t[1] > g[0]]
    print(f"gather_tile on queue {g[3]} stream {g[4]}: {(g[0]-t0)/1e3:9.1f} .. {(g[1]-t0)/1e3:9.1f} us "
          f"({(g[1]-g[0])/1e3:.1f} us); training kernels running meanwhile: {len(over)}"
          + (f" (queue {over[0][3]} stream {over[0][4]}, first at {(over[0][0]-t0)/1e3:.1f} us)" if over else ""))
print("first 14 dispatches (start us, duration us, queue, stream, kernel):")
for r in rows[:14]:
    print(f"  {(r[0]-t0)/1e3:9.1f} {(r[1]-r[0])/1e3:7.1f}  q{r[3]} s{r[4]}  {short(r[2])}")
