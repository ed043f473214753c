"""Reads a rocprofv3 kernel trace (CSV) of a `gcn DATASET.mat n master` run whose device slots share
one GPU (CCNN_CLI_DEVICE_MAP) and reports whether the slots' training steps were in flight together:
per stream the number of per-sample kernels, how many of them ran while a per-sample kernel of
ANOTHER stream ran, and the order in which the streams' first steps were dispatched.
    trace_slots.py TRACE_DIR > summary"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0],
                         r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
t0 = rows[0][0]
short = lambda n: n.replace("void ccnn::", "").replace("ccnn::", "").split("<")[0]
fb = [r for r in rows if "sample_fwd_bwd" in r[2] or "tile_fwd_bwd" in r[2]]
streams = sorted({r[4] for r in fb})
print(f"{len(rows)} kernel dispatches; per-sample training kernels on streams {streams} (queues {sorted({r[3] for r in fb})})")
for s in streams:
    mine = [r for r in fb if r[4] == s]
    others = [r for r in fb if r[4] != s]
    j, over = 0, 0
    for r in mine:
        while j < len(others) and others[j][1] <= r[0]:
            j += 1
        k = j
        while k < len(others) and others[k][0] < r[1]:
            if others[k][1] > r[0]:
                over += 1
                break
            k += 1
    span = (mine[-1][1] - mine[0][0]) / 1e3
    print(f"stream {s}: {len(mine)} per-sample kernels between {(mine[0][0]-t0)/1e3:.1f} and {(mine[-1][1]-t0)/1e3:.1f} us "
          f"({span / len(mine):.2f} us per step); {over} of them ran while another slot's per-sample kernel ran")
first = fb[:16]
print("the first 16 per-sample kernels (start us, duration us, stream): " + "  ".join(f"{(r[0]-t0)/1e3:.1f}/{(r[1]-r[0])/1e3:.1f}/s{r[4]}" for r in first))
tr = [r for r in rows if r[0] >= fb[0][0]][:24]
print("dispatches from the first training kernel on (start us, duration us, queue, stream, kernel):")
for r in tr:
    print(f"  {(r[0]-t0)/1e3:9.1f} {(r[1]-r[0])/1e3:7.1f}  q{r[3]} s{r[4]}  {short(r[2])}")
