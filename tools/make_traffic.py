#!/usr/bin/env python3
"""Writes profiles/traffic.json from the PMC summaries of one profile directory.

    make_traffic.py profiles/r3a_bench

Reads pmc_hbm.csv (the headline run: Pavia-U shape, B = 128), pmc_hbm_b2048.csv, pmc_hbm_f32_b512.csv
and pmc_hbm_f64_b100.csv (the canonical fp64 trainer's batch) as tools/profile_bench.sh leaves them, and stamps the digest of the training kernels' sources
(bench.py: kernel_source_digest) so that bench.py reports `roofline.traffic` only for the build
that was profiled.  hbm_bytes_per_launch = (2 FETCH_SIZE + WRITE_SIZE) KiB: gfx950 tallies a
128-byte read request as 64 bytes, which over-counts narrow loads - an upper bound."""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def digest():
    import hashlib
    h = hashlib.sha256()
    for name in ("mbgd.hip", "tile.hip", "forward.hip", "ccnn_device.h"):      # bench.py: KERNEL_SOURCES
        with open(os.path.join(ROOT, "gcn_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def counters(path):
    out = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            out.setdefault(row["Kernel"], {})[row["Counter"]] = float(row["Mean"])
    return out


def main():
    src = sys.argv[1].rstrip("/")
    records = []
    small = ("sample_fwd_bwd", "grad_reduce", "step_fused")
    large = ("tile_fwd_bwd", "grad_reduce")
    for fname, batch, dtype, kernels in (("pmc_hbm.csv", 128, "f32", small), ("pmc_hbm_b2048.csv", 2048, "f32", large),
                                         ("pmc_hbm_f64_b100.csv", 100, "f64", small), ("pmc_hbm_f32_b512.csv", 512, "f32", small)):
        path = os.path.join(src, fname)
        if not os.path.exists(path):
            continue
        for full, c in counters(path).items():
            m = re.match(r"ccnn::(\w+)<" + ("float" if dtype == "f32" else "double"), full)
            if not m or m.group(1) not in kernels or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
                continue
            records.append({"workload": "pavia_u", "batch": batch, "dtype": dtype, "kernel": m.group(1), "instance": full,
                            "fetch_size_kib": c["FETCH_SIZE"], "write_size_kib": c["WRITE_SIZE"],
                            "hbm_bytes_per_launch": int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)})
    doc = {
        "note": "HBM-side bytes per launch from rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE (separate passes, values in "
                "KiB). hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 tallies 128-byte read requests "
                "at 64 bytes; the doubling over-counts dword-wide loads, so the figure is an upper bound.",
        "command": "tools/profile_bench.sh, then tools/make_traffic.py " + os.path.relpath(src, ROOT),
        "summaries": os.path.relpath(src, ROOT) + "/",
        "kernel_source_digest": digest(),
        "records": records,
    }
    with open(os.path.join(ROOT, "profiles", "traffic.json"), "w") as f:
        json.dump(doc, f, indent=2)
        f.write("\n")
    print(f"{len(records)} records, kernels @{doc['kernel_source_digest']}")


if __name__ == "__main__":
    main()
