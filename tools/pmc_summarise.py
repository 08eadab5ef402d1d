#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel.

usage: pmc_summarise.py OUT.json WORKLOAD DIR [DIR...]

Every DIR is the output directory of one `rocprofv3 --pmc ... -- python3 bench.py ...` pass.  For each kernel
name the counter values are averaged over its dispatches (one launch = one dispatch).  FETCH_SIZE and
WRITE_SIZE are reported by rocprofv3 in KiB; on gfx950 FETCH_SIZE tallies the 128-byte requests of wide
coalesced reads at 64 bytes, so the fetch figure is doubled (upper bound for kernels that also issue narrow
reads) before the two are added into hbm_bytes_per_launch of the dominant kernel.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    """'void tp::k_scan_tiles<unsigned int>(...)' -> 'k_scan_tiles<unsigned int>'"""
    name = name.split("(")[0]
    if name.startswith("void "):
        name = name[5:]
    return name[4:] if name.startswith("tp::") else name


def main():
    out_path, workload, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    per = defaultdict(lambda: defaultdict(list))   # kernel -> counter -> values
    for d in dirs:
        for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(fn, newline="") as f:
                for row in csv.DictReader(f):
                    per[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for fn in glob.glob(os.path.join(d, "**", "*results.db"), recursive=True):
            import sqlite3
            db = sqlite3.connect(fn)
            for name, cname, value in db.execute("select kernel_name, counter_name, value from counters_collection"):
                per[short(name)][cname].append(float(value))
            db.close()
    kernels = {}
    for k, cs in per.items():
        if not k.startswith("k_"):
            continue
        kernels[k] = {c: {"avg": sum(v) / len(v), "launches": len(v)} for c, v in cs.items()}
    # HBM bytes of one step: every kernel's (2 x FETCH + WRITE) x launches, over the number of steps (the caption writer runs once a step)
    steps = 0
    for once_a_step in ("k_emit_runs", "k_emit", "k_lean_a", "k_fast"):
        steps = kernels.get(once_a_step, {}).get("FETCH_SIZE", {}).get("launches", 0)
        if steps:
            break
    if steps:
        total = 0.0
        per_kernel = {}
        for k, cs in kernels.items():
            if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
                b = (2.0 * cs["FETCH_SIZE"]["avg"] * cs["FETCH_SIZE"]["launches"] + cs["WRITE_SIZE"]["avg"] * cs["WRITE_SIZE"]["launches"]) * 1024.0 / steps
                per_kernel[k] = b
                total += b
    dom = max(per_kernel, key=per_kernel.get) if steps and per_kernel else sorted(kernels)[0]
    kd = kernels[dom]
    res = {"workload": workload, "kernel": dom, "kernels": kernels}
    if steps and per_kernel:
        res["hbm_bytes_per_step"] = total
        res["hbm_bytes_per_step_by_kernel"] = per_kernel
        res["steps"] = steps
    if "FETCH_SIZE" in kd and "WRITE_SIZE" in kd:
        fetch_raw = kd["FETCH_SIZE"]["avg"] * 1024.0
        write = kd["WRITE_SIZE"]["avg"] * 1024.0
        res["fetch_bytes_raw"] = fetch_raw
        res["fetch_bytes_corrected"] = 2.0 * fetch_raw
        res["write_bytes"] = write
        res["hbm_bytes_per_launch"] = 2.0 * fetch_raw + write
        res["note"] = "FETCH_SIZE KiB x1024 x2 (gfx950 wide-read correction) + WRITE_SIZE KiB x1024, averaged over launches of " + dom
    with open(out_path, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print(json.dumps({k: v for k, v in res.items() if k != "kernels"}))


if __name__ == "__main__":
    main()
