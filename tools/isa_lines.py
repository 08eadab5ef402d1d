#!/usr/bin/env python3
"""Static instruction counts per source line for one kernel of an ISA listing made with
`hipcc --cuda-device-only -S -gline-tables-only`: every instruction is charged to the innermost tp_fast.h line of
its inline chain that lies in one of the given line ranges (the bodies of the main functions), so helper calls count
at their call sites.  usage: isa_lines.py k.s kernel_symbol_substring lo-hi[,lo-hi...] [bucket_size]"""
import collections
import re
import sys

path, sym = sys.argv[1], sys.argv[2]
import os
FILE = os.environ.get("ISA_FILE", "tp_fast.h")
ranges = [tuple(int(x) for x in r.split("-")) for r in sys.argv[3].split(",")]
bucket = int(sys.argv[4]) if len(sys.argv) > 4 else 1
inside = False
cur = None
per = collections.defaultdict(collections.Counter)
tot = collections.Counter()
for ln in open(path):
    s = ln.strip()
    if not inside:
        if s.startswith("_Z") and ":" in s and sym in s.split(":")[0]:
            inside = True
        continue
    if s.startswith(".end_amdhsa_kernel"):
        break
    if s.startswith(".loc"):
        chain = re.findall(r"([A-Za-z0-9_./-]+):(\d+):\d+", s.split(";", 1)[1] if ";" in s else "")
        cur = None
        for f, l in chain:
            l = int(l)
            if f.endswith(FILE) and any(a <= l <= b for a, b in ranges):
                cur = l
                break
        if cur is None:
            cur = -1
        continue
    if not s or s.startswith(".") or s.startswith(";") or s.endswith(":"):
        continue
    op = s.split()[0]
    kind = "v" if op.startswith("v_") else "s" if op.startswith("s_") else "ds" if op.startswith("ds_") else "mem"
    key = cur if cur is None or cur < 0 else (cur // bucket) * bucket
    per[key][kind] += 1
    tot[kind] += 1
print("totals", dict(tot))
for k in sorted(per, key=lambda x: (x is None, x)):
    c = per[k]
    print("%6s  v %5d  s %5d  ds %4d  mem %4d" % (k, c["v"], c["s"], c["ds"], c["mem"]))
