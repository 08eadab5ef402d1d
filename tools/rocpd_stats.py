#!/usr/bin/env python3
"""Per-kernel statistics out of a rocprofv3 results.db (rocpd schema): name, calls, average / min / max duration in µs,
registers, LDS.  usage: rocpd_stats.py results.db [csv_out]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
rows = list(db.execute("select name, count(*), avg(duration), min(duration), max(duration), sum(duration) from kernels group by name order by sum(duration) desc"))
lines = ["name,calls,avg_us,min_us,max_us,total_us"]
for name, n, avg, mn, mx, tot in rows:
    short = name.split("(")[0].replace("void ", "")
    lines.append("%s,%d,%.2f,%.2f,%.2f,%.2f" % (short, n, avg / 1e3, mn / 1e3, mx / 1e3, tot / 1e3))
text = "\n".join(lines)
print(text)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text + "\n")
