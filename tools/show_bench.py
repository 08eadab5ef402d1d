#!/usr/bin/env python3
"""One line per bench record: files/s, kernel times, files in the general kernel.  usage: show_bench.py bench.json ..."""
import json
import sys
for p in sys.argv[1:]:
    d = json.load(open(p))
    print(p, round(d["value"] / 1e6, 1), "M files/s", {k: round(v, 3) for k, v in d["config"]["kernel_ms"].items()}, "general", d["config"]["files_in_general_kernel"])
