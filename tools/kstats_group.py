#!/usr/bin/env python
"""Groups a rocprofv3 kernel_stats.csv by kernel family (template name without arguments): total ms and share.  usage: kstats_group.py file.csv [steps]"""
import csv, re, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
agg = defaultdict(lambda: [0, 0.0])
for r in rows:
    n = re.sub(r"^void ", "", r["Name"]).replace("escx::", "")
    m = re.match(r"([A-Za-z0-9_]+)<([^>]*)>", n)
    key = n.split("(")[0]
    if m:
        args = [a.strip() for a in m.group(2).split(",")]
        types = [a for a in args if not re.fullmatch(r"-?\d+|true|false", a)]
        key = m.group(1) + ("<" + ",".join(types) + ">" if types else "")
    agg[key][0] += int(r["Calls"]); agg[key][1] += float(r["TotalDurationNs"]) / 1e6
tot = sum(v[1] for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{k:60s} calls {v[0]:6d}  {v[1] / steps:9.3f} ms/step  {100 * v[1] / tot:5.1f} %")
print(f"{'TOTAL':60s}               {tot / steps:9.3f} ms/step")
