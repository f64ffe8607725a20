#!/usr/bin/env python
"""Kernel time per family from a rocprofv3 kernel_stats.csv: kfam.py file.csv steps"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); steps = float(sys.argv[2])
fam = {}
for r in rows:
    n = r["Name"]
    k = ("mlp" if "mlp_fused" in n else "attn" if "attn_" in n else "rowgemm" if "rowgemm" in n else "combine" if "rows_combine" in n else "deembed" if "deembed" in n
         else "pvq" if ("CodeGather" in n or "ResidualGather" in n or "pvq" in n) else "other")
    fam[k] = fam.get(k, 0) + float(r["TotalDurationNs"]) / steps / 1e6
print("total %.2f ms/step  " % sum(fam.values()) + "  ".join(f"{k} {v:.2f}" for k, v in sorted(fam.items())))
