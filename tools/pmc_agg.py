#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc counter_collection.csv files: per kernel, mean counter value per dispatch, VGPR/LDS, mean duration.

usage: pmc_agg.py <dir-or-csv> [<dir-or-csv> ...] [--json out.json] [--top N]
"""
import csv, glob, json, os, re, sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("escx::", "")
    return re.sub(r"\(.*$", "", name)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    top = 14
    out_json = None
    av = sys.argv[1:]
    for i, a in enumerate(av):
        if a == "--top": top = int(av[i + 1]); args.remove(av[i + 1])
        if a == "--json": out_json = av[i + 1]; args.remove(av[i + 1])
    files = []
    for a in args:
        files += [a] if a.endswith(".csv") else glob.glob(os.path.join(a, "**", "*counter_collection.csv"), recursive=True)
    agg = defaultdict(lambda: {"n": defaultdict(int), "sum": defaultdict(float), "dur": 0.0, "nd": 0, "vgpr": 0, "agpr": 0, "lds": 0, "wg": 0, "grid": 0})
    for f in files:
        for r in csv.DictReader(open(f)):
            k = agg[short(r["Kernel_Name"])]
            c = r["Counter_Name"]
            k["n"][c] += 1; k["sum"][c] += float(r["Counter_Value"])
            k["dur"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; k["nd"] += 1
            k["vgpr"] = int(r["VGPR_Count"]); k["agpr"] = int(r["Accum_VGPR_Count"]); k["lds"] = int(r["LDS_Block_Size"]); k["wg"] = int(r["Workgroup_Size"])
            k["grid"] = max(k["grid"], int(r["Grid_Size"]))
    rows = []
    for name, k in agg.items():
        ctr = {c: k["sum"][c] / k["n"][c] for c in k["n"]}
        calls = max(k["n"].values())
        rows.append({"kernel": name, "calls": calls, "avg_us": k["dur"] / max(k["nd"], 1), "vgpr": k["vgpr"], "agpr": k["agpr"], "lds": k["lds"], "wg": k["wg"],
                     "max_grid": k["grid"], "counters": ctr})
    rows.sort(key=lambda r: -r["avg_us"] * r["calls"])
    for r in rows[:top]:
        print(f"{r['kernel'][:70]:70s} calls {r['calls']:5d} avg {r['avg_us']:8.1f} us  vgpr {r['vgpr']}+{r['agpr']} lds {r['lds']} wg {r['wg']}")
        for c, v in sorted(r["counters"].items()):
            print(f"      {c:32s} {v:16.1f}")
    if out_json:
        json.dump(rows, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
