"""Gaps and durations of the LAST single-clip encode+decode in a rocprofv3 kernel trace: python tools/b1_trace_stats.py trace.csv"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# one iteration = from one stft frame GEMM (FrameA) to the next
idx = [i for i, n in enumerate(names) if "FrameA" in n]
a, b = idx[-2], idx[-1]
it = rows[a:b]
t0 = int(it[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in it)
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in it)
gaps = [int(it[i + 1]["Start_Timestamp"]) - int(it[i]["End_Timestamp"]) for i in range(len(it) - 1)]
print(f"launches {len(it)}  span {(t1 - t0) / 1e3:.1f} us  sum of kernel durations {busy / 1e3:.1f} us  sum of positive gaps {sum(g for g in gaps if g > 0) / 1e3:.1f} us  median gap {sorted(gaps)[len(gaps) // 2] / 1e3:.2f} us")
fam = collections.defaultdict(lambda: [0, 0])
for r in it:
    k = r["Kernel_Name"].split("(")[0].replace("void escx::", "").replace("escx::", "")[:60]
    fam[k][0] += 1; fam[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (n, d) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"  {k:62s} x{n:3d}  {d / 1e3:8.1f} us  {d / n / 1e3:7.1f} us each")
