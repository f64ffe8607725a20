#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (kernel trace) into a per-kernel table: calls, total, average, share.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db > profiles/r1_kernel_stats.txt
"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = name.replace("escx::", "").replace("void ", "")
    return name if len(name) <= 150 else name[:147] + "..."


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       f"from kernels group by {name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    print(f"# {path}: {sum(r[1] for r in rows)} dispatches, {total / 1e6:.3f} ms of kernel time")
    print(f"{'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>9} {'max_us':>9} {'share':>7}  kernel")
    for n, c, t, a, mn, mx in rows:
        print(f"{c:7d} {t / 1e6:10.3f} {a / 1e3:10.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {100.0 * t / total:6.2f}%  {short(n)}")


if __name__ == "__main__":
    main(sys.argv[1])
