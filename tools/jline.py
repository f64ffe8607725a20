"""Print selected fields of the last JSON line on stdin: python tools/jline.py value ms_per_step roofline.frac"""
import json, sys
last = None
for l in sys.stdin:
    l = l.strip()
    if l.startswith("{"):
        last = l
d = json.loads(last)
out = []
for f in sys.argv[1:]:
    v = d
    for p in f.split("."):
        v = v.get(p) if isinstance(v, dict) else None
    out.append(f"{f}={v}")
print(" ".join(out))
