#!/usr/bin/env python
"""Per-kernel register / LDS / occupancy table from a hipcc -S listing (tuning aid).
usage: isa_stats.py file.s [name-filter-regex]"""
import re, sys
rx = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
name = None; st = {}
for line in open(sys.argv[1]):
    m = re.match(r"^(_Z\S+):\s+; @", line)
    if m: name = m.group(1); st = {}; continue
    m = re.match(r"^; (NumVgprs|NumAgprs|TotalNumVgprs|ScratchSize|Occupancy|LDSByteSize|NumSgprs): (\d+)", line)
    if m and name:
        st[m.group(1)] = int(m.group(2))
        if m.group(1) == "LDSByteSize":
            short = re.sub(r"^_ZN4escx\d+", "", name); short = re.sub(r"EEvNS_.*$", ">", short); short = short.replace("ILi", "<").replace("ELi", ",")
            if not rx or rx.search(short):
                print(f"{short:44s} vgpr {st.get('NumVgprs',0):3d} agpr {st.get('NumAgprs',0):3d} total {st.get('TotalNumVgprs',0):3d} sgpr {st.get('NumSgprs',0):3d} scratch {st.get('ScratchSize',0):4d} occ {st.get('Occupancy',0)} lds {st.get('LDSByteSize',0)}")
