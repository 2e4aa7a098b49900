#!/usr/bin/env python3
"""Static instruction counts per kernel of an AMDGPU assembly listing (hipcc -S --cuda-device-only): total, vector ALU (of which
double precision), scalar, LDS, memory.  Usage: python tools/isa_count.py file.s [name-filter]"""
import re
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)s_endpgm", txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if flt not in name:
        continue
    ins = []
    for l in body.split("\n"):
        t = l.strip()
        if l.startswith("\t") and t and not t.startswith((".", ";")):
            ins.append(t.split()[0])
    v = sum(1 for i in ins if i.startswith("v_"))
    f64 = sum(1 for i in ins if i.startswith("v_") and "f64" in i)
    s = sum(1 for i in ins if i.startswith("s_"))
    ds = sum(1 for i in ins if i.startswith("ds_"))
    g = sum(1 for i in ins if i.startswith(("global_", "flat_", "buffer_", "scratch_")))
    print(f"{name[:60]:60s} total {len(ins):6d}  valu {v:6d} (f64 {f64:5d})  salu {s:6d}  lds {ds:4d}  mem {g:4d}")
