#!/usr/bin/env python3
"""Compile rt_device.hip for gfx950 with the given extra flags and print the register / scratch use of every render kernel.
Usage: tools/kres.py [-DFOO ...] [-o out.so]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = "/tmp/kres.so"
extra = []
show_all = False
a = sys.argv[1:]
while a:
    x = a.pop(0)
    if x == "-o":
        out = a.pop(0)
    elif x == "--all":
        show_all = True
    else:
        extra.append(x)
cs = os.path.join(ROOT, "cuda_raytracing_amd", "csrc")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-mllvm", "-disable-machine-licm", "-fno-gpu-rdc",
       "-I" + os.path.join(ROOT, "include"), "-I" + cs, os.path.join(cs, "rt_device.hip"), os.path.join(cs, "host_scene.cpp"),
       os.path.join(cs, "rt_error.cpp"), "-Rpass-analysis=kernel-resource-usage", "-o", out] + extra
p = subprocess.run(cmd, capture_output=True, text=True)
if p.returncode:
    sys.stderr.write(p.stderr)
    sys.exit(p.returncode)
cur = None
rows = {}
for line in p.stderr.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+:\s+(.*?) \[-Rpass", line) or re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        m = re.search(r":\d+:\d+:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:") or t.startswith("Name:"):
        cur = t.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
for name, r in rows.items():
    if not show_all and not any(k in name for k in ("render_kernel", "path_kernel", "shadow_kernel", "fold_kernel")):
        continue
    m = re.match(r"_Z13render_kernelILb(\d)ELi(\d+)ELb(\d)E", name)
    d = "render_kernel<strict=%s, cap=%s, share=%s>" % m.groups() if m else name[:60]
    print(f"{d[:60]:60s} VGPR {r.get('VGPRs')} AGPR {r.get('AGPRs')} spill {r.get('VGPRs Spill', r.get('VGPR Spill'))} scratch {r.get('ScratchSize [bytes/lane]')} "
          f"occ {r.get('Occupancy [waves/SIMD]')} LDS {r.get('LDS Size [bytes/block]')} SGPR {r.get('TotalSGPRs')}")
