#!/usr/bin/env python3
"""Compile the HIP sources with -Rpass-analysis=kernel-resource-usage and print one row per kernel
(VGPR/AGPR/spill/scratch/occupancy) -- the first thing to look at after touching a kernel."""
import re, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "mac-network_amd", "csrc", "macx_api.hip")
raw = os.path.join(root, "mac-network_amd", "lib", "kernel_resources.raw")
if "--from-build" in sys.argv and os.path.exists(raw):      # what the last build.py run saw (no second compile)
    out = open(raw).read()
else:
    out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-I", os.path.join(root, "include"), src,
                          "-o", "/tmp/_kr.so", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
flt = [a for a in sys.argv[1:] if not a.startswith("--")]
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: .*?(Function Name|Name): (\S+)", line)
    if m:
        cur = {"name": m.group(2)}; rows.append(cur); continue
    m = re.search(r"remark: .*?\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
def demangle(n):
    try: return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip()
    except Exception: return n
print(f"{'kernel':90s} {'VGPR':>5s} {'AGPR':>5s} {'spill':>5s} {'scr':>5s} {'occ':>4s} {'SGPR':>5s}")
for r in rows:
    if flt and not any(f in r["name"] or f in demangle(r["name"]) for f in flt):
        continue
    n = demangle(r["name"]).replace("macx::", "")[:90]
    print(f"{n:90s} {r.get('VGPRs',-1):5d} {r.get('AGPRs',-1):5d} {r.get('VGPRs Spill',-1):5d} {r.get('ScratchSize',-1):5d} {r.get('Occupancy',-1):4d} {r.get('SGPRs',-1):5d}")
