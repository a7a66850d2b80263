#!/usr/bin/env python
"""Register / spill / LDS summary of every kernel of one .hip source (hipcc -Rpass-analysis=kernel-resource-usage):
    python tools/kernel_resources.py fvp_conv.hip [name-substring] [extra hipcc flags...]"""
import os
import re
import subprocess
import sys

csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "faster-voxelpose_amd", "csrc")
src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ""
flags = [a for a in sys.argv[2:] if a.startswith("-")]
extra = [] if src == "fvp_conv.hip" else ["-ffp-contract=off"]
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *extra, *flags,
                    "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(csrc, src), "-o", "/dev/null"],
                   capture_output=True, text=True)
cur = None
rows = {}
for line in r.stderr.splitlines():
    m = re.search(r"remark: +(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
for name, d in rows.items():
    try:
        dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        dem = name
    if pat and pat not in dem:
        continue
    print(f"{dem.split('(')[0][:90]:90s} VGPR {d.get('VGPRs', '?'):>3} AGPR {d.get('AGPRs', '?'):>3} vspill {d.get('VGPRs Spill', '?'):>3} "
          f"sspill {d.get('SGPRs Spill', '?'):>3} scratch {d.get('ScratchSize [bytes/lane]', '?'):>4} occ {d.get('Occupancy [waves/SIMD]', '?')} "
          f"LDS {d.get('LDS Size [bytes/block]', '?')}")
