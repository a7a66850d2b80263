#!/usr/bin/env python
"""Register / spill summary of every kernel of a BUILT library, read from its gfx950 code objects:

    python tools/kernel_resources.py [path/to/libfvp_hip.so] [name-substring]

For each kernel: VGPRs, SGPRs, spill counts (code-object metadata, `llvm-readelf --notes`), number of MFMAs and the
packed-f32 VALU instructions (`v_pk_{add,mul,fma}_f32`) that sit between the first and the last MFMA of a basic-block run
(MI355X_MICROARCH.md: packed f32 beside MFMAs is an anti-lever).  `scan_library()` is what tests/test_kernel_resources.py
gates the build on."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_LIB = os.path.join(ROOT, "faster-voxelpose_amd", "libfvp_hip.so")


def _code_objects(lib, tmp):
    dst = os.path.join(tmp, "lib.so")
    shutil.copy(lib, dst)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", dst], cwd=tmp, capture_output=True, check=True)
    return sorted(os.path.join(tmp, f) for f in os.listdir(tmp) if "amdgcn" in f)


def _metadata(co):
    out = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
    kernels, cur = [], None
    for line in out.splitlines():
        m = re.match(r"\s+(?:- )?\.(\w+):\s+(\S+)\s*$", line)
        if line.lstrip().startswith("- .agpr_count") or (line.lstrip().startswith("- .") and cur is None):
            cur = {}
            kernels.append(cur)
        elif re.match(r"\s+- \.\w+:", line) and not line.startswith("      "):
            cur = {}
            kernels.append(cur)
        if m and cur is not None and not line.startswith("      "):
            cur[m.group(1)] = m.group(2)
    return [k for k in kernels if "name" in k]


def _packed_between_mfma(co):
    """{kernel: (n_mfma, n_packed_f32_between_mfmas)} from the disassembly.  'Between' = inside a straight run of
    instructions (no label in between) that contains MFMAs both before and after the packed instruction, or within 40
    instructions of an MFMA in the same loop body."""
    out = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], capture_output=True, text=True,
                         check=True).stdout
    res, name, ins = {}, None, []

    def flush():
        if name is None:
            return
        mf = [i for i, s in enumerate(ins) if s.startswith("v_mfma")]
        pk = [i for i, s in enumerate(ins) if re.match(r"v_pk_(add|mul|fma)_f32", s)]
        near = 0
        for p in pk:
            before = any(0 < p - m <= 40 for m in mf)
            after = any(0 < m - p <= 40 for m in mf)
            near += bool(before and after)
        # VALU-writes-SGPR -> VMEM hazard (5 wait states): the inline-asm buffer accesses are invisible to
        # hipcc's hazard recognizer, so a v_readfirstlane whose SGPR is consumed by a buffer_* instruction among the next
        # five instructions would go unnoticed.  Count them (the descriptors are re-pointed with SALU today: expected 0).
        haz = 0
        for i, ins_i in enumerate(ins):
            m = re.match(r"v_readfirstlane_b32 s(\d+),", ins_i)
            if not m:
                continue
            n = int(m.group(1))
            for nxt in ins[i + 1:i + 6]:
                if not nxt.startswith("buffer_"):
                    continue
                used = set()
                for lo, hi in re.findall(r"s\[(\d+):(\d+)\]", nxt):
                    used.update(range(int(lo), int(hi) + 1))
                used.update(int(x) for x in re.findall(r"\bs(\d+)\b", nxt))
                haz += n in used
        res[name] = (len(mf), near, haz)

    for line in out.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            flush()
            name, ins = m.group(1), []
            continue
        s = line.strip()
        if s and not s.endswith(":") and name is not None:
            ins.append(s.split("//")[0].strip())
    flush()
    return res


def scan_library(lib=DEFAULT_LIB):
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for co in _code_objects(lib, tmp):
            pk = _packed_between_mfma(co)
            for k in _metadata(co):
                n_mfma, n_pk, n_haz = pk.get(k["name"], (0, 0, 0))
                rows.append(dict(name=k["name"], vgpr=int(k.get("vgpr_count", 0)), agpr=int(k.get("agpr_count", 0)),
                                 sgpr=int(k.get("sgpr_count", 0)), sgpr_spill=int(k.get("sgpr_spill_count", 0)),
                                 vgpr_spill=int(k.get("vgpr_spill_count", 0)), scratch=int(k.get("private_segment_fixed_size", 0)),
                                 mfma=n_mfma, packed_f32_between_mfma=n_pk, readfirstlane_to_buffer_hazards=n_haz))
    return rows


def demangle(names):
    exe = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
    if not exe:
        return list(names)
    r = subprocess.run([exe], input="\n".join(names), capture_output=True, text=True)
    return r.stdout.splitlines() if r.returncode == 0 else list(names)


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 and os.path.isfile(sys.argv[1]) else DEFAULT_LIB
    pat = next((a for a in sys.argv[1:] if not os.path.isfile(a)), "")
    rows = scan_library(lib)
    for r, d in zip(rows, demangle([r["name"] for r in rows])):
        d = d.split("(")[0]
        if pat and pat not in d:
            continue
        print(f"{d[:86]:86s} VGPR {r['vgpr']:3d} AGPR {r['agpr']:3d} SGPR {r['sgpr']:3d} sspill {r['sgpr_spill']:3d} "
              f"vspill {r['vgpr_spill']:3d} scratch {r['scratch']:4d} mfma {r['mfma']:4d} pk_f32@mfma {r['packed_f32_between_mfma']:3d}")
    bad = [r for r in rows if r["sgpr_spill"] or r["vgpr_spill"] or r["packed_f32_between_mfma"]]
    print(f"{len(rows)} kernels, {len(bad)} with spills or packed f32 beside MFMAs")
