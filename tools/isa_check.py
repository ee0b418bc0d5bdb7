"""Static check of k_physics<512>'s gfx950 code: registers, scratch bytes per lane, and the scratch / LDS / VALU instruction counts of the
solver's iteration loop (the ISA between the source lines of the loop head and of its closing stamp, via -gline-tables-only).
usage: python tools/isa_check.py [extra hipcc flags...]      (cross-compiles; no GPU needed)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "seqdex_amd", "csrc", "sdx_physics.hip")
KERN = "_Z9k_physicsILi512EEvPK8SdxConst6SdxBuf"

src = open(SRC).read().split("\n")
lo = next(i for i, l in enumerate(src) if "for (int it = it0;" in l) + 1
hi = next(i for i, l in enumerate(src) if "SSTAMP(22);" in l and i > lo) + 1
with tempfile.TemporaryDirectory() as td:
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-gline-tables-only",
           "-Rpass-analysis=kernel-resource-usage", "-save-temps=obj", "-c", SRC, "-o", os.path.join(td, "p.o")] + sys.argv[1:]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=td)
    rem = r.stderr
    blk = rem[rem.index("k_physicsILi512"):]
    for key in ("VGPRs:", "ScratchSize [bytes/lane]:", "SGPRs Spill:", "VGPRs Spill:", "Occupancy [waves/SIMD]:"):
        m = re.search(re.escape(key) + r"\s*(\d+)", blk)
        print(key, m.group(1) if m else "?")
    asm = [f for f in os.listdir(td) if f.endswith("gfx950.s")][0]
    inside, cur = False, None
    loop = {"scratch": 0, "ds": 0, "valu": 0, "salu": 0, "barrier": 0, "total": 0}
    tot_scratch = 0
    seen_lo = False
    fileno = None
    for line in open(os.path.join(td, asm)):
        if line.startswith(KERN + ":"):
            inside = True
            continue
        if not inside:
            m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"sdx_physics.hip"', line)
            if m:
                fileno = int(m.group(1))
            continue
        if ".end_amdhsa_kernel" in line or line.startswith("\t.section"):
            break
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", line)
        if m:
            if int(m.group(1)) == fileno:
                cur = int(m.group(2))
            continue
        t = line.strip()
        if not t or t[0] in ";." or t.endswith(":"):
            continue
        op = t.split()[0]
        if op.startswith("scratch_"):
            tot_scratch += 1
        if cur is not None and lo <= cur <= hi:
            loop["total"] += 1
            if op.startswith("scratch_"):
                loop["scratch"] += 1
            elif op.startswith("ds_"):
                loop["ds"] += 1
            elif op.startswith("v_"):
                loop["valu"] += 1
            elif op == "s_barrier":
                loop["barrier"] += 1
            elif op.startswith("s_"):
                loop["salu"] += 1
    print("scratch instructions in the kernel:", tot_scratch)
    print("iteration loop (source lines %d-%d):" % (lo, hi), loop)
