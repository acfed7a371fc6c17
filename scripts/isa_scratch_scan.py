"""Scratch traffic INSIDE loops: for every kernel of the given HIP sources, the loops (backward
branches of the gfx950 assembly) that contain scratch_load / scratch_store instructions, with
their MFMA and vector-memory-load counts.  A spill that is reloaded once per tile is harmless; a
`scratch_load` inside a K loop is a memory round trip -- and its `s_waitcnt vmcnt(0)` drains the
wave's whole in-order prefetch queue -- per iteration.  The resource-usage report
(-Rpass-analysis=kernel-resource-usage) gives scratch BYTES; this gives where they are touched.

    python scripts/isa_scratch_scan.py [file.hip ...]      (default: every file under vln-ce_amd/csrc)
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
srcs = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "vln-ce_amd", "csrc", "*.hip")))
for src in srcs:
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S",
                        "--cuda-device-only", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    print(f"## {os.path.relpath(src, ROOT)}")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\S+:", l) and "@" in l]
    for i0 in starts:
        try:
            i1 = next(j for j in range(i0, len(lines)) if ".Lfunc_end" in lines[j])
        except StopIteration:
            continue
        body = lines[i0:i1]
        name = subprocess.run(["c++filt", lines[i0].split(":")[0]],
                              capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(anonymous namespace\)::|vlnce_detail::", "", name).split("(")[0]
        total = sum("scratch_" in l for l in body)
        labels = {}
        for i, l in enumerate(body):
            m = re.match(r"^(\.LBB\d+_\d+):", l)
            if m:
                labels[m.group(1)] = i
        loops = []
        for i, l in enumerate(body):
            mm = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
            if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
                seg = body[labels[mm.group(1)]:i]
                ns = sum("scratch_" in x for x in seg)
                if ns:
                    loops.append((len(seg), sum("v_mfma" in x for x in seg),
                                  sum(bool(re.match(r"\s+(buffer|global)_load", x)) for x in seg), ns))
        if total == 0:
            continue
        inner = sorted(loops)[:3]
        print(f"  {name[:70]:70s} scratch instructions {total:3d}; in loops (lines, MFMAs, loads, scratch): "
              f"{inner if inner else 'none -- straight-line code only'}")
