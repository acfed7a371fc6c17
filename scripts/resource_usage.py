"""Per-kernel register / scratch / occupancy table of the HIP sources, from
`hipcc -Rpass-analysis=kernel-resource-usage` (cross-compiles without a GPU).

    python scripts/resource_usage.py [file.hip ...] > profiles/rNN_kernel_resource_usage.txt
"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
srcs = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "vln-ce_amd", "csrc", "*.hip")))
want = ("VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "SGPRs Spill",
        "VGPRs Spill", "LDS Size [bytes/block]")
print("# hipcc --offload-arch=gfx950 -O3 -std=c++17 -Rpass-analysis=kernel-resource-usage, one line per kernel")
for src in srcs:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                        "--cuda-device-only", "-c", src, "-o", "/dev/null",
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    print(f"## {os.path.relpath(src, ROOT)}")
    name, vals = None, {}
    for line in r.stderr.split("\n"):
        m = re.search(r"remark: .*Function Name: (\S+)", line)
        if m:
            if name:
                print(name + " | " + " | ".join(f"{k}: {vals.get(k, '?')}" for k in want))
            raw = m.group(1)
            dem = subprocess.run(["c++filt", raw], capture_output=True, text=True).stdout.strip()
            dem = re.sub(r"vlnce_detail::\(anonymous namespace\)::|vlnce_detail::|\(anonymous namespace\)::", "", dem)
            name, vals = re.sub(r"\(.*$", "", dem), {}
            continue
        m = re.search(r"remark:\s+(.+?): (\d+) \[-Rpass", line)
        if m and name:
            vals[m.group(1).strip()] = m.group(2)
    if name:
        print(name + " | " + " | ".join(f"{k}: {vals.get(k, '?')}" for k in want))
