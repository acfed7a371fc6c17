"""Summarise hipcc -Rpass-analysis=kernel-resource-usage logs in build/*.log."""
import glob
import re
import sys

for path in sorted(glob.glob("build/*.log")):
    txt = open(path).read()
    blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
    for b in blocks:
        name = b.split()[0]

        def g(k):
            m = re.search(re.escape(k) + r": (\d+)", b)
            return m.group(1) if m else "?"

        short = re.sub(r"^_ZN\d+_GLOBAL__N_1\d+", "", name)[:64]
        print(
            "%-8s %-64s vgpr=%s agpr=%s spill=%s scratch=%s occ=%s lds=%s"
            % (
                path.split("/")[-1][:-4], short, g("VGPRs"), g("AGPRs"), g("VGPRs Spill"),
                g("ScratchSize [bytes/lane]"), g("Occupancy [waves/SIMD]"),
                g("LDS Size [bytes/block]"),
            )
        )
