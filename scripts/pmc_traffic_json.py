"""profiles/rNN_pmc_traffic.json from the two `scripts/rocpd_pmc.py` tables of `bench.py --pmc-step`
(separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes): segment 1 is a 256 MiB device copy
(calibration of the counters' units), segment 2 the two visual trunks' forward.

    python scripts/pmc_traffic_json.py <fetch.txt> <write.txt> <out.json> <label for `source`>
"""
import json
import re
import sys

CONV = ("conv_p3_kernel", "conv_u3_kernel", "conv_s3_kernel", "conv_x3_kernel", "igemm_kernel",
        "stem7_kernel", "conv_m3_kernel")


def segments(path):
    segs, cur = {}, None
    for line in open(path):
        m = re.match(r"## segment (\d+)", line)
        if m:
            cur = segs.setdefault(int(m.group(1)), [])
            continue
        m = re.match(r"\S+\s+n=\s*(\d+) sum=(\S+) mean=\S+\s+(\S+)", line)
        if m and cur is not None:
            cur.append((int(m.group(1)), float(m.group(2)), m.group(3)))
    return segs


def conv_sum(seg):
    n = sum(c for c, _, name in seg if any(k in name for k in CONV))
    kb = sum(s for _, s, name in seg if any(k in name for k in CONV))
    return n, kb


fetch, write = segments(sys.argv[1]), segments(sys.argv[2])
copy_kb = 256 * 1024
# the layout's markers are the LAST three spin kernels of the run (other spin kernels, e.g. the
# stream-placement probe of streams.pick_concurrent_stream, come before them): count from the end
last = max(fetch)
f_frac = fetch[last - 2][0][1] / copy_kb
w_frac = write[last - 2][0][1] / copy_kb
nf, fkb = conv_sum(fetch[last - 1])
nw, wkb = conv_sum(write[last - 1])
assert nf == nw, (nf, nw)
fb, wb = fkb * 1024 / f_frac, wkb * 1024 / w_frac
out = {
    "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over "
              "`python bench.py --pmc-step`, segment 2 = the two visual trunks' forward (N=64, "
              "256x256): conv_p3 / conv_u3 / conv_s3 / conv_x3 / conv_m3 / stem7 / igemm dispatches only (" + sys.argv[4] + ")",
    "calibration": {"note": "256 MiB device copy in the same run (segment 1): FETCH_SIZE reports this "
                            "fraction of the bytes read (gfx950: 1/2), WRITE_SIZE this fraction of "
                            "the bytes written",
                    "fetch_reported_fraction": round(f_frac, 4),
                    "write_reported_fraction": round(w_frac, 4)},
    "conv_launches_per_step": nf,
    "fetch_bytes_per_step": int(fb), "write_bytes_per_step": int(wb),
    "hbm_bytes_per_launch": int((fb + wb) / nf),
    "algorithmic_bytes_per_step": 11950000000,
    "algorithmic_bytes_per_launch": int(11950000000 / nf),
    "note": "algorithmic bytes = inputs + weights + outputs of the 107 convolutions (8.33 GB) + the "
            "residual-block ends evaluated inside 15 of them (3.62 GB).  The bf16-plane kernels read "
            "their weights as 3 planes (1.5x the fp32 bytes; conv_p3 / conv_u3 re-read them per "
            "M-tile from L2).",
}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out)[:400])
