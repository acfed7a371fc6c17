"""Print the basic blocks of one kernel that contain MFMAs (the hot loop) from a hipcc -save-temps .s
    python scripts/isa_loop.py file.s <substring of the mangled kernel name> [context]
"""
import re
import sys

path, key = sys.argv[1], sys.argv[2]
ctx = int(sys.argv[3]) if len(sys.argv) > 3 else 0
lines = open(path).read().split("\n")
start = end = None
for i, l in enumerate(lines):
    if start is None and re.match(r"^_Z\S*:", l) and key in l:
        start = i
    elif start is not None and l.startswith("\t.end_amdhsa_kernel") or (start is not None and ".Lfunc_end" in l):
        end = i
        break
body = lines[start:end]
# basic blocks
blocks, cur = [], []
for l in body:
    if re.match(r"^\.LBB\d+_\d+:", l) or l.startswith("; %bb."):
        blocks.append(cur)
        cur = []
    cur.append(l)
blocks.append(cur)
for b in blocks:
    n = sum("v_mfma" in l for l in b)
    if n:
        ins = [l for l in b if l.startswith("\t") and not l.startswith("\t.")]
        kinds = {}
        for l in ins:
            op = l.split()[0]
            fam = ("mfma" if "mfma" in op else "ds" if op.startswith("ds_") else "buffer" if op.startswith("buffer")
                   else "s_wait" if op.startswith("s_waitcnt") else "salu" if op.startswith("s_") else
                   "lane" if "lane" in op else "valu" if op.startswith("v_") else op)
            kinds[fam] = kinds.get(fam, 0) + 1
        print(f"=== block {b[0].split()[0]}  {len(ins)} instrs: {kinds}")
        if ctx:
            print("\n".join(b))
