#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02x
mkdir -p $O
cd $GRAFT_REPO_ROOT
for t in 0 1 2 3 4; do
  VLNCE_X3_TILE=$t timeout 300 python scripts/convbench.py --n 64 --iters 20 --mode train > $O/conv_tile$t.log 2>&1
done
python - <<PY
import re
rows={}
for t in range(5):
    for l in open("$O/conv_tile%d.log"%t):
        f=l.split()
        if len(f)>=7 and f[0].startswith(("l1_","l2_","l3_","l4_")):
            rows.setdefault(f[0],{})[t]=float(f[4])
print("layer                     auto   128x128  64x128  128x64  64x64   best")
tot_auto=tot_best=0
cnt={l.split()[0]:int(l.split()[-1][1:]) for l in open("$O/conv_tile0.log") if l.split() and l.split()[0].startswith(("l1_","l2_","l3_","l4_"))}
for k,v in rows.items():
    best=min(v[t] for t in (1,2,3,4) if t in v)
    bt=[t for t in (1,2,3,4) if v.get(t)==best][0]
    print(f"{k:24s} {v.get(0,0):7.1f} {v.get(1,0):8.1f} {v.get(2,0):7.1f} {v.get(3,0):7.1f} {v.get(4,0):7.1f}   tile{bt} {'<-- ' if best < 0.95*v.get(0,1e9) else ''}")
    tot_auto+=v.get(0,0)*cnt[k]; tot_best+=best*cnt[k]
print("sum auto %.1f us, sum best %.1f us"%(tot_auto,tot_best))
PY
