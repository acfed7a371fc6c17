"""Per-kernel sums of a rocprofv3 --pmc counter from a rocpd sqlite database, split into the
segments that `bench.py --pmc-step` separates with marker kernels (torch.cuda._sleep).

   python scripts/rocpd_pmc.py <results.db> [marker-substring=spin_kernel]
Prints, per segment and (kernel, counter): dispatches, sum, mean.
"""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
marker = sys.argv[2] if len(sys.argv) > 2 else "spin_kernel"
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
info = [t for t in tabs if "info_pmc" in t][0]
ev = [t for t in tabs if "pmc_event" in t][0]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "info_kernel_symbol" in t][0]
marks = [r[0] for r in c.execute(
    f"select d.start from {kd} d join {ks} s on d.kernel_id = s.id "
    f"where s.kernel_name like '%{marker}%' order by d.start")]
bounds = [0] + marks + [1 << 62]
for seg in range(len(bounds) - 1):
    lo, hi = bounds[seg], bounds[seg + 1]
    q = f"""select s.kernel_name, p.name, count(*), sum(e.value) from {ev} e
            join {info} p on e.pmc_id = p.id join {kd} d on e.event_id = d.event_id
            join {ks} s on d.kernel_id = s.id where d.start > {lo} and d.start < {hi}
            group by 1, 2 order by 4 desc"""
    rows = list(c.execute(q))
    print(f"## segment {seg}: {sum(r[2] for r in rows)} dispatches")
    for name, ctr, n, tot in rows[:40]:
        name = re.sub(r"\(anonymous namespace\)::", "", name)[:100]
        print(f"{ctr:12s} n={n:5d} sum={tot:.6g} mean={tot / n:.6g}  {name}")
