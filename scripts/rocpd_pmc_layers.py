"""Per-layer PMC view of a `rocprofv3 --kernel-trace --pmc ...` run of scripts/convbench.py.

   python scripts/rocpd_pmc_layers.py <results.db> <launches-per-layer> [kernel-substring=igemm]

convbench issues every layer `3 warm-up + iters` times back to back, so consecutive groups of
that many dispatches of the conv kernel are one layer.  Prints, per group: mean duration of the
non-warm-up launches, the mean of every collected counter, and the derived figures
  clock_GHz  = GRBM_GUI_ACTIVE / duration            (effective shader clock, DVFS)
  mfma_busy  = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMD x 256 CU x GRBM_GUI_ACTIVE)
when those counters are present.
"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
group = int(sys.argv[2])
sub = sys.argv[3] if len(sys.argv) > 3 else "igemm"
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
info = [t for t in tabs if "info_pmc" in t][0]
ev = [t for t in tabs if "pmc_event" in t][0]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "info_kernel_symbol" in t][0]
cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
gcols = [x for x in ("grid_size_x", "workgroup_size_x", "grid_size", "workgroup_size") if x in cols]
sel = ", ".join(f"d.{g}" for g in gcols)
rows = list(c.execute(
    f"select d.event_id, d.start, d.end, s.kernel_name{', ' + sel if sel else ''} from {kd} d "
    f"join {ks} s on d.kernel_id = s.id where s.kernel_name like '%{sub}%' order by d.start"))
vals = {}
for eid, name, v in c.execute(
        f"select e.event_id, p.name, e.value from {ev} e join {info} p on e.pmc_id = p.id"):
    vals.setdefault(eid, {})[name] = vals.get(eid, {}).get(name, 0.0) + v
names = sorted({n for d in vals.values() for n in d})
print(f"# {len(rows)} dispatches of *{sub}*, {group} per layer; counters: {names}; dispatch cols {gcols}")
hdr = f"{'grp':>3s} {'n':>3s} {'us':>9s} " + " ".join(f"{n[-22:]:>22s}" for n in names) + "  clock_GHz mfma_busy"
print(hdr)
for g0 in range(0, len(rows), group):
    grp = rows[g0:g0 + group][3:] or rows[g0:g0 + group]
    dur = sum(r[2] - r[1] for r in grp) / len(grp) / 1e3
    mean = {n: sum(vals.get(r[0], {}).get(n, 0.0) for r in grp) / len(grp) for n in names}
    extra = ""
    if "GRBM_GUI_ACTIVE" in mean and dur > 0:
        ghz = mean["GRBM_GUI_ACTIVE"] / (dur * 1e3)
        extra += f"  {ghz:8.3f}"
        if "SQ_VALU_MFMA_BUSY_CYCLES" in mean and mean["GRBM_GUI_ACTIVE"] > 0:
            extra += f" {mean['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * mean['GRBM_GUI_ACTIVE']):8.3f}"
    grid = " ".join(str(x) for x in grp[0][4:])
    print(f"{g0 // group:3d} {len(grp):3d} {dur:9.1f} " + " ".join(f"{mean[n]:22.6g}" for n in names)
          + extra + f"  grid[{grid}]")
