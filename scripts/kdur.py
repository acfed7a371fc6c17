"""median kernel durations, in dispatch order groups: python scripts/kdur.py <db> <name-substring> <group-size>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = [(e - s) / 1e3 for s, e in db.execute(
    f"select start, end from kernels where name like '%{sys.argv[2]}%' order by start")]
n = int(sys.argv[3])
for i in range(0, len(rows), n):
    v = sorted(rows[i:i + n])
    print(f"group {i // n}: n={len(v)} median {v[len(v) // 2]:.2f} us  min {v[0]:.2f}")
