"""Per-dispatch PMC values of one kernel from a rocprofv3 rocpd database (run with --kernel-trace --pmc ...), in dispatch order.
    python scripts/rocpd_pmc_percall.py <db> <kernel-substring> [max_rows]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
idc = "dispatch_id" if "dispatch_id" in cols else cols[0]
rows = cur.execute(f"select {idc}, counter_name, value from counters_collection where kernel_name like ? order by {idc}", (f"%{sys.argv[2]}%",)).fetchall()
by = {}
for d, c, v in rows:
    by.setdefault(d, {})[c] = by.setdefault(d, {}).get(c, 0) + v
names = sorted({c for _, c, _ in rows})
print("dispatch", *names)
for i, (d, vals) in enumerate(sorted(by.items())):
    if i >= (int(sys.argv[3]) if len(sys.argv) > 3 else 16): break
    print(d, *[f"{vals.get(n, 0):.0f}" for n in names])
