"""Per-kernel average of a PMC counter from a rocprofv3 rocpd database (run with --kernel-trace --pmc <COUNTER>).
    python scripts/rocpd_pmc_summary.py <db> [kernel-substring ...]
Prints kernel, counter, dispatches, average value per dispatch (raw counter units; FETCH_SIZE / WRITE_SIZE are KiB)."""
import sqlite3
import sys


def main(path, filters):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) "
                       "from counters_collection group by kernel_name, counter_name order by 4 desc").fetchall()
    print(f"# PMC summary of {path}")
    print(f"{'calls':>6} {'avg':>16} {'min':>16} {'max':>16}  counter      kernel")
    for k, c, n, a, mn, mx in rows:
        if filters and not any(f in k for f in filters):
            continue
        print(f"{n:6d} {a:16.1f} {mn:16.1f} {mx:16.1f}  {c:12s} {k[:110]}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
