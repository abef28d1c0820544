"""Dump the kernel timeline of a rocprofv3 (rocpd sqlite) kernel trace: one row per dispatch with start / duration / queue,
for a window of dispatches, to see which kernels of different streams actually overlap.
    python scripts/rocpd_timeline.py <db> [first_dispatch [count]]
"""
import sqlite3
import sys


def main(path, first=0, count=120):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    extra = [c for c in ("queue_id", "stream_id", "queue", "stream") if c in cols]
    print("# columns of kernels:", cols)
    sel = ", ".join([name_col, "start", "end"] + extra)
    rows = cur.execute(f"select {sel} from kernels order by start").fetchall()
    rows = rows[first:first + count]
    t0 = rows[0][1]
    prev_end = {}
    print(f"{'start_us':>10} {'dur_us':>8} {'end_us':>10} {'/'.join(extra):>12}  kernel")
    for r in rows:
        n, s, e = r[0], r[1], r[2]
        n = n.replace("(anonymous namespace)::", "").split("(")[0][-60:]
        print(f"{(s - t0) / 1e3:10.2f} {(e - s) / 1e3:8.2f} {(e - t0) / 1e3:10.2f} {'/'.join(str(x) for x in r[3:]):>12}  {n}")


if __name__ == "__main__":
    a = sys.argv
    main(a[1], int(a[2]) if len(a) > 2 else 0, int(a[3]) if len(a) > 3 else 120)
