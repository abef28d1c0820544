"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / percentage.
    python scripts/rocpd_summary.py gpurun_out/prof1/r01_results.db > profiles/r01_kernel_stats.txt
"""
import sqlite3
import sys


def main(path, top=40):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       f"from kernels group by {name_col} order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print(f"# rocprofv3 --kernel-trace --stats summary of {path}")
    print(f"# total kernel time {tot/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    print(f"{'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>9} {'max_us':>9} {'pct':>6}  kernel")
    for n, c, s, a, mn, mx in rows[:top]:
        print(f"{c:7d} {s/1e6:10.3f} {a/1e3:10.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*s/tot:6.2f}  {n[:150]}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
