"""Per-dispatch view of a rocprofv3 (rocpd sqlite) kernel trace: the dispatches of one step in launch order with their
durations and the idle gap before each, averaged over the last N repetitions of the pattern that starts at `anchor`.
    python scripts/rocpd_sequence.py kt_results.db k_preprocess_fwd [reps]
"""
import sqlite3
import sys


def main(path, anchor, reps=10):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    starts = [i for i, r in enumerate(rows) if anchor in r[0]]
    if len(starts) < reps + 2:
        print("too few repetitions of", anchor); return
    starts = starts[-(reps + 1):]
    period = starts[1] - starts[0]
    if any(starts[i + 1] - starts[i] != period for i in range(reps)):
        print("# irregular pattern: lengths", [starts[i + 1] - starts[i] for i in range(reps)])
        period = min(starts[i + 1] - starts[i] for i in range(reps))
    print(f"# {path}: {period} dispatches per step, averaged over {reps} steps")
    print(f"{'#':>3} {'dur_us':>9} {'gap_us':>8}  kernel")
    tot = 0.0
    for j in range(period):
        d = g = 0.0
        for i in range(reps):
            n, s, e = rows[starts[i] + j]
            d += (e - s) / 1e3
            g += (s - rows[starts[i] + j - 1][2]) / 1e3
        n = rows[starts[0] + j][0]
        n = n.replace("(anonymous namespace)::", "").replace("void ", "")
        print(f"{j:3d} {d/reps:9.2f} {g/reps:8.2f}  {n[:80]}")
        tot += (d + g) / reps
    print(f"# sum of durations and gaps: {tot:.1f} us")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 10)
