"""Summarise a rocprofv3 --kernel-trace --stats output directory (csv or rocpd sqlite) into a
per-kernel table: calls, total ms, avg us, share.
usage: rocprof_summary.py <dir> [--steady MARKER SKIP] [--by-grid FILTER]
  --by-grid FILTER: second table -- the dispatches whose name contains FILTER, split by launch grid (one row per
  kernel x grid size): shows which tensor sizes a many-launch kernel (bn_*) spends its time on.
  --steady MARKER SKIP: only dispatches that start at or after the (SKIP+1)-th dispatch of the kernel whose name contains
  MARKER (one launch per step, e.g. cost_volume_fwd) are counted -- drops the warm-up steps, in which MIOpen's
  benchmark-mode solver search of the 2-D feature CNN runs thousands of candidate kernels."""
import csv
import glob
import os
import sqlite3
import sys
from collections import defaultdict


BY_GRID = defaultdict(lambda: [0, 0.0])


def from_csv(d, marker=None, skip=0, grid_filter=None):
    rows = defaultdict(lambda: [0, 0.0])
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    recs = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                name = r.get("Kernel_Name") or r.get("kernel_name")
                if grid_filter and grid_filter in name:
                    name_g = (name, int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0) * max(1, int(r.get("Grid_Size_Y") or 1)))
                else:
                    name_g = None
                recs.append((name, int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name_g))
    t0 = 0
    if marker:
        marks = sorted(s for n, s, e, _ in recs if marker in n)
        if len(marks) > skip:
            t0 = marks[skip]
            print(f"steady state: {len(marks) - skip} of {len(marks)} '{marker}' launches kept")
    for n, s, e, ng in recs:
        if s >= t0:
            rows[n][0] += 1
            rows[n][1] += (e - s) / 1e6
            if ng:
                BY_GRID[ng][0] += 1
                BY_GRID[ng][1] += (e - s) / 1e6
    return rows if files else None


def from_db(d):
    files = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    if not files:
        return None
    rows = defaultdict(lambda: [0, 0.0])
    for f in files:
        con = sqlite3.connect(f)
        cur = con.cursor()
        tabs = [t[0] for t in cur.execute("select name from sqlite_master where type in ('table','view')")]
        view = [t for t in tabs if t == "kernels"] or [t for t in tabs if "kernel_dispatch" in t]
        if not view:
            continue
        t = view[0]
        cols = [c[1] for c in cur.execute(f"pragma table_info({t})")]
        if t == "kernels":
            name_c = "name" if "name" in cols else cols[0]
            for n, s, e in cur.execute(f"select {name_c}, start, end from {t}"):
                rows[n][0] += 1
                rows[n][1] += (e - s) / 1e6
        else:
            sym = [x for x in tabs if "kernel_symbol" in x][0]
            q = (f"select s.kernel_name, d.start, d.end from {t} d join {sym} s on d.kernel_id = s.id")
            for n, s, e in cur.execute(q):
                rows[n][0] += 1
                rows[n][1] += (e - s) / 1e6
    return rows


def main():
    d = sys.argv[1]
    marker, skip = (sys.argv[3], int(sys.argv[4])) if len(sys.argv) > 4 and sys.argv[2] == "--steady" else (None, 0)
    gf = sys.argv[sys.argv.index("--by-grid") + 1] if "--by-grid" in sys.argv else None
    rows = from_csv(d, marker, skip, gf) or from_db(d)
    if not rows:
        print("no kernel trace found in", d)
        return
    tot = sum(v[1] for v in rows.values())
    print(f"{'calls':>7} {'total_ms':>10} {'avg_us':>10} {'share':>6}  kernel")
    for n, (c, ms) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:60]:
        print(f"{c:7d} {ms:10.3f} {ms / c * 1e3:10.1f} {ms / tot * 100:5.1f}%  {n[:150]}")
    print(f"total kernel time {tot:.3f} ms")
    if BY_GRID:
        print(f"\n{'calls':>7} {'total_ms':>10} {'avg_us':>10} {'threads':>10}  kernel (by launch grid)")
        for (n, g), (c, ms) in sorted(BY_GRID.items(), key=lambda kv: (kv[0][0], -kv[1][1])):
            print(f"{c:7d} {ms:10.3f} {ms / c * 1e3:10.1f} {g:10d}  {n[:90]}")


if __name__ == "__main__":
    main()
