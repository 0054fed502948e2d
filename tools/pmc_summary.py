"""Average PMC counter values per kernel from a rocprofv3 --pmc run (csv or sqlite). usage: pmc_summary.py <dir> [substr]"""
import csv
import glob
import os
import sqlite3
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = r.get("Kernel_Name", "")
                if sub in k:
                    a = acc[k[:90]][r["Counter_Name"]]
                    a[0] += 1
                    a[1] += float(r["Counter_Value"])
    if not acc:
        for f in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
            con = sqlite3.connect(f)
            cur = con.cursor()
            tabs = [t[0] for t in cur.execute("select name from sqlite_master where type in ('table','view')")]
            if "counters_collection" in tabs:
                cols = [c[1] for c in cur.execute("pragma table_info(counters_collection)")]
                kn = "kernel_name" if "kernel_name" in cols else "name"
                for k, c, v in cur.execute(f"select {kn}, counter_name, value from counters_collection"):
                    if sub in k:
                        a = acc[k[:90]][c]
                        a[0] += 1
                        a[1] += float(v)
            else:
                print("tables:", tabs)
    for k, cs in acc.items():
        print(k)
        for c, (n, v) in sorted(cs.items()):
            print(f"   {c:34s} n={n:4d} avg={v / n:16.1f}")


if __name__ == "__main__":
    main()
