#!/usr/bin/env python
"""Register / LDS / spill summary per kernel of a gfx950 assembly listing (hipcc -S --cuda-device-only)."""
import re
import subprocess
import sys


def main(path):
    s = open(path).read()
    rows = []
    for b in s.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", b).group(1)
        get = lambda k: re.search(r"\.%s:\s+(\d+)" % k, b).group(1)
        rows.append((name, get("vgpr_count"), b.split("\n")[0].strip(), get("sgpr_count"), get("vgpr_spill_count"),
                     get("group_segment_fixed_size")))
    dem = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
    for r, d in zip(rows, dem):
        d = d.replace("(anonymous namespace)::", "").replace("void ", "")
        print(f"{d[:86]:86s} vgpr={r[1]:>3s} agpr={r[2]:>3s} sgpr={r[3]:>3s} spill={r[4]:>3s} lds={r[5]}")


if __name__ == "__main__":
    main(sys.argv[1])
