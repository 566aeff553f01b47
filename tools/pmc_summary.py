#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter values per dispatch for kernels whose name matches a regular expression.
usage: pmc_summary.py <counter_collection.csv> <regex> [out.md]"""
import collections
import csv
import re
import sys


def main(path, sub, out=None):
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    dur = collections.defaultdict(list)
    for r in rows:
        k = r["Kernel_Name"]
        if not re.search(sub, k):
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
        if r.get("Start_Timestamp") and r.get("End_Timestamp"):
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    lines = []
    for k, v in agg.items():
        lines.append(f"### `{k[:120]}`")
        lines.append("| counter | mean per dispatch | dispatches |")
        lines.append("|---|---:|---:|")
        for c, val in sorted(v.items()):
            lines.append(f"| {c} | {val / cnt[(k, c)]:.0f} | {cnt[(k, c)]} |")
        if dur[k]:
            d = sorted(dur[k])
            lines.append(f"| (duration while counting, us: median) | {d[len(d) // 2]:.1f} | {len(d)} |")
        lines.append("")
    text = "\n".join(lines)
    if out:
        open(out, "a").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main(*sys.argv[1:4])
