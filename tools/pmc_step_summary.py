#!/usr/bin/env python3
"""Whole-step hardware counters: sum rocprofv3 --pmc counter values per KERNEL over every dispatch of one bench step.
usage: pmc_step_summary.py out.md label=<counter_collection.csv> ...   (one csv per counter pass)"""
import collections
import csv
import re
import subprocess
import sys


def short(name):
    if name.startswith("_Z"):
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*", "", name)[:96]      # (long enough to keep the template arguments that tell the FAST Linears apart)


def main(out, *passes):
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    for p in passes:
        rows = list(csv.DictReader(open(p)))
        seen = collections.Counter()
        part = collections.defaultdict(float)
        for r in rows:
            k = short(r["Kernel_Name"])
            part[(k, r["Counter_Name"])] += float(r["Counter_Value"])
            seen[(k, r["Counter_Name"])] += 1
        for (k, c), v in part.items():
            if c in tot[k]:
                continue          # (a kernel matched by the filter of more than one pass -- embed_ln_kernel by "ln_kernel<" -- counts once)
            tot[k][c] = v
            calls[k] = max(calls[k], seen[(k, c)])
    ctrs = sorted({c for v in tot.values() for c in v})
    lines = ["| kernel | dispatches | " + " | ".join(ctrs) + " |", "|---|---:|" + "---:|" * len(ctrs)]
    for k in sorted(tot, key=lambda k: -tot[k].get("GRBM_GUI_ACTIVE", tot[k].get("FETCH_SIZE", 0))):
        lines.append(f"| `{k}` | {calls[k]} | " + " | ".join(f"{tot[k].get(c, 0):.4g}" for c in ctrs) + " |")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], *sys.argv[2:])
