#!/usr/bin/env python3
"""Timeline of ONE training step from a `rocprofv3 --kernel-trace --output-format csv` run of tools/train_bench.py:
every dispatch of the last complete step in start order with its duration and the idle gap in front of it (time since the
latest end of any earlier dispatch), then the step's totals -- span, busy time (union of the dispatch intervals), idle time,
and the idle time grouped by the kernel that FOLLOWS the gap.

usage: train_timeline.py kernel_trace.csv [out.md] [--all]      (--all: list every dispatch, default: totals + per-kernel table)"""
import csv
import re
import subprocess
import sys
from collections import defaultdict


def short(name):
    if name.startswith("_Z"):
        try:
            name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        except Exception:
            pass
    name = re.sub(r"\(anonymous namespace\)::", "", name).replace("void ", "")
    m = re.match(r"([\w:]+(?:<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:60]


def main(path, out=None, full=False):
    rows = list(csv.DictReader(open(path)))
    col = lambda *names: next(n for n in names if n in rows[0])
    ks, ke, kn = col("Start_Timestamp", "start"), col("End_Timestamp", "end"), col("Kernel_Name", "name")
    kq = next((n for n in ("Stream_Id", "Queue_Id") if n in rows[0]), None)
    ev = sorted(((int(r[ks]), int(r[ke]), short(r[kn]), r.get(kq, "")) for r in rows), key=lambda e: e[0])
    starts = [i for i, e in enumerate(ev) if e[2].startswith("time_mlp1_kernel")]      # first kernel of d3dp_train_forward
    if len(starts) < 2:
        raise SystemExit("need two steps in the trace")
    a, b = starts[-2], starts[-1]
    step = ev[a:b]
    t0 = step[0][0]
    lines, gaps, durs, cnt = [], defaultdict(float), defaultdict(float), defaultdict(int)
    busy_end, busy, idle = step[0][0], 0.0, 0.0
    for s, e, n, q in step:
        gap = max(0, s - busy_end) / 1e3
        if s > busy_end:
            idle += gap
        busy += (max(e, busy_end) - max(s, busy_end)) / 1e3
        busy_end = max(busy_end, e)
        gaps[n] += gap; durs[n] += (e - s) / 1e3; cnt[n] += 1
        lines.append(f"| {(s - t0) / 1e3:9.1f} | {(e - s) / 1e3:7.1f} | {gap:5.1f} | {q} | `{n}` |")
    span = (busy_end - t0) / 1e3
    head = [f"one training step: {len(step)} dispatches, span {span / 1e3:.3f} ms, busy {busy / 1e3:.3f} ms, idle between dispatches "
            f"{idle / 1e3:.3f} ms ({idle / max(len(step) - 1, 1):.2f} us per boundary)", "",
            "| kernel | calls | total us | avg us | idle in front: total us | avg us |", "|---|---:|---:|---:|---:|---:|"]
    for n in sorted(durs, key=lambda k: -durs[k]):
        head.append(f"| `{n}` | {cnt[n]} | {durs[n]:.0f} | {durs[n] / cnt[n]:.1f} | {gaps[n]:.0f} | {gaps[n] / cnt[n]:.2f} |")
    text = "\n".join(head) + "\n"
    if full:
        text += "\n| start us | dur us | gap us | stream | kernel |\n|---:|---:|---:|---|---|\n" + "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    print(text if not full else "\n".join(head))


if __name__ == "__main__":
    a = [x for x in sys.argv[1:] if x != "--all"]
    main(a[0], a[1] if len(a) > 1 else None, "--all" in sys.argv)
