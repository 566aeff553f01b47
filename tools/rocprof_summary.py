#!/usr/bin/env python3
"""(durations in the rocpd `top_kernels` view are microseconds)
Summarise a rocprofv3 --kernel-trace --stats run (rocpd sqlite .db) as a small per-kernel table
(name, calls, total ms, avg us, % of GPU time) so it can be committed under profiles/."""
import re
import sqlite3
import sys


def demangle(name: str) -> str:
    if name.startswith("_Z"):
        try:
            import subprocess
            return subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        except Exception:
            return name
    return name


def short(name: str) -> str:
    name = demangle(name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    return name[:110]


def main(db, out=None):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    lines = ["| kernel | calls | total ms | avg us | % GPU time |", "|---|---:|---:|---:|---:|"]
    for name, calls, tot, avg, pct in rows:
        lines.append(f"| `{short(name)}` | {calls} | {tot / 1e3:.2f} | {avg:.1f} | {pct:.2f} |")
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main(*sys.argv[1:3])
