#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel-trace) into a per-kernel stats table:
calls, total/avg/min/max duration.  Usage: tools/rocpd_summary.py results.db [out.md]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for n, c, s, a, mn, mx in rows:
        lines.append("| %s | %d | %.3f | %.2f | %.2f | %.2f | %.1f |" % (n[:90], c, s / 1e6, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
