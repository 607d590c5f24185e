# -*- coding: utf-8 -*-
"""Summarise a rocprofv3 rocpd database (`*_results.db`) into a per-kernel table.

    python tools/prof_summary.py gpurun_out/prof/x_results.db [out.txt]
"""
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name,total_calls,total_duration,average,percentage "
                          "from top_kernels"))
    lines = ["%-112s %6s %13s %11s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct")]
    for name, calls, total, avg, pct in rows:
        lines.append("%-112s %6d %13.1f %11.2f %7.2f" % (name[:112], calls, total, avg, pct))
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == '__main__':
    main(*sys.argv[1:3])
