#!/usr/bin/env python3
"""Dump the per-kernel summary (rocprofv3 --kernel-trace --stats, rocpd sqlite output) as CSV text."""
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    cur = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels")
    lines = ["name,calls,total_us,avg_us,percent"]
    for name, calls, total, avg, pct in cur:
        lines.append('"%s",%d,%.3f,%.3f,%.2f' % (name.replace('"', "'"), calls, total, avg, pct))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main(*sys.argv[1:3])
