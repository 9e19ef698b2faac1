#!/usr/bin/env python
"""Turns a rocprofv3 (rocpd sqlite) kernel trace into the per-kernel stats table kept under profiles/."""
import sqlite3
import sys


def main(db_path, out_path=None):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels "
                      "order by total_duration desc").fetchall()
    lines = ["| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for name, calls, tot, avg, pct in rows:
        short = name.split("(")[0][-70:]
        lines.append(f"| {short} | {calls} | {tot / 1e3:.3f} | {avg:.2f} | {pct:.2f} |")  # rocpd view units: microseconds
    text = "\n".join(lines)
    if out_path:
        with open(out_path, "a") as fh:
            fh.write(text + "\n")
    print(text)


if __name__ == "__main__":
    main(*sys.argv[1:])
