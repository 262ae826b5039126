#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd (.db) kernel trace: per-kernel calls / total / mean / min / max (us).
Usage: python tools/rocpd_stats.py results.db [--csv out.csv]"""
import sqlite3
import sys


def stats(path):
    db = sqlite3.connect(path)
    q = """select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start)
           from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
           group by s.kernel_name order by 3 desc"""
    rows = db.execute(q).fetchall()
    tot = sum(r[2] for r in rows) or 1
    return [(r[0], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot) for r in rows]


if __name__ == "__main__":
    rows = stats(sys.argv[1])
    lines = ["name,calls,total_us,mean_us,min_us,max_us,pct"]
    for r in rows:
        lines.append('"%s",%d,%.1f,%.2f,%.2f,%.2f,%.2f' % r)
    out = "\n".join(lines)
    if "--csv" in sys.argv:
        open(sys.argv[sys.argv.index("--csv") + 1], "w").write(out + "\n")
    print(out)
