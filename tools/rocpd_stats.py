"""Kernel statistics (the `rocprofv3 --stats` table) from a rocpd SQLite result file.

usage: python tools/rocpd_stats.py results.db [out.csv] [--skip-first N]
"""
import csv
import sqlite3
import sys


def main():
    db = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else None
    con = sqlite3.connect(db)
    rows = con.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    table = [("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")]
    for n, k, s, a, mn, mx in rows:
        table.append((n, k, s, "%.1f" % a, "%.2f" % (100.0 * s / tot), mn, mx))
    if out:
        with open(out, "w", newline="") as f:
            csv.writer(f, quoting=csv.QUOTE_NONNUMERIC).writerows(table)
    for r in table[:45]:
        print("%-100s %6s %12s %10s %6s" % (str(r[0])[:100], r[1], r[2], r[3], r[4]))


if __name__ == "__main__":
    main()
