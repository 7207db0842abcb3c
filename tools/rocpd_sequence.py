"""Ordered kernel sequence of ONE step (between the last two optimizer kernels) from a rocpd SQLite result file.

usage: python tools/rocpd_sequence.py results.db [marker substring, default adam_hyp_kernel]
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 else "adam_hyp_kernel"
    con = sqlite3.connect(db)
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    if len(marks) < 2:
        print("fewer than two marker kernels")
        return
    a, b = marks[-2], marks[-1]
    t0 = rows[a][2]
    print("# %d kernels between the last two %s, %.3f ms" % (b - a, marker, (rows[b][2] - t0) / 1e6))
    for name, s, e in rows[a + 1:b + 1]:
        print("%9.1f %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, name[:110]))


if __name__ == "__main__":
    main()
