"""Summarise a rocprofv3 rocpd (sqlite) kernel trace into the per-kernel stats table we commit under profiles/."""
import sqlite3
import sys


def main(db, out=None):
    import glob
    import os
    if os.path.isdir(db):   # rocprofv3 -d <dir>: <dir>/<host>/<pid>_results.db
        found = sorted(glob.glob(os.path.join(db, "**", "*.db"), recursive=True), key=os.path.getsize)
        if not found:
            raise SystemExit("no rocpd .db under %s" % db)
        db = found[-1]
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = c.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
                     "group by %s order by sum(end-start) desc" % (name_col, name_col)).fetchall()
    total = sum(r[2] for r in rows)
    lines = ["%-100s %8s %14s %12s %12s %12s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct")]
    for n, cnt, tot, avg, mn, mx in rows:
        n = n if len(n) <= 100 else n[:97] + "..."
        lines.append("%-100s %8d %14d %12.0f %12d %12d %6.2f%%" % (n, cnt, tot, avg, mn, mx, 100.0 * tot / total))
    txt = "\n".join(lines)
    if out:
        open(out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
