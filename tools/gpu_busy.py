"""How busy is the GPU during a run?  From a rocprofv3 rocpd (sqlite) kernel trace: wall span of the kernels, the UNION of
their execution intervals (device busy time: kernels on different streams overlap), the plain sum, the number of launches, and the
gaps between consecutive kernels by size class.  python tools/gpu_busy.py <rocprofv3 -d dir | .db> [skip_fraction]
(skip_fraction: leading part of the trace to drop -- warm-up; default 0.3)"""
import glob
import os
import sqlite3
import sys


def main(db, skip=0.3):
    if os.path.isdir(db):
        found = sorted(glob.glob(os.path.join(db, "**", "*.db"), recursive=True), key=os.path.getsize)
        if not found:
            raise SystemExit("no rocpd .db under %s" % db)
        db = found[-1]
    c = sqlite3.connect(db)
    rows = c.execute("select start, end from kernels order by start").fetchall()
    rows = rows[int(len(rows) * skip):]
    if not rows:
        raise SystemExit("no kernels")
    t0, t1 = rows[0][0], max(e for _, e in rows)
    busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
    gaps = []
    for s, e in rows[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append(s - cur_e)
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    total = sum(e - s for s, e in rows)
    span = t1 - t0
    print("kernels %d, span %.3f ms, busy (union) %.3f ms = %.1f %%, sum of kernel times %.3f ms (overlap %.3f ms)" % (
        len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span, total / 1e6, (total - busy) / 1e6))
    for lo, hi in ((0, 2e3), (2e3, 5e3), (5e3, 1e4), (1e4, 3e4), (3e4, 1e5), (1e5, 1e12)):
        g = [x for x in gaps if lo <= x < hi]
        print("  idle gaps %6.0f .. %-8.0f ns: %6d, %.3f ms in all" % (lo, hi, len(g), sum(g) / 1e6))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.3)
