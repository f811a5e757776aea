"""Diagnostic: kernels of a rocprofv3 --kernel-trace database by launch count (python tools/kernels_by_count.py <dir> [n])."""
import glob, sqlite3, sys
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for f in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    c = sqlite3.connect(f)
    print(c.execute("select sum(duration) / 1e6, count(*) from kernels").fetchone())
    for name, k, s, a in c.execute("select name, count(*), sum(duration), avg(duration) from kernels group by name order by count(*) desc limit ?", (n,)):
        print(k, round(s / 1e6, 2), round(a / 1e3, 2), name[:150])
