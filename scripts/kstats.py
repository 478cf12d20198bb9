"""Per-kernel duration table from a rocprofv3 rocpd database (kernel-trace run without --output-format csv)."""
import sqlite3
import sys

for path in sys.argv[1:]:
    c = sqlite3.connect(path)
    print(path)
    for name, n, avg, mn in c.execute(
            "select name, count(*), avg(end-start), min(end-start) from kernels group by name order by 3 desc"):
        if name.startswith("__amd"):
            continue
        print(f"  {avg / 1e3:9.1f} us avg  {mn / 1e3:9.1f} us min  x{n:<6d} {name[:90]}")
