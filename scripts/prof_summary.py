"""Dev tool: per-kernel summary (calls, avg us, total ms) of a rocprofv3 rocpd database."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, count(*), avg(end-start), sum(end-start) from kernels group by name order by sum(end-start) desc"))
tot = sum(r[3] for r in rows)
for name, n, avg, s in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print(f'{s / 1e6:9.2f} ms {100 * s / tot:5.1f}%  n={n:5d}  avg {avg / 1e3:8.1f} us  {name[:90]}')
print(f'{tot / 1e6:9.2f} ms total')
