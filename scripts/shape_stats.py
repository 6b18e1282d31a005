"""Dev tool: per-kernel, per-launch-grid rows (calls, average us, ms per batch) of a rocprofv3 rocpd database -> CSV on stdout.
Usage: shape_stats.py <db> <batches>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
nb = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(db.execute("select name, grid_x, grid_y, grid_z, workgroup_x, count(*), avg(end-start), sum(end-start), min(end-start), max(end-start) "
                       "from kernels group by name, grid_x, grid_y, grid_z order by sum(end-start) desc"))
print('"kernel","workgroups_x","workgroups_y","workgroups_z","calls_per_batch","avg_us","min_us","max_us","ms_per_batch"')
for name, gx, gy, gz, wx, n, avg, tot, mn, mx in rows:
    if tot / 1e6 / nb < 0.02:
        continue
    print(f'"{name.split("(")[0].replace("void ", "")}",{gx // max(wx, 1)},{gy},{gz},{n / nb:.1f},{avg / 1e3:.1f},{mn / 1e3:.1f},{mx / 1e3:.1f},{tot / 1e6 / nb:.3f}')
