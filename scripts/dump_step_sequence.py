"""Dev tool: the kernel launches of ONE optimisation step in issue order (name, grid, duration, gap to the previous kernel's end) from a
rocprofv3 kernel trace of `bench.py --steps 1 --warmup 1 --no-graph --lanes 1 --no-roofline --no-cpu-baseline`.
Usage: dump_step_sequence.py <db> [step index from the end, default 3]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = list(db.execute("select name, grid_x, grid_y, grid_z, workgroup_x, start, end from kernels order by start"))
# a step starts with la_affine_fwd_kernel
starts = [i for i, r in enumerate(rows) if 'la_affine_fwd' in r[0]]
lo, hi = starts[-back - 1], starts[-back]
prev_end = rows[lo][5]
tot = 0.0
print(f'# step of {hi - lo} launches')
for name, gx, gy, gz, wx, st, en in rows[lo:hi]:
    short = name.split('(')[0].replace('void ', '')
    print(f'{short[:60]:60s} ({gx // max(wx, 1)},{gy},{gz})  {1e-3 * (en - st):7.1f} us  gap {1e-3 * (st - prev_end):6.1f}')
    tot += en - st
    prev_end = en
print(f'# sum of kernel times {tot * 1e-3:.1f} us, wall {1e-3 * (rows[hi][5] - rows[lo][5]):.1f} us')
