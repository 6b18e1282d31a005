"""Dev tool: how the two stream lanes share the chip, from a rocprofv3 kernel trace of the timed schedule (graph replay, lanes side by
side): wall time of the traced batches split by what is running -- no kernel, only small / HBM-class kernels, one MFMA-class launch
(halo / flat / 32^2 split-K), two of them at once.      lanes_timeline.py <db> [skip_first_ms]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, grid_x, grid_y, grid_z, workgroup_x, start, end from kernels order by start"))
t0 = rows[0][5]
skip = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 0.0


def is_m(name, wgs):
    if 'la_conv_bf16_halo_kernel' in name:
        return True
    if 'la_conv_bf16_kernel' in name and (', false,' in name or wgs >= 400):
        return True
    return False


ev = []
for name, gx, gy, gz, wx, st, en in rows:
    if st - t0 < skip:
        continue
    wgs = (gx // max(wx, 1)) * gy * gz
    m = is_m(name, wgs)
    ev.append((st, 1, m))
    ev.append((en, -1, m))
ev.sort()
nm = no = 0
last = ev[0][0]
acc = {'idle': 0.0, 'other only': 0.0, 'one M': 0.0, 'two+ M': 0.0, 'M + other': 0.0}
for t, d, m in ev:
    dt = t - last
    if dt > 0:
        if nm == 0 and no == 0:
            acc['idle'] += dt
        elif nm == 0:
            acc['other only'] += dt
        elif nm >= 2:
            acc['two+ M'] += dt
        elif no > 0:
            acc['M + other'] += dt
        else:
            acc['one M'] += dt
    last = t
    if m:
        nm += d
    else:
        no += d
tot = sum(acc.values())
print(f'traced wall {tot * 1e-6:.1f} ms')
for k, v in acc.items():
    print(f'  {k:12s} {v * 1e-6:8.2f} ms  {100 * v / tot:5.1f} %')
