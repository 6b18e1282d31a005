"""dev tool: largest per-shape differences between two shape_stats CSVs (ms per batch, A - B)."""
import csv, sys
def load(f):
    d = {}
    for r in csv.DictReader(open(f)):
        d[(r['kernel'], r['workgroups_x'], r['workgroups_y'], r['workgroups_z'])] = (float(r['calls_per_batch']), float(r['avg_us']), float(r['ms_per_batch']))
    return d
a, b = load(sys.argv[1]), load(sys.argv[2])
print('total ms per batch: A %.2f  B %.2f' % (sum(v[2] for v in a.values()), sum(v[2] for v in b.values())))
rows = sorted(((a.get(k, (0, 0, 0))[2] - b.get(k, (0, 0, 0))[2], k) for k in set(a) | set(b)), key=lambda r: -abs(r[0]))
for d, k in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 20]:
    va, vb = a.get(k, (0, 0, 0)), b.get(k, (0, 0, 0))
    print(f"{d:+8.3f} ms  {k[0][:46]:46s} {','.join(k[1:]):14s} A: n={va[0]:.0f} {va[1]:.1f}us   B: n={vb[0]:.0f} {vb[1]:.1f}us")
