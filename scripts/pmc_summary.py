"""Dev tool: per-kernel mean of every PMC counter in a rocprofv3 rocpd database (one row per kernel name)."""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else '%'
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for name, cn, v, d in db.execute("select kernel_name, counter_name, value, duration from counters_collection where kernel_name like ?", (pat,)):
    acc[name][cn].append(v)
for name, cs in acc.items():
    print(name[:100])
    for cn, vs in sorted(cs.items()):
        print(f'    {cn:36s} n={len(vs):4d} mean {sum(vs) / len(vs):16.1f}')
