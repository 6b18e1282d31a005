"""Dev tool: sums of arbitrary PMC counters per kernel name from one rocprofv3 --pmc pass.  Usage: pmc_generic.py <db> [name-filter]"""
import collections, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else 'la_conv'
acc = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(float); cnt = collections.defaultdict(int); seen = set()
for name, cn, v, d, did in db.execute("select kernel_name, counter_name, value, duration, dispatch_id from counters_collection where kernel_name like ?", ('%' + flt + '%',)):
    key = name.split('(')[0].replace('void ', '')
    acc[key][cn] += v
    if (did, key) not in seen:
        seen.add((did, key)); dur[key] += d; cnt[key] += 1
for k in acc:
    print(k, 'launches', cnt[k], 'total_ms %.2f' % (dur[k] / 1e6))
    for cn, v in sorted(acc[k].items()):
        print('   %-28s %.4g  (per launch %.4g, per us %.4g)' % (cn, v, v / cnt[k], v / (dur[k] / 1e3)))
