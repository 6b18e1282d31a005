"""Dev tool: MFMA busy fraction per contraction kernel from one rocprofv3 --pmc pass of `bench.py --steps 1 --warmup 0`
(counters SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY).  Usage: pmc_mfma.py <db> <out.json>"""
import collections
import json
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
acc = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(float)
cnt = collections.defaultdict(int)
seen = set()
for name, cn, v, d, did in db.execute("select kernel_name, counter_name, value, duration, dispatch_id from counters_collection where kernel_name like '%la_conv%'"):
    key = name.split('(')[0].replace('void ', '')
    acc[key][cn] += v
    if (did, key) not in seen:
        seen.add((did, key))
        dur[key] += d
        cnt[key] += 1
out = {'note': 'SQ_VALU_MFMA_BUSY_CYCLES counts cycles (16 per v_mfma_f32_16x16x32_f16, the instruction of the default contraction kernels; 32 per v_mfma_f32_32x32x16_f16, the 64- / 32-row forms); SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles summed over '
               'waves.  mfma_busy_frac = MFMA busy cycles / (1024 SIMDs x kernel time x 2.4 GHz): a LOWER bound on the pipe utilisation '
               '(the sustained clock under MFMA load is 1.9-2.3 GHz).  Kernel durations under --pmc are longer than in the kernel-trace run.',
       'kernels': {}}
for k in acc:
    t_ns = dur[k]
    busy = acc[k].get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)
    out['kernels'][k] = {
        'launches': cnt[k], 'total_ms': t_ns / 1e6, 'mfma_busy_cycles': busy,
        'mfma_busy_frac_at_2.4GHz': busy / (1024 * t_ns * 2.4) if t_ns else None,
        'wave_quadcycles': acc[k].get('SQ_WAVE_CYCLES'), 'wait_any_frac': (acc[k].get('SQ_WAIT_ANY', 0) / acc[k]['SQ_WAVE_CYCLES']) if acc[k].get('SQ_WAVE_CYCLES') else None,
        'wait_inst_any_frac': (acc[k].get('SQ_WAIT_INST_ANY', 0) / acc[k]['SQ_WAVE_CYCLES']) if acc[k].get('SQ_WAVE_CYCLES') else None,
    }
json.dump(out, open(sys.argv[2], 'w'), indent=1)
print(json.dumps(out['kernels'], indent=1)[:1500])
