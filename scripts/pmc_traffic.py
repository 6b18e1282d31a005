"""Dev tool: HBM traffic of the contraction path from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of
`bench.py --steps 1 --warmup 0`.  Usage: pmc_traffic.py <fetch.db> <write.db> <precision> <out.json>"""
import json
import sqlite3
import sys

PAT = ('la_conv', 'la_presplit', 'la_plane_absmax')


def total(path, counter):
    db = sqlite3.connect(path)
    tot, launches = 0.0, 0
    for name, cn, v in db.execute("select kernel_name, counter_name, value from counters_collection"):
        if cn != counter or not any(p in name for p in PAT):
            continue
        tot += v
        if ('la_conv_bf16' in name or 'la_conv_igemm' in name):
            launches += 1
    return tot * 1024.0, launches          # KB units


fetch, n1 = total(sys.argv[1], 'FETCH_SIZE')
write, n2 = total(sys.argv[2], 'WRITE_SIZE')
n = max(n1, n2, 1)
out = {
    'precision': sys.argv[3], 'launches': n, 'bytes_per_launch': (fetch + write) / n, 'fetch_bytes_per_launch': fetch / n,
    'write_bytes_per_launch': write / n, 'whole_batch_bytes': fetch + write,
    'note': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (KB units x1024), summed over the contraction kernels + '
            'their pre-split / absmax / split-K finish kernels and divided by the la_conv launches of one batch (bench.py --steps 1 '
            '--warmup 0). FETCH_SIZE is UNCORRECTED: the guide calibrates the gfx950 x2 under-count only for 16 B/lane streaming '
            'reads; these kernels mix 4 B/lane gathers with 16 B/lane weight fragments, so true fetched bytes lie between 1x and 2x '
            'the fetch figure.',
}
json.dump(out, open(sys.argv[4], 'w'), indent=1)
print(json.dumps(out)[:400])
