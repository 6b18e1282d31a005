"""Dev tool: print value / ms per batch / per-class launches and ms of bench JSON lines (files given as arguments)."""
import json
import sys

for path in sys.argv[1:]:
    for line in open(path):
        line = line.strip()
        if not line.startswith('{'):
            continue
        d = json.loads(line)
        cls = d.get('roofline', {}).get('classes', {})
        print(f"{path}: {d['value']:.2f} {d['unit']}  {d['ms_per_step']:.2f} ms | " +
              '  '.join(f"{k} {v['launches_per_batch']}:{v['ms_per_batch']:.2f}" for k, v in cls.items()))
