"""profiles/<tag>_kernel_stats.json from a rocprofv3 --kernel-trace --stats summary (kernel_stats.csv) of scripts/train_c5.py: per-kernel
average duration (us) and launches per training step, keyed to the library's sources (bench.py::kernel_stats_lookup reports them only on these).
usage: python scripts/parse_kernel_stats.py kernel_stats.csv STEPS OUT.json"""
import csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnkeras_amd._native import source_hash
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
avg, calls = {}, {}
for r in rows:
    n = r['Name']
    if 'gnn::' not in n and 'anonymous namespace' not in n: continue
    key = n[:n.find('(')].replace('void ', '').replace('gnn::', '').replace('(anonymous namespace)::', '')
    if float(r['TotalDurationNs']) / 1e3 / steps < 20.0: continue          # (kernels below 20 us a step: not listed)
    avg[key] = round(float(r['AverageNs']) / 1e3, 1)
    calls[key] = round(int(r['Calls']) / steps, 2)
json.dump({'source_hash': source_hash(), 'steps_profiled': steps, 'avg_us': avg, 'calls_per_step': calls}, open(sys.argv[3], 'w'), indent=1)
print(json.dumps(avg))
