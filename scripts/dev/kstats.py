"""Pretty-print a rocprofv3 kernel_stats.csv: per kernel calls, average us, us per step.  usage: kstats.py file.csv [steps]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = 0.0
for r in rows:
    n = r['Name']; n = n[:n.find('(')] if '(' in n else n
    t = float(r['TotalDurationNs']) / 1e3 / steps; tot += t
    print(f"{n[:72]:72s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:9.1f} us  per step {t:9.1f} us  {float(r['Percentage']):5.2f}%")
print(f"{'(listed kernels)':72s} per step {tot:9.1f} us")
