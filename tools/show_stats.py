"""Print the top rows of a rocprofv3 kernel_stats.csv (per-step averages):  python tools/show_stats.py <csv> [steps]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else None
tot = sum(float(r['TotalDurationNs']) for r in rows)
if steps:
    print(f'GPU time per step: {tot / steps / 1e3:.1f} us')
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    per = f" per-step={float(r['TotalDurationNs']) / steps / 1e3:7.1f}" if steps else ''
    print(f"{r['Name'][:58]:58s} calls={r['Calls']:>4s} avg={float(r['AverageNs']) / 1e3:8.1f}us{per} pct={float(r['Percentage']):5.1f}")
