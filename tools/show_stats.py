"""Print the kernel_stats.csv of a rocprofv3 --stats run (first matching file under the given directory)."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True))[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:n]:
    print(f"{r['Name'][:100]:100s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.1f}  {100*float(r['TotalDurationNs'])/tot:5.1f}%")
