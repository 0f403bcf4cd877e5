"""Top kernels of a rocprofv3 --kernel-trace --stats run: python tools/kstats.py <dir> [steps] [rows]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f'total kernel time {tot / steps / 1e6:.2f} ms per step over {sum(int(r["Calls"]) for r in rows) / steps:.0f} launches per step')
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{r['Name'][:100]:100s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs']) / 1e3:9.1f} ms/step {float(r['TotalDurationNs']) / steps / 1e6:7.2f}")
