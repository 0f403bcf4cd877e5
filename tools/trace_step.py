"""Timeline of one benchmark step from a rocprofv3 --kernel-trace CSV: per kernel start / end relative to the step's first kernel,
the stream (queue) it ran on, and the idle time of the device between the edge kernels.
usage: python tools/trace_step.py <kernel_trace.csv> [step_index]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
name = lambda r: r['Kernel_Name'].split('(')[0].replace('void ', '')[:44]
edge = [i for i, r in enumerate(rows) if 'k_edge_layer' in r['Kernel_Name']]
steps = [edge[i:i + 9] for i in range(0, len(edge) - 8, 9)]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(steps) - 2
lo = steps[k - 1][-1] + 1 if k > 0 else 0
hi = steps[k + 1][0] if k + 1 < len(steps) else len(rows)
# the step starts after the previous step's last edge kernel's followers: take the first kernel after the previous output stage
t0 = None
for i in range(lo, hi):
    r = rows[i]
    if t0 is None and ('k_edge_in' in r['Kernel_Name'] or 'k_linear' in r['Kernel_Name'] or 'k_y_in' in r['Kernel_Name']):
        t0 = int(r['Start_Timestamp'])
        first = i
for i in range(first - 12, hi):
    r = rows[i]
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    if s > 16_000_000:
        break
    print(f"{s / 1e3:10.1f} {e / 1e3:10.1f} {(e - s) / 1e3:8.1f}  q{r.get('Queue_Id', '?'):>3} {name(r)}")
