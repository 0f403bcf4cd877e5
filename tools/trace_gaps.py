"""Idle analysis of a rocprofv3 kernel trace of bench.py: python tools/trace_gaps.py gpurun_out/prof_x/b_kernel_trace.csv
Prints the gaps between consecutive k_edge_layer launches of the last step and the timeline inside one of them."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
edges = [i for i, r in enumerate(rows) if 'k_edge_layer' in r['Kernel_Name']]
last = edges[-9:]
gaps = [(int(rows[b]['Start_Timestamp']) - int(rows[a]['End_Timestamp'])) / 1e3 for a, b in zip(last[:-1], last[1:])]
print('gaps between edge kernels of the last step (us):', [round(g, 1) for g in gaps], 'mean', round(sum(gaps) / len(gaps), 1))
a, b = last[-4], last[-3]
t0 = int(rows[a]['End_Timestamp'])
for r in rows[a + 1:b + 1]:
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} -> {(int(r['End_Timestamp']) - t0) / 1e3:8.1f}  s{r['Stream_Id']}  {r['Kernel_Name'][:60]}")
