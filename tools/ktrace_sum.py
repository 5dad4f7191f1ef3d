"""Per-launch kernel durations of one plan run from a rocprofv3 --kernel-trace CSV (median over the runs in the trace).

    python tools/ktrace_sum.py <dir-or-csv> [first-kernel-substring=u8_max_kernel]
A "run" starts at each occurrence of the first kernel; runs with the most common length are aligned position by position.
"""
import collections
import csv
import glob
import os
import sys

path = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else 'u8_max_kernel'
if os.path.isdir(path):
    path = glob.glob(os.path.join(path, '**', '*kernel_trace.csv'), recursive=True)[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if first in r['Kernel_Name']]
runs = [rows[a:b] for a, b in zip(starts, starts[1:] + [len(rows)])]
length = collections.Counter(len(r) for r in runs).most_common(1)[0][0]
runs = [r for r in runs if len(r) == length][2:]          # drop the first two (cold) runs
tot = 0.0
for i in range(length):
    d = sorted((int(r[i]['End_Timestamp']) - int(r[i]['Start_Timestamp'])) / 1000 for r in runs)
    med = d[len(d) // 2]
    tot += med
    nm = runs[0][i]['Kernel_Name'].replace('(anonymous namespace)::', '')
    print(f'{i:3d} {nm[:84]:84s} {med:8.1f} us')
spans = sorted((int(r[-1]['End_Timestamp']) - int(r[0]['Start_Timestamp'])) / 1000 for r in runs)
print(f'runs {len(runs)}  launches {length}  sum of kernels {tot:.1f} us  span first-start..last-end {spans[len(spans) // 2]:.1f} us')
