"""Dev tool: how much do kernels of different streams overlap?  Reads a rocprofv3 --kernel-trace CSV of `bench.py --steps N` and prints,
for the steady part, the busy time (union of kernel intervals), the sum of kernel durations, the mean number of kernels in flight and the
per-kernel duration inflation relative to the one-batch-in-flight profile (profiles/<tag>_kernel_trace_per_launch.csv)."""
import collections
import csv
import glob
import os
import sys

path = sys.argv[1]
if os.path.isdir(path):
    path = glob.glob(os.path.join(path, '**', '*kernel_trace.csv'), recursive=True)[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r['Start_Timestamp']))
n = len(rows)
rows = rows[n // 3: n - n // 10]                                   # steady part
iv = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', r.get('Stream_Id', '?'))) for r in rows]
t0, t1 = iv[0][0], max(e for _, e, _, _ in iv)
# union
busy, cur_s, cur_e = 0, None, None
for s, e, _, _ in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _, _ in iv)
print(f'kernels {len(iv)}  span {(t1 - t0) / 1e3:.1f} us  busy(union) {busy / 1e3:.1f} us  sum of durations {tot / 1e3:.1f} us  mean in flight {tot / busy:.2f}  idle {(1 - busy / (t1 - t0)) * 100:.1f} %')
per = collections.defaultdict(list)
for s, e, nm, _ in iv:
    per[nm.replace('(anonymous namespace)::', '')[:60]].append((e - s) / 1e3)
print('per kernel (mean us, count):')
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:24]:
    print(f'  {k:60s} {sum(v) / len(v):8.1f} x{len(v)}')
print('queues:', collections.Counter(q for _, _, _, q in iv))
