"""gpurun_out/prof3/ (rocprofv3 CSVs of tools/one_step.py: --kernel-trace --stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE) -> profiles/<tag>_*.

    python tools/prof_post3.py r03_x2
    python tools/prof_post3.py r04_x2 prof4/throughput launch_names_throughput.json        (round 4: one directory per launch schedule)

A step of the plan is found in the kernel stream by its first kernel (u8_max); launches that issue two kernels (split-K conv + finish)
are folded; the hipMemsetAsync of the f16x2 plan (a fill kernel) is listed on its own."""
import collections
import csv
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r03_x2'
src = os.path.join(root, 'gpurun_out', sys.argv[2] if len(sys.argv) > 2 else 'prof3')
meta = json.load(open(os.path.join(root, 'gpurun_out', sys.argv[3] if len(sys.argv) > 3 else 'launch_names.json')))
L, kpl, alg = meta['launches'], meta['kernels_per_launch'], meta['alg_bytes_per_image']
basis = 0.5 if meta.get('precision') == 'f16x2' else 1.0      # SURVEY 8(d) counts fp16 bytes; the f16x2 plan reports 4 B per element


def steps_of(rows, key):
    rows = sorted(rows, key=key)
    starts = [i for i, r in enumerate(rows) if 'u8_max_kernel' in r['Kernel_Name']]
    runs = [rows[a:b] for a, b in zip(starts, starts[1:] + [len(rows)])]
    want = sum(kpl)
    runs = [[r for r in run if 'fillBuffer' not in r['Kernel_Name']][:want] for run in runs]
    return [r for r in runs if len(r) == want][1:]            # drop the first (cold) step


def fold(values):
    out, k = [], 0
    for n in kpl:
        out.append(sum(values[k:k + n]))
        k += n
    return out


tr = list(csv.DictReader(open(os.path.join(src, 'stats', 'p_kernel_trace.csv'))))
runs = steps_of(tr, lambda r: int(r['Start_Timestamp']))
dur = [fold([(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in run]) for run in runs]
avg = [sum(d[i] for d in dur) / len(dur) for i in range(len(L))]
first_kernel = []
k = 0
for n in kpl:
    first_kernel.append(runs[0][k])
    k += n
traffic = {}
for cnt, sub in (('FETCH_SIZE', 'fetch'), ('WRITE_SIZE', 'write')):
    path = os.path.join(src, sub, 'p_counter_collection.csv')
    if not os.path.exists(path):
        continue
    rows = [r for r in csv.DictReader(open(path)) if r['Counter_Name'] == cnt]
    rr = steps_of(rows, lambda r: int(r['Dispatch_Id']))
    per = [fold([float(r['Counter_Value']) for r in run]) for run in rr]
    traffic[cnt] = [sum(p[i] for p in per) / len(per) * 1024 * (2 if cnt == 'FETCH_SIZE' else 1) for i in range(len(L))]
with open(os.path.join(root, 'profiles', f'{tag}_kernel_trace_per_launch.csv'), 'w', newline='') as fh:
    fh.write(f'# rocprofv3 --kernel-trace --stats on tools/one_step.py ({meta.get("precision")} plan, {meta.get("schedule", "per-layer")} schedule, B=32, one batch in flight): average duration per '
             f'launch of the step over {len(dur)} steps (split-K conv + its finishing pass folded); HBM bytes from separate --pmc FETCH_SIZE / '
             'WRITE_SIZE passes (FETCH x2 per MI355X_MICROARCH.md); algorithmic bytes = SURVEY 8(d) basis (fp16 in + out)\n')
    wr = csv.writer(fh)
    wr.writerow(['launch', 'name', 'kernel', 'kernels', 'avg_us', 'vgpr', 'lds', 'fetch_bytes_x2', 'write_bytes', 'total_bytes', 'algorithmic_bytes'])
    for i, nm in enumerate(L):
        f = traffic.get('FETCH_SIZE', [None] * len(L))[i]
        w = traffic.get('WRITE_SIZE', [None] * len(L))[i]
        a = alg[i] * 32 * basis if i < len(alg) else None
        r0 = first_kernel[i]
        wr.writerow([i, nm, r0['Kernel_Name'].replace('(anonymous namespace)::', '')[:80], kpl[i], round(avg[i], 2), r0['VGPR_Count'], r0['LDS_Block_Size'],
                     None if f is None else round(f), None if w is None else round(w), None if f is None or w is None else round(f + w),
                     None if a is None else round(a)])
    wr.writerow(['', 'SUM', '', sum(kpl), round(sum(avg), 1), '', '', *(round(sum(traffic[c])) if c in traffic else '' for c in ('FETCH_SIZE', 'WRITE_SIZE')), '', ''])
if len(traffic) == 2:
    json.dump({f'{i}:{nm}': round(traffic['FETCH_SIZE'][i] + traffic['WRITE_SIZE'][i]) for i, nm in enumerate(L)},
              open(os.path.join(root, 'profiles', f'{tag}_hbm_traffic.json'), 'w'), indent=1)
st = os.path.join(src, 'stats', 'p_kernel_stats.csv')
if os.path.exists(st):
    open(os.path.join(root, 'profiles', f'{tag}_kernel_stats.csv'), 'w').write(open(st).read())
for extra, dst in (('bench_line.json', f'{tag.split("_")[0]}_bench_line.json'), ('x2_per_launch.txt', f'{tag.split("_")[0]}_f16x2_per_launch.txt'),
                   ('train/p_kernel_stats.csv', f'{tag.split("_")[0]}_train_kernel_stats.csv'), ('bench_f16.json', f'{tag.split("_")[0]}_bench_line_f16.json')):
    if os.path.exists(os.path.join(src, extra)):
        open(os.path.join(root, 'profiles', dst), 'w').write(open(os.path.join(src, extra)).read())
print(f'{tag}: {len(L)} launches, sum of kernels {sum(avg):.1f} us over {len(dur)} steps' +
      (f", HBM traffic {sum(traffic['FETCH_SIZE']) / 1e6 + sum(traffic['WRITE_SIZE']) / 1e6:.0f} MB/step" if len(traffic) == 2 else ''))
