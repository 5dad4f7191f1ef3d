"""gpurun_out/prof/ (rocprofv3 CSVs of tools/one_step.py and bench.py) -> profiles/r01_* summaries (run locally after the GPU call)."""
import csv, json, os, sys, collections
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, 'gpurun_out', 'prof')
tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
names = json.load(open(os.path.join(root, 'gpurun_out', 'launch_names.json')))
L, steps = names['launches'], names['steps']

def per_launch(counter_csv, counter):
    rows = [r for r in csv.DictReader(open(counter_csv)) if r['Counter_Name'] == counter]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    rows = rows[-len(L) * steps:]
    acc = collections.defaultdict(list)
    for i, r in enumerate(rows):
        acc[i % len(L)].append((float(r['Counter_Value']), r['Kernel_Name'], int(r['Grid_Size'])))
    return acc

out = []
f = per_launch(os.path.join(src, 'fetch', 'p_counter_collection.csv'), 'FETCH_SIZE')
w = per_launch(os.path.join(src, 'write', 'p_counter_collection.csv'), 'WRITE_SIZE')
by_launch = {}
for i, nm in enumerate(L):
    fetch = sum(v[0] for v in f[i]) / len(f[i]) * 1024 * 2       # KiB; x2: gfx950 reports half of wide coalesced reads (MI355X_MICROARCH.md)
    write = sum(v[0] for v in w[i]) / len(w[i]) * 1024
    alg = names['alg_bytes_per_image'][i] * 32 if i < len(names['alg_bytes_per_image']) else None
    out.append([i, nm, f[i][0][1][:70], f[i][0][2], round(fetch), round(write), round(fetch + write), alg])
    by_launch[f'{i}:{nm}'] = round(fetch + write)
with open(os.path.join(root, 'profiles', f'{tag}_hbm_traffic.csv'), 'w', newline='') as fh:
    fh.write('# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on tools/one_step.py (= one bench.py step, B=32); per launch.\n')
    fh.write('# FETCH_SIZE (KiB) doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads); WRITE_SIZE as reported.\n')
    wr = csv.writer(fh)
    wr.writerow(['launch', 'name', 'kernel', 'grid', 'fetch_bytes_x2', 'write_bytes', 'total_bytes', 'algorithmic_bytes'])
    wr.writerows(out)
json.dump(by_launch, open(os.path.join(root, 'profiles', f'{tag}_hbm_traffic.json'), 'w'), indent=1)
st = os.path.join(src, 'stats', 'p_kernel_stats.csv')
if os.path.exists(st):
    with open(st) as a, open(os.path.join(root, 'profiles', f'{tag}_kernel_stats.csv'), 'w') as b:
        b.write(a.read())
print('ok', len(out), 'launches; total traffic MB/step', sum(r[6] for r in out) / 1e6)

# per-launch durations from the kernel trace of the same script (dispatch order, last `steps` steps)
tr = os.path.join(src, 'stats', 'p_kernel_trace.csv')
if os.path.exists(tr):
    rows = sorted(csv.DictReader(open(tr)), key=lambda r: int(r['Start_Timestamp']))
    n_steps = min(40, len(rows) // len(L) - 2)
    rows = rows[-len(L) * n_steps:]
    acc = collections.defaultdict(list)
    for i, r in enumerate(rows):
        acc[i % len(L)].append(((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r))
    with open(os.path.join(root, 'profiles', f'{tag}_kernel_trace_per_launch.csv'), 'w', newline='') as fh:
        fh.write('# rocprofv3 --kernel-trace on tools/one_step.py 50 (= bench.py step, B=32, one batch in flight): duration per launch of '
                 f'the step, dispatch order, last {n_steps} steps\n')
        wr = csv.writer(fh)
        wr.writerow(['launch', 'name', 'kernel', 'avg_us', 'min_us', 'max_us', 'vgpr', 'lds'])
        tot = 0.0
        for i, nm in enumerate(L):
            d = [v[0] for v in acc[i]]
            r0 = acc[i][0][1]
            tot += sum(d) / len(d)
            wr.writerow([i, nm, r0['Kernel_Name'][:90], round(sum(d) / len(d), 2), round(min(d), 2), round(max(d), 2), r0['VGPR_Count'],
                         r0['LDS_Block_Size']])
    print('per-launch trace: sum of averages %.1f us' % tot)
for extra, dst in (('train/p_kernel_stats.csv', f'{tag}_train_kernel_stats.csv'), ('bench_line.json', f'{tag}_bench_line.json'),
                   ('x2_per_launch.txt', f'{tag}_f16x2_per_launch.txt'), ('drift_v1.txt', f'{tag}_fp16_drift_per_layer.txt')):
    if os.path.exists(os.path.join(src, extra)):
        with open(os.path.join(src, extra)) as a, open(os.path.join(root, 'profiles', dst), 'w') as b:
            b.write(a.read())
