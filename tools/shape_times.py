"""Per grid shape: launches per step and median duration of the named kernels in a rocprofv3 kernel trace of tools/train_step.py.
    python tools/shape_times.py <kernel_trace.csv> <executions of the step in the trace> <kernel substring>..."""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
execs = float(sys.argv[2])
for pat in sys.argv[3:]:
    d = collections.defaultdict(list)
    for r in rows:
        if pat in r['Kernel_Name']:
            d[(int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), int(r['Grid_Size_Y']), int(r['Grid_Size_Z']))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    print(pat)
    tot = 0
    for k, v in sorted(d.items(), key=lambda kv: -sorted(kv[1])[len(kv[1]) // 2] * len(kv[1])):
        v.sort()
        n = len(v) / execs
        tot += v[len(v) // 2] * n
        print('   grid', k, 'per step %.1f' % n, 'median %.1f us' % v[len(v) // 2])
    print('   total per step %.0f us' % tot)
