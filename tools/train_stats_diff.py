"""Per-kernel comparison of two rocprofv3 kernel_stats.csv of the training step: python tools/train_stats_diff.py new.csv old.csv [replays_new replays_old]
(per-step figures: calls and microseconds divided by the number of graph replays + eager steps in the trace; default = calls of yolo_loss_kernel / 2)."""
import csv, sys
def load(p):
    d = {}
    for r in csv.DictReader(open(p)):
        d[r['Name'].split('(')[0][:64]] = (int(r['Calls']), float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e3)
    return d
new, old = load(sys.argv[1]), load(sys.argv[2])
def steps(d):
    k = [v for n, v in d.items() if n.startswith('yolo_loss_kernel')]
    return k[0][0] / 2.0                                    # two output layers per step
na = float(sys.argv[3]) if len(sys.argv) > 3 else steps(new)
oa = float(sys.argv[4]) if len(sys.argv) > 4 else steps(old)
keys = sorted(set(new) | set(old), key=lambda k: -(new.get(k, (0, 0, 0))[2] / na + old.get(k, (0, 0, 0))[2] / oa))
for k in keys[:int(sys.argv[5]) if len(sys.argv) > 5 else 45]:
    n, o = new.get(k, (0, 0, 0)), old.get(k, (0, 0, 0))
    print(f'{k:64s} new {n[0] / na:6.1f} x {n[1]:7.1f} = {n[2] / na:8.1f} | old {o[0] / oa:6.1f} x {o[1]:7.1f} = {o[2] / oa:8.1f}')
skip = ('letterbox', 'u8_normalise', 'u8_image_max')
print('launches per step new %.0f old %.0f' % (sum(v[0] for k, v in new.items() if not k.startswith(skip)) / na, sum(v[0] for k, v in old.items() if not k.startswith(skip)) / oa))
print('sum of kernels per step (us) new %.0f old %.0f' % (sum(v[2] for k, v in new.items() if not k.startswith(skip)) / na, sum(v[2] for k, v in old.items() if not k.startswith(skip)) / oa))
