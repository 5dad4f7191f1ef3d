"""Dev tool (developer build, `make DEV=1`): per-launch times of the f16x2 plan under several environment settings, side by side.

    python tools/xsweep.py "YK_X_NS=2" "YK_X_NS=4" "YK_X_CFG=0 YK_X_NS=3" ...
"""
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cols = []
for setting in ['base'] + sys.argv[1:]:
    env = dict(os.environ)
    if setting != 'base':
        for kv in setting.split():
            k, v = kv.split('=')
            env[k] = v
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'xbench.py')], env=env, capture_output=True, text=True, cwd=root).stdout
    rows = []
    for line in out.splitlines():
        parts = line.split()
        if len(parts) >= 3 and parts[2] == 'us':
            rows.append((parts[0], float(parts[1])))
        elif line.startswith('sum'):
            rows.append(('SUM', float(parts[1])))
        elif line.startswith('max|ref|'):
            rows.append(('err ' + parts[-1], 0.0))
    cols.append((setting, rows))
n = max(len(r) for _, r in cols)
print(' ' * 58 + ''.join(f'{s[-22:]:>24s}' for s, _ in cols))
for i in range(n):
    name = ''
    vals = []
    for _, r in cols:
        if i < len(r):
            name = name or r[i][0].split('[')[0]
            vals.append(f'{r[i][0].split("[")[1][:-1] if "[" in r[i][0] else "":>15s} {r[i][1]:8.1f}')
        else:
            vals.append(' ' * 24)
    print(f'{name:58s}' + ''.join(vals))
