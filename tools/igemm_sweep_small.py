import os, subprocess, sys
root = '.'
shapes = [(104, 104, 64, 128, 3), (52, 52, 128, 256, 3), (26, 26, 256, 512, 3)]
for (h, w, c1, c2, k) in shapes:
    line = f'{h}x{w} {c1}->{c2} k{k}: '
    for cfg in ['', '5', '11']:   # auto (64x128), 128x128, 256x128 on eight waves (developer build)
        env = dict(os.environ, YK_IGEMM_FORCE=cfg, YK_FORCE_MINK='64', PROFILE='1', KS=str(k))
        out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'igemm_one.py'), str(h), str(w), str(c1), str(c2), '16'],
                             env=env, capture_output=True, text=True).stdout
        hit = [l for l in out.splitlines() if f'_{c1}to{c2}[' in l]
        line += f' cfg{cfg or "A"}=' + (hit[0].split('[')[1].split(']')[0].replace('igemm_', '') + ':' + hit[0].split()[-2] if hit else 'ERR')
    print(line, flush=True)
