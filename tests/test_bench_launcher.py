"""bench.py's own N-rank launcher (BASELINE metric "1/2/4/8-GPU scaling"): `python bench.py --gpus N` must start N
ranks by itself, report n_gpus = N with the slowest rank defining the time, and refuse to oversubscribe a box with fewer devices.
CPU only: `--stub` swaps the GPU step for a sleep and RCCL for gloo; everything else is the bench's real code path."""
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def _run(*args, timeout=240):
    return subprocess.run([sys.executable, str(ROOT / 'bench.py'), *args], capture_output=True, text=True, timeout=timeout, cwd=str(ROOT))


def _line(r):
    rows = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert r.returncode == 0 and len(rows) == 1, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    return json.loads(rows[0])


def test_gpus_2_starts_two_ranks_by_itself_and_the_slow_rank_defines_the_time():
    out = _line(_run('--gpus', '2', '--stub', '--steps', '20', '--warmup', '2'))
    assert out['n_gpus'] == 2 and out['steps'] == 20 and out['warmup'] == 2 and out['stub'] is True
    assert out['config']['global_batch'] == 64 and out['config']['images_of_rank0'] == 32 and out['scaling'] == 'weak'
    # rank 1 sleeps 3 ms per step, rank 0 2 ms: max over ranks
    assert out['ms_per_step'] >= 3.0
    assert abs(out['value'] - 64 / (out['ms_per_step'] * 1e-3)) / out['value'] < 0.01          # whole-job images / slowest rank's time


def test_single_rank_stub_line_has_the_contract_fields():
    out = _line(_run('--stub', '--steps', '10', '--warmup', '1'))
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
              'data', 'config'):
        assert k in out
    assert out['n_gpus'] == 1 and out['vs_baseline'] is None


def test_more_ranks_than_devices_is_refused_loudly():
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    r = _run('--gpus', str(have + 2), '--steps', '1', '--warmup', '0')
    assert r.returncode != 0
    assert f'{have + 2} ranks requested, {have} device(s) visible' in (r.stderr + r.stdout)


def test_a_rank_count_that_contradicts_the_launcher_is_refused():
    import os
    env = dict(os.environ, WORLD_SIZE='2', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, str(ROOT / 'bench.py'), '--gpus', '4', '--stub'], capture_output=True, text=True, timeout=120,
                       cwd=str(ROOT), env=env)
    assert r.returncode != 0 and 'the launcher started 2 ranks' in (r.stderr + r.stdout)


def test_gpus_8_stub_eight_ranks_report_every_rank_and_the_slowest_defines_the_time():
    """SURVEY 8(e) / BASELINE "1/2/4/8-GPU scaling": the 8-rank launch path (rendezvous on 127.0.0.1, image i -> rank i mod 8, barrier,
    max over ranks, per-rank rates in the line) exercised without an 8-GPU node."""
    out = _line(_run('--gpus', '8', '--stub', '--steps', '10', '--warmup', '1', timeout=400))
    assert out['n_gpus'] == 8 and out['config']['global_batch'] == 256 and out['config']['images_of_rank0'] == 32
    pr = out['from_host_per_rank']
    assert len(pr['images_per_sec']) == 8 and pr['min'] == min(pr['images_per_sec']) and pr['max'] == max(pr['images_per_sec'])
    # rank r sleeps 2 ms * (1 + r / 2) per step: rank 7 (9 ms) is the slowest and defines the whole-job time
    assert pr['images_per_sec'][7] == pr['min'] and pr['images_per_sec'][0] == pr['max']
    assert out['ms_per_step'] >= 9.0
    assert abs(out['value'] - 256 / (out['ms_per_step'] * 1e-3)) / out['value'] < 0.01
