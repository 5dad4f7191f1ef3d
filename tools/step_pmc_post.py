"""gpurun_out/pmc3/{a,b,c}/p_counter_collection.csv (tools/step_pmc.sh) -> profiles/<tag>_step_pmc.json: every launch of one bench step of the
f16x2 plan under the SQ / TCC counters (last of the three steps; split-K conv + finishing pass folded)."""
import collections
import csv
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r03_x2'
src = os.path.join(root, 'gpurun_out', sys.argv[2] if len(sys.argv) > 2 else 'pmc3')              # round 5: one directory per launch schedule
meta = json.load(open(os.path.join(root, 'gpurun_out', sys.argv[3] if len(sys.argv) > 3 else 'launch_names.json')))
L, kpl = meta['launches'], meta['kernels_per_launch']
val = collections.defaultdict(dict)            # launch -> counter -> value
waves_grid = {}
for sub in 'abc':
    path = os.path.join(src, sub, 'p_counter_collection.csv')
    if not os.path.exists(path):
        import glob
        hits = glob.glob(os.path.join(src, sub, '**', '*counter_collection.csv'), recursive=True)
        if not hits:
            continue
        path = hits[0]
    by_disp = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        d = by_disp.setdefault(int(r['Dispatch_Id']), {'name': r['Kernel_Name'], 'c': {}, 'grid': int(r['Grid_Size']), 'wg': int(r['Workgroup_Size'])})
        d['c'][r['Counter_Name']] = d['c'].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    disp = [by_disp[k] for k in sorted(by_disp)]
    starts = [i for i, d in enumerate(disp) if 'u8_max_kernel' in d['name']]
    run = [d for d in disp[starts[-1]:] if 'fillBuffer' not in d['name']][:sum(kpl)]
    k = 0
    for li, n in enumerate(kpl):
        for d in run[k:k + n]:
            for c, v in d['c'].items():
                val[li][c] = val[li].get(c, 0.0) + v
        waves_grid[li] = sum(d['grid'] // 64 for d in run[k:k + n])
        k += n
out = []
for li, nm in enumerate(L):
    v = val.get(li, {})
    wc = v.get('SQ_WAVE_CYCLES', 0.0)
    kc = v.get('GRBM_GUI_ACTIVE', 0.0) / 8.0
    waves = v.get('SQ_WAVES', waves_grid.get(li, 0))
    g = lambda name: v.get(name)
    frac = lambda name: round(v[name] / wc, 4) if name in v and wc else None
    per_wave = lambda name: round(v[name] / waves, 1) if name in v and waves else None
    out.append({'launch': li, 'name': nm, 'kernel_cycles': round(kc) if kc else None, 'waves': int(waves), 'wave_cycles': wc,
                'busy_inst_any': frac('SQ_ACTIVE_INST_ANY'), 'busy_valu': frac('SQ_ACTIVE_INST_VALU'), 'busy_lds': frac('SQ_ACTIVE_INST_LDS'),
                'busy_vmem': frac('SQ_ACTIVE_INST_VMEM'), 'wait_inst_any': frac('SQ_WAIT_INST_ANY'), 'wait_any': frac('SQ_WAIT_ANY'),
                'valu_per_wave': per_wave('SQ_INSTS_VALU'), 'salu_per_wave': per_wave('SQ_INSTS_SALU'), 'lds_per_wave': per_wave('SQ_INSTS_LDS'),
                'vmem_rd_per_wave': per_wave('SQ_INSTS_VMEM_RD'), 'mfma_f16_ops_per_wave': per_wave('SQ_INSTS_VALU_MFMA_MOPS_F16'),
                'lds_conflict_frac': round(v['SQ_LDS_BANK_CONFLICT'] / v['SQ_LDS_IDX_ACTIVE'], 4) if v.get('SQ_LDS_IDX_ACTIVE') else None,
                'l2_hit': round(v['TCC_HIT_sum'] / v['TCC_REQ_sum'], 3) if v.get('TCC_REQ_sum') else None,
                'mfma_busy': round(v['SQ_VALU_MFMA_BUSY_CYCLES'] / (kc * 1024), 4) if kc and 'SQ_VALU_MFMA_BUSY_CYCLES' in v else None,
                'occupancy_waves_per_simd': round(wc / (kc * 1024), 2) if kc and wc else None,
                'valu_issue_util': round(v['SQ_INSTS_VALU'] * 4 / (kc * 1024), 3) if kc and 'SQ_INSTS_VALU' in v else None})
doc = {'_schedule': meta.get('schedule'),
       '_what': 'rocprofv3 --pmc, three separate passes (no trace domains: tools/step_pmc.sh / refresh_profiles5.sh) over tools/one_step.py 3 f16x2 <schedule> (= one bench.py step, B=32, one '
                'batch in flight); last step, per launch.  kernel_cycles = GRBM_GUI_ACTIVE / 8 XCDs; busy_* / wait_* are fractions of SQ_WAVE_CYCLES '
                '(wave-cycles with an instruction of that class in flight / waiting); occupancy = SQ_WAVE_CYCLES / (kernel_cycles * 1024 SIMDs); '
                'mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (kernel_cycles * 1024); lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; '
                'valu_issue_util = SQ_INSTS_VALU * 4 cycles / (kernel_cycles * 1024 SIMDs): the share of the launch during which a SIMD is issuing a VALU instruction',
       'launches': out}
json.dump(doc, open(os.path.join(root, 'profiles', f'{tag}_step_pmc.json'), 'w'), indent=1)
for o in out:
    print(f"{o['launch']:2d} {o['name'][:52]:52s} cyc {o['kernel_cycles']:7d} valu/wave {o['valu_per_wave']:7.1f} valu_issue {o['valu_issue_util']:.2f} mfma {o['mfma_busy']:.3f} lds/wave {o['lds_per_wave']:6.1f} conflict {o['lds_conflict_frac']} l2hit {o['l2_hit']}")
