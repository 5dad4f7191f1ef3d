"""gpurun_out/<call>/inflight_d*.json (tools/inflight_counters.py, device-wide counters while D batches are in flight) -> derived chip-wide
utilisations.   python tools/inflight_post.py out.json in1.json [in2.json ...]   (later files add / override counter sets of the same depth)
Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave; SQ_VALU_MFMA_BUSY_CYCLES, SQ_LDS_*,
GRBM_GUI_ACTIVE count cycles; GRBM has one instance per XCD (8), SQ one per shader engine (32), TCC/TCP/TA are summed over their instances."""
import json, sys
CUS, SIMDS, XCDS = 256, 1024, 8
out, ins = sys.argv[1], sys.argv[2:]
by_depth = {}
for f in ins:
    d = json.load(open(f))
    e = by_depth.setdefault(d['depth'], {'depth': d['depth'], 'batch': d['batch'], 'steps': d['steps'], 'sets': {}, 'plain_images_per_sec': d['plain_images_per_sec']})
    e['sets'].update({k: v for k, v in d['sets'].items() if 'counters' in v})
res = {}
for depth, d in sorted(by_depth.items()):
    S = d['sets']

    def c(set_, name):
        return S[set_]['counters'][name]['sum']

    def cyc(set_):                                                   # shader cycles of the region (per XCD average)
        return c(set_, 'GRBM_GUI_ACTIVE') / XCDS
    r = {'images_per_sec_plain': round(d['plain_images_per_sec'], 1),
         'images_per_sec_under_counters': {k: round(v['images_per_sec'], 1) for k, v in S.items()},
         'steps_per_region': d['steps'], 'batch': d['batch']}
    if 'waves' in S:
        sc = cyc('waves') * SIMDS
        wc = 4 * c('waves', 'SQ_WAVE_CYCLES')
        r['shader_clock_GHz'] = round(cyc('waves') / S['waves']['seconds'] / 1e9, 3)
        r['resident_waves_per_simd'] = round(wc / sc, 3)
        r['wave_cycle_split'] = {'parked_at_waitcnt_or_barrier': round(c('waves', 'SQ_WAIT_ANY') / c('waves', 'SQ_WAVE_CYCLES'), 3),
                                 'stalled_at_issue': round(c('waves', 'SQ_WAIT_INST_ANY') / c('waves', 'SQ_WAVE_CYCLES'), 3),
                                 'issuing': round(c('waves', 'SQ_ACTIVE_INST_ANY') / c('waves', 'SQ_WAVE_CYCLES'), 3)}
        r['cu_busy_frac'] = round(4 * c('waves', 'SQ_BUSY_CU_CYCLES') / (cyc('waves') * CUS) / 4, 3)
        r['waves_launched_per_step'] = round(c('waves', 'SQ_WAVES') / d['steps'], 1)
    if 'issue' in S:
        sc = cyc('issue') * SIMDS
        r['per_simd_issue_utilisation'] = {
            'valu': round(4 * c('issue', 'SQ_ACTIVE_INST_VALU') / sc, 3),
            'mfma_pipe_busy': round(c('issue', 'SQ_VALU_MFMA_BUSY_CYCLES') / sc, 3),
            'salu_per_cu_scalar_unit': round(4 * c('issue', 'SQ_ACTIVE_INST_SCA') / (cyc('issue') * CUS) / 4, 3),
            'lds_issue': round(4 * c('issue', 'SQ_ACTIVE_INST_LDS') / sc, 3),
            'vmem_issue': round(4 * c('issue', 'SQ_ACTIVE_INST_VMEM') / sc, 3),
            'misc (barrier, nop, sleep)': round(4 * c('issue', 'SQ_ACTIVE_INST_MISC') / sc, 3)}
    if 'insts' in S:
        st, B = d['steps'], d['batch']
        r['wave_instructions_per_image'] = {k[9:].lower(): round(c('insts', k) / st / B) for k in
                                            ('SQ_INSTS_VALU', 'SQ_INSTS_MFMA', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR', 'SQ_INSTS_SALU', 'SQ_INSTS_SMEM')}
        r['valu_active_lanes_avg'] = round(c('insts', 'SQ_THREAD_CYCLES_VALU') / c('insts', 'SQ_INSTS_VALU'), 1)
    if 'coexec' in S:
        v = c('coexec', 'SQ_ACTIVE_INST_VALU')
        r['valu_mix'] = {k[14:].lower(): round(c('coexec', k) / v, 3) for k in ('SQ_INSTS_VALU_INT32', 'SQ_INSTS_VALU_FMA_F32', 'SQ_INSTS_VALU_MUL_F32', 'SQ_INSTS_VALU_CVT', 'SQ_INSTS_VALU_ADD_F32')}
        r['mfma_cycles_coexecuting_with_valu_frac'] = round(c('coexec', 'SQ_VALU_MFMA_COEXEC_CYCLES') / c('coexec', 'SQ_VALU_MFMA_BUSY_CYCLES'), 3)
    if 'lds' in S:
        cc = cyc('lds') * CUS
        r['lds'] = {'active_frac_of_cu_cycles': round(c('lds', 'SQ_LDS_IDX_ACTIVE') / cc, 3),
                    'bank_conflict_frac_of_active': round(c('lds', 'SQ_LDS_BANK_CONFLICT') / c('lds', 'SQ_LDS_IDX_ACTIVE'), 3),
                    'vmem_ta_addr_fifo_full_frac_of_cu_cycles': round(c('lds', 'SQ_VMEM_TA_ADDR_FIFO_FULL') / cc, 3),
                    'vmem_ta_cmd_fifo_full_frac_of_cu_cycles': round(c('lds', 'SQ_VMEM_TA_CMD_FIFO_FULL') / cc, 3)}
    if 'l2' in S and 'hbm' in S:
        sec = S['l2']['seconds']
        r['l2'] = {'hit_rate': round(c('l2', 'TCC_HIT_sum') / (c('l2', 'TCC_HIT_sum') + c('l2', 'TCC_MISS_sum')), 3),
                   'requests_per_sec_G': round(c('l2', 'TCC_REQ_sum') / sec / 1e9, 1)}
        rd = c('l2', 'TCC_EA0_RDREQ_sum') * 64 * 2 / d['steps']      # 64 B per request as FETCH_SIZE tallies them, x2 per the guide's gfx950 correction
        wr = c('hbm', 'TCC_EA0_WRREQ_64B_sum') * 64 / d['steps'] + (c('hbm', 'TCC_EA0_WRREQ_sum') - c('hbm', 'TCC_EA0_WRREQ_64B_sum')) * 32 / d['steps']
        us = S['l2']['us_per_step']
        r['hbm'] = {'read_MB_per_step': round(rd / 1e6, 1), 'write_MB_per_step': round(wr / 1e6, 1),
                    'GBps': round((rd + wr) / us / 1e3, 1), 'frac_of_8TBps_peak': round((rd + wr) / us / 1e3 / 8000, 3)}
    if 'spi_b' in S:
        r['workgroup_dispatch_stall_cycles (SPI_RA_*_CSN, summed over 32 SEs)'] = {
            k: S[s]['counters'][k]['sum'] for s in ('spi_a', 'spi_b', 'spi_c', 'spi_d') if s in S for k in S[s]['counters'] if k.startswith('SPI_')}
    if 'ta' in S and 'td' in S:
        cc = cyc('ta') * CUS
        r['texture_path'] = {'ta_busy_frac_of_cu_cycles': round(c('ta', 'TA_TA_BUSY_sum') / cc, 3),
                             'td_busy_frac_of_cu_cycles': round(c('td', 'TD_TD_BUSY_sum') / (cyc('td') * CUS), 3),
                             'lds_dma_wave_instructions_per_image': round(c('ta', 'TA_BUFFER_READ_LDS_WAVEFRONTS_sum') / d['steps'] / d['batch'])}
    res[f'{depth}_batches_in_flight'] = r
json.dump(res, open(out, 'w'), indent=1)
print(json.dumps(res, indent=1))
