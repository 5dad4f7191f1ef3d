#!/bin/bash
# counters of ONE implicit-GEMM launch (Darknet 52x52 128->256 3x3, B=32, f16), two tile configurations, two --pmc passes each
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5pmc; rm -rf $O; mkdir -p $O
export YK_LIB_PATH=$R/k210_yolo_framework_amd/csrc/libyolo_hip_dev.so YK_FORCE_MINK=64 ITERS=3
cd /tmp && export TMPDIR=/tmp
for v in "9 2" "11 3" "14 2"; do
  set -- $v; tag=c$1n$2
  export YK_IGEMM_FORCE=$1 YK_NS=$2 YK_SPLIT_FORCE=1
  timeout -k 5 100 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/$tag/a -o p -- python $R/tools/igemm_one.py 52 52 128 256 32 > $O/$tag.a.log 2>&1
  timeout -k 5 100 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES TCC_HIT_sum TCC_REQ_sum --output-format csv -d $O/$tag/b -o p -- python $R/tools/igemm_one.py 52 52 128 256 32 > $O/$tag.b.log 2>&1
  timeout -k 5 100 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INST_CYCLES_VMEM_RD TCC_EA0_RDREQ_sum --output-format csv -d $O/$tag/c -o p -- python $R/tools/igemm_one.py 52 52 128 256 32 > $O/$tag.c.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections, os
O='gpurun_out/r5pmc'
for tag in sorted(d for d in os.listdir(O) if os.path.isdir(os.path.join(O,d))):
    vals=collections.OrderedDict()
    for sub in 'abc':
        fs=glob.glob(f'{O}/{tag}/{sub}/**/*counter_collection.csv', recursive=True)
        if not fs: continue
        rows=[r for r in csv.DictReader(open(fs[0])) if 'igemm_pipe_kernel' in r['Kernel_Name']]
        if not rows: continue
        last=max(int(r['Dispatch_Id']) for r in rows)
        for r in rows:
            if int(r['Dispatch_Id'])==last:
                vals[r['Counter_Name']]=vals.get(r['Counter_Name'],0.0)+float(r['Counter_Value'])
                vals['_grid']=r['Grid_Size']; vals['_wg']=r['Workgroup_Size']; vals['_lds']=r.get('LDS_Block_Size'); vals['_vgpr']=r.get('VGPR_Count')
    wc=vals.get('SQ_WAVE_CYCLES',1); kc=vals.get('GRBM_GUI_ACTIVE',0)/8
    print(tag, {k:(round(v) if isinstance(v,float) else v) for k,v in vals.items()})
    if kc:
        print('   kernel cycles', round(kc), ' occupancy waves/SIMD', round(wc/(kc*1024),2), ' mfma_busy', round(vals.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(kc*1024),3),
              ' wait_any', round(vals.get('SQ_WAIT_ANY',0)/wc,3), ' wait_inst_any', round(vals.get('SQ_WAIT_INST_ANY',0)/wc,3), ' active_any', round(vals.get('SQ_ACTIVE_INST_ANY',0)/wc,3),
              ' wait_inst_lds', round(vals.get('SQ_WAIT_INST_LDS',0)/wc,3), ' lds_idx_active/kc/256', round(vals.get('SQ_LDS_IDX_ACTIVE',0)/(kc*256),3),
              ' lds_conflict', round(vals.get('SQ_LDS_BANK_CONFLICT',0)/max(1,vals.get('SQ_LDS_IDX_ACTIVE',1)),3), ' l2_hit', round(vals.get('TCC_HIT_sum',0)/max(1,vals.get('TCC_REQ_sum',1)),3),
              ' active_vmem', round(vals.get('SQ_ACTIVE_INST_VMEM',0)/wc,3), ' active_lds', round(vals.get('SQ_ACTIVE_INST_LDS',0)/wc,3))
PY
