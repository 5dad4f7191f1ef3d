#!/bin/bash
# Darknet-53 (configs[4]) f16, 32 images: rocprofv3 kernel trace + stats, and one --pmc pass with the matrix-pipe counter (separate runs)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5dark; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python $R/tools/darknet_layers.py f16 32 > $O/stats.log 2>&1; echo stats rc=$?
timeout -k 5 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/pmc -o p -- python $R/tools/darknet_layers.py f16 32 > $O/pmc.log 2>&1; echo pmc rc=$?
cd $R
python - <<'PY'
import csv, glob, collections
O='gpurun_out/r5dark'
f=glob.glob(O+'/pmc/**/*counter_collection.csv', recursive=True)[0]
agg=collections.defaultdict(lambda: collections.Counter()); n=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].split('(')[0][:60]
    agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
    if r['Counter_Name']=='GRBM_GUI_ACTIVE': n[k]+=1
rows=[]
for k,c in agg.items():
    kc=c['GRBM_GUI_ACTIVE']/8
    if kc<=0: continue
    rows.append((kc, k, n[k], c['SQ_VALU_MFMA_BUSY_CYCLES']/(kc*1024), c['SQ_WAIT_ANY']/max(1,c['SQ_WAVE_CYCLES'])))
rows.sort(reverse=True)
tot=sum(r[0] for r in rows)
with open(O+'/darknet_pmc_by_kernel.txt','w') as fo:
    fo.write('# Darknet-53 f16, 32 images: per kernel (all dispatches of the run summed): share of GPU cycles, matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs), waves parked = SQ_WAIT_ANY / SQ_WAVE_CYCLES\n')
    for kc,k,nn,mf,wa in rows[:14]:
        line=f'{kc/tot*100:5.1f} % of cycles  {nn:5d} dispatches  mfma_busy {mf:.3f}  wait_any {wa:.3f}  {k}'
        print(line); fo.write(line+'\n')
    allm=sum(agg[k]['SQ_VALU_MFMA_BUSY_CYCLES'] for k in agg)/(tot*1024)
    line=f'whole run: matrix pipe busy {allm:.3f} of all GPU-active cycles'
    print(line); fo.write(line+'\n')
PY
