#!/bin/bash
# Run ON THE GPU BOX (through gpurun), round 5: for BOTH launch schedules of the f16x2 plan - rocprofv3 kernel trace + stats, the two HBM
# counter passes (FETCH_SIZE, WRITE_SIZE: separate runs, no trace domains) and three SQ / TCC counter passes of one bench step; the
# training step's kernel stats; the bench line.  Summaries are copied into profiles/ by
#   python tools/prof_post3.py r05_x2 prof5/throughput launch_names_throughput.json ; python tools/prof_post3.py r05_x2lat prof5/latency launch_names_latency.json
#   python tools/step_pmc_post.py r05_x2 prof5/throughput/pmc launch_names_throughput.json ; python tools/step_pmc_post.py r05_x2lat prof5/latency/pmc launch_names_latency.json
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof5; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for s in throughput latency; do
  mkdir -p $O/$s/pmc
  timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$s/stats -o p -- python $R/tools/one_step.py 40 f16x2 $s > $O/$s/stats.log 2>&1; echo $s stats rc=$?
  timeout -k 5 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/$s/fetch -o p -- python $R/tools/one_step.py 4 f16x2 $s > $O/$s/fetch.log 2>&1; echo $s fetch rc=$?
  timeout -k 5 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/$s/write -o p -- python $R/tools/one_step.py 4 f16x2 $s > $O/$s/write.log 2>&1; echo $s write rc=$?
  timeout -k 5 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/$s/pmc/a -o p -- python $R/tools/one_step.py 3 f16x2 $s > $O/$s/pmc/a.log 2>&1; echo $s pmc a rc=$?
  timeout -k 5 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d $O/$s/pmc/b -o p -- python $R/tools/one_step.py 3 f16x2 $s > $O/$s/pmc/b.log 2>&1; echo $s pmc b rc=$?
  timeout -k 5 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES TCC_HIT_sum TCC_REQ_sum --output-format csv -d $O/$s/pmc/c -o p -- python $R/tools/one_step.py 3 f16x2 $s > $O/$s/pmc/c.log 2>&1; echo $s pmc c rc=$?
done
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o p -- python $R/bench.py --mode train --steps 5 --warmup 2 > $O/train.log 2>&1; echo train rc=$?
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o p -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-secondary > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err; echo bench-rocprof rc=$?
cd $R
cp gpurun_out/launch_names_throughput.json gpurun_out/launch_names_latency.json $O/ 2>/dev/null
# the large raw traces are not needed back: keep the CSVs the post-processing reads
find $O -name "*.db" -delete 2>/dev/null
du -sh $O | tail -1
