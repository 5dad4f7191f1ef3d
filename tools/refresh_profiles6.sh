#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash tools/refresh_profiles6.sh'), round 6: for BOTH launch schedules of the f16x2 plan - rocprofv3 kernel trace +
# stats, the two HBM counter passes (FETCH_SIZE, WRITE_SIZE: separate runs, no trace domains) and, for the plan behind `value`, three SQ / TCC counter
# passes of one bench step; the device-wide counters of the in-flight regime (tools/inflight_counters.py); the training step's and Darknet-53's kernel
# stats; the bench line.  Summaries are copied into profiles/ here by
#   python tools/prof_post3.py r06_x2 prof6/throughput launch_names_throughput.json ; python tools/prof_post3.py r06_x2lat prof6/latency launch_names_latency.json
#   python tools/step_pmc_post.py r06_x2 prof6/throughput/pmc launch_names_throughput.json
#   python tools/inflight_post.py profiles/r06_inflight_pmc.json gpurun_out/prof6/inflight_d4.json gpurun_out/prof6/inflight_d1.json
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof6; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for s in throughput latency; do
  mkdir -p $O/$s/pmc
  timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$s/stats -o p -- python $R/tools/one_step.py 40 f16x2 $s > $O/$s/stats.log 2>&1; echo $s stats rc=$?
  timeout -k 5 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/$s/fetch -o p -- python $R/tools/one_step.py 4 f16x2 $s > $O/$s/fetch.log 2>&1; echo $s fetch rc=$?
  timeout -k 5 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/$s/write -o p -- python $R/tools/one_step.py 4 f16x2 $s > $O/$s/write.log 2>&1; echo $s write rc=$?
done
s=throughput
timeout -k 5 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/$s/pmc/a -o p -- python $R/tools/one_step.py 3 f16x2 $s > $O/$s/pmc/a.log 2>&1; echo $s pmc a rc=$?
timeout -k 5 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d $O/$s/pmc/b -o p -- python $R/tools/one_step.py 3 f16x2 $s > $O/$s/pmc/b.log 2>&1; echo $s pmc b rc=$?
timeout -k 5 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES TCC_HIT_sum TCC_REQ_sum --output-format csv -d $O/$s/pmc/c -o p -- python $R/tools/one_step.py 3 f16x2 $s > $O/$s/pmc/c.log 2>&1; echo $s pmc c rc=$?
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o p -- python $R/bench.py --mode train --steps 5 --warmup 2 > $O/train.log 2>&1; echo train rc=$?
YK_NET=yolo timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/darknet -o p -- python $R/tools/inflight.py 12 1 1 f16 > $O/darknet.log 2>&1; echo darknet rc=$?; tail -1 $O/darknet.log
cd $R
( ROCP_TOOL_LIBRARIES=$R/tools/devcount/libdevcount.so timeout 400 python tools/inflight_counters.py 4 400 $O/inflight_d4.json ) > $O/inflight_d4.log 2>&1; echo inflight d4 rc=$?
( ROCP_TOOL_LIBRARIES=$R/tools/devcount/libdevcount.so timeout 400 python tools/inflight_counters.py 1 400 $O/inflight_d1.json ) > $O/inflight_d1.log 2>&1; echo inflight d1 rc=$?
( timeout 900 python bench.py ) > $O/bench_line.json 2> $O/bench_line.err; echo bench rc=$?; cut -c1-400 $O/bench_line.json
( timeout 600 python tools/netbench.py tiny_yolo yolo_mobilev2 yolo ) > $O/other_configs.txt 2>&1; tail -8 $O/other_configs.txt
cp gpurun_out/launch_names_throughput.json gpurun_out/launch_names_latency.json $O/ 2>/dev/null
find $O -name "*.db" -delete 2>/dev/null
du -sh $O | tail -1
