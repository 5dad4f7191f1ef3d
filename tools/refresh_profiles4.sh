#!/bin/bash
# Run ON THE GPU BOX (through gpurun): for BOTH launch schedules of the f16x2 plan (throughput = the plan behind bench.py's `value`,
# latency = the plan of its one-batch number) the rocprofv3 kernel trace + the two HBM counter passes of one bench step; then the phase
# timelines of the two cluster launches, the training step's kernel stats and the bench line.  Summaries are copied into profiles/ by
#   python tools/prof_post3.py r04_x2 prof4/throughput launch_names_throughput.json
#   python tools/prof_post3.py r04_x2lat prof4/latency launch_names_latency.json
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof4; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for s in throughput latency; do
  mkdir -p $O/$s
  timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$s/stats -o p -- python $R/tools/one_step.py 40 f16x2 $s > $O/$s/stats.log 2>&1; echo $s stats rc=$?
  timeout -k 5 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/$s/fetch -o p -- python $R/tools/one_step.py 4 f16x2 $s > $O/$s/fetch.log 2>&1; echo $s fetch rc=$?
  timeout -k 5 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/$s/write -o p -- python $R/tools/one_step.py 4 f16x2 $s > $O/$s/write.log 2>&1; echo $s write rc=$?
done
if [ "$1" = "full" ]; then
  timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o p -- python $R/bench.py --mode train --steps 5 --warmup 2 > $O/train.log 2>&1; echo train rc=$?
  timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o p -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-secondary > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err; echo bench-rocprof rc=$?
fi
cd $R
timeout -k 5 120 python tools/xbench.py > $O/x2_per_launch.txt 2>&1
YK_LIB_PATH=$R/k210_yolo_framework_amd/csrc/libyolo_hip_dev.so timeout -k 5 120 python tools/xpersist_phase.py > $O/persist_phases.txt 2>&1
YK_LIB_PATH=$R/k210_yolo_framework_amd/csrc/libyolo_hip_dev.so timeout -k 5 120 python tools/xheads_phase.py > $O/heads_phases.txt 2>&1
if [ "$1" = "full" ]; then
  timeout -k 5 700 python bench.py > $O/bench_line.json 2> $O/bench.err; echo bench rc=$?; tail -c 300 $O/bench_line.json
fi
ls $O/throughput/stats $O/latency/stats | head
