#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel trace + the two HBM counter passes of one bench step of the f16x2 plan (the
# headline mode), then the bench lines.  Summaries are copied into profiles/ by `python tools/prof_post3.py r03_x2` afterwards.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof3; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python $R/tools/one_step.py 40 f16x2 > $O/stats.log 2>&1; echo stats rc=$?
timeout -k 5 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- python $R/tools/one_step.py 4 f16x2 > $O/fetch.log 2>&1; echo fetch rc=$?
timeout -k 5 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- python $R/tools/one_step.py 4 f16x2 > $O/write.log 2>&1; echo write rc=$?
if [ "$1" = "full" ]; then
  timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o p -- python $R/bench.py --mode train --steps 5 --warmup 2 > $O/train.log 2>&1; echo train rc=$?
fi
cd $R
timeout -k 5 120 python tools/xbench.py > $O/x2_per_launch.txt 2>&1
if [ "$1" = "full" ]; then
  timeout -k 5 500 python bench.py > $O/bench_line.json 2> $O/bench.err; echo bench rc=$?; tail -c 300 $O/bench_line.json
fi
ls $O/stats $O/fetch $O/write
