#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel stats + the two HBM counter passes of one bench step, then the bench line,
# the per-layer fp16 drift table and the other networks' throughput.  Summaries are copied into profiles/ by tools/profiles_post.py.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python $R/tools/one_step.py 50 > $O/stats.log 2>&1; echo stats rc=$?
timeout -k 5 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- python $R/tools/one_step.py 3 > $O/fetch.log 2>&1; echo fetch rc=$?
timeout -k 5 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- python $R/tools/one_step.py 3 > $O/write.log 2>&1; echo write rc=$?
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o p -- python $R/bench.py --mode train --steps 5 --warmup 2 > $O/train.log 2>&1; echo train rc=$?
timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/x2 -o p -- python $R/tools/xone.py 20 > $O/x2.log 2>&1; echo x2 rc=$?
cd $R
timeout -k 5 120 python tools/xbench.py > $O/x2_per_launch.txt 2>&1
timeout -k 5 120 python tests/diag_drift.py yolo_mobilev1 0.75 > $O/drift_v1.txt 2>&1; echo drift rc=$?
timeout -k 5 200 python tools/pipebench.py tiny_yolo 1.0 416 416 64 > $O/pipe_tiny.txt 2>&1
timeout -k 5 200 python tools/pipebench.py yolo_mobilev2 1.0 224 320 32 > $O/pipe_v2.txt 2>&1
timeout -k 5 200 python tools/pipebench.py yolo 1.0 416 416 16 > $O/pipe_yolo.txt 2>&1
timeout -k 5 400 python bench.py > $O/bench_line.json 2> $O/bench.err; echo bench rc=$?; tail -c 300 $O/bench_line.json
ls $O/stats $O/fetch $O/write $O/train
