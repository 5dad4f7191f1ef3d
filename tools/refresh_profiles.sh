#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel stats + the two HBM counter passes of one bench step, then the bench line.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python $R/tools/one_step.py 50 > $O/stats.log 2>&1; echo stats rc=$?
timeout -k 5 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- python $R/tools/one_step.py 3 > $O/fetch.log 2>&1; echo fetch rc=$?
timeout -k 5 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- python $R/tools/one_step.py 3 > $O/write.log 2>&1; echo write rc=$?
cd $R && timeout -k 5 300 python bench.py > $O/bench_line.json 2> $O/bench.err; echo bench rc=$?; tail -c 600 $O/bench_line.json
ls $O/stats $O/fetch $O/write
