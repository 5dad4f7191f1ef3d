#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c9; mkdir -p $O
( timeout 900 python tools/r05_map_synth.py --steps 2500 --out gpurun_out/r5c9/map_eval.json ) > $O/map.log 2>&1; tail -12 $O/map.log | cut -c1-400
