#!/bin/bash
# Run ON THE GPU BOX (through gpurun): three --pmc passes (no trace domains) over one bench step of the f16x2 plan; summarised per launch
# by `python tools/step_pmc_post.py r03_x2` into profiles/<tag>_step_pmc.json.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc3; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 5 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/a -o p -- python $R/tools/one_step.py 3 f16x2 > $O/a.log 2>&1; echo a rc=$?
timeout -k 5 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d $O/b -o p -- python $R/tools/one_step.py 3 f16x2 > $O/b.log 2>&1; echo b rc=$?
timeout -k 5 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES TCC_HIT_sum TCC_REQ_sum --output-format csv -d $O/c -o p -- python $R/tools/one_step.py 3 f16x2 > $O/c.log 2>&1; echo c rc=$?
ls $O/a $O/b $O/c 2>&1 | head -12; tail -n 3 $O/b.log; tail -n 3 $O/c.log
