#!/bin/bash
# SQ counters of the weight-gradient kernels (one-workgroup and two-workgroups-per-CU) and of the NT / NN kernels for comparison.  usage: tools/pmc_dw.sh TAG
tag=$1; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
cat > /tmp/pmc_dw_in.txt <<'EOT'
pmc: SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pmc: SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM
EOT
( cd /tmp && timeout 600 rocprofv3 -i /tmp/pmc_dw_in.txt --kernel-trace -d /tmp/sqdw_$tag -o r -- python $GRAFT_REPO_ROOT/tools/dw_check.py --quick > /dev/null 2> $GRAFT_REPO_ROOT/$out/sq.err )
python tools/pmc_show.py $(find /tmp/sqdw_$tag -name '*.db') > $out/pmc_sq_dw.txt
( cd /tmp && timeout 600 rocprofv3 -i /tmp/pmc_dw_in.txt --kernel-trace -d /tmp/sqg_$tag -o r -- python $GRAFT_REPO_ROOT/tools/gemm_bench.py --iters 1 --only dec > /dev/null 2>> $GRAFT_REPO_ROOT/$out/sq.err )
python tools/pmc_show.py $(find /tmp/sqg_$tag -name '*.db') > $out/pmc_sq_gemm.txt
cat $out/pmc_sq_dw.txt $out/pmc_sq_gemm.txt
