#!/bin/bash
( timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -q -x -k "lnfold or resid_stats or folded or gemm_epilogues or residual_epilogue" 2>&1 | tail -4 )
bash tools/r3_j.sh
