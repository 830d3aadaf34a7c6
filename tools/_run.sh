( timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "gemm" 2>&1 | tail -4 )
python tools/epi_abl.py e0; CSMAE_EPI_POINTERS=1 python tools/epi_abl.py e0
bash tools/ab_env.sh CSMAE_EPI_POINTERS=1 3
