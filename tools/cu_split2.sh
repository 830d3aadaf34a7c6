#!/bin/bash
# CU-partition experiment, repaired.  hipExtStreamCreateWithCUMask creates a BLOCKING stream: beside work on the legacy null
# stream every launch synchronises implicitly, which is what round 4's "all-ones mask costs +8.7 ms" control measured.  Here the whole step runs
# on a non-blocking stream of its own (CSMAE_DEBUG=bench_stream), so a masked stream has nothing to block against.  usage: tools/cu_split2.sh OUT [rounds]
out=$1; rounds=${2:-2}
python - <<'PY' | tee -a $out
import ctypes, sys
sys.path.insert(0, "cross-scale-mae_amd")
import torch
from csmae_hip import ops
hip = ctypes.CDLL("libamdhip64.so")
f = ctypes.c_uint(99)
m = ops.cu_masked_stream(0, 256)
hip.hipStreamGetFlags(ctypes.c_void_p(m.cuda_stream), ctypes.byref(f))
g = ctypes.c_uint(99)
hip.hipStreamGetFlags(ctypes.c_void_p(torch.cuda.Stream().cuda_stream), ctypes.byref(g))
print(f"# hipStreamGetFlags: CU-masked stream {f.value} (hipStreamNonBlocking = 1), torch pool stream {g.value}")
PY
run() { timeout 300 env ${1//+/ } python bench.py --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
S="X=1 CSMAE_DEBUG=bench_stream CSMAE_DEBUG=bench_stream,dw_cus=32 CSMAE_DEBUG=bench_stream,dw_cus=24 CSMAE_DEBUG=bench_stream,dw_cus=20 CSMAE_DEBUG=bench_stream,dw_cus=16 CSMAE_DEBUG=bench_stream,dw_cus=12,bwd_main_cus=96:256 CSMAE_DEBUG=bench_stream,dw_cus=16,bwd_main_cus=128:256 CSMAE_DEBUG=bench_stream,dw_cus=20,bwd_main_cus=160:256 CSMAE_DEBUG=bench_stream,bwd_main_cus=0:256"
for i in $(seq $rounds); do for s in $S; do echo "$s $(run $s)" | tee -a $out; done; done
