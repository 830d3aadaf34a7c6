#!/bin/bash
# A/B the step time of two builds of the HIP library on the same GPU box, interleaved: tools/ab.sh build/lib_base.so [rounds]
base=$1; rounds=${2:-3}
lib=cross-scale-mae_amd/csmae_hip/libcsmae_hip.so
cp $lib /tmp/lib_new.so
run() { timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in $(seq $rounds); do
  cp $base $lib; echo "base $(run)"
  cp /tmp/lib_new.so $lib; echo "new  $(run)"
done
