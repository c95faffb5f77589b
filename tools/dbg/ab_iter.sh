#!/bin/bash
# Development aid: tools/iter_time.py alternating over library variants on ONE box (box-to-box differences are +-1 us): tools/dbg/ab_iter.sh main p10 p0
cd "$(dirname "$0")/../.."
for rep in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = main ]; then L=$PWD/mfm_amd/lib/libmfm_hip.so; else L=$PWD/mfm_amd/lib/libmfm_hip_$v.so; fi
    echo "== $v: $(MFM_LIB=$L timeout -k 10 120 python tools/iter_time.py 2>&1 | grep -E 'wall|kernels' | tr '\n' ' ')"
  done
done
