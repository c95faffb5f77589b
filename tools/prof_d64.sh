#!/bin/bash
# rocprofv3 kernel statistics of the flow step at the reference's phi-four default shape (d = 64, 1024 chains, --hutch): the shape-specialised
# kernel with the automatic chains-per-workgroup choice, with 16 per workgroup, and the generic tile.  usage (GPU box): tools/prof_d64.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_d64
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/auto -- python3 $R/tools/d64_time.py one > $OUT/auto.log 2>&1
export MFM_FLOW_LIVE=16
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/live16 -- python3 $R/tools/d64_time.py one > $OUT/live16.log 2>&1
unset MFM_FLOW_LIVE
export MFM_GENERIC_ODE=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/generic -- python3 $R/tools/d64_time.py one > $OUT/generic.log 2>&1
for v in auto live16 generic; do
  f=$(ls $OUT/$v/*/*kernel_stats.csv | head -1)
  echo "== $v"; head -3 $f | cut -c1-200
  cp $f $OUT/${v}_kernel_stats.csv
done
