#!/bin/bash
# rocprofv3 kernel trace + stats of the pines workload (wide kernel family).  usage (on the GPU box): tools/prof_pines.sh TAG
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r01}
OUT=$R/gpurun_out/prof_pines_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --workload pines --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
head -25 $OUT/kernel_stats.csv | cut -c1-200
