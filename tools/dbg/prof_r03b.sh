# Round-3 refresh of the kernel-trace summaries after the MALA step moved into the training kernel (trace + stats only; the PMC
# passes of tools/prof_round.sh / prof_workload.sh concern kernels that did not change).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for wl in gaussian-mixture 4-mode; do
  OUT=$R/gpurun_out/prof_r03b_$wl; rm -rf $OUT; mkdir -p $OUT
  if [ $wl = headline ]; then W=""; else W="--workload $wl"; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $W --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
  cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
  echo "== $wl"; head -7 $OUT/kernel_stats.csv | cut -c1-140; tail -c 600 $OUT/bench_under_rocprof.json | head -c 300; echo
done
