cd $GRAFT_REPO_ROOT
for lib in "" $GRAFT_REPO_ROOT/mfm_amd/lib/libmfm_hip_maxilp.so; do
  for w in phi-four gaussian-mixture 4-mode pines; do
    if [ -z "$lib" ]; then unset MFM_LIB; else export MFM_LIB=$lib; fi
    python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${lib:+maxilp}', '$w', d['value'], d['ms_per_step'], d['roofline'].get('frac'), d['roofline'].get('avg_launch_ms'), d['config'].get('iteration_ms_excluding_flow_kernel'))"
  done
done
