# Development aid: bench.py with the product library and with variant libraries, alternating, on the same box.
# usage: tools/dbg/ab_bench.sh name1 [name2 ...]   (mfm_amd/lib/libmfm_hip_<name>.so; "product" = the product library)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = product ]; then unset MFM_LIB; else export MFM_LIB=$PWD/mfm_amd/lib/libmfm_hip_$v.so; fi
  python bench.py --no-cpu-baseline > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err
  python - $v <<PY
import json, sys
j = json.loads(open(f"gpurun_out/ab_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], j["value"], j["config"]["iteration_ms_excluding_flow_kernel"], j["config"]["flow_step_avg_ms"], j["kernels_avg_us_instrumented_pass"])
PY
done; done
