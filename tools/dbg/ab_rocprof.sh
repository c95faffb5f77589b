# Development aid: rocprofv3 kernel stats of bench.py for the product library and variant libraries on the same box.
# usage: tools/dbg/ab_rocprof.sh name1 [name2 ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = product ]; then unset MFM_LIB; else export MFM_LIB=$R/mfm_amd/lib/libmfm_hip_$v.so; fi
  rm -rf $R/gpurun_out/abprof_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abprof_$v -- python3 $R/bench.py --steps 303 --warmup 101 --no-cpu-baseline > $R/gpurun_out/abprof_$v.json 2> $R/gpurun_out/abprof_$v.err
  echo "== $v"; python3 - $R/gpurun_out/abprof_$v <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:7]:
    print(f"{r['Name'][:64]:64s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.2f} us")
PY
done
