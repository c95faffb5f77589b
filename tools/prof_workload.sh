#!/bin/bash
# rocprofv3 evidence for a non-headline bench workload: kernel trace + stats, then the HBM-traffic PMC passes (separate runs, no
# trace domains besides --kernel-trace).  usage (on the GPU box): tools/prof_workload.sh WORKLOAD [TAG] [STEPS WARMUP for the PMC passes]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
WL=$1; TAG=${2:-r02}; PS=${3:-101}; PW=${4:-101}
OUT=$R/gpurun_out/prof_${WL}_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --workload $WL --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
echo "trace done"; head -8 $OUT/kernel_stats.csv | cut -c1-160
i=0
for set in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/bench.py --workload $WL --steps $PS --warmup $PW --no-cpu-baseline > $OUT/p$i.json 2> $OUT/p$i.err
  echo "pmc pass $i done"
done
python3 - $OUT $PS $PW <<'PY'
import csv, sys, glob, json, collections
out, steps, warm = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out + "/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:70]
        if k.startswith("void at::") or k.startswith("__amd"): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
res = {k: {c: {"mean_per_launch": v / n[k][c], "launches": n[k][c], "total": v} for c, v in cs.items()} for k, cs in acc.items()}
tot = collections.defaultdict(float)
for k, cs in acc.items():
    for c, v in cs.items(): tot[c] += v
res["_all_kernels_total"] = {c: v for c, v in tot.items()}
res["_run"] = {"bench_steps": steps, "bench_warmup": warm, "note": "totals cover every launch of the run (bench priming + state preparation + warm-up + timed + instrumented iterations)"}
json.dump(res, open(out + "/pmc_summary.json", "w"), indent=1)
for k, cs in sorted(res.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", {}).get("total", 0) if isinstance(kv[1].get("FETCH_SIZE"), dict) else 0)[:6]:
    if k.startswith("_"): continue
    print(k, {c: "%.4g" % v["mean_per_launch"] for c, v in cs.items()})
PY
