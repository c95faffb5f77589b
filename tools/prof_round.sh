#!/bin/bash
# rocprofv3 evidence for profiles/: PMC passes of the bench command (separate runs, no trace domains besides --kernel-trace),
# then kernel trace + stats of it.  usage (on the GPU box): tools/prof_round.sh TAG
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r02}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE FETCH_SIZE" \
           "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/bench.py --steps 101 --warmup 201 --no-cpu-baseline > $OUT/p$i.json 2> $OUT/p$i.err
  echo "pmc pass $i done"
done
python3 - $OUT <<'PY'
import csv, sys, glob, json, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out + "/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        if k.startswith("void at::") or k.startswith("__amd"): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
res = {k: {c: {"mean_per_launch": v / n[k][c], "launches": n[k][c]} for c, v in cs.items()} for k, cs in acc.items()}
json.dump(res, open(out + "/pmc_summary.json", "w"), indent=1)
for k, cs in res.items():
    if "flow_step" in k or "fm_fwd" in k: print(k, {c: "%.4g" % v["mean_per_launch"] for c, v in cs.items()})
PY
# the PMC summary goes into profiles/ BEFORE the traced run, whose JSON line quotes `traffic` from that committed file
cp $OUT/pmc_summary.json $R/profiles/${TAG}_pmc_summary.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
echo "trace done"; head -5 $OUT/kernel_stats.csv | cut -c1-150
cp $OUT/kernel_stats.csv $R/gpurun_out/${TAG}_bench_kernel_stats.csv; cp $OUT/bench_under_rocprof.json $R/gpurun_out/${TAG}_bench_under_rocprof.json; cp $OUT/pmc_summary.json $R/gpurun_out/${TAG}_pmc_summary.json
