#!/bin/bash
# rocprofv3 evidence for a non-headline bench workload: the HBM-traffic PMC passes (separate runs, no trace domains besides
# --kernel-trace), then kernel trace + stats.  usage (on the GPU box): tools/prof_workload.sh WORKLOAD [TAG] [STEPS WARMUP for the PMC passes]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
WL=$1; TAG=${2:-r02}; PS=${3:-101}; PW=${4:-101}
OUT=$R/gpurun_out/prof_${WL}_$TAG
mkdir -p $OUT
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
# the wide family's training step is many launches (bench.py's roofline entry for pines is the whole step): sum the counters over the
# dispatches from fm_prologue_kernel to adamw_vec_kernel, in dispatch order, per PMC pass
seg = collections.defaultdict(float); nseg = 0
for f in glob.glob(out + "/p*/*/*counter_collection.csv"):
    disp = collections.OrderedDict()
    for r in sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"])):
        d = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "c": collections.defaultdict(float)})
        d["c"][r["Counter_Name"]] += float(r["Counter_Value"])
    inside, cur, nst = False, None, 0
    for d in disp.values():
        k = d["name"]
        if k.startswith("void at::") or k.startswith("__amd"): continue
        if "fm_prologue_kernel" in k: inside, cur = True, collections.defaultdict(float)
        if inside:
            for c, v in d["c"].items(): cur[c] += v
        if inside and "adamw_vec_kernel" in k:
            inside = False; nst += 1
            for c, v in cur.items(): seg[c] += v
    for c in set(c for d in disp.values() for c in d["c"]): res.setdefault("_fm_train_step", {}).setdefault(c, {"steps": 0})["steps"] += nst
for c, v in seg.items(): res["_fm_train_step"][c]["mean_per_step"] = v / max(1, res["_fm_train_step"][c]["steps"])
res["_run"] = {"bench_steps": steps, "bench_warmup": warm, "note": "totals cover every launch of the run (bench priming + state preparation + warm-up + timed + instrumented iterations)"}
json.dump(res, open(out + "/pmc_summary.json", "w"), indent=1)
for k, cs in sorted(res.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", {}).get("total", 0) if isinstance(kv[1].get("FETCH_SIZE"), dict) else 0)[:6]:
    if k.startswith("_"): continue
    print(k, {c: "%.4g" % v["mean_per_launch"] for c, v in cs.items()})
PY
# the PMC summary goes into profiles/ BEFORE the traced run, whose JSON line quotes `traffic` from that committed file
case $WL in 4-mode) SHORT=4mode;; gaussian-mixture) SHORT=gmm;; *) SHORT=$WL;; esac
cp $OUT/pmc_summary.json $R/profiles/${TAG}_${SHORT}_pmc_summary.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --workload $WL --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
echo "trace done"; head -8 $OUT/kernel_stats.csv | cut -c1-160
cp $OUT/kernel_stats.csv $R/gpurun_out/${TAG}_${SHORT}_kernel_stats.csv; cp $OUT/bench_under_rocprof.json $R/gpurun_out/${TAG}_${SHORT}_bench_under_rocprof.json; cp $OUT/pmc_summary.json $R/gpurun_out/${TAG}_${SHORT}_pmc_summary.json
