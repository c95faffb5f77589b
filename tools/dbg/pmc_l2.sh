# Development aid: L2 hit / miss requests per kernel (one PMC pass; kernel-trace only, no other trace domain).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_l2
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/pmc_l2 -- python3 $R/bench.py --steps 101 --warmup 101 --no-cpu-baseline > $R/gpurun_out/pmc_l2.json 2> $R/gpurun_out/pmc_l2.err
python3 - $R/gpurun_out/pmc_l2 <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:50]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in acc:
    if any(s in k for s in ("wgrad", "fm_fwd", "reduce_slabs", "adamw")):
        print(k, {c: round(v / n[k][c] / 1e6, 3) for c, v in acc[k].items()}, "M requests per launch")
PY
