#!/bin/bash
# PMC passes (separate runs) for the MFMA kernels; usage: prof_pmc.sh OUTDIR "set1" "set2" ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
i=0
for set in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/bench.py --steps 101 --warmup 0 --no-cpu-baseline > $OUT/p$i.json 2> $OUT/p$i.err
  f=$(ls $OUT/p$i/*/*counter_collection.csv | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:40]
    if not any(s in k for s in ("flow_step", "fm_fwd_bwd")): continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in acc:
    print(k, {c: "%.4g" % (v / n[k][c]) for c, v in acc[k].items()})
PY
done
