#!/bin/bash
# instruction-cache counters of the flow-step kernel (its solver loop is ~65 KB of code, the I-cache 64 KB per CU pair)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_icache
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/p -- python3 $R/bench.py --steps 101 --warmup 201 --no-cpu-baseline > $OUT/p.json 2> $OUT/p.err
python3 - $OUT <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out + "/p/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:50]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k, cs in acc.items():
    if "flow_step" in k or "fm_fwd" in k or "wgrad" in k:
        print(k, {c: "%.4g" % (v / n[k][c]) for c, v in cs.items()})
PY
tail -3 $OUT/p.err
