#!/bin/bash
# SQ counters of the wide GEMM microbenchmark (development aid).  usage on the GPU box: bash tools/mb/pmc_wgemm.sh [binary]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; B=${1:-wgemm}; OUT=$R/gpurun_out/pmc_$B; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/p1 -- $R/tools/mb/$B > $OUT/p1.log 2> $OUT/p1.err
python3 - $OUT <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
acc = collections.OrderedDict()
for f in glob.glob(out + "/p1/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:50], r["Grid_Size"], r.get("Workgroup_Size", ""))
        d = acc.setdefault(k, collections.defaultdict(float)); d[r["Counter_Name"]] += float(r["Counter_Value"]); d["_n_" + r["Counter_Name"]] += 1
for k, d in acc.items():
    n = d["_n_SQ_WAVE_CYCLES"] or 1
    wc = d["SQ_WAVE_CYCLES"] / n
    print(k, "launches %d" % n, "| per launch: wave quad-cycles %.3g, parked %.0f %%, issue-stall %.0f %% (LDS %.0f %%), active %.0f %%, VMEM-active %.1f %%, MFMA busy cycles %.3g, SQ busy %.3g" % (
        wc, 100 * d["SQ_WAIT_ANY"] / n / wc, 100 * d["SQ_WAIT_INST_ANY"] / n / wc, 100 * d["SQ_WAIT_INST_LDS"] / n / wc, 100 * d["SQ_ACTIVE_INST_ANY"] / n / wc,
        100 * d["SQ_ACTIVE_INST_VMEM"] / n / wc, d["SQ_VALU_MFMA_BUSY_CYCLES"] / n, d["SQ_BUSY_CYCLES"] / n))
PY
