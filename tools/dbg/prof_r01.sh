#!/bin/bash
# rocprofv3 runs for profiles/ (kernel trace + stats; then PMC passes in separate runs)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r01
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 101 --warmup 101 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
ls -R $OUT/trace | head -20
rocprofv3 -L 2>/dev/null | grep -E "SQ_WAVE_CYCLES|SQ_BUSY_CYCLES|SQ_WAIT_ANY|SQ_WAIT_INST_ANY|SQ_ACTIVE_INST_ANY|SQ_VALU_MFMA_BUSY|MFMA_MOPS|SQ_WAIT_INST_LDS|LDS_BANK_CONFLICT|FETCH_SIZE|WRITE_SIZE|SQ_INSTS_VALU_MFMA|SQ_ACTIVE_INST_VALU|SQ_ACTIVE_INST_LDS|SQ_INST_CYCLES_VMEM|SQ_WAIT_INST_VMEM|GRBM_GUI_ACTIVE" | cut -c1-160 | sort -u | head -40
