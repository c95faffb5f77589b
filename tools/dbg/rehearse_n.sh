#!/bin/bash
# Development aid: the N > 1 call sequences on a one-rank RCCL communicator (bench.py --steps 101 after a full warm-up cycle), one box.
cd "$(dirname "$0")/../.."
run() { echo "== $*"; env "$@" timeout -k 10 200 python bench.py --steps 101 --warmup 201 --no-cpu-baseline --no-fixed-step 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.4g  iteration_ms_excluding_flow_kernel %.5f  kernels %s' % (d['value'], d['config']['iteration_ms_excluding_flow_kernel'], d.get('kernels_avg_us_instrumented_pass')))"; }
run MFM_X=plain
run MFM_COLLECTIVES_AT_WORLD1=1                                                # default: in-library communicator, one mfm_train_iter, all-reduce in line
run MFM_COLLECTIVES_AT_WORLD1=1 MFM_RCCL_COMM_STREAM=1                         # the all-reduce through the communication stream (event hops)
run MFM_COLLECTIVES_AT_WORLD1=1 MFM_NO_FUSED_AT_N=1                            # split calls, deferred optimizer step, in-library all-reduce on its stream
run MFM_COLLECTIVES_AT_WORLD1=1 MFM_TORCH_ALLREDUCE=1                          # split calls, torch.distributed's all-reduce
run MFM_SPLIT_CALLS=1
