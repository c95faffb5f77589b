cd $GRAFT_REPO_ROOT
for l in auto 16 auto 16; do
  if [ $l = auto ]; then unset MFM_FLOW_LIVE; else export MFM_FLOW_LIVE=$l; fi
  timeout -k 10 300 python -m mfm_amd.multi_modal --example phi-four --learning_iter 4000 --mcmc_per_flow_steps 10 --hutch --seed 1 --log_every 4000 2>&1 | grep -E "pm" | head -1 | awk -v l=$l '{print "chains/workgroup " l ": logpdf " $1 "  train_time column: " $(NF-2)}'
done
