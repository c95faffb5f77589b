# Development aid: the reference's first phi-four command line (exact trace: no --hutch), solves on the wide family's solver (default) and
# on the fused family's own generic tile (MFM_TILE_EXACT=1): train_time of the result row
cd $GRAFT_REPO_ROOT
for v in wide tile; do
  if [ $v = tile ]; then export MFM_TILE_EXACT=1; else unset MFM_TILE_EXACT; fi
  timeout -k 10 900 python -m mfm_amd.multi_modal --example phi-four --learning_iter 400 --mcmc_per_flow_steps 10 --seed 1 --log_every 400 2>&1 | grep -E "pm" | head -1 | awk -v l=$v '{print "exact-trace solves on " l ": logpdf " $1 "  train_time: " $(NF-2)}'
done
