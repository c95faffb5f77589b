# Development aid: train_time of the reference's pines default (d = 1600, 128 chains, hidden 1024) with and without --hutch
cd $GRAFT_REPO_ROOT
for v in "--hutch" ""; do
  t0=$SECONDS
  timeout -k 10 900 python -m mfm_amd.multi_modal --example pines --learning_iter 110 --mcmc_per_flow_steps 10 $v --seed 1 --log_every 1000 2>&1 | grep -E "pm|Traceback|rror" | head -1 | awk -v l="$v" '{print "pines default " l ": logpdf " $1 "  train_time: " $(NF-2)}'
  echo "wall $((SECONDS - t0)) s"
done
