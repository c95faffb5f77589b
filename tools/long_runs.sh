# Full-length runs of the reference README's command lines (development aid): sanity of the final metrics after 1,000 / 10,000 iterations.
cd $GRAFT_REPO_ROOT
for a in "--example 4-mode --learning_iter 1000 --mcmc_per_flow_steps 10" "--example 4-mode --learning_iter 1000 --mcmc_per_flow_steps 100 --hutch" "--example gaussian-mixture --learning_iter 10000 --mcmc_per_flow_steps 100" "--example phi-four --learning_iter 1000 --mcmc_per_flow_steps 10 --hutch"; do
  echo "=== $a"
  s=$(date +%s)
  timeout -k 10 900 python -m mfm_amd.multi_modal $a --seed 1 --log_every 1000 > gpurun_out/long_one.log 2>&1
  echo "rc $? in $(( $(date +%s) - s )) s"
  tail -25 gpurun_out/long_one.log
done
