cd $GRAFT_REPO_ROOT
for a in "--example phi-four --learning_iter 2000 --mcmc_per_flow_steps 10 --hutch" "--example phi-four --learning_iter 1000 --mcmc_per_flow_steps 10 --hutch --hidden_x 128 128 128 --hidden_t 128 128 128 --hidden_xt 128 128 128" "--example 4-mode --learning_iter 1000 --mcmc_per_flow_steps 10 --hidden_x 64 64 64 --hidden_t 64 --hidden_xt 64 64 64"; do
  echo "=== $a"
  s=$(date +%s)
  timeout -k 10 500 python -m mfm_amd.multi_modal $a --seed 1 --log_every 500 > gpurun_out/long_two.log 2>&1
  echo "rc $? in $(( $(date +%s) - s )) s"
  grep -E "Logpdf of flow|Max mean disc|pm|Traceback|Error" gpurun_out/long_two.log | head -6
  grep -E "loss" gpurun_out/long_two.log | tail -2 | cut -c1-200
done
