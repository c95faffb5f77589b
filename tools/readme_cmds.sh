set -e
cd $GRAFT_REPO_ROOT
for a in "--example 4-mode --learning_iter 44 --mcmc_per_flow_steps -1" "--example 4-mode --learning_iter 44 --mcmc_per_flow_steps 1000" "--example 4-mode --learning_iter 44 --mcmc_per_flow_steps 1 --hutch" "--example 4-mode --learning_iter 44 --mcmc_per_flow_steps 10" "--example gaussian-mixture --learning_iter 33 --mcmc_per_flow_steps 10 --hutch" "--example gaussian-mixture --learning_iter 33 --mcmc_per_flow_steps 10 --check" "--example phi-four --learning_iter 33 --mcmc_per_flow_steps 10" "--example phi-four --learning_iter 33 --mcmc_per_flow_steps 10 --hutch" "--example pines --learning_iter 12 --mcmc_per_flow_steps 5 --hutch" "--example pines --learning_iter 12 --mcmc_per_flow_steps 5" "--example pines --learning_iter 7 --mcmc_per_flow_steps 100" "--example 4-mode --learning_iter 22 --mcmc_per_flow_steps 10 --non_linearity gelu" "--example 4-mode --do_smc --learning_iter 10" "--example phi-four --learning_iter 22 --mcmc_per_flow_steps 10 --hutch --num_chain 100" "--example 4-mode --learning_iter 22 --mcmc_per_flow_steps 10 --num_chain 100" "--example phi-four --learning_iter 22 --mcmc_per_flow_steps 10 --hutch --hidden_x 128 128 128 --hidden_t 128 --hidden_xt 128 128 128" "--example 4-mode --learning_iter 22 --mcmc_per_flow_steps 10 --hidden_x 64 64 64 --hidden_t 64 64 64 --hidden_xt 64"; do
  echo "=== $a"
  t0=$SECONDS
  timeout -k 10 400 python -m mfm_amd.multi_modal $a --seed 1 2>&1 | grep -E "Logpdf of flow|Max mean disc of flow|Error|error|Traceback|pm" | head -6
  echo "wall $((SECONDS - t0)) s"
done
