# Development aid: the command lines with hidden lists of other lengths / widths (README.md), with wall times
cd $GRAFT_REPO_ROOT
for a in "--example phi-four --learning_iter 22 --mcmc_per_flow_steps 10 --hutch --hidden_x 128 128 128 --hidden_t 128 --hidden_xt 128 128 128" "--example 4-mode --learning_iter 22 --mcmc_per_flow_steps 10 --hidden_x 64 64 64 --hidden_t 64 64 64 --hidden_xt 64" "--example phi-four --learning_iter 22 --mcmc_per_flow_steps 10 --hutch --hidden_x 100 100 --hidden_t 100 100 --hidden_xt 100 100" "--example phi-four --learning_iter 22 --mcmc_per_flow_steps 10 --hutch"; do
  echo "=== $a"
  t0=$SECONDS
  timeout -k 10 400 python -m mfm_amd.multi_modal $a --seed 1 2>&1 | grep -E "Logpdf of flow|Max mean disc of flow|Error|error|Traceback|pm" | head -8
  echo "wall $((SECONDS - t0)) s"
done
