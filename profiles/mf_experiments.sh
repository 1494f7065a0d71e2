# Compile-time timing experiments of k_conv_mf (results are WRONG by construction):  EXPS="0 1 32 33" bash profiles/mf_experiments.sh
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports GRAFT_REPO_ROOT)}"
# the diagnostic build replaces the in-tree library: put the production build back on every exit path
trap 'env -u JAMUN_EXTRA_CFLAGS python3 jamun_amd/csrc/build.py > /dev/null' EXIT
for d in ${EXPS:-0 1 2 4 8 16 32 6 7 63}; do
  export JAMUN_EXTRA_CFLAGS="-DMF_EXP=$d"
  python3 jamun_amd/csrc/build.py > /dev/null
  python3 bench.py --config ${CFG:-cfg2} --no-cpu-baseline --no-secondary --repeats 3 2>/dev/null | python3 -c "
import json,sys
try:
  d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('exp $d', 'conv0 ms', round(d['kernel_avg_ms']['conv0'],4), 'ms/step', round(d['ms_per_step'],3))
except Exception as e: print('exp $d failed', e)"
done
