# One GPU-box call: a selection of the -m gpu suite (pytest -k expression), output under gpurun_out/$R.
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=${1:-r04}
K=${2:-dateline}
mkdir -p gpurun_out/$R
python -m pytest tests -m gpu -q -k "$K" --durations=10 > gpurun_out/$R/gpu_tests_k.log 2>&1 || { tail -80 gpurun_out/$R/gpu_tests_k.log; exit 1; }
tail -15 gpurun_out/$R/gpu_tests_k.log
