# W2 experiment: time steps in pairs (one transect gather per two steps): parity test, then the in-process A/B of whole passes
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=${1:-r04}
mkdir -p gpurun_out/$R
python -m pytest tests -m gpu -q -k "pairs or partial_steps or sharding" > gpurun_out/$R/k3_pairs_tests.log 2>&1 || { tail -60 gpurun_out/$R/k3_pairs_tests.log; exit 1; }
tail -3 gpurun_out/$R/k3_pairs_tests.log
NEMOFLUX_AMD_LIB=$GRAFT_REPO_ROOT/nemoflux_amd/libnemoflux_amd.so python tools/ab_pass.py "0:k3_pairs=0,0:k3_pairs=1" > gpurun_out/$R/ab_pass_k3_pairs.txt 2>&1 || { tail -30 gpurun_out/$R/ab_pass_k3_pairs.txt; exit 1; }
cat gpurun_out/$R/ab_pass_k3_pairs.txt
