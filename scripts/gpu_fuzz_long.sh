set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$1
true
true
python tools/fuzz_weights.py 30000 12 > gpurun_out/$1/fuzz_weights_12.txt 2>&1 || { tail -20 gpurun_out/$1/fuzz_weights_12.txt; exit 1; }
tail -2 gpurun_out/$1/fuzz_weights_12.txt
python tools/fuzz_flux.py 4000 13 > gpurun_out/$1/fuzz_flux_13.txt 2>&1 || { tail -20 gpurun_out/$1/fuzz_flux_13.txt; exit 1; }
tail -1 gpurun_out/$1/fuzz_flux_13.txt
