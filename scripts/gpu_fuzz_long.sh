# Long differential fuzz campaigns on one MI355X (about 2.5 + 2.5 + 9 minutes): K2 / point location and K1 against the oracle.
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$1
for seed in ${2:-11 12}; do
  python tools/fuzz_weights.py 30000 $seed > gpurun_out/$1/fuzz_weights_$seed.txt 2>&1 || { tail -20 gpurun_out/$1/fuzz_weights_$seed.txt; exit 1; }
  tail -2 gpurun_out/$1/fuzz_weights_$seed.txt
done
python tools/fuzz_flux.py 4000 13 > gpurun_out/$1/fuzz_flux_13.txt 2>&1 || { tail -20 gpurun_out/$1/fuzz_flux_13.txt; exit 1; }
tail -1 gpurun_out/$1/fuzz_flux_13.txt
