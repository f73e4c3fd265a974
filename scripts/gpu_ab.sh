# In-process A/B of the flux-kernel variants (tuning build of the library), float32 and float64.
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
make -C nemoflux_amd/csrc tuning -j8 -s   # the diagnostic library is built here, on the GPU box: build/ never travels
R=${1:-r02}
mkdir -p gpurun_out/$R
python tools/ab_flux.py --dtype float32 --variants ${2:-0,5,3,14,4,11,21,28} > gpurun_out/$R/ab_flux_f32.txt 2>&1
cat gpurun_out/$R/ab_flux_f32.txt
python tools/ab_flux.py --dtype float64 --variants ${3:-0,5,6,13,3,12,40,21,28} > gpurun_out/$R/ab_flux_f64.txt 2>&1
cat gpurun_out/$R/ab_flux_f64.txt
