set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
make -C nemoflux_amd/csrc tuning -j8 -s   # the diagnostic library is built here, on the GPU box: build/ never travels
R=${1:-r02}
mkdir -p gpurun_out/$R
AB_NOCHECK=1 python tools/ab_pass.py "$2" > gpurun_out/$R/ab_pass_$3.txt 2>&1
cat gpurun_out/$R/ab_pass_$3.txt
