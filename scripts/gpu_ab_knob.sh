# in-process A/B of whole passes between two settings of a tuning knob of the shipped library: bash scripts/gpu_ab_knob.sh <tag> "<spec>" <name>
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$1
NEMOFLUX_AMD_LIB=$GRAFT_REPO_ROOT/nemoflux_amd/libnemoflux_amd.so python tools/ab_pass.py "$2" > gpurun_out/$1/ab_pass_$3.txt 2>&1 || { tail -30 gpurun_out/$1/ab_pass_$3.txt; exit 1; }
cat gpurun_out/$1/ab_pass_$3.txt
