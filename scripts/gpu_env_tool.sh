# bash scripts/gpu_env_tool.sh <tag> <name> "<ENV=VAL ...>" <python file> [args]
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=$1; N=$2; E=$3; shift 3
mkdir -p gpurun_out/$R
env $E python "$@" > gpurun_out/$R/$N.txt 2>&1 || { tail -40 gpurun_out/$R/$N.txt; exit 1; }
cat gpurun_out/$R/$N.txt
