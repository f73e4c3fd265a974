# One GPU-box call of a tool: bash scripts/gpu_tool.sh <tag> <output name> <python file> [args...]
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=$1; N=$2; shift 2
mkdir -p gpurun_out/$R
python "$@" > gpurun_out/$R/$N.txt 2>&1 || { tail -40 gpurun_out/$R/$N.txt; exit 1; }
cat gpurun_out/$R/$N.txt
