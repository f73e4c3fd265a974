# Where does a group's staging time go?  The pipelined file-backed leg with the stager's trace, with and without the early upload.
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=${1:-r03g}
mkdir -p gpurun_out/$R
for early in 1 0; do
  for bg in 8 4; do
    echo "=== NF_EARLY_UPLOAD=$early NF_GATHER_THREADS_BG=$bg" >> gpurun_out/$R/stage_trace.txt
    NF_EARLY_UPLOAD=$early NF_GATHER_THREADS_BG=$bg NF_STAGE_TRACE=1 NF_TIMING_LEGS=pipelined python tools/filebacked_timing.py 1440 1021 75 ${2:-24} 2>&1 | grep -a 'staging:\|pipelined' | tail -17 >> gpurun_out/$R/stage_trace.txt
  done
done
cat gpurun_out/$R/stage_trace.txt
