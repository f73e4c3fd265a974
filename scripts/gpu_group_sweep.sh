# How many time steps per decode launch?  The pipelined file-backed leg with groups of 6 (one wave of streams on the chip's
# 1024 decoder slots), 12 / 13 (two waves) and 20 (three) on a 26-step C3-sized file image.
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=${1:-r03r}
mkdir -p gpurun_out/$R
for g in 6 13 20; do
  echo "=== NF_INFLATE_GROUP=$g" >> gpurun_out/$R/group_sweep.txt
  NF_INFLATE_GROUP=$g NF_STAGE_TRACE=1 NF_TIMING_LEGS=pipelined python tools/filebacked_timing.py 1440 1021 75 ${2:-26} 2>&1 | grep -a 'staging:\|pipelined' | tail -9 >> gpurun_out/$R/group_sweep.txt
done
cat gpurun_out/$R/group_sweep.txt
