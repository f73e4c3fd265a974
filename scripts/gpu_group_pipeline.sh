# The pipelined file-backed pass (tools/filebacked_timing.py, C3-sized float32 image, 24 steps) against the group size and the
# early-upload form: G = 13 (two waves of decoder streams, the default for long series) leaves a 24-step series only two
# groups -- nothing to overlap the first with.
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=${1:-r06}
mkdir -p gpurun_out/$R
: > gpurun_out/$R/group_pipeline.txt
for cfg in "13 1" "6 1" "6 0" "8 1" "4 1" "13 0"; do
  set -- $cfg
  echo "=== NF_INFLATE_GROUP=$1 NF_DIRECT_UPLOAD=$2" >> gpurun_out/$R/group_pipeline.txt
  NF_INFLATE_GROUP=$1 NF_DIRECT_UPLOAD=$2 NF_STAGE_TRACE=1 NF_TIMING_LEGS=pipelined python tools/filebacked_timing.py 1440 1021 75 24 2>&1 | grep -a 'staging:\|device inflate' | tail -9 >> gpurun_out/$R/group_pipeline.txt
done
cat gpurun_out/$R/group_pipeline.txt
