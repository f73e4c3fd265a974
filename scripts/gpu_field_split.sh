set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=${1:-r04}
mkdir -p gpurun_out/$R
python -m pytest tests -m gpu -q -x -k "field_split or ragged or two_missing or all_steps_in_one_launch or partial_steps" > gpurun_out/$R/field_split_tests.log 2>&1 || { tail -60 gpurun_out/$R/field_split_tests.log; exit 1; }
tail -3 gpurun_out/$R/field_split_tests.log
python tools/size_sweep.py > gpurun_out/$R/size_sweep.txt 2>&1 || { tail -30 gpurun_out/$R/size_sweep.txt; exit 1; }
cat gpurun_out/$R/size_sweep.txt
