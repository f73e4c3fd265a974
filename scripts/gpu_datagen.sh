set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=${1:-r04}
mkdir -p gpurun_out/$R
python -m pytest tests -m gpu -q -k "datagen" > gpurun_out/$R/datagen_tests.log 2>&1 || { tail -60 gpurun_out/$R/datagen_tests.log; exit 1; }
tail -3 gpurun_out/$R/datagen_tests.log
python tools/datagen_timing.py 12 > gpurun_out/$R/datagen_timing.txt 2>&1 || { tail -30 gpurun_out/$R/datagen_timing.txt; exit 1; }
cat gpurun_out/$R/datagen_timing.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/prof_dg -- python3 tools/datagen_timing.py 12 > gpurun_out/$R/prof_dg.log 2>&1
python3 - <<PY
import glob, csv
for f in glob.glob('gpurun_out/$R/prof_dg/**/*kernel_stats.csv', recursive=True):
    for row in list(csv.reader(open(f)))[:8]:
        print(row[:7])
PY
