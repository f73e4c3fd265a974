# One GPU-box call: the whole -m gpu suite, then the default bench line.  Output under gpurun_out/$R.
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=${1:-r02}
mkdir -p gpurun_out/$R
python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/$R/gpu_tests.log 2>&1 || { tail -60 gpurun_out/$R/gpu_tests.log; exit 1; }
tail -25 gpurun_out/$R/gpu_tests.log
python bench.py > gpurun_out/$R/bench_default.json 2> gpurun_out/$R/bench_default.err || { tail -30 gpurun_out/$R/bench_default.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/$R/bench_default.json'))
print(d['value'], d['ms_per_step'], d['roofline'], d['accuracy'])
print('f32', d['f32'])
print('cpu', d['cpu_baseline'])
"
