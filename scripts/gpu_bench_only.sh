set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$1
python bench.py > gpurun_out/$1/bench_default.json 2> gpurun_out/$1/bench_default.err || { tail -30 gpurun_out/$1/bench_default.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/$1/bench_default.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])
print('f32', d['f32']['value'], d['f32']['roofline']['frac'])
print('c3', json.dumps(d['c3'], indent=1))
print('ingest', d['ingest'])
"
