set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=r06
python tools/filebacked_timing.py 1440 1021 75 24 > gpurun_out/$R/filebacked_timing.txt 2>&1 || { tail -20 gpurun_out/$R/filebacked_timing.txt; exit 1; }
cat gpurun_out/$R/filebacked_timing.txt
NF_STAGE_TRACE=1 NF_TIMING_LEGS=device python tools/filebacked_timing.py 1440 1021 75 24 2>&1 | grep -a 'staging:' | tail -18 > gpurun_out/$R/stage_trace.txt || true
cat gpurun_out/$R/stage_trace.txt
python bench.py --no-cpu --no-f32 --no-c3 --steps 5 > gpurun_out/$R/bench_ingest.json 2> gpurun_out/$R/bench_ingest.err || { tail -30 gpurun_out/$R/bench_ingest.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/$R/bench_ingest.json')); print(json.dumps(d['ingest'], indent=1))"
