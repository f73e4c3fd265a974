# File-ingest path on the GPU box: parity suites (toy + full size), per-kind decode rates, the file-backed pass.
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=${1:-r03b}
mkdir -p gpurun_out/$R && rm -rf gpurun_out/$R/rate
timeout -k 10 900 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_ingest_full.py -m gpu -x -q --durations=8 > gpurun_out/$R/ingest_tests.log 2>&1 || { tail -40 gpurun_out/$R/ingest_tests.log; exit 1; }
tail -14 gpurun_out/$R/ingest_tests.log
timeout -k 10 600 python -m pytest tests -m gpu -q -k "file_backed or cf_encoded" 2>&1 | tail -2
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$R/rate -- python3 tools/inflate_rate.py > gpurun_out/$R/rate.log 2>&1
f=$(find gpurun_out/$R/rate -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' | tee gpurun_out/$R/inflate_rate.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'k_inflate' in r['Kernel_Name']]
place = [r for r in csv.DictReader(open(sys.argv[1])) if 'k_place' in r['Kernel_Name']]
names = ['plane0 stored', 'plane2 literals+short matches', 'plane3 long matches', 'whole level', 'literals 4-bit codes', 'literals 8-bit codes', 'matches of 8 bytes']
for k, r in enumerate(rows):
    if k % 2:
        print(f"{names[k // 2]:32s} 256 streams: k_inflate {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6:8.2f} ms"
              f"   k_place {(int(place[k]['End_Timestamp']) - int(place[k]['Start_Timestamp'])) / 1e6:6.2f} ms")
PY
python tools/filebacked_timing.py 1440 1021 75 ${2:-24} > gpurun_out/$R/filebacked_timing.txt 2>&1 || { tail -20 gpurun_out/$R/filebacked_timing.txt; exit 1; }
cat gpurun_out/$R/filebacked_timing.txt
NF_STAGE_TRACE=1 NF_TIMING_LEGS=device python tools/filebacked_timing.py 1440 1021 75 ${2:-24} 2>&1 | grep -a 'staging:' | tail -18 > gpurun_out/$R/stage_trace.txt || true
cat gpurun_out/$R/stage_trace.txt
