# correctness of the device decoder, then nf::k_inflate durations by kind of data (tools/inflate_rate.py under rocprofv3)
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=${1:-r02j}
mkdir -p gpurun_out/$R && rm -rf gpurun_out/$R/rate
timeout -k 10 600 python -m pytest tests -m gpu -q -k "inflate or unshuffle or tiled or file_backed" 2>&1 | tail -2
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$R/rate -- python3 tools/inflate_rate.py > gpurun_out/$R/rate.log 2>&1
f=$(find gpurun_out/$R/rate -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'k_inflate' in r['Kernel_Name']]
names = ['plane0 stored', 'plane2 literals+short matches', 'plane3 long matches', 'whole level']
for k, r in enumerate(rows):
    if k % 2:
        print(f"{names[k // 2]:32s} 256 streams: {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6:8.2f} ms")
PY
