set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$1
python -m pytest tests -m gpu -q -k "geometry or box or sa_T" 2>&1 | tail -2
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$1/geom -- python3 tools/geom_timing.py > gpurun_out/$1/geom.txt 2>&1
cat gpurun_out/$1/geom.txt | grep -v amdgpu.ids
python3 - <<PY
import glob, csv
for f in glob.glob('gpurun_out/$1/geom/**/*kernel_stats.csv', recursive=True):
    for row in list(csv.reader(open(f)))[:5]:
        print(row[:4])
PY
