# kernel shares of the weight build (K2) at 512 transects on the ORCA12-like grid: bash scripts/gpu_weights_profile.sh <tag> <name>
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=${1:-r05}; N=${2:-weights_profile}
mkdir -p gpurun_out/$R
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/$N -- python3 tools/weights_scaling.py profile 512 > gpurun_out/$R/$N.log 2>&1
python3 - $R $N <<'PY'
import csv, glob, sys
R, N = sys.argv[1], sys.argv[2]
fn = glob.glob(f'gpurun_out/{R}/{N}/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(fn)))
out = open(f'gpurun_out/{R}/{N}.txt', 'w')
print(open(f'gpurun_out/{R}/{N}.log').read().strip().split('\n')[-1], file=out)
for r in rows[:14]:
    print(f"{r['Name'][:90]:90s} calls {int(r['Calls']):5d} total {int(r['TotalDurationNs']) / 1e6:9.2f} ms avg {float(r['AverageNs']) / 1e6:9.3f} ms  {float(r['Percentage']):5.1f} %", file=out)
out.close()
print(open(f'gpurun_out/{R}/{N}.txt').read())
PY
