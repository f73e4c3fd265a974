# How the headline depends on the length of the timed region (sustained load): bench.py with --steps 5 / 10 / 20 / 50 / 200 on one box.
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$1
: > gpurun_out/$1/steps_sweep.txt
for K in 5 10 20 50 200 10; do
  python bench.py --no-cpu --no-f32 --no-ingest --no-c3 --steps $K --warmup 2 > gpurun_out/$1/s$K.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/$1/s$K.json'))
print('steps %4d: value %.4e  ms/pass %.3f  K1 %.4f ms  frac %.4f  wall_frac %.4f' % ($K, d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['wall_frac']))" | tee -a gpurun_out/$1/steps_sweep.txt
done
