set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --steps 10 --warmup 2 > gpurun_out/bench2.json 2> gpurun_out/bench2.err
cat gpurun_out/bench2.json | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], d['roofline'], d['accuracy'], d['config']['weight_entries'])"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r01 -- python3 bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/bench_prof.json 2> gpurun_out/bench_prof.err
find gpurun_out/prof_r01 -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -12 {}'
for v in "NF_FLUX_UZ=3" "NF_FLUX_UZ=4" "NF_FLUX_UZ=8" "NF_FLUX_NT=0" "NF_XCD_MAP=0"; do
  echo "== $v"; env $v python bench.py --steps 5 --warmup 1 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['achieved'])"
done
