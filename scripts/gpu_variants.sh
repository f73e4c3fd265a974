set -e
cd $GRAFT_REPO_ROOT
true
for v in ${VARIANTS:-0 1 2 3 4 5 6 7 8 9 10 11 12 13}; do
  NF_FLUX_VARIANT=$v python bench.py --steps 6 --warmup 2 --no-cpu --batch 0 --nt 6 ${BENCH_ARGS} 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('variant $v', d['roofline']['avg_launch_ms'], d['roofline']['achieved'], d['value'])"
done
