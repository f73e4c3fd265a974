set -e
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
for v in 33554432 134217728; do
  NF_BATCH_CELLSTEPS=$v python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('cellsteps $v', round(d['ms_per_step'],3), d['roofline']['avg_launch_ms'], d['roofline']['launches'], d['roofline']['achieved'], f\"{d['value']:.4e}\", d['accuracy']['max_abs_err_vs_fluxexact'])"
done; done
