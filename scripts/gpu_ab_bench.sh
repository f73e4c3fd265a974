set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "variants" 2>&1 | tail -2
for r in 1 2 3; do
for v in 0 50 40; do
  NF_FLUX_VARIANT=$v python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('variant $v', round(d['ms_per_step'],3), d['roofline']['avg_launch_ms'], f\"{d['value']:.4e}\")"
done; done
