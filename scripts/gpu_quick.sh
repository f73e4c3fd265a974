set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -5
python bench.py --steps 10 --warmup 2 --no-cpu > gpurun_out/bench3.json 2> gpurun_out/bench3.err
python -c "import json; d=json.load(open('gpurun_out/bench3.json')); print(d['value'], d['ms_per_step'], d['roofline'], d['accuracy'])"
