# bench.py under torch.distributed.run with N ranks sharing the one GPU of the box (gloo: the rehearsal hook), N = 4 and 6,
# against the N = 1 line of the same small problem: totals to 1e-13.  Keeps to the 6-process limit of a box.
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=$1
mkdir -p gpurun_out/$R
ARGS="--nx 144 --ny 72 --nz 9 --nt 5 --batch 6 --steps 2 --warmup 1 --no-cpu --no-f32 --no-ingest --no-c3 --dump-totals"
python bench.py --gpus 1 $ARGS > gpurun_out/$R/n1.json 2> gpurun_out/$R/n1.err
for N in 4 6; do
  NF_FORCE_DEVICE=0 NF_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500+N)) bench.py --gpus $N $ARGS > gpurun_out/$R/n$N.json 2> gpurun_out/$R/n$N.err || { tail -30 gpurun_out/$R/n$N.err; exit 1; }
done
python - <<PY
import json, numpy
one = json.load(open('gpurun_out/$R/n1.json'))
for N in (4, 6):
    d = json.loads([l for l in open('gpurun_out/$R/n%d.json' % N) if l.startswith('{')][0])
    a, b = numpy.array(d['totals']), numpy.array(one['totals'])
    print(N, 'ranks: n_gpus', d['n_gpus'], 'max rel diff vs N=1', float(numpy.abs(a - b).max() / numpy.abs(b).max()),
          'slabs', [r['slabs'] for r in d['ranks']], 'launches', [r['launches_per_pass'] for r in d['ranks']])
    assert numpy.abs(a - b).max() <= 1e-13 * numpy.abs(b).max()
print('rehearsal OK')
PY
