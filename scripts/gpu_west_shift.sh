# Round-4 verdict W5, the one bounded store experiment: the one-field flux kernel with its west slots built from lane-shifted
# values (aligned 16-byte stores) against today's 8-byte stores -- in-process A/B of K1 on the ORCA025-like step (both
# dtypes), then WRITE_SIZE of both forms from two counter passes.  bash scripts/gpu_west_shift.sh <tag>
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=${1:-r05}
mkdir -p gpurun_out/$R
O=gpurun_out/$R/west_shift.txt
python -m pytest tests -m gpu -q -k "field_split_bit_identical or headline_kernel_bit_exact" > gpurun_out/$R/west_shift_tests.log 2>&1 || { tail -40 gpurun_out/$R/west_shift_tests.log; exit 1; }
tail -2 gpurun_out/$R/west_shift_tests.log > $O
for dt in float32 float64; do
  echo "== ORCA025-like 1440 x 1021 x 75, one step, $dt: west_shift 0 | 1 (in-process, interleaved rounds)" >> $O
  python tools/ab_flux.py --variants 0 --rounds 15 --nt 4 --nx 1440 --ny 1021 --dtype $dt --knobs 'west_shift=0;west_shift=1' >> $O 2>&1
done
for ws in 0 1; do
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/$R/ws_write_$ws -- python3 bench.py --only-c3 --steps 5 --knob west_shift=$ws > /dev/null 2> gpurun_out/$R/ws_write_$ws.err
  python3 - $R $ws >> $O <<'PY'
import csv, glob, sys
R, ws = sys.argv[1], sys.argv[2]
fn = glob.glob(f'gpurun_out/{R}/ws_write_{ws}/**/*counter_collection.csv', recursive=True)[0]
acc = {}
for r in csv.DictReader(open(fn)):
    if 'k_flux_field' in r['Kernel_Name'] and r['Counter_Name'] == 'WRITE_SIZE':
        key = 'float' if 'k_flux_field<float' in r['Kernel_Name'] else 'double'
        acc.setdefault(key, []).append(float(r['Counter_Value']))
for k, v in acc.items():
    v = v[len(v) // 2:]
    print(f'west_shift={ws} k_flux_field<{k}>: WRITE_SIZE {sum(v) / len(v):.0f} KiB per launch over {len(v)} launches (algorithmic 6 planes x 1470240 x 8 B = {6 * 1470240 * 8 / 1024:.0f} KiB)')
PY
done
cat $O
