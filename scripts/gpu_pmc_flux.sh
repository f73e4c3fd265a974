# Where does the epilogue's time go?  PMC counters of the flux kernel's default build against the diagnostic builds without
# stores (21) and with two planes only (28), collected on the tuning library by tools/ab_flux.py.  One counter group per pass.
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
make -C nemoflux_amd/csrc tuning -j8 -s   # the diagnostic library is built here, on the GPU box: build/ never travels
R=${1:-r02f}
mkdir -p gpurun_out/$R
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES" \
           "TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_EA0_RDREQ_LEVEL" \
           "TCC_EA0_WRREQ_LEVEL TCC_EA0_RDREQ TCC_EA0_WRREQ" \
           "TCC_BUBBLE TCC_IB_STALL TCC_TAG_STALL"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/$R/pmc$i -- python3 tools/ab_flux.py --dtype ${2:-float64} --variants 0,21,28 --rounds 2 --nt 3 > gpurun_out/$R/pmc$i.log 2>&1 || { tail -5 gpurun_out/$R/pmc$i.log; }
  echo "group $i done"
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/$R/pmc*/')):
    fs = glob.glob(d + '**/*counter_collection.csv', recursive=True)
    if not fs: print(d, 'no counters'); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name']
        if 'k_flux' not in k: continue
        acc[k.split('(')[0][-40:]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, cs in acc.items():
        print(k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, 'n=', len(next(iter(cs.values()))))
PY
