# Per-rank compute evidence for the N > 1 strong-scaling run, on ONE GPU (round-3 verdict item 2b): for N in 2 4 8 every
# rank r's pass is timed by `bench.py --emulate-rank r/N` (its slab range, its launches, no reduce); for N = 8 also with the
# six-plane epilogue kept on partial steps (--knob partial_step_planes=1 = the code before round 4).  NOT a scaling curve:
# no RCCL, no xGMI, one GPU after the other.  Output: gpurun_out/$R/rank_emulation.txt (copied to profiles/ by hand).
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=${1:-r04}
O=gpurun_out/$R
mkdir -p $O
: > $O/rank_emulation.jsonl
python bench.py --no-cpu --no-f32 --no-ingest --no-c3 --steps 20 --warmup 3 > $O/emu_n1.json 2> $O/emu_n1.err
for N in 2 4 8; do
  for r in $(seq 0 $((N-1))); do
    python bench.py --emulate-rank $r/$N --steps 20 --warmup 3 >> $O/rank_emulation.jsonl 2>> $O/emu.err
    echo "emulated $r/$N"
  done
done
for r in $(seq 0 7); do
  python bench.py --emulate-rank $r/8 --steps 20 --warmup 3 --knob partial_step_planes=1 >> $O/rank_emulation.jsonl 2>> $O/emu.err
  echo "emulated $r/8 (six planes on partial steps)"
done
python scripts/summarize_rank_emulation.py $O/emu_n1.json $O/rank_emulation.jsonl > $O/rank_emulation.txt
cat $O/rank_emulation.txt
