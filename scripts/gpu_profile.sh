# Profile of the headline bench (float64 workload + float32 sub-record in one process): kernel trace + stats, then HBM traffic counters in two separate PMC passes
# (MI355X_MICROARCH.md: FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2 -- they do not fit one pass); the same three passes for the
# file-ingest kernels (tools/inflate_rate.py: nf::k_inflate / nf::k_place4 on 256 streams per launch).  Every number that
# profiles/README.md and DESIGN.md quote is printed by scripts/summarize_profile.py into <tag>_numbers.md.
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=${1:-r03}
COMMIT=${2:-unknown}     # git rev-parse --short HEAD of the snapshot, given by the caller (the box has no .git)
mkdir -p gpurun_out/$R
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/trace -- python3 bench.py --steps 5 --warmup 1 --no-cpu --no-c3 > gpurun_out/$R/bench_trace.json 2> gpurun_out/$R/bench_trace.err
echo "trace done"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/c3_trace -- python3 bench.py --only-c3 --steps 20 > gpurun_out/$R/bench_c3.json 2> gpurun_out/$R/bench_c3.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/$R/c3_fetch -- python3 bench.py --only-c3 --steps 5 > /dev/null 2> gpurun_out/$R/bench_c3_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/$R/c3_write -- python3 bench.py --only-c3 --steps 5 > /dev/null 2> gpurun_out/$R/bench_c3_write.err
echo "c3 passes done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/$R/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-c3 > gpurun_out/$R/bench_fetch.json 2> gpurun_out/$R/bench_fetch.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/$R/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-c3 > gpurun_out/$R/bench_write.json 2> gpurun_out/$R/bench_write.err
echo "write done"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/ingest_trace -- python3 tools/inflate_rate.py > gpurun_out/$R/ingest_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/$R/ingest_fetch -- python3 tools/inflate_rate.py > gpurun_out/$R/ingest_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/$R/ingest_write -- python3 tools/inflate_rate.py > gpurun_out/$R/ingest_write.log 2>&1
echo "ingest done"
python3 scripts/summarize_profile.py gpurun_out/$R $R $COMMIT
