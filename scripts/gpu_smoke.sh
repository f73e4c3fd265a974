set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$1
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/$1/smoke.log 2>&1 || { tail -30 gpurun_out/$1/smoke.log; exit 1; }
tail -3 gpurun_out/$1/smoke.log
python -m pytest tests -m gpu -q -k "plain_c_client or readme_examples" 2>&1 | tail -3
