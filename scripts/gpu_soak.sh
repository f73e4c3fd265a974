# the whole -m gpu suite three times in a row on one box (flakiness check)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$1
for i in 1 2 3; do
  python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/$1/run$i.log 2>&1; rc=$?
  tail -1 gpurun_out/$1/run$i.log
  [ $rc -eq 0 ] || { tail -40 gpurun_out/$1/run$i.log; exit 1; }
done
