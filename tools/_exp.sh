cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for f in 0 287 1024 2048; do
  HVPR_EXP_FILL=$f rocprofv3 --kernel-trace --stats -d gpurun_out/pg_f$f -o g -- python3 tools/bench_group.py --car1 > /dev/null 2>&1
done
