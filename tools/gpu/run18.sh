cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in 0 1 0 1; do
PF_FUSE_BN_BWD_STATS=$v timeout 600 python bench.py --steps 10 --warmup 5 --no_cpu_baseline 2>&1 | tail -1 | cut -c100-200
done
