# round 2, GPU call 16: where do the wavefront cycles of the shared-tile backward-filter kernel go (PMC, two shapes)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/gpu/pmc_kernel.sh wrw2_14_1024_256 wrw 14 1024 256
bash tools/gpu/pmc_kernel.sh wrw2_56_64_256 wrw 56 64 256
for f in gpurun_out/pmc_wrw2_*; do echo "== $f"; cat $f | cut -c1-250; done
