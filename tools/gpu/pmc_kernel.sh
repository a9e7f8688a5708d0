# PMC passes over one kernel configuration:  bash tools/gpu/pmc_kernel.sh <tag> <one_kernel.py args...>
# pass 1: where the wavefront cycles go; pass 2: LDS / instruction mix.  Output: gpurun_out/pmc_<tag>_<pass>.csv (per-kernel means)
tag=$1; shift
cd /tmp
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"; do
  name=$(echo $pass | cut -d' ' -f1)
  rm -rf /tmp/pmc_$tag
  timeout 300 rocprofv3 --pmc $pass --output-format csv -d /tmp/pmc_$tag -o p -- python $GRAFT_REPO_ROOT/tools/gpu/one_kernel.py "$@" > /tmp/pmc_$tag.log 2>&1
  python $GRAFT_REPO_ROOT/tools/pmc_table.py $(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_${name}.csv 2>/dev/null || tail -3 /tmp/pmc_$tag.log
done
cd $GRAFT_REPO_ROOT
