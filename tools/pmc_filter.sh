#!/bin/bash
# tools/pmc_filter.sh: SQ / LDS / TA counters of the Filter::run kernels (bench.py --filter), one rocprofv3 --pmc pass per group
set -e
out=$GRAFT_REPO_ROOT/gpurun_out/pmcf
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 0 --no-config5 --no-config4 --filter --cpu-seconds 0"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_LDS_ADDR_CONFLICT" \
           "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $out/p$i -o p$i -- $B > $out/p$i.log 2>&1 || echo "pass $i failed"
done
cd $GRAFT_REPO_ROOT
for k in "k_filter_neighbor<2048" k_groups_edges k_filter_exact k_filter_outside k_depth_maps k_filter_vimages k_index_count; do
  python3 tools/pmc_kernel.py $k $(find $out -name "*counter_collection.csv") > "$out/pmc_${k%%<*}.json"
done
