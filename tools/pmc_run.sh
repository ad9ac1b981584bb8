#!/bin/bash
# usage (on the GPU box): bash tools/pmc_run.sh <codec> [kernel-substring]
# Separate rocprofv3 --pmc passes (kernel-trace only, as the pool requires) over a 2 GiB single-step bench run.
codec=${1:-lz4}
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/pmc_$codec
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
           "SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAIT_ANY" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_INT64 SQ_BUSY_CU_CYCLES"; do   # (FETCH_SIZE and WRITE_SIZE: one pass EACH, tools/round_profile.sh - a pass that asks for both aborts and hangs)
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/$i -o p -- python3 $R/bench.py --codec $codec --gib 2 --steps 1 --warmup 0 --no-cpu-baseline --no-by-kind --base-mib 64 > $out/$i.log 2>&1
done
python3 $R/tools/pmc_summary.py $out ${2:-${codec}_}
