#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/pmc_icache
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -E "icache|ifetch|SQC_" | head -60 > $out/avail.txt
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_INPUT_VALID_READYB SQC_TC_INST_REQ"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/$i -o p -- python3 $R/bench.py --codec ${1:-zstd} --gib 2 --steps 1 --warmup 0 --no-cpu-baseline --no-by-kind --base-mib 64 > $out/$i.log 2>&1
done
python3 $R/tools/pmc_summary.py $out ${1:-zstd}_
