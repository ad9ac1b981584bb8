#!/bin/bash
# usage (on the GPU box): bash tools/pmc_cache.sh <tag> [codec ...]
# Where a compress kernel's memory traffic is served (verdict r02 item 4: "TCC hit / miss / MALL counters"): separate
# rocprofv3 --pmc passes (kernel-trace only) over a 2 GiB single-step bench run, L2 (TCC) requests / hits / misses,
# reads sent down to the fabric (EA) by size and how many of them reach DRAM (the rest are served by the Infinity Cache),
# write requests, and the L1 -> L2 read latency.  Output: gpurun_out/<tag>/cache_<codec>.txt
tag=${1:-r03_cache}
shift
codecs=${@:-snappy lz4 zstd}
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in $codecs; do
  rm -rf $out/pmc_$c; mkdir -p $out/pmc_$c
  i=0
  for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
             "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_ATOMIC_sum" \
             "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum TCC_WRITE_sum" \
             "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/pmc_$c/$i -o p -- python3 $R/bench.py --codec $c --gib 2 --steps 1 --warmup 0 --no-cpu-baseline --no-by-kind --base-mib 64 > $out/pmc_$c/$i.log 2>&1
  done
  python3 $R/tools/pmc_summary.py $out/pmc_$c ${c}_ > $out/cache_$c.txt 2>&1
  rm -rf $out/pmc_$c
done
cat $out/cache_*.txt
