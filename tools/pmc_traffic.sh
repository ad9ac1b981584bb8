#!/bin/bash
# usage (on the GPU box): bash tools/pmc_traffic.sh <codec> [kernel-substring]
# FETCH_SIZE and WRITE_SIZE (one counter per rocprofv3 --pmc pass, kernel-trace only) of a 2 GiB single-step bench run: the quick
# look at a kernel's HBM traffic between two builds (tools/round_profile.sh makes the 8 GiB passes that are kept under profiles/).
codec=${1:-zstd}
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/pmc_traffic_$codec
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/$ctr -o p -- python3 $R/bench.py --codec $codec --gib 2 --steps 1 --warmup 0 --no-cpu-baseline --no-by-kind --base-mib 64 > $out/$ctr.log 2>&1
done
python3 $R/tools/pmc_summary.py $out ${2:-${codec}_}
