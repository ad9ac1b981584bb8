#!/bin/bash
# usage (on the GPU box): bash tools/round_profile.sh <tag> [codec ...]      (default: lz4 snappy zstd)
# Full bench lines (with cpu_baseline) for the three codecs; per codec a rocprofv3 --kernel-trace --stats run of the same
# bench command and separate --pmc passes (FETCH_SIZE, WRITE_SIZE: ONE counter per pass - a pass that asks for both aborts
# and hangs) for HBM traffic.  Outputs: gpurun_out/<tag>/
tag=${1:-r01}
shift
codecs=${@:-lz4 snappy zstd}
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
for c in $codecs; do
  timeout 600 python3 bench.py --codec $c > $out/bench_$c.json 2> $out/bench_$c.err
done
cd /tmp && export TMPDIR=/tmp
for c in $codecs; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$c -o $c -- python3 $R/bench.py --codec $c --no-cpu-baseline --no-by-kind > $out/trace_bench_$c.json 2> $out/trace_$c.err
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_${c}_$ctr -o p -- python3 $R/bench.py --codec $c --no-cpu-baseline --no-by-kind --steps 1 --warmup 0 > $out/pmc_${c}_$ctr.json 2> $out/pmc_${c}_$ctr.err
  done
done
# FETCH_SIZE calibration for scattered 8-byte loads (bench.py `fetch_calibration`)
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_scatter8 -o p -- python3 $R/tools/probe_scatter.py > $out/scatter8.json 2> $out/scatter8.err
ls -R $out | head -80
