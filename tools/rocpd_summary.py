#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd database (--kernel-trace --stats) into the per-kernel text summary kept under profiles/.
usage: tools/rocpd_summary.py <results.db> [bench.log] > profiles/<name>.txt"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
print("# rocprofv3 --kernel-trace --stats summary (durations in microseconds)")
print(f"{'calls':>6} {'total_us':>14} {'avg_us':>14} {'pct':>7}  kernel")
for name, calls, total, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print(f"{calls:>6} {total:>14.1f} {avg:>14.1f} {pct:>7.3f}  {name[:110]}")
if len(sys.argv) > 2:
    for line in open(sys.argv[2]):
        if line.startswith("{"):
            print("\n# bench.py line of the same (profiled) run\n" + line.strip())
