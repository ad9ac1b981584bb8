#!/usr/bin/env python3
"""Static instruction counts of one kernel of a hipcc -S listing, split at the `; <marker>` comments the sources plant with
asm volatile (e.g. "; batch begin" / "; resolve begin" in csrc/lz4_kernels.hip), by issue class.  What is counted is every
instruction that lies between two markers in the listing - rarely taken side paths included - so it bounds a stage from above.
A kernel that holds several inlined instances of one function repeats its markers: `--first <marker>` starts a new table at every
occurrence of that marker (e.g. "batch begin"), so each instance is counted on its own.
usage: tools/isa_region_count.py <listing.s> <kernel symbol substring> [--first <marker>]"""
import re, sys, collections
path, sym = sys.argv[1], sys.argv[2]
first = sys.argv[sys.argv.index("--first") + 1] if "--first" in sys.argv else None
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^[_A-Za-z0-9]+:", l) and sym in l)
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
def cls(op):
    if op.startswith("s_cbranch") or op.startswith("s_branch") or op.startswith("s_setpc"): return "branch"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"): return "wait/nop"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "smem"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("global_load") or op.startswith("buffer_load") or op.startswith("flat_load") or op.startswith("scratch_load"): return "vmem_rd"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"): return "vmem_wr"
    if op.startswith("v_"): return "valu"
    return "other"
tables = [collections.OrderedDict()]
regions, cur = tables[0], "prologue"
regions[cur] = collections.Counter()
for l in lines[start + 1:end + 1]:
    t = l.strip()
    m = re.match(r"^; ([a-z0-9 +<>-]+ (begin|end))$", t)
    if m:
        if first and m.group(1) == first and any(k != "prologue" for k in regions):
            tables.append(collections.OrderedDict()); regions = tables[-1]
        cur = "after: " + m.group(1)
        regions.setdefault(cur, collections.Counter())
        continue
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":") or t.startswith("//"): continue
    op = t.split()[0]
    if not re.match(r"^[a-z_0-9]+$", op): continue
    regions[cur][cls(op)] += 1
cols = ["salu", "branch", "valu", "lds", "vmem_rd", "vmem_wr", "smem", "wait/nop"]
for i, regions in enumerate(tables):
    if len(tables) > 1: print(f"--- instance {i}")
    print(f"{'region':34s}" + "".join(f"{c:>9s}" for c in cols) + "    total")
    for r, c in regions.items():
        print(f"{r:34s}" + "".join(f"{c[k]:9d}" for k in cols) + f"{sum(c.values()):9d}")
