#!/usr/bin/env python3
"""Turn gpurun_out/<tag>/ (tools/round_profile.sh) into the files kept under profiles/:
bench_<round>_<codec>_<ver>.json, <round>_<codec>_<ver>_kernel_trace_and_pmc.txt, traffic.json.
usage: tools/collect_profiles.py <tag> <ver> [codec ...]   e.g.  r02_v1 v1   (round = the tag's prefix up to the first "_")"""
import csv, glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, ver = sys.argv[1], sys.argv[2]
RND = tag.split("_")[0]
O = os.path.join(ROOT, "gpurun_out", tag)
P = os.path.join(ROOT, "profiles")
tpath = os.path.join(P, "traffic.json")
traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
for c in (sys.argv[3:] or ["lz4", "snappy", "zstd"]):
    line = [l for l in open(os.path.join(O, f"bench_{c}.json")) if l.startswith("{")][-1]
    open(os.path.join(P, f"bench_{RND}_{c}_{ver}.json"), "w").write(line)
    bench = json.loads(line)
    dom = bench["roofline"]["kernel"]
    name = f"{RND}_{c}_{ver}_kernel_trace_and_pmc.txt"
    with open(os.path.join(P, name), "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats of: python3 bench.py --codec {c} --no-cpu-baseline --no-by-kind  ({RND}, {ver} kernels, 1x MI355X, 8 GiB)\n")
        rows = list(csv.reader(open(os.path.join(O, f"trace_{c}", f"{c}_kernel_stats.csv"))))
        f.write(",".join(f'"{x}"' for x in rows[0]) + "\n")
        for r in rows[1:]:
            if "aoclgpu" in r[0]:
                f.write(",".join(f'"{x}"' if i == 0 else x for i, x in enumerate(r)) + "\n")
        f.write("# (kernels of the bench harness itself - torch copies, compares, reductions - left out)\n\n# bench.py line of the same (profiled) run\n")
        f.write([l for l in open(os.path.join(O, f"trace_bench_{c}.json")) if l.startswith("{")][-1])
        f.write(f"\n# separate --pmc passes (one counter per pass, kernel-trace only), python3 bench.py --codec {c} --no-cpu-baseline --no-by-kind --steps 1 --warmup 0; raw counter values (KB) per launch\n")
        kb = {}
        per_call = int(((bench["roofline"].get("launches") or {}).get("per_compress_call") or {}).get(dom, 1))   # level 3: one launch per slab
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            txt = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), os.path.join(O, f"pmc_{c}_{ctr}"), ("zstd_" if c.startswith("zstd") else c + "_")], capture_output=True, text=True).stdout
            f.write(txt)
            cur = None
            for l in txt.splitlines():
                if not l.startswith(" "):
                    cur = l.strip()
                elif ctr in l and cur and dom.split("_kernel")[0] in cur:
                    kb[ctr] = int(float(l.split()[1])) * per_call          # (the summary prints the mean per dispatch)
    if len(kb) == 2:
        traffic[c] = {"kernel": dom, "gib_per_gpu": bench["config"]["gib_per_gpu"], "chunk_bytes": bench["config"]["chunk_bytes"],
                      "fetch_kb": kb["FETCH_SIZE"], "write_kb": kb["WRITE_SIZE"],
                      "kernel_source_sha16": bench["roofline"].get("kernel_source_sha16"),     # bench.py drops `traffic` when the sources changed since
                      "source": f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE, separate passes, raw KB per launch)"}
        if per_call > 1:                                                 # per compress CALL: the dominant kernel's launches of one step, summed
            traffic[c]["launches"] = f"{per_call} x {dom} per compress call (one per slab): fetch_kb / write_kb = {per_call} x the mean per launch"
# scattered-load calibration of FETCH_SIZE (tools/probe_scatter.py under --pmc FETCH_SIZE)
try:
    rec = json.loads([l for l in open(os.path.join(O, "scatter8.json")) if l.startswith("{")][-1])
    txt = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), os.path.join(O, "pmc_scatter8"), "probe_scatter8"], capture_output=True, text=True).stdout
    kbv = [float(l.split()[1]) for l in txt.splitlines() if "FETCH_SIZE" in l]
    if kbv:
        name = f"{RND}_scatter8_fetch_calibration.txt"
        open(os.path.join(P, name), "w").write(f"# rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 tools/probe_scatter.py   ({rec['loads']} independent 8-byte loads at random addresses of a 2 GiB buffer)\n" + txt +
                                                  f"# FETCH_SIZE is in KB: {kbv[0] * 1000.0 / rec['loads']:.1f} bytes fetched per 8-byte load\n")
        traffic["scatter8_calibration"] = {"loads": rec["loads"], "fetch_kb": kbv[0], "fetch_bytes_per_load": round(kbv[0] * 1000.0 / rec["loads"], 2),
                                           "source": f"profiles/{name}"}
except (OSError, ValueError, IndexError, KeyError):
    pass
json.dump(traffic, open(tpath, "w"), indent=2)
print(json.dumps(traffic, indent=1))
