"""In-tree build of the gfx950 engine: csrc/*.hip -> aocl-compression_amd/libaocl_compression_gpu.so.

hipcc compiles each translation unit for gfx950 (device code is cross-compiled; no GPU needed).
The link step does NOT hard-wire a particular libamdhip64: the shared object records a plain
``NEEDED libamdhip64.so`` (via an empty link-time stub), so that
  * in a process that already loaded a HIP runtime under that name (PyTorch-ROCm bundles its own
    copy and loads it first) the engine binds to THAT runtime -- one HIP runtime per process, so
    torch-allocated device pointers and torch streams can be handed to the C-ABI directly;
  * in a plain C process the runtime is found through RUNPATH (/opt/rocm/lib).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libaocl_compression_gpu.so")
# the same code with only the engine's own names exported (aoclgpu_*, aocl_llc_*, RAP helpers): for programs that also link
# upstream libzstd / liblz4 / libsnappy - e.g. users of the sequence producer - where the drop-in native names must not clash
CORE_LIB = os.path.join(HERE, "libaocl_compression_gpu_core.so")
CLI = os.path.join(HERE, "aocl_compression_bench")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HIPCC = os.path.join(ROCM, "bin", "hipcc")
ARCH = "gfx950"
CLI_SOURCE = "bench_cli.cpp"
SOURCES = ["capi.hip", "lz4_kernels.hip", "snappy_kernels.hip", "zstd_kernels.hip", "rap.hip", "aocl_api.cpp", "native_api.cpp", "zstd_api.cpp", "lz4frame_api.cpp", "exchange.cpp"]
CFLAGS = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-value", "-Wno-unused-function",
          *os.environ.get("AOCLGPU_CFLAGS", "").split()]


def _newer(src: str, out: str, deps) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in [src, *deps])


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + r.stderr)
        raise RuntimeError("build step failed: " + " ".join(cmd[:3]))
    return r


def build_sanitized(verbose: bool = False) -> str:
    """AOCLGPU_SANITIZE=1 / `python build.py --sanitize`: the HOST side of the library (the .cpp shims and the host halves of
    the .hip files) with AddressSanitizer + UndefinedBehaviorSanitizer -> libaocl_compression_gpu_asan.so.  Device code is
    compiled as always (the GPU sanitizers need xnack+ boxes this pool does not have).  CPU-only use: tests/test_sanitized.py
    preloads the clang ASan runtime and drives the host logic (ABI structs, level tables, frame-header parsing, error paths)."""
    return build_variant("asan", ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-g", "-Wno-option-ignored"], verbose=verbose,
                         linker=[os.path.join(ROCM, "lib", "llvm", "bin", "clang++"), "-fsanitize=address,undefined", "-shared-libsan"])


def asan_runtime() -> str:
    r = subprocess.run([os.path.join(ROCM, "lib", "llvm", "bin", "clang++"), "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True)
    return r.stdout.strip()


def build_variant(name: str, cflags, verbose: bool = False, linker=None) -> str:
    """Development builds next to the product (kernel experiments, profiling instantiations): the same sources with extra
    flags -> libaocl_compression_gpu_<name>.so, objects under build/<name>/.  Loaded by tools/ through AOCLGPU_LIB (api.lib_path)."""
    objdir = os.path.join(OBJDIR, name)
    os.makedirs(objdir, exist_ok=True)
    out = os.path.join(HERE, f"libaocl_compression_gpu_{name}.so")
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers += [os.path.join(HERE, "..", "include", f) for f in os.listdir(os.path.join(HERE, "..", "include"))]
    stamp = os.path.join(objdir, "flags.txt")
    flags = " ".join(cflags)
    same = os.path.exists(stamp) and open(stamp).read() == flags
    jobs, objs = [], []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(objdir, s + ".o")
        objs.append(obj)
        if not same or _newer(src, obj, headers):
            cmd = [HIPCC, f"--offload-arch={ARCH}", *CFLAGS, *cflags, "-I" + os.path.join(HERE, "..", "include")]
            if s.endswith(".cpp"):
                cmd += ["-x", "hip"]
            jobs.append(cmd + ["-c", src, "-o", obj])
    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(_run, jobs))
    stubdir = os.path.join(OBJDIR, "stub")
    if jobs or not os.path.exists(out):
        _run([*(linker or ["g++"]), "-shared", "-fPIC", *objs, "-o", out, "-L" + stubdir, "-Wl,--no-as-needed", "-lamdhip64",
              "-Wl,--as-needed", "-Wl,-rpath," + os.path.join(ROCM, "lib"), "-lpthread", "-ldl"])
    with open(stamp, "w") as f:
        f.write(flags)
    if verbose:
        print("built", out)
    return out


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers += [os.path.join(HERE, "..", "include", f) for f in os.listdir(os.path.join(HERE, "..", "include"))]
    jobs, objs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJDIR, s + ".o")
        objs.append(obj)
        if force or _newer(src, obj, headers):
            cmd = [HIPCC, f"--offload-arch={ARCH}", *CFLAGS, "-I" + os.path.join(HERE, "..", "include")]
            if s.endswith(".cpp"):
                cmd += ["-x", "hip"]
            jobs.append(cmd + ["-c", src, "-o", obj])
    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(_run, jobs))
    if force or jobs or not os.path.exists(LIB) or not os.path.exists(CORE_LIB):
        stubdir = os.path.join(OBJDIR, "stub")
        os.makedirs(stubdir, exist_ok=True)
        stub_c = os.path.join(stubdir, "empty.c")
        with open(stub_c, "w") as f:
            f.write("/* link-time stand-in so the engine records NEEDED libamdhip64.so (no version) */\n")
        _run(["gcc", "-shared", "-fPIC", stub_c, "-o", os.path.join(stubdir, "libamdhip64.so")])
        _run(["g++", "-shared", "-fPIC", *objs, "-o", LIB, "-L" + stubdir, "-Wl,--no-as-needed", "-lamdhip64",
              "-Wl,--as-needed", "-Wl,-rpath," + os.path.join(ROCM, "lib"), "-lpthread", "-ldl", "-Wl,-Bsymbolic-functions"])
        vmap = os.path.join(stubdir, "core.map")
        with open(vmap, "w") as f:
            f.write("{ global: aoclgpu_*; aocl_llc_*; aocl_get_rap_frame_bound_mt; aocl_skip_rap_frame_mt; local: *; };\n")
        _run(["g++", "-shared", "-fPIC", *objs, "-o", CORE_LIB, "-L" + stubdir, "-Wl,--no-as-needed", "-lamdhip64",
              "-Wl,--as-needed", "-Wl,-rpath," + os.path.join(ROCM, "lib"), "-lpthread", "-ldl", "-Wl,--version-script=" + vmap])
    # the bench CLI (counterpart of the reference's aocl_compression_bench, test/codec_bench.c)
    cli_src = os.path.join(CSRC, "bench_cli.cpp")
    if force or not os.path.exists(CLI) or _newer(cli_src, CLI, headers + [LIB]):
        _run(["g++", "-O2", "-std=c++17", "-Wall", "-I" + os.path.join(HERE, "..", "include"), cli_src, "-o", CLI,
              "-L" + HERE, "-laocl_compression_gpu", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + os.path.join(ROCM, "lib"),
              "-Wl,--allow-shlib-undefined"])
    if verbose:
        print("built", LIB, "and", CLI)
    return LIB


def clean_variants(verbose: bool = False) -> None:
    """Development builds (build_variant: kernel experiments, profiling instantiations) are scratch: the product build removes
    their libraries and object trees so that they do not ship with every GPU run or pile up in the package directory.
    Kept: the product, the core library and the sanitised host build.  AOCLGPU_KEEP_VARIANTS=1 (or --keep-variants) keeps them."""
    import shutil
    keep = {os.path.basename(LIB), os.path.basename(CORE_LIB), "libaocl_compression_gpu_asan.so"}
    for f in os.listdir(HERE):
        if f.startswith("libaocl_compression_gpu_") and f.endswith(".so") and f not in keep:
            os.unlink(os.path.join(HERE, f))
            if verbose:
                print("removed", f)
    if os.path.isdir(OBJDIR):
        for d in os.listdir(OBJDIR):
            full = os.path.join(OBJDIR, d)
            if os.path.isdir(full) and d not in ("stub", "asan"):
                shutil.rmtree(full, ignore_errors=True)


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    if "--variant" not in sys.argv and "--keep-variants" not in sys.argv and os.environ.get("AOCLGPU_KEEP_VARIANTS") != "1":
        clean_variants(verbose=True)
    if "--sanitize" in sys.argv or os.environ.get("AOCLGPU_SANITIZE") == "1":
        build_sanitized(verbose=True)
    # python build.py --variant prof -DAOCLGPU_EXPERIMENTS -DZSTD_PROF
    if "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        build_variant(sys.argv[i + 1], sys.argv[i + 2:], verbose=True)
