#!/bin/bash
# oracle/build_ref.sh -- TEST INFRASTRUCTURE ONLY.
# Compiles the REAL reference (amd/aocl-compression 4.2.0) hot-path codecs with gcc/g++ straight from
# the sources where they lie under $REF (read-only; nothing is copied into this repo) into
# oracle/_ref/libaocl_ref.so.  It is the ground truth the C restatement (liboracle.so) and the HIP
# path are pinned against, and bench.py's cpu_baseline (kind "reference").
#
# Flags mirror the reference's Release build (CMakeLists.txt:378-380,466-468).  Codecs outside the
# hot path are dropped with the reference's own AOCL_EXCLUDE_* switches (CMakeLists.txt:82-88); every
# parity-changing option stays at its default OFF.  The reference's own build
# system (cmake) is NOT run.
set -e
REF=${REF:-/root/reference}
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"; OBJ="$OUT/obj"
if [ ! -f "$REF/api/api.cpp" ]; then
  echo "build_ref: reference tree $REF absent; keeping prebuilt $OUT/libaocl_ref.so (if any)"; exit 0
fi
if [ -f "$OUT/libaocl_ref.so" ] && [ -f "$OUT/libaocl_ref_mt.so" ] && [ "$OUT/libaocl_ref_mt.so" -nt "$0" ]; then
  echo "build_ref: $OUT/libaocl_ref.so up to date"; exit 0
fi
build_variant() {  # $1 = output library, $2 = object dir, $3 = extra defines, $4 = extra C sources
OUTLIB="$1"; OBJ="$2"; EXTRA="$3"
mkdir -p "$OBJ"
FLAGS="-O3 -DNDEBUG -fPIC -fomit-frame-pointer -fstrict-aliasing -ffast-math -fopenmp -w $EXTRA \
 -DAOCL_DYNAMIC_DISPATCHER -DAOCL_EXCLUDE_BZIP2 -DAOCL_EXCLUDE_LZMA -DAOCL_EXCLUDE_ZLIB \
 -I$REF -I$REF/api -I$REF/utils -I$REF/algos/common"
VERS=(-DAOCL_BUILD_VERSION="\"Build_oracle\"" -DCFLAGS_SET1="\"-O3\"" -DCFLAGS_SET2="\"\"" -DCCompiler="\"gcc\"" -DCXXCompiler="\"g++\"")
SNAP="-msse4.1 -DHAVE_BUILTIN_CTZ=1 -DHAVE_BUILTIN_EXPECT=1 -DHAVE_SYS_UIO_H"
pids=()
for f in utils/utils.cpp api/api.cpp api/codec.cpp; do cc1 g++ "$REF/$f" "${VERS[@]}"; done
for f in snappy-sinksource snappy-stubs-internal snappy snappy-c; do cc1 g++ "$REF/algos/snappy/$f.cc" $SNAP "${VERS[@]}"; done
waitall
cc1 gcc "$REF/algos/lz4/lz4.c"; cc1 gcc "$REF/algos/lz4/lz4hc.c"; cc1 gcc "$REF/algos/lz4/xxhash.c"   # lz4hc: producer of LZ4HC streams for the decoder tests
cc1 gcc "$REF/algos/lz4/lz4frame.c"                        # LZ4 frame format (LZ4_FRAME_FORMAT_SUPPORT, CMakeLists.txt:311-325): ground truth of csrc/lz4frame_api.cpp
for f in $4; do cc1 gcc "$REF/$f"; done
n=0
for f in "$REF"/algos/zstd/lib/common/*.c "$REF"/algos/zstd/lib/compress/*.c "$REF"/algos/zstd/lib/decompress/*.c "$REF"/algos/zstd/lib/decompress/*.S; do
  case "$f" in *code_alignment*) continue;; esac
  cc1 gcc "$f"; n=$((n+1)); if [ $((n % 8)) -eq 0 ]; then waitall; fi
done
waitall
g++ -shared -fopenmp -o "$OUTLIB" "$OBJ"/*.o
echo "build_ref: built $OUTLIB"
}
cc1() { # $1 compiler, $2 src, rest flags
  local comp=$1 src=$2; shift 2
  local o="$OBJ/$(echo "${src#$REF/}" | tr '/' '_').o"
  $comp $FLAGS "$@" -c "$src" -o "$o" &
  pids+=($!)
}
waitall() { for p in "${pids[@]}"; do wait "$p" || { echo "build_ref: compile failed"; exit 1; }; done; pids=(); }
# default build (single-threaded codecs: the parity ground truth and the cpu_baseline functions)
build_variant "$OUT/libaocl_ref.so" "$OUT/obj" "" ""
# AOCL_ENABLE_THREADS build (CMakeLists.txt:81,256-273,473-474): the RAP-stream producer/consumer the GPU RAP
# pack / unpack (csrc/rap.hip) is checked against
build_variant "$OUT/libaocl_ref_mt.so" "$OUT/obj_mt" "-DAOCL_ENABLE_THREADS" "threads/threads.c"
