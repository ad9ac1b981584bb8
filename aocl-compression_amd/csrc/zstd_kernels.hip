// csrc/zstd_kernels.hip -- placeholder until the zstd kernels land (launchers fail loudly).
#include "common.hip.h"
#include "host_util.h"
extern "C" int aoclgpu_zstd_compress_chunks(const aoclgpu_batch*, void*) { aoclgpu::set_error_msg("zstd compress: not implemented"); return AOCLGPU_ERR_ARG; }
extern "C" int aoclgpu_zstd_decompress_chunks(const aoclgpu_batch*, void*) { aoclgpu::set_error_msg("zstd decompress: not implemented"); return AOCLGPU_ERR_ARG; }
