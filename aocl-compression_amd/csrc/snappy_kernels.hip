// csrc/snappy_kernels.hip -- placeholder until the snappy kernels land (launchers fail loudly).
#include "common.hip.h"
#include "host_util.h"
extern "C" int aoclgpu_snappy_compress_chunks(const aoclgpu_batch*, void*) { aoclgpu::set_error_msg("snappy compress: not implemented"); return AOCLGPU_ERR_ARG; }
extern "C" int aoclgpu_snappy_decompress_chunks(const aoclgpu_batch*, void*) { aoclgpu::set_error_msg("snappy decompress: not implemented"); return AOCLGPU_ERR_ARG; }
