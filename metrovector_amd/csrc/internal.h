// internal.h — pieces of api.hip the other translation units of libmvf_gpu.so use.
#pragma once

#include "../../include/mvf_status.h"

#include <cstdint>
#include <cstring>
#include <string>

namespace mvf {

// records the calling thread's failure detail (mvfgpu_last_error_message) and returns `status`
int set_fail(int status, const std::string& msg);

// Out-structs of the C ABI start with a caller-set `struct_size` (include/mvf_gpu.h "OUT-STRUCTS GROW"): copy at most
// that many bytes of `full` and report how many were filled.
template <class T>
int copy_out_struct(T* out, T full) {
    const uint32_t have = out->struct_size;
    if (have < 8u) return set_fail(MVF_ERR_INVALID_ARGUMENT, "struct_size not set (MVFGPU_INIT the out-struct before the call)");
    const uint32_t n = have < (uint32_t)sizeof(T) ? have : (uint32_t)sizeof(T);
    full.struct_size = n;
    std::memcpy(out, &full, n);
    return 0;
}

}  // namespace mvf
