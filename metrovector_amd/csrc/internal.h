// internal.h — pieces of api.hip the other translation units of libmvf_gpu.so use.
#pragma once

#include <string>

namespace mvf {

// records the calling thread's failure detail (mvfgpu_last_error_message) and returns `status`
int set_fail(int status, const std::string& msg);

}  // namespace mvf
