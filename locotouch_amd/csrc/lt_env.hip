// placeholder until the kernels land
#include <hip/hip_runtime.h>
#include "lt_internal.h"
int lt_launch_reset_all(const lt_env*, void*) { return (int)hipErrorNotSupported; }
int lt_launch_step(const lt_env*, const float*, void*) { return (int)hipErrorNotSupported; }
int lt_launch_eval_terms(const lt_env*, void*) { return (int)hipErrorNotSupported; }
int lt_launch_set_command_ranges(const lt_env*, const float*, int, float, void*) { return (int)hipErrorNotSupported; }
const char* lt_hip_error_string(int err) { return hipGetErrorString((hipError_t)err); }
