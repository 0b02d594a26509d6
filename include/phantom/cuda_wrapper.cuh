// cuda_wrapper.cuh -- stands in for the reference's include/cuda_wrapper.cuh:19-283 (installed as include/phantom/cuda_wrapper.cuh, CMakeLists.txt:67-70):
// cudaStream_t alias, cuda_stream_wrapper, cuda_auto_ptr, make_cuda_auto_ptr.
// The declarations live in one header, phantom-fhe_amd/host/phantom.h (the MI355X host mirror over the C ABI of
// include/phantom_amd.h); this file only gives it the reference's file name, so that `#include "cuda_wrapper.cuh"` (with
// -I include/phantom) and `#include <phantom/cuda_wrapper.cuh>` (with -I include) resolve as they do against the reference.
#pragma once
#include "../../phantom-fhe_amd/host/phantom.h"
