// ntt.cuh -- stands in for the reference's include/ntt.cuh:6-226 (installed as include/phantom/ntt.cuh, CMakeLists.txt:67-70):
// DModulus, DNTTTable (handle) and the nwt_2d_radix8_* launchers.
// The declarations live in one header, phantom-fhe_amd/host/phantom.h (the MI355X host mirror over the C ABI of
// include/phantom_amd.h); this file only gives it the reference's file name, so that `#include "ntt.cuh"` (with
// -I include/phantom) and `#include <phantom/ntt.cuh>` (with -I include) resolve as they do against the reference.
#pragma once
#include "../../phantom-fhe_amd/host/phantom.h"
