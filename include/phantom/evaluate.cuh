// evaluate.cuh -- stands in for the reference's include/evaluate.cuh:12-400 (installed as include/phantom/evaluate.cuh, CMakeLists.txt:67-70):
// phantom::key_switch_inner_prod, keyswitch_inplace and the evaluate.* functions of the hot path.
// The declarations live in one header, phantom-fhe_amd/host/phantom.h (the MI355X host mirror over the C ABI of
// include/phantom_amd.h); this file only gives it the reference's file name, so that `#include "evaluate.cuh"` (with
// -I include/phantom) and `#include <phantom/evaluate.cuh>` (with -I include) resolve as they do against the reference.
#pragma once
#include "../../phantom-fhe_amd/host/phantom.h"
