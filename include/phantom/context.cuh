// context.cuh -- stands in for the reference's include/context.cuh:19-273 (installed as include/phantom/context.cuh, CMakeLists.txt:67-70):
// phantom::ContextData, PhantomContext (gpu_rns_tables(), get_context_data(i).gpu_rns_tool()).
// The declarations live in one header, phantom-fhe_amd/host/phantom.h (the MI355X host mirror over the C ABI of
// include/phantom_amd.h); this file only gives it the reference's file name, so that `#include "context.cuh"` (with
// -I include/phantom) and `#include <phantom/context.cuh>` (with -I include) resolve as they do against the reference.
#pragma once
#include "../../phantom-fhe_amd/host/phantom.h"
