// rns_bconv.cuh -- stands in for the reference's include/rns_bconv.cuh:3-87 (installed as include/phantom/rns_bconv.cuh, CMakeLists.txt:67-70):
// DBaseConverter: bConv_BEHZ / bConv_HPS.
// The declarations live in one header, phantom-fhe_amd/host/phantom.h (the MI355X host mirror over the C ABI of
// include/phantom_amd.h); this file only gives it the reference's file name, so that `#include "rns_bconv.cuh"` (with
// -I include/phantom) and `#include <phantom/rns_bconv.cuh>` (with -I include) resolve as they do against the reference.
#pragma once
#include "../../phantom-fhe_amd/host/phantom.h"
