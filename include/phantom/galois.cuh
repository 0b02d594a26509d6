// galois.cuh -- stands in for the reference's include/galois.cuh:16-131 (installed as include/phantom/galois.cuh, CMakeLists.txt:67-70):
// get_elt_from_step / get_elts_from_steps and the Galois entry points of evaluate.*.
// The declarations live in one header, phantom-fhe_amd/host/phantom.h (the MI355X host mirror over the C ABI of
// include/phantom_amd.h); this file only gives it the reference's file name, so that `#include "galois.cuh"` (with
// -I include/phantom) and `#include <phantom/galois.cuh>` (with -I include) resolve as they do against the reference.
#pragma once
#include "../../phantom-fhe_amd/host/phantom.h"
