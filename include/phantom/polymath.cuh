// polymath.cuh -- stands in for the reference's include/polymath.cuh:6-307 (installed as include/phantom/polymath.cuh, CMakeLists.txt:67-70):
// the residue-wise kernels as host launchers (first argument: the table handle instead of <<<grid, block>>>).
// The declarations live in one header, phantom-fhe_amd/host/phantom.h (the MI355X host mirror over the C ABI of
// include/phantom_amd.h); this file only gives it the reference's file name, so that `#include "polymath.cuh"` (with
// -I include/phantom) and `#include <phantom/polymath.cuh>` (with -I include) resolve as they do against the reference.
#pragma once
#include "../../phantom-fhe_amd/host/phantom.h"
