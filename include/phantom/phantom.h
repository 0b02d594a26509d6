// phantom.h -- stands in for the reference's include/phantom.h (installed as include/phantom/phantom.h, CMakeLists.txt:67-70):
// the umbrella header.
// The declarations live in one header, phantom-fhe_amd/host/phantom.h (the MI355X host mirror over the C ABI of
// include/phantom_amd.h); this file only gives it the reference's file name, so that `#include "phantom.h"` (with
// -I include/phantom) and `#include <phantom/phantom.h>` (with -I include) resolve as they do against the reference.
#pragma once
#include "../../phantom-fhe_amd/host/phantom.h"
