// secretkey.h -- stands in for the reference's include/secretkey.h:102-220 (installed as include/phantom/secretkey.h, CMakeLists.txt:67-70):
// PhantomRelinKey, PhantomGaloisKey (key layout [dnum][2][#QP][N] + device pointer table).
// The declarations live in one header, phantom-fhe_amd/host/phantom.h (the MI355X host mirror over the C ABI of
// include/phantom_amd.h); this file only gives it the reference's file name, so that `#include "secretkey.h"` (with
// -I include/phantom) and `#include <phantom/secretkey.h>` (with -I include) resolve as they do against the reference.
#pragma once
#include "../../phantom-fhe_amd/host/phantom.h"
