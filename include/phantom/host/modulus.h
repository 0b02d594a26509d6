// host/modulus.h -- stands in for the reference's include/host/modulus.h:29-330 (installed as include/phantom/host/modulus.h, CMakeLists.txt:67-70):
// arith::Modulus, CoeffModulus::{MaxBitCount, BFVDefault, Create}, PlainModulus::Batching, sec_level_type.
// The declarations live in one header, phantom-fhe_amd/host/phantom.h (the MI355X host mirror over the C ABI of
// include/phantom_amd.h); this file only gives it the reference's file name, so that `#include "host/modulus.h"` (with
// -I include/phantom) and `#include <phantom/host/modulus.h>` (with -I include) resolve as they do against the reference.
#pragma once
#include "../../../phantom-fhe_amd/host/phantom.h"
