// ciphertext.h -- stands in for the reference's include/ciphertext.h:7-214 (installed as include/phantom/ciphertext.h, CMakeLists.txt:67-70):
// PhantomCiphertext incl. save / load in the reference byte format.
// The declarations live in one header, phantom-fhe_amd/host/phantom.h (the MI355X host mirror over the C ABI of
// include/phantom_amd.h); this file only gives it the reference's file name, so that `#include "ciphertext.h"` (with
// -I include/phantom) and `#include <phantom/ciphertext.h>` (with -I include) resolve as they do against the reference.
#pragma once
#include "../../phantom-fhe_amd/host/phantom.h"
