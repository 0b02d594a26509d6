// host/encryptionparams.h -- stands in for the reference's include/host/encryptionparams.h:19-246 (installed as include/phantom/host/encryptionparams.h, CMakeLists.txt:67-70):
// scheme_type, mul_tech_type, EncryptionParameters.
// The declarations live in one header, phantom-fhe_amd/host/phantom.h (the MI355X host mirror over the C ABI of
// include/phantom_amd.h); this file only gives it the reference's file name, so that `#include "host/encryptionparams.h"` (with
// -I include/phantom) and `#include <phantom/host/encryptionparams.h>` (with -I include) resolve as they do against the reference.
#pragma once
#include "../../../phantom-fhe_amd/host/phantom.h"
