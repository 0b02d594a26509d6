// rns.cuh -- stands in for the reference's include/rns.cuh:13-236 (installed as include/phantom/rns.cuh, CMakeLists.txt:67-70):
// phantom::DRNSTool (handle): modup / moddown_from_NTT / divide_and_round_q_last(_ntt) / mod_t_and_divide_q_last_ntt / HPS leveled scaling.
// The declarations live in one header, phantom-fhe_amd/host/phantom.h (the MI355X host mirror over the C ABI of
// include/phantom_amd.h); this file only gives it the reference's file name, so that `#include "rns.cuh"` (with
// -I include/phantom) and `#include <phantom/rns.cuh>` (with -I include) resolve as they do against the reference.
#pragma once
#include "../../phantom-fhe_amd/host/phantom.h"
