// C-ABI of a polynomial-primitives-only library (-DFEATURE_MERSENNE31 / -DFEATURE_BABY_BEAR_X4):
// fields the reference defines as types only (ff/mersenne31.hpp, bb31_4_t in ff/baby_bear.hpp) with
// no NTT parameter set; the generic primitives of polynomial/*.cuh are their natural entry points.
#include "../ntt/field_select.hpp"
#include "common_api.hpp"

using namespace sppark_amd;
typedef ntt_fr_t fr_t;

template<class Fn> static RustError guarded(Fn&& fn)
{
    try { fn(); return rust_ok(); }
    catch (const hip_error& e) { (void)hipGetLastError(); return rust_err(e.code(), e.what()); }
    catch (const std::exception& e) { return rust_err(-1, e.what()); }
    catch (...) { return rust_err(-1, "unknown exception"); }
}

#include "poly_api.hpp"
