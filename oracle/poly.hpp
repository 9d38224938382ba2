// ORACLE — TEST INFRASTRUCTURE ONLY.
// Sequential restatement of the observable semantics of the reference's polynomial primitives
// (device-only templates there; the reference has no CPU version and no tests for them, so the
// pin is the definition-level Python big-int vectors of tests/golden/make_golden.py):
//
//   prefix_op   polynomial/prefix_op.cuh:17-47 (Add / Multiply functors), :49-322 (inclusive
//               scan: every lane combines the value shuffled up from lane - offset WITH its own)
//   evaluate    polynomial/evaluate.cuh:307-412      ret[j] = sum_i coeffs[i] * x[j]^i
//   div_by_x_minus_z  polynomial/div_by_x_minus_z.cuh:132-157 (column picture and the meaning
//               of |rotate|), :447-486
#pragma once
#include <cstddef>

namespace oracle {

template<class F> static void prefix_op(F* out, const F* in, size_t len, int op)    // op 0: Add, 1: Multiply
{
    if (len == 0) return;
    F run = in[0];
    out[0] = run;
    for (size_t i = 1; i < len; i++) {
        run = op == 0 ? run + in[i] : run * in[i];
        out[i] = run;
    }
}

template<class F> static void poly_evaluate(F* ret, const F* x, size_t n, const F* coeffs, size_t len)
{
    for (size_t j = 0; j < n; j++) {
        F acc = coeffs[0] - coeffs[0];                      // zero, whatever the representation
        for (size_t i = len; i--;) acc = coeffs[i] + x[j] * acc;
        ret[j] = acc;
    }
}

// B_k = sum_{m >= k} c_m z^(m-k).  !rotate: inout[k] = B_k (remainder p(z) first, quotient after it);
// rotate: quotient moved to the front, remainder last.
template<class F> static void div_by_x_minus_z(F* inout, size_t len, F z, bool rotate)
{
    if (len == 0) return;
    for (size_t k = len - 1; k--;) inout[k] = inout[k] + z * inout[k + 1];
    if (rotate) {
        F rem = inout[0];
        for (size_t k = 1; k < len; k++) inout[k - 1] = inout[k];
        inout[len - 1] = rem;
    }
}

} // namespace oracle
