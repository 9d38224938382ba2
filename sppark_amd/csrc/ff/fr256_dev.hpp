// 256-bit scalar fields (BLS12-381 Fr, alt_bn128 Fr) as NTT element types: the
// Montgomery class of mont_dev.hpp plus the interface ntt_kernels.hpp expects
// (the reference's "wide" kernels, ntt/kernels/{ct,gs}_mixed_radix_wide.cu, on
// fr_t = mont_t<255|254,...>, ff/bls12-381.hpp:75-83, ff/alt_bn128.hpp:68-82).
// Wire format: 8 x u32 little-endian, Montgomery form with R = 2^256.
#pragma once
#include "mont_dev.hpp"

namespace sppark_amd {

template<class P> struct alignas(16) fr256_dev : mont_dev<P> {
    typedef mont_dev<P> base;
    static constexpr unsigned TWO_ADICITY = P::TWO_ADICITY;
    static constexpr bool SHIFT_ROOTS = false;

    fr256_dev() = default;
    __host__ __device__ fr256_dev(const base& b) : base(b) {}
    SPPARK_DEVFN static fr256_dev one() { return base::one(); }
    SPPARK_DEVFN friend fr256_dev operator+(const fr256_dev& a, const fr256_dev& b)
    {   return static_cast<const base&>(a) + static_cast<const base&>(b);   }
    SPPARK_DEVFN friend fr256_dev operator-(const fr256_dev& a, const fr256_dev& b)
    {   return static_cast<const base&>(a) - static_cast<const base&>(b);   }
    SPPARK_DEVFN friend fr256_dev operator*(const fr256_dev& a, const fr256_dev& b)
    {   return static_cast<const base&>(a) * static_cast<const base&>(b);   }
    // x * w_{2^R}^k from the per-(size, direction) table inner[(1 << R) + k]
    template<bool INV>
    SPPARK_DEVFN static fr256_dev mul_root(const fr256_dev& x, unsigned R, unsigned k, const fr256_dev* inner)
    {   return k ? x * inner[(1u << R) + k] : x;   }
    template<bool INV>
    SPPARK_DEVFN static constexpr bool root_neg(unsigned, unsigned) { return false; }
    SPPARK_DEVFN static void bfly(const fr256_dev& a, const fr256_dev& b, fr256_dev& s, fr256_dev& d)
    {   fr256_dev t = a - b; s = a + b; d = t;   }
};

} // namespace sppark_amd
