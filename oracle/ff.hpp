// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under sppark_amd/ may include,
// link or execute this file; only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg use it (as the checker, never as the product).
//
// CPU restatement of the reference's field layer for the MSM/NTT hot path.
//
//  * mont_t<P>   : N-bit Montgomery field on 64-bit limbs.  Restates the
//                  semantics the reference gets from blst's blst_384_t /
//                  blst_256_t (un-vendored dependency, blst v0.3.16 per
//                  /root/reference/go.mod:5; call sites
//                  ff/bls12-381.hpp:90-138, ff/alt_bn128.hpp:86-131) and that
//                  the device class ff/mont_t.cuh:33-1217 implements with
//                  32-bit limbs.  Algorithm here: schoolbook product followed
//                  by word-serial Montgomery reduction (SOS form), chosen to
//                  be *different* from the product's CIOS/FIPS device code so
//                  that the two do not share bugs.  Residues mod p are unique,
//                  so any correct implementation yields identical bytes.
//  * gl64        : Goldilocks p = 2^64-2^32+1, canonical u64, non-Montgomery
//                  (ff/gl64_t.cuh:39-587).
//  * bb31        : BabyBear p = 0x78000001, Montgomery R = 2^32
//                  (ff/mont32_t.cuh:19-425, ff/baby_bear.hpp:19).
//
// PINNED (round 5): tests/test_oracle.py::test_oracle_field_equals_the_reference_device_field holds mont_t<P> -- base and
// scalar field of all five curves, + - * sqr to from -- against outputs of the reference's OWN device field classes
// (fp_t / fr_t over ff/mont_t.hip, ff/bls12-381.hpp:63-83 ...), recorded on an MI355X by tests/golden/make_ref_field_golden.py
// into tests/golden/ref_field_golden.json.  The layers above the field (ec.hpp, msm.hpp) are pinned as DESIGN.md section 2 says.
//
// Constants (modulus, R^2, R, -1/p mod 2^64) are the reference's tables,
// cited at each definition, and are re-derived from Python big-ints by
// tests/test_oracle_fields.py.
#pragma once
#include <cstdint>
#include <cstddef>
#include <cstring>

namespace oracle {

typedef unsigned __int128 u128;

// ---------------------------------------------------------------------------
// Generic Montgomery field, P supplies: N (64-bit limbs), NBITS, MOD, RR, ONE, M0
// ---------------------------------------------------------------------------
template<class P> class mont_t {
public:
    static const size_t N = P::N;
    static const size_t nbits = P::NBITS;
    static const unsigned int degree = 1;
    using mem_t = mont_t;
    typedef unsigned char pow_t[(P::NBITS + 7) / 8];

    uint64_t v[N];

    mont_t() = default;

    static const uint64_t* modulus() { return P::MOD; }

    static mont_t from_limbs(const uint64_t* p)
    {   mont_t r; for (size_t i = 0; i < N; i++) r.v[i] = p[i]; return r;   }

    // one(or_zero): R mod p, or zero when the flag is set (ec/xyzz_t.hpp:24-26
    // uses one(is_inf) to build ZZZ/ZZ of a point at infinity).
    static mont_t one(bool or_zero = false)
    {
        mont_t r;
        for (size_t i = 0; i < N; i++) r.v[i] = or_zero ? 0 : P::ONE[i];
        return r;
    }

    bool is_zero() const
    {   uint64_t acc = 0; for (size_t i = 0; i < N; i++) acc |= v[i]; return acc == 0;   }
    void zero() { for (size_t i = 0; i < N; i++) v[i] = 0; }

    friend bool operator==(const mont_t& a, const mont_t& b)
    {   uint64_t acc = 0; for (size_t i = 0; i < N; i++) acc |= a.v[i] ^ b.v[i]; return acc == 0;   }
    friend bool operator!=(const mont_t& a, const mont_t& b) { return !(a == b); }

private:
    // r = a - MOD if a >= MOD (carry = bit above the top limb)
    static void final_sub(uint64_t r[], const uint64_t a[], uint64_t carry)
    {
        uint64_t t[N], borrow = 0;
        for (size_t i = 0; i < N; i++) {
            u128 d = (u128)a[i] - P::MOD[i] - borrow;
            t[i] = (uint64_t)d;
            borrow = (uint64_t)(d >> 64) & 1;
        }
        bool ge = carry || !borrow;
        for (size_t i = 0; i < N; i++) r[i] = ge ? t[i] : a[i];
    }

public:
    mont_t& operator+=(const mont_t& b)
    {
        uint64_t t[N], c = 0;
        for (size_t i = 0; i < N; i++) {
            u128 s = (u128)v[i] + b.v[i] + c;
            t[i] = (uint64_t)s; c = (uint64_t)(s >> 64);
        }
        final_sub(v, t, c);
        return *this;
    }
    mont_t& operator-=(const mont_t& b)
    {
        uint64_t borrow = 0;
        for (size_t i = 0; i < N; i++) {
            u128 d = (u128)v[i] - b.v[i] - borrow;
            v[i] = (uint64_t)d; borrow = (uint64_t)(d >> 64) & 1;
        }
        if (borrow) {
            uint64_t c = 0;
            for (size_t i = 0; i < N; i++) {
                u128 s = (u128)v[i] + P::MOD[i] + c;
                v[i] = (uint64_t)s; c = (uint64_t)(s >> 64);
            }
        }
        return *this;
    }
    friend mont_t operator+(mont_t a, const mont_t& b) { return a += b; }
    friend mont_t operator-(mont_t a, const mont_t& b) { return a -= b; }

    // Montgomery product a*b/R mod p.
    mont_t& operator*=(const mont_t& b)
    {
        uint64_t t[2*N + 1];
        for (size_t i = 0; i < 2*N + 1; i++) t[i] = 0;
        for (size_t i = 0; i < N; i++) {
            uint64_t c = 0;
            for (size_t j = 0; j < N; j++) {
                u128 s = (u128)v[i] * b.v[j] + t[i+j] + c;
                t[i+j] = (uint64_t)s; c = (uint64_t)(s >> 64);
            }
            t[i+N] = c;
        }
        for (size_t i = 0; i < N; i++) {
            uint64_t m = t[i] * P::M0, c = 0;
            for (size_t j = 0; j < N; j++) {
                u128 s = (u128)m * P::MOD[j] + t[i+j] + c;
                t[i+j] = (uint64_t)s; c = (uint64_t)(s >> 64);
            }
            for (size_t k = i + N; c != 0 && k < 2*N + 1; k++) {
                u128 s = (u128)t[k] + c;
                t[k] = (uint64_t)s; c = (uint64_t)(s >> 64);
            }
        }
        final_sub(v, &t[N], t[2*N]);
        return *this;
    }
    friend mont_t operator*(mont_t a, const mont_t& b) { return a *= b; }

    // a^2 is spelled `a^2` in the reference templates (ec/xyzz_t.hpp:139).
    friend mont_t operator^(mont_t a, int p)
    {   if (p != 2) __builtin_trap(); a *= mont_t(a); return a;   }
    mont_t& operator^=(int p)
    {   if (p != 2) __builtin_trap(); mont_t t = *this; return *this *= t;   }

    mont_t& operator<<=(unsigned l)
    {   while (l--) { mont_t t = *this; *this += t; } return *this;   }
    friend mont_t operator<<(mont_t a, unsigned l) { return a <<= l; }

    mont_t& cneg(bool flag)
    {
        if (flag && !is_zero()) {
            mont_t z; z.zero(); z -= *this; *this = z;
        }
        return *this;
    }
    friend mont_t czero(const mont_t& a, int set_z)
    {   mont_t r = a; if (set_z) r.zero(); return r;   }

    // to Montgomery form: a*RR/R ; from: a*1/R
    mont_t& to()   { return *this *= from_limbs(P::RR); }
    mont_t& from()
    {   mont_t o; o.zero(); o.v[0] = 1; return *this *= o;   }

    // from-Montgomery, little-endian bytes (msm/pippenger.hpp:236-241 calls
    // scalar_t::to_scalar on Montgomery-form scalars).
    void to_scalar(pow_t& out) const
    {
        mont_t t = *this; t.from();
        for (size_t i = 0; i < sizeof(pow_t); i++)
            out[i] = (unsigned char)(t.v[i/8] >> (8*(i%8)));
    }

    // Fermat inversion a^(p-2); 1/0 = 0.
    mont_t reciprocal() const
    {
        uint64_t e[N]; uint64_t borrow = 2;
        for (size_t i = 0; i < N; i++) {
            u128 d = (u128)P::MOD[i] - borrow;
            e[i] = (uint64_t)d; borrow = (uint64_t)(d >> 64) & 1;
        }
        mont_t r = one(), b = *this;
        for (size_t i = 0; i < 64*N; i++) {
            if ((e[i/64] >> (i%64)) & 1) r *= b;
            b ^= 2;
        }
        return r;
    }
    friend mont_t operator/(int one_, const mont_t& a)
    {   if (one_ != 1) __builtin_trap(); return a.reciprocal();   }

    // NTT conventions of ntt/parameters/{bls12_381,alt_bn128}.h:11-14: group_gen
    // (7 resp. 5) and forward_roots_of_unity[S] = group_gen^((r-1)/2^S)
    // (checked against the reference tables by tests/test_oracle.py).
    static const unsigned TWO_ADICITY = P::TWO_ADICITY;
    static mont_t group_gen()
    {   mont_t g; g.zero(); for (unsigned i = 0; i < P::GEN; i++) g += one(); return g;   }
    static mont_t top_root()
    {
        uint64_t e[N];
        for (size_t i = 0; i < N; i++) e[i] = P::MOD[i];
        e[0] -= 1;
        mont_t r = one(), b = group_gen();
        for (size_t bit = P::TWO_ADICITY; bit < 64 * N; bit++) {
            if ((e[bit / 64] >> (bit % 64)) & 1) r *= b;
            b ^= 2;
        }
        return r;
    }
};

// ---------------------------------------------------------------------------
// Parameter sets.  TO_LIMB values are the reference's tables.
// ---------------------------------------------------------------------------
struct bls12_381_fp_params {
    static const unsigned TWO_ADICITY = 0, GEN = 0;   // NTT domain (scalar fields only)        // ff/bls12-381.hpp:100-116
    static const size_t N = 6, NBITS = 381;
    static constexpr uint64_t MOD[6] = {
        0xb9feffffffffaaab, 0x1eabfffeb153ffff, 0x6730d2a0f6b0f624,
        0x64774b84f38512bf, 0x4b1ba7b6434bacd7, 0x1a0111ea397fe69a };
    static constexpr uint64_t RR[6] = {
        0xf4df1f341c341746, 0x0a76e6a609d104f1, 0x8de5476c4c95b6d5,
        0x67eb88a9939d83c0, 0x9a793e85b519952d, 0x11988fe592cae3aa };
    static constexpr uint64_t ONE[6] = {
        0x760900000002fffd, 0xebf4000bc40c0002, 0x5f48985753c758ba,
        0x77ce585370525745, 0x5c071a97a256ec6d, 0x15f65ec3fa80e493 };
    static const uint64_t M0 = 0x89f3fffcfffcfffd;
};
struct bls12_381_fr_params {
    static const unsigned TWO_ADICITY = 32, GEN = 7;   // NTT domain (scalar fields only)        // ff/bls12-381.hpp:125-138
    static const size_t N = 4, NBITS = 255;
    static constexpr uint64_t MOD[4] = {
        0xffffffff00000001, 0x53bda402fffe5bfe, 0x3339d80809a1d805, 0x73eda753299d7d48 };
    static constexpr uint64_t RR[4] = {
        0xc999e990f3f29c6d, 0x2b6cedcb87925c23, 0x05d314967254398f, 0x0748d9d99f59ff11 };
    static constexpr uint64_t ONE[4] = {
        0x00000001fffffffe, 0x5884b7fa00034802, 0x998c4fefecbc4ff5, 0x1824b159acc5056f };
    static const uint64_t M0 = 0xfffffffeffffffff;
};
struct alt_bn128_fp_params {
    static const unsigned TWO_ADICITY = 0, GEN = 0;   // NTT domain (scalar fields only)        // ff/alt_bn128.hpp:88-101
    static const size_t N = 4, NBITS = 254;
    static constexpr uint64_t MOD[4] = {
        0x3c208c16d87cfd47, 0x97816a916871ca8d, 0xb85045b68181585d, 0x30644e72e131a029 };
    static constexpr uint64_t RR[4] = {
        0xf32cfc5b538afa89, 0xb5e71911d44501fb, 0x47ab1eff0a417ff6, 0x06d89f71cab8351f };
    static constexpr uint64_t ONE[4] = {
        0xd35d438dc58f0d9d, 0x0a78eb28f5c70b3d, 0x666ea36f7879462c, 0x0e0a77c19a07df2f };
    static const uint64_t M0 = 0x87d20782e4866389;
};
struct alt_bn128_fr_params {
    static const unsigned TWO_ADICITY = 28, GEN = 5;   // NTT domain (scalar fields only)        // ff/alt_bn128.hpp:111-124
    static const size_t N = 4, NBITS = 254;
    static constexpr uint64_t MOD[4] = {
        0x43e1f593f0000001, 0x2833e84879b97091, 0xb85045b68181585d, 0x30644e72e131a029 };
    static constexpr uint64_t RR[4] = {
        0x1bb8e645ae216da7, 0x53fe3ab1e35c59e3, 0x8c49833d53bb8085, 0x0216d0b17f4e44a5 };
    static constexpr uint64_t ONE[4] = {
        0xac96341c4ffffffb, 0x36fc76959f60cd29, 0x666ea36f7879462e, 0x0e0a77c19a07df2f };
    static const uint64_t M0 = 0xc2e1f593efffffff;
};

struct bls12_377_fp_params {
    static const unsigned TWO_ADICITY = 0, GEN = 0;   // NTT domain (scalar fields only)        // ff/bls12-377.hpp:13-33
    static const size_t N = 6, NBITS = 377;
    static const unsigned FP2_NR = 5;                  // Fp2 = Fp[u]/(u^2 + 5), ff/bls12-377-fp2.hpp
    static constexpr uint64_t MOD[6] = {
        0x8508c00000000001, 0x170b5d4430000000, 0x1ef3622fba094800, 0x1a22d9f300f5138f, 0xc63b05c06ca1493b, 0x01ae3a4617c510ea };
    static constexpr uint64_t RR[6] = {
        0xb786686c9400cd22, 0x0329fcaab00431b1, 0x22a5f11162d6b46d, 0xbfdf7d03827dc3ac, 0x837e92f041790bf9, 0x006dfccb1e914b88 };
    static constexpr uint64_t ONE[6] = {
        0x02cdffffffffff68, 0x51409f837fffffb1, 0x9f7db3a98a7d3ff2, 0x7b4e97b76e7c6305, 0x4cf495bf803c84e8, 0x008d6661e2fdf49a };
    static const uint64_t M0 = 0x8508bfffffffffff;
};
struct bls12_377_fr_params {
    static const unsigned TWO_ADICITY = 47, GEN = 22;   // NTT domain (scalar fields only)        // ff/bls12-377.hpp:35-51, ntt/parameters/bls12_377.h:11-16
    static const size_t N = 4, NBITS = 253;
    static constexpr uint64_t MOD[4] = {
        0x0a11800000000001, 0x59aa76fed0000001, 0x60b44d1e5c37b001, 0x12ab655e9a2ca556 };
    static constexpr uint64_t RR[4] = {
        0x25d577bab861857b, 0xcc2c27b58860591f, 0xa7cc008fe5dc8593, 0x011fdae7eff1c939 };
    static constexpr uint64_t ONE[4] = {
        0x7d1c7ffffffffff3, 0x7257f50f6ffffff2, 0x16d81575512c0fee, 0x0d4bda322bbb9a9d };
    static const uint64_t M0 = 0x0a117fffffffffff;
};

struct pasta_p_params {
    static const unsigned TWO_ADICITY = 32, GEN = 5;   // the Pallas base field = the Vesta scalar field (ff/pasta.hpp: Pallas_P; roots: ntt/parameters/pallas.h)
    static const size_t N = 4, NBITS = 255;
    static constexpr uint64_t MOD[4] = {
        0x992d30ed00000001, 0x224698fc094cf91b, 0x0000000000000000, 0x4000000000000000 };
    static constexpr uint64_t RR[4] = {
        0x8c78ecb30000000f, 0xd7d30dbd8b0de0e7, 0x7797a99bc3c95d18, 0x096d41af7b9cb714 };
    static constexpr uint64_t ONE[4] = {
        0x34786d38fffffffd, 0x992c350be41914ad, 0xffffffffffffffff, 0x3fffffffffffffff };
    static const uint64_t M0 = 0x992d30ecffffffff;
};
struct pasta_q_params {
    static const unsigned TWO_ADICITY = 32, GEN = 5;   // the Vesta base field = the Pallas scalar field (ff/pasta.hpp: Vesta_P; roots: ntt/parameters/vesta.h)
    static const size_t N = 4, NBITS = 255;
    static constexpr uint64_t MOD[4] = {
        0x8c46eb2100000001, 0x224698fc0994a8dd, 0x0000000000000000, 0x4000000000000000 };
    static constexpr uint64_t RR[4] = {
        0xfc9678ff0000000f, 0x67bb433d891a16e3, 0x7fae231004ccf590, 0x096d41af7ccfdaa9 };
    static constexpr uint64_t ONE[4] = {
        0x5b2b3e9cfffffffd, 0x992c350be3420567, 0xffffffffffffffff, 0x3fffffffffffffff };
    static const uint64_t M0 = 0x8c46eb20ffffffff;
};

// ---------------------------------------------------------------------------
// Goldilocks (ff/gl64_t.cuh:39-587): canonical residues in a u64.
// ---------------------------------------------------------------------------
struct gl64 {
    static const uint64_t MOD = 0xffffffff00000001ULL;
    uint64_t v;
    gl64() = default;
    explicit gl64(uint64_t x) : v(x % MOD) {}
    static gl64 one() { return gl64(1); }
    friend gl64 operator+(gl64 a, gl64 b)
    {   u128 s = (u128)a.v + b.v; if (s >= MOD) s -= MOD; gl64 r; r.v = (uint64_t)s; return r;   }
    friend gl64 operator-(gl64 a, gl64 b)
    {   gl64 r; r.v = a.v >= b.v ? a.v - b.v : a.v + (MOD - b.v); return r;   }
    friend gl64 operator*(gl64 a, gl64 b)
    {   gl64 r; r.v = (uint64_t)(((u128)a.v * b.v) % MOD); return r;   }
    gl64& operator*=(gl64 b) { return *this = *this * b; }
    friend bool operator==(gl64 a, gl64 b) { return a.v == b.v; }
    uint64_t raw() const { return v; }
    static gl64 from_raw(uint64_t x) { gl64 r; r.v = x; return r; }
    // reference NTT root tables, ntt/parameters/goldilocks.h:84-159 (default
    // "canonical" generator branch): group_gen = 7, the 2^32-th root below;
    // every other entry of forward_roots_of_unity[] is a repeated square of it.
    // plonky2() = true selects the -DGOLDILOCKS_PLONKY2 branch (goldilocks.h:7-82):
    // group_gen = 0xc65c18b67785d900, forward_roots_of_unity[32] = 0x64fdd1a46201e246.
    static const unsigned TWO_ADICITY = 32;
    static bool& plonky2()    { static bool on = false; return on; }
    static gl64 group_gen()   { return plonky2() ? gl64(0xc65c18b67785d900ULL) : gl64(7); }
    static gl64 top_root()    { return plonky2() ? gl64(0x64fdd1a46201e246ULL) : gl64(0x185629dcda58878cULL); }
};

// ---------------------------------------------------------------------------
// BabyBear (ff/baby_bear.hpp:19; ff/mont32_t.cuh): Montgomery, R = 2^32.
// Wire format = Montgomery residues (poc/ntt-cuda/tests/ntt.rs:50-53).
// ---------------------------------------------------------------------------
struct bb31 {
    static const uint32_t MOD = 0x78000001u;
    static const uint32_t M   = 0x77ffffffu;     // -1/MOD mod 2^32
    static const uint32_t RR  = 0x45dddde3u;     // 2^64 mod MOD
    static const uint32_t ONE = 0x0ffffffeu;     // 2^32 mod MOD
    uint32_t v;                                  // Montgomery residue
    bb31() = default;
    static bb31 from_raw(uint32_t x) { bb31 r; r.v = x; return r; }
    static bb31 from_canonical(uint32_t x)
    {   bb31 r; r.v = (uint32_t)((((uint64_t)(x % MOD)) << 32) % MOD); return r;   }
    uint32_t to_canonical() const
    {   bb31 o; o.v = 1; return (*this * o).v;   }
    static bb31 one() { return from_raw(ONE); }
    uint32_t raw() const { return v; }
    friend bb31 operator+(bb31 a, bb31 b)
    {   uint64_t s = (uint64_t)a.v + b.v; if (s >= MOD) s -= MOD; return from_raw((uint32_t)s);   }
    friend bb31 operator-(bb31 a, bb31 b)
    {   return from_raw(a.v >= b.v ? a.v - b.v : a.v + (MOD - b.v));   }
    friend bb31 operator*(bb31 a, bb31 b)
    {
        uint64_t t = (uint64_t)a.v * b.v;
        uint32_t m = (uint32_t)t * M;
        uint64_t u = (t + (uint64_t)m * MOD) >> 32;       // < 2*MOD
        if (u >= MOD) u -= MOD;
        return from_raw((uint32_t)u);
    }
    bb31& operator*=(bb31 b) { return *this = *this * b; }
    friend bool operator==(bb31 a, bb31 b) { return a.v == b.v; }
    // ntt/parameters/baby_bear.h:76-175 (default branch): group_gen = 3,
    // forward_roots_of_unity[27] = 0x1ffffedc (Montgomery).
    // canonical_roots() = true selects the -DBABY_BEAR_CANONICAL branch (baby_bear.h:7-74):
    // group_gen = 31 (the smallest primitive root), forward_roots_of_unity[27] = 0x57fab6ee.
    static const unsigned TWO_ADICITY = 27;
    static bool& canonical_roots() { static bool on = false; return on; }
    static bb31 group_gen()   { return from_canonical(canonical_roots() ? 31 : 3); }
    static bb31 top_root()    { return from_raw(canonical_roots() ? 0x57fab6eeu : 0x1ffffedcu); }
};

// ---------------------------------------------------------------------------
// Mersenne31 (ff/mersenne31.hpp:13-60), p = 2^31 - 1: canonical residues, the reference's memory form
// (mrs31_t::mem_t); and the BabyBear quartic extension F_p[x]/(x^4 - beta), beta = -11 (default) or
// +11 (-DBABY_BEAR_CANONICAL) (bb31_4_t, ff/baby_bear.hpp:70-446): four Montgomery residues.
// Field types only in the reference (no NTT parameters); here: operands of the polynomial primitives.
// ---------------------------------------------------------------------------
struct mrs31 {
    static const uint32_t MOD = 0x7fffffffu;
    uint32_t v;
    mrs31() = default;
    static mrs31 from_raw(uint32_t x) { mrs31 r; r.v = x; return r; }
    static mrs31 one() { return from_raw(1); }
    friend mrs31 operator+(mrs31 a, mrs31 b) { return from_raw((uint32_t)(((uint64_t)a.v + b.v) % MOD)); }
    friend mrs31 operator-(mrs31 a, mrs31 b) { return from_raw((uint32_t)(((uint64_t)a.v + MOD - b.v) % MOD)); }
    friend mrs31 operator*(mrs31 a, mrs31 b) { return from_raw((uint32_t)(((uint64_t)a.v * b.v) % MOD)); }
};
struct bb31_4 {
    bb31 c[4];
    static bool& canonical_beta() { static bool on = false; return on; }        // -DBABY_BEAR_CANONICAL
    static bb31 beta() { return canonical_beta() ? bb31::from_canonical(11) : bb31::from_canonical(bb31::MOD - 11); }
    static bb31_4 one() { bb31_4 r; r.c[0] = bb31::one(); r.c[1] = r.c[2] = r.c[3] = bb31::from_raw(0); return r; }
    friend bb31_4 operator+(const bb31_4& a, const bb31_4& b) { bb31_4 r; for (int i = 0; i < 4; i++) r.c[i] = a.c[i] + b.c[i]; return r; }
    friend bb31_4 operator-(const bb31_4& a, const bb31_4& b) { bb31_4 r; for (int i = 0; i < 4; i++) r.c[i] = a.c[i] - b.c[i]; return r; }
    friend bb31_4 operator*(const bb31_4& a, const bb31_4& b)
    {
        bb31 t[7];
        for (int k = 0; k < 7; k++) t[k] = bb31::from_raw(0);
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) t[i + j] = t[i + j] + a.c[i] * b.c[j];
        bb31_4 r;
        for (int k = 0; k < 3; k++) r.c[k] = t[k] + beta() * t[k + 4];
        r.c[3] = t[3];
        return r;
    }
};

template<class F> static inline F fpow(F b, uint64_t e)
{
    F r = F::one();
    while (e) { if (e & 1) r *= b; b *= b; e >>= 1; }
    return r;
}

// ---------------------------------------------------------------------------
// Fp2 = Fp[u]/(u^2 + 1): coordinate field of G2 for BLS12-381 and alt_bn128.  Memory image
// c0 | c1 (fp_mont x[2], ff/bls12-381-fp2.hpp:33-34).  The reference's host-side Fp2 is blst's
// (not vendored); this class offers the same interface mont_t does, so that the reference's
// point templates (and this oracle's) instantiate over it.  Schoolbook products on purpose
// (the device code uses Karatsuba).
// Fp2 = Fp[u]/(u^2 + NR): NR = 1 for BLS12-381 and alt_bn128 (ff/bls12-381-fp2.hpp, ff/alt_bn128-fp2.hpp),
// 5 for BLS12-377 (ff/bls12-377-fp2.hpp)
template<class FP> struct fp2_nonresidue { static const unsigned value = 1; };
template<class FP> class fp2_t {
    static FP mul_nr(const FP& x) { FP r = x; for (unsigned k = 1; k < fp2_nonresidue<FP>::value; k++) r += x; return r; }
public:
    static const unsigned int degree = 2;
    using mem_t = fp2_t;
    FP c0, c1;

    fp2_t() = default;
    fp2_t(const FP& a, const FP& b) : c0(a), c1(b) {}
    static fp2_t one(bool or_zero = false) { fp2_t r; r.c0 = FP::one(or_zero); r.c1.zero(); return r; }
    bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    void zero() { c0.zero(); c1.zero(); }
    friend bool operator==(const fp2_t& a, const fp2_t& b) { return a.c0 == b.c0 && a.c1 == b.c1; }
    friend bool operator!=(const fp2_t& a, const fp2_t& b) { return !(a == b); }

    fp2_t& operator+=(const fp2_t& b) { c0 += b.c0; c1 += b.c1; return *this; }
    fp2_t& operator-=(const fp2_t& b) { c0 -= b.c0; c1 -= b.c1; return *this; }
    fp2_t& operator*=(const fp2_t& b)
    {
        FP r0 = c0 * b.c0 - mul_nr(c1 * b.c1), r1 = c0 * b.c1 + c1 * b.c0;
        c0 = r0; c1 = r1;
        return *this;
    }
    friend fp2_t operator+(fp2_t a, const fp2_t& b) { return a += b; }
    friend fp2_t operator-(fp2_t a, const fp2_t& b) { return a -= b; }
    friend fp2_t operator*(fp2_t a, const fp2_t& b) { return a *= b; }
    fp2_t& operator^=(int p) { (void)p; return *this *= *this; }       // only ^2 is used
    friend fp2_t operator^(fp2_t a, int p) { return a ^= p; }
    fp2_t& operator<<=(unsigned l) { c0 <<= l; c1 <<= l; return *this; }
    friend fp2_t operator<<(fp2_t a, unsigned l) { return a <<= l; }
    fp2_t& cneg(bool flag) { c0.cneg(flag); c1.cneg(flag); return *this; }
    friend fp2_t czero(const fp2_t& a, int set_z) { fp2_t r; r.c0 = czero(a.c0, set_z); r.c1 = czero(a.c1, set_z); return r; }
    // 1/(a0 + a1 u) = (a0 - a1 u)/(a0^2 + NR a1^2)
    fp2_t reciprocal() const
    {
        FP n = (c0 * c0 + mul_nr(c1 * c1)).reciprocal();
        fp2_t r; r.c0 = c0 * n; r.c1 = c1 * n; r.c1.cneg(true);
        return r;
    }
    friend fp2_t operator/(int one_, const fp2_t& a) { (void)one_; return a.reciprocal(); }
};

typedef mont_t<bls12_381_fp_params> bls12_381_fp;
typedef mont_t<bls12_381_fr_params> bls12_381_fr;
typedef mont_t<alt_bn128_fp_params> alt_bn128_fp;
typedef mont_t<alt_bn128_fr_params> alt_bn128_fr;
typedef mont_t<bls12_377_fp_params> bls12_377_fp;
typedef mont_t<bls12_377_fr_params> bls12_377_fr;
template<> struct fp2_nonresidue<bls12_377_fp> { static const unsigned value = 5; };
typedef fp2_t<bls12_377_fp> bls12_377_fp2;
typedef mont_t<pasta_p_params> pasta_p;          // Pallas base field = Vesta scalar field
typedef mont_t<pasta_q_params> pasta_q;          // Vesta base field = Pallas scalar field
typedef fp2_t<bls12_381_fp> bls12_381_fp2;
typedef fp2_t<alt_bn128_fp> alt_bn128_fp2;

} // namespace oracle
