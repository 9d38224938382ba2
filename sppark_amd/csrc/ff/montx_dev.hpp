// Loosely-reduced, reduced-radix Montgomery field for the MSM bucket kernels (gfx950).
//
// Multiply-adds, multiplies, carry instructions and 64-bit adds/shifts of a wave64 all issue at the same
// rate on CDNA4 (only plain 32-bit adds / logic ops are ~1.5x cheaper: profiles/r02_ubench_instruction_rates.log),
// so the currency is the instruction COUNT.  With full 32-bit limbs (mont_dev.hpp) a 12-limb
// product is 288 mads + 288 carry adds + bookkeeping = ~680 instructions and a modular
// add/sub ~50.  Here (LB = 28 for a 381-bit modulus: NL = 14 limbs):
//
//   * a partial product of normalised limbs is < 2^56 and a whole column of a*b AND m*p
//     products (28 of them) fits ONE 64-bit accumulator: a column is a pure chain of
//     v_mad_u64_u32, no carry instructions; between columns the accumulator moves down by
//     LB bits with one 64-bit shift;
//   * the Montgomery radix is 2^(LB*NL) = 2^392, a factor ~2500 above the modulus: values
//     are only loosely reduced.  A product of inputs < ka*p and < kb*p is
//     < (ka*kb*p/R + 1)*p; a + b adds the bounds; a - b is a + K*p - b where K*p is written
//     with "fat" limbs that dominate b's limbs one by one.  No conditional subtraction and
//     no carry chain on the fast path of a point addition;
//   * the head-room inside the accumulator (2^64 / (28 * 2^56) = 2^3.2) lets the LEFT operand
//     of a product be un-normalised (limbs < 2^31, i.e. the direct result of a lazy add/sub);
//     norm() (3 instructions per limb) is needed only before squaring such a value or
//     multiplying two of them.
//
// Montgomery domain: x is held as x * 2^(LB*NL) mod p; the reference's wire form is
// x * 2^(32*N) (ff/bls12-381.hpp:13-33).  from_std()/to_std() convert (one product each).
// The limb/value bounds are the caller's contract and are stated at every use in
// ec/xyzzx_dev.hpp.
#pragma once
#include "mont_dev.hpp"

namespace sppark_amd {

// acc += a * b (64-bit accumulate; the carry-out is architecturally written but never set: the
// column bound keeps acc below 2^64).  Inline asm because hipcc, given the C expression,
// (i) strength-reduces products by the modulus limbs into shift/add sequences and (ii) splits
// long accumulation chains and re-joins them with 64-bit adds -- both cost more than they save
// (measured: 6.0e9 vs 7.1e9 mixed additions/s).  b in a VGPR / in an SGPR (modulus limbs).
SPPARK_DEVFN void macx(u64& acc, u32 a, u32 b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
#else
    acc += (u64)a * b;
#endif
}
SPPARK_DEVFN void macxs(u64& acc, u32 a, u32 b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "s"(b) : "vcc");
#else
    acc += (u64)a * b;
#endif
}
// two independent multiply-adds in ONE asm statement: hipcc pads every asm statement with an
// s_nop (it cannot see inside), so pairing halves the padding of the interleaved product pairs
SPPARK_DEVFN void macx2(u64& acc0, u32 a0, u32 b0, u64& acc1, u32 a1, u32 b1)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1"
        : "+v"(acc0), "+v"(acc1) : "v"(a0), "v"(b0), "v"(a1), "v"(b1) : "vcc");
#else
    acc0 += (u64)a0 * b0; acc1 += (u64)a1 * b1;
#endif
}
SPPARK_DEVFN void macxs2(u64& acc0, u32 a0, u64& acc1, u32 a1, u32 b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_mad_u64_u32 %0, vcc, %2, %4, %0\n\tv_mad_u64_u32 %1, vcc, %3, %4, %1"
        : "+v"(acc0), "+v"(acc1) : "v"(a0), "v"(a1), "s"(b) : "vcc");
#else
    acc0 += (u64)a0 * b; acc1 += (u64)a1 * b;
#endif
}

#if defined(SPPARK_HOST_EMULATION) && defined(SPPARK_TRACK_BOUNDS)
extern "C" void sppark_bound_violation(const char* what, double got, double limit);
#endif
template<class P, int LB> struct montx_dev {
    static constexpr int NW = P::N;                         // 32-bit words of the standard wire form
    // head-room above the modulus: >= 8 bits with 28-bit limbs (rho = 2^RBITS / p >= 256: a product of any two values the
    // point formulas hold is < (1 + small) p); >= 6 bits with 29-bit limbs -- NINE limbs for a 254- or 255-bit modulus,
    // rho = 169 for alt_bn128 and 128 for the Pasta fields: the largest product of the formulas, P^2 with P < 13 p, is then
    // < 2 p resp. < 2.33 p, and nothing asks more of it (it is only ever the normalised right operand of the next
    // products; the machine-checked bounds of tests/emu/emu_bounds.cpp say so for every operation: ec/xyzzx_dev.hpp, TIGHT)
    static constexpr int HEAD = LB <= 28 ? 8 : 6;
    static constexpr int NL = (P::NBITS + HEAD + LB - 1) / LB;
    // what 32-bit limbs of LB significant bits leave: a lazy sum may reach CAP * 2^LB before it wraps, the left operand
    // of a product 2^31 = FAT_L * 2^LB.  TIGHT (LB = 29: 8 and 4 instead of 16 and 8) is what the point formulas key on.
    static constexpr int CAP = 1 << (32 - LB);
    static constexpr int FAT_L = 1 << (31 - LB);
    static constexpr bool TIGHT = LB > 28;
    // unmasked quotient digits (quotient_digit<true>) need NL (2^32 2^LB + 2^2LB) + carry < 2^64: 28-bit limbs only
    static constexpr bool FAT_M_OK = (double)NL * (4294967296.0 * (double)(1ull << LB) + (double)(1ull << LB) * (double)(1ull << LB))
                                     + 68719476736.0 < 18446744073709551616.0;
    // mul_add's two left operands: limbs < 2^31 and <= 6 2^LB (LB = 28); <= 3 2^LB each with 29-bit limbs
    static constexpr int MA_A0 = TIGHT ? 3 : FAT_L, MA_A1 = TIGHT ? 3 : 6;
    // sqr()'s operand: limbs <= SQR_L 2^LB (cross products use the doubled operand)
    static constexpr int SQR_L = TIGHT ? 2 : 4;
    static constexpr int N = NL;                            // words of the in-memory image (internal form)
    static constexpr u32 MASK = (1u << LB) - 1;
    static constexpr int RBITS = LB * NL;                   // Montgomery radix 2^RBITS
    static_assert((double)NL * (double)(1ull << 31) * (double)(1ull << LB) + (double)NL * (double)(1ull << LB) * (double)(1ull << LB)
                  < 18446744073709551616.0, "a column (one fat operand) must fit the 64-bit accumulator");
    u32 l[NL];

    // ---- machine-checked contracts (host emulation only: tests/emu/emu_bounds.cpp, -DSPPARK_TRACK_BOUNDS) ----------------
    // Every value carries what the comments of this file and of ec/xyzzx*_dev.hpp CLAIM about it -- value < bv * p, every
    // limb <= bl * (2^LB - 1), so that sums of normalised values and the fat constants of sub / neg are exact: a fat limb
    // is below (B + 1) 2^LB, and "limbs < 2^31" reads bl <= 2^(31-LB) -- every operation checks its operands against its stated contract and derives the claim of its
    // result from the rules written at its definition.  The numbers in the limbs play no part: what is checked is the
    // worst case the bounds allow, for ALL inputs, not the sample that happens to be run.  Compiled out everywhere else.
#if defined(SPPARK_HOST_EMULATION) && defined(SPPARK_TRACK_BOUNDS)
# define SPPARK_BND(...) __VA_ARGS__
    double bv = -1.0, bl = -1.0;                                // unset until a rule or the harness sets them
    static double rho()                                         // 2^RBITS / p
    {
        double p = 0.0;
        for (int i = NW - 1; i >= 0; i--) p = p * 4294967296.0 + (double)P::MOD[i];
        double r = 1.0;
        for (int i = 0; i < RBITS; i++) r *= 2.0;
        return r / p;
    }
    static constexpr double FAT_LEFT = (double)(1u << (31 - LB));   // limbs < 2^31 in units of 2^LB
    static void bnd(bool ok, const char* what, double got, double lim) { if (!ok) sppark_bound_violation(what, got, lim); }
    void bnd_set(double v, double lm) { bv = v; bl = lm; }
    void bnd_known(const char* what) const { bnd(bv >= 0.0 && bl >= 0.0, what, bv, 0.0); }
#else
# define SPPARK_BND(...)
#endif

    SPPARK_DEVFN static constexpr u32 limb_of(const u32* w, int j)
    {
        const int bit = LB * j, wi = bit >> 5, sh = bit & 31;
        if (wi >= NW) return 0;
        u64 two = w[wi];
        if (wi + 1 < NW) two |= (u64)w[wi + 1] << 32;
        return (u32)(two >> sh) & MASK;                     // the top limb of a value < 2^(32 NW) is short anyway
    }
    SPPARK_DEVFN static constexpr u32 mod_limb(int j) { return limb_of(P::MOD, j); }

    // Compile-time constant tables.  They are static constexpr OBJECTS (not constexpr function
    // calls in device code, which hipcc would happily evaluate at run time, per limb).
    struct limbs_t { u32 l[NL]; };
    // 2^(32*NW + e) mod p, by doubling P::ONE e times, as internal-width limbs
    SPPARK_DEVFN static constexpr limbs_t make_pow2(int e)
    {
        u32 w[NW] = {};
        for (int i = 0; i < NW; i++) w[i] = P::ONE[i];
        for (int k = 0; k < e; k++) {
            u32 c = 0;
            for (int i = 0; i < NW; i++) { u32 t = (w[i] << 1) | c; c = w[i] >> 31; w[i] = t; }
            u32 d[NW] = {}; u64 bw = 0;                     // conditional subtraction of p
            for (int i = 0; i < NW; i++) { u64 t = (u64)w[i] - P::MOD[i] - bw; d[i] = (u32)t; bw = (t >> 63) & 1; }
            if (c || !bw) for (int i = 0; i < NW; i++) w[i] = d[i];
        }
        limbs_t r{};
        for (int j = 0; j < NL; j++) r.l[j] = limb_of(w, j);
        return r;
    }
    template<int E> struct pow2_tab { static constexpr limbs_t T = make_pow2(E); };
    // K*p in "fat" form: every limb but the top borrows B*2^LB from the next one, so limb j is
    // >= B*(2^LB - 1) for j < NL-1 and the value is still exactly K*p.
    SPPARK_DEVFN static constexpr limbs_t make_fat(int K, int B)
    {
        limbs_t r{}; u64 carry = 0;
        for (int j = 0; j < NL; j++) {
            u64 v = (u64)K * mod_limb(j) + carry; carry = v >> LB;
            r.l[j] = j == NL - 1 ? (u32)v : (u32)(v & MASK);
            if (j < NL - 1) r.l[j] += (u32)B << LB;
            if (j > 0) r.l[j] -= (u32)B;
        }
        return r;
    }
    template<int K, int B> struct fat_tab { static constexpr limbs_t T = make_fat(K, B); };
    // k*p for k < KMAX, normalised limbs (candidates of is_zero_mod)
    template<int KMAX> struct multiples_t { u32 l[KMAX][NL]; };
    template<int KMAX> SPPARK_DEVFN static constexpr multiples_t<KMAX> make_multiples()
    {
        multiples_t<KMAX> r{};
        for (int k = 0; k < KMAX; k++) {
            u64 carry = 0;
            for (int j = 0; j < NL; j++) {
                u64 v = (u64)k * mod_limb(j) + carry; carry = v >> LB;
                r.l[k][j] = j == NL - 1 ? (u32)v : (u32)(v & MASK);
            }
        }
        return r;
    }
    template<int KMAX> struct multiples_tab { static constexpr multiples_t<KMAX> T = make_multiples<KMAX>(); };

    // 1 in the internal domain = 2^RBITS mod p
    SPPARK_DEVFN static montx_dev one()
    {   montx_dev r; for (int j = 0; j < NL; j++) r.l[j] = pow2_tab<RBITS - 32 * NW>::T.l[j]; SPPARK_BND(r.bnd_set(1.0, 1.0);) return r;   }

    // standard wire form (x * 2^(32 NW), 32-bit words, canonical) -> internal (x * 2^RBITS), normalised, < 2p.
    // The two domains differ by 2^SH, SH = RBITS - 32 NW (8 bits on fourteen 28-bit limbs, 5 on nine 29-bit ones), so the
    // conversion is a SHIFT and the subtraction of a small multiple of p -- not a product (round 6: k_convert_points did two
    // 14 x 14 products + reductions per point, 787 of its 1030 vector instructions, and ran 3.9 ms at 2^26 points for 15 GB
    // of traffic).  With t = w 2^SH < 2^SH p and the quotient estimate
    //     q = (top32(w) * QM) >> (62 - SH),   top32 = bits [PB - 32, PB) of the value, QM = floor(2^62 / (top32(p) + 1)),
    // q <= floor(t / p) (both roundings go down) and q >= floor(t / p) - 1 (together they lose less than 2^(SH - 29) <= 2^-5;
    // ten 28-bit limbs over a 254-bit field, alt_bn128's G2, have SH = 24), so t - q p is in [0, 2p).  Canonical input
    // (w < p) is the wire format's contract; any w < 2^PB still lands below 2p.
    // (plain constexpr: evaluated for the constants below by the host AND the device pass of a translation unit)
    static constexpr int mod_bits()
    {
        int i = NW - 1;
        while (i > 0 && P::MOD[i] == 0) i--;
        int b = 32;
        while (b > 1 && !((P::MOD[i] >> (b - 1)) & 1)) b--;
        return 32 * i + b;
    }
    static constexpr int SH = RBITS - 32 * NW, PB = mod_bits(), TB = PB - 32;
    static_assert(SH >= 0 && SH < LB && SH <= 24 && PB > 64 && PB <= 32 * NW, "shift conversion: domain offset and modulus size");
    static constexpr u32 top32(const u32* w)                  // bits [TB, TB + 32)
    {
        const int wi = TB >> 5, sh = TB & 31;
        u64 two = w[wi];
        if (wi + 1 < NW) two |= (u64)w[wi + 1] << 32;
        return (u32)(two >> sh);
    }
    static constexpr u64 PT = (u64)top32(P::MOD) + 1;
    static constexpr u32 QM = (u32)(((u64)1 << 62) / PT);           // < 2^31: top32(p) has its top bit set
    SPPARK_DEVFN static constexpr u32 limb_of_shifted(const u32* w, int j)      // limb j of w * 2^SH
    {
        const int bit = LB * j - SH;
        if (bit < 0) return (u32)((u64)w[0] << (-bit)) & MASK;
        const int wi = bit >> 5, sh = bit & 31;
        if (wi >= NW) return 0;
        u64 two = w[wi];
        if (wi + 1 < NW) two |= (u64)w[wi + 1] << 32;
        return j == NL - 1 ? (u32)(two >> sh) : (u32)(two >> sh) & MASK;
    }
    SPPARK_DEVFN static montx_dev from_std(const u32* w)
    {
        const u32 q = (u32)(((u64)top32(w) * QM) >> (62 - SH));
        montx_dev r;
        long long c = 0;
        #pragma unroll
        for (int j = 0; j < NL; j++) {
            c += (long long)limb_of_shifted(w, j) - (long long)((u64)q * mod_limb(j));
            r.l[j] = j == NL - 1 ? (u32)c : ((u32)c & MASK);
            c >>= LB;
        }
        SPPARK_BND(r.bnd_set(2.0, 1.0);)                            // in [0, 2p), limbs normalised (see above)
        return r;
    }
    // internal (any admissible lazy value: limbs < 2^31) -> canonical standard wire words:
    // v * 2^(32 NW) / 2^RBITS, then the conditional subtraction and re-limbing
    SPPARK_DEVFN void to_std(u32* w) const
    {
        montx_dev k;
        #pragma unroll
        for (int j = 0; j < NL; j++) k.l[j] = pow2_tab<0>::T.l[j];
        SPPARK_BND(k.bnd_set(1.0, 1.0);)
        (*this * k).cond_sub_p().to_words(w);                       // the product: < v*p/2^RBITS + p < 2p
    }
    // a normalised value < 2p -> its canonical representative: r - p if r >= p, limb-wise with borrow
    SPPARK_DEVFN montx_dev cond_sub_p() const
    {
        SPPARK_BND(bnd_known("cond_sub_p: operand"); bnd(bl <= 1.0, "cond_sub_p: normalised", bl, 1.0); bnd(bv <= 2.0, "cond_sub_p: < 2p", bv, 2.0);)
        montx_dev r = *this;
        SPPARK_BND(r.bnd_set(1.0, 1.0);)
        u32 d[NL]; int bw = 0;
        #pragma unroll
        for (int j = 0; j < NL; j++) {
            int v = (int)r.l[j] - (int)mod_limb(j) - bw;
            d[j] = j == NL - 1 ? (u32)v : ((u32)v & MASK); bw = (v >> 31) & 1;
        }
        #pragma unroll
        for (int j = 0; j < NL; j++) r.l[j] = bw ? r.l[j] : d[j];
        return r;
    }
    // the integer itself (no change of Montgomery domain): NW 32-bit words <-> NL limbs.  to_words: normalised limbs,
    // value < 2^(32 NW)
    SPPARK_DEVFN static montx_dev from_words(const u32* w)
    {
        montx_dev a;
        #pragma unroll
        for (int j = 0; j < NL; j++) a.l[j] = limb_of(w, j);
        return a;
    }
    SPPARK_DEVFN void to_words(u32* w) const
    {
        #pragma unroll
        for (int i = 0; i < NW; i++) {
            const int bit = 32 * i, j = bit / LB, sh = bit % LB;
            u64 acc = (u64)l[j] >> sh;
            if (j + 1 < NL) acc |= (u64)l[j + 1] << (LB - sh);
            if (j + 2 < NL && 2 * LB - sh < 32) acc |= (u64)l[j + 2] << (2 * LB - sh);
            w[i] = (u32)acc;
        }
    }

    // in-memory image between the MSM kernels = the limbs themselves
    SPPARK_DEVFN static montx_dev from_wire(const u32* w)
    {   montx_dev r; for (int j = 0; j < NL; j++) r.l[j] = w[j]; return r;   }
    SPPARK_DEVFN void to_wire(u32* w) const { for (int j = 0; j < NL; j++) w[j] = l[j]; }

    SPPARK_DEVFN static montx_dev zero() { montx_dev r; for (int j = 0; j < NL; j++) r.l[j] = 0; SPPARK_BND(r.bnd_set(0.0, 0.0);) return r; }
    SPPARK_DEVFN bool limbs_all_zero() const
    {   u32 acc = l[0]; for (int j = 1; j < NL; j++) acc |= l[j]; return acc == 0;   }

    // carry propagation: limbs < 2^LB afterwards (the top limb takes what is left; same value)
    SPPARK_DEVFN montx_dev norm() const
    {
        montx_dev r; u32 c = 0;
        #pragma unroll
        for (int j = 0; j < NL - 1; j++) { u32 v = l[j] + c; r.l[j] = v & MASK; c = v >> LB; }
        r.l[NL - 1] = l[NL - 1] + c;
        // (limb + carry must not wrap: limbs <= (CAP - 1) 2^LB; the top limb of a value < bv p is far below 2^LB)
        SPPARK_BND(bnd_known("norm: operand"); bnd(bl <= (double)(CAP - 1), "norm: limbs", bl, (double)(CAP - 1)); r.bnd_set(bv, bl < 1.0 ? bl : 1.0);)
        return r;
    }

    // a + b limb-wise, no carries
    SPPARK_DEVFN friend montx_dev operator+(const montx_dev& a, const montx_dev& b)
    {
        montx_dev r;
        #pragma unroll
        for (int j = 0; j < NL; j++) r.l[j] = a.l[j] + b.l[j];
        SPPARK_BND(a.bnd_known("+: left"); b.bnd_known("+: right"); r.bnd_set(a.bv + b.bv, a.bl + b.bl); bnd(r.bl < (double)CAP, "+: limbs wrap", r.bl, (double)CAP);)
        return r;
    }

    // a + K*p - b.  Contract: b's limbs <= B*(2^LB - 1) (B = 1: normalised), b < (K-1)*p
    // (so that the top limbs cannot underflow either), a's limbs + (B+1)*2^LB < 2^32.
    template<int K, int B = 1> SPPARK_DEVFN static montx_dev sub(const montx_dev& a, const montx_dev& b)
    {
        montx_dev r;
        #pragma unroll
        for (int j = 0; j < NL; j++) r.l[j] = a.l[j] + (fat_tab<K, B>::T.l[j] - b.l[j]);
        SPPARK_BND(a.bnd_known("sub: minuend"); b.bnd_known("sub: subtrahend");
                   bnd(b.bv <= (double)(K - 1), "sub<K>: subtrahend < (K-1) p", b.bv, (double)(K - 1));
                   bnd(b.bl <= (double)B, "sub<K,B>: subtrahend's limbs <= B 2^LB", b.bl, (double)B);
                   r.bnd_set(a.bv + (double)K, a.bl + (double)(B + 1));
                   bnd(r.bl < (double)CAP, "sub: limbs wrap", r.bl, (double)CAP);)
        return r;
    }
    // K*p - b, same contract
    template<int K, int B = 1> SPPARK_DEVFN static montx_dev neg(const montx_dev& b)
    {
        montx_dev r;
        #pragma unroll
        for (int j = 0; j < NL; j++) r.l[j] = fat_tab<K, B>::T.l[j] - b.l[j];
        SPPARK_BND(b.bnd_known("neg: operand");
                   bnd(b.bv <= (double)(K - 1), "neg<K>: operand < (K-1) p", b.bv, (double)(K - 1));
                   bnd(b.bl <= (double)B, "neg<K,B>: operand's limbs <= B 2^LB", b.bl, (double)B);
                   r.bnd_set((double)K, (double)(B + 1));)
        return r;
    }

    // accumulator >> LB: ONE 64-bit shift.  (Round 1 split it into v_alignbit_b32 + v_lshrrev_b32 on the
    // belief that a 64-bit shift costs two issue slots; measured A/B on one box, 2^26 points:
    // 114.0 ms with the pair, 112.3 ms with the single instruction -- with two waves per SIMD the
    // kernel is bound by the NUMBER of instructions a wave gets to issue, tools/gpu_r2_job19.sh.)
    SPPARK_DEVFN static u64 shift_down(u64 A)
    {
#if defined(__HIP_DEVICE_COMPILE__)
        asm("v_lshrrev_b64 %0, %1, %0" : "+v"(A) : "n"(LB));
        return A;
#else
        return A >> LB;
#endif
    }

    // Montgomery product a*b / 2^RBITS (mod p).  Contract: b normalised (limbs < 2^LB), a's
    // limbs < 2^31 (NL*2^(31+LB) + NL*2^(2LB) < 2^64).  Output normalised, value < a*b/2^RBITS + p.
    SPPARK_DEVFN friend montx_dev operator*(const montx_dev& a, const montx_dev& b)
    {
        constexpr u32 PINV = P::M0 & MASK;                  // -1/p mod 2^LB
        u32 m[NL];
        montx_dev r;
        u64 A = 0;
        #pragma unroll
        for (int k = 0; k < 2 * NL; k++) {
            const int lo = k < NL ? 0 : k - NL + 1, hi = k < NL ? k : NL - 1;    // i range with 0 <= k-i < NL
            if (k <= 2 * NL - 2) {
                #pragma unroll
                for (int i = lo; i <= hi; i++) macx(A, a.l[i], b.l[k - i]);
                #pragma unroll
                for (int i = lo; i <= hi; i++) if (i < k) macxs(A, m[i], mod_limb(k - i));   // m[i] known for i < k
            }
            if (k < NL) {
                m[k] = ((u32)A * PINV) & MASK;
                macxs(A, m[k], mod_limb(0));
            } else {
                r.l[k - NL] = (u32)A & MASK;
            }
            A = shift_down(A);
        }
        SPPARK_BND(a.bnd_known("*: left"); b.bnd_known("*: right");
                   bnd(b.bl <= 1.0, "*: right operand normalised", b.bl, 1.0);
                   bnd(a.bl <= FAT_LEFT + 1e-6, "*: left operand's limbs < 2^31", a.bl, FAT_LEFT);
                   r.bnd_set(a.bv * b.bv / rho() + 1.0, 1.0);)
        return r;
    }
    // The Montgomery quotient digit of column k.  m = A * (-1/p) mod 2^LB makes A + m*p divisible by 2^LB;
    // so does any m' = m (mod 2^LB), and m' = the low 32 bits of A * (-1/p) saves the mask.  It costs
    // head-room: m' < 2^32 makes the m*p terms of a column 16 times larger and adds (m' - m) * p / 2^LB
    // <= 16 p / 2^LB to the value per later column -- negligible except for the LAST digit, which stays
    // masked.  FAT_M is therefore allowed only when BOTH operands are normalised:
    //   NL * (2^32 * 2^LB + 2^LB * 2^LB) + carry < 2^64   (static_assert below),
    // and the result is < a*b/2^RBITS + (1 + 2^-20) p, normalised as before.
    template<bool FAT_M> SPPARK_DEVFN static u32 quotient_digit(u64 A, int k)
    {
        constexpr u32 PINV = P::M0 & MASK;
        // (where the accumulator has no room for unmasked digits -- 29-bit limbs -- the request is ignored: masked digits
        // satisfy every caller's contract)
        if (FAT_M && FAT_M_OK && k < NL - 1) return (u32)A * (u32)P::M0;
        return ((u32)A * PINV) & MASK;
    }

    // Two independent products with their multiply-add chains interleaved: a v_mad_u64_u32
    // that feeds the next one through its addend costs an extra wait state (and hipcc pads
    // it with an s_nop); alternating two accumulators hides it and halves the padding.
    // NORM0 / NORM1: the left operand of that product is normalised too (unmasked quotient digits).
    template<bool NORM0 = false, bool NORM1 = false>
    SPPARK_DEVFN static void mul2(montx_dev& r0, montx_dev& r1,
                                  const montx_dev& a0, const montx_dev& b0,
                                  const montx_dev& a1, const montx_dev& b1)
    {
        u32 m0[NL], m1[NL];
        u64 A0 = 0, A1 = 0;
        #pragma unroll
        for (int k = 0; k < 2 * NL; k++) {
            const int lo = k < NL ? 0 : k - NL + 1, hi = k < NL ? k : NL - 1;
            if (k <= 2 * NL - 2) {
                #pragma unroll
                for (int i = lo; i <= hi; i++) macx2(A0, a0.l[i], b0.l[k - i], A1, a1.l[i], b1.l[k - i]);
                #pragma unroll
                for (int i = lo; i <= hi; i++) if (i < k) macxs2(A0, m0[i], A1, m1[i], mod_limb(k - i));
            }
            if (k < NL) {
                m0[k] = quotient_digit<NORM0>(A0, k); m1[k] = quotient_digit<NORM1>(A1, k);
                macxs2(A0, m0[k], A1, m1[k], mod_limb(0));
            } else {
                r0.l[k - NL] = (u32)A0 & MASK; r1.l[k - NL] = (u32)A1 & MASK;
            }
            A0 = shift_down(A0); A1 = shift_down(A1);
        }
        SPPARK_BND(a0.bnd_known("mul2: a0"); b0.bnd_known("mul2: b0"); a1.bnd_known("mul2: a1"); b1.bnd_known("mul2: b1");
                   bnd(b0.bl <= 1.0 && b1.bl <= 1.0, "mul2: right operands normalised", b0.bl > b1.bl ? b0.bl : b1.bl, 1.0);
                   bnd(a0.bl <= (NORM0 ? 1.0 : FAT_LEFT) + 1e-6, NORM0 ? "mul2<NORM0>: a0 normalised" : "mul2: a0's limbs < 2^31", a0.bl, NORM0 ? 1.0 : FAT_LEFT);
                   bnd(a1.bl <= (NORM1 ? 1.0 : FAT_LEFT) + 1e-6, NORM1 ? "mul2<NORM1>: a1 normalised" : "mul2: a1's limbs < 2^31", a1.bl, NORM1 ? 1.0 : FAT_LEFT);
                   r0.bnd_set(a0.bv * b0.bv / rho() + 1.0 + (NORM0 ? 1e-6 : 0.0), 1.0);
                   r1.bnd_set(a1.bv * b1.bv / rho() + 1.0 + (NORM1 ? 1e-6 : 0.0), 1.0);)
    }
    // (a0*b0 + a1*b1) / 2^RBITS with ONE Montgomery reduction (a sum of two products needs no more
    // than one: NL^2 multiply-adds saved against two separate products and a subtraction).
    // Contract: b0, b1 normalised; a0's limbs < 2^31, a1's limbs < 6*2^LB, so that a column of both
    // products and the m*p terms stays below NL*(8 + 6 + 1)*2^(2*LB) < 2^64 (29-bit limbs: both <= 3*2^LB, 9*7*2^58).  The two products run
    // in two accumulators (their chains interleave like mul2's) that are joined once per column.
    // Output normalised, value < (a0*b0 + a1*b1)/2^RBITS + p.
    SPPARK_DEVFN static montx_dev mul_add(const montx_dev& a0, const montx_dev& b0,
                                          const montx_dev& a1, const montx_dev& b1)
    {
        static_assert((double)NL * ((double)MA_A0 + (double)MA_A1 + 1.0) * (double)(1ull << LB) * (double)(1ull << LB) + 68719476736.0
                      < 18446744073709551616.0, "a column of two products must fit the 64-bit accumulator");
        constexpr u32 PINV = P::M0 & MASK;
        u32 m[NL];
        montx_dev r;
        u64 A = 0;
        #pragma unroll
        for (int k = 0; k < 2 * NL; k++) {
            const int lo = k < NL ? 0 : k - NL + 1, hi = k < NL ? k : NL - 1;
            if (k <= 2 * NL - 2) {
                u64 B = 0;
                #pragma unroll
                for (int i = lo; i <= hi; i++) macx2(A, a0.l[i], b0.l[k - i], B, a1.l[i], b1.l[k - i]);
                // m[i] is known for i < k; the m*p terms alternate between the two chains
                const int mhi = hi < k ? hi : k - 1;
                #pragma unroll
                for (int i = lo; i + 1 <= mhi; i += 2) {
#if defined(__HIP_DEVICE_COMPILE__)
                    asm("v_mad_u64_u32 %0, vcc, %2, %4, %0\n\tv_mad_u64_u32 %1, vcc, %3, %5, %1"
                        : "+v"(A), "+v"(B) : "v"(m[i]), "v"(m[i + 1]), "s"(mod_limb(k - i)), "s"(mod_limb(k - i - 1)) : "vcc");
#else
                    A += (u64)m[i] * mod_limb(k - i); B += (u64)m[i + 1] * mod_limb(k - i - 1);
#endif
                }
                if (mhi >= lo && ((mhi - lo + 1) & 1)) macxs(A, m[mhi], mod_limb(k - mhi));
                A += B;
            }
            if (k < NL) {
                m[k] = ((u32)A * PINV) & MASK;
                macxs(A, m[k], mod_limb(0));
            } else {
                r.l[k - NL] = (u32)A & MASK;
            }
            A = shift_down(A);
        }
        SPPARK_BND(a0.bnd_known("mul_add: a0"); b0.bnd_known("mul_add: b0"); a1.bnd_known("mul_add: a1"); b1.bnd_known("mul_add: b1");
                   bnd(b0.bl <= 1.0 && b1.bl <= 1.0, "mul_add: right operands normalised", b0.bl > b1.bl ? b0.bl : b1.bl, 1.0);
                   bnd(a0.bl <= (double)MA_A0 + 1e-6, "mul_add: a0's limbs <= MA_A0 2^LB", a0.bl, (double)MA_A0);
                   bnd(a1.bl <= (double)MA_A1, "mul_add: a1's limbs <= MA_A1 2^LB", a1.bl, (double)MA_A1);
                   r.bnd_set((a0.bv * b0.bv + a1.bv * b1.bv) / rho() + 1.0, 1.0);)
        return r;
    }

    // two squares, interleaved likewise (inputs normalised: unmasked quotient digits -- a cross term uses
    // the doubled limb, (NL/2) * 2^(2LB+1) per column, the same bound as NL * 2^(2LB))
    SPPARK_DEVFN static void sqr2(montx_dev& r0, montx_dev& r1, const montx_dev& a0, const montx_dev& a1)
    {
        u32 m0[NL], m1[NL], d0[NL], d1[NL];
        #pragma unroll
        for (int j = 0; j < NL; j++) { d0[j] = a0.l[j] << 1; d1[j] = a1.l[j] << 1; }
        u64 A0 = 0, A1 = 0;
        #pragma unroll
        for (int k = 0; k < 2 * NL; k++) {
            const int lo = k < NL ? 0 : k - NL + 1, hi = k < NL ? k : NL - 1;
            if (k <= 2 * NL - 2) {
                #pragma unroll
                for (int i = lo; i <= hi; i++) {
                    const int j = k - i;
                    if (i > j) continue;
                    if (i == j) macx2(A0, a0.l[i], a0.l[i], A1, a1.l[i], a1.l[i]);
                    else        macx2(A0, a0.l[i], d0[j], A1, a1.l[i], d1[j]);
                }
                #pragma unroll
                for (int i = lo; i <= hi; i++) if (i < k) macxs2(A0, m0[i], A1, m1[i], mod_limb(k - i));
            }
            if (k < NL) {
                m0[k] = quotient_digit<true>(A0, k); m1[k] = quotient_digit<true>(A1, k);
                macxs2(A0, m0[k], A1, m1[k], mod_limb(0));
            } else {
                r0.l[k - NL] = (u32)A0 & MASK; r1.l[k - NL] = (u32)A1 & MASK;
            }
            A0 = shift_down(A0); A1 = shift_down(A1);
        }
        SPPARK_BND(a0.bnd_known("sqr2: a0"); a1.bnd_known("sqr2: a1");
                   bnd(a0.bl <= 1.0 && a1.bl <= 1.0, "sqr2: operands normalised", a0.bl > a1.bl ? a0.bl : a1.bl, 1.0);
                   r0.bnd_set(a0.bv * a0.bv / rho() + 1.0 + 1e-6, 1.0); r1.bnd_set(a1.bv * a1.bv / rho() + 1.0 + 1e-6, 1.0);)
    }

    // a^2 / 2^RBITS.  Contract: limbs < 2^(LB+2) (cross products use the doubled operand:
    // NL/2 of them < 2^(2LB+5) each).
    SPPARK_DEVFN montx_dev sqr() const
    {
        // a column: at most NL/2 cross terms l[i] * 2 l[j], one diagonal term, NL quotient-digit terms, the carry
        static_assert(((double)(NL / 2) * 2.0 * SQR_L * SQR_L + (double)(SQR_L * SQR_L) + (double)NL) * (double)(1ull << LB) * (double)(1ull << LB)
                      + 68719476736.0 < 18446744073709551616.0, "a column of a square of limbs <= SQR_L 2^LB must fit the 64-bit accumulator");
        constexpr u32 PINV = P::M0 & MASK;
        u32 m[NL], d[NL];
        #pragma unroll
        for (int j = 0; j < NL; j++) d[j] = l[j] << 1;
        montx_dev r;
        u64 A = 0;
        #pragma unroll
        for (int k = 0; k < 2 * NL; k++) {
            if (k <= 2 * NL - 2) {
                #pragma unroll
                for (int i = 0; i < NL; i++) {
                    const int j = k - i;
                    if (j < 0 || j >= NL || i > j) continue;
                    if (i == j) macx(A, l[i], l[i]);
                    else        macx(A, l[i], d[j]);
                }
                #pragma unroll
                for (int i = 0; i < NL; i++) {
                    const int j = k - i;
                    if (j < 0 || j >= NL || i >= k) continue;
                    macxs(A, m[i], mod_limb(j));
                }
            }
            if (k < NL) {
                m[k] = ((u32)A * PINV) & MASK;
                macxs(A, m[k], mod_limb(0));
            } else {
                r.l[k - NL] = (u32)A & MASK;
            }
            A = shift_down(A);
        }
        SPPARK_BND(bnd_known("sqr: operand"); bnd(bl <= (double)SQR_L, "sqr: limbs <= SQR_L 2^LB", bl, (double)SQR_L); r.bnd_set(bv * bv / rho() + 1.0, 1.0);)
        return r;
    }

    // a^(p-2) = 1/a by square-and-multiply over the words of the modulus (|*this| normalised, value != 0 mod p;
    // result normalised, < 2p).  ~NBITS squares + NBITS/2 products: used once per table entry of the fixed-base
    // mode (msm_kernels.hpp k_fixed_base_table), never on an MSM's hot path.
    SPPARK_DEVFN montx_dev inverse() const
    {
        montx_dev r = one();
        bool started = false;
        #pragma unroll 1
        for (int w = NW - 1; w >= 0; w--) {
            u32 e = 0, borrow = 2;                              // word w of p - 2 (BLS12-377's p ends in ...00000001)
            for (int k = 0; k <= w; k++) { e = P::MOD[k] - borrow; borrow = P::MOD[k] < borrow ? 1u : 0u; }
            #pragma unroll 1
            for (int b = 31; b >= 0; b--) {
                if (started) r = r.sqr();
                if ((e >> b) & 1u) { r = started ? r * *this : *this; started = true; }
            }
        }
        return r;
    }

    // value == 0 (mod p) for a NORMALISED value < KMAX*p: it must be one of 0, p, 2p, ...;
    // the low limb filters out almost everything before the exact comparison.
    template<int KMAX> SPPARK_DEVFN bool is_zero_mod() const
    {
        SPPARK_BND(bnd_known("is_zero_mod: operand"); bnd(bl <= 1.0, "is_zero_mod: normalised", bl, 1.0);
                   bnd(bv <= (double)KMAX, "is_zero_mod<KMAX>: value < KMAX p", bv, (double)KMAX);)
        // (A one-multiply filter -- k*p has the low limb l[0] iff k = l[0] / p mod 2^LB -- saves ~30 scalar and
        // vector instructions per mixed addition, but hipcc then allocates 269 registers for k_accumulate
        // instead of 238: ONE wave per SIMD, 151 ms instead of 114 at 2^26 points.  Measured and reverted,
        // profiles/r03_msm_diet_ab.log; tests/test_build_resources.py now pins the register budget.)
        bool maybe = false;
        #pragma unroll
        for (int kk = 0; kk < KMAX; kk++) maybe |= (l[0] == multiples_tab<KMAX>::T.l[kk][0]);
        if (!maybe) return false;
        bool hit = false;
        #pragma unroll 1
        for (int k = 0; k < KMAX; k++) {                    // rare: keep it small, not unrolled over k
            u32 diff = 0;
            #pragma unroll
            for (int j = 0; j < NL; j++) diff |= multiples_tab<KMAX>::T.l[k][j] ^ l[j];
            hit |= diff == 0;
        }
        return hit;
    }
};

} // namespace sppark_amd
