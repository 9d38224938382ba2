// ORACLE — TEST INFRASTRUCTURE ONLY (see ff.hpp header).
// C entry points (ctypes) over the CPU restatement: field ops, G1 helpers,
// the three MSM evaluators and the NTT.  Built by oracle/Makefile into
// oracle/liboracle.so.
#include "ff.hpp"
#include "ec.hpp"
#include "msm.hpp"
#include "ntt.hpp"
#include <vector>
#include <cstring>

using namespace oracle;

namespace {

template<class F> F load_f(const void* p)
{   F r; memcpy(r.v, p, sizeof(r.v)); return r;   }
template<class F> void store_f(void* p, const F& a)
{   memcpy(p, a.v, sizeof(a.v));   }

template<class F> int field_op(int op, uint64_t* out, const uint64_t* a, const uint64_t* b)
{
    F x = load_f<F>(a), y;
    if (b) y = load_f<F>(b); else y.zero();
    switch (op) {
        case 0: x += y; break;
        case 1: x -= y; break;
        case 2: x *= y; break;
        case 3: x = x.reciprocal(); break;
        case 4: x.to(); break;
        case 5: x.from(); break;
        case 6: x.cneg(true); break;
        case 7: x ^= 2; break;
        default: return -1;
    }
    store_f(out, x);
    return 0;
}

struct splitmix64 {
    uint64_t s;
    uint64_t next()
    {
        uint64_t z = (s += 0x9e3779b97f4a7c15ULL);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
        return z ^ (z >> 31);
    }
};

// canonical big-endian hex -> Montgomery field element
template<class F> F from_hex(const char* hex)
{
    F r; r.zero();
    size_t len = strlen(hex);
    for (size_t i = 0; i < len; i++) {
        char ch = hex[len - 1 - i];
        uint64_t nib = ch <= '9' ? ch - '0' : (ch | 0x20) - 'a' + 10;
        r.v[i / 16] |= nib << (4 * (i % 16));
    }
    r.to();
    return r;
}

template<class FP, class FR> struct curve_t {
    typedef FP fp; typedef FR fr;
    static const size_t fp_bytes = sizeof(FP), fr_bits = FR::nbits;

    static affine<FP> load_affine(const unsigned char* p, size_t stride)
    {
        affine<FP> a;
        memcpy(a.X.v, p, fp_bytes); memcpy(a.Y.v, p + fp_bytes, fp_bytes);
        // flagged wire format (ec/affine_t.hpp:64-122, Affine_inf_t): the byte
        // after Y is the infinity flag; the plain format encodes inf as X=Y=0.
        if (stride > 2 * fp_bytes && (p[2 * fp_bytes] & 1)) a.set_inf();
        return a;
    }
    static void store_affine(unsigned char* p, const affine<FP>& a, size_t stride)
    {
        memset(p, 0, stride);
        memcpy(p, a.X.v, fp_bytes); memcpy(p + fp_bytes, a.Y.v, fp_bytes);
        if (stride > 2 * fp_bytes) p[2 * fp_bytes] = a.is_inf();
    }
    static jacobian<FP> load_jac(const unsigned char* p)
    {
        jacobian<FP> j;
        memcpy(j.X.v, p, fp_bytes); memcpy(j.Y.v, p + fp_bytes, fp_bytes);
        memcpy(j.Z.v, p + 2 * fp_bytes, fp_bytes);
        return j;
    }
    static void store_jac(unsigned char* p, const jacobian<FP>& j)
    {
        memcpy(p, j.X.v, fp_bytes); memcpy(p + fp_bytes, j.Y.v, fp_bytes);
        memcpy(p + 2 * fp_bytes, j.Z.v, fp_bytes);
    }
};

struct bls12_381_g1 : curve_t<bls12_381_fp, bls12_381_fr> {
    // standard G1 generator (not stored in the reference)
    static affine<fp> generator()
    {
        affine<fp> g;
        g.X = from_hex<fp>("17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb");
        g.Y = from_hex<fp>("08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1");
        return g;
    }
    static fp b() { return from_hex<fp>("4"); }
};
struct alt_bn128_g1 : curve_t<alt_bn128_fp, alt_bn128_fr> {
    static affine<fp> generator()
    {   affine<fp> g; g.X = from_hex<fp>("1"); g.Y = from_hex<fp>("2"); return g;   }
    static fp b() { return from_hex<fp>("3"); }
};

template<class C>
std::vector<unsigned char> plain_scalars(const unsigned char* scalars, size_t n, int mont)
{
    const size_t nbytes = (C::fr_bits + 7) / 8;
    std::vector<unsigned char> out(n * nbytes);
    for (size_t i = 0; i < n; i++) {
        if (mont) {
            typename C::fr s = load_f<typename C::fr>(scalars + i * sizeof(typename C::fr));
            typename C::fr::pow_t le;
            s.to_scalar(le);
            memcpy(&out[i * nbytes], le, nbytes);
        } else {
            memcpy(&out[i * nbytes], scalars + i * sizeof(typename C::fr), nbytes);
        }
    }
    return out;
}


template<class C>
void gen_points(unsigned char* out, size_t stride, size_t n, uint64_t seed)
{
    typedef typename C::fp fp;
    splitmix64 rng{seed};
    affine<fp> g = C::generator();
    for (size_t i = 0; i < n; i++) {
        unsigned char k[32];
        for (int w = 0; w < 4; w++) { uint64_t x = rng.next(); memcpy(k + 8 * w, &x, 8); }
        k[31] &= 0x1f;                                  // < 2^253 < r
        jacobian<fp> j;
        mult(j, g, k, 253);
        C::store_affine(out + i * stride, j.to_affine(), stride);
    }
}

template<class C>
void g1_mul(unsigned char* out, const unsigned char* in, size_t stride, const unsigned char* k)
{
    jacobian<typename C::fp> j;
    mult(j, C::load_affine(in, stride), k, C::fr_bits);
    C::store_affine(out, j.to_affine(), stride);
}

template<class C> int on_curve_impl(const unsigned char* in, size_t stride)
{   return on_curve(C::load_affine(in, stride), C::b());   }

template<class C> void jac_to_affine_impl(unsigned char* out, const unsigned char* in, size_t stride)
{   C::store_affine(out, C::load_jac(in).to_affine(), stride);   }

template<class C> void xyzz_to_affine_impl(unsigned char* out, const unsigned char* in, size_t stride)
{
    xyzz<typename C::fp> p;
    const size_t fb = C::fp_bytes;
    memcpy(p.X.v, in, fb); memcpy(p.Y.v, in + fb, fb);
    memcpy(p.ZZZ.v, in + 2 * fb, fb); memcpy(p.ZZ.v, in + 3 * fb, fb);
    C::store_affine(out, p.to_affine(), stride);
}

template<class C> void jac_add_impl(unsigned char* out, const unsigned char* a, const unsigned char* b)
{   auto p = C::load_jac(a); p.add(C::load_jac(b)); C::store_jac(out, p);   }

template<class C> void jac_dbl_impl(unsigned char* out, const unsigned char* a)
{   auto p = C::load_jac(a); p.dbl(); C::store_jac(out, p);   }

template<class C> int jac_eq_impl(const unsigned char* a, const unsigned char* b)
{   return C::load_jac(a) == C::load_jac(b);   }

template<class C>
int msm_dispatch(int algo, unsigned char* out_jac, const unsigned char* points, size_t stride,
                 size_t n, const unsigned char* scalars, int mont, size_t param)
{
    typedef typename C::fp fp;
    std::vector<affine<fp>> pts(n);
    for (size_t i = 0; i < n; i++) pts[i] = C::load_affine(points + i * stride, stride);
    std::vector<unsigned char> sc = plain_scalars<C>(scalars, n, mont);
    jacobian<fp> r;
    switch (algo) {
        case 0: mult_pippenger(r, pts.data(), n, sc.data(), C::fr_bits, param); break;
        case 1: msm_naive(r, pts.data(), n, sc.data(), C::fr_bits); break;
        case 2: msm_signed(r, pts.data(), n, sc.data(), C::fr_bits, param, C::fr::modulus()); break;
        default: return -1;
    }
    C::store_jac(out_jac, r);
    return 0;
}

} // namespace

#define CURVE_DISPATCH(curve, call_bls, call_bn) \
    switch (curve) { case 0: call_bls; break; case 1: call_bn; break; default: return -1; }
#define CURVE_RETURN(curve, call_bls, call_bn) \
    switch (curve) { case 0: return call_bls; case 1: return call_bn; default: return -1; }

extern "C" {

// field: 0 bls12_381 fp, 1 bls12_381 fr, 2 alt_bn128 fp, 3 alt_bn128 fr
// op: 0 add 1 sub 2 montmul 3 inverse 4 to_mont 5 from_mont 6 neg 7 sqr
int oracle_field_op(int field, int op, uint64_t* out, const uint64_t* a, const uint64_t* b)
{
    switch (field) {
        case 0: return field_op<bls12_381_fp>(op, out, a, b);
        case 1: return field_op<bls12_381_fr>(op, out, a, b);
        case 2: return field_op<alt_bn128_fp>(op, out, a, b);
        case 3: return field_op<alt_bn128_fr>(op, out, a, b);
    }
    return -1;
}

int oracle_g1_generator(int curve, unsigned char* out, size_t stride)
{
    CURVE_DISPATCH(curve,
        bls12_381_g1::store_affine(out, bls12_381_g1::generator(), stride),
        alt_bn128_g1::store_affine(out, alt_bn128_g1::generator(), stride));
    return 0;
}

// P_i = k_i * G with k_i from splitmix64(seed), 253-bit
int oracle_g1_gen_points(int curve, unsigned char* out, size_t stride, size_t n, uint64_t seed)
{
    CURVE_DISPATCH(curve, gen_points<bls12_381_g1>(out, stride, n, seed),
                          gen_points<alt_bn128_g1>(out, stride, n, seed));
    return 0;
}

int oracle_g1_mul(int curve, unsigned char* out, const unsigned char* in, size_t stride, const unsigned char* scalar_le)
{
    CURVE_DISPATCH(curve, g1_mul<bls12_381_g1>(out, in, stride, scalar_le),
                          g1_mul<alt_bn128_g1>(out, in, stride, scalar_le));
    return 0;
}

int oracle_g1_on_curve(int curve, const unsigned char* in, size_t stride)
{   CURVE_RETURN(curve, on_curve_impl<bls12_381_g1>(in, stride), on_curve_impl<alt_bn128_g1>(in, stride));   }

int oracle_jac_to_affine(int curve, unsigned char* out, const unsigned char* in_jac, size_t stride)
{
    CURVE_DISPATCH(curve, jac_to_affine_impl<bls12_381_g1>(out, in_jac, stride),
                          jac_to_affine_impl<alt_bn128_g1>(out, in_jac, stride));
    return 0;
}

int oracle_xyzz_to_affine(int curve, unsigned char* out, const unsigned char* in_xyzz, size_t stride)
{
    CURVE_DISPATCH(curve, xyzz_to_affine_impl<bls12_381_g1>(out, in_xyzz, stride),
                          xyzz_to_affine_impl<alt_bn128_g1>(out, in_xyzz, stride));
    return 0;
}

int oracle_jac_add(int curve, unsigned char* out, const unsigned char* a, const unsigned char* b)
{
    CURVE_DISPATCH(curve, jac_add_impl<bls12_381_g1>(out, a, b), jac_add_impl<alt_bn128_g1>(out, a, b));
    return 0;
}

int oracle_jac_dbl(int curve, unsigned char* out, const unsigned char* a)
{
    CURVE_DISPATCH(curve, jac_dbl_impl<bls12_381_g1>(out, a), jac_dbl_impl<alt_bn128_g1>(out, a));
    return 0;
}

int oracle_jac_eq(int curve, const unsigned char* a, const unsigned char* b)
{   CURVE_RETURN(curve, jac_eq_impl<bls12_381_g1>(a, b), jac_eq_impl<alt_bn128_g1>(a, b));   }

// algo 0: restated msm/pippenger.hpp with |param| = ncpus (0/1 -> serial path)
// algo 1: naive sum of double-and-add
// algo 2: signed-window model of the GPU semantics with |param| = window bits
// out_jac: X|Y|Z Montgomery limbs (144 B BLS12-381, 96 B alt_bn128)
int oracle_msm(int curve, int algo, unsigned char* out_jac, const unsigned char* points,
               size_t stride, size_t npoints, const unsigned char* scalars, int mont, size_t param)
{
    CURVE_RETURN(curve,
        msm_dispatch<bls12_381_g1>(algo, out_jac, points, stride, npoints, scalars, mont, param),
        msm_dispatch<alt_bn128_g1>(algo, out_jac, points, stride, npoints, scalars, mont, param));
}

void oracle_ntt_gl64(uint64_t* inout, unsigned lg, int order, int direction, int type)
{   ntt(reinterpret_cast<gl64*>(inout), lg, order, direction, type);   }
void oracle_ntt_bb31(uint32_t* inout, unsigned lg, int order, int direction, int type)
{   ntt(reinterpret_cast<bb31*>(inout), lg, order, direction, type);   }
void oracle_ntt_naive_gl64(uint64_t* out, const uint64_t* in, unsigned lg, int inv)
{   ntt_naive(reinterpret_cast<gl64*>(out), reinterpret_cast<const gl64*>(in), lg, inv != 0);   }
void oracle_ntt_naive_bb31(uint32_t* out, const uint32_t* in, unsigned lg, int inv)
{   ntt_naive(reinterpret_cast<bb31*>(out), reinterpret_cast<const bb31*>(in), lg, inv != 0);   }

// 256-bit scalar fields: field 0 = BLS12-381 Fr, 1 = alt_bn128 Fr; elements are 4 x u64 Montgomery limbs
void oracle_ntt_fr(int field, uint64_t* inout, unsigned lg, int order, int direction, int type)
{
    if (field == 0) ntt(reinterpret_cast<bls12_381_fr*>(inout), lg, order, direction, type);
    else            ntt(reinterpret_cast<alt_bn128_fr*>(inout), lg, order, direction, type);
}
void oracle_ntt_naive_fr(int field, uint64_t* out, const uint64_t* in, unsigned lg, int inv)
{
    if (field == 0) ntt_naive(reinterpret_cast<bls12_381_fr*>(out), reinterpret_cast<const bls12_381_fr*>(in), lg, inv != 0);
    else            ntt_naive(reinterpret_cast<alt_bn128_fr*>(out), reinterpret_cast<const alt_bn128_fr*>(in), lg, inv != 0);
}
void oracle_fr_root(int field, uint64_t* out, unsigned lg)
{
    if (field == 0) { auto w = root_of_unity<bls12_381_fr>(lg); memcpy(out, w.v, 32); }

    else            { auto w = root_of_unity<alt_bn128_fr>(lg); memcpy(out, w.v, 32); }
}
// LDE family; field: 0 gl64, 1 bb31, 2 bls12_381 fr, 3 alt_bn128 fr
void oracle_lde(int field, void* inout, unsigned lg_domain, unsigned lg_blowup, void* aux)
{
    switch (field) {
        case 0: lde(reinterpret_cast<gl64*>(inout), lg_domain, lg_blowup, reinterpret_cast<gl64*>(aux)); break;
        case 1: lde(reinterpret_cast<bb31*>(inout), lg_domain, lg_blowup, reinterpret_cast<bb31*>(aux)); break;
        case 2: lde(reinterpret_cast<bls12_381_fr*>(inout), lg_domain, lg_blowup, reinterpret_cast<bls12_381_fr*>(aux)); break;
        default: lde(reinterpret_cast<alt_bn128_fr*>(inout), lg_domain, lg_blowup, reinterpret_cast<alt_bn128_fr*>(aux)); break;
    }
}
void oracle_lde_powers(int field, void* inout, unsigned lg)
{
    switch (field) {
        case 0: lde_powers_bitrev(reinterpret_cast<gl64*>(inout), lg); break;
        case 1: lde_powers_bitrev(reinterpret_cast<bb31*>(inout), lg); break;
        case 2: lde_powers_bitrev(reinterpret_cast<bls12_381_fr*>(inout), lg); break;
        default: lde_powers_bitrev(reinterpret_cast<alt_bn128_fr*>(inout), lg); break;
    }
}
void oracle_lde_expand(int field, void* out, const void* in, unsigned lg_domain, unsigned lg_blowup)
{
    switch (field) {
        case 0: lde_expand(reinterpret_cast<gl64*>(out), reinterpret_cast<const gl64*>(in), lg_domain, lg_blowup); break;
        case 1: lde_expand(reinterpret_cast<bb31*>(out), reinterpret_cast<const bb31*>(in), lg_domain, lg_blowup); break;
        case 2: lde_expand(reinterpret_cast<bls12_381_fr*>(out), reinterpret_cast<const bls12_381_fr*>(in), lg_domain, lg_blowup); break;
        default: lde_expand(reinterpret_cast<alt_bn128_fr*>(out), reinterpret_cast<const alt_bn128_fr*>(in), lg_domain, lg_blowup); break;
    }
}

uint64_t oracle_gl64_root(unsigned lg) { return root_of_unity<gl64>(lg).raw(); }
uint32_t oracle_bb31_root(unsigned lg) { return root_of_unity<bb31>(lg).raw(); }
uint64_t oracle_gl64_mul(uint64_t a, uint64_t b) { return (gl64::from_raw(a) * gl64::from_raw(b)).raw(); }
uint32_t oracle_bb31_mul(uint32_t a, uint32_t b) { return (bb31::from_raw(a) * bb31::from_raw(b)).raw(); }
uint32_t oracle_bb31_to_mont(uint32_t a) { return bb31::from_canonical(a).raw(); }
uint32_t oracle_bb31_from_mont(uint32_t a) { return bb31::from_raw(a).to_canonical(); }

} // extern "C"
