// ORACLE — TEST INFRASTRUCTURE ONLY (see ff.hpp header).
// C entry points (ctypes) over the CPU restatement: field ops, G1 helpers,
// the three MSM evaluators and the NTT.  Built by oracle/Makefile into
// oracle/liboracle.so.
#include "ff.hpp"
#include "ec.hpp"
#include "msm.hpp"
#include "ntt.hpp"
#include "poly.hpp"
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
        memcpy((void*)&a.X, p, fp_bytes); memcpy((void*)&a.Y, p + fp_bytes, fp_bytes);
        // flagged wire format (ec/affine_t.hpp:64-122, Affine_inf_t): the byte
        // after Y is the infinity flag; the plain format encodes inf as X=Y=0.
        if (stride > 2 * fp_bytes && (p[2 * fp_bytes] & 1)) a.set_inf();
        return a;
    }
    static void store_affine(unsigned char* p, const affine<FP>& a, size_t stride)
    {
        memset(p, 0, stride);
        memcpy(p, &a.X, fp_bytes); memcpy(p + fp_bytes, &a.Y, fp_bytes);
        if (stride > 2 * fp_bytes) p[2 * fp_bytes] = a.is_inf();
    }
    static jacobian<FP> load_jac(const unsigned char* p)
    {
        jacobian<FP> j;
        memcpy((void*)&j.X, p, fp_bytes); memcpy((void*)&j.Y, p + fp_bytes, fp_bytes);
        memcpy((void*)&j.Z, p + 2 * fp_bytes, fp_bytes);
        return j;
    }
    static void store_jac(unsigned char* p, const jacobian<FP>& j)
    {
        memcpy(p, &j.X, fp_bytes); memcpy(p + fp_bytes, &j.Y, fp_bytes);
        memcpy(p + 2 * fp_bytes, &j.Z, fp_bytes);
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

// G2: the twists over Fp2 = Fp[u]/(u^2 + 1).  Standard generators (checked on-curve by
// tests/test_oracle.py and, independently, with Python big-ints in tests/golden/make_golden.py)
template<class FP2, class FP> static FP2 fp2_hex(const char* c0, const char* c1)
{   FP2 r; r.c0 = from_hex<FP>(c0); r.c1 = from_hex<FP>(c1); return r;   }
struct bls12_381_g2 : curve_t<bls12_381_fp2, bls12_381_fr> {
    static affine<fp> generator()
    {
        affine<fp> g;
        g.X = fp2_hex<fp, bls12_381_fp>("024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8",
                                        "13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e");
        g.Y = fp2_hex<fp, bls12_381_fp>("0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801",
                                        "0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be");
        return g;
    }
    static fp b() { return fp2_hex<fp, bls12_381_fp>("4", "4"); }                 // y^2 = x^3 + 4(1 + u)
};
struct alt_bn128_g2 : curve_t<alt_bn128_fp2, alt_bn128_fr> {
    static affine<fp> generator()
    {
        affine<fp> g;
        g.X = fp2_hex<fp, alt_bn128_fp>("1800deef121f1e76426a00665e5c4479674322d4f75edadd46debd5cd992f6ed",
                                        "198e9393920d483a7260bfb731fb5d25f1aa493335a9e71297e485b7aef312c2");
        g.Y = fp2_hex<fp, alt_bn128_fp>("12c85ea5db8c6deb4aab71808dcb408fe3d1e7690c43d37b4ce6cc0166fa7daa",
                                        "090689d0585ff075ec9e99ad690c3395bc4b313370b38ef355acdadcd122975b");
        return g;
    }
    static fp b()                                                               // y^2 = x^3 + 3/(9 + u)
    {   return fp2_hex<fp, alt_bn128_fp>("2b149d40ceb8aaae81be18991be06ac3b5b4c5e559dbefa33267e6dc24a138e5",
                                         "009713b03af0fed4cd2cafadeed8fdf4a74fa084e52d1852e4a2bd0685c315d2");   }
};

// BLS12-377 (ff/bls12-377.hpp): y^2 = x^3 + 1; the twist over Fp2 = Fp[u]/(u^2 + 5) is y^2 = x^3 + 1/u.
// G1: the standard generator (ark-bls12-377); G2: a point of order r derived and checked with Python
// big-ints (cofactor clearing of the first twist point with x = k + u)
struct bls12_377_g1 : curve_t<bls12_377_fp, bls12_377_fr> {
    static affine<fp> generator()
    {
        affine<fp> g;
        g.X = from_hex<fp>("008848defe740a67c8fc6225bf87ff5485951e2caa9d41bb188282c8bd37cb5cd5481512ffcd394eeab9b16eb21be9ef");
        g.Y = from_hex<fp>("01914a69c5102eff1f674f5d30afeec4bd7fb348ca3e52d96d182ad44fb82305c2fe3d3634a9591afd82de55559c8ea6");
        return g;
    }
    static fp b() { return from_hex<fp>("1"); }
};
struct bls12_377_g2 : curve_t<bls12_377_fp2, bls12_377_fr> {
    static affine<fp> generator()
    {
        affine<fp> g;
        g.X = fp2_hex<fp, bls12_377_fp>("6f72205595a839df693176b247c2fa251f7e02a29061e50540dc9e1c2bf1957bf1bab2288c257c2cb36b58f2418bc9",
                                        "138c24b2b4e17888beed0a9802aac837cdea39890effe00072f754ecb0152dd6cb524f281298966dbaeca23d3e462b8");
        g.Y = fp2_hex<fp, bls12_377_fp>("16235fdea6c3faf2a83d3730f6ab2c033ef6c2739002946f7dc48e4688bca1af1c9b417d58220817e0dc644b5e7d916",
                                        "707ac6cc7d192827fc54eb83267f3bed8511bd3c74f63a1ea75eabb66476769c8786f2af2a75166f33142379b4963c");
        return g;
    }
    static fp b()                                                               // (0, -1/5) = 1/u
    {   return fp2_hex<fp, bls12_377_fp>("0", "010222f6db0fd6f343bd03737460c589dc7b4f91cd5fd889129207b63c6bf8000dd39e5c1ccccccd1c9ed9999999999a");   }
};

// The Pasta cycle (ff/pasta.hpp): y^2 = x^3 + 5 over each of the two fields, generator (-1, 2);
// Pallas over pasta_p with scalars in pasta_q, Vesta the other way round.
struct pallas_g1 : curve_t<pasta_p, pasta_q> {
    static affine<fp> generator()
    {   affine<fp> g; g.X = from_hex<fp>("1"); g.X.cneg(true); g.Y = from_hex<fp>("2"); return g;   }
    static fp b() { return from_hex<fp>("5"); }
};
struct vesta_g1 : curve_t<pasta_q, pasta_p> {
    static affine<fp> generator()
    {   affine<fp> g; g.X = from_hex<fp>("1"); g.X.cneg(true); g.Y = from_hex<fp>("2"); return g;   }
    static fp b() { return from_hex<fp>("5"); }
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
    memcpy((void*)&p.X, in, fb); memcpy((void*)&p.Y, in + fb, fb);
    memcpy((void*)&p.ZZZ, in + 2 * fb, fb); memcpy((void*)&p.ZZ, in + 3 * fb, fb);
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

template<class C> void store_generator(unsigned char* out, size_t stride)
{   C::store_affine(out, C::generator(), stride);   }

} // namespace

// curve: 0 BLS12-381 G1, 1 alt_bn128 G1, 2 BLS12-381 G2, 3 alt_bn128 G2, 4 BLS12-377 G1, 5 BLS12-377 G2, 6 Pallas, 7 Vesta
#define CURVE_DO(curve, FN, ...)                                                         \
    switch (curve) {                                                                     \
        case 0: FN<bls12_381_g1>(__VA_ARGS__); break; case 1: FN<alt_bn128_g1>(__VA_ARGS__); break; \
        case 2: FN<bls12_381_g2>(__VA_ARGS__); break; case 3: FN<alt_bn128_g2>(__VA_ARGS__); break; \
        case 4: FN<bls12_377_g1>(__VA_ARGS__); break; case 5: FN<bls12_377_g2>(__VA_ARGS__); break; \
        case 6: FN<pallas_g1>(__VA_ARGS__); break; case 7: FN<vesta_g1>(__VA_ARGS__); break; \
        default: return -1;                                                              \
    }
#define CURVE_RET(curve, FN, ...)                                                        \
    switch (curve) {                                                                     \
        case 0: return FN<bls12_381_g1>(__VA_ARGS__); case 1: return FN<alt_bn128_g1>(__VA_ARGS__); \
        case 2: return FN<bls12_381_g2>(__VA_ARGS__); case 3: return FN<alt_bn128_g2>(__VA_ARGS__); \
        case 4: return FN<bls12_377_g1>(__VA_ARGS__); case 5: return FN<bls12_377_g2>(__VA_ARGS__); \
        case 6: return FN<pallas_g1>(__VA_ARGS__); case 7: return FN<vesta_g1>(__VA_ARGS__); \
        default: return -1;                                                              \
    }

extern "C" {

// field: 0 bls12_381 fp, 1 bls12_381 fr, 2 alt_bn128 fp, 3 alt_bn128 fr, 4 bls12_377 fp, 5 bls12_377 fr
// op: 0 add 1 sub 2 montmul 3 inverse 4 to_mont 5 from_mont 6 neg 7 sqr
int oracle_field_op(int field, int op, uint64_t* out, const uint64_t* a, const uint64_t* b)
{
    switch (field) {
        case 0: return field_op<bls12_381_fp>(op, out, a, b);
        case 1: return field_op<bls12_381_fr>(op, out, a, b);
        case 2: return field_op<alt_bn128_fp>(op, out, a, b);
        case 3: return field_op<alt_bn128_fr>(op, out, a, b);
        case 4: return field_op<bls12_377_fp>(op, out, a, b);
        case 5: return field_op<bls12_377_fr>(op, out, a, b);
        case 6: return field_op<pasta_p>(op, out, a, b);
        case 7: return field_op<pasta_q>(op, out, a, b);
    }
    return -1;
}

int oracle_g1_generator(int curve, unsigned char* out, size_t stride)
{   CURVE_DO(curve, store_generator, out, stride); return 0;   }

// P_i = k_i * G with k_i from splitmix64(seed), 253-bit
int oracle_g1_gen_points(int curve, unsigned char* out, size_t stride, size_t n, uint64_t seed)
{   CURVE_DO(curve, gen_points, out, stride, n, seed); return 0;   }

int oracle_g1_mul(int curve, unsigned char* out, const unsigned char* in, size_t stride, const unsigned char* scalar_le)
{   CURVE_DO(curve, g1_mul, out, in, stride, scalar_le); return 0;   }

int oracle_g1_on_curve(int curve, const unsigned char* in, size_t stride)
{   CURVE_RET(curve, on_curve_impl, in, stride);   }

int oracle_jac_to_affine(int curve, unsigned char* out, const unsigned char* in_jac, size_t stride)
{   CURVE_DO(curve, jac_to_affine_impl, out, in_jac, stride); return 0;   }

int oracle_xyzz_to_affine(int curve, unsigned char* out, const unsigned char* in_xyzz, size_t stride)
{   CURVE_DO(curve, xyzz_to_affine_impl, out, in_xyzz, stride); return 0;   }

int oracle_jac_add(int curve, unsigned char* out, const unsigned char* a, const unsigned char* b)
{   CURVE_DO(curve, jac_add_impl, out, a, b); return 0;   }

int oracle_jac_dbl(int curve, unsigned char* out, const unsigned char* a)
{   CURVE_DO(curve, jac_dbl_impl, out, a); return 0;   }

int oracle_jac_eq(int curve, const unsigned char* a, const unsigned char* b)
{   CURVE_RET(curve, jac_eq_impl, a, b);   }

// algo 0: restated msm/pippenger.hpp with |param| = ncpus (0/1 -> serial path)
// algo 1: naive sum of double-and-add
// algo 2: signed-window model of the GPU semantics with |param| = window bits
// out_jac: X|Y|Z Montgomery limbs (144 B BLS12-381 G1, 96 B alt_bn128 G1, 288 / 192 B for G2)
int oracle_msm(int curve, int algo, unsigned char* out_jac, const unsigned char* points,
               size_t stride, size_t npoints, const unsigned char* scalars, int mont, size_t param)
{   CURVE_RET(curve, msm_dispatch, algo, out_jac, points, stride, npoints, scalars, mont, param);   }

void oracle_ntt_gl64(uint64_t* inout, unsigned lg, int order, int direction, int type)
{   ntt(reinterpret_cast<gl64*>(inout), lg, order, direction, type);   }
void oracle_ntt_bb31(uint32_t* inout, unsigned lg, int order, int direction, int type)
{   ntt(reinterpret_cast<bb31*>(inout), lg, order, direction, type);   }
void oracle_ntt_naive_gl64(uint64_t* out, const uint64_t* in, unsigned lg, int inv)
{   ntt_naive(reinterpret_cast<gl64*>(out), reinterpret_cast<const gl64*>(in), lg, inv != 0);   }
void oracle_ntt_naive_bb31(uint32_t* out, const uint32_t* in, unsigned lg, int inv)
{   ntt_naive(reinterpret_cast<bb31*>(out), reinterpret_cast<const bb31*>(in), lg, inv != 0);   }

// 256-bit scalar fields: field 0 = BLS12-381 Fr, 1 = alt_bn128 Fr, 4 = BLS12-377 Fr (the curve ids); elements are 4 x u64 Montgomery limbs
void oracle_ntt_fr(int field, uint64_t* inout, unsigned lg, int order, int direction, int type)
{
    if (field == 0)      ntt(reinterpret_cast<bls12_381_fr*>(inout), lg, order, direction, type);
    else if (field == 4) ntt(reinterpret_cast<bls12_377_fr*>(inout), lg, order, direction, type);
    else if (field == 6) ntt(reinterpret_cast<pasta_q*>(inout), lg, order, direction, type);       // Pallas: Fr = pasta_q
    else if (field == 7) ntt(reinterpret_cast<pasta_p*>(inout), lg, order, direction, type);       // Vesta:  Fr = pasta_p
    else                 ntt(reinterpret_cast<alt_bn128_fr*>(inout), lg, order, direction, type);
}
void oracle_ntt_naive_fr(int field, uint64_t* out, const uint64_t* in, unsigned lg, int inv)
{
    if (field == 0)      ntt_naive(reinterpret_cast<bls12_381_fr*>(out), reinterpret_cast<const bls12_381_fr*>(in), lg, inv != 0);
    else if (field == 4) ntt_naive(reinterpret_cast<bls12_377_fr*>(out), reinterpret_cast<const bls12_377_fr*>(in), lg, inv != 0);
    else if (field == 6) ntt_naive(reinterpret_cast<pasta_q*>(out), reinterpret_cast<const pasta_q*>(in), lg, inv != 0);
    else if (field == 7) ntt_naive(reinterpret_cast<pasta_p*>(out), reinterpret_cast<const pasta_p*>(in), lg, inv != 0);
    else                 ntt_naive(reinterpret_cast<alt_bn128_fr*>(out), reinterpret_cast<const alt_bn128_fr*>(in), lg, inv != 0);
}
void oracle_fr_root(int field, uint64_t* out, unsigned lg)
{
    if (field == 0) { auto w = root_of_unity<bls12_381_fr>(lg); memcpy(out, w.v, 32); }
    else if (field == 4) { auto w = root_of_unity<bls12_377_fr>(lg); memcpy(out, w.v, 32); }
    else if (field == 6) { auto w = root_of_unity<pasta_q>(lg); memcpy(out, w.v, 32); }
    else if (field == 7) { auto w = root_of_unity<pasta_p>(lg); memcpy(out, w.v, 32); }

    else            { auto w = root_of_unity<alt_bn128_fr>(lg); memcpy(out, w.v, 32); }
}
// LDE family; field: 0 gl64, 1 bb31, 2 bls12_381 fr, 3 alt_bn128 fr
void oracle_lde(int field, void* inout, unsigned lg_domain, unsigned lg_blowup, void* aux)
{
    switch (field) {
        case 0: lde(reinterpret_cast<gl64*>(inout), lg_domain, lg_blowup, reinterpret_cast<gl64*>(aux)); break;
        case 1: lde(reinterpret_cast<bb31*>(inout), lg_domain, lg_blowup, reinterpret_cast<bb31*>(aux)); break;
        case 2: lde(reinterpret_cast<bls12_381_fr*>(inout), lg_domain, lg_blowup, reinterpret_cast<bls12_381_fr*>(aux)); break;
        case 4: lde(reinterpret_cast<bls12_377_fr*>(inout), lg_domain, lg_blowup, reinterpret_cast<bls12_377_fr*>(aux)); break;
        case 6: lde(reinterpret_cast<pasta_q*>(inout), lg_domain, lg_blowup, reinterpret_cast<pasta_q*>(aux)); break;
        case 7: lde(reinterpret_cast<pasta_p*>(inout), lg_domain, lg_blowup, reinterpret_cast<pasta_p*>(aux)); break;
        default: lde(reinterpret_cast<alt_bn128_fr*>(inout), lg_domain, lg_blowup, reinterpret_cast<alt_bn128_fr*>(aux)); break;
    }
}
void oracle_lde_powers(int field, void* inout, unsigned lg)
{
    switch (field) {
        case 0: lde_powers_bitrev(reinterpret_cast<gl64*>(inout), lg); break;
        case 1: lde_powers_bitrev(reinterpret_cast<bb31*>(inout), lg); break;
        case 2: lde_powers_bitrev(reinterpret_cast<bls12_381_fr*>(inout), lg); break;
        case 4: lde_powers_bitrev(reinterpret_cast<bls12_377_fr*>(inout), lg); break;
        case 6: lde_powers_bitrev(reinterpret_cast<pasta_q*>(inout), lg); break;
        case 7: lde_powers_bitrev(reinterpret_cast<pasta_p*>(inout), lg); break;
        default: lde_powers_bitrev(reinterpret_cast<alt_bn128_fr*>(inout), lg); break;
    }
}
void oracle_lde_expand(int field, void* out, const void* in, unsigned lg_domain, unsigned lg_blowup)
{
    switch (field) {
        case 0: lde_expand(reinterpret_cast<gl64*>(out), reinterpret_cast<const gl64*>(in), lg_domain, lg_blowup); break;
        case 1: lde_expand(reinterpret_cast<bb31*>(out), reinterpret_cast<const bb31*>(in), lg_domain, lg_blowup); break;
        case 2: lde_expand(reinterpret_cast<bls12_381_fr*>(out), reinterpret_cast<const bls12_381_fr*>(in), lg_domain, lg_blowup); break;
        case 4: lde_expand(reinterpret_cast<bls12_377_fr*>(out), reinterpret_cast<const bls12_377_fr*>(in), lg_domain, lg_blowup); break;
        case 6: lde_expand(reinterpret_cast<pasta_q*>(out), reinterpret_cast<const pasta_q*>(in), lg_domain, lg_blowup); break;
        case 7: lde_expand(reinterpret_cast<pasta_p*>(out), reinterpret_cast<const pasta_p*>(in), lg_domain, lg_blowup); break;
        default: lde_expand(reinterpret_cast<alt_bn128_fr*>(out), reinterpret_cast<const alt_bn128_fr*>(in), lg_domain, lg_blowup); break;
    }
}

// polynomial primitives; field: 0 gl64, 1 bb31, 2 bls12_381 fr, 3 alt_bn128 fr, 4 bls12_377 fr, 6 / 7 Pasta, 8 Mersenne31, 9 bb31_4
#define POLY_DISPATCH(field, CALL)                                          \
    switch (field) {                                                        \
        case 0: { typedef gl64 F; CALL; } break;                            \
        case 1: { typedef bb31 F; CALL; } break;                            \
        case 2: { typedef bls12_381_fr F; CALL; } break;                    \
        case 4: { typedef bls12_377_fr F; CALL; } break;                    \
        case 6: { typedef pasta_q F; CALL; } break;                         \
        case 7: { typedef pasta_p F; CALL; } break;                         \
        case 8: { typedef mrs31 F; CALL; } break;                           \
        case 9: { typedef bb31_4 F; CALL; } break;                          \
        default: { typedef alt_bn128_fr F; CALL; } break;                   \
    }
void oracle_prefix_op(int field, void* out, const void* in, size_t len, int op)
{   POLY_DISPATCH(field, prefix_op((F*)out, (const F*)in, len, op))   }
void oracle_poly_evaluate(int field, void* ret, const void* x, size_t n, const void* coeffs, size_t len)
{   POLY_DISPATCH(field, poly_evaluate((F*)ret, (const F*)x, n, (const F*)coeffs, len))   }
void oracle_div_by_x_minus_z(int field, void* inout, size_t len, const void* z, int rotate)
{   POLY_DISPATCH(field, div_by_x_minus_z((F*)inout, len, *(const F*)z, rotate != 0))   }

// compile-time root conventions of the reference, as a run-time switch of the oracle:
// GOLDILOCKS_PLONKY2 (ntt/parameters/goldilocks.h:7-82), BABY_BEAR_CANONICAL (baby_bear.h:7-74)
void oracle_set_root_conventions(int goldilocks_plonky2, int baby_bear_canonical)
{   gl64::plonky2() = goldilocks_plonky2 != 0; bb31::canonical_roots() = baby_bear_canonical != 0;   }

uint64_t oracle_gl64_root(unsigned lg) { return root_of_unity<gl64>(lg).raw(); }
uint32_t oracle_bb31_root(unsigned lg) { return root_of_unity<bb31>(lg).raw(); }
uint64_t oracle_gl64_mul(uint64_t a, uint64_t b) { return (gl64::from_raw(a) * gl64::from_raw(b)).raw(); }
uint32_t oracle_bb31_mul(uint32_t a, uint32_t b) { return (bb31::from_raw(a) * bb31::from_raw(b)).raw(); }
uint32_t oracle_bb31_to_mont(uint32_t a) { return bb31::from_canonical(a).raw(); }
uint32_t oracle_bb31_from_mont(uint32_t a) { return bb31::from_raw(a).to_canonical(); }

} // extern "C"
