// Field parameter sets for the device (32-bit limbs) and host (64-bit limbs)
// Montgomery classes.  The numbers are fixed by the curves and by the
// reference's choice of Montgomery radix R = 2^(32*N); they are the values of
// the reference tables cited per struct and are re-derived from Python
// big-ints in tests/test_params.py.
#pragma once
#include <stdint.h>

namespace sppark_amd {

#define SPPARK_LO(x) (uint32_t)((x) & 0xffffffffu)
#define SPPARK_HI(x) (uint32_t)((uint64_t)(x) >> 32)
#define SPPARK_L2(x) SPPARK_LO(x), SPPARK_HI(x)

// BLS12-381 base field, ff/bls12-381.hpp:13-33 (device) / :100-116 (host)
struct bls12_381_fp_p {
    static constexpr int N = 12, N64 = 6, NBITS = 381;
    static constexpr unsigned FP2_NR = 1;               // Fp2 = Fp[u]/(u^2 + 1)  (ff/bls12-381-fp2.hpp)
    static constexpr uint64_t MOD64[6] = {
        0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
        0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL };
    static constexpr uint64_t RR64[6] = {           // 2^768 mod p
        0xf4df1f341c341746ULL, 0x0a76e6a609d104f1ULL, 0x8de5476c4c95b6d5ULL,
        0x67eb88a9939d83c0ULL, 0x9a793e85b519952dULL, 0x11988fe592cae3aaULL };
    static constexpr uint64_t ONE64[6] = {          // 2^384 mod p
        0x760900000002fffdULL, 0xebf4000bc40c0002ULL, 0x5f48985753c758baULL,
        0x77ce585370525745ULL, 0x5c071a97a256ec6dULL, 0x15f65ec3fa80e493ULL };
    static constexpr uint64_t M0_64 = 0x89f3fffcfffcfffdULL;   // -1/p mod 2^64
    static constexpr uint32_t M0 = 0xfffcfffdu;                // -1/p mod 2^32
    static constexpr uint32_t MOD[12] = {
        SPPARK_L2(MOD64[0]), SPPARK_L2(MOD64[1]), SPPARK_L2(MOD64[2]),
        SPPARK_L2(MOD64[3]), SPPARK_L2(MOD64[4]), SPPARK_L2(MOD64[5]) };
    static constexpr uint32_t RR[12] = {
        SPPARK_L2(RR64[0]), SPPARK_L2(RR64[1]), SPPARK_L2(RR64[2]),
        SPPARK_L2(RR64[3]), SPPARK_L2(RR64[4]), SPPARK_L2(RR64[5]) };
    static constexpr uint32_t ONE[12] = {
        SPPARK_L2(ONE64[0]), SPPARK_L2(ONE64[1]), SPPARK_L2(ONE64[2]),
        SPPARK_L2(ONE64[3]), SPPARK_L2(ONE64[4]), SPPARK_L2(ONE64[5]) };
};

// BLS12-381 scalar field, ff/bls12-381.hpp:35-51 / :125-138
struct bls12_381_fr_p {
    static constexpr int N = 8, N64 = 4, NBITS = 255;
    // NTT: group_gen = 7, roots[k] = 7^((r-1)/2^k), S = 32 (ntt/parameters/bls12_381.h:11-14)
    static constexpr unsigned TWO_ADICITY = 32, GROUP_GEN = 7;
    static constexpr uint64_t MOD64[4] = {
        0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL };
    static constexpr uint64_t RR64[4] = {           // 2^512 mod r
        0xc999e990f3f29c6dULL, 0x2b6cedcb87925c23ULL, 0x05d314967254398fULL, 0x0748d9d99f59ff11ULL };
    static constexpr uint64_t ONE64[4] = {          // 2^256 mod r
        0x00000001fffffffeULL, 0x5884b7fa00034802ULL, 0x998c4fefecbc4ff5ULL, 0x1824b159acc5056fULL };
    static constexpr uint64_t M0_64 = 0xfffffffeffffffffULL;
    static constexpr uint32_t M0 = 0xffffffffu;
    static constexpr uint32_t MOD[8] = {
        SPPARK_L2(MOD64[0]), SPPARK_L2(MOD64[1]), SPPARK_L2(MOD64[2]), SPPARK_L2(MOD64[3]) };
    static constexpr uint32_t RR[8] = {
        SPPARK_L2(RR64[0]), SPPARK_L2(RR64[1]), SPPARK_L2(RR64[2]), SPPARK_L2(RR64[3]) };
    static constexpr uint32_t ONE[8] = {
        SPPARK_L2(ONE64[0]), SPPARK_L2(ONE64[1]), SPPARK_L2(ONE64[2]), SPPARK_L2(ONE64[3]) };
};

// alt_bn128 (BN254) base field, ff/alt_bn128.hpp:13-30 / :88-101
struct alt_bn128_fp_p {
    static constexpr int N = 8, N64 = 4, NBITS = 254;
    static constexpr unsigned FP2_NR = 1;               // Fp2 = Fp[u]/(u^2 + 1)  (ff/alt_bn128-fp2.hpp)
    static constexpr uint64_t MOD64[4] = {
        0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL };
    static constexpr uint64_t RR64[4] = {
        0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL };
    static constexpr uint64_t ONE64[4] = {
        0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL };
    static constexpr uint64_t M0_64 = 0x87d20782e4866389ULL;
    static constexpr uint32_t M0 = 0xe4866389u;
    static constexpr uint32_t MOD[8] = {
        SPPARK_L2(MOD64[0]), SPPARK_L2(MOD64[1]), SPPARK_L2(MOD64[2]), SPPARK_L2(MOD64[3]) };
    static constexpr uint32_t RR[8] = {
        SPPARK_L2(RR64[0]), SPPARK_L2(RR64[1]), SPPARK_L2(RR64[2]), SPPARK_L2(RR64[3]) };
    static constexpr uint32_t ONE[8] = {
        SPPARK_L2(ONE64[0]), SPPARK_L2(ONE64[1]), SPPARK_L2(ONE64[2]), SPPARK_L2(ONE64[3]) };
};

// alt_bn128 scalar field, ff/alt_bn128.hpp:32-48 / :111-124
struct alt_bn128_fr_p {
    static constexpr int N = 8, N64 = 4, NBITS = 254;
    // NTT: group_gen = 5, roots[k] = 5^((r-1)/2^k), S = 28 (ntt/parameters/alt_bn128.h:11-14)
    static constexpr unsigned TWO_ADICITY = 28, GROUP_GEN = 5;
    static constexpr uint64_t MOD64[4] = {
        0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL };
    static constexpr uint64_t RR64[4] = {
        0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL };
    static constexpr uint64_t ONE64[4] = {
        0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL };
    static constexpr uint64_t M0_64 = 0xc2e1f593efffffffULL;
    static constexpr uint32_t M0 = 0xefffffffu;
    static constexpr uint32_t MOD[8] = {
        SPPARK_L2(MOD64[0]), SPPARK_L2(MOD64[1]), SPPARK_L2(MOD64[2]), SPPARK_L2(MOD64[3]) };
    static constexpr uint32_t RR[8] = {
        SPPARK_L2(RR64[0]), SPPARK_L2(RR64[1]), SPPARK_L2(RR64[2]), SPPARK_L2(RR64[3]) };
    static constexpr uint32_t ONE[8] = {
        SPPARK_L2(ONE64[0]), SPPARK_L2(ONE64[1]), SPPARK_L2(ONE64[2]), SPPARK_L2(ONE64[3]) };
};

// BLS12-377 base field, ff/bls12-377.hpp:13-33
struct bls12_377_fp_p {
    static constexpr int N = 12, N64 = 6, NBITS = 377;
    static constexpr unsigned FP2_NR = 5;               // Fp2 = Fp[u]/(u^2 + 5)  (ff/bls12-377-fp2.hpp)
    static constexpr uint64_t MOD64[6] = {
        0x8508c00000000001ULL, 0x170b5d4430000000ULL, 0x1ef3622fba094800ULL, 0x1a22d9f300f5138fULL, 0xc63b05c06ca1493bULL, 0x01ae3a4617c510eaULL };
    static constexpr uint64_t RR64[6] = {
        0xb786686c9400cd22ULL, 0x0329fcaab00431b1ULL, 0x22a5f11162d6b46dULL, 0xbfdf7d03827dc3acULL, 0x837e92f041790bf9ULL, 0x006dfccb1e914b88ULL };
    static constexpr uint64_t ONE64[6] = {
        0x02cdffffffffff68ULL, 0x51409f837fffffb1ULL, 0x9f7db3a98a7d3ff2ULL, 0x7b4e97b76e7c6305ULL, 0x4cf495bf803c84e8ULL, 0x008d6661e2fdf49aULL };
    static constexpr uint64_t M0_64 = 0x8508bfffffffffffULL;
    static constexpr uint32_t M0 = 0xffffffffu;
    static constexpr uint32_t MOD[12] = { SPPARK_L2(MOD64[0]), SPPARK_L2(MOD64[1]), SPPARK_L2(MOD64[2]), SPPARK_L2(MOD64[3]), SPPARK_L2(MOD64[4]), SPPARK_L2(MOD64[5]) };
    static constexpr uint32_t RR[12] = { SPPARK_L2(RR64[0]), SPPARK_L2(RR64[1]), SPPARK_L2(RR64[2]), SPPARK_L2(RR64[3]), SPPARK_L2(RR64[4]), SPPARK_L2(RR64[5]) };
    static constexpr uint32_t ONE[12] = { SPPARK_L2(ONE64[0]), SPPARK_L2(ONE64[1]), SPPARK_L2(ONE64[2]), SPPARK_L2(ONE64[3]), SPPARK_L2(ONE64[4]), SPPARK_L2(ONE64[5]) };
};
// BLS12-377 scalar field, ff/bls12-377.hpp:35-51
struct bls12_377_fr_p {
    static constexpr int N = 8, N64 = 4, NBITS = 253;
    // NTT: group_gen = 22, roots[k] = 22^((r-1)/2^k), S = 47 (ntt/parameters/bls12_377.h:11-16)
    static constexpr unsigned TWO_ADICITY = 47, GROUP_GEN = 22;
    static constexpr uint64_t MOD64[4] = {
        0x0a11800000000001ULL, 0x59aa76fed0000001ULL, 0x60b44d1e5c37b001ULL, 0x12ab655e9a2ca556ULL };
    static constexpr uint64_t RR64[4] = {
        0x25d577bab861857bULL, 0xcc2c27b58860591fULL, 0xa7cc008fe5dc8593ULL, 0x011fdae7eff1c939ULL };
    static constexpr uint64_t ONE64[4] = {
        0x7d1c7ffffffffff3ULL, 0x7257f50f6ffffff2ULL, 0x16d81575512c0feeULL, 0x0d4bda322bbb9a9dULL };
    static constexpr uint64_t M0_64 = 0x0a117fffffffffffULL;
    static constexpr uint32_t M0 = 0xffffffffu;
    static constexpr uint32_t MOD[8] = { SPPARK_L2(MOD64[0]), SPPARK_L2(MOD64[1]), SPPARK_L2(MOD64[2]), SPPARK_L2(MOD64[3]) };
    static constexpr uint32_t RR[8] = { SPPARK_L2(RR64[0]), SPPARK_L2(RR64[1]), SPPARK_L2(RR64[2]), SPPARK_L2(RR64[3]) };
    static constexpr uint32_t ONE[8] = { SPPARK_L2(ONE64[0]), SPPARK_L2(ONE64[1]), SPPARK_L2(ONE64[2]), SPPARK_L2(ONE64[3]) };
};

// the Pallas base field = the Vesta scalar field (ff/pasta.hpp: Pallas_P; roots: ntt/parameters/pallas.h)
struct pasta_p_p {
    static constexpr int N = 8, N64 = 4, NBITS = 255;
    static constexpr unsigned FP2_NR = 1;               // (no pairing, no G2: unused)
    // NTT: group_gen = 5, roots[k] = 5^((p-1)/2^k), S = 32
    static constexpr unsigned TWO_ADICITY = 32, GROUP_GEN = 5;
    static constexpr uint64_t MOD64[4] = {
        0x992d30ed00000001ULL, 0x224698fc094cf91bULL, 0x0000000000000000ULL, 0x4000000000000000ULL };
    static constexpr uint64_t RR64[4] = {
        0x8c78ecb30000000fULL, 0xd7d30dbd8b0de0e7ULL, 0x7797a99bc3c95d18ULL, 0x096d41af7b9cb714ULL };
    static constexpr uint64_t ONE64[4] = {
        0x34786d38fffffffdULL, 0x992c350be41914adULL, 0xffffffffffffffffULL, 0x3fffffffffffffffULL };
    static constexpr uint64_t M0_64 = 0x992d30ecffffffffULL;
    static constexpr uint32_t M0 = 0xffffffffu;
    static constexpr uint32_t MOD[8] = { SPPARK_L2(MOD64[0]), SPPARK_L2(MOD64[1]), SPPARK_L2(MOD64[2]), SPPARK_L2(MOD64[3]) };
    static constexpr uint32_t RR[8] = { SPPARK_L2(RR64[0]), SPPARK_L2(RR64[1]), SPPARK_L2(RR64[2]), SPPARK_L2(RR64[3]) };
    static constexpr uint32_t ONE[8] = { SPPARK_L2(ONE64[0]), SPPARK_L2(ONE64[1]), SPPARK_L2(ONE64[2]), SPPARK_L2(ONE64[3]) };
};
// the Vesta base field = the Pallas scalar field (ff/pasta.hpp: Vesta_P; roots: ntt/parameters/vesta.h)
struct pasta_q_p {
    static constexpr int N = 8, N64 = 4, NBITS = 255;
    static constexpr unsigned FP2_NR = 1;               // (no pairing, no G2: unused)
    // NTT: group_gen = 5, roots[k] = 5^((p-1)/2^k), S = 32
    static constexpr unsigned TWO_ADICITY = 32, GROUP_GEN = 5;
    static constexpr uint64_t MOD64[4] = {
        0x8c46eb2100000001ULL, 0x224698fc0994a8ddULL, 0x0000000000000000ULL, 0x4000000000000000ULL };
    static constexpr uint64_t RR64[4] = {
        0xfc9678ff0000000fULL, 0x67bb433d891a16e3ULL, 0x7fae231004ccf590ULL, 0x096d41af7ccfdaa9ULL };
    static constexpr uint64_t ONE64[4] = {
        0x5b2b3e9cfffffffdULL, 0x992c350be3420567ULL, 0xffffffffffffffffULL, 0x3fffffffffffffffULL };
    static constexpr uint64_t M0_64 = 0x8c46eb20ffffffffULL;
    static constexpr uint32_t M0 = 0xffffffffu;
    static constexpr uint32_t MOD[8] = { SPPARK_L2(MOD64[0]), SPPARK_L2(MOD64[1]), SPPARK_L2(MOD64[2]), SPPARK_L2(MOD64[3]) };
    static constexpr uint32_t RR[8] = { SPPARK_L2(RR64[0]), SPPARK_L2(RR64[1]), SPPARK_L2(RR64[2]), SPPARK_L2(RR64[3]) };
    static constexpr uint32_t ONE[8] = { SPPARK_L2(ONE64[0]), SPPARK_L2(ONE64[1]), SPPARK_L2(ONE64[2]), SPPARK_L2(ONE64[3]) };
};

// G1 curve descriptions: y^2 = x^3 + b, generator in Montgomery form.
// The reference stores no generator (its tests take points from arkworks);
// these are the standard generators, used only by the synthetic-input
// generator sppark_g1_generate().
struct bls12_381_g1_p {
    typedef bls12_381_fp_p fp; typedef bls12_381_fr_p fr;
    static constexpr uint64_t GX64[6] = {
        0x5cb38790fd530c16ULL, 0x7817fc679976fff5ULL, 0x154f95c7143ba1c1ULL,
        0xf0ae6acdf3d0e747ULL, 0xedce6ecc21dbf440ULL, 0x120177419e0bfb75ULL };
    static constexpr uint64_t GY64[6] = {
        0xbaac93d50ce72271ULL, 0x8c22631a7918fd8eULL, 0xdd595f13570725ceULL,
        0x51ac582950405194ULL, 0x0e1c8c3fad0059c0ULL, 0x0bbc3efc5008a26aULL };
};
struct alt_bn128_g1_p {
    typedef alt_bn128_fp_p fp; typedef alt_bn128_fr_p fr;
    static constexpr uint64_t GX64[4] = {       // 1 * R
        0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL };
    static constexpr uint64_t GY64[4] = {       // 2 * R
        0xa6ba871b8b1e1b3aULL, 0x14f1d651eb8e167bULL, 0xccdd46def0f28c58ULL, 0x1c14ef83340fbe5eULL };
};

// y^2 = x^3 + 1; the standard generator (arkworks ark-bls12-377 G1_GENERATOR_X/Y), Montgomery form
struct bls12_377_g1_p {
    typedef bls12_377_fp_p fp; typedef bls12_377_fr_p fr;
    static constexpr uint64_t GX64[6] = {
        0x260f33b9772451f4ULL, 0xc54dd773169d5658ULL, 0x5c1551c469a510ddULL, 0x761662e4425e1698ULL, 0xc97d78cc6f065272ULL, 0x00a41206b361fd4dULL };
    static constexpr uint64_t GY64[6] = {
        0x8193961fb8cb81f3ULL, 0x00638d4c5f44adb8ULL, 0xfafaf3dad4daf54aULL, 0xc27849e2d655cd18ULL, 0x2ec3ddb401d52814ULL, 0x007da93326303c71ULL };
};

// Pallas: y^2 = x^3 + 5 over pasta_p_p, generator (-1, 2), Montgomery form
struct pallas_g1_p {
    typedef pasta_p_p fp; typedef pasta_q_p fr;
    static constexpr uint64_t GX64[4] = {
        0x64b4c3b400000004ULL, 0x891a63f02533e46eULL, 0x0000000000000000ULL, 0x0000000000000000ULL };
    static constexpr uint64_t GY64[4] = {
        0xcfc3a984fffffff9ULL, 0x1011d11bbee5303eULL, 0xffffffffffffffffULL, 0x3fffffffffffffffULL };
};
// Vesta: y^2 = x^3 + 5 over pasta_q_p, generator (-1, 2), Montgomery form
struct vesta_g1_p {
    typedef pasta_q_p fp; typedef pasta_p_p fr;
    static constexpr uint64_t GX64[4] = {
        0x311bac8400000004ULL, 0x891a63f02652a376ULL, 0x0000000000000000ULL, 0x0000000000000000ULL };
    static constexpr uint64_t GY64[4] = {
        0x2a0f9218fffffff9ULL, 0x1011d11bbcef61f1ULL, 0xffffffffffffffffULL, 0x3fffffffffffffffULL };
};

} // namespace sppark_amd
