// Host field of the MSM tail (ff/mont_host.hpp): the adc-chain product / square against the plain CIOS loop (restated below as the
// checker), random + edge operands, and the time of the 255 doublings every MSM ends in.  Build + run: tools/jobs/r6_35_host_horner.sh
#include "ff/params.hpp"
#include "ff/mont_host.hpp"
#include "ec/jacobian_host.hpp"
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
using namespace sppark_amd;
template<class P> int run(const char* name, int iters){
  typedef mont_host<P> F;
  // reference product: slow CIOS restated here
  auto slow = [](const F& a, const F& b){
    typedef unsigned __int128 u128; constexpr int N = F::N;
    uint64_t t[N + 2] = {0};
    for (int i = 0; i < N; i++) {
        uint64_t c = 0;
        for (int j = 0; j < N; j++) { u128 s = (u128)a.v[j] * b.v[i] + t[j] + c; t[j] = (uint64_t)s; c = (uint64_t)(s >> 64); }
        u128 s = (u128)t[N] + c; t[N] = (uint64_t)s; t[N + 1] = (uint64_t)(s >> 64);
        uint64_t m = t[0] * P::M0_64;
        c = (uint64_t)(((u128)m * P::MOD64[0] + t[0]) >> 64);
        for (int j = 1; j < N; j++) { u128 q = (u128)m * P::MOD64[j] + t[j] + c; t[j - 1] = (uint64_t)q; c = (uint64_t)(q >> 64); }
        s = (u128)t[N] + c; t[N - 1] = (uint64_t)s; t[N] = t[N + 1] + (uint64_t)(s >> 64);
    }
    F r; F::cond_sub(r.v, t, t[N]); return r; };
  srand(1); int bad = 0;
  for (int it = 0; it < iters; it++) {
    F a, b;
    for (int i = 0; i < F::N; i++) { a.v[i] = ((uint64_t)rand() << 33) ^ ((uint64_t)rand() << 11) ^ rand(); b.v[i] = ((uint64_t)rand() << 33) ^ ((uint64_t)rand() << 11) ^ rand(); }
    if (it % 7 == 0) for (int i = 0; i < F::N; i++) a.v[i] = ~0ull;
    if (it % 11 == 0) for (int i = 0; i < F::N; i++) b.v[i] = ~0ull;
    // bring below MOD: a = a*1 style reduce through slow (inputs < 2^(64N) are fine for Montgomery with N words? keep canonical:)
    a = slow(a, F::one()); b = slow(b, F::one());
    if (it % 13 == 0) { for (int i = 0; i < F::N; i++) a.v[i] = P::MOD64[i]; a.v[0] -= 1; }
    F r1 = a * b, r2 = slow(a, b), s1 = a.sqr(), s2 = slow(a, a);
    if (!(r1 == r2) || !(s1 == s2)) bad++;
  }
  F a = F::one(); a = a + a + a; F b = a*a + a;
  auto t0 = std::chrono::steady_clock::now();
  for (int i=0;i<1000000;i++) a = a*b;
  auto t1 = std::chrono::steady_clock::now();
  double m = std::chrono::duration<double,std::nano>(t1-t0).count()/1e6;
  t0 = std::chrono::steady_clock::now();
  for (int i=0;i<1000000;i++) a = a.sqr();
  t1 = std::chrono::steady_clock::now();
  double q = std::chrono::duration<double,std::nano>(t1-t0).count()/1e6;
  jacobian_host<F> pt; pt.X = a; pt.Y = b; pt.Z = F::one();
  t0 = std::chrono::steady_clock::now();
  for (int i=0;i<255*100;i++) pt.dbl();
  t1 = std::chrono::steady_clock::now();
  printf("%s: mismatches %d; mul %.1f ns, sqr %.1f ns, 255 doublings %.1f us (%llx)\n", name, bad, m, q, std::chrono::duration<double,std::micro>(t1-t0).count()/100, (unsigned long long)(a.v[0]^pt.X.v[0]));
  return bad;
}
#if defined(SPPARK_HOST_MULX) && defined(SPPARK_HOST_ADC)
// the two products of the fast path side by side (alternating, best of ten): adc chains in C against mulx / adcx / adox
typedef mont_host<bls12_381_fp_p> F6;
__attribute__((noinline)) static F6 mul_c(const F6& a, const F6& b)
{
    unsigned long long t[8] = {0};
    for (int i = 0; i < 6; i++) { F6::mac_row(t, a.v, b.v[i]); F6::red_row(t); }
    uint64_t u[7]; for (int i = 0; i <= 6; i++) u[i] = t[i];
    F6 r; F6::cond_sub(r.v, u, u[6]); return r;
}
__attribute__((noinline)) static F6 mul_asm(const F6& a, const F6& b)
{
    uint64_t u[6]; mont_mul_x86_6(u, a.v, b.v, bls12_381_fp_p::MOD64, bls12_381_fp_p::M0_64);
    F6 r; F6::cond_sub(r.v, u, 0); return r;
}
static void ab()
{
    F6 a = F6::one(); a = a + a + a; F6 b = a * a + a;
    double best[2] = {1e9, 1e9};
    for (int rep = 0; rep < 10; rep++) for (int k = 0; k < 2; k++) {
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 500000; i++) a = k ? mul_asm(a, b) : mul_c(a, b);
        auto t1 = std::chrono::steady_clock::now();
        double m = std::chrono::duration<double,std::nano>(t1 - t0).count() / 5e5;
        if (m < best[k]) best[k] = m;
    }
    printf("6 limbs, side by side: adc chains in C %.1f ns, mulx / adcx / adox %.1f ns (mulx+adx on this host: %d) (%llx)\n", best[0], best[1],
           (int)host_has_mulx_adx(), (unsigned long long)a.v[0]);
}
#else
static void ab() {}
#endif
int main(int argc, char** argv){
  if (argc > 2) ab();
  const int iters = argc > 1 ? atoi(argv[1]) : 200000;
  return run<bls12_381_fp_p>("bls12_381 fp", iters) + run<alt_bn128_fp_p>("alt_bn128 fp", iters) + run<bls12_377_fp_p>("bls12_377 fp", iters) != 0;
}
