// HOST TEST HARNESS (tests only): the CONTRACTS of the loosely-reduced field, machine-checked.
//
// ff/montx_dev.hpp computes on values that are only loosely reduced (value < K p for a K stated at each use, limbs that
// may exceed 2^LB by a stated factor) and every operation has a precondition on its operands: a subtrahend below (K-1) p
// with limbs <= B 2^LB, a normalised right operand of a product, a left operand whose limbs stay below 2^31, ...  The
// point formulas of ec/xyzzx_dev.hpp (G1) and ec/xyzzx2_dev.hpp over ff/fp2x_dev.hpp (G2) are written against those
// contracts and promise invariants for what they leave in memory (X < 10 p with limbs <= 5 2^LB, ...).  Random tests can
// only sample them.  With -DSPPARK_TRACK_BOUNDS every field value carries its CLAIMED bounds, every operation checks its
// operands against its precondition and derives the claim of its result by the rule written at its definition; this
// harness starts every point operation from the LOOSEST operands the invariants allow and checks that no precondition
// is violated on the way and that the results are inside the invariants again -- for all inputs, not for a sample.
// (The limb values themselves are ordinary points, so that each formula takes the path under test.)
#define SPPARK_HOST_EMULATION 1
#define SPPARK_TRACK_BOUNDS 1
#include "../../sppark_amd/csrc/msm/curve_select.hpp"
#include <cstdio>
#include <cstring>
#include <vector>

using namespace sppark_amd;

static int g_violations = 0;
static char g_first[512];
extern "C" void sppark_bound_violation(const char* what, double got, double limit)
{
    if (!g_violations) snprintf(g_first, sizeof(g_first), "%s: %.6g against %.6g", what, got, limit);
    g_violations++;
}
static void expect(bool ok, const char* what, double got, double lim) { if (!ok) sppark_bound_violation(what, got, lim); }

#ifdef SPPARK_G2
typedef fp2_d F;
typedef F::fp B;                                        // the base field carries the bounds
static void set_b(F& x, double v, double l) { x.c0.bnd_set(v, l); x.c1.bnd_set(v, l); }
static double val_b(const F& x) { return x.c0.bv > x.c1.bv ? x.c0.bv : x.c1.bv; }
static double limb_b(const F& x) { return x.c0.bl > x.c1.bl ? x.c0.bl : x.c1.bl; }
static bool unset_b(const F& x) { return x.c0.bv < 0 || x.c1.bv < 0 || x.c0.bl < 0 || x.c1.bl < 0; }
// ec/xyzzx2_dev.hpp: all four coordinates normalised, X < 9 p, Y < 5 p, ZZ, ZZZ < 2 p
static constexpr double INV[4][2] = {{9, 1}, {5, 1}, {2, 1}, {2, 1}};
#else
typedef msm_fp_d F;
static void set_b(F& x, double v, double l) { x.bnd_set(v, l); }
static double val_b(const F& x) { return x.bv; }
static double limb_b(const F& x) { return x.bl; }
static bool unset_b(const F& x) { return x.bv < 0 || x.bl < 0; }
// ec/xyzzx_dev.hpp: X < 10 p with limbs <= 5 2^LB, Y < 5 p with limbs <= 3 2^LB, ZZ, ZZZ < 2 p normalised;
// with 29-bit limbs (alt_bn128 G1): X < 10 p normalised, Y < 3 p with limbs <= 2 2^LB
static constexpr double INV[4][2] = {{10, F::TIGHT ? 1.0 : 5.0}, {F::TIGHT ? 3.0 : 5.0, F::TIGHT ? 2.0 : 3.0}, {2, 1}, {2, 1}};
#endif
typedef xyzz_dev<F> bucket;

// a bucket as loose as the invariants allow / an affine point as loose as its contract allows (< 2 p, normalised)
static void loosen(bucket& b)
{
    if (b.is_inf()) return;
    set_b(b.X, INV[0][0], INV[0][1]); set_b(b.Y, INV[1][0], INV[1][1]); set_b(b.ZZZ, INV[2][0], INV[2][1]); set_b(b.ZZ, INV[3][0], INV[3][1]);
}
static void check(const char* op, const bucket& b)
{
    if (b.is_inf()) return;
    const F* c[4] = {&b.X, &b.Y, &b.ZZZ, &b.ZZ};
    static const char* name[4] = {"X", "Y", "ZZZ", "ZZ"};
    char what[160];
    for (int k = 0; k < 4; k++) {
        snprintf(what, sizeof(what), "%s leaves %s outside its invariant (unset)", op, name[k]);
        expect(!unset_b(*c[k]), what, -1, 0);
        snprintf(what, sizeof(what), "%s leaves %s outside its value invariant", op, name[k]);
        expect(val_b(*c[k]) <= INV[k][0], what, val_b(*c[k]), INV[k][0]);
        snprintf(what, sizeof(what), "%s leaves %s outside its limb invariant", op, name[k]);
        expect(limb_b(*c[k]) <= INV[k][1], what, limb_b(*c[k]), INV[k][1]);
    }
}

// points: n >= 4 DISTINCT affine points of the curve (standard wire form, |stride| bytes apart), none at infinity.
// Returns the number of violated contracts; |msg| receives the first one.
extern "C" int emu_bounds_run(const unsigned char* points, size_t stride, size_t n, char* msg, size_t msglen)
{
    g_violations = 0; g_first[0] = 0;
    if (n < 4) return -1;
    // the conversion of wire points (k_convert_points) and what it writes: < 2 p, normalised
    std::vector<uint4> conv((size_t)n * affine_loader<F>::STRIDE / 16 + 1);
    for (size_t i = 0; i < n; i++) affine_loader<F>::template convert<false>((unsigned char*)conv.data(), points, i, (unsigned)stride);
    auto pt = [&](size_t i) {
        affine_dev<F> p = load_affine<F, false>((const unsigned char*)conv.data(), i, 0);
        set_b(p.X, 2, 1); set_b(p.Y, 2, 1);                         // the contract of an affine record
        return p;
    };
    {   // from_std itself stays inside that contract
        u32 w[F::NW];
        memcpy(w, points, sizeof(w));
        F x = F::from_std(w);
        expect(val_b(x) <= 2.0 && limb_b(x) <= 1.0, "from_std leaves the affine contract", val_b(x), 2.0);
    }
    for (int negate = 0; negate < 2; negate++) {
        // a general bucket: p0 + p1, then as loose as its invariant allows
        bucket a; a.set(pt(0), false); check("set", a);
        a.madd(pt(1), negate); check("madd after set", a);
        loosen(a);
        bucket m = a; m.madd(pt(2), negate); check("madd", m);
        bucket q; q.set(pt(2), negate); q.madd(pt(3), !negate); loosen(q);
        bucket s = a; s.add(q); check("add", s);
        bucket d = a; d.dbl(); check("dbl", d);
        bucket sp = a; sp.add_pairs(q); check("add_pairs", sp);
        bucket dp = a; dp.dbl_pairs(); check("dbl_pairs", dp);
        // the same point again: the doubling inside madd (dbl_affine), then from loose operands
        bucket e; e.set(pt(0), negate); loosen(e); e.madd(pt(0), negate); check("madd of the same point", e);
        // a + a by the full addition: the doubling inside add
        bucket f = a; f.add(a); check("add of the same bucket", f);
        // the results written in the reference's wire form (k_finalize: to_std on a loose value)
        bucket g = a; g.madd(pt(2), negate); loosen(g);
        xyzz_mem<F::NW> out; g.store_std(&out);
        // chains: a result is an admissible operand again without any help
        bucket h; h.set(pt(0), false);
        for (size_t i = 1; i < n; i++) { h.madd(pt(i), (i + negate) & 1); check("madd in a chain", h); }
        bucket h2 = h; h2.add(m); check("add in a chain", h2); h2.dbl(); check("dbl in a chain", h2); h2.add(h2); check("add of itself in a chain", h2);
    }
    if (msg && msglen) { strncpy(msg, g_first, msglen - 1); msg[msglen - 1] = 0; }
    return g_violations;
}

// the checker checks: deliberately broken contracts must be reported (returns how many were)
extern "C" int emu_bounds_selftest()
{
    g_violations = 0; g_first[0] = 0;
#ifdef SPPARK_G2
    typedef B fp;
#else
    typedef F fp;
#endif
    fp a = fp::one(), b = fp::one();
    a.bnd_set(5.0, 1.0);
    (void)fp::template sub<3>(b, a);                        // subtrahend not below 2 p
    a.bnd_set(1.0, 3.0);
    (void)(b * a);                                          // right operand of a product not normalised
    a.bnd_set(1.0, 9.0);
    (void)(a * b);                                          // left operand's limbs above 2^31
    fp u = b; u.bnd_set(-1.0, -1.0);
    (void)(u + b);                                          // a value nobody made a claim about
    a.bnd_set(14.0, 1.0);
    (void)a.template is_zero_mod<13>();                     // more multiples of p than the test compares with
    return g_violations;
}
