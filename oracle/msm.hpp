// ORACLE — TEST INFRASTRUCTURE ONLY (see ff.hpp header).
//
// CPU restatement of the reference's host Pippenger, msm/pippenger.hpp:13-374
// (the path BASELINE.json calls "msm/pippenger.hpp on host CPU"):
//
//   get_wval            <- :13-29    unsigned window extraction (no Booth)
//   window_size         <- :31-38
//   integrate_buckets   <- :40-56    running-sum  sum_k (k+1)*B_k
//   tile                <- :81-104   one (point range) x (bit window) tile
//   bit_length          <- :106-157  (num_bits)
//   breakdown           <- :160-190  (nx, ny, window) from nbits/window/ncpus
//   mult                <- :192-214  double-and-add for npoints == 1
//   mult_pippenger      <- :218-350  single-thread top-down path and the
//                                    pooled tile-grid path with strictly
//                                    top-down row consumption
//
// Plus two definition-level checkers that are NOT in the reference:
//   msm_naive  : sum_i double-and-add(s_i, P_i)
//   msm_signed : a model of the GPU recoding semantics documented in
//                msm/pippenger.cuh:63-69,84-118 and msm/sort.cuh:92 (signed
//                windows, bucket = |digit| - 1, scalar-level abs()).
//
// npoints == 0 is undefined in the reference (tile() decrements 0,
// msm/pippenger.hpp:94); this restatement defines it as "result = infinity".
#pragma once
#include "ec.hpp"
#include <vector>
#include <thread>
#include <atomic>
#include <mutex>
#include <condition_variable>
#include <algorithm>
#include <tuple>

namespace oracle {

// up to 25 bits starting at bit |off| of the little-endian byte string |d|,
// never reading past the byte that holds the last requested bit.
static inline size_t get_wval(const unsigned char* d, size_t off, size_t bits)
{
    if (bits == 0) return 0;
    size_t first = off / 8, last = (off + bits - 1) / 8;
    uint64_t acc = 0;
    for (size_t k = first, sh = 0; k <= last && sh < 64; k++, sh += 8)
        acc |= (uint64_t)d[k] << sh;
    return (size_t)(acc >> (off % 8));
}

static inline size_t window_size(size_t npoints)
{
    size_t lg = 0;
    while (npoints >>= 1) lg++;
    if (lg > 12) return lg - 3;
    if (lg > 4)  return lg - 2;
    return lg ? 2 : 1;
}

static inline size_t bit_length(size_t l)
{   size_t n = 0; while (l) { n++; l >>= 1; } return n;   }

static inline std::tuple<size_t, size_t, size_t>
breakdown(size_t nbits, size_t window, size_t ncpus)
{
    size_t nx, ny, wnd;

    if (nbits > window * ncpus) {
        nx = 1;
        wnd = bit_length(ncpus / 4);
        if (window + wnd > 18) {
            wnd = window - wnd;
        } else {
            wnd = (nbits / window + ncpus - 1) / ncpus;
            wnd = ((nbits / (window + 1) + ncpus - 1) / ncpus < wnd) ? window + 1 : window;
        }
    } else {
        nx = 2;
        wnd = window - 2;
        while ((nbits / wnd + 1) * nx < ncpus) {
            nx += 1;
            wnd = window - bit_length(3 * nx / 2);
        }
        nx -= 1;
        wnd = window - bit_length(3 * nx / 2);
    }
    ny = nbits / wnd + 1;
    wnd = nbits / ny + 1;
    return std::make_tuple(nx, ny, wnd);
}

// sum_{k=0}^{2^wbits-1} (k+1) * buckets[k]; buckets are reset to infinity.
template<class F>
static void integrate_buckets(jacobian<F>& out, std::vector<xyzz<F>>& buckets, size_t wbits)
{
    size_t n = (size_t)1 << wbits;
    xyzz<F> acc = buckets[--n], ret = acc;
    buckets[n].inf();
    while (n--) {
        acc.add(buckets[n]);
        ret.add(acc);
        buckets[n].inf();
    }
    out = ret.to_jacobian();
}

// one tile: points [0,npoints) x scalar bits [bit0, bit0+wbits)
template<class F>
static void tile(jacobian<F>& ret, const affine<F>* points, size_t npoints,
                 const unsigned char* scalars, size_t nbytes,
                 std::vector<xyzz<F>>& buckets, size_t bit0, size_t wbits, size_t cbits)
{
    size_t wmask = ((size_t)1 << wbits) - 1, cmask = ((size_t)1 << cbits) - 1;
    for (size_t i = 0; i < npoints; i++) {
        size_t wval = get_wval(scalars + i * nbytes, bit0, wbits) & wmask & cmask;
        if (wval) buckets[wval - 1].add(points[i]);
    }
    integrate_buckets(ret, buckets, cbits);
}

// double-and-add, msm/pippenger.hpp:192-214
template<class F>
static void mult(jacobian<F>& ret, const affine<F>& point, const unsigned char* scalar, size_t nbits)
{
    ret.inf();
    if (point.is_inf()) return;
    jacobian<F> p; p.set(point);
    size_t top = nbits;
    auto bit = [&](size_t i) { return (scalar[i / 8] >> (i % 8)) & 1; };
    while (top && !bit(top - 1)) top--;
    while (top--) {
        ret.dbl();
        if (bit(top)) ret.add(p);
    }
}

// scalars: |npoints| little-endian byte strings of |nbytes| each, already out
// of Montgomery form (the caller applies to_scalar when mont == true, as
// msm/pippenger.hpp:232-243 does).  ncpus < 2 selects the serial path.
template<class F>
static void mult_pippenger(jacobian<F>& ret, const affine<F>* points, size_t npoints,
                           const unsigned char* scalars, size_t nbits, size_t ncpus)
{
    const size_t nbytes = (nbits + 7) / 8;
    ret.inf();
    if (npoints == 0) return;

    size_t window = window_size(npoints);

    if (ncpus < 2 || npoints < 32) {
        if (npoints == 1) { mult(ret, points[0], scalars, nbits); return; }

        std::vector<xyzz<F>> buckets((size_t)1 << window);
        for (auto& b : buckets) b.inf();
        jacobian<F> p;

        // top excess bits modulo the window size come first
        size_t wbits = nbits % window, cbits = wbits + 1, bit0 = nbits;
        while (bit0 -= wbits) {
            tile(p, points, npoints, scalars, nbytes, buckets, bit0, wbits, cbits);
            ret.add(p);
            for (size_t i = 0; i < window; i++) ret.dbl();
            cbits = wbits = window;
        }
        tile(p, points, npoints, scalars, nbytes, buckets, 0, wbits, cbits);
        ret.add(p);
        return;
    }

    size_t nx, ny;
    std::tie(nx, ny, window) = breakdown(nbits, window, ncpus);

    struct tile_t { size_t x, dx, y, dy; jacobian<F> p; };
    std::vector<tile_t> grid(nx * ny);

    size_t dx = npoints / nx, y = window * (ny - 1), total = 0;
    for (; total < nx; total++) {
        grid[total].x = total * dx; grid[total].dx = dx;
        grid[total].y = y;          grid[total].dy = nbits - y;
    }
    grid[total - 1].dx = npoints - grid[total - 1].x;
    while (y) {
        y -= window;
        for (size_t i = 0; i < nx; i++, total++) {
            grid[total].x = grid[i].x; grid[total].dx = grid[i].dx;
            grid[total].y = y;         grid[total].dy = window;
        }
    }

    std::vector<std::atomic<size_t>> row_done(ny);
    for (auto& r : row_done) r = 0;
    std::atomic<size_t> counter(0);
    std::mutex mtx; std::condition_variable cv;

    size_t n_workers = std::min(ncpus, total);
    std::vector<std::thread> workers;
    for (size_t w = 0; w < n_workers; w++) {
        workers.emplace_back([&, window, total, nbits, nx]() {
            std::vector<xyzz<F>> buckets((size_t)1 << window);
            for (auto& b : buckets) b.inf();
            for (size_t work; (work = counter++) < total;) {
                tile_t& t = grid[work];
                tile(t.p, &points[t.x], t.dx, scalars + t.x * nbytes, nbytes,
                     buckets, t.y, t.dy, t.dy + (t.dy < window));
                if (++row_done[t.y / window] == nx) {
                    std::lock_guard<std::mutex> lk(mtx);
                    cv.notify_all();
                }
            }
        });
    }

    // consume rows strictly top-down: add the row's tiles, then |window|
    // doublings before the next row (msm/pippenger.hpp:333-349).
    size_t row = 0;
    for (size_t r = ny; r--;) {
        {
            std::unique_lock<std::mutex> lk(mtx);
            cv.wait(lk, [&] { return row_done[r] == nx; });
        }
        for (size_t i = 0; i < nx; i++) ret.add(grid[row++].p);
        if (r) for (size_t i = 0; i < window; i++) ret.dbl();
    }
    for (auto& t : workers) t.join();
}

// ---- definition-level checkers (not in the reference) ----------------------
template<class F>
static void msm_naive(jacobian<F>& ret, const affine<F>* points, size_t npoints,
                      const unsigned char* scalars, size_t nbits)
{
    const size_t nbytes = (nbits + 7) / 8;
    ret.inf();
    for (size_t i = 0; i < npoints; i++) {
        jacobian<F> t;
        mult(t, points[i], scalars + i * nbytes, nbits);
        ret.add(t);
    }
}

// Signed-window bucket method as the GPU path defines it:
// s > r/2 is replaced by r - s with every digit negated; window w covers bits
// [w*c, w*c + c); digits live in [-2^(c-1)+1, 2^(c-1)]; bucket = |digit| - 1.
// |order| = little-endian 64-bit limbs of the group order r.
template<class F>
static void msm_signed(jacobian<F>& ret, const affine<F>* points, size_t npoints,
                       const unsigned char* scalars, size_t nbits, size_t c,
                       const uint64_t* order)
{
    const size_t nbytes = (nbits + 7) / 8, nlimbs = (nbits + 63) / 64;
    const size_t nwins = (nbits - 1) / c + 1;
    std::vector<std::vector<xyzz<F>>> buckets(nwins, std::vector<xyzz<F>>((size_t)1 << (c - 1)));
    for (auto& row : buckets) for (auto& b : row) b.inf();

    for (size_t i = 0; i < npoints; i++) {
        uint64_t s[8] = {0}, half[8], neg[8];
        for (size_t k = 0; k < nbytes; k++) s[k / 8] |= (uint64_t)scalars[i * nbytes + k] << (8 * (k % 8));
        // half = (r-1)/2 ; neg = r - s
        for (size_t k = 0; k < nlimbs; k++)
            half[k] = (order[k] >> 1) | (k + 1 < nlimbs ? order[k + 1] << 63 : 0);
        bool gt = false;
        for (size_t k = nlimbs; k--;) if (s[k] != half[k]) { gt = s[k] > half[k]; break; }
        if (gt) {
            uint64_t borrow = 0;
            for (size_t k = 0; k < nlimbs; k++) {
                u128 d = (u128)order[k] - s[k] - borrow;
                neg[k] = (uint64_t)d; borrow = (uint64_t)(d >> 64) & 1;
            }
            for (size_t k = 0; k < nlimbs; k++) s[k] = neg[k];
        }
        unsigned char le[64] = {0};
        for (size_t k = 0; k < nbytes; k++) le[k] = (unsigned char)(s[k / 8] >> (8 * (k % 8)));

        size_t carry = 0;
        for (size_t w = 0; w < nwins; w++) {
            size_t bits = std::min(c, nbits - w * c);
            size_t d = (get_wval(le, w * c, bits) & (((size_t)1 << bits) - 1)) + carry;
            bool minus = false;
            carry = 0;
            if (d > ((size_t)1 << (c - 1))) { d = ((size_t)1 << c) - d; minus = true; carry = 1; }
            if (d) buckets[w][d - 1].add(points[i], minus != gt);
        }
    }

    ret.inf();
    for (size_t w = nwins; w--;) {
        jacobian<F> p;
        integrate_buckets(p, buckets[w], c - 1);
        ret.add(p);
        if (w) for (size_t k = 0; k < c; k++) ret.dbl();
    }
}

} // namespace oracle
