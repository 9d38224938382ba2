// The plan of an MSM -- window size, split of the bucket index between the two sort levels, run length, fan-ins --
// as a function of the point count and the tunables.  Host logic only (no HIP): msm_driver.hpp launches what it says,
// tests/emu/emu_plan.cpp checks its invariants over every size in the GPU-less container.
#pragma once
#include <algorithm>
#include <cstddef>

namespace sppark_amd {

struct msm_plan {
    unsigned n, wbits, nwins, NB;       // wbits = longest window, NB = 2^(wbits-1) buckets per window
    unsigned nbits;                     // scalar bits, split evenly over the windows
    unsigned HB, LB, NA;                // bucket index = (k_hi : k_lo), NA = 2^HB partitions per window
    unsigned L, chunks_per_win;         // accumulate run length
    unsigned nslabs, slab_sz;           // hist/scatter point slabs
    unsigned F;                         // reduce_runs fan-in
    unsigned K;                         // bucket-reduction chunk
    unsigned K1;                        // ... of the first level (buckets per work item)
    unsigned G, wpg;                    // window groups, windows per group (the last group may be shorter)
    unsigned big;                       // level-A partitions above this many entries go to the cooperative level B (0 = the tunable / 2^18)
    // level-A records in 4 bytes (msm_sort_kernels.hpp): index bits kept in the record (0 = 8-byte records), log2 of the
    // slabs per index group, index groups (boundaries level B searches)
    unsigned IB, SH, NG;
};

struct msm_tunables {                   // 0 = automatic
    unsigned wbits = 0, L = 0, F = 0, K = 0, nslabs = 0, LB = 0;
    unsigned big = 0;                   // level-A partitions above this many entries are sorted cooperatively (0 = 2^18)
    unsigned records = 0;               // level-A sort records: 0 = 4 bytes unless a slab count is given, 1 = 8 bytes, 2 = 4 bytes also with a given slab count (rounded to power-of-two slabs)
    unsigned groups = 0;                // window groups (1 = everything on one stream)
    unsigned top = 0;                   // bucket sums: items per window handed to the subset-sum top (0 = 4096, 1 = never)
    unsigned join = 0;                  // record list: 1 = no k_join_runs (every segment through the fan-in tree), 2 = no one-launch narrow end, 3 = no low-latency bucket-sum kernels, 4 = no cooperative (four waves per operation) kernels (A/B switches)
    unsigned K1 = 0;                    // bucket sums: buckets per work item of the first level (0 = K)
    unsigned g2_coop = 0;               // G2 only: the accumulation with one Fp2 component per wave (msm_g2c_kernels.hpp): 0 = for the 14-limb base fields, 1 = always, 2 = never
    unsigned long_runs = 0;             // set by the driver for G2 over the 14-limb base fields (the wave-pair accumulation): the run lengths of that kernel (make_plan)
    size_t resident_lanes = 0;          // lanes of k_accumulate the device holds at once (set by the driver from the occupancy query; 0 = unknown)
    size_t chunk = 0;                   // points per chunk of the chunked path (0 = automatic)
    size_t max_scratch = 0;             // upper bound for the scratch blob in bytes (0 = what the device has)
};

static inline unsigned lg2_floor(size_t x) { unsigned r = 0; while (x >>= 1) r++; return r; }

static inline msm_plan make_plan(size_t npoints, unsigned scalar_bits, const msm_tunables& t)
{
    msm_plan p;
    p.n = (unsigned)npoints;
    unsigned lg = lg2_floor(npoints ? npoints : 1);
    // window: ~2^6 entries per bucket on average; both halves of the bucket index
    // must fit LDS counters (2^15 u32 = 128 KB of the 160 KB) => wbits - 1 <= 30, capped at 24
    // measured optima (profiles/r01_msm_small_sweep.log): ~2^4 entries per bucket at
    // 2^20..2^26; below that the serial depth of the reduction levels dominates and
    // much smaller windows (more, shorter windows in parallel) win
    // (round 3, with the sort split following the size and the cheaper tail: 2^17..2^19 moved from 8 / 8 / 11 to
    // 14 / 15 / 16 bits -- 2^18: 2.15 -> 1.82 ms, 2^19: 3.10 -> 2.42 ms, profiles/r03_msm_small_grid2.log)
    // (round 4, with the cooperative tail: 8 bits at 2^15 too -- 0.85 -> 0.73 ms, profiles/r04_msm_small_grid.log)
    // (round 6, with the piece tree and the small windows' sums: 8 bits at 2^14 too -- 0.68 -> 0.59 ms wall; 2^13 is level between
    // 4 and 8 bits and stays, profiles/r06_msm_small_grid2.log)
    unsigned autow = lg >= 22 ? std::min(22u, lg - 4) : lg >= 19 ? 16u : lg == 18 ? 15u : lg == 17 ? 14u
                   : lg >= 14 ? 8u : std::max(4u, lg > 10 ? lg - 10 : 0u);
    p.wbits = t.wbits ? t.wbits : autow;
    p.wbits = std::min(24u, std::max(2u, p.wbits));
    p.nwins = (scalar_bits - 1) / p.wbits + 1;      // as pippenger.cuh:365
    p.nbits = scalar_bits;
    p.wbits = scalar_bits / p.nwins + (scalar_bits % p.nwins ? 1 : 0);     // even split (window_len)
    p.NB = 1u << (p.wbits - 1);
    // Split of the bucket index between the two sort levels: 2^HB level-A partitions per window of ~2^14 entries each
    // -- what k_sortB's register path takes in one piece (18 K) -- but at least ~2^10 (partition, window) work-groups.
    // Until round 3 it was "as many partitions as possible" (HB = 12), right for 2^26 points only: at 2^20 that
    // is 65 536 work-groups of 256 entries (digits + sort 0.79 ms -> 0.33 ms with 2^7 partitions; 2^22: 1.31 -> 0.79,
    // 2^23: 2.04 -> 1.50, 2^24: 3.61 -> 3.07; profiles/r03_msm_sort_split.log)
    // (by the CEILING of lg n: 1.5 * 2^20 points in 2^6 partitions would be 24 K entries each, above the register path)
    const unsigned lgc = lg + ((npoints & (npoints - 1)) ? 1 : 0);
    unsigned hb = lgc > 14 ? lgc - 14 : 0;
    {
        const unsigned lgw = lg2_floor(p.nwins);
        hb = std::max(hb, lgw < 10 ? 10 - lgw : 0u);
        hb = std::min(hb, std::min(p.wbits - 1, 12u));
    }
    p.LB = t.LB ? std::min(t.LB, p.wbits - 1) : p.wbits - 1 - hb;
    if (p.LB > 13) p.LB = 13;                       // 2^LB LDS counters + scan words
    if (p.wbits - 1 - p.LB > 15) p.LB = p.wbits - 1 - 15;
    p.HB = p.wbits - 1 - p.LB;
    p.NA = 1u << p.HB;
    size_t entries = (size_t)p.n * p.nwins;
    // run length: 64 entries per lane, 128 from 2^22 points on, 256 from 2^25 (every chunk boundary costs one
    // full addition in k_join_runs; the accumulation itself is flat in L as long as there are > 10^5 lanes per
    // window.  2^23: tail 3.46 -> 2.67 ms with 128, 2^26: 11.5 -> 10.4 ms with 256, profiles/r03_msm_tail.log)
    // Below 2^22 points: 8..128 entries -- what still fills one round of 131 072 resident lanes -- twice what round 2 used
    // (2^14..2^18: -5..-9 %, profiles/r03_msm_small_grid.log); round 6: 128 instead of 64 at 2^20 / 2^21 (the accumulation is
    // the same one or two full rounds, the record list is half: 2^21 BLS12-381 6.81 -> 6.64 ms, BLS12-377 6.85 -> 6.32, Pallas
    // 3.63 -> 3.25, alt_bn128 unchanged; 2^20 3.78 -> 3.66 / unchanged; profiles/r06_msm_run_length_mid.log)
    unsigned L = t.L ? t.L : lg >= 22 ? (lg >= 25 ? 256u : 128u)
                           : 1u << lg2_floor(std::min<size_t>(128, std::max<size_t>(8, entries / 131072)));
    // Round 4: FIT THE GRID TO THE DEVICE.  k_accumulate's lanes all run the same L additions, so its time is
    // (rounds of resident waves) x L: 17 windows x 2^18 points / 32 = 139 264 lanes are 1.06 x the 131 072 that two waves
    // per SIMD hold -- two rounds, the second one for 6 % of the work (0.96 ms where 16 windows take 0.70,
    // profiles/r04_msm_small_grid.log).  With R = resident_lanes known, a grid that needs MORE than one round gets the
    // smallest run length that fits k rounds exactly, k = the rounds of the power-of-two choice above or one less
    // (fewer, longer runs also leave fewer records) -- if that saves at least 8 % of rounds x L; a grid that fills its
    // rounds already (2^16, 2^19, 2^20, 2^23 ... 2^26 points) keeps its power of two.  2^17: 19 x 8192 lanes at L = 16
    // (1.19 rounds) -> L = 20, one round: 1.41 -> 1.24 ms; 2^18: L = 32 -> 35: 1.95 -> 1.65 ms; 300 000 points: L = 32 -> 40:
    // 2.02 -> 1.84 ms (profiles/r04_msm_fit_rounds.log).  Not applied to a grid below one round
    // (shorter runs there only add records), and R counts at most two waves per SIMD: a third one (ten-limb fields)
    // adds no throughput (alt_bn128 at 2^20: 1.33 rounds of three waves at L = 64 run faster than one round at L = 86).
    // (In WORK-GROUPS: the grid is ceil(chunks / 256) groups of 256 lanes per window, and the groups, not the lanes, are
    // what must fit -- 300 000 points at L = 39 are 130 781 lanes but 17 x 31 = 527 groups for 512 places: 1.17 ms
    // instead of 0.97 at L = 32; at L = 40 they are 510.)
    // G2 by wave pairs (msm_g2c_kernels.hpp): an addition costs three of G1's and a run boundary a full Fp2 addition in the join,
    // half as many runs are resident, and the fit above is calibrated on G1's kernel -- so longer runs, from a sweep of its own
    // (BLS12-381 G2, profiles/r06_g2_run_length.log): 2^16 3.39 -> 3.13 ms and 2^18 6.55 -> 6.29 with 32, 2^20 13.9 -> 13.3 and
    // 2^21 24.1 -> 22.9 with 128 (2^19: 64 and 2^22: 128 were the choices already)
    const bool g2_runs = !t.L && t.long_runs && lg >= 16 && lg < 25;
    if (g2_runs) L = lg >= 20 ? 128u : lg >= 19 ? 64u : 32u;
    if (!t.L && !g2_runs && t.resident_lanes >= 256 && p.n >= 4096) {
        const size_t R = t.resident_lanes / 256, T0 = (size_t)p.nwins * (((p.n + L - 1) / L + 255) / 256);
        const size_t k_hi = (T0 + R - 1) / R, k_lo = k_hi > 1 ? k_hi - 1 : 1;
        if (k_hi >= 2 && k_hi <= 64) {
            size_t best_cost = 0; unsigned best_L = L;
            for (size_t k = k_lo; k <= k_hi; k++) {
                const size_t q = (k * R / p.nwins) * 256;           // chunks per window whose work-groups fit k rounds
                if (q == 0) continue;
                const size_t Lk = (p.n + q - 1) / q;
                if (Lk < 4 || Lk > 1024) continue;
                const size_t cost = k * Lk * 100 + (k - k_lo) * 3 * Lk;      // prefer the longer runs unless the shorter ones save > 3 %
                if (best_cost == 0 || cost < best_cost) { best_cost = cost; best_L = (unsigned)Lk; }
            }
            // (>= 8 %: the fit assumes full lists -- uniform scalars; with half of the digits zero a fitted L = 161 at 2^22
            // made the accumulation 13 % SLOWER than L = 128 for a 3 % gain on uniform ones, profiles/r04_msm_skew.log)
            if (best_cost && best_cost <= k_hi * (size_t)L * 92) L = best_L;
        }
    }
    p.L = L;
    p.chunks_per_win = (p.n + L - 1) / L;
    // point slabs of the level-A histogram / scatter: >= 8 from 2^14 points on (2^18: digits + sort 0.31 -> 0.17 ms with 8)
    p.nslabs = t.nslabs ? t.nslabs
             : (unsigned)std::min<size_t>(64, std::max<size_t>(npoints / 131072, std::min<size_t>(8, std::max<size_t>(1, npoints / 2048))));
    p.slab_sz = (p.n + p.nslabs - 1) / p.nslabs;
    // 4-byte level-A records: sign | index mod 2^IB | k_lo, IB = 31 - LB; the index bits above IB follow from the position of
    // a record in its partition, which needs slabs of a power of two <= 2^IB points (so that an index group is a whole
    // number of slabs) and at most 128 groups.  Not with an explicit slab count (the tunable means what it says) unless
    // tunables.records = 2 asks for both (the slabs are then the power of two below n / nslabs); records = 1: never.
    p.IB = p.SH = 0; p.NG = 1;
    if (!t.nslabs && p.LB < 16) {
        const unsigned IB = 31 - p.LB;
        // (rounded DOWN: a size just above a power of two keeps its work-group count -- 2^26 + 1 points are 65 slabs of 2^20,
        // not 33 of 2^21, which would be two rounds of twice the work on 256 compute units instead of three)
        const unsigned lgs = std::min(lg2_floor(p.slab_sz ? p.slab_sz : 1), IB);
        const size_t ss = (size_t)1 << lgs, ns = (npoints + ss - 1) / ss, ng = (((ns ? ns : 1) - 1) >> (IB - lgs)) + 1;
        // (at most 2 * 64 + 1 slabs: the heuristic's 64, doubled by the rounding, plus a ragged one -- what the per-slab
        // histograms H and k_scan_slabs' serial loop are sized for)
        if (ng <= 128 && ns <= 129) { p.slab_sz = (unsigned)ss; p.nslabs = (unsigned)std::max<size_t>(1, ns); p.IB = IB; p.SH = IB - lgs; p.NG = (unsigned)ng; }
    }
    // fan-in of the record tree (< 3 would never shrink the list): 4 up to 2^20 points, where the buckets are longer
    // than the join's walk and the tree does the work -- one addition per work item and level instead of three
    // (above 2^18 the join leaves the tree nothing to do and every level is an empty launch of ~6 us: fewer, wider ones)
    // (round 4: 16 wherever k_join_runs runs first -- the driver's condition n / NB <= 4 L -- : the tree then sees only the
    // records of long segments, with uniform scalars none at all, and every level it does not have is an empty launch of
    // ~6 us saved: ten levels become four at 2^18 points)
    const bool joined = t.join != 1 && (size_t)p.n / p.NB <= (size_t)4 * p.L;
    p.F = std::max(4u, t.F ? t.F : (joined ? 16u : 4u));
    p.K = t.K ? t.K : (lg <= 22 ? 4 : 8);
    p.K = std::min(p.K, p.NB);
    // first level: 16 buckets per work item once a window has >= 2^21 of them (2^26 points: tail 11.35 -> 11.03 ms)
    // 8 where that hands 4096 partial sums per window straight to the subset-sum top (2^15 buckets: no chunked level at all)
    // (round 6, profiles/r06_msm_sums_sweep.log: 2^14 buckets likewise -- 2^18 points, tail 0.69 -> 0.64 ms; 128 buckets --
    // 2^15 / 2^16 points -- go to the top as they are, the first level only replaces the buckets without entries: 0.51 -> 0.48 ms)
    p.K1 = std::min(t.K1 ? t.K1 : (p.NB >= (1u << 21) ? 16u : (p.NB == (1u << 15) || p.NB == (1u << 14)) ? 8u : p.NB == 128u ? 1u : p.K), p.NB);
    // window groups: ONE by default.  Measured on MI355X (profiles/r02_msm_groups.log): whatever
    // the sort of the next group gains by running beside k_accumulate, the accumulation loses --
    // 2^26 points: 168.5 ms with one group, 169-190 ms with 2..12 groups, with the sort stream on
    // all CUs or confined to 4..64 of them, at any stream priority.  Groups remain useful to bound
    // the sort scratch (two group-sized sets instead of W) and are a tunable.
    unsigned G = t.groups ? t.groups : 1u;
    G = std::max(1u, std::min(G, p.nwins));
    p.wpg = (p.nwins + G - 1) / G;
    p.G = (p.nwins + p.wpg - 1) / p.wpg;
    p.big = 0;
    return p;
}

// The plan of a fixed-base MSM (msm_driver.hpp invoke_fixed): ONE window of |fb_wbits| bits over fb_nwins * n entries --
// the multiples are baked into the table, so every real window's digits select from one bucket set.
// |register_stage|: entries level B of the sort keeps in registers (msm_sort_kernels.hpp SORTB_STAGE).
static inline msm_plan make_fixed_plan(size_t n, unsigned fb_wbits, unsigned fb_nwins, unsigned register_stage, const msm_tunables& tune)
{
    msm_plan p;
    const size_t entries = (size_t)fb_nwins * n;
    const unsigned lg = lg2_floor(entries ? entries : 1);
    p.n = (unsigned)entries; p.wbits = fb_wbits; p.nwins = 1; p.nbits = fb_wbits;
    p.NB = 1u << (p.wbits - 1);
    // level-A partitions of ~2^14 entries as in make_plan, but never more than the 2^12 the LDS-staged scatter
    // takes: with 2^15 partitions the direct scatter (8-byte stores to 32 768 open rows) alone is 16.7 ms at
    // 11 x 2^26 entries, against 4.8 ms staged; the partitions are then ~2^17.5 entries and go to level B's
    // cooperative form in slices (profiles/r03_msm_fixed_base.log)
    unsigned hb = lg > 14 ? lg - 14 : 0;
    hb = std::min(std::max(hb, 10u), std::min(p.wbits - 1, 12u));
    p.LB = tune.LB ? std::min(tune.LB, p.wbits - 1) : p.wbits - 1 - hb;
    if (p.LB > 13) p.LB = 13;
    if (p.wbits - 1 - p.LB > 15) p.LB = p.wbits - 1 - 15;
    p.HB = p.wbits - 1 - p.LB; p.NA = 1u << p.HB;
    p.L = tune.L ? tune.L : 1u << lg2_floor(std::min<size_t>(256, std::max<size_t>(8, entries / 131072)));
    p.chunks_per_win = (p.n + p.L - 1) / p.L;
    p.nslabs = tune.nslabs ? tune.nslabs : (unsigned)std::min<size_t>(1024, std::max<size_t>(entries / 131072, std::min<size_t>(8, std::max<size_t>(1, entries / 2048))));
    p.slab_sz = (p.n + p.nslabs - 1) / p.nslabs;
    p.F = std::max(4u, tune.F ? tune.F : 8u);
    p.K = std::min(tune.K ? tune.K : 8u, p.NB);
    p.K1 = std::min(tune.K1 ? tune.K1 : (p.NB >= (1u << 21) ? 16u : p.NB == (1u << 15) ? 8u : p.K), p.NB);
    p.G = 1; p.wpg = 1;
    // every partition beyond level B's register form goes to the cooperative form in LDS-staged slices: the
    // partitions are ALL of one size class here, and one work-group walking 50..300 K entries twice is the slower way
    // (2^25 points: digits + sort 9.1 -> 6.4 ms, 2^24: 4.2 -> 3.3 ms)
    p.big = register_stage;
    p.IB = p.SH = 0; p.NG = 1;          // 8-byte level-A records (the entry index is fb_nwins x the point index)
    return p;
}

} // namespace sppark_amd
