/*
 * sppark_amd C ABI — the drop-in boundary.
 *
 * One shared object per field/curve, exactly as the reference builds one per
 * -DFEATURE_* (rust/src/build.rs ccmd(), poc/{msm,ntt}-cuda/build.rs):
 *
 *   libsppark_bls12_381.so  mult_pippenger_inf, mult_pippenger          (BLS12-381 G1)
 *                           mult_pippenger_fp2_inf                      (BLS12-381 G2)
 *                           compute_ntt                                 (BLS12-381 Fr, 2-adicity 32)
 *   libsppark_bn254.so      mult_pippenger_inf, mult_pippenger          (alt_bn128 G1)
 *                           mult_pippenger_fp2_inf                      (alt_bn128 G2)
 *                           compute_ntt                                 (alt_bn128 Fr, 2-adicity 28)
 *   libsppark_bls12_377.so  the same four entry points over BLS12-377   (G1, G2 with u^2 = -5, Fr: 2-adicity 47)
 *   libsppark_pallas.so     mult_pippenger_inf, mult_pippenger, compute_ntt over the scalar field (Pasta; no G2)
 *   libsppark_vesta.so      likewise
 *   libsppark_gl64.so       compute_ntt                                 (Goldilocks)
 *   libsppark_bb31.so       compute_ntt                                 (BabyBear)
 *   libsppark_gl64_plonky2.so / libsppark_bb31_canonical.so   the -DGOLDILOCKS_PLONKY2 / -DBABY_BEAR_CANONICAL root conventions
 *   libsppark_m31.so / libsppark_bb31x4.so   section 3 only (Mersenne31, BabyBear quartic extension)
 *
 * plus, in every library, the four symbols of util/all_gpus.cpp:65-86.
 * Section 1 below declares exactly what the reference's Rust/Go callers bind;
 * section 2 is this implementation's extension surface (contexts that keep
 * scratch memory and inputs resident in HBM, stream selection, kernel timers,
 * host-side point helpers).  Plain pointers and sizes only.
 *
 * All field elements are little-endian limbs in Montgomery form (R = 2^384 for
 * the BLS12-381 base field, 2^256 for 256-bit fields, 2^32 for BabyBear), except
 * Goldilocks (canonical u64) and MSM scalars (plain little-endian integers < r).
 * Pointers may be HOST or DEVICE (HIP) pointers; the library detects which.
 */
#ifndef SPPARK_AMD_H
#define SPPARK_AMD_H

#include <stddef.h>
#include <stdint.h>
#include <stdbool.h>

#ifdef __cplusplus
extern "C" {
#endif

/* util/rusterror.h:18-36 — returned BY VALUE (RAX:RDX on SysV x86-64).
 * code == 0: success; otherwise the negated HIP error code.  message is
 * malloc'ed (strdup) or NULL; the caller frees it (Rust: libc free on drop,
 * rust/src/lib.rs:9-23; Go: drop_error_message, go/sppark.go:51-60). */
typedef struct { int code; char *message; } SppError;

/* ------------------------------------------------------------------------ */
/* 1. The reference's FFI surface                                            */
/* ------------------------------------------------------------------------ */

/* poc/msm-cuda/cuda/pippenger_inf.cu:29-34.  points: array of Affine_inf_t
 * (X | Y | infinity-flag byte) with stride ffi_affine_sz (104 for arkworks
 * BLS12-381 G1Affine, 72 for bn254); scalars: 32-byte integers < r, NOT in
 * Montgomery form; out: Jacobian X|Y|Z (144 B / 96 B).  On error out = infinity. */
SppError mult_pippenger_inf(void *out, const void *points, size_t npoints,
                            const void *scalars, size_t ffi_affine_sz);

/* poc/msm-cuda/cuda/pippenger_inf.cu:41-47.  The same over G2: coordinates are Fp2 elements
 * c0 | c1 (ff/bls12-381-fp2.hpp:33-34), so points are X(2 fp) | Y(2 fp) | flag byte with stride
 * ffi_affine_sz (200 for arkworks BLS12-381 G2Affine, 136 for bn254) and out is 3 Fp2
 * coordinates (288 B / 192 B).  Called by multi_scalar_mult_fp2_arkworks
 * (poc/msm-cuda/src/lib.rs:84-119). */
SppError mult_pippenger_fp2_inf(void *out, const void *points, size_t npoints,
                                const void *scalars, size_t ffi_affine_sz);
/* Not in the reference: which bucket-accumulation kernel mult_pippenger_fp2_inf runs (process-wide).  0 = automatic:
 * a PAIR of waves per 64 mixed additions, one Fp2 component per wave (msm/msm_g2c_kernels.hpp), over the 14-limb base
 * fields (BLS12-381, BLS12-377: 2^22 points 47.8 -> 40.2 ms), one lane per addition over the 10-limb one (alt_bn128);
 * 1 / 2 force either.  Same result either way; the tests run both against the oracle.
 * A TEST AND TUNING HOOK, not part of the drop-in surface: the switch is one process-wide variable per library, read at the
 * start of every mult_pippenger_fp2_inf call -- set it while no G2 call of this library is in flight on another thread (a
 * call that overlaps the change runs entirely under the old or the new value, but which one is unspecified), and set it
 * back to 0 when done. */
SppError sppark_msm_g2_path(unsigned mode);

/* poc/msm-cuda/cuda/pippenger.cu:20-25.  points: Affine_t (X | Y, infinity
 * encoded as all-zero), stride 2*sizeof(fp); same scalars/out as above. */
SppError mult_pippenger(void *out, const void *points, size_t npoints,
                        const void *scalars);

/* poc/ntt-cuda/cuda/ntt_api.cu:25-36.  In-place transform of 2^lg_domain_size
 * elements.  order: NN=0 NR=1 RN=2 RR=3; direction: forward=0 inverse=1;
 * type: standard=0 coset=1 (ntt/ntt.cuh:33-36).  lg_domain_size == 0 is a
 * successful no-op (ntt/ntt.cuh:220-221). */
SppError compute_ntt(size_t device_id, void *inout, uint32_t lg_domain_size,
                     int ntt_order, int ntt_direction, int ntt_type);

/* util/all_gpus.cpp:65-86 */
bool  cuda_available(void);
void  drop_gpu_ptr_t(void **ref);           /* gpu_ptr_t<void>& : one pointer to a ref-counted block */
void *clone_gpu_ptr_t(void *const *ref);    /* returns gpu_ptr_t<void>::by_value                      */
void  drop_error_message(char *msg);
/* poc/go/poc.cu:17-32 — Go bridge smoke test: launches a trivial kernel; message always set */
SppError cuda_func(void *ptr);

/* ------------------------------------------------------------------------ */
/* 2. Extension surface (device-resident inputs, reusable scratch, timers)   */
/*    Counterpart of the reference's C++-only msm_t / NTT::Base_dev_ptr API  */
/*    (msm/pippenger.cuh:351-388,582-610; ntt/ntt.cuh:344-350).              */
/* ------------------------------------------------------------------------ */

/* The one-shot entry points of section 1 borrow a context (streams + scratch memory) from a
 * process-wide pool keyed by device instead of allocating and freeing GBs on every call as the
 * reference's per-call msm_t does (msm/pippenger.cuh:730-747).  A context goes back to the pool
 * after the call and keeps its scratch only while that is below SPPARK_MSM_CACHE_BYTES
 * (environment, default 32 GiB); nothing is tied to the calling thread.  This call frees the
 * scratch of every idle pooled context (all devices) and the idle staging buffers of the library's
 * NTT / LDE / polynomial entry points (sppark_ntt_release_cached's pool).
 *
 * Stream contract of the one-shot entry points: they are synchronous.  Host inputs are copied in
 * chunks that overlap the arithmetic.  When an input is a DEVICE pointer the call first waits for
 * the whole device (hipDeviceSynchronize), so data produced on any stream is complete; results
 * are equal as group elements from run to run, but the Jacobian bytes may differ (the order of
 * additions inside a bucket depends on LDS-atomic cursors): compare after normalisation, as the
 * reference's tests do (poc/msm-cuda/tests/msm.rs:26-38). */
void sppark_msm_release_cached(void);

/* msm/batch_addition.cuh:25-132, the bitmap variants batch_addition / batch_diff (C++ templates in
 * the reference, whose kernels leave one partial sum per warp for sum_up(), :167-181; here the total
 * is returned).  out = sum of points[i] over the bits set in bitmap (bit k of word w = point 32*w+k,
 * ceil(npoints/32) words).  refmap != NULL: the selection is bitmap XOR refmap and a point that is
 * only in refmap is subtracted (batch_addition.cuh:60-61) -- the difference between two selections.
 * points: Affine_t / Affine_inf_t records with stride ffi_affine_sz; all pointers host or device. */
SppError sppark_batch_addition(void *out, const void *points, size_t npoints,
                               const uint32_t *bitmap, const uint32_t *refmap, size_t ffi_affine_sz);

/* number of usable devices = ngpus() of util/all_gpus.cpp:62-63 (the filtered list) */
size_t sppark_ngpus(void);

/* Multi-GPU G1 MSM inside one process (north star: "MSM shards the point/scalar vector across
 * the 8 GPUs of one node").  The reference models one gpu_t per device in a process
 * (util/all_gpus.cpp:39-63, util/gpu_t.cuh:173-267) but its msm_t drives one of them; here the
 * vector is cut into ndev contiguous shards, shard i runs on device i on its own host thread
 * (own context, own streams), and the ndev Jacobian partial sums are added on the host -- one
 * exchange step of 144 bytes per device.  points / scalars: HOST pointers, same formats as
 * mult_pippenger_inf (mont != 0: scalars in Montgomery form); ndev == 0: all devices.
 * The multi-PROCESS variant (one rank per GPU, RCCL all-gather of the partial sums) is
 * sppark_msm_rccl below (native callers) and sppark_amd/multi_gpu.py (torch.distributed). */
SppError sppark_msm_multi(void *out, const void *points, size_t npoints, const void *scalars,
                          int mont, size_t ffi_affine_sz, unsigned ndev);
/* General form: nshards independent (points[i], npoints[i], scalars[i]) triples; shard i runs on
 * device device_ids[i] (index in the filtered list; NULL: device i).  Each pointer is a host
 * pointer or a pointer into the memory of the shard's own device; a device may be named twice. */
SppError sppark_msm_multi_shards(void *out, const void *const *points, const size_t *npoints,
                                 const void *const *scalars, int mont, size_t ffi_affine_sz,
                                 unsigned nshards, const int *device_ids);
/* The same two with per-device timings: out_ms[i] = wall-clock milliseconds shard i's device spent on its MSM
 * (host-to-device copies included), ndev / nshards entries -- load imbalance between the devices is visible
 * to the caller (with ndev == 0 provide sppark_ngpus() entries). */
SppError sppark_msm_multi_ms(void *out, const void *points, size_t npoints, const void *scalars,
                             int mont, size_t ffi_affine_sz, unsigned ndev, float *out_ms);
SppError sppark_msm_multi_shards_ms(void *out, const void *const *points, const size_t *npoints,
                                    const void *const *scalars, int mont, size_t ffi_affine_sz,
                                    unsigned nshards, const int *device_ids, float *out_ms);

/* One PROCESS per GPU (north star: "sharded across the GPUs of one node, RCCL partial-sum exchange"): the exchange
 * step over the CALLER's communicator.  nccl_comm: an ncclComm_t the caller made with ncclCommInitRank (one rank per
 * process, its device current); stream: a hipStream_t of that device (NULL: the null stream).  Every rank passes its
 * Jacobian partial sum (host memory; the `out` of any MSM entry point above: 3 field elements, g2 != 0: 3 Fp2
 * elements); ONE ncclAllGather of that many bytes per rank, then every rank adds the nranks points on its host --
 * elliptic-curve addition is not an RCCL reduction operator, and N host additions are microseconds.  All ranks
 * return the same point.  The library does not link RCCL: ncclAllGather / ncclCommCount are looked up in
 * $SPPARK_RCCL_LIB when set, else in the copy the process already holds (the one that made the communicator), else in
 * librccl.so.1;
 * code ENOSYS when there is none.  The reference has no counterpart (one gpu_t per msm_t, msm/pippenger.cuh:328-353). */
SppError sppark_msm_rccl_sum(void *out, const void *partial, int g2, void *nccl_comm, void *stream);
/* A rank's whole share: the local G1 MSM over its shard on the calling thread's current device (arguments of
 * mult_pippenger_inf, plus mont), then the exchange above.  Collective: every rank calls it (npoints == 0 allowed);
 * a rank whose local MSM fails still contributes the point at infinity so that the others do not hang, and reports
 * its own error. */
SppError sppark_msm_rccl(void *out, const void *points, size_t npoints, const void *scalars,
                         int mont, size_t ffi_affine_sz, void *nccl_comm, void *stream);

typedef struct sppark_msm_ctx sppark_msm_ctx;

/* device_id indexes the filtered device list, -1 = current HIP device.
 * stream: a hipStream_t to launch on, or NULL for a private stream; every call on the private
 * stream first waits for the work already queued on the legacy default stream (device-resident
 * inputs produced on any OTHER stream must be complete, or that stream must be passed here). */
SppError sppark_msm_create(sppark_msm_ctx **ctx, int device_id, void *stream);
void     sppark_msm_destroy(sppark_msm_ctx *ctx);
SppError sppark_msm_set_stream(sppark_msm_ctx *ctx, void *stream);
/* wbits/L/F/K/nslabs = 0 keeps the automatic choice (an explicit nslabs also keeps the sort's level-A records at 8 bytes:
 * the 4-byte form needs point slabs of a power of two, which the automatic choice provides) */
SppError sppark_msm_tune(sppark_msm_ctx *ctx, unsigned wbits, unsigned L, unsigned F,
                         unsigned K, unsigned nslabs);
/* low_bits of the bucket index sorted by the second LDS level (0 = automatic) */
SppError sppark_msm_tune_sort(sppark_msm_ctx *ctx, unsigned low_bits);
/* sort partitions with more entries than this are split over several work-groups (skewed scalars;
 * 0 = automatic, 2^18) */
SppError sppark_msm_tune_split(sppark_msm_ctx *ctx, unsigned big_partition);
/* bucket sums: windows with at most top_items partial sums are finished by the subset-sum kernels
 * (k_bucket_top_bits / k_bucket_top_sum); 0 = automatic (4096), 1 = never */
SppError sppark_msm_tune_sums(sppark_msm_ctx *ctx, unsigned top_items);
/* The tail of an MSM.  join: 0 = the record segments of at most eight records (with uniform scalars: all
 * of them) are summed by one launch (k_join_runs) and the fan-in tree only sees the longer ones; 1 = every
 * segment goes through the tree; 2 / 3 = A/B switches: 2 = without the one-launch narrow end of the tree
 * (k_reduce_tail), 3 = without the low-latency bucket-sum kernels for small grids, 4 = without the cooperative (four waves
 * per operation) kernels, 5 = without the piece tree of the small sizes, 6 = the point conversion with one lane per point
 * instead of the coalesced form, 7 = the subset-sum top with a work-group per sum instead of per piece of a sum, 8 = every level of the piece
 * tree a launch of its own (no k_piece_tail_coop), 16 + x = that launch from the first level of at most 2^x work items, 10 = the latency-bound chunked bucket-sum levels with one lane per work
 * item (k_bucket_level1_lat / _levelN_lat) instead of their sums on two / three waves (k_bucket_level1_pipe / _levelN_pipe).  k1: buckets per work item of the first bucket-sum level (a power of
 * two; 0 = the same as the other levels). */
SppError sppark_msm_tune_tail(sppark_msm_ctx *ctx, unsigned join, unsigned k1);
/* Pipeline shape.  groups: the windows are sorted and accumulated in this many groups, the digits +
 * sort of group g+1 on a second stream beside the bucket accumulation of group g (0 / 1 = one group,
 * the default: on MI355X the overlap gains nothing; more groups shrink the sort
 * scratch).  chunk_points: host-resident inputs, and inputs whose scratch would not fit the device,
 * are processed in chunks of this many points, the copy of chunk c+1 under the arithmetic of chunk c
 * (0 = automatic).  max_scratch_bytes: upper bound of the scratch memory the context may allocate;
 * the chunk is halved until it fits (0 = what is free). */
SppError sppark_msm_tune_pipeline(sppark_msm_ctx *ctx, unsigned groups, size_t chunk_points,
                                  size_t max_scratch_bytes);
/* the sort's side of the plan: out = { point slabs, points per slab, index bits kept in a 4-byte level-A sort record (0: 8-byte
 * records), log2 slabs per index group, index groups, window groups, first-level bucket chunk, pieces per bucket the piece
 * tree of a small MSM takes (0: the record list goes through k_join_runs / the fan-in tree) } */
void sppark_msm_plan_sort(const sppark_msm_ctx *ctx, size_t npoints, unsigned out[8]);
/* level-A sort records: 0 = 4 bytes unless sppark_msm_tune gives a slab count, 1 = 8 bytes, 2 = 4 bytes also with a given slab
 * count (the slabs are then the power of two below npoints / nslabs): record format and slab count can be varied independently */
SppError sppark_msm_tune_records(sppark_msm_ctx *ctx, unsigned records);
/* chunks the last invoke was cut into / window groups the context would use for npoints */
unsigned sppark_msm_last_chunks(const sppark_msm_ctx *ctx);
/* invocations of this context whose record list and bucket sums ran twice: small MSMs sum the pieces of a bucket by a tree sized
 * for the average bucket (msm_piece_kernels.hpp); a bucket beyond that size (skewed scalars) is found after the fact and the
 * fan-in tree then runs over what was left.  A counter for tests and tuning; the result is exact either way. */
unsigned sppark_msm_tail_redone(const sppark_msm_ctx *ctx);
unsigned sppark_msm_plan_groups(const sppark_msm_ctx *ctx, size_t npoints);
SppError sppark_msm_reserve(sppark_msm_ctx *ctx, size_t npoints, size_t ffi_affine_sz,
                            int host_points, int host_scalars);
/* Preloaded bases (msm_t(points, np, ffi_affine_sz), msm/pippenger.cuh:351-385): copy npoints
 * points (host or device pointer) into HBM owned by the context; afterwards
 * sppark_msm_invoke(ctx, out, NULL, n <= npoints, scalars, ...) is the reference's
 * invoke(out, scalars) (pippenger.cuh:604-605) on the first n of them.  npoints == 0 drops them. */
SppError sppark_msm_set_points(sppark_msm_ctx *ctx, const void *points, size_t npoints,
                               size_t ffi_affine_sz);
size_t   sppark_msm_preloaded(const sppark_msm_ctx *ctx);
/* Fixed-base mode of the preloaded bases (G1 contexts; no counterpart in the reference, whose msm_t keeps the
 * bases themselves, pippenger.cuh:351-385): besides the points, the context keeps 2^(off_j) * P_i for every
 * window j of a wide signed-digit split of the scalar (W = ceil(scalar bits / c) windows, c ~ lg npoints, at
 * most 26; W x the memory, built once on the device with one field inversion per entry).
 * sppark_msm_invoke(ctx, out, NULL, npoints, ...) over EXACTLY these npoints points then sorts the
 * W * npoints (digit, multiple) pairs into ONE set of 2^(c-1) buckets: no per-window bucket sums, no
 * doublings between windows, W instead of ~12 additions per point.  Any other length falls back to the
 * ordinary path on the points themselves; the result is the same group element either way.
 * Measured on MI355X: -7 % at 2^24 .. 2^26 points, -4 % at 2^23, slower below (the plain
 * path's points are shared by all its windows and stay in the Infinity Cache; the tables are gathered from HBM
 * once each) -- so below 2^23 points the call is a plain sppark_msm_set_points unless a window width was forced.
 * sppark_msm_fixed_base_windows: W of the tables the context holds, 0 without them.  Refused
 * (hipErrorInvalidValue) when W * npoints >= 2^31.  sppark_msm_tune's wbits, when set, is the c the
 * tables are built with (8..26); sppark_msm_tune_pipeline's chunking switches the mode off. */
SppError sppark_msm_set_points_fixed_base(sppark_msm_ctx *ctx, const void *points, size_t npoints,
                                          size_t ffi_affine_sz);
unsigned sppark_msm_fixed_base_windows(const sppark_msm_ctx *ctx);
/* mont != 0: scalars are in Montgomery form (msm_t::invoke's `mont`).
 * points == NULL: use the preloaded points (ffi_affine_sz is then ignored). */
SppError sppark_msm_invoke(sppark_msm_ctx *ctx, void *out, const void *points, size_t npoints,
                           const void *scalars, int mont, size_t ffi_affine_sz);
SppError sppark_msm_enable_timing(sppark_msm_ctx *ctx, int on);
/* which: 0 = before the first accumulation launch (exposed sort + point conversion), 1 = the
 * k_accumulate launches (sum over the window groups), 2 = all device work, 3 = number of
 * k_accumulate launches; ms of the last invoke (of its first chunk when it was chunked) */
float    sppark_msm_kernel_ms(const sppark_msm_ctx *ctx, int which);
size_t   sppark_msm_scratch_bytes(const sppark_msm_ctx *ctx);
/* plan for npoints: {window bits, windows, buckets/window, run length, partitions, low bits, fan-in, chunk} */
void     sppark_msm_plan(const sppark_msm_ctx *ctx, size_t npoints, unsigned out[8]);

/* Host-side helpers on Jacobian points (no GPU needed): out = sum of n points;
 * affine = (x, y) of a Jacobian point, infinity -> all-zero.  Used by the
 * multi-GPU combine step and by callers that want affine results. */
void sppark_g1_jacobian_sum(void *out, const void *points, size_t n);
void sppark_g1_to_affine(void *out_xy, const void *jacobian);
/* the same for G2 points (Jacobian X|Y|Z, each coordinate c0 | c1) */
void sppark_g2_jacobian_sum(void *out, const void *points, size_t n);
void sppark_g2_to_affine(void *out_xy, const void *jacobian);
/* P_i = k_i * G for i < n on the DEVICE, k_i = splitmix64(seed) stream, 253-bit;
 * out: affine, stride bytes apart (flag byte written when stride > 2*sizeof(fp)).
 * out may be a host or device pointer.  Synthetic-input generator for benches. */
SppError sppark_g1_generate(void *out, size_t stride, size_t n, uint64_t seed);
/* ALL-DISTINCT synthetic inputs with known discrete logarithms: P_i = (a + i*b) * G for i < n, affine, written on the
 * device (out: DEVICE pointer, stride a multiple of 8; bytes above the coordinates cleared).  a, b: 128-bit little-endian
 * word pairs, a >= 1, b < 2^96, a < 2^126, n < 2^30 (every a + i*b is distinct and non-zero modulo the group order).
 * The arbitrary-point oracle of the reference's MSM test (poc/msm-cuda/tests/msm.rs:19-39) does not reach 2^26 points;
 * on this vector sum s_i P_i = (sum s_i (a + i b) mod r) * G needs one scalar multiplication. */
SppError sppark_g1_generate_progression(void *out, size_t stride, size_t n, const uint64_t a[2], const uint64_t b[2]);

/* gpu_ptr_t<void> handles for C callers (the reference creates them from C++ only):
 * allocate `bytes` of device memory behind a ref-counted handle / read its device pointer.
 * Release with drop_gpu_ptr_t, share with clone_gpu_ptr_t. */
void *sppark_gpu_ptr_alloc(size_t bytes);
void *sppark_gpu_ptr_get(void *const *ref);

/* NTT on a device- or host-resident buffer with an explicit stream.  On a device buffer and a non-NULL stream the call
 * only ENQUEUES the kernels (as NTT::Base_dev_ptr, ntt/ntt.cuh:344-350); with stream == NULL, or on a host buffer, it
 * returns when the result is in place (as compute_ntt).  Callers that time or pipeline transforms pass their stream. */
SppError sppark_ntt(size_t device_id, void *inout, uint32_t lg_domain_size,
                    int ntt_order, int ntt_direction, int ntt_type, void *stream);

/* Low-degree extension -- C++-only in the reference (class NTT, ntt/ntt.cuh).
 * sppark_lde = NTT::LDE_aux (ntt.cuh:283-336) / NTT::LDE (:338-340): inout (host or device)
 * holds 2^lg_domain_size evaluations in natural order in its first elements and has room for
 * 2^(lg_domain_size+lg_blowup); on return it holds the evaluations of the same polynomial on
 * the coset g*<w_ext> in natural order.  aux_out (NULL or 2^lg_domain_size elements, host or
 * device) receives the polynomial's coefficients in natural order. */
SppError sppark_lde(size_t device_id, void *inout, uint32_t lg_domain_size, uint32_t lg_blowup,
                    void *aux_out, void *stream);
/* NTT::LDE_powers(stream, d_inout, lg) (ntt.cuh:352-356): d_inout[i] *= g^bitrev(i), device buffer */
SppError sppark_lde_powers(size_t device_id, void *d_inout, uint32_t lg_domain_size, void *stream);
/* NTT::LDE_expand (ntt.cuh:358-365): d_out[i << lg_blowup] = d_in[i], zeros elsewhere; device
 * buffers.  As in the reference the two buffers either do not overlap or d_in is aligned to the END
 * of d_out (d_in == d_out + 2^(lg_domain_size+lg_blowup) - 2^lg_domain_size, lg_blowup >= 1): the
 * expansion is then done in place.  Any other overlap is an invalid-value error. */
SppError sppark_lde_expand(size_t device_id, void *d_out, const void *d_in, uint32_t lg_domain_size,
                           uint32_t lg_blowup, void *stream);
/* compute_ntt / sppark_lde on host buffers, the LDE's coefficient copy and the polynomial primitives keep their device
 * scratch between calls: at most four idle buffers and at most SPPARK_SCRATCH_CACHE_BYTES (environment, default 1 GiB)
 * of idle memory PER LIBRARY (every .so has its own pool); an allocation that fails frees the idle buffers and is tried
 * once more before the call reports out-of-memory.  The twiddle tables are cached per (device, size, direction); the
 * inter-pass tables of the 256-bit fields can be as large as the data (512 MB for a 2^24 transform) and are shared by
 * all transform sizes.  This call frees the idle scratch buffers AND every cached twiddle table of this library (they
 * are rebuilt on the next call); it must not run concurrently with a transform of the same library. */
void     sppark_ntt_release_cached(void);
/* diagnostics: idle bytes of the scratch pool / number of cached twiddle tables of this library */
size_t   sppark_ntt_cached_scratch_bytes(void);
size_t   sppark_ntt_cached_tables(void);

/* ------------------------------------------------------------------------ */
/* 3. Polynomial primitives over the library's NTT field (every library)      */
/*    C++ templates only in the reference (polynomial/<name>.cuh); buffers host or */
/*    device, elements in the wire format of compute_ntt.                     */
/* ------------------------------------------------------------------------ */

/* polynomial/prefix_op.cuh:324-396 with the functors of :17-47: inclusive scan
 * out[i] = inp[0] (op) ... (op) inp[i]; op: 0 = Add, 1 = Multiply.  out may equal inp. */
SppError sppark_prefix_op(size_t device_id, void *out, const void *inp, size_t len, int op, void *stream);
/* polynomial/evaluate.cuh:307-412: ret[j] = sum_{i < len} coeffs[i] * x[j]^i for j < n */
SppError sppark_poly_evaluate(size_t device_id, void *ret, const void *x, size_t n,
                              const void *coeffs, size_t len, void *stream);
/* polynomial/div_by_x_minus_z.cuh:447-486: in-place division of sum_i inout[i] X^i by (X - z), z = one
 * field element.  rotate == 0: inout[0] = remainder = p(z), inout[1..] = quotient; rotate != 0: the
 * quotient in inout[0 .. len-2], the remainder in inout[len-1] (div_by_x_minus_z.cuh:152-157). */
SppError sppark_div_by_x_minus_z(size_t device_id, void *inout, size_t len, const void *z, int rotate, void *stream);

#ifdef __cplusplus
}
#endif
#endif
