/* trh.h -- C ABI of libtrh.so, the MI355X (gfx950) backend for the MSM / NTT / IPA-commitment
 * hot path of halo2_proofs as driven by the tiny-ram-halo2 TinyRAM circuit.
 *
 * The reference has no FFI of its own: the seam is the set of Rust functions that
 * `halo2_proofs::plonk::create_proof` calls (reference call sites
 * /root/reference/src/test_utils.rs:21, 23-25, 41-49, 89-104; crate pinned at
 * /root/reference/Cargo.lock:619-621, pasta_curves at :847-858).  Each entry point below names
 * the Rust interface it replaces.  INTEGRATION.md shows the `extern "C"` block a maintainer
 * adds to the halo2_proofs fork.
 *
 * Data formats (identical to the Rust in-memory layout, so slices cross without repacking):
 *   field element  4 x u64 little-endian limbs, Montgomery form (R = 2^256), fully reduced
 *                  -- `pasta_curves::Fp` / `Fq`.
 *   affine base    8 x u64: x[4], y[4]; the all-zero pattern is the identity.  (pasta's
 *                  `Affine` carries a separate flag: the Rust shim repacks, see INTEGRATION.md.)
 *   point result   12 x u64: Jacobian X[4], Y[4], Z[4] as in pasta's `Point`, normalised to
 *                  Z = 1; the identity is all-zero.
 * Curves: pallas (base Fp, scalar Fq), vesta (base Fq, scalar Fp; `EqAffine`, the curve the
 * reference proves over).  Fields: fp, fq.
 *
 * Errors: every function returns 0 on success or a negative TRH_E* code; the message is
 * available from trh_last_error() (thread-local).  Nothing aborts or throws across the ABI;
 * the Rust shim maps a non-zero return to the panic the reference would raise.
 * Threading and contexts: a CONTEXT is bound to one GPU and owns every scratch buffer, table cache and
 * in-flight state, behind one lock.  trh_init() makes the process default; every entry point may be
 * called from any host thread (it makes the context's device current for the calling thread itself --
 * hipSetDevice is per thread) and calls on one context are serialised, on the host by its lock and on
 * the device by ordering each call's stream behind the stream of the previous call (they share the
 * scratch).  One MSM may be in flight per context (trh_msm_dev_enqueue .. _finish); a second enqueue
 * returns TRH_EBUSY.  Threads that want to OVERLAP work (an MSM beside an NTT, two MSMs) give each
 * thread its own context: trh_ctx_create + trh_ctx_set_current (thread-local binding; several contexts
 * per GPU are fine).  Handles (trh_bases_t, trh_domain_t, trh_expr_t) are device memory: usable from
 * every context of the device they were created on.  Callbacks (trh_ipa_create_proof's transcript /
 * rng) run with the context locked: they may call host-side helpers (trh_point_sum, trh_last_error)
 * but must not start device work on the same context.
 * Multi-GPU: trh_init_multi() binds a device group; base sets of at least trh_set_shard_min() points
 * created afterwards are range-sharded over the group and trh_msm / trh_msm_dev / trh_best_multiexp_*
 * run one local Pippenger per GPU and add the partial points on the host.
 * There is no CPU fallback: without a usable HIP device trh_init() fails and every compute
 * entry point returns TRH_ENODEV.
 */
#ifndef TRH_H
#define TRH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TRH_OK 0
#define TRH_EINVAL (-1)   /* bad argument (the Rust side panics on these: length mismatch, n != 1<<log_n) */
#define TRH_ENODEV (-2)   /* no HIP device / trh_init not called */
#define TRH_EHIP (-3)     /* HIP runtime error */
#define TRH_ENOMEM (-4)
#define TRH_EBUSY (-5)    /* the context already has an MSM in flight; trh_set_option while a context exists */
#define TRH_ESELFTEST (-6) /* trh_init: the known-answer self-test of the device arithmetic failed (trh_last_error() names the primitive) */

#define TRH_PALLAS 0
#define TRH_VESTA 1
#define TRH_FP 0
#define TRH_FQ 1

/* ---- lifecycle ------------------------------------------------------------------------ */
int trh_init(int device);            /* creates the process-default context on one GPU; idempotent */
void trh_shutdown(void);             /* destroys the default context / the device group (not contexts made by trh_ctx_create) */
const char* trh_last_error(void);
int trh_device_count(void);
const char* trh_version(void);       /* "trh <version> (gfx950, build <hash of the sources>)" */
/* The library's switches.  Each has an environment variable TRH_<NAME IN UPPER CASE> that is read ONCE (at the first trh_init /
 * trh_set_option); trh_set_option overrides it for hosts that must not touch the environment (a multi-threaded prover: getenv racing a
 * setenv elsewhere in the process is undefined behaviour, so no entry point of this library calls getenv after that).  Options are
 * process-wide and fixed while any context exists: call before trh_init, or after trh_shutdown (TRH_EBUSY otherwise).  Values are
 * decimal integers.  Names (defaults): pool_mb (4096), stage_slot_mb (16), copy_threads (-1 = by core count), bases_cache (0),
 * force_no_peer (0), ipa_fold (1 = the library's choice of the round after which an opening collapses its generators; 0 never), trace (0; bit 0
 * host-pointer entries, bit 1 IPA rounds), msm_chunk_gb (4), sparse (1), reduce_q4 (1), bin_sort (1), selftest (1) -- DESIGN.md section 8.   */
int trh_set_option(const char* name, const char* value);
int trh_get_option(const char* name, long* value);

/* ---- device group: the MSM range-sharded over the GPUs of one node (one host process, as the reference's prover is:
 * /root/reference/src/test_utils.rs:37-54).  devices[0] becomes the process default, exactly as trh_init(devices[0]);
 * a device may be listed more than once (logical shards on one GPU).  Base sets with at least trh_set_shard_min()
 * points (default 2^20) created afterwards by trh_bases_create_* / trh_bases_generate / trh_best_multiexp_* hold
 * shard g = points [g * ceil(n / G), ...) on devices[g]; MSMs over them enqueue one local Pippenger per device on that
 * device's own stream from the calling thread, copy the G partial points (96 B each) device-to-host and add them on
 * the host.  Device-resident scalars (trh_msm_dev) are handed to the other GPUs with peer copies.                    */
int trh_init_multi(const int* devices, int n_devices);
int trh_group_size(void);
int trh_group_peer_access(void);  /* 1: every pair of distinct group devices has peer access; 0: some pair has not (trh_last_error() after trh_init_multi names the first) */
int trh_set_shard_min(size_t n_points);

/* ---- explicit contexts (see "Threading and contexts" above) ------------------------------------------------------- */
typedef struct trh_ctx* trh_ctx_t;
int trh_ctx_create(int device, trh_ctx_t* out);
void trh_ctx_destroy(trh_ctx_t ctx);
int trh_ctx_set_current(trh_ctx_t ctx_or_null);  /* binds the calling THREAD; NULL = back to the process default */
int trh_ctx_device(trh_ctx_t ctx_or_null);       /* device index of the context (NULL: of the calling thread's), -1 if none */
/* the context's own non-blocking stream (a hipStream_t), for hosts without a HIP toolchain of their own (the Rust shim, the C++ driver):
 * pass it as the `stream` argument of the *_dev entries so that two contexts' work overlaps instead of meeting on the null stream */
void* trh_ctx_stream(trh_ctx_t ctx_or_null);

/* ---- halo2_proofs::arithmetic::best_multiexp(coeffs, bases) -> C::Curve ------------------
 * coeffs: n x 4 u64 (scalar field, Montgomery -- the memory image of `&[C::Scalar]`),
 * bases: n x 8 u64 affine, out: 12 u64.  Host pointers: the pairs cross PCIe in ranges, range t + 1 while range t is
 * computed (96 B per pair: bound by the link, 2^24 pairs in ~max(link, 17 ms)).             */
int trh_best_multiexp_pallas(const uint64_t* coeffs, const uint64_t* bases, size_t n, uint64_t out_xyz[12]);
int trh_best_multiexp_vesta(const uint64_t* coeffs, const uint64_t* bases, size_t n, uint64_t out_xyz[12]);

/* ---- halo2_proofs::arithmetic::best_fft(a, omega, log_n) ---------------------------------
 * a: 2^log_n x 4 u64 in place, natural order in and out; omega: 4 u64 Montgomery, must be a
 * primitive 2^log_n-th root of unity (what EvaluationDomain passes).  Host pointers.       */
int trh_best_fft_fp(uint64_t* a, const uint64_t omega[4], uint32_t log_n);
int trh_best_fft_fq(uint64_t* a, const uint64_t omega[4], uint32_t log_n);

/* the same for `count` host slices with one omega (one pointer per column): the uploads, transforms and downloads of
 * consecutive columns overlap on three streams (the link carries both directions at once); results as `count` single calls */
int trh_best_fft_batch_fp(uint64_t* const* a, size_t count, const uint64_t omega[4], uint32_t log_n);
int trh_best_fft_batch_fq(uint64_t* const* a, size_t count, const uint64_t omega[4], uint32_t log_n);

/* ---- host memory of the host-pointer entry points.  A pageable slice crosses PCIe through the library's pinned rings
 * (csrc/hostio.hip).  A caller that keeps long-lived buffers (Params.g_lagrange, a polynomial arena) may page-lock them once --
 * trh_host_register on memory it owns, or trh_host_alloc -- and the DMA engine then reads / writes them directly.
 * trh_io_stats: bytes moved by the host-pointer entry points on the calling thread's context and the host seconds spent in
 * the copies, per direction (overlapping copies each count their own time); reset != 0 clears the counters.              */
int trh_host_register(void* host, size_t bytes);
int trh_host_unregister(void* host);
int trh_host_alloc(void** host, size_t bytes);
int trh_host_free(void* host);
typedef struct trh_io_stats {
    double h2d_bytes, d2h_bytes, h2d_seconds, d2h_seconds;
    double h2d_zero_bytes; /* of h2d_bytes: pinned slots that were zero throughout (the padding of trh_best_fft's zero-padded vectors)
                              and were cleared on the device instead of crossing the link */
} trh_io_stats_t;
int trh_io_stats(trh_io_stats_t* out, int reset);

/* ---- device-resident base sets: poly::commitment::Params { g, g_lagrange, w } -------------
 * `Params::commit` / `commit_lagrange` run hundreds of MSMs over the same bases; upload once. */
typedef struct trh_bases* trh_bases_t;
int trh_bases_create_pallas(const uint64_t* xy_host, size_t n, trh_bases_t* out);
int trh_bases_create_vesta(const uint64_t* xy_host, size_t n, trh_bases_t* out);
/* wrap n bases already in device memory (no copy, not owned) */
int trh_bases_wrap_device(int curve, const void* xy_dev, size_t n, trh_bases_t* out);
/* synthetic set P_i = (s0 + (first + i) * d) * G, G = (-1, 2), generated on the device */
int trh_bases_generate(int curve, uint64_t s0, uint64_t d, uint64_t first, size_t n, trh_bases_t* out);
int trh_bases_download(trh_bases_t b, size_t offset, size_t n, uint64_t* xy_host);
const void* trh_bases_device_ptr(trh_bases_t b);
size_t trh_bases_len(trh_bases_t b);
int trh_bases_shards(trh_bases_t b);  /* devices the set is range-sharded over (trh_init_multi); 1 for a single-device set */
void trh_bases_destroy(trh_bases_t b);
/* Fixed-base tables for an owned set (Params.g / g_lagrange serve ~500 commitments per proof): stores
 * 2^(c j) * P_i for every window j (W x n x 128 B of HBM), after which a full-range MSM over the handle puts the
 * digits of all windows into ONE bucket set -- one bucket reduction per MSM instead of W, wider windows, no
 * Horner pass over windows.  window_bits 0 = automatic (<= 17); requires W * n <= 2^24.  Results are the same
 * group elements; MSMs over a sub-range (offset != 0 or n < len) keep using the per-window path.           */
int trh_bases_precompute(trh_bases_t b, int window_bits);   /* must not run while an MSM over `b` is in flight on another context; the same holds for trh_bases_destroy */
int trh_bases_precomputed_window_bits(trh_bases_t b); /* 0 when no table is attached */
/* sizes the calling context's MSM scratch for batches of `batch` MSMs (commitments) over the first n bases of the set, without running one:
 * setup-time, so that the first commitment batch of a process allocates nothing (see trh_domain_reserve) */
int trh_bases_reserve(trh_bases_t b, size_t n, size_t batch);

/* MSM over bases[offset .. offset+n) with host scalars (Params::commit / commit_lagrange) */
int trh_msm(trh_bases_t bases, size_t offset, const uint64_t* scalars_host, size_t n,
            int scalars_are_montgomery, uint64_t out_xyz[12]);
/* same with scalars already in device memory, enqueued on `stream` (a hipStream_t, may be 0);
 * synchronises the stream before returning the point to the host.                           */
int trh_msm_dev(trh_bases_t bases, size_t offset, const void* scalars_dev, size_t n,
                int scalars_are_montgomery, void* stream, uint64_t out_xyz[12]);
/* asynchronous halves of trh_msm_dev: enqueue leaves the per-window sums on the device,
 * finish synchronises, folds the windows on the host and returns the point.  One MSM in flight per
 * context (TRH_EBUSY otherwise); finish must name the base set and be called on the context of its enqueue.
 * Over a base set WITH fixed-base tables (trh_bases_precompute) the enqueue is not fully asynchronous: a sampler reads ~1024 scalars
 * and votes on the flag-like unit path, which synchronises `stream` once (everything queued on it before the call completes first);
 * option sparse = 0 (or a set without tables) keeps the enqueue free of host synchronisations.  With trh_set_timing(1) such an MSM
 * reports zero phase times (trh_last_timing): the sampler's paths are not timed phase by phase.                                  */
int trh_msm_dev_enqueue(trh_bases_t bases, size_t offset, const void* scalars_dev, size_t n,
                        int scalars_are_montgomery, void* stream);
int trh_msm_dev_finish(trh_bases_t bases, void* stream, uint64_t out_xyz[12]);
/* batch of MSMs over the same bases (one per column): scalars_dev holds batch x n x 4 u64.  Batches of >= 8 items run in chunks
 * of <= 64 and synchronise `stream` once or twice per chunk (a sampler's vote decides whether the chunk is flag-like and its +-1 digits
 * are summed straight from the table; the entry counts of the chunk are read back to size the accumulation's segments for sparse --
 * witness-shaped -- columns), so the call is not fully asynchronous even before the final hand-over of the points. */
int trh_msm_batch_dev(trh_bases_t bases, size_t offset, const void* scalars_dev, size_t n, size_t batch,
                      int scalars_are_montgomery, void* stream, uint64_t* out_xyz /* batch x 12 */);
/* Params::commit / commit_lagrange for `batch` polynomials already in device memory (batch x n x 4 u64, back to back):
 * item b is the MSM of polys[b] || blinds[b] over the handle's n + 1 bases (g or g_lagrange followed by w) -- the blind
 * is read from its own small array, so the polynomials are not copied to make room for it.  blinds: batch x 4 u64, host. */
int trh_commit_batch_dev(trh_bases_t bases, const void* polys_dev, size_t n, size_t batch, const uint64_t* blinds_host,
                         void* stream, uint64_t* out_xyz /* batch x 12 */);
/* the same for polynomials in HOST memory (polys_host[b]: n scalars, Montgomery): the columns cross PCIe in chunks, chunk
 * j + 1 while the batched MSM of chunk j runs */
int trh_commit_batch_host(trh_bases_t bases, const uint64_t* const* polys_host, size_t n, size_t batch, const uint64_t* blinds_host,
                          uint64_t* out_xyz /* batch x 12 */);
/* window width override for tuning (0 = automatic) */
int trh_msm_set_window_bits(int c); /* 0 or 2..18 */

/* sum of `count` points (Jacobian 12 x u64 each) on the host: combines the per-GPU partial
 * results of a range-sharded MSM after the all-gather.                                       */
int trh_point_sum(int curve, const uint64_t* points_xyz, size_t count, uint64_t out_xyz[12]);

/* ---- NTT on device memory (EvaluationDomain call sites) ----------------------------------- */
int trh_ntt_dev(int field, void* a_dev, uint32_t log_n, const uint64_t omega[4], size_t batch, void* stream);
/* a[i] *= factor (e.g. n^-1 after an inverse transform), device memory */
int trh_field_scale_dev(int field, void* a_dev, size_t n, const uint64_t factor[4], void* stream);
/* a[i] *= factors[i % period] (factors: period x 4 u64, host memory, period <= 64).  Covers
 * halo2's `distribute_powers_zeta` coset shift (period 3: 1, zeta, zeta^2) and
 * `divide_by_vanishing_poly` (period 2^(extended_k - k) table of (X^n - 1)^-1 values).        */
int trh_field_scale_periodic_dev(int field, void* a_dev, size_t n, const uint64_t* factors, uint32_t period, void* stream);
/* same for `rows` polynomials stored back to back (row_len elements each): only the first
 * active_len elements of each row are scaled and the period restarts with each row, i.e.
 * a[r][c] *= factors[c % period] for c < active_len (a batch of coeff_to_extended inputs).     */
int trh_field_scale_rows_dev(int field, void* a_dev, size_t rows, size_t row_len, size_t active_len,
                             const uint64_t* factors, uint32_t period, void* stream);

/* ---- halo2_proofs::poly::EvaluationDomain on device polynomials -------------------------------
 * trh_domain_create(field, j, k) mirrors EvaluationDomain::new(j, k): quotient_poly_degree = j - 1,
 * extended_k = smallest e with 2^e >= 2^k (j - 1); omega / extended_omega from ROOT_OF_UNITY, coset
 * generator ZETA, t_evaluations = (X^n - 1)^-1 on the coset.  Polynomials are `batch` rows of 2^k
 * (or 2^extended_k) field elements stored back to back in device memory.                        */
typedef struct trh_domain* trh_domain_t;
int trh_domain_create(int field, uint32_t j, uint32_t k, trh_domain_t* out);
/* Setup-time sizing, so that the FIRST proof of a process runs like every later one (the reference proves 1 - 10 circuits per process,
 * /root/reference/src/test_utils.rs:37-54, and its user waits for the first): the transforms' twiddle / coset-block tables (also built by
 * trh_domain_create) and their scratch for batches of `batch` polynomials, on the calling thread's context.                              */
int trh_domain_reserve(trh_domain_t d, size_t batch);
void trh_domain_destroy(trh_domain_t d);
uint32_t trh_domain_extended_k(trh_domain_t d);
/* which: 0 omega, 1 omega_inv, 2 extended_omega, 3 extended_omega_inv, 4 ifft_divisor,
 *        5 extended_ifft_divisor, 6 g_coset (zeta), 7 g_coset_inv (zeta^2); Montgomery limbs */
int trh_domain_constant(trh_domain_t d, int which, uint64_t out[4]);
int trh_domain_lagrange_to_coeff(trh_domain_t d, void* a_dev, size_t batch, void* stream);            /* in place */
int trh_domain_coeff_to_extended(trh_domain_t d, const void* coeff_dev, void* ext_dev, size_t batch, void* stream);
int trh_domain_extended_to_coeff(trh_domain_t d, void* a_dev, size_t batch, void* stream);            /* in place; caller truncates */
int trh_domain_divide_by_vanishing_poly(trh_domain_t d, void* a_dev, size_t batch, void* stream);     /* in place */

/* The extended domain as coset blocks: its 2^extended_k points zeta * extended_omega^i, i = q * 2^(extended_k - k) + r, are the
 * 2^(extended_k - k) cosets (zeta extended_omega^r) * omega^q of the size-2^k subgroup.  coeff_to_extended_blocks writes, for each
 * of `batch` polynomials, blocks r = 0 .. n_blocks - 1 (2^k values each, back to back): entry q of block r is entry
 * q * 2^(extended_k - k) + r of coeff_to_extended's output.  A block is a size-2^k transform (two passes at k = 18 instead of the
 * three of the zero-padded 2^21 one), Rotation(1) is q + 1 inside a block (trh_expr_eval_blocks_dev), and the quotient h(X) --
 * degree < (j - 1) 2^k -- is determined by j - 1 = trh_domain_quotient_blocks() blocks (5 of 8 for the reference's circuit), so the
 * prover needs only those of every column.  blocks_to_quotient: (j - 1) x 2^k values of the numerator on blocks 0 .. j - 2
 * (overwritten) -> the (j - 1) x 2^k coefficients of h(X) = numerator / (X^(2^k) - 1) (divide_by_vanishing = 1; with 0 the input
 * is taken as h's own values): what extended_to_coeff(divide_by_vanishing_poly(.)) returns, truncated to (j - 1) 2^k, whenever the
 * vanishing polynomial divides the numerator (it does in create_proof).                                                          */
uint32_t trh_domain_quotient_blocks(trh_domain_t d);
int trh_domain_coeff_to_extended_blocks(trh_domain_t d, const void* coeff_dev, void* ext_dev, size_t batch, uint32_t n_blocks, void* stream);
int trh_domain_blocks_to_quotient(trh_domain_t d, void* num_blocks_dev, void* h_coeff_dev, int divide_by_vanishing, void* stream);

/* the same on HOST polynomials (one pointer per column; what the Rust EvaluationDomain's methods take), pipelined over PCIe:
 * lagrange_to_coeff in place; coeff_to_extended reads 2^k coefficients and writes 2^extended_k values (only the non-zero
 * coefficients go up); extended_to_coeff in place on one polynomial of 2^extended_k values, optionally preceded by
 * divide_by_vanishing_poly (the two steps create_proof applies to h(X))                                                     */
int trh_domain_lagrange_to_coeff_host(trh_domain_t d, uint64_t* const* a, size_t count);
int trh_domain_coeff_to_extended_host(trh_domain_t d, const uint64_t* const* coeff, uint64_t* const* ext, size_t count);
int trh_domain_extended_to_coeff_host(trh_domain_t d, uint64_t* a, int divide_by_vanishing_first);
/* the coset-block forms on host polynomials, for a host-side h(X) evaluation that has adopted the block layout: n_blocks x 2^k values
 * per column come down instead of 2^extended_k (5/8 of the bytes when n_blocks = trh_domain_quotient_blocks()), the quotient's numerator
 * goes up as (j - 1) x 2^k values and h(X)'s (j - 1) x 2^k coefficients come back                                                     */
int trh_domain_coeff_to_extended_blocks_host(trh_domain_t d, const uint64_t* const* coeff, uint64_t* const* ext, size_t count, uint32_t n_blocks);
int trh_domain_blocks_to_quotient_host(trh_domain_t d, const uint64_t* num_blocks, uint64_t* h_coeff, int divide_by_vanishing);

/* ---- IPA opening rounds: poly::commitment::prover::create_proof (device memory) ---------------
 * The two half-size MSMs of a round run through trh_msm_dev on the live G' buffer
 * (trh_bases_wrap_device + offset); these are the remaining per-round primitives.            */
/* halo2_proofs::arithmetic::compute_inner_product(a, b): sum a[i] * b[i] -> out (Montgomery) */
int trh_field_inner_product_dev(int field, const void* a_dev, const void* b_dev, size_t n, void* stream, uint64_t out[4]);
/* halo2_proofs::arithmetic::eval_polynomial for `batch` polynomials (n coefficients each, back to back in device memory)
 * at one point -- the evaluations create_proof sends before the multiopen argument; out: batch x 4 u64 on the host */
int trh_poly_eval_batch_dev(int field, const void* polys_dev, size_t n, size_t batch, const uint64_t point[4], void* stream, uint64_t* out);
/* y[i] += c * x[i]: the folds p'[i] += u^-1 p'[i + half], b[i] += u b[i + half] */
int trh_field_axpy_dev(int field, void* y_dev, const void* x_dev, size_t n, const uint64_t c_mont[4], void* stream);
/* out[i] = x^i for i < n: the evaluation vector b = (1, x3, x3^2, ...) */
int trh_field_powers_dev(int field, void* out_dev, size_t n, const uint64_t x_mont[4], void* stream);
/* parallel_generator_collapse: g_lo[i] = g_lo[i] + u * g_hi[i], normalised to affine (u: scalar
 * field element, Montgomery; g_*: `half` 64-byte affine PODs)                                   */
int trh_bases_fold_dev(int curve, void* g_lo_dev, const void* g_hi_dev, size_t half, const uint64_t u_mont[4], void* stream);

/* The whole opening in one call: poly::commitment::prover::create_proof(params, rng, transcript,
 * p_poly, p_blind, x_3).  p', b and G' never leave the device; per round only L_j, R_j and the
 * challenge cross the boundary, through the caller's transcript (BLAKE2b on the Rust side) and
 * randomness callbacks.  g_w: resident bases g (2^k points) followed by w -- or g, w and u (2^k + 2 points; the last one must
 * equal u_xy): with fixed-base tables attached to that set (trh_bases_precompute, once per Params) every MSM of the opening runs
 * in fixed-base mode (k = 18: 20 -> 15.5 ms).  u_xy: Params.u;
 * p_poly_dev / s_poly_dev: 2^k coefficients in device memory (s_poly: the caller's random polynomial,
 * its constant term is adjusted here so that s(x3) = 0); scalars are Montgomery limbs.
 * Writes to the transcript exactly what the Rust prover writes: S, then L_j, R_j per round, then c, f.
 * A zero round challenge (the Rust prover's `u_j.invert().unwrap()` panics) returns TRH_EINVAL.          */
typedef struct trh_transcript {
    void* ctx;
    void (*write_point)(void* ctx, const uint64_t xyz[12]);           /* normalised Jacobian, Z = 1 */
    void (*write_scalar)(void* ctx, const uint64_t s_mont[4]);
    void (*squeeze_challenge_scalar)(void* ctx, uint64_t out_mont[4]);
} trh_transcript_t;
typedef void (*trh_rng_scalar_fn)(void* ctx, uint64_t out_mont[4]);    /* C::Scalar::random(rng) */
int trh_ipa_create_proof(trh_bases_t g_w, const uint64_t u_xy[8], uint32_t k, const void* p_poly_dev,
                         const uint64_t p_blind[4], const uint64_t x3[4], const void* s_poly_dev,
                         const uint64_t s_blind[4], const trh_transcript_t* transcript,
                         trh_rng_scalar_fn rng, void* rng_ctx, void* stream,
                         uint64_t out_c[4], uint64_t out_f[4]);

/* ---- grand-product building blocks of the permutation / lookup arguments ----------------------
 * (plonk/permutation/prover.rs, plonk/lookup/prover.rs: batch_invert of the denominators, then the
 * running product z[0] = 1, z[i] = z[i-1] * numerator[i-1] / denominator[i-1])                    */
/* ff::BatchInvert: a[i] <- a[i]^-1 in place, zeros stay zero */
int trh_field_batch_invert_dev(int field, void* a_dev, size_t n, void* stream);
/* the division the product columns need, in the same pass: a[i] <- num[i] * a[i]^-1 (a zero denominator stays zero) */
int trh_field_batch_invert_mul_dev(int field, void* a_dev, const void* num_dev, size_t n, void* stream);
/* the products the running product is taken over, for every product column of a proof in ONE launch:
 *     out[r][i] = prod_{t in row r} (x_t[i] + c_t * y_t[i] + g_t)      (y_t NULL: x_t[i] + g_t),   i < n
 * row r owns terms[row_start[r] .. row_start[r + 1]) (row_start has rows + 1 entries, row_start[0] = 0).  Permutation chunk
 * (plonk/permutation/prover.rs Argument::commit): numerator terms (v_j, omega^i column, beta * delta^(col j), gamma), denominator
 * terms (v_j, sigma_j, beta, gamma); lookup product (plonk/lookup/prover.rs commit_product): numerator (A, -, -, beta)(S, -, -, gamma),
 * denominator (A', -, -, beta)(S', -, -, gamma).  c / g: Montgomery words; columns: n elements in device memory.              */
typedef struct trh_product_term {
    const void* x;
    const void* y;
    uint64_t c[4];
    uint64_t g[4];
} trh_product_term_t;
int trh_product_terms_dev(int field, const trh_product_term_t* terms, const uint32_t* row_start, uint32_t rows, size_t n, void* out_dev, void* stream);
/* out[i] = prod_{j < i} a[j], out[0] = 1 (exclusive scan; out must not alias a) */
int trh_field_prefix_product_dev(int field, const void* a_dev, void* out_dev, size_t n, void* stream);
/* the same for `rows` independent vectors of n elements stored back to back (all product columns of a proof at once) */
int trh_field_prefix_product_rows_dev(int field, const void* a_dev, void* out_dev, size_t n, size_t rows, void* stream);
/* out[i] = sum_{j < i} a[j], out[0] = 0 */
int trh_field_prefix_sum_dev(int field, const void* a_dev, void* out_dev, size_t n, void* stream);

/* ---- multiopen building blocks (poly/multiopen/prover.rs: the x1 / x4 linear combinations of the queried polynomials
 *      and the division of (q(X) - r(X)) by the (X - point) factors) ------------------------------------------------- */
/* out[i] = sum_b coeffs[b] * polys[b][i]; polys: batch x n back to back in device memory, coeffs: batch x 4 u64 on the host */
int trh_poly_lincomb_dev(int field, const void* polys_dev, size_t n, size_t batch, const uint64_t* coeffs_host, void* out_dev, void* stream);
/* arithmetic::kate_division(a, z): quotient of a(X) (n coefficients) by (X - z), n - 1 coefficients, remainder a(z) dropped.
 * pz_dev / pzinv_dev: z^i and z^-i for i < n (trh_field_powers_dev; shared by every polynomial divided at this point, z != 0),
 * scratch_dev: 2 n elements.                                                                                          */
int trh_poly_kate_division_dev(int field, const void* a_dev, size_t n, const void* pz_dev, const void* pzinv_dev, void* scratch_dev, void* q_dev, void* stream);

/* ---- lookup argument: plonk/lookup/prover.rs `permute_expression_pair` ------------------------------------------
 * out_input = the first usable_rows input values sorted (field Ord = canonical integer order); out_table[row] = out_input[row]
 * where that row starts a run of equal values (one instance of the value leaves the table multiset), the remaining rows take
 * the left-over table values in ascending order starting from the LAST repeated row, exactly as the Rust code fills them.
 * Returns TRH_EINVAL when an input value does not occur in the table (halo2: Error::ConstraintSystemFailure).  The blinding
 * rows behind usable_rows are the caller's.  Synchronises the stream.                                                       */
int trh_lookup_permute_dev(int field, const void* input_dev, const void* table_dev, size_t usable_rows, void* out_input_dev,
                           void* out_table_dev, void* stream);
/* every lookup of a proof at once (the reference's circuit has 31): `batch` input columns and `batch` table columns, column l at
 * element offset l * row_stride (row_stride >= usable_rows: whole 2^k-row columns can be passed with usable_rows < 2^k; the rows behind
 * usable_rows are not touched in the outputs, which use the same stride).  One set of launches and one host synchronisation for
 * the whole batch; TRH_EINVAL names the first lookup with an input value that does not occur in its table.                        */
int trh_lookup_permute_batch_dev(int field, const void* inputs_dev, const void* tables_dev, size_t usable_rows, size_t row_stride, size_t batch,
                                 void* out_inputs_dev, void* out_tables_dev, void* stream);

/* ---- gate expressions over resident columns: the h(X) numerator of plonk::create_proof ---------
 * halo2's `Expression<F>` (Constant / Selector / Fixed / Advice / Instance query at a Rotation, Negated, Sum,
 * Product, Scaled) compiled by the caller to a straight-line program for a stack machine that every row of the
 * (extended) domain runs on its own: PUSH_* push a value, ADD / SUB / MUL replace the two top entries (next op top),
 * NEG / SQR / MUL_CONST / ADD_CONST rewrite the top, FOLD does acc = acc * const[a] + top and pops (the challenge-y
 * Horner over gates), STORE_TOP / STORE_ACC write output a.  Locals hold shared sub-expressions.
 * A column query reads column[(row + rotation * rot_step) mod 2^log_n]: on the extended coset Rotation(r) is
 * r * 2^(extended_k - k) rows.  Constants are Montgomery limbs.                                           */
enum {
    TRH_EXPR_PUSH_COLUMN = 0, /* a = column index, rotation */
    TRH_EXPR_PUSH_CONST = 1,  /* a = constant index */
    TRH_EXPR_PUSH_LOCAL = 2,  /* a = local index */
    TRH_EXPR_ADD = 3,
    TRH_EXPR_SUB = 4,         /* next - top */
    TRH_EXPR_MUL = 5,
    TRH_EXPR_NEG = 6,
    TRH_EXPR_SQR = 7,
    TRH_EXPR_MUL_CONST = 8,   /* a = constant index */
    TRH_EXPR_ADD_CONST = 9,
    TRH_EXPR_STORE_LOCAL = 10, /* local[a] = top (kept) */
    TRH_EXPR_FOLD = 11,       /* acc = acc * const[a] + top; pop */
    TRH_EXPR_STORE_TOP = 12,  /* output[a][row] = top; pop */
    TRH_EXPR_STORE_ACC = 13   /* output[a][row] = acc */
};
typedef struct {
    uint32_t op;
    uint32_t a;
    int32_t rotation;
} trh_expr_insn_t;
typedef struct trh_expr* trh_expr_t;
/* validates the program (stack discipline, index ranges), fixes the LDS slot of every spill / refill and uploads it */
int trh_expr_create(int field, const trh_expr_insn_t* insns, size_t n_insn, const uint64_t* consts /* n_consts x 4 */, size_t n_consts,
                    size_t n_columns, size_t n_outputs, size_t n_locals, trh_expr_t* out);
void trh_expr_destroy(trh_expr_t e);
uint32_t trh_expr_lds_slots(trh_expr_t e); /* stack entries below the two register ones + locals */
/* replace one constant: the per-proof challenges (y, beta, gamma, theta) of an otherwise fixed program */
int trh_expr_set_const(trh_expr_t e, uint32_t index, const uint64_t value[4]);
/* columns_dev / outputs_dev: host arrays of device pointers (2^log_n x 4 u64 each); synchronises the stream */
int trh_expr_eval_dev(trh_expr_t e, const void* const* columns_dev, void* const* outputs_dev, uint32_t log_n, uint32_t rot_step, void* stream);
/* the same over columns in the coset-block layout of trh_domain_coeff_to_extended_blocks: n_blocks x 2^block_log rows per column,
 * Rotation(r) reads row (q + r) mod 2^block_log of the same block */
int trh_expr_eval_blocks_dev(trh_expr_t e, const void* const* columns_dev, void* const* outputs_dev, uint32_t block_log, uint32_t n_blocks, void* stream);

/* ---- best_fft over curve points: Params::new's g -> g_lagrange -----------------------------
 * halo2_proofs::arithmetic::best_fft::<C::Curve>(a, omega, log_n): a'[i] = sum_j [omega^(i j)] a[j].
 * points_dev: 2^log_n affine PODs in device memory, transformed in place (natural order in and out)
 * and normalised to affine; omega: scalar field element, Montgomery, primitive 2^log_n-th root;
 * scale_or_null: when non-NULL every output is multiplied by this scalar (Params::new uses n^-1).  */
int trh_point_fft_dev(int curve, void* points_dev, uint32_t log_n, const uint64_t omega[4],
                      const uint64_t* scale_or_null, void* stream);

/* ---- element-wise field / group ops on device memory (parity tests of the device arithmetic;
 *      op: 0 add, 1 sub, 2 mul, 3 sqr, 4 neg, 5 inv, 6 to_mont, 7 from_mont) ------------------ */
int trh_field_op_dev(int field, int op, const void* a_dev, const void* b_dev, void* out_dev, size_t n, void* stream);
/* op: 0 out = p + q (both Jacobian 12 x u64), 1 out = p + q (q affine 8 x u64), 2 out = 2p; 3 / 4: p + q / 2p through the quad-lane
 * arithmetic of the small MSMs' bucket reduction (csrc/curve_q4.h); out Jacobian normalised */
int trh_point_op_dev(int curve, int op, const void* p_dev, const void* q_dev, void* out_dev, size_t n, void* stream);

/* ---- plain device memory helpers so that non-HIP hosts (Rust, ctypes) can stage buffers ----
 * trh_free keeps freed blocks for reuse by trh_malloc (per device, by rounded size; it returns when the device has drained, as
 * hipFree does): at most `pool_mb` MiB (trh_set_option / TRH_POOL_MB, default 4096; 0 switches the pool off), evicting the device's largest idle
 * blocks first.  The idle blocks are invisible to any other allocator in the process (torch's caching allocator, say): libtrh
 * returns them when one of its own allocations fails, and trh_pool_trim() returns them on request.  Freeing a block twice is
 * reported (TRH_EINVAL) while it still waits in the pool. */
int trh_malloc(void** dev, size_t bytes);
int trh_free(void* dev);
int trh_pool_trim(void);
size_t trh_pool_idle_bytes(void);
int trh_memcpy_h2d(void* dev, const void* host, size_t bytes);
int trh_memcpy_d2h(void* host, const void* dev, size_t bytes);
int trh_stream_synchronize(void* stream);
/* counters of the calling thread's context, by name.  "msm_lean_retries": MSMs of callers that vouched for uniformly random scalars (the
 * opening's rounds) whose whole-bin LDS sort met a bin that did not fit and that were therefore run a second time with the chunked passes;
 * "msm_small_launches": MSMs that ran as ONE launch (up to 8448 pairs, batches of up to four); "ipa_generator_collapses": openings that
 * collapsed their generators from the fixed-base table (option ipa_fold). */
int trh_stat(const char* name, uint64_t* value);
/* GPU-side timing without HIP headers: events recorded on a stream between the steps (a hipEvent_t each), read afterwards.
 * trh_event_elapsed_ms waits for `end` and returns the device time between the two records. */
int trh_event_create(void** out_event);
int trh_event_record(void* event, void* stream);
int trh_event_elapsed_ms(void* start, void* end, float* ms);
void trh_event_destroy(void* event);

/* ---- timing of the last MSM / NTT on this context (HIP events on the launch stream) -------- */
typedef struct trh_timing {
    float total_ms;        /* first kernel start -> last kernel end */
    float digits_ms;       /* MSM: scalar recode + histogram */
    float sort_ms;         /* MSM: bucket offsets + scatter */
    float accumulate_ms;   /* MSM: bucket accumulation (the dominant kernel) */
    float reduce_ms;       /* MSM: bucket reduction + window sums */
    int window_bits;
    int windows;
    float accumulate_kernel_ms; /* msm_accumulate_seg_kernel alone (accumulate_ms also covers the per-bucket combine) */
} trh_timing_t;
int trh_set_timing(int enabled);
int trh_last_timing(trh_timing_t* out);

#ifdef __cplusplus
}
#endif
#endif /* TRH_H */
