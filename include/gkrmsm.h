/* gkrmsm.h -- C ABI of libgkrmsm_hip.so: the MI355X (gfx950) drop-in for the Pippenger-MSM +
 * GKR-sumcheck hot path of morgana-proofs/GKR-MSM.
 *
 * The reference is a pure-Rust crate with no FFI of its own; every entry point below names the Rust
 * seam (file:line under /root/reference) that a `extern "C"` shim would route here (INTEGRATION.md
 * shows the shim).  Conventions:
 *   - every function returns 0 on success, non-zero GM_ERR_* otherwise; gm_last_error() gives text
 *     (the reference convention is panic!/assert!, the Rust shim turns non-zero into panic!);
 *   - field elements are BLS12-381 Fr in the reference's in-memory form: Montgomery (R = 2^256),
 *     4 x u64 little-endian limbs, 32 bytes, i.e. `&[Fr]` passes as `*const u64` unchanged;
 *   - Bandersnatch points are twisted-Edwards `Affine{x,y}` = 64 bytes (x then y);
 *   - pointers named d_* are DEVICE pointers (hipMalloc / torch data_ptr), h_* are host pointers;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); no call synchronises the
 *     device unless it returns data into an h_* buffer;
 *   - no hipMalloc/hipFree happens inside *_run / round / bind calls (workspaces live in handles).
 */
#ifndef GKRMSM_H
#define GKRMSM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GM_OK 0
#define GM_ERR_INVALID 1   /* bad argument (mirrors an assert! in the reference) */
#define GM_ERR_HIP 2       /* HIP runtime failure */
#define GM_ERR_NO_DEVICE 3 /* no gfx950 device visible */
#define GM_ERR_STATE 4     /* call order violated (e.g. bind before unipoly: vecvec_eq.rs:305-307) */
#define GM_ERR_VERIFY 5    /* the verifier rejected the proof (an assert! in the reference's verify functions) */

/* ---------------------------------------------------------------- runtime
 * Threads.  Handles are not shared between host threads while a call on them is in progress, but different threads may work
 * on different handles at the same time (a proving service: one thread = one stream + one plan / witness; rayon workers in
 * the reference play the same role).  Everything the library keeps between calls is per thread (pinned staging, the host
 * tables, the last error) or behind a lock (the device-memory cache, which hands a freed block only to the thread that
 * freed it; the co-residency budget of the persistent round kernel; the G1 engine's scratch and fixed-base registry: ONE G1 call
 * at a time per device, so whole proofs from several threads overlap one proof's G1 work with the others' sumcheck rounds).
 * A thread must outlive the handles it created.
 * bench.py's `sumcheck.concurrent_provers` / `full_gen2_prover.concurrent_provers`, tests/test_prover_gpu.py::
 * test_provers_on_concurrent_host_threads and tests/test_pippenger_full_gpu.py::test_whole_proofs_on_concurrent_host_threads use this. */
const char* gm_last_error(void);
const char* gm_version(void);
int32_t gm_device_count(int32_t* out_count);
int32_t gm_set_device(int32_t device);
int32_t gm_stream_sync(void* stream);
/* a hipStream_t of the current device for callers that do not link the HIP runtime themselves (plain C, the Rust shim); created
 * non-blocking: it never synchronises with the default stream.  Every `void* stream` argument of this header takes it (or any
 * hipStream_t of the caller's; NULL = the default stream). */
int32_t gm_stream_create(void** out_stream);
int32_t gm_stream_destroy(void* stream);
/* plain device memory helpers for non-torch callers (the Rust shim) */
int32_t gm_malloc(void** out_d_ptr, size_t bytes);
int32_t gm_free(void* d_ptr);
/* Handles and workspaces keep the large device blocks they free on an idle list (re-allocating freshly freed HBM is slow
 * and hipFree synchronises); this returns the idle blocks to the driver. */
int32_t gm_release_cached_memory(void);
/* gm_reserve: take `bytes` of device memory from the driver now -- set-up time, next to the SRS load -- for the library's pool to cut
 * its blocks from.  The driver hands memory out at ~25-40 GiB/s, so a process's FIRST proof otherwise pays for every byte of its
 * trace (gen-1 at 2^20 x 2^8: 5.0 s against 0.33 s in steady state).  Per device; additive; what the reserve cannot serve still
 * comes from the driver.  gm_unreserve gives the reserve back (GM_ERR_STATE while handles hold blocks of it).
 * gm_memory_stats: out8 = {driver allocations so far, their bytes, bytes idling in the pool, bytes reserved, of which in use, blocks
 * cut from the reserve so far, 0, 0}. */
int32_t gm_reserve(uint64_t bytes);
int32_t gm_unreserve(void);
int32_t gm_memory_stats(uint64_t* out8);
/* Small sumcheck rounds keep kernels WAITING on the device for the caller's next challenge (a one-wave gate in front of a
 * pre-enqueued fold; the persistent tail-round kernel), and the host waits for their results.  Every such wait is bounded: after
 * `ms` milliseconds (default 20000; 0 restores the default) the waiting kernel flags a status word and leaves, and the call in
 * progress returns GM_ERR_STATE -- a caller whose transcript dies never wedges the GPU.  Process-wide; affects later launches. */
int32_t gm_set_wait_timeout_ms(uint32_t ms);
/* Diagnostics of the co-residency budget the persistent round kernel's launches share ("Threads" above): workgroups this process
 * has in flight on the current device and the budget (occupancy x compute units).  0 in flight whenever no proof is running. */
int32_t gm_stage_slots(uint32_t* in_flight, uint32_t* capacity);
int32_t gm_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes, void* stream);
int32_t gm_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes, void* stream);
int32_t gm_memcpy_d2d(void* d_dst, const void* d_src, size_t bytes, void* stream); /* asynchronous */

/* ---------------------------------------------------------------- AlgFn descriptors
 * Replaces the Rust generics `Fun: AlgFn<F>` (src/cleanup/utils/algfn.rs:21-34).  A function is a
 * left-to-right stack of up to 4 segments (primitive id, repeat count):
 *   StackedAlgFn::new(f1, RepeatedAlgFn::new(f2, n))  ==  nseg=2, prim={f1,f2}, count={1,n}
 * Primitive ids: twisted_edwards_ops.rs:150-156 plus algfn.rs Id/BitCheck and gen-1 pt_bit_choice. */
#define GM_FN_AFF_L1 1   /* affine_twisted_edwards_add_l1   (deg 2, 4 -> 3)  */
#define GM_FN_AFF_L2 2   /* affine_twisted_edwards_add_l2   (deg 2, 3 -> 3)  */
#define GM_FN_AFF_L3 3   /* affine_twisted_edwards_add_l3   (deg 2, 3 -> 3)  */
#define GM_FN_PROJ_L1 4  /* twisted_edwards_add_l1          (deg 2, 6 -> 4)  */
#define GM_FN_PROJ_L2 5  /* twisted_edwards_add_l2          (deg 2, 4 -> 4)  */
#define GM_FN_PROJ_L3 6  /* twisted_edwards_add_l3          (deg 2, 4 -> 3)  */
#define GM_FN_TRI_L1 7   /* triangle_twisted_edwards_add_l1 (deg 2, 12 -> 12) */
#define GM_FN_ID 8       /* IdAlgFn::new(1); use count=n for IdAlgFn::new(n) */
#define GM_FN_BITCHECK 9 /* BitCheckFn */
#define GM_FN_PT_BIT_CHOICE 10 /* gkr_msm_simple.rs:82-84 */
#define GM_FN_ADD_INVERSES 11  /* AddInversesFn  pushforward/pushforward.rs:255-281  (deg 2, 2 -> 2) */
#define GM_FN_LOGUP_LAYER 12   /* LogupLayerFn   pushforward/logup_mainphase.rs:30-61 (deg 2, 4 -> 2) */
#define GM_FN_MAX_SEG 4

typedef struct gm_fn {
    int32_t nseg;
    int32_t prim[GM_FN_MAX_SEG];
    int32_t count[GM_FN_MAX_SEG];
} gm_fn;

int32_t gm_fn_shape(const gm_fn* f, int32_t* n_ins, int32_t* n_outs, int32_t* deg);

/* ---------------------------------------------------------------- field batch ops (a1)
 * Elementwise Fr arithmetic over device arrays; replaces ark-ff operator calls inside the
 * reference's rayon loops (src/utils.rs:22-49).  op: 0 add, 1 sub, 2 mul, 3 neg(a), 4 inverse(a),
 * 5 to-Montgomery(a), 6 from-Montgomery(a), 7 mul_by_a(a) = -5a, 8 mul_by_d(a). */
int32_t gm_fr_batch(int32_t op, const uint64_t* d_a, const uint64_t* d_b, uint64_t* d_out, uint64_t n,
                    void* stream);
/* same ops on host memory, scalar code (host glue; used by the CPU-side tests) */
int32_t gm_fr_host(int32_t op, const uint64_t* h_a, const uint64_t* h_b, uint64_t* h_out, uint64_t n);
/* exec an AlgFn on host rows: h_in = n rows x n_ins elements, h_out = n rows x n_outs elements */
int32_t gm_fn_host(const gm_fn* f, const uint64_t* h_in, uint64_t* h_out, uint64_t n);

/* ---------------------------------------------------------------- dense polynomials (a2, a3, a4, a5)
 * Columns are separate device arrays of 2^n elements; `d_in` / `d_out` are HOST arrays of device pointers.
 *   gm_dense_map        Vec::algfn_map                 cleanup/polys/dense.rs:141-184   (AlgFnUtils::map algfn.rs:55-80)
 *   gm_dense_map_split  Vec::algfn_map_split           cleanup/polys/dense.rs:115-139   split on index bit
 *                       `split_lo_bit` (= SplitIdx::lo_usize), outputs 2*n_outs columns of len/2 in the
 *                       bundle-interleaved order [L-bundle, R-bundle, ...] of dense.rs:137-138
 *   gm_dense_bind       bind_dense_poly                cleanup/protocols/sumcheck.rs:160-163 (== bind_21 on 21-form,
 *                       dense.rs:54-61): out[c][i] = in[c][2i] + t (in[c][2i+1] - in[c][2i]) for k columns
 *   gm_eq_table         eq_poly_sequence_from_multiplier(..).last()   utils.rs:222-250; point[0] is the MSB;
 *                       d_scratch: 2^nvars elements of workspace (holds the lower levels) */
int32_t gm_dense_map(const gm_fn* f, const uint64_t* const* d_in, uint64_t* const* d_out, uint64_t len, void* stream);
int32_t gm_dense_map_split(const gm_fn* f, const uint64_t* const* d_in, uint64_t* const* d_out, uint64_t len,
                           uint32_t split_lo_bit, uint32_t bundle, void* stream);
int32_t gm_dense_bind(const uint64_t* const* d_in, uint64_t* const* d_out, uint32_t k, uint64_t len,
                      const uint64_t* h_t, void* stream);
int32_t gm_eq_table(const uint64_t* h_multiplier, const uint64_t* h_point, uint32_t nvars, uint64_t* d_scratch,
                    uint64_t* d_out, void* stream);

/* ---------------------------------------------------------------- VecVec polynomials (a2, a3, a12)
 * gm_vv = k `VecVecPolynomial`s sharing one row structure (cleanup/polys/vecvec.rs:149-160): rows stored back to
 * back on the device, odd rows padded with row_pad (vecvec.rs:181-186).
 *   gm_vv_from_host           VecVecPolynomial::new for k polys (h_data[c] = rows of poly c concatenated, unpadded)
 *   gm_vv_from_msm            the bucket image of PushForwardState::new (pushforward.rs:342-349, 380-381, 411-426,
 *                             477-487) from the last gm_msm_run: polys (x, y, z), pads (0, 1, 0)
 *   gm_vv_map                 vecvec_map                    vecvec.rs:480-540
 *   gm_vv_map_split           vecvec_map_split, LO(0)       vecvec.rs:542-606   (output: 2*n_outs polys, bundled)
 *   gm_vv_map_split_to_dense  vecvec_map_split_to_dense     vecvec.rs:608-654   (d_out: 2*n_outs columns of 2^col_logsize)
 *   gm_vv_slice / gm_vv_concat  &polys[a..b] / Vec::extend as used by GlueSplit::witness (splits.rs:172-176)
 *   gm_vv_to_dense            Densify::to_dense             vecvec.rs:446-476 */
typedef struct gm_vv gm_vv;
struct gm_msm_plan;   /* the MSM plan handle, declared with the MSM section below */
int32_t gm_vv_from_host(uint32_t k, uint32_t nrows, const uint32_t* h_row_len, const uint64_t* const* h_data,
                        const uint64_t* h_row_pad, const uint64_t* h_col_pad, uint32_t row_logsize,
                        uint32_t col_logsize, gm_vv** out, void* stream);
int32_t gm_vv_from_msm(const struct gm_msm_plan* plan, const uint64_t* d_points_xy, uint32_t y_logsize, gm_vv** out,
                       void* stream);
int32_t gm_vv_map(const gm_fn* f, const gm_vv* in, gm_vv** out, void* stream);
int32_t gm_vv_map_split(const gm_fn* f, const gm_vv* in, uint32_t bundle, gm_vv** out, void* stream);
int32_t gm_vv_map_split_to_dense(const gm_fn* f, const gm_vv* in, uint32_t bundle, uint64_t* const* d_out,
                                 void* stream);
int32_t gm_vv_slice(const gm_vv* in, uint32_t first, uint32_t count, gm_vv** out);
int32_t gm_vv_concat(const gm_vv* a, const gm_vv* b, gm_vv** out);
int32_t gm_vv_to_dense(const gm_vv* v, uint64_t* const* d_out, void* stream);
int32_t gm_vv_info(const gm_vv* v, uint32_t* k, uint32_t* nrows, uint64_t* total_cells, uint32_t* row_logsize,
                   uint32_t* col_logsize);
int32_t gm_vv_read(const gm_vv* v, uint32_t col, uint32_t* h_off, uint64_t* h_cells, void* stream);
int32_t gm_vv_pads(const gm_vv* v, uint64_t* h_row_pad, uint64_t* h_col_pad);
int32_t gm_vv_destroy(gm_vv* v);

/* ---------------------------------------------------------------- sumcheck objects (a6-a9)
 * The `Sumcheckable::{unipoly, bind, final_evals}` seam (cleanup/protocols/sumchecks/vecvec_eq.rs:218-225) called
 * by GenericSumcheckProtocol::prove (cleanup/protocols/sumcheck.rs:101-123):
 *   gm_sc_dense_deg2_create   DenseDeg2SumcheckObject::new + rlc(gamma)   dense_eq.rs:27-59  -> ..ObjectSO :61-173
 *   gm_sc_vecvec_deg2_create  VecVecDeg2SumcheckObject::new + rlc(gamma)  vecvec_eq.rs:35-70 -> ..ObjectSO :72-398
 *                             (sparse rounds, then bind_into_dense :157-190 and the dense rounds)
 *   gm_sc_dense_create        DenseSumcheckObjectSO::new                  sumcheck.rs:248-257; kind 0: F = EqWrapper(
 *                             GammaWrapper(f, gamma)) over f.n_ins columns + the eq column (:706-829), kind 1: Prod3Fn
 *                             (pushforward.rs:27-49), kind 2: FoldedProdAlgFn(gamma, nargs) (multiopen_reduction.rs:13-42)
 *                             over nargs polynomials followed by nargs eq tables, f = {GM_FN_ID, count nargs}
 * h_claims: n_outs evaluation claims (folded with gamma as in rlc); h_point: num_vars elements, point[0] = MSB.
 * Input columns are read, never written.  gm_sc_unipoly returns the deg+1 coefficients (low to high) of the
 * round polynomial, i.e. `unipoly().as_vec()`; the caller drops the linear one (compress_coefficients :27-31). */
typedef struct gm_sc gm_sc;
int32_t gm_sc_dense_deg2_create(const gm_fn* f, uint32_t num_vars, const uint64_t* const* d_cols,
                                const uint64_t* h_point, const uint64_t* h_gamma, const uint64_t* h_claims, gm_sc** out,
                                void* stream);
int32_t gm_sc_vecvec_deg2_create(const gm_fn* f, const gm_vv* polys, const uint64_t* h_point, const uint64_t* h_gamma,
                                 const uint64_t* h_claims, gm_sc** out, void* stream);
int32_t gm_sc_dense_create(int32_t kind, const gm_fn* f, uint32_t num_vars, const uint64_t* const* d_cols,
                           const uint64_t* h_gamma, const uint64_t* h_claim, gm_sc** out, void* stream);
int32_t gm_sc_unipoly(gm_sc* so, uint64_t* h_coeffs, uint32_t* n_coeffs);
int32_t gm_sc_bind(gm_sc* so, const uint64_t* h_t);
int32_t gm_sc_final_evals(gm_sc* so, uint64_t* h_evals, uint32_t* n_evals);
int32_t gm_sc_claim(const gm_sc* so, uint64_t* h_claim);
int32_t gm_sc_destroy(gm_sc* so);

/* Measurement: the large (non-split) round kernels of the sumcheck objects created on the calling thread are
 * bracketed with HIP events on their launch stream.  mode 0 = off, 1 = time those launches (cheap: two event records per large
 * round; exact pair counts of sparse rounds come back by a 4-byte asynchronous copy), 2 = additionally account the
 * algorithmic bytes of every other round kernel and of every fold.  gm_sc_profile_read synchronises `stream`, returns one row
 * per kernel class (since the last read / mode change) and the two byte totals, and resets the records.
 * Algorithmic bytes: a round kernel reads both cells of every pair of its k input columns (64 k bytes per pair, + 32 for the eq
 * weight of the eq-factored objects); a fold moves 96 bytes per output cell and column. */
typedef struct gm_sc_profile_row {
    char kernel[64];     /* e.g. "k_round_deg2_lean<PROJ_L1,vecvec>" */
    uint32_t launches;
    uint32_t k_cols;
    double total_ms;     /* sum of the launches' durations (HIP events) */
    double max_ms;
    double pairs;        /* pairs processed over all launches */
    double alg_bytes;    /* algorithmic bytes read over all launches */
    double fr_mul;       /* field multiplications over all launches */
    double max_ms_pairs; /* pairs of the launch that took max_ms (the largest one: its own roofline figure) */
} gm_sc_profile_row;
/* k_stage (persistent stage kernel) launches of this process so far and how many of them left before their first round (residency
 * barrier; sharded: another rank's launch failed) -- which path a proof took, for tests and benches */
int32_t gm_sc_stage_counts(uint64_t* launched, uint64_t* left_early);
int32_t gm_sc_profile(int32_t mode);
int32_t gm_sc_profile_read(gm_sc_profile_row* rows, uint32_t cap, uint32_t* n_rows, double* other_round_bytes, double* fold_bytes,
                           void* stream);

/* ---------------------------------------------------------------- Pippenger MSM (a11, a12, a14)
 * The reference's bucketed MSM over Bandersnatch:
 *   digits + bucket scatter      PushForwardState::new      pushforward/pushforward.rs:351-361, 401-429
 *   bucket sums (pairwise tree)  bintree_add witness        gkrs/bintree_add.rs:137-239
 *   bucket reduction             triangle_add witness       gkrs/triangle_add.rs:101-158, pippenger_ending.rs:47-58
 *   final recombination          verify_pippenger           pippenger.rs:586-602
 * and, as a whole, the `VariableBaseMsmNonaffine::msm_bigint_nonaff(bases, bigints) -> G` call shape
 * (msm_nonaffine.rs:41-50) for G = Bandersnatch.
 *
 * A plan owns all workspaces for one shape; windows [y_begin, y_end) of the y_size windows are
 * processed by this plan (multi-GPU: one plan per rank, windows partitioned, no data-path exchange
 * until the (d+1)-points-per-window outputs are gathered). */
typedef struct gm_msm_plan gm_msm_plan;

int32_t gm_msm_plan_create(uint32_t x_logsize, uint32_t d_logsize, uint32_t y_size, uint32_t y_begin,
                           uint32_t y_end, gm_msm_plan** out);
int32_t gm_msm_plan_destroy(gm_msm_plan* plan);
size_t gm_msm_plan_workspace_bytes(const gm_msm_plan* plan);

/* scalars: canonical bigints (what `into_bigint()` yields, msm_nonaffine.rs:21-23), 4 x u64 LE each.
 * Runs digits -> bucketize -> bucket sums -> bucket reduction; asynchronous on `stream`. */
int32_t gm_msm_run(gm_msm_plan* plan, const uint64_t* d_points_xy, const uint64_t* d_scalars, void* stream);

/* Results of the last run (device pointers owned by the plan, valid until the next run/destroy):
 *   bucket sums: 3 dense columns X,Y,Z of n_rows_local = (y_end-y_begin) << d_logsize elements each,
 *                row (y - y_begin) << d | digit  == bintree `last_step` output restricted to our windows;
 *   window points: 3*(d_logsize+1) columns of (y_end-y_begin) elements, column order
 *                [X0,Y0,Z0, X1,Y1,Z1, ...] == triangle `last_step` output (pippenger.rs:531-534);
 *   digits (u16) / counter (u32): [(y - y_begin) * N + x]                (pushforward.rs:351-361, 423);
 *   row_len (u32): bucket populations, n_rows_local entries. */
int32_t gm_msm_bucket_sums(const gm_msm_plan* plan, const uint64_t** d_x, const uint64_t** d_y,
                           const uint64_t** d_z, uint64_t* n_rows_local);
int32_t gm_msm_window_points(const gm_msm_plan* plan, const uint64_t** d_cols, uint64_t* n_cols,
                             uint64_t* col_len);
int32_t gm_msm_digits(const gm_msm_plan* plan, const uint16_t** d_digits, const uint32_t** d_counter,
                      const uint32_t** d_row_len);

/* Fr-side data of the pushforward argument derived from the bucketing of the last run (a12/a13):
 *   gm_msm_phase1_polys  PushForwardState::new   pushforward.rs:489-510  c, d ([y][x] as field elements), ac_c, ac_d
 *                        (negated access counts, 2^x_logsize and 2^d_logsize elements)
 *   gm_msm_second_phase  PushForwardState::second_phase  pushforward.rs:572-596  c_pull = eq_c[counter], d_pull = eq_d[digit]
 *                        for the point h_r = [r_y | r_d | r_c]  (the G1 commitments of these columns are SURVEY 8f-1) */
int32_t gm_msm_phase1_polys(const gm_msm_plan* plan, uint64_t* d_c, uint64_t* d_d, uint64_t* d_ac_c, uint64_t* d_ac_d,
                            void* stream);
int32_t gm_msm_second_phase(const gm_msm_plan* plan, const uint64_t* h_r, uint32_t y_logsize, uint64_t* d_c_pull,
                            uint64_t* d_d_pull, void* stream);

/* Bench instrumentation: HIP events on the launch stream around the stages of gm_msm_run.
 * mode 0 off, 1 = only the dominant kernel (level-0 bucket add), 2 = every stage.
 * gm_msm_profile_read returns ms per stage of the last run (7 floats: digits, histogram, chunk scan +
 * offsets, scatter, level-0 add, levels >= 1, bucket reduction; -1 where not recorded). */
int32_t gm_msm_profile(gm_msm_plan* plan, int32_t mode);
int32_t gm_msm_profile_read(gm_msm_plan* plan, float* h_ms, int32_t n);
/* how the last gm_msm_run was launched: *fused01 = 1 when bintree levels 0 and 1 ran as ONE kernel (k_add_level01; GM_MSM_FUSE01=0
 * switches that off): stage 4 of gm_msm_profile_read then covers both levels, stage 5 the levels from 2 on. */
int32_t gm_msm_run_info(const gm_msm_plan* p, int32_t* fused01);
/* cells of the row layout of every level of the LAST run (x_logsize + 1 counts: level l's input layout, pad cells included;
 * level l >= 1 adds h_cells[l] / 2 pairs); synchronises the stream.  Bench accounting of the level kernels' bytes and products. */
int32_t gm_msm_level_cells(const gm_msm_plan* plan, uint64_t* h_cells, uint32_t n, void* stream);

/* Final recombination acc = sum_w 2^(d*w) sum_{i>=1} 2^(i-1) P[i][w] on the host from the window points
 * of ALL windows (h_cols: 3*(d+1) columns x n_windows, Montgomery); writes affine (x,y), 8 x u64. */
int32_t gm_msm_combine_host(const uint64_t* h_cols, uint32_t d_logsize, uint32_t n_windows,
                            uint64_t* h_out_xy);

/* One-shot convenience: plan over all windows + run + D2H + combine; synchronises `stream`. */
int32_t gm_msm_te(const uint64_t* d_points_xy, const uint64_t* d_scalars, uint32_t x_logsize,
                  uint32_t d_logsize, uint32_t nbits, uint64_t* h_out_xy, void* stream);

/* ---------------------------------------------------------------- Fiat-Shamir seam
 * The transcript (merlin/STROBE hashing, sequential) stays with the caller.  The whole-protocol drivers below come in
 * two forms: `*_tr` takes the caller's live transcript as two callbacks (what the Rust shim passes: thin wrappers over
 * TProofTranscript2::write_scalars / challenge(128), cleanup/proof_transcript.rs:109-136, or gen-1
 * TranscriptReceiver::append_scalars / TranscriptSender::challenge_scalar, transcript.rs:70-101); the tape form takes
 * pre-drawn challenges and returns the messages (tests, benches, replay).
 *   write_scalars: n field elements in the reference's in-memory form (Montgomery, 4 x u64), in write order; may be NULL
 *   challenge:     `challenge_vec(n, bitsize)` (proof_transcript.rs:41-45; n = 1 is `challenge(bitsize)`): ONE squeeze of
 *                  n * ceil(bitsize / 8) bytes, every chunk `from_le_bytes_mod_order`; out = n canonical elements (4 x u64 LE each).
 *                  gen-2 draws (1, 128) everywhere except (4, 512) at the start of the pushforward argument and (1, 512) in the
 *                  opening; gen-1's `challenge_scalar` (64 bytes reduced mod p, transcript.rs:96-101) is requested as (1, 512)
 *   write_points:  n G1 points in the affine wire form (12 x u64 each), `write_points::<G1>` (proof_transcript.rs:59-69); only
 *                  the Knuckles opening uses it; may be NULL
 * A non-zero return from any of them aborts the prover with GM_ERR_STATE. */
typedef struct gm_transcript {
    void* ctx;
    int32_t (*write_scalars)(void* ctx, const uint64_t* elems, uint64_t n);
    int32_t (*challenge)(void* ctx, uint32_t n, uint32_t bitsize, uint64_t* out);
    int32_t (*write_points)(void* ctx, const uint64_t* aff_points, uint64_t n);
} gm_transcript;

/* A host-side ProofTranscript2 (cleanup/proof_transcript.rs:76-136) for callers without a Rust transcript: merlin
 * v1.0 (STROBE-128 / Keccak-f[1600]) with empty labels, scalars as 32-byte LE canonical elements, G1 points in ark-bls12-381's
 * 48-byte compressed encoding, challenges = from_le_bytes_mod_order.  gm_merlin_transcript fills a gm_transcript whose callbacks
 * feed it; gm_merlin_proof returns the proof bytes (every message written, concatenated).  A restatement of the published
 * formats pinned by public vectors -- not by bytes of the Rust binary.  No GPU work. */
typedef struct gm_merlin gm_merlin;
int32_t gm_merlin_create(const uint8_t* pparam, uint64_t len, gm_merlin** out);
int32_t gm_merlin_destroy(gm_merlin* t);
int32_t gm_merlin_transcript(gm_merlin* t, gm_transcript* out);
int32_t gm_merlin_proof(const gm_merlin* t, const uint8_t** bytes, uint64_t* len);
int32_t gm_merlin_append_message(gm_merlin* t, const uint8_t* label, uint64_t label_len, const uint8_t* msg, uint64_t len);
int32_t gm_merlin_challenge_bytes(gm_merlin* t, const uint8_t* label, uint64_t label_len, uint8_t* dest, uint64_t len);
int32_t gm_keccak_f1600(uint8_t* state200);

/* ---------------------------------------------------------------- the verifier
 * Pippenger::verify (cleanup/protocols/pippenger.rs:296-406) and everything under it -- the sumcheck, GKR, logup, pushforward,
 * multi-open and Knuckles verifiers -- on the host, as in the reference (no device work).  The transcript seam in verifier mode
 * (TProofTranscript2::read_scalars / challenge / read_points, proof_transcript.rs:33-62; PTMode::Verifier reads the proof and
 * feeds the sponge):
 *   read_scalars: next n field elements of the proof -> Montgomery, 4 x u64 each
 *   challenge:    as in gm_transcript
 *   read_points:  next n G1 points of the proof -> affine wire form (12 x u64 each); like TProofTranscript2::read_points
 *                 (deserialize_compressed with Validate::Yes) the callback hands over VALIDATED points: on the curve and in the
 *                 prime-order subgroup (the built-in reader does; the recorded form, gm_pippenger_verify, checks both itself)
 * A non-zero return from a read means the proof is too short or malformed: the verifier returns GM_ERR_VERIFY.
 *
 * gm_pippenger_verify_tr: claims = the point r_y (y_logsize elements) and the 3 (d_logsize + 1) evaluations of the MSM's dense
 *   output there (verify_pippenger, pippenger.rs:562-587, forms them from the claimed result); h_g0_aff = the first SRS element
 *   (KzgVerifyingKey::g0), h_k = the Knuckles generator k (Montgomery).  GM_OK: every check up to the deferred pairing passed and
 *   h_pair = (A, B), affine, 2 x 12 u64 (required: Pippenger::verify ends with verify_pair(ps_pair), pippenger.rs:403-405, so the
 *   proof is ACCEPTED only when gm_kzg_verify_pair(h_pair, h0, h1) returns GM_OK as well); GM_ERR_VERIFY: a check failed
 *   (gm_last_error() names it).  Shapes are bounded: x_logsize <= 30, d_logsize, y_logsize <= 16, y_size * d_logsize <= 256.
 * gm_pippenger_verify: the same over recorded messages and a challenge tape (the prover's outputs from gm_pippenger_prove);
 *   additionally insists that every message was read.
 * gm_kzg_verify_pair: KzgVerifyingKey::verify_pair (commitments/kzg.rs:61-67), e(A, h0) == e(B, h1) on BLS12-381; h0 = [1]_2 and
 *   h1 = [tau]_2 as G2 affine points: x.c0, x.c1, y.c0, y.c1, each Montgomery 6 x u64.  GM_OK = equal.
 * gm_merlin_create_verifier / gm_merlin_reader: ProofTranscript2::start_verifier over proof bytes (proof_transcript.rs:104-107). */
typedef struct gm_transcript_reader {
    void* ctx;
    int32_t (*read_scalars)(void* ctx, uint64_t n, uint64_t* out);
    int32_t (*challenge)(void* ctx, uint32_t n, uint32_t bitsize, uint64_t* out);
    int32_t (*read_points)(void* ctx, uint64_t n, uint64_t* out_aff);
    /* non-zero: read_points has already checked prime-order subgroup membership (ark's deserialize_compressed with Validate::Yes,
     * proof_transcript.rs read_points; gm_merlin_reader sets it).  Zero -- the default of a zero-initialised struct -- makes the
     * verifier run the check itself on every point read (~70 us each). */
    uint32_t points_validated;
    uint32_t reserved;
} gm_transcript_reader;
int32_t gm_pippenger_verify_tr(uint32_t x_logsize, uint32_t d_logsize, uint32_t y_size, uint32_t y_logsize,
                               uint32_t commitment_log_multiplicity, const uint64_t* h_claim_point, const uint64_t* h_claim_evs,
                               const uint64_t* h_g0_aff, const uint64_t* h_k, const gm_transcript_reader* tr, uint64_t* h_pair);
int32_t gm_pippenger_verify(uint32_t x_logsize, uint32_t d_logsize, uint32_t y_size, uint32_t y_logsize,
                            uint32_t commitment_log_multiplicity, const uint64_t* h_claim_point, const uint64_t* h_claim_evs,
                            const uint64_t* h_g0_aff, const uint64_t* h_k, const uint64_t* h_scalars, uint64_t n_scalars,
                            const uint64_t* h_points_aff, uint64_t n_points, const uint64_t* h_tape, uint64_t n_tape,
                            uint64_t* h_pair, uint64_t* tape_used);
/* gen-1: the verifier side of gkr_msm_prove's GKR -- BintreeVerifier / SumcheckPolyMapVerifier / SplitVerifier ::round
 * (protocol/bintree.rs:313-395, protocol/sumcheck.rs:595-657, protocol/split.rs:99-115) -- over the stream gm_gkr_msm_prove
 * emits (h_msgs) and its challenges; returns the final EvalClaim about the base layer (point of lp + lb elements, 3 evaluations:
 * bit, px, py), which the caller checks against its commitments (the reference's gkr_msm_prove stops there too). */
int32_t gm_gkr_msm_verify(uint32_t log_num_points, uint32_t log_num_scalar_bits, const uint64_t* h_msgs, uint64_t n_msgs,
                          const uint64_t* h_tape, uint64_t n_tape, uint64_t* h_final_point, uint32_t* n_final_point,
                          uint64_t* h_final_evs, uint64_t* tape_used, uint64_t* rounds);
int32_t gm_gkr_msm_verify_tr(uint32_t log_num_points, uint32_t log_num_scalar_bits, const gm_transcript_reader* tr,
                             uint64_t* h_final_point, uint32_t* n_final_point, uint64_t* h_final_evs, uint64_t* rounds);
int32_t gm_kzg_verify_pair(const uint64_t* h_pair, const uint64_t* h_h0, const uint64_t* h_h1);
int32_t gm_kzg_mock_vk(const uint64_t* h_tau, uint64_t* h_h0, uint64_t* h_h1);   /* [1]_2, [tau]_2 of KzgProvingKey::mock_setup */
int32_t gm_pairing(const uint64_t* h_p_aff, const uint64_t* h_q_aff, uint64_t* h_gt);
int32_t gm_merlin_create_verifier(const uint8_t* pparam, uint64_t len, const uint8_t* proof, uint64_t proof_len, gm_merlin** out);
int32_t gm_merlin_reader(gm_merlin* t, gm_transcript_reader* out);
int32_t gm_merlin_unread(const gm_merlin* t, uint64_t* n);

/* ---------------------------------------------------------------- multi-GPU seam
 * One process per GPU.  The path shards by MSM window: rank g owns windows [g*y_size/G, (g+1)*y_size/G), i.e. the bucket
 * rows (y << d_logsize | digit) of its windows -- MSM, witness build, round sums and folds of the bintree GKR are all local to
 * those rows.  The exchange steps are tiny and go through ONE collective on host buffers, either the library's own RCCL
 * backing (gm_comm_rccl_*, below) or a caller-provided one (the CPU tests use torch.distributed over gloo; a Rust shim can
 * back it with anything):
 *   - per sumcheck round: all-gather of the 2-3 partial round sums (<= 96 bytes per rank), added mod p by every rank;
 *   - once per proof: all-gather of the bucket sums (3 * 2^(y_logsize + d_logsize) field elements in total), after which the
 *     bucket-reduction (triangle) GKR runs replicated, and of a dense layer's slices once they are down to 2^8 elements per
 *     column and rank (GM_SC_SHARD_GATHER_LOG): the remaining 8 + log2(G) rounds of that layer run replicated, unsharded.
 * all_gather: `h_buf` holds world * bytes_per_rank bytes; the caller has filled slot `rank`; on return every slot is filled.
 * Every rank must run the same sequence of calls (the provers are deterministic given the same challenges). */
typedef struct gm_pull {
    uint32_t peer, reserved;
    uint64_t src_offset, bytes;
    void* d_dst;
} gm_pull;
typedef struct gm_comm {
    void* ctx;
    uint32_t rank, world;   /* world: a power of two dividing y_size */
    int32_t (*all_gather)(void* ctx, void* h_buf, uint64_t bytes_per_rank);
    /* optional (NULL: the host form above serves every exchange): the same collective on DEVICE buffers, asynchronous on `stream`
     * -- d_recv receives world * bytes_per_rank bytes, rank-major.  When present the per-round sums of the sharded provers never
     * leave the device before they are added up: round kernel -> device slot -> this all-gather -> a one-wave sum -> ONE report
     * to the host (gm_comm_rccl_as_comm sets it: ncclAllGather). */
    int32_t (*all_gather_dev)(void* ctx, const void* d_send, void* d_recv, uint64_t bytes_per_rank, void* stream);
    /* optional (NULL: bulk redistributions are staged through `all_gather` on host buffers): device to device, collective -- every
     * rank exposes `src_bytes` bytes at d_src and pulls n pieces, piece k = bytes [src_offset, src_offset + bytes) of rank `peer`'s
     * d_src into d_dst (a rank may pull from itself).  Returns when this rank's pieces have arrived AND every rank has finished
     * reading (d_src may be reused).  Return value 100 = the device path is not available (some rank could not export or open a
     * mapping): every rank gets the same answer from the same call, nothing was copied, the caller stages the redistribution through
     * `all_gather` instead.  The sharded pushforward argument re-spreads the halves of its logup tree with it
     * (gm_comm_shm_as_comm sets it: HIP IPC handles exchanged through the shared memory, hipMemcpyAsync between the devices). */
    int32_t (*pull_dev)(void* ctx, const void* d_src, uint64_t src_bytes, uint32_t n, const struct gm_pull* pieces, void* stream);
} gm_comm;
/* A rank's view of the KZG proving key (kzg_pk.ptau_1, kzg.rs:123-132) in a sharded proof: the segments of the key that are
 * resident on this rank's device, as (device pointer, index of the first point, number of points), affine wire form.  With
 * commitment_log_multiplicity = clm the key has 2^(x_logsize + clm + 1) - 1 points (pippenger.rs:475-480: 51 GB at x_logsize 24,
 * clm 4) and no rank needs it whole; gm_pippenger_sharded_key_ranges lists what a rank reads.  A step whose range is not resident
 * fails with GM_ERR_INVALID naming the range.  One segment covering the whole key is the small-shape / test form. */
typedef struct gm_key_view {
    uint32_t n_segments, reserved;
    const uint64_t* const* d_segment;
    const uint64_t* first;
    const uint64_t* count;
} gm_key_view;

/* Diagnosis of a sharded run: where the calling thread's time inside the communicator went since the last reset.  out8 = {small
 * host all-gathers (<= 4 KiB: round sums, agreement words, group elements -- mostly waiting for the slowest rank): microseconds,
 * count; larger host all-gathers: microseconds, count, bytes per rank; pull_dev: microseconds, count, bytes pulled}. */
int32_t gm_shard_clock(int32_t reset, double* out8);
/* host-only self-test of a gm_comm (no GPU): sums the field elements h_vals[0..n) of all ranks in place (Montgomery) */
int32_t gm_comm_sum_fr(const gm_comm* comm, uint64_t* h_vals, uint32_t n);

/* Native backing of the seam: RCCL over xGMI, one communicator per process (= per GPU).  RCCL is bound at run time (dlopen;
 * a copy already in the process, e.g. PyTorch's, is reused), so single-GPU callers never load it.
 *   gm_comm_rccl_unique_id       ncclGetUniqueId on one rank; the caller carries the 128 bytes to the others
 *   gm_comm_rccl_create          ncclCommInitRank on the current device (collective: every rank calls it)
 *   gm_comm_rccl_as_comm         a gm_comm for gm_pip_witness_create_sharded whose all_gather is ncclAllGather on a device
 *                                staging buffer on the communicator's stream (payloads <= 64 KiB go through pinned memory)
 *   gm_comm_rccl_all_gather_dev  device buffers, asynchronous on `stream`: the window points of a window-sharded MSM
 *                                (gm_msm_window_points: 3 (d_logsize + 1) * windows_per_rank elements per rank)
 *   gm_comm_rccl_broadcast_dev   operand replication: points / scalars from `root` to every rank (the one link-bound step)
 * Nothing is REDUCED by RCCL: field and curve additions are not RCCL operations, every rank adds the gathered parts itself. */
#define GM_RCCL_UNIQUE_ID_BYTES 128
typedef struct gm_rccl gm_rccl;
int32_t gm_comm_rccl_unique_id(uint8_t* out_id128);
int32_t gm_comm_rccl_create(const uint8_t* id128, uint32_t rank, uint32_t world, gm_rccl** out, void* stream);
int32_t gm_comm_rccl_destroy(gm_rccl* r);
int32_t gm_comm_rccl_as_comm(gm_rccl* r, gm_comm* out);
int32_t gm_comm_rccl_all_gather_dev(gm_rccl* r, const void* d_send, void* d_recv, uint64_t bytes_per_rank, void* stream);
int32_t gm_comm_rccl_broadcast_dev(gm_rccl* r, void* d_buf, uint64_t bytes, uint32_t root, void* stream);
int32_t gm_comm_rccl_stats(const gm_rccl* r, uint64_t* host_all_gathers, uint64_t* bytes_per_rank_total);

/* One-node backing of the seam over POSIX shared memory (csrc/shm_comm.hip): the per-round sums of a sharded proof are wanted on the
 * HOST (they go into the caller's transcript), so the ranks of one node exchange them between their host threads -- a cache-line
 * transfer, well under a microsecond, where a device collective costs a kernel launch and a ring per round.  With this gm_comm
 * (host all_gather, no all_gather_dev) the device side of a sharded round is the unsharded one: pre-enqueued folds and the
 * persistent stage kernel included.  RCCL keeps the transfers that carry volume (operand replication, window points).
 *   gm_comm_shm_create   collective: every rank calls it with the same name ("/gm-<job id>", unique per job) and world; rank 0
 *                        creates the object (an existing name is an error), all return once every rank has mapped it, the name
 *                        is removed again at that point (nothing is left behind in /dev/shm if a rank dies later)
 *   gm_comm_shm_as_comm  the gm_comm for gm_pip_witness_create_sharded; payloads of any size (256 KiB chunks)
 * Every wait is bounded by gm_set_wait_timeout_ms. */
typedef struct gm_shm gm_shm;
int32_t gm_comm_shm_create(const char* name, uint32_t rank, uint32_t world, gm_shm** out);
int32_t gm_comm_shm_destroy(gm_shm* c);
int32_t gm_comm_shm_as_comm(gm_shm* c, gm_comm* out);
int32_t gm_comm_shm_stats(const gm_shm* c, uint64_t* all_gathers, uint64_t* bytes_per_rank_total);
/* pull_dev's cache of opened peer allocations (HIP IPC): least recently used first out, bounded by GM_SHM_MAX_OPENED (default 1024); the mappings belong to the process and outlive a communicator
 * BETWEEN calls -- a mapping the running call resolved an address into is never closed under it.  Counters for tests / diagnosis. */
int32_t gm_comm_shm_ipc_stats(const gm_shm* c, uint64_t* opens, uint64_t* closes, uint64_t* held);

/* ---------------------------------------------------------------- "prove image part" (a10, a11)
 * Host-side driver over the kernels, mirroring PippengerWG::new (pippenger.rs:37-70, without the BLS12-381 G1
 * commitments: SURVEY 8f-1) and Pippenger::prove's "prove image part" span (pippenger.rs:138-141):
 *   gm_pip_witness_create    image (gm_vv_from_msm) -> GlueSplit::witness (splits.rs:172-176) -> PippengerEndingWG::new
 *                            (pippenger_ending.rs:32-95): bintree witness, last_step, two HI splits, triangle witness,
 *                            and the dense output of pippenger.rs:531-534
 *   gm_pip_prove_image_part  PippengerBucketed::prove (pippenger_ending.rs:142-149) + GlueSplit::prove (splits.rs:185-197).
 * The Fiat-Shamir transcript stays with the caller: h_tape holds the challenges in the order the protocol draws
 * them (canonical 4 x u64, values < 2^128 as `challenge(128)` yields); prover messages come back in write order. */
typedef struct gm_pip_witness gm_pip_witness;
int32_t gm_pip_witness_create(const gm_msm_plan* plan, const uint64_t* d_points_xy, uint32_t y_logsize,
                              gm_pip_witness** out, void* stream);
/* Sharded form: `plan` covers this rank's windows only (gm_msm_plan_create(.., y_begin, y_end)), y_size = the global window
 * count; requires y_size = 2^y_logsize and comm->world | y_size.  The prove calls below then run the sharded protocol on
 * every rank (same messages and final claims on all ranks, bit-identical to the unsharded run).  `comm` must outlive w. */
int32_t gm_pip_witness_create_sharded(const gm_msm_plan* plan, const uint64_t* d_points_xy, uint32_t y_logsize,
                                      const gm_comm* comm, gm_pip_witness** out, void* stream);
int32_t gm_pip_witness_destroy(gm_pip_witness* w);
/* "claim computation" of run_pippenger (pippenger.rs:531-541): h_evs[c] = evaluate_poly(dense_output[c], h_point), point of
 * y_logsize elements, 3 (d_logsize + 1) evaluations out -- the ClaimsBefore of Pippenger::prove / verify */
int32_t gm_pip_witness_claims(const gm_pip_witness* w, const uint64_t* h_point, uint64_t* h_evs, uint32_t* n_evs);
int32_t gm_pip_witness_outputs(const gm_pip_witness* w, const uint64_t** d_output_cols, uint32_t* n_output_cols,
                               uint64_t* output_len, const uint64_t** d_bucket_sum_cols);
uint64_t gm_pip_witness_bytes(const gm_pip_witness* w);
int32_t gm_pip_prove_image_part(const gm_pip_witness* w, const uint64_t* h_claim_point, const uint64_t* h_claim_evs,
                                const uint64_t* h_tape, uint64_t n_tape, uint64_t* h_msgs, uint64_t msgs_cap,
                                uint64_t* n_msgs, uint64_t* h_final_point, uint32_t* n_final_point,
                                uint64_t* h_final_evs, uint64_t* tape_used, uint64_t* rounds);

int32_t gm_pip_prove_image_part_tr(const gm_pip_witness* w, const uint64_t* h_claim_point, const uint64_t* h_claim_evs,
                                   const gm_transcript* tr, uint64_t* h_final_point, uint32_t* n_final_point,
                                   uint64_t* h_final_evs, uint64_t* n_challenges, uint64_t* rounds);

/* ---------------------------------------------------------------- the two GKR circuits on their own (a10, a11)
 * TriangleAddWG / TriangleAdd (gkrs/triangle_add.rs:160-250) and VecVecBintreeAddWG / VecVecBintreeAdd (gkrs/bintree_add.rs:85-126):
 * the witness generators and SimpleGKR provers the Pippenger prover is built from, callable directly as the reference's
 * own tests and benches call them (triangle_add.rs:277-393, bintree_add.rs:401-505).
 *   gm_triangle_witness_create  TriangleAddWG::new(inputs, num_vars, HI(split_hi)); d_cols = 12 device columns of 2^num_vars
 *                               (the output of two HI(split_hi) split-maps of the X, Y, Z bucket columns); borrowed, not copied
 *   gm_bintree_witness_create   VecVecBintreeAddWG::new_common(VecVecMAP(inputs), row_logsize, num_adds, do_bitcheck) with
 *                               row_logsize = inputs' row_logsize + 1 (inputs are the LO(0) split of the point columns)
 *   gm_gkr_witness_output       builder::witness::last_step of the last advice, as dense device columns (a VecVec result is
 *                               densified): 3 (num_layers + 3) columns for the triangle, 3 for the bintree
 *   gm_gkr_prove(_tr)           TriangleAdd::prove / VecVecBintreeAdd::prove: claims on the output -> claims on the inputs */
typedef struct gm_gkr_witness gm_gkr_witness;
int32_t gm_triangle_witness_create(const uint64_t* const* d_cols, uint32_t num_vars, uint32_t split_hi, gm_gkr_witness** out,
                                   void* stream);
int32_t gm_bintree_witness_create(const gm_vv* inputs, uint32_t num_adds, int32_t do_bitcheck, gm_gkr_witness** out, void* stream);
int32_t gm_gkr_witness_destroy(gm_gkr_witness* w);
int32_t gm_gkr_witness_output(const gm_gkr_witness* w, const uint64_t** d_cols, uint32_t cols_cap, uint32_t* n_cols,
                              uint32_t* num_vars);
int32_t gm_gkr_prove(const gm_gkr_witness* w, const uint64_t* h_claim_point, const uint64_t* h_claim_evs, const uint64_t* h_tape,
                     uint64_t n_tape, uint64_t* h_msgs, uint64_t msgs_cap, uint64_t* n_msgs, uint64_t* h_final_point,
                     uint32_t* n_final_point, uint64_t* h_final_evs, uint32_t* n_final_evs, uint64_t* tape_used, uint64_t* rounds);
int32_t gm_gkr_prove_tr(const gm_gkr_witness* w, const uint64_t* h_claim_point, const uint64_t* h_claim_evs,
                        const gm_transcript* tr, uint64_t* h_final_point, uint32_t* n_final_point, uint64_t* h_final_evs,
                        uint32_t* n_final_evs, uint64_t* n_challenges, uint64_t* rounds);

/* ---------------------------------------------------------------- "prove pushforward" (a7, a10, a12, a13)
 * PushforwardProtocol::prove (pushforward/pushforward.rs:640-846) with LogupMainphaseProtocol (pushforward/logup_mainphase.rs:83-208)
 * on the Fr columns of the plan's last gm_msm_run: c / d / ac_c / ac_d (PushForwardState::new :489-510), c_pull / d_pull
 * (second_phase :572-596, point = the image part's final point [r_y | r_d | r_c]), p_0 / p_1 = the points' coordinates.
 *   h_claim_point / h_claim_evs: the final claims of gm_pip_prove_image_part (y_logsize + d_logsize + x_logsize coordinates, 3 evs)
 *   challenges: 4 of 512 bits (canonical field elements on the tape), then `challenge(128)` values, in draw order
 *   outputs = PushforwardFinalClaims (:626-631): gamma; claims_about_matrix: point (x_logsize + y_logsize coordinates) and
 *   [p_folded, c_pull, d_pull, c, d] evaluations; claims_ac_c: point (x_logsize) + [ac_c, table_c] evaluations; claims_ac_d:
 *   point (d_logsize) + [ac_d, table_d] evaluations.  (The G1 commitments written to the transcript around this call,
 *   pippenger.rs:126-133, 152-153, are gm_msm_g1_outer / gm_g1_msm / gm_g1_msm_nonaff results.) */
int32_t gm_pushforward_prove(const gm_msm_plan* plan, const uint64_t* d_points_xy, uint32_t y_logsize,
                             const uint64_t* h_claim_point, const uint64_t* h_claim_evs, const uint64_t* h_tape, uint64_t n_tape,
                             uint64_t* h_msgs, uint64_t msgs_cap, uint64_t* n_msgs, uint64_t* h_gamma, uint64_t* h_matrix_point,
                             uint64_t* h_matrix_evs, uint64_t* h_ac_c_point, uint64_t* h_ac_c_evs, uint64_t* h_ac_d_point,
                             uint64_t* h_ac_d_evs, uint64_t* tape_used, uint64_t* rounds, void* stream);
/* The argument with the matrix sharded by windows: `plan` covers this rank's windows only (gm_msm_plan_create(..,
 * y_begin, y_end) after its gm_msm_run), y_size = the global window count = 2^y_logsize, comm->world | y_size.  Every array of the
 * argument is indexed (window, point): a rank holds its windows' slice of each, the sumchecks run on the slices (round sums through
 * `comm`, the last log2(world) rounds replicated), the access counts are summed over the ranks, and the halves of every level of the
 * logup tree are re-spread over the ranks.  Every rank runs the same transcript and obtains the unsharded argument's messages and
 * claims, bit for bit.  `comm` must outlive the call. */
int32_t gm_pushforward_prove_sharded(const gm_msm_plan* plan, const uint64_t* d_points_xy, uint32_t y_logsize,
                                     const struct gm_comm* comm, const uint64_t* h_claim_point, const uint64_t* h_claim_evs,
                                     const uint64_t* h_tape, uint64_t n_tape, uint64_t* h_msgs, uint64_t msgs_cap, uint64_t* n_msgs,
                                     uint64_t* h_gamma, uint64_t* h_matrix_point, uint64_t* h_matrix_evs, uint64_t* h_ac_c_point,
                                     uint64_t* h_ac_c_evs, uint64_t* h_ac_d_point, uint64_t* h_ac_d_evs, uint64_t* tape_used,
                                     uint64_t* rounds, void* stream);
int32_t gm_pushforward_prove_tr(const gm_msm_plan* plan, const uint64_t* d_points_xy, uint32_t y_logsize,
                                const uint64_t* h_claim_point, const uint64_t* h_claim_evs, const gm_transcript* tr,
                                uint64_t* h_gamma, uint64_t* h_matrix_point, uint64_t* h_matrix_evs, uint64_t* h_ac_c_point,
                                uint64_t* h_ac_c_evs, uint64_t* h_ac_d_point, uint64_t* h_ac_d_evs, uint64_t* n_challenges,
                                uint64_t* rounds, void* stream);

/* MultiOpenReduction::prove (cleanup/protocols/multiopen_reduction.rs:65-93; the first step of the "open" span, pippenger.rs:222-258):
 * nargs (<= 8) device columns of 2^nvars elements with one claim each -- h_points: nargs x nvars coordinates, h_evs: nargs
 * evaluations -- are reduced to nargs claims at one common point (h_out_point: nvars coordinates, h_out_evs: nargs evaluations). */
int32_t gm_multiopen_prove(uint32_t nvars, uint32_t nargs, const uint64_t* const* d_polys, const uint64_t* h_points,
                           const uint64_t* h_evs, const uint64_t* h_tape, uint64_t n_tape, uint64_t* h_msgs, uint64_t msgs_cap,
                           uint64_t* n_msgs, uint64_t* h_out_point, uint64_t* h_out_evs, uint64_t* tape_used, uint64_t* rounds,
                           void* stream);
int32_t gm_multiopen_prove_tr(uint32_t nvars, uint32_t nargs, const uint64_t* const* d_polys, const uint64_t* h_points,
                              const uint64_t* h_evs, const gm_transcript* tr, uint64_t* h_out_point, uint64_t* h_out_evs,
                              uint64_t* n_challenges, uint64_t* rounds, void* stream);

/* div_by_linear + ev (commitments/kzg.rs:73-81, 142-150) on a device polynomial (coefficients lowest first): h_ev = poly(pt) = the
 * remainder; d_quotient (len - 1 coefficients; may be NULL) = poly / (X - pt).  KzgProvingKey::open = this + gm_g1_msm. */
int32_t gm_kzg_div_by_linear(const uint64_t* d_poly, uint64_t len, const uint64_t* h_pt, uint64_t* d_quotient, uint64_t* h_ev,
                             void* stream);

/* Knuckles opening: KnucklesOpeningProtocol::prove (cleanup/protocols/opening.rs:39-98) = compute_t
 * (commitments/knuckles.rs:111-154), three KZG commitments / openings (kzg.rs:73-81, 123-132) and the deferred pairing pair.
 *   gm_knuckles_setup   the `inverses` table of KnucklesProvingKey::new (knuckles.rs:64-82): 2^(num_vars+1) - 1 elements
 *   gm_knuckles_open    d_basis_aff: kzg_pk.ptau_1 (>= 2^(num_vars+1) - 1 affine points); d_poly: poly_len <= 2^num_vars
 *                       coefficients; h_point: num_vars coordinates; h_claimed_ev; h_commitment_aff: the claim's commitment.
 *                       Transcript order: point T, challenge x, scalars [T(x), P(x)], challenge lambda, point, scalar T(kx),
 *                       point, challenge fin (3 challenges on the tape).
 *                       h_proof (48 x u64): t_comm (12) | t_x (4) | p_x (4) | p_lt_x_proof (12) | t_kx (4) | t_kx_proof (12);
 *                       h_pair (24 x u64): the deferred pair (A, B) with <A, H0> = <B, H1>. */
int32_t gm_knuckles_setup(const uint64_t* h_k, uint32_t num_vars, uint64_t* d_inverses, void* stream);
int32_t gm_knuckles_open(const uint64_t* d_basis_aff, const uint64_t* d_inverses, const uint64_t* h_k, uint32_t num_vars,
                         const uint64_t* d_poly, uint64_t poly_len, const uint64_t* h_point, const uint64_t* h_claimed_ev,
                         const uint64_t* h_commitment_aff, const uint64_t* h_tape, uint64_t n_tape, uint64_t* h_proof,
                         uint64_t* h_pair, void* stream);
int32_t gm_knuckles_open_tr(const uint64_t* d_basis_aff, const uint64_t* d_inverses, const uint64_t* h_k, uint32_t num_vars,
                            const uint64_t* d_poly, uint64_t poly_len, const uint64_t* h_point, const uint64_t* h_claimed_ev,
                            const uint64_t* h_commitment_aff, const gm_transcript* tr, uint64_t* h_proof, uint64_t* h_pair,
                            void* stream);
/* The opening with polynomial, key and inverses distributed over the ranks of a gm_comm (config E: 2^28 coefficients, a
 * 51 GB key, a 16 GiB table t).  Collective; rank r of G holds, with N = 2^num_vars and S = 2N / G:
 *   d_poly_slice      coefficients [r N / G, (r + 1) N / G) of the zero-padded polynomial
 *   d_inverses_slice  entries [r S, (r + 1) S) of the inverses table that exist (2N - 1 in all): gm_knuckles_setup_range
 *   key               must hold points [r S, (r + 1) S) that exist (and point 0 on rank 0)
 * compute_t's passes (knuckles.rs:131-146) read a halo from the lower neighbour (gm_comm::pull_dev, or its host fall-back),
 * evaluations and quotients (kzg.rs:73-81, 142-150) chain one value per rank, every commitment (kzg.rs:123-132) is the rank's MSM
 * over its key range combined by gm_g1_combine_parts.  Same proof and pair on every rank as gm_knuckles_open over the whole. */
int32_t gm_knuckles_setup_range(const uint64_t* h_k, uint32_t num_vars, uint64_t first, uint64_t count, uint64_t* d_inverses,
                                void* stream);
int32_t gm_knuckles_open_sharded(const gm_comm* comm, const gm_key_view* key, const uint64_t* d_inverses_slice, const uint64_t* h_k,
                                 uint32_t num_vars, const uint64_t* d_poly_slice, const uint64_t* h_point, const uint64_t* h_claimed_ev,
                                 const uint64_t* h_commitment_aff, const uint64_t* h_tape, uint64_t n_tape, uint64_t* h_proof,
                                 uint64_t* h_pair, void* stream);
int32_t gm_knuckles_open_sharded_tr(const gm_comm* comm, const gm_key_view* key, const uint64_t* d_inverses_slice, const uint64_t* h_k,
                                    uint32_t num_vars, const uint64_t* d_poly_slice, const uint64_t* h_point, const uint64_t* h_claimed_ev,
                                    const uint64_t* h_commitment_aff, const gm_transcript* tr, uint64_t* h_proof, uint64_t* h_pair,
                                    void* stream);

/* ---------------------------------------------------------------- the whole gen-2 prover
 * PippengerWG::new (cleanup/protocols/pippenger.rs:37-70) and Pippenger::prove (pippenger.rs:118-290) behind two calls; pure
 * orchestration of the entry points above (phase-1 commitments, image part, second phase + its commitments, pushforward
 * argument, MultiOpenReduction, Knuckles opening) with the scalar / G1 glue of the "open" span on the host.
 *   gm_pippenger_wg_create  after gm_msm_run(plan, ...): witness of the image part, outer buckets, commitments c[], d[] (one per
 *                           2^clm windows), p_0, p_1, ac_c, ac_d.  d_kzg_basis_aff: >= 2^(x_logsize + clm + 1) - 1 affine points.
 *   gm_pippenger_wg_witness the gm_pip_witness inside (gm_pip_witness_outputs gives the dense output the claims are about)
 *   gm_pippenger_prove      claims = (r_y, the 3 (d_logsize + 1) evaluations of the dense output at r_y).  Scalars written go to
 *                           h_msgs, G1 points (12 x u64 affine each) to h_points, both in write order; h_pair = the deferred
 *                           pairing pair (A, B), <A, H0> = <B, H1>.  d_knuckles_inverses / h_k: gm_knuckles_setup for
 *                           num_vars = x_logsize + clm.  The `_tr` form drives the caller's live transcript instead. */
typedef struct gm_pippenger_wg gm_pippenger_wg;
int32_t gm_pippenger_wg_create(const gm_msm_plan* plan, const uint64_t* d_points_xy, uint32_t y_logsize,
                               uint32_t commitment_log_multiplicity, const uint64_t* d_kzg_basis_aff, gm_pippenger_wg** out,
                               void* stream);
/* The same for one rank of a window-sharded proof (BASELINE.json configs[4]).  Collective: every rank calls it with the
 * plan of ITS windows, the same shape and clm, a gm_key_view of the key ranges resident on its device and the communicator (which
 * must outlive the handle).  PippengerWG::new's G1 work is linear in the committed columns (pushforward.rs:417-420, 504-524;
 * kzg.rs:123-132): a rank commits what its windows / its index range contribute, one group element per commitment crosses the
 * communicator, and the handle carries the COMBINED commitments -- equal on every rank to gm_pippenger_wg_create's over the whole
 * key.  gm_pippenger_prove(_tr) on such a handle runs Pippenger::prove sharded (image part, pushforward argument, the opening
 * witnesses and MultiOpenReduction on slices, gm_knuckles_open_sharded); there d_knuckles_inverses is the rank's
 * gm_knuckles_setup_range slice.  Every rank obtains the proof and pairing pair of the unsharded prover.
 * gm_pippenger_sharded_key_ranges: the <= 4 ranges of kzg_basis() a rank reads ([0] the key slices of its windows' outer buckets,
 * which must lie in ONE segment of the view; [1] its share of p_0 / p_1 / ac_c; [2] of ac_d; [3] its range of the opening). */
int32_t gm_pippenger_wg_create_sharded(const gm_msm_plan* plan, const uint64_t* d_points_xy, uint32_t y_logsize,
                                       uint32_t commitment_log_multiplicity, const gm_key_view* key, const gm_comm* comm,
                                       gm_pippenger_wg** out, void* stream);
int32_t gm_pippenger_sharded_key_ranges(uint32_t x_logsize, uint32_t d_logsize, uint32_t y_logsize,
                                        uint32_t commitment_log_multiplicity, uint32_t rank, uint32_t world, uint64_t* first4,
                                        uint64_t* count4);
/* The calling thread's last gm_pippenger_prove(_tr) by the reference's tracing spans (pippenger.rs:121-159), milliseconds of host
 * wall time: out8 = {prove image part, phase-2 commitments, prove pushforward, open: witnesses + commitment combinations, open:
 * MultiOpenReduction, open: Knuckles, 0, 0}. */
int32_t gm_pippenger_last_spans(double* out8);
int32_t gm_pippenger_wg_destroy(gm_pippenger_wg* wg);
int32_t gm_pippenger_wg_witness(const gm_pippenger_wg* wg, const gm_pip_witness** w);
int32_t gm_pippenger_prove(const gm_pippenger_wg* wg, const uint64_t* h_claim_point, const uint64_t* h_claim_evs,
                           const uint64_t* d_knuckles_inverses, const uint64_t* h_k, const uint64_t* h_tape, uint64_t n_tape,
                           uint64_t* h_msgs, uint64_t msgs_cap, uint64_t* n_msgs, uint64_t* h_points, uint64_t points_cap,
                           uint64_t* n_points, uint64_t* h_pair, uint64_t* tape_used, uint64_t* rounds);
int32_t gm_pippenger_prove_tr(const gm_pippenger_wg* wg, const uint64_t* h_claim_point, const uint64_t* h_claim_evs,
                              const uint64_t* d_knuckles_inverses, const uint64_t* h_k, const gm_transcript* tr, uint64_t* h_pair,
                              uint64_t* n_challenges, uint64_t* rounds);

/* ---------------------------------------------------------------- gen-1 prover (a6, a16)
 * gkr_msm_prove (src/gkr_msm_simple.rs:86-338) without the BLS12-381 G1 column commitments: base polys
 * (bit, px, py) over index point*2^lb + bit, BintreeProtocol::witness (protocol/bintree.rs:168-184) over the layer list of
 * gkr_msm_simple.rs:248-269, output claims, and the BintreeProver::round loop (bintree.rs:213-288) with
 * SumcheckPolyMapProver / SplitProver (protocol/sumcheck.rs:185-257, protocol/split.rs:66-82).
 *   d_scalar_bits: 2^lp * 2^lb bytes, the flattened Vec<Vec<bool>> (1 = true);  h_tape: challenges in draw order, canonical
 *   field elements (gen-1 challenges are 64 bytes reduced mod p, transcript.rs:96-101);  h_msgs: everything the prover appends
 *   to the transcript, in order (output polys, round polynomials as full coefficient vectors, final evaluations);
 *   h_output: the three output polys (3 * 2^lb);  final EvalClaim: point (lp + lb elements) and 3 evaluations (bit, px, py). */
int32_t gm_gkr_msm_prove(const uint64_t* d_points_xy, const uint8_t* d_scalar_bits, uint32_t log_num_points,
                         uint32_t log_num_scalar_bits, const uint64_t* h_tape, uint64_t n_tape, uint64_t* h_msgs,
                         uint64_t msgs_cap, uint64_t* n_msgs, uint64_t* h_output, uint64_t* h_final_point,
                         uint32_t* n_final_point, uint64_t* h_final_evs, uint64_t* tape_used, uint64_t* rounds,
                         double* witness_ms, void* stream);

int32_t gm_gkr_msm_prove_tr(const uint64_t* d_points_xy, const uint8_t* d_scalar_bits, uint32_t log_num_points,
                            uint32_t log_num_scalar_bits, const gm_transcript* tr, uint64_t* h_output,
                            uint64_t* h_final_point, uint32_t* n_final_point, uint64_t* h_final_evs, uint64_t* n_challenges,
                            uint64_t* rounds, void* stream);

/* ---------------------------------------------------------------- gen-1 fragmented polynomials, any shape (a3, a4, a5)
 * FragmentedPoly{data, consts, shape} (src/polynomial/fragmented.rs:383-388) with `data` / `consts` in device memory and the
 * shape -- a short list of fragments -- on the host, exactly the reference's Fragment{mem_idx, len, content, start}
 * (fragmented.rs:36-41).  Fragments must tile the index range in order; a Data fragment's mem_idx is the number of data
 * cells before it, a Consts fragment's mem_idx indexes `consts` (the invariants Shape::finalize asserts, :168-183).
 *   gm_frag_shape_full_split  Shape::full_split + prune_consts   fragmented.rs:285-364  -> fragments of both halves, the
 *                             permutation of the constants (new consts[i] = old consts[perm[i]]) and the new data length
 *   gm_frag_split             FragmentedPoly::split              fragmented.rs:676-732  (even / odd halves; outputs sized by
 *                             gm_frag_shape_full_split: data_len cells and n_perm constants each)
 *   gm_frag_bind              FragmentedPoly::bind               fragmented.rs:736-746  l + t (r - l) on data and consts
 *   gm_frag_to_dense          FragmentedPoly::into_vec           fragmented.rs:831-846
 *   gm_segment_split          compute_segment_split              src/copoly.rs:139-148  (host only)
 *   gm_frag_eq_materialize    EqPoly::materialize_eq_with_shape  src/copoly.rs:492-567: CopolyData{values (device, one per
 *                             data cell), sums (host, one per constant)} of multiplier * eq(point, .), point[0] = MSB */
#define GM_FRAG_DATA 0
#define GM_FRAG_CONSTS 1
typedef struct gm_fragment {
    uint64_t mem_idx, len, start;
    uint32_t content; /* GM_FRAG_DATA / GM_FRAG_CONSTS */
    uint32_t reserved;
} gm_fragment;
int32_t gm_frag_shape_full_split(const gm_fragment* frags, uint32_t n_frags, uint64_t num_consts, gm_fragment* out_frags,
                                 uint32_t out_cap, uint32_t* n_out, uint64_t* out_perm, uint32_t perm_cap, uint32_t* n_perm,
                                 uint64_t* out_data_len);
int32_t gm_frag_split(const gm_fragment* frags, uint32_t n_frags, uint64_t num_consts, const uint64_t* d_data,
                      const uint64_t* d_consts, uint64_t* d_l_data, uint64_t* d_r_data, uint64_t* d_l_consts,
                      uint64_t* d_r_consts, void* stream);
int32_t gm_frag_bind(const gm_fragment* frags, uint32_t n_frags, uint64_t num_consts, const uint64_t* d_data,
                     const uint64_t* d_consts, const uint64_t* h_t, uint64_t* d_out_data, uint64_t* d_out_consts, void* stream);
int32_t gm_frag_to_dense(const gm_fragment* frags, uint32_t n_frags, uint64_t num_consts, const uint64_t* d_data,
                         const uint64_t* d_consts, uint64_t* d_out, void* stream);
int32_t gm_segment_split(uint64_t start, uint64_t end, uint64_t* out_starts, uint8_t* out_loglengths, uint32_t cap,
                         uint32_t* n_out);
int32_t gm_frag_eq_materialize(const gm_fragment* frags, uint32_t n_frags, uint64_t num_consts, const uint64_t* h_multiplier,
                               const uint64_t* h_point, uint32_t nvars, uint64_t* d_values, uint64_t* h_sums, void* stream);

/* ---------------------------------------------------------------- BLS12-381 G1 (a12 G1 part, a13, a14, a15, a17)
 * The reference reaches G1 through ark-ec 0.4.2 (un-vendored).  Wire forms here (the Rust shim marshals, ark's structs are
 * not repr(C)):  Fq = Montgomery (R = 2^384) 6 x u64 LE;  affine point = x, y (12 x u64, 96 bytes), the point at infinity is
 * (0, 0);  Jacobian point = X, Y, Z (18 x u64, 144 bytes), infinity is Z = 0 (`Projective<g1::Config>`: x = X/Z^2, y = Y/Z^3).
 * `Projective` equality in ark-ec is equality of the group element: single results come back in affine form (`into_affine`),
 * bucket arrays as Jacobian points whose coordinates are NOT specified beyond the element they represent.
 * Scalars: canonical bigints (4 x u64) or, with scalars_mont = 1, Fr elements in Montgomery form (`into_bigint()` is applied
 * on the device, msm_nonaffine.rs:21-23).  nbits: scalar width to process (255 for full Fr; smaller when the caller knows a
 * bound, the analogue of the max_num_bits early exit msm_nonaffine.rs:93-104).  Each call synchronises `stream`.
 *
 *   gm_g1_msm            <G1 as VariableBaseMSM>::msm(&[G1Affine], &[Fr])             KzgProvingKey::commit kzg.rs:123-126,
 *                                                                                  CommitmentKey::commit_vec gkr_msm_simple.rs:59-62
 *   gm_g1_msm_nonaff     VariableBaseMsmNonaffine::msm_nonaff / msm_bigint_nonaff   msm_nonaffine.rs:34-50 (bases: &[G1Projective])
 *   gm_g1_bucket_sums    buckets[mapping[i]] += bases[i]                            pullback.rs:41-57
 *   gm_g1_pullback_msm   Pullback::bucketed_msm (image: Fr Montgomery)              pullback.rs:27-59
 *   gm_g1_weighted_sum   acc = sum_i i * bucket[g][i] for n_groups bucket arrays     pushforward.rs:504-524 (running-sum loop)
 *   gm_g1_prepare_bases  prepare_bases: ceil(n/gamma) tables of 2^gamma - 1 affine   binary_msm.rs:32-48
 *   gm_g1_binary_msm     binary_msm(coefs, tables)  (coefs = prepare_coefs bytes)   binary_msm.rs:19-29
 *   gm_msm_g1_outer      d_outer_buckets, c_outer_buckets, d_comm, c_comm of PushForwardState::new from the digits / counter of
 *                        the plan's last gm_msm_run                                 pushforward.rs:395-456, 504-524
 *                        (phase 2, pushforward.rs:596-605 = gm_g1_msm_nonaff over these bucket arrays with the eq tables)
 *   gm_g1_host / gm_g1_batch  elementwise point arithmetic on host / device memory: op 0 add (jac, jac), 1 double (jac),
 *                        2 mixed add (jac, aff), 3 into_affine (jac -> aff), 4 affine + affine -> jac; host only: 5 on-curve
 *                        check (one u64 per point), 6 Fq multiplication, 7 Fq multiplication, device formulation */
int32_t gm_g1_msm(const uint64_t* d_bases_aff, const uint64_t* d_scalars, uint64_t n, int32_t scalars_mont, uint32_t nbits,
                  uint64_t* h_out_aff, void* stream);
int32_t gm_g1_msm_nonaff(const uint64_t* d_bases_jac, const uint64_t* d_scalars, uint64_t n, int32_t scalars_mont,
                         uint32_t nbits, uint64_t* h_out_aff, void* stream);
/* n_groups MSMs over projective bases against ONE scalar array: group g = d_bases_jac[g * stride .. + h_n[g]) with scalars[0 .. h_n[g]);
 * the pull commitments of second_phase (pushforward.rs:596-605: every outer-bucket array against the same eq table) in one call */
int32_t gm_g1_msm_nonaff_grouped(const uint64_t* d_bases_jac, uint64_t stride, const uint32_t* h_n, uint32_t n_groups,
                                 const uint64_t* d_scalars, int32_t scalars_mont, uint32_t nbits, uint64_t* h_out_aff, void* stream);
int32_t gm_g1_bucket_sums(const uint64_t* d_bases_aff, const uint32_t* d_mapping, uint64_t n, uint32_t n_buckets,
                          uint64_t* d_out_jac, void* stream);
int32_t gm_g1_pullback_msm(const uint64_t* d_bases_aff, const uint32_t* d_mapping, uint64_t n, const uint64_t* d_image,
                           uint32_t image_len, uint64_t* h_out_aff, void* stream);
int32_t gm_g1_weighted_sum(const uint64_t* d_buckets_jac, uint32_t n_groups, uint32_t group_len, uint64_t* h_out_aff,
                           void* stream);
int32_t gm_g1_prepare_bases(const uint64_t* d_bases_aff, uint64_t n, uint32_t gamma, uint64_t* d_tables_aff, void* stream);
int32_t gm_g1_binary_msm(const uint8_t* d_coefs, const uint64_t* d_tables_aff, uint64_t n_chunks, uint32_t gamma,
                         uint64_t* h_out_aff, void* stream);
int32_t gm_msm_g1_outer(const gm_msm_plan* plan, const uint64_t* d_basis_aff, uint32_t commitment_log_multiplicity,
                        uint64_t* d_d_outer_jac, uint64_t* d_c_outer_jac, uint64_t c_outer_cap, uint32_t* c_stride,
                        uint64_t* h_d_comm_aff, uint64_t* h_c_comm_aff, void* stream);
/* One rank's part of gm_msm_g1_outer for a window-sharded plan, and the cross-GPU combine (pushforward.rs:395-396,
 * 431-456: with commitment_log_multiplicity > 0 a commitment matrix spans 2^clm windows, i.e. several ranks).  The rank holds only
 * the KZG key slices of its windows: d_basis_local = n_slots * 2^x_logsize affine points, h_slot[s] = index of slice s
 * (kzg_basis[s * 2^x_logsize ..]) in it or -1.  Outputs: the rank's PARTIAL outer buckets of matrices first_matrix ..
 * first_matrix + n_matrices - 1 and its share of d_comm / c_comm (n_matrices Jacobian points each).  The protocol only ever takes
 * linear images of the outer buckets, so gm_g1_combine_parts -- all-gather of n points per rank, added on the host -- is the whole
 * exchange: one group element per matrix and commitment. */
int32_t gm_msm_g1_outer_part(const gm_msm_plan* plan, const uint64_t* d_basis_local, const int32_t* h_slot, uint32_t clm,
                             uint64_t* d_d_outer, uint64_t* d_c_outer, uint64_t c_outer_cap, uint32_t* c_stride,
                             uint32_t* first_matrix, uint32_t* n_matrices, uint64_t* h_d_part_jac, uint64_t* h_c_part_jac, void* stream);
int32_t gm_g1_combine_parts(const struct gm_comm* comm, const uint64_t* h_parts_jac, uint32_t n, uint64_t* h_out_aff);
int32_t gm_g1_generator(uint64_t* h_out_aff);   /* the standard generator, affine wire form (mock_setup's g0) */
int32_t gm_g1_to_affine(const uint64_t* d_in_jac, uint64_t n, uint64_t* d_out_aff, void* stream);
int32_t gm_g1_from_affine(const uint64_t* d_in_aff, uint64_t n, uint64_t* d_out_jac, void* stream);
int32_t gm_g1_host(int32_t op, const uint64_t* h_a, const uint64_t* h_b, uint64_t* h_out, uint64_t n);
int32_t gm_g1_batch(int32_t op, const uint64_t* d_a, const uint64_t* d_b, uint64_t* d_out, uint64_t n, void* stream);
/* synthetic SRS for bench / tests: n points k_i * G (G = the standard G1 generator), affine */
int32_t gm_g1_gen_points(uint64_t* d_points_aff, uint64_t n, uint64_t seed, void* stream);
/* KzgProvingKey::mock_setup (commitments/kzg.rs:84-98): ptau_1[i] = tau^i * g0, i < n (tau: Fr Montgomery; g0: affine).  For
 * tests and the bench's end-to-end check: with a known tau the pairing equation <A, H0> = <B, H1> of a proof reads A = tau * B. */
int32_t gm_g1_mock_srs(const uint64_t* h_tau, const uint64_t* h_g0_aff, uint64_t n, uint64_t* d_out_aff, void* stream);
/* Fixed-base precomputation for a base array many MSMs will use (the KZG proving key: every commit / open of a proof runs
 * against it): 2^(16 w) * base_i for the 16 windows of a 255-bit scalar, affine, 1536 bytes per base (3.2 GB at 2^21 bases).
 * Afterwards gm_g1_msm calls whose d_bases_aff is this pointer (n up to the registered length) drop every window into one
 * set of 2^16 buckets: the same group element with ~28 % fewer field multiplications.  One-off cost ~0.4 s at 2^21 bases.
 * The bases must stay unchanged and allocated until the release. */
int32_t gm_g1_fixed_base_register(const uint64_t* d_bases_aff, uint64_t n, void* stream);
int32_t gm_g1_fixed_base_release(const uint64_t* d_bases_aff);
/* frees the grow-only device scratch the G1 calls on the CURRENT device share (one scratch and one queue of G1 calls per device) */
int32_t gm_g1_release_scratch(void);

/* The G1 column commitments gkr_msm_prove makes before the GKR (gkr_msm_simple.rs:117-151): 2^log_num_bit_columns bit columns
 * through CommitmentKey::commit_bitvec = binary_msm(prepare_coefs(bits, gamma), binary_extended_bases) and the point column
 * (x coordinates, y coordinates, zero padding) through commit_vec = G::msm.  d_bases_aff: col_size = 2^(lp + lb - log_cols) affine
 * bases; d_binary_tables_aff: gm_g1_prepare_bases(d_bases_aff, col_size, gamma).  Outputs: affine points, in transcript order. */
int32_t gm_gkr_msm_commit(const uint64_t* d_points_xy, const uint8_t* d_scalar_bits, uint32_t log_num_points,
                          uint32_t log_num_scalar_bits, uint32_t log_num_bit_columns, const uint64_t* d_bases_aff,
                          const uint64_t* d_binary_tables_aff, uint32_t gamma, uint64_t* h_bit_comms_aff, uint64_t* h_pts_comm_aff,
                          void* stream);

/* Bandersnatch ScalarField (Montgomery, as stored by ark `Fr` of ark-ed-on-bls12-381-bandersnatch)
 * -> canonical bigint: the `into_bigint()` of pushforward.rs:352 / msm_nonaffine.rs:21-23. */
int32_t gm_bs_scalars_into_bigint(const uint64_t* d_in, uint64_t* d_out, uint64_t n, void* stream);

/* Synthetic inputs (bench / tests): n points k_i*G of the prime-order subgroup, affine Montgomery. */
int32_t gm_gen_points(uint64_t* d_points_xy, uint64_t n, uint64_t seed, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GKRMSM_H */
