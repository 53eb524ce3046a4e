/* TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement ("oracle") of the Pippenger-MSM + GKR-sumcheck hot path of morgana-proofs/GKR-MSM.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (libgkrmsm_hip.so) never links or calls it.
 *
 * Parity status: the Rust reference cannot be built in this environment (no cargo/rustc, un-vendored
 * arkworks/liblasso/merlin), and its tests hold no byte-level golden vectors.  The oracle is pinned by
 *   (1) the reference's integer KAT  COEFF_D  (src/utils.rs:35),
 *   (2) the algebraic identities the reference's own tests assert (layers == affine group law,
 *       GKR output == MSM, optimised round polynomials == naive ones, prover/verifier agreement),
 *   (3) an independent Python big-int restatement (oracle/pyref) and fixtures generated from it
 *       (tests/golden, scripts/make_golden.py).
 * Byte parity against the Rust binary itself is therefore "parity unpinned" (see DESIGN.md).
 *
 * All field elements are BLS12-381 Fr in Montgomery form, 4 x u64 little-endian (ark-ff layout).
 */
#ifndef GKRMSM_ORACLE_H
#define GKRMSM_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t l[4]; } or_fr;

/* AlgFn descriptor: same encoding as gm_fn in include/gkrmsm.h (kept independent on purpose) */
typedef struct { int32_t nseg; int32_t prim[4]; int32_t count[4]; } or_fn;

/* ---- field (src/utils.rs:22-49) */
void or_fr_add(or_fr* r, const or_fr* a, const or_fr* b);
void or_fr_sub(or_fr* r, const or_fr* a, const or_fr* b);
void or_fr_mul(or_fr* r, const or_fr* a, const or_fr* b);
void or_fr_neg(or_fr* r, const or_fr* a);
void or_fr_inv(or_fr* r, const or_fr* a);
void or_fr_mul_by_a(or_fr* r, const or_fr* a);
void or_fr_mul_by_d(or_fr* r, const or_fr* a);
void or_fr_batch(int op, const or_fr* a, const or_fr* b, or_fr* out, uint64_t n);
void or_coeff_d(or_fr* r);

/* ---- AlgFn (cleanup/utils/twisted_edwards_ops.rs:10-80, algfn.rs:129-292) */
int or_fn_n_ins(const or_fn* f);
int or_fn_n_outs(const or_fn* f);
void or_fn_exec(const or_fn* f, const or_fr* in, or_fr* out);

/* ---- dense polynomials (cleanup/polys/dense.rs, cleanup/protocols/sumcheck.rs:160-163, utils.rs:222-250) */
void or_dense_bind(const or_fr* in, uint64_t n, const or_fr* t, or_fr* out /* n/2 */);
void or_dense_make21(or_fr* v, uint64_t n);
void or_dense_bind21(const or_fr* in, uint64_t n, const or_fr* t, or_fr* out /* n/2 */);
void or_dense_map(const or_fn* f, const or_fr* const* cols_in, uint64_t n, or_fr* const* cols_out);
/* split on bit `lo_bit` of the index, bundle interleave as dense.rs:137-138; outputs 2*n_outs cols of n/2 */
void or_dense_map_split(const or_fn* f, const or_fr* const* cols_in, uint64_t n, uint32_t lo_bit, uint32_t bundle,
                        or_fr* const* cols_out);
void or_eq_table(const or_fr* mult, const or_fr* pt, uint32_t nvars, or_fr* out /* 2^nvars */);

/* ---- Pippenger MSM (pushforward.rs:351-429 Fr part, bintree_add.rs:137-239, triangle_add.rs:101-158,
 *      pippenger_ending.rs:46-58, pippenger.rs:531-534, 586-602).
 * Windows [y_begin, y_end).  Outputs (any may be NULL):
 *   digits  u16 [(y-y_begin)*N + x], counter u32 likewise, row_len u32 [nrows_local]
 *   bucket sums bx,by,bz [nrows_local] ; window points cols [3*(d+1)][nwin_local] ; threads = OpenMP threads */
int or_msm(const or_fr* points_xy, const uint64_t* scalars /* N x 4 canonical */, uint32_t x_logsize,
           uint32_t d_logsize, uint32_t y_size, uint32_t y_begin, uint32_t y_end, int threads, uint16_t* digits,
           uint32_t* counter, uint32_t* row_len, or_fr* bx, or_fr* by, or_fr* bz, or_fr* window_cols);
/* Horner recombination of ALL windows' points -> affine (x, y) */
void or_msm_combine(const or_fr* window_cols, uint32_t d_logsize, uint32_t n_windows, or_fr* out_xy);

#ifdef __cplusplus
}
#endif

/* ---- "prove image part" on the CPU (added after the first header block; same restatement rules):
 *   PippengerWG::new (Fr part)  pippenger.rs:37-70, splits.rs:172-176, pippenger_ending.rs:32-95
 *   PippengerBucketed::prove + GlueSplit::prove   pippenger_ending.rs:142-149, splits.rs:185-197
 * with the sumcheck objects of sumchecks/dense_eq.rs:61-173, sumchecks/vecvec_eq.rs:72-398, sumcheck.rs:237-347.
 * Challenges come from a tape (canonical 4 x u64 each, < 2^128); messages are returned in write order. */
#ifdef __cplusplus
extern "C" {
#endif
typedef struct or_pip_witness or_pip_witness;
/* CPU-baseline variant switch: 0 = fair (all loops threaded), 1 = reference-faithful (serial where the reference is serial) */
void or_set_reference_faithful(int on);
or_pip_witness* or_pip_witness_create(const or_fr* points_xy, const uint64_t* scalars, uint32_t x_logsize,
                                      uint32_t d_logsize, uint32_t y_size, uint32_t y_logsize, int threads);
void or_pip_witness_destroy(or_pip_witness* w);
/* dense output (pippenger.rs:531-534): 3*(d+1) columns of 2^y_logsize, written column-major into out */
void or_pip_witness_output(const or_pip_witness* w, or_fr* out);
int or_pip_prove_image_part(or_pip_witness* w, const or_fr* claim_point, const or_fr* claim_evs, const uint64_t* tape,
                            uint64_t n_tape, or_fr* msgs, uint64_t msgs_cap, uint64_t* n_msgs, or_fr* final_point,
                            uint32_t* n_final_point, or_fr* final_evs, uint64_t* tape_used, uint64_t* rounds, int threads);
/* gen-1 prover gkr_msm_prove, Fr part (gkr_msm_simple.rs:86-338); bits: 2^lp * 2^lb bytes; tape: canonical field elements */
int or_gkr_msm_prove(const or_fr* points_xy, const uint8_t* bits, uint32_t lp, uint32_t lb, const uint64_t* tape,
                     uint64_t n_tape, or_fr* msgs, uint64_t msgs_cap, uint64_t* n_msgs, or_fr* output, or_fr* final_point,
                     uint32_t* n_final_point, or_fr* final_evs, uint64_t* tape_used, uint64_t* rounds, int threads);
#ifdef __cplusplus
}
#endif
#endif /* GKRMSM_ORACLE_H */
