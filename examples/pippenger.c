/* The reference's example (examples/pippenger.rs: build_pippenger_data -> run_pippenger -> verify_pippenger) on top of the C ABI,
 * in plain C: what a caller of libgkrmsm_hip.so writes when there is no Rust around it.
 *
 *   ./pippenger [--x-logsize N] [--d-logsize D] [--nbits S] [--commitment-log-multiplicity M]
 *
 * Synthetic points (k_i * G on Bandersnatch, generated on the GPU) and uniformly random nbits-bit scalars, a mock-setup SRS
 * (KzgProvingKey::mock_setup with a random tau), the built-in ProofTranscript2 (merlin).  Prints the spans the reference's
 * tracing tree prints, the proof size, and the verifier's verdict; exit code 0 iff the proof verifies.
 * Build: make examples      (gcc, links against gkr_msm_amd/libgkrmsm_hip.so) */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "gkrmsm.h"

#define CHECK(call)                                                                                 \
    do {                                                                                            \
        int32_t rc__ = (call);                                                                      \
        if (rc__ != GM_OK) {                                                                        \
            fprintf(stderr, "%s:%d: %s -> %d: %s\n", __FILE__, __LINE__, #call, rc__, gm_last_error()); \
            exit(2);                                                                                \
        }                                                                                           \
    } while (0)

static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

static uint64_t rng_state = 0x474b524d534dull;
static uint64_t next_u64(void) { /* SplitMix64 */
    uint64_t z = (rng_state += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
/* a uniformly random element below 2^252 (< p), Montgomery form */
static void random_fr(uint64_t out[4]) {
    uint64_t c[4] = {next_u64(), next_u64(), next_u64(), next_u64() & ((1ull << 60) - 1)};
    CHECK(gm_fr_host(5, c, NULL, out, 1));
}

int main(int argc, char** argv) {
    uint32_t x_log = 10, d_log = 8, nbits = 128, clm = 0;
    for (int i = 1; i + 1 < argc; i += 2) {
        const uint32_t v = (uint32_t)strtoul(argv[i + 1], NULL, 10);
        if (!strcmp(argv[i], "--x-logsize") || !strcmp(argv[i], "-x")) x_log = v;
        else if (!strcmp(argv[i], "--d-logsize") || !strcmp(argv[i], "-d")) d_log = v;
        else if (!strcmp(argv[i], "--nbits") || !strcmp(argv[i], "-s")) nbits = v;
        else if (!strcmp(argv[i], "--commitment-log-multiplicity")) clm = v;
        else { fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
    }
    const uint32_t y_size = (nbits + d_log - 1) / d_log;
    uint32_t y_log = 0;
    while ((1u << y_log) < y_size) y_log++;
    const uint64_t n = 1ull << x_log, nv = x_log + clm, srs_len = (2ull << nv) - 1;
    int32_t ndev = 0;
    CHECK(gm_device_count(&ndev));
    if (ndev < 1) { fprintf(stderr, "no gfx950 device\n"); return 2; }
    CHECK(gm_set_device(0));
    printf("x_logsize %u, d_logsize %u, nbits %u -> y_size %u (y_logsize %u), commitment_log_multiplicity %u\n", x_log, d_log, nbits,
           y_size, y_log, clm);

    /* ---- build_pippenger_data (pippenger.rs:462-497) */
    double t0 = now_ms();
    void *d_pts = NULL, *d_sc = NULL, *d_srs = NULL, *d_inv = NULL;
    CHECK(gm_malloc(&d_pts, n * 64));
    CHECK(gm_malloc(&d_sc, n * 32));
    CHECK(gm_malloc(&d_srs, srs_len * 96));
    CHECK(gm_malloc(&d_inv, srs_len * 32));
    CHECK(gm_gen_points((uint64_t*)d_pts, n, 0x474b524d534dull, NULL));
    uint64_t* sc = (uint64_t*)calloc(n, 32);
    for (uint64_t i = 0; i < n; i++) {   /* nbits uniformly random bits, canonical little-endian limbs */
        for (uint32_t w = 0; w < 4; w++) {
            const uint32_t lo = 64 * w;
            uint64_t v = next_u64();
            if (nbits <= lo) v = 0;
            else if (nbits < lo + 64) v &= (1ull << (nbits - lo)) - 1;
            if (w == 3) v &= (1ull << 60) - 1;   /* stay below the Bandersnatch group order */
            sc[4 * i + w] = v;
        }
    }
    CHECK(gm_memcpy_h2d(d_sc, sc, n * 32, NULL));
    uint64_t tau[4], k[4], two[4] = {2, 0, 0, 0}, g0[12], h0[24], h1[24];
    random_fr(tau);
    CHECK(gm_fr_host(5, two, NULL, k, 1));
    CHECK(gm_g1_generator(g0));
    CHECK(gm_g1_mock_srs(tau, g0, srs_len, (uint64_t*)d_srs, NULL));
    CHECK(gm_g1_fixed_base_register((const uint64_t*)d_srs, srs_len, NULL));   /* proving-key precomputation */
    CHECK(gm_knuckles_setup(k, (uint32_t)nv, (uint64_t*)d_inv, NULL));
    CHECK(gm_kzg_mock_vk(tau, h0, h1));
    uint64_t* r = (uint64_t*)calloc(y_log ? y_log : 1, 32);
    for (uint32_t i = 0; i < y_log; i++) random_fr(r + 4 * i);
    CHECK(gm_stream_sync(NULL));
    printf("build data                               %9.1f ms\n", now_ms() - t0);

    /* ---- run_pippenger (pippenger.rs:500-560) */
    t0 = now_ms();
    gm_msm_plan* plan = NULL;
    gm_pippenger_wg* wg = NULL;
    const gm_pip_witness* wit = NULL;
    CHECK(gm_msm_plan_create(x_log, d_log, y_size, 0, y_size, &plan));
    CHECK(gm_msm_run(plan, (const uint64_t*)d_pts, (const uint64_t*)d_sc, NULL));
    CHECK(gm_pippenger_wg_create(plan, (const uint64_t*)d_pts, y_log, clm, (const uint64_t*)d_srs, &wg, NULL));
    CHECK(gm_pippenger_wg_witness(wg, &wit));
    uint64_t* evs = (uint64_t*)calloc(3 * (d_log + 1), 32);
    uint32_t n_evs = 0;
    CHECK(gm_pip_witness_claims(wit, r, evs, &n_evs));
    printf("compute buckets and commit phase 1       %9.1f ms\n", now_ms() - t0);

    t0 = now_ms();
    static const uint8_t label[] = "pippenger";
    gm_merlin* pt = NULL;
    gm_transcript tr;
    uint64_t pair[24], n_ch = 0, rounds = 0;
    CHECK(gm_merlin_create(label, sizeof(label) - 1, &pt));
    CHECK(gm_merlin_transcript(pt, &tr));
    CHECK(gm_pippenger_prove_tr(wg, r, evs, (const uint64_t*)d_inv, k, &tr, pair, &n_ch, &rounds));
    const uint8_t* proof = NULL;
    uint64_t proof_len = 0;
    CHECK(gm_merlin_proof(pt, &proof, &proof_len));
    printf("Pippenger::prove                         %9.1f ms   (%llu sumcheck rounds, %llu challenges, proof %llu bytes)\n",
           now_ms() - t0, (unsigned long long)rounds, (unsigned long long)n_ch, (unsigned long long)proof_len);

    /* the MSM result the proof is about: recombine the dense output (verify_pippenger, pippenger.rs:589-602) */
    const uint64_t* d_cols = NULL;
    uint64_t ncols = 0, col_len = 0, msm_xy[8];
    CHECK(gm_msm_window_points(plan, &d_cols, &ncols, &col_len));
    uint64_t* cols = (uint64_t*)malloc(ncols * col_len * 32);
    CHECK(gm_memcpy_d2h(cols, d_cols, ncols * col_len * 32, NULL));
    CHECK(gm_stream_sync(NULL));
    CHECK(gm_msm_combine_host(cols, d_log, y_size, msm_xy));
    uint64_t canon[8];
    CHECK(gm_fr_host(6, msm_xy, NULL, canon, 2));
    printf("msm result x = 0x%016llx%016llx%016llx%016llx\n", (unsigned long long)canon[3], (unsigned long long)canon[2],
           (unsigned long long)canon[1], (unsigned long long)canon[0]);

    /* ---- verify_pippenger (pippenger.rs:562-587) from the proof bytes alone */
    t0 = now_ms();
    gm_merlin* vt = NULL;
    gm_transcript_reader rd;
    uint64_t vpair[24], unread = 0;
    CHECK(gm_merlin_create_verifier(label, sizeof(label) - 1, proof, proof_len, &vt));
    CHECK(gm_merlin_reader(vt, &rd));
    int32_t rc = gm_pippenger_verify_tr(x_log, d_log, y_size, y_log, clm, r, evs, g0, k, &rd, vpair);
    if (rc == GM_OK) CHECK(gm_merlin_unread(vt, &unread));
    if (rc == GM_OK && unread == 0 && memcmp(pair, vpair, sizeof(pair)) == 0) rc = gm_kzg_verify_pair(vpair, h0, h1);
    else if (rc == GM_OK) rc = GM_ERR_VERIFY;
    printf("Pippenger::verify + verify_pair          %9.1f ms\n", now_ms() - t0);
    if (rc == GM_OK) printf("proof verified\n");
    else printf("PROOF REJECTED (%d): %s\n", rc, gm_last_error());

    /* a flipped proof byte must be rejected */
    uint8_t* bad = (uint8_t*)malloc(proof_len);
    memcpy(bad, proof, proof_len);
    bad[proof_len / 2] ^= 1;
    gm_merlin* bt = NULL;
    CHECK(gm_merlin_create_verifier(label, sizeof(label) - 1, bad, proof_len, &bt));
    CHECK(gm_merlin_reader(bt, &rd));
    int32_t rc_bad = gm_pippenger_verify_tr(x_log, d_log, y_size, y_log, clm, r, evs, g0, k, &rd, vpair);
    if (rc_bad == GM_OK) rc_bad = gm_kzg_verify_pair(vpair, h0, h1);
    printf("tampered proof %s\n", rc_bad == GM_ERR_VERIFY ? "rejected" : "NOT REJECTED");

    gm_merlin_destroy(bt);
    gm_merlin_destroy(vt);
    gm_merlin_destroy(pt);
    gm_pippenger_wg_destroy(wg);
    gm_msm_plan_destroy(plan);
    gm_g1_fixed_base_release((const uint64_t*)d_srs);
    gm_free(d_inv); gm_free(d_srs); gm_free(d_sc); gm_free(d_pts);
    free(bad); free(cols); free(evs); free(r); free(sc);
    return (rc == GM_OK && rc_bad == GM_ERR_VERIFY) ? 0 : 1;
}
