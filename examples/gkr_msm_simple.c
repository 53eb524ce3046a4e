/* gen-1: the flow of the reference's gkr_msm_simple_test (src/gkr_msm_simple.rs:363-427) -- commit the bit columns and the
 * point column, run gkr_msm_prove -- plus the verifier side the reference only exercises in protocol/bintree.rs' own tests,
 * in plain C on top of the C ABI.
 *
 *   ./gkr_msm_simple [--log-num-points P] [--log-num-scalar-bits B] [--log-num-bit-columns C] [--gamma G]
 *
 * Exit code 0 iff the verifier accepts the prover's transcript, ends on the prover's final claim, and rejects a tampered one. */
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
static uint64_t rng_state = 0x67656e31ull;
static uint64_t next_u64(void) {
    uint64_t z = (rng_state += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

int main(int argc, char** argv) {
    uint32_t lp = 5, lb = 8, lcols = 6, gamma = 5;   /* the reference test's shape */
    for (int i = 1; i + 1 < argc; i += 2) {
        const uint32_t v = (uint32_t)strtoul(argv[i + 1], NULL, 10);
        if (!strcmp(argv[i], "--log-num-points")) lp = v;
        else if (!strcmp(argv[i], "--log-num-scalar-bits")) lb = v;
        else if (!strcmp(argv[i], "--log-num-bit-columns")) lcols = v;
        else if (!strcmp(argv[i], "--gamma")) gamma = v;
        else { fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
    }
    if (lcols > lp + lb || gamma < 1 || gamma > 8) { fprintf(stderr, "bad shape\n"); return 2; }
    const uint64_t npts = 1ull << lp, size = 1ull << (lp + lb), ncols = 1ull << lcols, col_size = size >> lcols;
    const uint64_t nchunks = (col_size + gamma - 1) / gamma, tab = (1ull << gamma) - 1;
    CHECK(gm_set_device(0));
    printf("log_num_points %u, log_num_scalar_bits %u, log_num_bit_columns %u, gamma %u\n", lp, lb, lcols, gamma);

    void *d_pts = NULL, *d_bits = NULL, *d_bases = NULL, *d_tables = NULL;
    CHECK(gm_malloc(&d_pts, npts * 64));
    CHECK(gm_malloc(&d_bits, size));
    CHECK(gm_malloc(&d_bases, col_size * 96));
    CHECK(gm_malloc(&d_tables, nchunks * tab * 96));
    CHECK(gm_gen_points((uint64_t*)d_pts, npts, 0x67656e31ull, NULL));
    uint8_t* bits = (uint8_t*)malloc(size);   /* bits[point * 2^lb + bit] (gkr_msm_simple.rs:120) */
    for (uint64_t i = 0; i < size; i += 64) {
        uint64_t w = next_u64();
        for (uint64_t b = 0; b < 64 && i + b < size; b++) bits[i + b] = (uint8_t)((w >> b) & 1);
    }
    CHECK(gm_memcpy_h2d(d_bits, bits, size, NULL));
    CHECK(gm_g1_gen_points((uint64_t*)d_bases, col_size, 7, NULL));
    CHECK(gm_g1_prepare_bases((const uint64_t*)d_bases, col_size, gamma, (uint64_t*)d_tables, NULL));
    CHECK(gm_stream_sync(NULL));

    /* ---- commitments (gkr_msm_simple.rs:117-151), onto the transcript */
    double t0 = now_ms();
    uint64_t* comms = (uint64_t*)calloc(ncols + 1, 96);
    CHECK(gm_gkr_msm_commit((const uint64_t*)d_pts, (const uint8_t*)d_bits, lp, lb, lcols, (const uint64_t*)d_bases,
                            (const uint64_t*)d_tables, gamma, comms, comms + 12 * ncols, NULL));
    printf("commit %llu bit columns + the point column  %9.1f ms\n", (unsigned long long)ncols, now_ms() - t0);
    static const uint8_t label[] = "test";
    gm_merlin* pt = NULL;
    gm_transcript tr;
    CHECK(gm_merlin_create(label, sizeof(label) - 1, &pt));
    CHECK(gm_merlin_transcript(pt, &tr));
    if (tr.write_points(tr.ctx, comms, ncols + 1)) { fprintf(stderr, "write_points failed\n"); return 2; }

    /* ---- gkr_msm_prove */
    t0 = now_ms();
    const uint64_t nout = 1ull << lb;
    uint64_t* output = (uint64_t*)calloc(3 * nout, 32);
    uint64_t fpt[64 * 4], fev[3 * 4], n_ch = 0, rounds = 0;
    uint32_t n_fpt = 0;
    CHECK(gm_gkr_msm_prove_tr((const uint64_t*)d_pts, (const uint8_t*)d_bits, lp, lb, &tr, output, fpt, &n_fpt, fev, &n_ch, &rounds,
                              NULL));
    const uint8_t* proof = NULL;
    uint64_t proof_len = 0;
    CHECK(gm_merlin_proof(pt, &proof, &proof_len));
    printf("gkr_msm_prove                               %9.1f ms   (%llu sumcheck rounds, %llu challenges, transcript %llu bytes)\n",
           now_ms() - t0, (unsigned long long)rounds, (unsigned long long)n_ch, (unsigned long long)proof_len);

    /* ---- verifier: replay from the bytes */
    t0 = now_ms();
    int ok = 1;
    for (int pass = 0; pass < 2; pass++) {
        uint8_t* buf = (uint8_t*)malloc(proof_len);
        memcpy(buf, proof, proof_len);
        if (pass == 1) buf[48 * (ncols + 1) + 32 * 3 * nout + 40] ^= 1;   /* a byte of the first round polynomial */
        gm_merlin* vt = NULL;
        gm_transcript_reader rd;
        CHECK(gm_merlin_create_verifier(label, sizeof(label) - 1, buf, proof_len, &vt));
        CHECK(gm_merlin_reader(vt, &rd));
        uint64_t* vcomms = (uint64_t*)calloc(ncols + 1, 96);
        uint64_t vpt[64 * 4], vev[3 * 4], vrounds = 0, unread = 0;
        uint32_t n_vpt = 0;
        int32_t rc = rd.read_points(rd.ctx, ncols + 1, vcomms) ? GM_ERR_VERIFY : GM_OK;
        if (rc == GM_OK) rc = gm_gkr_msm_verify_tr(lp, lb, &rd, vpt, &n_vpt, vev, &vrounds);
        if (rc == GM_OK) CHECK(gm_merlin_unread(vt, &unread));
        const int same = rc == GM_OK && unread == 0 && n_vpt == n_fpt && !memcmp(vpt, fpt, 32 * n_fpt) && !memcmp(vev, fev, 96) &&
                         !memcmp(vcomms, comms, 96 * (ncols + 1));
        if (pass == 0) {
            printf("verifier                                    %9.1f ms\n", now_ms() - t0);
            printf(same ? "transcript verified: same final claim (%u coordinates) as the prover\n" : "TRANSCRIPT REJECTED (%u)\n", n_vpt);
            ok = ok && same;
        } else {
            printf("tampered transcript %s\n", rc == GM_ERR_VERIFY ? "rejected" : "NOT REJECTED");
            ok = ok && rc == GM_ERR_VERIFY;
        }
        gm_merlin_destroy(vt);
        free(vcomms);
        free(buf);
    }
    gm_merlin_destroy(pt);
    gm_free(d_tables); gm_free(d_bases); gm_free(d_bits); gm_free(d_pts);
    free(output); free(comms); free(bits);
    return ok ? 0 : 1;
}
