/* The whole gen-2 proof sharded by MSM windows over G ranks (BASELINE.json configs[4]'s structure), in plain C: ONE process
 * drives every GPU of the node, one host thread per rank (rank r on device r mod #devices), the ranks' host threads exchange the
 * round sums through the library's shared-memory communicator and pull each other's device buffers by address.
 *
 *   ./pippenger_sharded [--ranks G] [--x-logsize N] [--d-logsize D] [--nbits S] [--commitment-log-multiplicity M]
 *
 * Every rank: the plan of ITS windows (gm_msm_plan_create(..., y0, y1)), a view of the key ranges it reads
 * (gm_pippenger_sharded_key_ranges), gm_pippenger_wg_create_sharded, its slice of the Knuckles inverses table, and
 * gm_pippenger_prove_tr under its own merlin transcript.  Every rank ends with the SAME proof bytes and pairing pair (the unsharded
 * prover's); rank 0 verifies them with the host verifier and the pairing check.  Exit code 0 iff all of that holds.
 * The y_size = ceil(nbits / d) windows must be a power of two and divisible by G (a power of two).
 * Build: make examples      (gcc -pthread, links against gkr_msm_amd/libgkrmsm_hip.so) */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

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
static void random_fr(uint64_t out[4]) {
    uint64_t c[4] = {next_u64(), next_u64(), next_u64(), next_u64() & ((1ull << 60) - 1)};
    CHECK(gm_fr_host(5, c, NULL, out, 1));
}

/* what every rank thread shares (read-only) and what it leaves behind */
struct job {
    uint32_t world, x_log, d_log, nbits, clm, y_size, y_log;
    int32_t ndev;
    char shm_name[64];
    const uint64_t* sc;          /* n x 4 u64: canonical scalars */
    uint64_t tau[4], k[4], g0[12];
    const uint64_t* r;           /* y_log x 4 u64 */
    pthread_barrier_t bar;
};
struct rank_out {
    struct job* job;
    uint32_t rank;
    uint8_t* proof;
    uint64_t proof_len, pair[24], rounds, key_points;
    uint64_t evs[4 * 3 * 17];
    double new_ms, prove_ms;
};

static void* rank_main(void* arg) {
    struct rank_out* o = (struct rank_out*)arg;
    const struct job* j = o->job;
    const uint32_t rank = o->rank, world = j->world;
    const uint64_t n = 1ull << j->x_log, nv = j->x_log + j->clm, srs_len = (2ull << nv) - 1;
    CHECK(gm_set_device((int32_t)(rank % (uint32_t)j->ndev)));
    void* s = NULL;
    CHECK(gm_stream_create(&s));

    /* operands: replicated on every rank (points generated on the device, scalars uploaded) */
    void *d_pts = NULL, *d_sc = NULL, *d_srs = NULL, *d_inv = NULL;
    CHECK(gm_malloc(&d_pts, n * 64));
    CHECK(gm_malloc(&d_sc, n * 32));
    CHECK(gm_gen_points((uint64_t*)d_pts, n, 0x474b524d534dull, s));
    CHECK(gm_memcpy_h2d(d_sc, j->sc, n * 32, s));

    /* the proving key: a rank only needs the <= 4 ranges gm_pippenger_sharded_key_ranges names (at config E: 6.4 + 6.4 GB of the 51 GB
     * key).  A deployment uploads just those; the mock set-up here generates the powers of tau on the device and the view points INTO
     * them -- the ranges merged where they touch, range [0] inside one segment as the library asks. */
    CHECK(gm_malloc(&d_srs, srs_len * 96));
    CHECK(gm_g1_mock_srs(j->tau, j->g0, srs_len, (uint64_t*)d_srs, s));
    uint64_t first4[4], count4[4];
    CHECK(gm_pippenger_sharded_key_ranges(j->x_log, j->d_log, j->y_log, j->clm, rank, world, first4, count4));
    uint64_t lo[4], hi[4];
    uint32_t nseg = 0;
    for (int a = 0; a < 4; a++) {           /* insertion sort by first index, then merge */
        if (!count4[a]) continue;
        uint32_t p = nseg++;
        while (p > 0 && lo[p - 1] > first4[a]) { lo[p] = lo[p - 1]; hi[p] = hi[p - 1]; p--; }
        lo[p] = first4[a]; hi[p] = first4[a] + count4[a];
    }
    uint32_t m = 0;
    for (uint32_t a = 0; a < nseg; a++) {
        if (m && lo[a] <= hi[m - 1]) { if (hi[a] > hi[m - 1]) hi[m - 1] = hi[a]; }
        else { lo[m] = lo[a]; hi[m] = hi[a]; m++; }
    }
    const uint64_t* seg_ptr[4];
    uint64_t seg_first[4], seg_count[4];
    o->key_points = 0;
    for (uint32_t a = 0; a < m; a++) {
        seg_ptr[a] = (const uint64_t*)d_srs + 12 * lo[a];
        seg_first[a] = lo[a];
        seg_count[a] = hi[a] - lo[a];
        o->key_points += seg_count[a];
    }
    const gm_key_view key = {m, 0, seg_ptr, seg_first, seg_count};

    /* the rank's slice of the Knuckles inverses table: entries [rank S, (rank + 1) S) of the 2N - 1 that exist, S = 2N / world */
    const uint64_t S = (2ull << nv) / world, inv_first = rank * S;
    const uint64_t inv_count = inv_first + S <= srs_len ? S : (srs_len > inv_first ? srs_len - inv_first : 0);
    CHECK(gm_malloc(&d_inv, (inv_count ? inv_count : 1) * 32));
    CHECK(gm_knuckles_setup_range(j->k, (uint32_t)nv, inv_first, inv_count, (uint64_t*)d_inv, s));

    /* the communicator: collective over the rank threads (processes on one node use it the same way) */
    gm_shm* shm = NULL;
    gm_comm comm;
    CHECK(gm_comm_shm_create(j->shm_name, rank, world, &shm));
    CHECK(gm_comm_shm_as_comm(shm, &comm));
    CHECK(gm_stream_sync(s));

    /* ---- PippengerWG::new, sharded: the rank's windows [y0, y1) */
    pthread_barrier_wait((pthread_barrier_t*)&j->bar);
    double t0 = now_ms();
    const uint32_t wpr = j->y_size / world, y0 = rank * wpr, y1 = y0 + wpr;
    gm_msm_plan* plan = NULL;
    gm_pippenger_wg* wg = NULL;
    const gm_pip_witness* wit = NULL;
    CHECK(gm_msm_plan_create(j->x_log, j->d_log, j->y_size, y0, y1, &plan));
    CHECK(gm_msm_run(plan, (const uint64_t*)d_pts, (const uint64_t*)d_sc, s));
    CHECK(gm_pippenger_wg_create_sharded(plan, (const uint64_t*)d_pts, j->y_log, j->clm, &key, &comm, &wg, s));
    CHECK(gm_pippenger_wg_witness(wg, &wit));
    uint32_t n_evs = 0;
    CHECK(gm_pip_witness_claims(wit, j->r, o->evs, &n_evs));   /* the dense output is whole on every rank (bucket sums are gathered) */
    o->new_ms = now_ms() - t0;

    /* ---- Pippenger::prove, sharded, under this rank's own merlin transcript: identical bytes on every rank */
    t0 = now_ms();
    static const uint8_t label[] = "pippenger";
    gm_merlin* pt = NULL;
    gm_transcript tr;
    uint64_t n_ch = 0;
    CHECK(gm_merlin_create(label, sizeof(label) - 1, &pt));
    CHECK(gm_merlin_transcript(pt, &tr));
    CHECK(gm_pippenger_prove_tr(wg, j->r, o->evs, (const uint64_t*)d_inv, j->k, &tr, o->pair, &n_ch, &o->rounds));
    o->prove_ms = now_ms() - t0;
    const uint8_t* proof = NULL;
    CHECK(gm_merlin_proof(pt, &proof, &o->proof_len));
    o->proof = (uint8_t*)malloc(o->proof_len);
    memcpy(o->proof, proof, o->proof_len);

    pthread_barrier_wait((pthread_barrier_t*)&j->bar);   /* nobody tears its buffers down while a peer may still read them */
    gm_merlin_destroy(pt);
    gm_pippenger_wg_destroy(wg);
    gm_msm_plan_destroy(plan);
    gm_comm_shm_destroy(shm);
    gm_free(d_inv); gm_free(d_srs); gm_free(d_sc); gm_free(d_pts);
    CHECK(gm_stream_destroy(s));
    return NULL;
}

int main(int argc, char** argv) {
    static struct job j;
    j.world = 2; j.x_log = 8; j.d_log = 4; j.nbits = 32; j.clm = 1;
    for (int i = 1; i + 1 < argc; i += 2) {
        const uint32_t v = (uint32_t)strtoul(argv[i + 1], NULL, 10);
        if (!strcmp(argv[i], "--ranks")) j.world = v;
        else if (!strcmp(argv[i], "--x-logsize") || !strcmp(argv[i], "-x")) j.x_log = v;
        else if (!strcmp(argv[i], "--d-logsize") || !strcmp(argv[i], "-d")) j.d_log = v;
        else if (!strcmp(argv[i], "--nbits") || !strcmp(argv[i], "-s")) j.nbits = v;
        else if (!strcmp(argv[i], "--commitment-log-multiplicity")) j.clm = v;
        else { fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
    }
    j.y_size = (j.nbits + j.d_log - 1) / j.d_log;
    while ((1u << j.y_log) < j.y_size) j.y_log++;
    if ((1u << j.y_log) != j.y_size || j.world == 0 || (j.world & (j.world - 1)) || j.y_size % j.world || j.world > 64 || j.d_log > 16) {
        fprintf(stderr, "need y_size = nbits / d_logsize a power of two, divisible by --ranks (a power of two)\n");
        return 2;
    }
    CHECK(gm_device_count(&j.ndev));
    if (j.ndev < 1) { fprintf(stderr, "no gfx950 device\n"); return 2; }
    printf("%u ranks on %d device(s); x_logsize %u, d_logsize %u, nbits %u -> %u windows, %u per rank; commitment_log_multiplicity %u\n",
           j.world, j.ndev, j.x_log, j.d_log, j.nbits, j.y_size, j.y_size / j.world, j.clm);
    snprintf(j.shm_name, sizeof(j.shm_name), "/gm-example-%d", (int)getpid());

    const uint64_t n = 1ull << j.x_log;
    uint64_t* sc = (uint64_t*)calloc(n, 32);
    for (uint64_t i = 0; i < n; i++)
        for (uint32_t w = 0; w < 4; w++) {
            const uint32_t lo = 64 * w;
            uint64_t v = next_u64();
            if (j.nbits <= lo) v = 0;
            else if (j.nbits < lo + 64) v &= (1ull << (j.nbits - lo)) - 1;
            if (w == 3) v &= (1ull << 60) - 1;
            sc[4 * i + w] = v;
        }
    j.sc = sc;
    uint64_t two[4] = {2, 0, 0, 0}, h0[24], h1[24];
    random_fr(j.tau);
    CHECK(gm_fr_host(5, two, NULL, j.k, 1));
    CHECK(gm_g1_generator(j.g0));
    CHECK(gm_kzg_mock_vk(j.tau, h0, h1));
    uint64_t* r = (uint64_t*)calloc(j.y_log ? j.y_log : 1, 32);
    for (uint32_t i = 0; i < j.y_log; i++) random_fr(r + 4 * i);
    j.r = r;
    pthread_barrier_init(&j.bar, NULL, j.world);

    struct rank_out* out = (struct rank_out*)calloc(j.world, sizeof(struct rank_out));
    pthread_t* th = (pthread_t*)calloc(j.world, sizeof(pthread_t));
    for (uint32_t k = 0; k < j.world; k++) {
        out[k].job = &j;
        out[k].rank = k;
        if (pthread_create(&th[k], NULL, rank_main, &out[k])) { fprintf(stderr, "pthread_create failed\n"); return 2; }
    }
    for (uint32_t k = 0; k < j.world; k++) pthread_join(th[k], NULL);

    int same = 1;
    const uint64_t srs_len = (2ull << (j.x_log + j.clm)) - 1;
    for (uint32_t k = 0; k < j.world; k++) {
        printf("rank %u: PippengerWG::new %8.1f ms, Pippenger::prove %8.1f ms (%llu rounds), key points held %llu of %llu\n", k, out[k].new_ms,
               out[k].prove_ms, (unsigned long long)out[k].rounds, (unsigned long long)out[k].key_points, (unsigned long long)srs_len);
        if (out[k].proof_len != out[0].proof_len || memcmp(out[k].proof, out[0].proof, out[0].proof_len) ||
            memcmp(out[k].pair, out[0].pair, sizeof(out[0].pair)))
            same = 0;
    }
    printf("proof %llu bytes; %s\n", (unsigned long long)out[0].proof_len,
           same ? "all ranks hold the same proof and pairing pair" : "RANKS DISAGREE");

    /* ---- verify_pippenger (pippenger.rs:562-587) on rank 0's bytes: host verifier + pairing */
    static const uint8_t label[] = "pippenger";
    gm_merlin* vt = NULL;
    gm_transcript_reader rd;
    uint64_t vpair[24], unread = 0;
    CHECK(gm_merlin_create_verifier(label, sizeof(label) - 1, out[0].proof, out[0].proof_len, &vt));
    CHECK(gm_merlin_reader(vt, &rd));
    int32_t rc = gm_pippenger_verify_tr(j.x_log, j.d_log, j.y_size, j.y_log, j.clm, j.r, out[0].evs, j.g0, j.k, &rd, vpair);
    if (rc == GM_OK) CHECK(gm_merlin_unread(vt, &unread));
    if (rc == GM_OK && unread == 0 && memcmp(out[0].pair, vpair, sizeof(vpair)) == 0) rc = gm_kzg_verify_pair(vpair, h0, h1);
    else if (rc == GM_OK) rc = GM_ERR_VERIFY;
    if (rc == GM_OK) printf("proof verified\n");
    else printf("PROOF REJECTED (%d): %s\n", rc, gm_last_error());
    gm_merlin_destroy(vt);
    for (uint32_t k = 0; k < j.world; k++) free(out[k].proof);
    free(out); free(th); free(r); free(sc);
    return (rc == GM_OK && same) ? 0 : 1;
}
