// A host-side ProofTranscript2 for callers without a Rust transcript at hand: merlin v1.0 (STROBE-128 over
// Keccak-f[1600]) used exactly as /root/reference/src/cleanup/proof_transcript.rs:76-136 does --
//   start_prover(pparam)  = merlin::Transcript::new(pparam)
//   write_raw_msg(bytes)  = append_message(b"", bytes) + proof.extend(bytes)
//   raw_challenge(n)      = challenge_bytes(b"", n bytes)
//   write_scalars         = 32-byte little-endian canonical elements (ark-serialize compressed Fp)
//   write_points::<G1>    = 48-byte compressed points in ark-bls12-381 0.4's encoding (big-endian x; the three top bits of byte 0
//                           are: compressed, infinity, y lexicographically largest)
//   challenge(bits)       = from_le_bytes_mod_order(raw_challenge(ceil(bits / 8)))
// merlin (3.0.0, Cargo.lock:976-977) and ark-serialize are un-vendored dependencies: this is a restatement of their published
// formats, pinned by public vectors (SHA3-256 for the permutation, merlin's own "test protocol" challenge for the
// transcript framing; tests/test_merlin_cpu.py) -- not by bytes of the Rust binary (parity unpinned).
// No GPU work here: it plugs into the provers through gm_transcript.
#include <string>
#include <vector>

#include "g1.hip.h"
#include "internal.hpp"

using namespace gm;

namespace {

inline uint64_t rotl64(uint64_t x, int s) { return (x << s) | (x >> (64 - s)); }

void keccak_f1600(uint64_t st[25]) {
    static const uint64_t RC[24] = {0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
                                    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
                                    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
                                    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
                                    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
                                    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    static const int ROT[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
    static const int PIL[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
    for (int round = 0; round < 24; round++) {
        uint64_t bc[5];
        for (int i = 0; i < 5; i++) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
        for (int i = 0; i < 5; i++) {
            const uint64_t t = bc[(i + 4) % 5] ^ rotl64(bc[(i + 1) % 5], 1);
            for (int j = 0; j < 25; j += 5) st[j + i] ^= t;
        }
        uint64_t t = st[1];
        for (int i = 0; i < 24; i++) {
            const int j = PIL[i];
            const uint64_t b = st[j];
            st[j] = rotl64(t, ROT[i]);
            t = b;
        }
        for (int j = 0; j < 25; j += 5) {
            for (int i = 0; i < 5; i++) bc[i] = st[j + i];
            for (int i = 0; i < 5; i++) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
        }
        st[0] ^= RC[round];
    }
}

// STROBE-128 as merlin uses it (merlin/src/strobe.rs)
struct Strobe128 {
    static constexpr int R = 166;
    static constexpr uint8_t FI = 1, FA = 2, FC = 4, FM = 16, FK = 32;   // STROBE flags (T = 8 is unused by merlin)
    uint8_t st[200];
    uint8_t pos = 0, pos_begin = 0, cur_flags = 0;
    void permute() {
        uint64_t w[25];
        for (int i = 0; i < 25; i++) {
            uint64_t v = 0;
            for (int b = 7; b >= 0; b--) v = (v << 8) | st[8 * i + b];
            w[i] = v;
        }
        keccak_f1600(w);
        for (int i = 0; i < 25; i++)
            for (int b = 0; b < 8; b++) st[8 * i + b] = (uint8_t)(w[i] >> (8 * b));
    }
    void run_f() {
        st[pos] ^= pos_begin;
        st[pos + 1] ^= 0x04;
        st[R + 1] ^= 0x80;
        permute();
        pos = 0;
        pos_begin = 0;
    }
    void absorb(const uint8_t* d, size_t n) {
        for (size_t i = 0; i < n; i++) {
            st[pos] ^= d[i];
            pos++;
            if (pos == R) run_f();
        }
    }
    void squeeze(uint8_t* d, size_t n) {
        for (size_t i = 0; i < n; i++) {
            d[i] = st[pos];
            st[pos] = 0;
            pos++;
            if (pos == R) run_f();
        }
    }
    void begin_op(uint8_t flags, bool more) {
        if (more) return;  // continuing the same operation (merlin asserts cur_flags == flags)
        const uint8_t old_begin = pos_begin;
        pos_begin = pos + 1;
        cur_flags = flags;
        const uint8_t hdr[2] = {old_begin, flags};
        absorb(hdr, 2);
        const bool force_f = (flags & (FC | FK)) != 0;
        if (force_f && pos != 0) run_f();
    }
    explicit Strobe128(const uint8_t* label, size_t n) {
        memset(st, 0, 200);
        const uint8_t init[6] = {1, R + 2, 1, 0, 1, 96};
        memcpy(st, init, 6);
        memcpy(st + 6, "STROBEv1.0.2", 12);
        permute();
        meta_ad(label, n, false);
    }
    void meta_ad(const uint8_t* d, size_t n, bool more) { begin_op(FM | FA, more); absorb(d, n); }
    void ad(const uint8_t* d, size_t n, bool more) { begin_op(FA, more); absorb(d, n); }
    void prf(uint8_t* d, size_t n, bool more) { begin_op(FI | FA | FC, more); squeeze(d, n); }
};

struct Merlin {
    Strobe128 s;
    explicit Merlin(const uint8_t* label, size_t n) : s(reinterpret_cast<const uint8_t*>("Merlin v1.0"), 11) {
        append_message(reinterpret_cast<const uint8_t*>("dom-sep"), 7, label, n);
    }
    void append_message(const uint8_t* label, size_t ln, const uint8_t* msg, size_t n) {
        const uint32_t len = (uint32_t)n;
        const uint8_t le[4] = {(uint8_t)len, (uint8_t)(len >> 8), (uint8_t)(len >> 16), (uint8_t)(len >> 24)};
        s.meta_ad(label, ln, false);
        s.meta_ad(le, 4, true);
        s.ad(msg, n, false);
    }
    void challenge_bytes(const uint8_t* label, size_t ln, uint8_t* dest, size_t n) {
        const uint32_t len = (uint32_t)n;
        const uint8_t le[4] = {(uint8_t)len, (uint8_t)(len >> 8), (uint8_t)(len >> 16), (uint8_t)(len >> 24)};
        s.meta_ad(label, ln, false);
        s.meta_ad(le, 4, true);
        s.prf(dest, n, false);
    }
};

// F::from_le_bytes_mod_order: the little-endian integer reduced mod p, returned canonical
Fr from_le_bytes_mod_order(const uint8_t* b, size_t n) {
    const Fr c256 = fr_from_u64(256);
    Fr acc = fr_zero();
    for (size_t i = n; i-- > 0;) acc = fr_add(fr_mul(acc, c256), fr_from_u64(b[i]));
    return fr_from_mont(acc);
}

// (q - 1) / 2, big-endian bytes, for the "lexicographically largest" flag
bool fq_is_lex_largest(const Fq& y_canon) {
    static const uint32_t HALF[12] = {0xffffd555u, 0xdcff7fffu, 0x58a9ffffu, 0x0f55ffffu, 0x7b587b12u, 0xb3986950u,
                                      0x79c2895fu, 0xb23ba5c2u, 0x21a5d66bu, 0x258dd3dbu, 0x1cbff34du, 0x0d0088f5u};
    for (int i = 11; i >= 0; i--) {
        if (y_canon.l[i] > HALF[i]) return true;
        if (y_canon.l[i] < HALF[i]) return false;
    }
    return false;
}

void g1_compress(const G1Aff& p, uint8_t out[48]) {
    memset(out, 0, 48);
    if (g1_aff_is_inf(p)) { out[0] = 0x80 | 0x40; return; }
    const Fq x = fq_from_mont(p.x), y = fq_from_mont(p.y);
    for (int i = 0; i < 12; i++)
        for (int b = 0; b < 4; b++) out[47 - (4 * i + b)] = (uint8_t)(x.l[i] >> (8 * b));
    out[0] |= 0x80;
    if (fq_is_lex_largest(y)) out[0] |= 0x20;
}

// Affine::deserialize_compressed with validation (ark-ec short_weierstrass; flags in the top bits of byte 0: 0x80 compressed,
// 0x40 infinity, 0x20 y is the lexicographically largest root); false = malformed, off the curve or outside the subgroup
bool g1_decompress(const uint8_t in[48], G1Aff* out) {
    if (!(in[0] & 0x80)) return false;
    if (in[0] & 0x40) {
        if (in[0] & 0x20) return false;
        for (int i = 0; i < 48; i++)
            if ((i == 0 ? (in[0] & 0x1f) : in[i]) != 0) return false;
        out->x = fq_zero(); out->y = fq_zero();
        return true;
    }
    Fq xc;
    for (int i = 0; i < 12; i++) {
        uint32_t v = 0;
        for (int b = 3; b >= 0; b--) {
            uint8_t byte = in[47 - (4 * i + b)];
            if (4 * i + b == 47) byte &= 0x1f;
            v = (v << 8) | byte;
        }
        xc.l[i] = v;
    }
    for (int i = 11; i >= 0; i--) {  // canonical: x < q
        if (xc.l[i] < fq_p(i)) break;
        if (xc.l[i] > fq_p(i) || i == 0) return false;
    }
    const Fq x = fq_to_mont(xc);
    const Fq rhs = fq_add(fq_mul(fq_sqr(x), x), fq_dbl(fq_dbl(fq_one())));
    // q = 3 mod 4: sqrt = rhs^((q + 1) / 4)
    static const uint64_t E[6] = {0xee7fbfffffffeaabull, 0x07aaffffac54ffffull, 0xd9cc34a83dac3d89ull,
                                  0xd91dd2e13ce144afull, 0x92c6e9ed90d2eb35ull, 0x0680447a8e5ff9a6ull};
    Fq y = fq_one();
    for (int i = 5; i >= 0; i--)
        for (int b = 63; b >= 0; b--) {
            y = fq_sqr(y);
            if ((E[i] >> b) & 1) y = fq_mul(y, rhs);
        }
    if (!fq_eq(fq_sqr(y), rhs)) return false;  // not on the curve
    if (fq_is_lex_largest(fq_from_mont(y)) != ((in[0] & 0x20) != 0)) y = fq_neg(y);
    out->x = x; out->y = y;
    return g1_aff_in_subgroup_host(*out);  // Validate::Yes also rejects points outside the prime-order subgroup
}

}  // namespace

struct gm_merlin {
    Merlin m;
    std::vector<uint8_t> proof;
    bool verifier = false;   // PTMode::Verifier (proof_transcript.rs:10-14): messages are read from `proof` at `ctr`
    size_t ctr = 0;
    gm_merlin(const uint8_t* l, size_t n) : m(l, n) {}
    void write_raw(const uint8_t* d, size_t n) {
        m.append_message(nullptr, 0, d, n);
        proof.insert(proof.end(), d, d + n);
    }
    // read_raw_msg (proof_transcript.rs:119-130)
    const uint8_t* read_raw(size_t n) {
        if (ctr + n > proof.size()) return nullptr;
        const uint8_t* p = proof.data() + ctr;
        ctr += n;
        m.append_message(nullptr, 0, p, n);
        return p;
    }
};

static int32_t mt_write_scalars(void* ctx, const uint64_t* e, uint64_t n) {
    gm_merlin* t = static_cast<gm_merlin*>(ctx);
    if (t->verifier) return 1;  // write_raw_msg panics in verifier mode (proof_transcript.rs:134)
    std::vector<uint8_t> buf(32 * n);
    for (uint64_t i = 0; i < n; i++) {
        Fr v;
        memcpy(&v, e + 4 * i, 32);
        v = fr_from_mont(v);
        memcpy(buf.data() + 32 * i, &v, 32);  // little-endian canonical
    }
    t->write_raw(buf.data(), buf.size());
    return 0;
}
static int32_t mt_write_points(void* ctx, const uint64_t* aff, uint64_t n) {
    gm_merlin* t = static_cast<gm_merlin*>(ctx);
    if (t->verifier) return 1;
    std::vector<uint8_t> buf(48 * n);
    for (uint64_t i = 0; i < n; i++) {
        G1Aff p;
        memcpy(&p, aff + 12 * i, sizeof(G1Aff));
        g1_compress(p, buf.data() + 48 * i);
    }
    t->write_raw(buf.data(), buf.size());
    return 0;
}
static int32_t mt_challenge(void* ctx, uint32_t n, uint32_t bits, uint64_t* out) {
    gm_merlin* t = static_cast<gm_merlin*>(ctx);
    const size_t bs = (bits + 7) / 8;
    std::vector<uint8_t> buf(bs * n);
    t->m.challenge_bytes(nullptr, 0, buf.data(), buf.size());
    for (uint32_t i = 0; i < n; i++) {
        const Fr c = from_le_bytes_mod_order(buf.data() + bs * i, bs);
        memcpy(out + 4 * i, &c, 32);
    }
    return 0;
}

// read_scalars / read_points (proof_transcript.rs:46-49, 59-62): 1 = out of bounds, 2 = malformed element
static int32_t mt_read_scalars(void* ctx, uint64_t n, uint64_t* out) {
    gm_merlin* t = static_cast<gm_merlin*>(ctx);
    if (!t->verifier) return 3;
    const uint8_t* p = t->read_raw(32 * n);
    if (!p) return 1;
    for (uint64_t i = 0; i < n; i++) {
        Fr v;
        memcpy(&v, p + 32 * i, 32);
        for (int l = 7; l >= 0; l--) {  // deserialize_compressed rejects values >= p
            if (v.l[l] < fr_p(l)) break;
            if (v.l[l] > fr_p(l) || l == 0) return 2;
        }
        v = fr_to_mont(v);
        memcpy(out + 4 * i, &v, 32);
    }
    return 0;
}
static int32_t mt_read_points(void* ctx, uint64_t n, uint64_t* out) {
    gm_merlin* t = static_cast<gm_merlin*>(ctx);
    if (!t->verifier) return 3;
    const uint8_t* p = t->read_raw(48 * n);
    if (!p) return 1;
    for (uint64_t i = 0; i < n; i++) {
        G1Aff a;
        if (!g1_decompress(p + 48 * i, &a)) return 2;
        memcpy(out + 12 * i, &a, sizeof(G1Aff));
    }
    return 0;
}

extern "C" int32_t gm_merlin_create_verifier(const uint8_t* pparam, uint64_t len, const uint8_t* proof, uint64_t proof_len, gm_merlin** out) {
    GM_REQUIRE(out && (pparam || len == 0) && (proof || proof_len == 0), "null argument");
    gm_merlin* t = new gm_merlin(pparam, len);
    t->verifier = true;
    t->proof.assign(proof, proof + proof_len);
    *out = t;
    return GM_OK;
}
extern "C" int32_t gm_merlin_reader(gm_merlin* t, gm_transcript_reader* out) {
    GM_REQUIRE(t && out && t->verifier, "not a verifier transcript");
    out->ctx = t;
    out->read_scalars = mt_read_scalars;
    out->challenge = mt_challenge;
    out->read_points = mt_read_points;
    out->points_validated = 1;   // mt_read_points decompresses with the subgroup check (g1_decompress)
    out->reserved = 0;
    return GM_OK;
}
// bytes of the proof not read yet (a verifier may insist on 0)
extern "C" int32_t gm_merlin_unread(const gm_merlin* t, uint64_t* n) {
    GM_REQUIRE(t && n && t->verifier, "not a verifier transcript");
    *n = t->proof.size() - t->ctr;
    return GM_OK;
}

extern "C" int32_t gm_merlin_create(const uint8_t* pparam, uint64_t len, gm_merlin** out) {
    GM_REQUIRE(out && (pparam || len == 0), "null argument");
    *out = new gm_merlin(pparam, len);
    return GM_OK;
}
extern "C" int32_t gm_merlin_destroy(gm_merlin* t) {
    delete t;
    return GM_OK;
}
extern "C" int32_t gm_merlin_transcript(gm_merlin* t, gm_transcript* out) {
    GM_REQUIRE(t && out, "null argument");
    out->ctx = t;
    out->write_scalars = mt_write_scalars;
    out->challenge = mt_challenge;
    out->write_points = mt_write_points;
    return GM_OK;
}
// the proof = every message written so far, concatenated (ProofTranscript2::end)
extern "C" int32_t gm_merlin_proof(const gm_merlin* t, const uint8_t** bytes, uint64_t* len) {
    GM_REQUIRE(t && bytes && len, "null argument");
    *bytes = t->proof.data();
    *len = t->proof.size();
    return GM_OK;
}
// raw access for tests / other users of the same transcript
extern "C" int32_t gm_merlin_append_message(gm_merlin* t, const uint8_t* label, uint64_t label_len, const uint8_t* msg, uint64_t len) {
    GM_REQUIRE(t, "null argument");
    t->m.append_message(label, label_len, msg, len);
    return GM_OK;
}
extern "C" int32_t gm_merlin_challenge_bytes(gm_merlin* t, const uint8_t* label, uint64_t label_len, uint8_t* dest, uint64_t len) {
    GM_REQUIRE(t && dest, "null argument");
    t->m.challenge_bytes(label, label_len, dest, len);
    return GM_OK;
}
// Keccak-f[1600] on a 200-byte little-endian state (test hook: SHA3 known answers pin the permutation)
extern "C" int32_t gm_keccak_f1600(uint8_t* state200) {
    GM_REQUIRE(state200, "null argument");
    uint64_t w[25];
    for (int i = 0; i < 25; i++) {
        uint64_t v = 0;
        for (int b = 7; b >= 0; b--) v = (v << 8) | state200[8 * i + b];
        w[i] = v;
    }
    keccak_f1600(w);
    for (int i = 0; i < 25; i++)
        for (int b = 0; b < 8; b++) state200[8 * i + b] = (uint8_t)(w[i] >> (8 * b));
    return GM_OK;
}
