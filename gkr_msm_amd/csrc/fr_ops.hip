// Runtime entry points, elementwise field batch ops, synthetic inputs and the host-side scalar glue
// (final MSM recombination).  See include/gkrmsm.h.
#include "algfn.hip.h"
#include "common.hpp"

using namespace gm;

namespace gm {

__device__ __forceinline__ Fr fr_op(int op, const Fr& a, const Fr& b) {
    switch (op) {
        case 0: return fr_add(a, b);
        case 1: return fr_sub(a, b);
        case 2: return fr_mul(a, b);
        case 3: return fr_neg(a);
        case 4: return fr_inv(a);
        case 5: return fr_to_mont(a);
        case 6: return fr_from_mont(a);
        case 7: return fr_mul_by_a(a);
        default: return fr_mul_by_d(a);
    }
}

__global__ void k_fr_batch(int op, const Fr* __restrict__ a, const Fr* __restrict__ b, Fr* __restrict__ out,
                           uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        Fr x = fr_load(a + i);
        Fr y = (op <= 2) ? fr_load(b + i) : x;
        fr_store(out + i, fr_op(op, x, y));
    }
}

static Fr fr_op_host(int op, const Fr& a, const Fr& b) {
    switch (op) {
        case 0: return fr_add(a, b);
        case 1: return fr_sub(a, b);
        case 2: return fr_mul(a, b);
        case 3: return fr_neg(a);
        case 4: return fr_inv(a);
        case 5: return fr_to_mont(a);
        case 6: return fr_from_mont(a);
        case 7: return fr_mul_by_a(a);
        case 9: return fr_mul_c(a, b);  // the 32-bit-limb formulation the device compiles when the asm path is off
        default: return fr_mul_by_d(a);
    }
}

// ---- Bandersnatch scalar field: Montgomery -> canonical (one Montgomery reduction mod the group order)
__device__ __forceinline__ uint32_t bs_q(int i) {
    switch (i) {
        case 0: return 0x2876e7e1u; case 1: return 0x74fd06b5u; case 2: return 0x74190471u;
        case 3: return 0xff8f8700u; case 4: return 0x02687600u; case 5: return 0x0cce7602u;
        case 6: return 0xca675f52u; default: return 0x1cfb69d4u;
    }
}

__global__ void k_bs_into_bigint(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t t[9];
#pragma unroll
    for (int j = 0; j < 8; j++) t[j] = in[i * 8 + j];
    t[8] = 0;
    const uint32_t inv = 0x5cc063dfu;  // -q^-1 mod 2^32
#pragma unroll
    for (int r = 0; r < 8; r++) {
        uint32_t m = t[0] * inv;
        uint64_t k = ((uint64_t)m * bs_q(0) + t[0]) >> 32;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            k += (uint64_t)m * bs_q(j) + t[j];
            t[j - 1] = (uint32_t)k;
            k >>= 32;
        }
        k += t[8];
        t[7] = (uint32_t)k;
        t[8] = (uint32_t)(k >> 32);
    }
    // t < 2q : conditional subtract
    uint32_t d[8];
    uint64_t borrow = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        uint64_t v = (uint64_t)t[j] - bs_q(j) - borrow;
        d[j] = (uint32_t)v;
        borrow = (v >> 32) & 1;
    }
#pragma unroll
    for (int j = 0; j < 8; j++) out[i * 8 + j] = borrow ? t[j] : d[j];
}

// ---- synthetic points: P_i = k_i * G, k_i = i-th output of SplitMix64(seed); affine, Montgomery
struct P3 { Fr x, y, z; };

GM_HD P3 p3_add(const P3& p, const P3& q) {
    Fr in[6] = {p.x, p.y, p.z, q.x, q.y, q.z};
    Fr a[4], b[4], c[3];
    proj_l1(in, a);
    proj_l2(a, b);
    proj_l3(b, c);
    P3 r;
    r.x = c[0]; r.y = c[1]; r.z = c[2];
    return r;
}

// projective doubling (dbl-2008-bbjlp, a = -5): 3 products + 4 squares where the generic addition spends 13 products.  Used by the
// host recombination only, whose result is an affine (canonical) point: any correct formula gives the same output.
GM_HD P3 p3_dbl(const P3& p) {
    const Fr B = fr_sqr(fr_add(p.x, p.y)), C = fr_sqr(p.x), D = fr_sqr(p.y);
    const Fr E = fr_mul_by_a(C), F = fr_add(E, D), H = fr_sqr(p.z);
    const Fr J = fr_sub(F, fr_dbl(H));
    P3 r;
    r.x = fr_mul(fr_sub(fr_sub(B, C), D), J);
    r.y = fr_mul(F, fr_sub(E, D));
    r.z = fr_mul(F, J);
    return r;
}

GM_HD Fr bs_gen_x() {
    Fr r;
    r.l[0] = 0xe7ab47f5u; r.l[1] = 0xec2627e1u; r.l[2] = 0x4f01aa9cu; r.l[3] = 0x3e63de48u;
    r.l[4] = 0x53946dc4u; r.l[5] = 0xfe0f5c3bu; r.l[6] = 0xaeb2cfcdu; r.l[7] = 0x2d71920bu;
    return r;
}
GM_HD Fr bs_gen_y() {
    Fr r;
    r.l[0] = 0x1895bd34u; r.l[1] = 0x4e30593eu; r.l[2] = 0x32afbe4bu; r.l[3] = 0x156d738fu;
    r.l[4] = 0xcdeb75f4u; r.l[5] = 0x45ef0b1cu; r.l[6] = 0x37d2e71fu; r.l[7] = 0x6a7cca00u;
    return r;
}

GM_HD uint64_t splitmix64_at(uint64_t seed, uint64_t i) {
    uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void k_gen_points(Fr* __restrict__ out_xy, uint64_t n, uint64_t seed) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t k = splitmix64_at(seed, i) | 1ull;
    P3 g; g.x = bs_gen_x(); g.y = bs_gen_y(); g.z = fr_one();
    P3 acc; acc.x = fr_zero(); acc.y = fr_one(); acc.z = fr_one();
    for (int b = 63; b >= 0; b--) {
        acc = p3_add(acc, acc);
        if ((k >> b) & 1) acc = p3_add(acc, g);
    }
    Fr zi = fr_inv(acc.z);
    fr_store(out_xy + 2 * i, fr_mul(acc.x, zi));
    fr_store(out_xy + 2 * i + 1, fr_mul(acc.y, zi));
}

}  // namespace gm

// ------------------------------------------------------------------------------------------- C ABI
extern "C" const char* gm_last_error(void) { return err_buf(); }
extern "C" const char* gm_version(void) { return "gkrmsm-hip 0.1 (gfx950)"; }

extern "C" int32_t gm_device_count(int32_t* out_count) {
    GM_REQUIRE(out_count, "null out");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *out_count = 0; return set_err(GM_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *out_count = n;
    return GM_OK;
}
extern "C" int32_t gm_set_device(int32_t device) {
    GM_REQUIRE(device >= 0 && device < GM_MAX_DEVICES, "device id %d: this library's per-device tables hold %d devices", device, GM_MAX_DEVICES);
    GM_HIP(hipSetDevice(device));
    return GM_OK;
}
extern "C" int32_t gm_stream_sync(void* stream) { GM_HIP(hipStreamSynchronize(as_stream(stream))); return GM_OK; }
// a stream of the current device for callers without the HIP runtime of their own (plain C, the Rust shim): non-blocking, i.e. it does
// not synchronise with the device's default stream -- rank threads that share a device must not (their gate kernels wait for each other's hosts)
extern "C" int32_t gm_stream_create(void** out) {
    GM_REQUIRE(out, "null argument");
    hipStream_t s = nullptr;
    GM_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *out = s;
    return GM_OK;
}
extern "C" int32_t gm_stream_destroy(void* stream) {
    if (stream) GM_HIP(hipStreamDestroy(as_stream(stream)));
    return GM_OK;
}
extern "C" int32_t gm_malloc(void** out, size_t bytes) { GM_REQUIRE(out, "null out"); GM_HIP(hipMalloc(out, bytes ? bytes : 16)); return GM_OK; }
extern "C" int32_t gm_free(void* p) { GM_HIP(hipFree(p)); return GM_OK; }
extern "C" int32_t gm_release_cached_memory(void) { gm::dev_pool().release(); return GM_OK; }
// Take `bytes` of device memory from the driver NOW (set-up time, like loading the SRS) for the library's pool to cut its blocks
// from: the first proof then allocates at pool speed instead of the driver's ~25-40 GiB/s (DevPool, common.hpp).  Per device
// (gm_set_device); may be called more than once; requests the slabs cannot serve fall back to the driver.
extern "C" int32_t gm_reserve(uint64_t bytes) {
    GM_REQUIRE(bytes >= 256, "nothing to reserve");
    hipError_t e = gm::dev_pool().reserve((size_t)bytes);
    if (e != hipSuccess) return gm::set_err(GM_ERR_HIP, "gm_reserve(%llu): %s", (unsigned long long)bytes, hipGetErrorString(e));
    return GM_OK;
}
extern "C" int32_t gm_unreserve(void) {
    if (!gm::dev_pool().unreserve()) return gm::set_err(GM_ERR_STATE, "gm_unreserve: blocks cut from the reserved memory are still in use (destroy the handles first)");
    return GM_OK;
}
// out8 = {driver allocations so far, their bytes, bytes idling in the pool, bytes reserved, of which cut, blocks cut from the reserve so far, 0, 0}
extern "C" int32_t gm_memory_stats(uint64_t* out8) {
    GM_REQUIRE(out8, "null argument");
    gm::DevPool& p = gm::dev_pool();
    std::lock_guard<std::mutex> g(p.mu);
    out8[0] = p.n_driver_allocs; out8[1] = p.driver_alloc_bytes; out8[2] = p.idle_bytes; out8[3] = p.slab_bytes; out8[4] = p.slab_used;
    out8[5] = p.n_slab_cuts; out8[6] = out8[7] = 0;
    return GM_OK;
}
extern "C" int32_t gm_memcpy_h2d(void* d, const void* h, size_t bytes, void* stream) {
    GM_HIP(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, as_stream(stream)));
    GM_HIP(hipStreamSynchronize(as_stream(stream)));
    return GM_OK;
}
extern "C" int32_t gm_memcpy_d2h(void* h, const void* d, size_t bytes, void* stream) {
    GM_HIP(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    GM_HIP(hipStreamSynchronize(as_stream(stream)));
    return GM_OK;
}

extern "C" int32_t gm_memcpy_d2d(void* d, const void* sPtr, size_t bytes, void* stream) {
    GM_HIP(hipMemcpyAsync(d, sPtr, bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
    return GM_OK;
}

extern "C" int32_t gm_fn_shape(const gm_fn* f, int32_t* n_ins, int32_t* n_outs, int32_t* deg) {
    GM_REQUIRE(f && f->nseg >= 1 && f->nseg <= GM_FN_MAX_SEG, "bad gm_fn");
    GmFn g;
    g.nseg = f->nseg;
    for (int s = 0; s < f->nseg; s++) {
        GM_REQUIRE(f->prim[s] >= 1 && f->prim[s] <= 10 && f->count[s] >= 0, "bad gm_fn segment %d", s);
        g.prim[s] = f->prim[s];
        g.count[s] = f->count[s];
    }
    if (n_ins) *n_ins = fn_n_ins(g);
    if (n_outs) *n_outs = fn_n_outs(g);
    if (deg) *deg = fn_deg(g);
    return GM_OK;
}

extern "C" int32_t gm_fr_batch(int32_t op, const uint64_t* d_a, const uint64_t* d_b, uint64_t* d_out, uint64_t n,
                               void* stream) {
    GM_REQUIRE(op >= 0 && op <= 8, "bad op %d", op);
    GM_REQUIRE(d_a && d_out && (op > 2 || d_b), "null argument");
    if (n == 0) return GM_OK;
    unsigned blocks = ceil_div(n, 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_fr_batch, dim3(blocks), dim3(256), 0, as_stream(stream), op, reinterpret_cast<const Fr*>(d_a),
                       reinterpret_cast<const Fr*>(d_b), reinterpret_cast<Fr*>(d_out), n);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

extern "C" int32_t gm_fr_host(int32_t op, const uint64_t* h_a, const uint64_t* h_b, uint64_t* h_out, uint64_t n) {
    GM_REQUIRE(op >= 0 && op <= 9, "bad op %d", op);
    GM_REQUIRE(h_a && h_out && ((op > 2 && op != 9) || h_b), "null argument");
    for (uint64_t i = 0; i < n; i++) {
        Fr a, b;
        memcpy(&a, h_a + 4 * i, 32);
        if (op <= 2 || op == 9) memcpy(&b, h_b + 4 * i, 32); else b = a;
        Fr r = fr_op_host(op, a, b);
        memcpy(h_out + 4 * i, &r, 32);
    }
    return GM_OK;
}

extern "C" int32_t gm_fn_host(const gm_fn* f, const uint64_t* h_in, uint64_t* h_out, uint64_t n) {
    int32_t ni, no, dg;
    int32_t rc = gm_fn_shape(f, &ni, &no, &dg);
    if (rc) return rc;
    GM_REQUIRE(h_in && h_out, "null argument");
    Fr in[64], out[64];
    GM_REQUIRE(ni <= 64 && no <= 64, "function too wide");
    for (uint64_t r = 0; r < n; r++) {
        memcpy(in, h_in + 4 * r * ni, 32 * (size_t)ni);
        int io = 0, oo = 0;
        for (int s = 0; s < f->nseg; s++)
            for (int c = 0; c < f->count[s]; c++) {
                prim_exec(f->prim[s], in + io, out + oo);
                io += prim_n_ins(f->prim[s]);
                oo += prim_n_outs(f->prim[s]);
            }
        memcpy(h_out + 4 * r * no, out, 32 * (size_t)no);
    }
    return GM_OK;
}

extern "C" int32_t gm_bs_scalars_into_bigint(const uint64_t* d_in, uint64_t* d_out, uint64_t n, void* stream) {
    GM_REQUIRE(d_in && d_out, "null argument");
    if (n == 0) return GM_OK;
    hipLaunchKernelGGL(k_bs_into_bigint, dim3(ceil_div(n, 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const uint32_t*>(d_in), reinterpret_cast<uint32_t*>(d_out), n);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

extern "C" int32_t gm_gen_points(uint64_t* d_points_xy, uint64_t n, uint64_t seed, void* stream) {
    GM_REQUIRE(d_points_xy, "null argument");
    if (n == 0) return GM_OK;
    hipLaunchKernelGGL(k_gen_points, dim3(ceil_div(n, 128)), dim3(128), 0, as_stream(stream),
                       reinterpret_cast<Fr*>(d_points_xy), n, seed);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

// ---- host-only 4 x 64-bit form of Fr for the recombination below: the same Montgomery representation (R = 2^256; the 8 x 32 limbs
// of Fr read as 4 x u64), CIOS with 128-bit products -- about twice as fast on the host as the portable 8 x 32 code the device shares.
namespace {
struct F4 { uint64_t l[4]; };
constexpr uint64_t F4_P[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};
constexpr uint64_t F4_INV = 0xfffffffeffffffffull;   // -p^-1 mod 2^64
inline F4 f4_from(const Fr& a) { F4 r; memcpy(r.l, a.l, 32); return r; }
inline Fr f4_to(const F4& a) { Fr r; memcpy(r.l, a.l, 32); return r; }
inline bool f4_geq_p(const uint64_t* t) {
    for (int i = 3; i >= 0; i--) { if (t[i] > F4_P[i]) return true; if (t[i] < F4_P[i]) return false; }
    return true;
}
inline void f4_sub_p(uint64_t* t) {
    unsigned __int128 br = 0;
    for (int i = 0; i < 4; i++) { const unsigned __int128 d = (unsigned __int128)t[i] - F4_P[i] - br; t[i] = (uint64_t)d; br = (d >> 64) & 1; }
}
inline F4 f4_mul(const F4& a, const F4& b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        unsigned __int128 c = 0;
        for (int j = 0; j < 4; j++) { c += (unsigned __int128)a.l[j] * b.l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        const uint64_t m = t[0] * F4_INV;
        c = (unsigned __int128)m * F4_P[0] + t[0]; c >>= 64;
        for (int j = 1; j < 4; j++) { c += (unsigned __int128)m * F4_P[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    if (t[4] || f4_geq_p(t)) f4_sub_p(t);
    F4 r; memcpy(r.l, t, 32);
    return r;
}
inline F4 f4_add(const F4& a, const F4& b) {
    uint64_t t[4]; unsigned __int128 c = 0;
    for (int i = 0; i < 4; i++) { c += (unsigned __int128)a.l[i] + b.l[i]; t[i] = (uint64_t)c; c >>= 64; }
    if (c || f4_geq_p(t)) f4_sub_p(t);   // a, b < p < 2^255: no carry out, the test on c is for form
    F4 r; memcpy(r.l, t, 32);
    return r;
}
inline F4 f4_sub(const F4& a, const F4& b) {
    uint64_t t[4]; unsigned __int128 br = 0;
    for (int i = 0; i < 4; i++) { const unsigned __int128 d = (unsigned __int128)a.l[i] - b.l[i] - br; t[i] = (uint64_t)d; br = (d >> 64) & 1; }
    if (br) { unsigned __int128 c = 0; for (int i = 0; i < 4; i++) { c += (unsigned __int128)t[i] + F4_P[i]; t[i] = (uint64_t)c; c >>= 64; } }
    F4 r; memcpy(r.l, t, 32);
    return r;
}
inline F4 f4_x5(const F4& a) { const F4 a2 = f4_add(a, a), a4 = f4_add(a2, a2); return f4_add(a4, a); }
struct Q3 { F4 x, y, z; };
// projective doubling (dbl-2008-bbjlp, a = -5): E = a C = -5 C
inline Q3 q3_dbl(const Q3& p) {
    const F4 s = f4_add(p.x, p.y);
    const F4 B = f4_mul(s, s), C = f4_mul(p.x, p.x), D = f4_mul(p.y, p.y), H = f4_mul(p.z, p.z);
    const F4 zero = {{0, 0, 0, 0}};
    const F4 E = f4_sub(zero, f4_x5(C)), F = f4_add(E, D), J = f4_sub(F, f4_add(H, H));
    Q3 r;
    r.x = f4_mul(f4_sub(f4_sub(B, C), D), J);
    r.y = f4_mul(F, f4_sub(E, D));
    r.z = f4_mul(F, J);
    return r;
}
// projective addition (add-2008-bbjlp, a = -5): A = Z1 Z2, B = A^2, C = X1 X2, D = Y1 Y2, E = d C D, F = B - E, G = B + E,
// X3 = A F ((X1 + Y1)(X2 + Y2) - C - D), Y3 = A G (D - a C), Z3 = F G
inline Q3 q3_add(const Q3& p, const Q3& q, const F4& coeff_d) {
    const F4 A = f4_mul(p.z, q.z), B = f4_mul(A, A), C = f4_mul(p.x, q.x), D = f4_mul(p.y, q.y);
    const F4 E = f4_mul(coeff_d, f4_mul(C, D)), F = f4_sub(B, E), G = f4_add(B, E);
    const F4 S = f4_sub(f4_sub(f4_mul(f4_add(p.x, p.y), f4_add(q.x, q.y)), C), D);
    Q3 r;
    r.x = f4_mul(f4_mul(A, F), S);
    r.y = f4_mul(f4_mul(A, G), f4_add(D, f4_x5(C)));
    r.z = f4_mul(F, G);
    return r;
}
}  // namespace

// acc = sum over (window w, i >= 1) of 2^(d*w + i - 1) * P[i][w], Horner from the top
// (/root/reference/src/cleanup/protocols/pippenger.rs:586-602); affine output (canonical: independent of the formulas used on the
// way -- 256 doublings + 256 additions at config B, ~100 us in the 4 x 64 host form).
extern "C" int32_t gm_msm_combine_host(const uint64_t* h_cols, uint32_t d_logsize, uint32_t n_windows,
                                       uint64_t* h_out_xy) {
    GM_REQUIRE(h_cols && h_out_xy && n_windows >= 1 && d_logsize >= 1, "bad argument");
    const Fr* cols = reinterpret_cast<const Fr*>(h_cols);
    const F4 one = f4_from(fr_one()), dcoef = f4_from(fr_coeff_d());
    Q3 acc;
    acc.x = F4{{0, 0, 0, 0}}; acc.y = one; acc.z = one;
    for (int64_t w = (int64_t)n_windows - 1; w >= 0; w--) {
        for (int64_t i = d_logsize; i >= 1; i--) {
            Q3 p;
            p.x = f4_from(cols[(uint64_t)(3 * i + 0) * n_windows + w]);
            p.y = f4_from(cols[(uint64_t)(3 * i + 1) * n_windows + w]);
            p.z = f4_from(cols[(uint64_t)(3 * i + 2) * n_windows + w]);
            acc = q3_dbl(acc);
            acc = q3_add(acc, p, dcoef);
        }
    }
    const Fr az = f4_to(acc.z);
    const Fr zi = fr_inv(az);
    const Fr x = fr_mul(f4_to(acc.x), zi), y = fr_mul(f4_to(acc.y), zi);
    memcpy(h_out_xy, &x, 32);
    memcpy(h_out_xy + 4, &y, 32);
    return GM_OK;
}

extern "C" int32_t gm_msm_te(const uint64_t* d_points_xy, const uint64_t* d_scalars, uint32_t x_logsize,
                             uint32_t d_logsize, uint32_t nbits, uint64_t* h_out_xy, void* stream) {
    GM_REQUIRE(h_out_xy, "null out");
    GM_REQUIRE(d_logsize >= 2 && d_logsize <= 10, "d_logsize %u out of range", d_logsize);
    const uint32_t y_size = (nbits + d_logsize - 1) / d_logsize;
    gm_msm_plan* plan = nullptr;
    int32_t rc = gm_msm_plan_create(x_logsize, d_logsize, y_size, 0, y_size, &plan);
    if (rc) return rc;
    rc = gm_msm_run(plan, d_points_xy, d_scalars, stream);
    if (rc) { gm_msm_plan_destroy(plan); return rc; }
    const uint64_t* d_cols; uint64_t n_cols, col_len;
    gm_msm_window_points(plan, &d_cols, &n_cols, &col_len);
    uint64_t* h = new uint64_t[n_cols * col_len * 4];
    rc = gm_memcpy_d2h(h, d_cols, n_cols * col_len * 32, stream);
    if (rc == GM_OK) rc = gm_msm_combine_host(h, d_logsize, (uint32_t)col_len, h_out_xy);
    delete[] h;
    gm_msm_plan_destroy(plan);
    return rc;
}
