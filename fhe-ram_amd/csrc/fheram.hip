// C-ABI library of the MI355X-native FHE-RAM evaluator (include/fheram.h).
// Host orchestration of Ram::read / read_prepare_write / write (reference: src/ram.rs) over the
// fused HIP kernels in kernels.hpp.  No CPU compute path exists here: every ciphertext
// operation is a kernel launch, and a missing GPU is a hard error.
#include "../../include/fheram.h"
#include "kernels.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

using namespace fk;

namespace {

typedef unsigned __int128 u128;
thread_local std::string g_create_err;

uint64_t mulmod_u(uint64_t a, uint64_t b) { return (uint64_t)((u128)a * b % P_U64); }
uint64_t powmod_u(uint64_t a, uint64_t e) {
    uint64_t r = 1;
    while (e) { if (e & 1) r = mulmod_u(r, a); a = mulmod_u(a, a); e >>= 1; }
    return r;
}
double centred(uint64_t v) { return v > P_U64 / 2 ? -(double)(P_U64 - v) : (double)v; }
unsigned brv(unsigned x, int bits) { unsigned r = 0; for (int i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; } return r; }

// Twiddle table in the LDS layout of ntt_dev.hpp: W[2^s + J] (W[i] = psi^bitrev12(i)) is stored
// at 2^s + j*E^Q + hi with s = LOGE*Q + u, J = hi*2^u + j.
std::vector<double> make_twiddles() {
    std::vector<double> tw(N, 0.0);
    for (int s = 0; s < LOGN; s++) {
        const int Q = s / LOGE, u = s % LOGE, HQ = 1 << (LOGE * Q);
        for (int J = 0; J < (1 << s); J++) {
            const int hi = J >> u, j = J & ((1 << u) - 1);
            const uint64_t w = powmod_u(PSI_8192, brv((unsigned)((1 << s) + J), LOGN));
            tw[(1 << s) + j * HQ + hi] = centred(w);
        }
    }
    return tw;
}

struct ProfCls {
    uint64_t launches = 0, blocks = 0;
    double ms = 0.0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

}  // namespace

struct fheram_ctx {
    fheram_params p;
    int device = 0;
    hipStream_t stream = nullptr;    // main stream: every op is ordered on it
    hipStream_t stream2 = nullptr;   // side stream for work that is independent inside one op (write path)
    hipStream_t cur = nullptr;       // stream the launchers currently enqueue on
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // derived
    int ws = 0, n2 = 0, n_digits = 0;
    size_t rows = 0;        // GLWE rows per sub-RAM held by THIS context (all of them unless sharded)
    size_t rows_glob = 0;   // rows per sub-RAM of the whole RAM
    int shard = 0, n_shards = 1;   // row sharding: this context owns rows r = shard (mod n_shards)
    std::vector<std::vector<int>> base2d;
    static constexpr int S_CT = 3, S_ADDR = 4, S_EVK = 4, S_INV = 5, DNUM_CT = 3, DNUM_GGSW = 4;
    static constexpr size_t GLWE = (size_t)S_CT * 2 * N;                   // elements of a ct
    static constexpr size_t GLWE4 = (size_t)S_ADDR * 2 * N;                // one GGSW row ct
    static constexpr size_t GGSW = (size_t)DNUM_CT * 2 * GLWE4;            // elements of a GGSW
    static constexpr size_t ATK = (size_t)DNUM_CT * S_EVK * 2 * N;         // trace key
    static constexpr size_t EVK5 = (size_t)DNUM_GGSW * S_INV * 2 * N;      // inverse / tensor key
    // device
    double* d_tw = nullptr;
    double ninv = 0.0;
    double* d_atk = nullptr;       // [log_n] prepared trace keys
    double* d_atk_inv = nullptr;
    double* d_tsk = nullptr;
    int64_t gal[LOGN];
    bool keys_loaded = false;
    int32_t* d_data = nullptr;     // [ws][rows] GLWE
    int32_t* d_tree = nullptr;     // [ws] GLWE (tree[0][0])
    int32_t* d_scrA = nullptr;     // [ws][rows]  ping-pong arenas: every fused kernel is out of place
    int32_t* d_scrB = nullptr;     // [ws][rows]
    int32_t* d_scrC = nullptr;     // [ws][rows]
    int32_t* d_scrD = nullptr;     // [ws][rows]
    int32_t* d_res = nullptr;      // [ws]
    int32_t* d_tmp = nullptr;      // [ws]
    int32_t* d_tmp2 = nullptr;     // [ws]
    int32_t* d_w = nullptr;        // [ws]
    double* d_big = nullptr;       // [LIMB_SPLIT_MAX ciphertexts] un-normalised limbs of the limb-parallel path
    double* d_big2 = nullptr;      // same, for launches on the side stream
    int limb_split = 1;            // FHERAM_LIMB_SPLIT=0 disables the limb-parallel path
    int use_graph = 0;             // FHERAM_GRAPH=1: replay each op's launch sequence from a hipGraph (per address)
    int32_t* d_part = nullptr;     // [ws]            this shard's partial pack / the un-rotated ct_lo
    int32_t* d_gat[3] = {nullptr, nullptr, nullptr};   // [n_shards][ws] gathered partials + ping-pong (root)
    int nco = 0;                   // output columns per workgroup: 1 = split by column (2 workgroups per
                                   // ciphertext), 2 = one workgroup, 0 = choose per launch from the batch size
    int cus = 256;
    double* d_prep = nullptr;      // [max digits per coordinate] prepared GGSW
    double* d_prep2 = nullptr;     // second set (inverse coordinate 0, prepared on the side stream)
    int32_t* d_ggsw_tmp = nullptr; // [max digits per coordinate] std GGSW (inversion result)
    int32_t* d_ggsw_tmp2 = nullptr;
    int max_digits = 0;
    bool initialized = false, state = false, words_staged = false;
    std::vector<int32_t> h_i32;    // host staging
    // profiling
    bool profile = false;
    std::map<std::string, ProfCls> prof;
    std::vector<hipEvent_t> ev_pool;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    std::string err;
};

struct fheram_addr {
    fheram_ctx* ctx;   // owner (identity check only after creation)
    int32_t* d_ggsw;   // [n_digits] std-form GGSW, int32
    int n_digits;
    int device;
    hipGraphExec_t graph[3] = {nullptr, nullptr, nullptr};   // captured launch sequences: read, read_prepare_write, write
};

struct fheram_secret {
    fheram_ctx* ctx;
    int device;
    std::vector<int32_t> sk;   // N coefficients in {-1, 0, 1}
    double* d_hat;             // prepared (transform domain, 1/N folded in)
};

namespace {

int fail(fheram_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg; else g_create_err = msg;
    return code;
}
#define HIPCHK(c, call)                                                                     \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail((c), FHERAM_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

GlweRef ref(int32_t* p, long sy, long sx) { return GlweRef{p, sy, sx}; }

hipEvent_t get_event(fheram_ctx* c) {
    if (!c->ev_pool.empty()) { hipEvent_t e = c->ev_pool.back(); c->ev_pool.pop_back(); return e; }
    hipEvent_t e; hipEventCreate(&e); return e;
}
struct ProfScope {
    fheram_ctx* c; ProfCls* cls = nullptr; hipEvent_t a = nullptr;
    ProfScope(fheram_ctx* c_, const char* name, uint64_t blocks) : c(c_) {
        if (!c->profile) return;
        cls = &c->prof[name];
        cls->launches++; cls->blocks += blocks;
        a = get_event(c);
        hipEventRecord(a, c->cur);
    }
    ~ProfScope() {
        if (!cls) return;
        hipEvent_t b = get_event(c);
        hipEventRecord(b, c->cur);
        cls->pending.emplace_back(a, b);
    }
};
void prof_collect(fheram_ctx* c) {
    for (auto& kv : c->prof) {
        for (auto& pr : kv.second.pending) {
            hipEventSynchronize(pr.second);
            float ms = 0.f;
            hipEventElapsedTime(&ms, pr.first, pr.second);
            kv.second.ms += ms;
            c->ev_pool.push_back(pr.first); c->ev_pool.push_back(pr.second);
        }
        kv.second.pending.clear();
    }
}

int ilog2_ceil(size_t x) { int k = 0; while (((size_t)1 << k) < x) k++; return k; }
int galois_mod(int64_t g) { const int64_t m = 2 * N; return (int)(((g % m) + m) % m); }
int galois_inv_mod(int g) {   // g odd; g^(N-1) = g^-1 mod 2N
    int64_t r = 1, b = g, e = N - 1, m = 2 * N;
    while (e) { if (e & 1) r = r * b % m; b = b * b % m; e >>= 1; }
    return (int)r;
}
int64_t galois_element(int i) {   // GLWE::trace_galois_elements (keys.rs:39,158)
    if (i == 0) return -1;
    int64_t g = 5, e = (int64_t)1 << (i - 1), r = 1, m = 2 * N;
    while (e) { if (e & 1) r = r * g % m; g = g * g % m; e >>= 1; }
    return r;
}

// ---- narrowing / widening between the int64 ABI layout and the int32 device layout ---------
bool narrow(const int64_t* src, int32_t* dst, size_t n) {
    int64_t bad = 0;
    for (size_t i = 0; i < n; i++) {
        const int64_t v = src[i];
        bad |= (v > 65536) | (v < -65536);
        dst[i] = (int32_t)v;
    }
    return bad == 0;
}
void widen(const int32_t* src, int64_t* dst, size_t n) { for (size_t i = 0; i < n; i++) dst[i] = src[i]; }

int upload_i64(fheram_ctx* c, int32_t* dst, const int64_t* src, size_t n) {
    c->h_i32.resize(n);
    if (!narrow(src, c->h_i32.data(), n)) return fail(c, FHERAM_ERR_RANGE, "limb out of the normalised range [-2^16, 2^16]");
    HIPCHK(c, hipMemcpyAsync(dst, c->h_i32.data(), n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return FHERAM_OK;
}
int download_i64(fheram_ctx* c, int64_t* dst, const int32_t* src, size_t n) {
    c->h_i32.resize(n);
    HIPCHK(c, hipMemcpyAsync(c->h_i32.data(), src, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    widen(c->h_i32.data(), dst, n);
    return FHERAM_OK;
}

// ---- kernel launchers ---------------------------------------------------------------------
constexpr int LIMB_SPLIT_MAX = 64;   // ciphertexts per launch the limb-parallel path is used for (at most)
constexpr int EW_SLICES = 8;   // workgroups per ciphertext of the elementwise kernels (blockIdx.z)
// One workgroup per ciphertext does the least work (no repeated forward transforms); splitting by
// output column doubles the number of workgroups, which pays while the batch cannot fill the CUs.
int pick_nco(const fheram_ctx* c, int gx, int gy) {
    if (c->nco != 0) return c->nco;
    return ((long)gx * gy * 2 <= c->cus) ? 1 : 2;
}
// gal != 0: automorphism key of Galois element gal, prepared as NTT(phi_gal(K)) (see k_prepare)
void launch_prepare(fheram_ctx* c, const int32_t* in, double* out, int npoly, int64_t gal = 0) {
    ProfScope ps(c, "prepare", npoly);
    const int ginv = gal == 0 ? 0 : galois_inv_mod(galois_mod(gal));
    hipLaunchKernelGGL(k_prepare, dim3(npoly), dim3(T), LDS_BYTES, c->cur, in, out, c->d_tw, c->ninv, ginv);
}
// res = a (x) ggsw over a (gx, gy) grid of ciphertexts; res must not alias a
// Limb-parallel path: 2*SK workgroups per ciphertext + a normalisation pass, chosen while even the
// column split leaves most CUs idle.
double* big_of(const fheram_ctx* c) { return c->cur == c->stream2 ? c->d_big2 : c->d_big; }
bool use_limb_split(const fheram_ctx* c, int gx, int gy, int sk) {
    return c->limb_split && (long)gx * gy <= LIMB_SPLIT_MAX && (long)gx * gy * 2 * sk <= c->cus;
}
void launch_ep(fheram_ctx* c, GlweRef a, GlweRef res, const double* ggsw, int gx, int gy) {
    if (gx <= 0 || gy <= 0) return;
    ProfScope ps(c, "ext_product", (uint64_t)gx * gy);
    if (use_limb_split(c, gx, gy, 4)) {
        hipLaunchKernelGGL((k_ext_product<3, 4, 1, 1>), dim3(gx, gy, 8), dim3(T), LDS_BYTES, c->cur, a, res, ggsw, c->d_tw, big_of(c));
        hipLaunchKernelGGL((k_ext_product<3, 4, 1, 2>), dim3(gx, gy, 2), dim3(T), 0, c->cur, a, res, ggsw, c->d_tw, big_of(c));
        return;
    }
    if (pick_nco(c, gx, gy) == 1) hipLaunchKernelGGL((k_ext_product<3, 4, 1>), dim3(gx, gy, 2), dim3(T), LDS_BYTES, c->cur, a, res, ggsw, c->d_tw, big_of(c));
    else hipLaunchKernelGGL((k_ext_product<3, 4, 2>), dim3(gx, gy, 1), dim3(T), LDS_BYTES, c->cur, a, res, ggsw, c->d_tw, big_of(c));
}
template <int MODE, int SX, int SK, int SO>
void launch_ks(fheram_ctx* c, const KsArgs& ka, int gx, int gy) {
    if (gx <= 0 || gy <= 0) return;
    ProfScope ps(c, "keyswitch", (uint64_t)gx * gy);
    if (use_limb_split(c, gx, gy, SK)) {
        KsArgs kb = ka;
        kb.big = big_of(c);
        hipLaunchKernelGGL((k_keyswitch<MODE, SX, SK, SO, 1, 1>), dim3(gx, gy, 2 * SK), dim3(T), LDS_BYTES, c->cur, kb);
        hipLaunchKernelGGL((k_keyswitch_norm<MODE, SX, SK, SO>), dim3(gx, gy, 2 * (N / 256)), dim3(256), 0, c->cur, kb);
        return;
    }
    if (pick_nco(c, gx, gy) == 1) hipLaunchKernelGGL((k_keyswitch<MODE, SX, SK, SO, 1>), dim3(gx, gy, 2), dim3(T), LDS_BYTES, c->cur, ka);
    else hipLaunchKernelGGL((k_keyswitch<MODE, SX, SK, SO, 2>), dim3(gx, gy, 1), dim3(T), LDS_BYTES, c->cur, ka);
}
void launch_copy(fheram_ctx* c, GlweRef src, GlweRef dst, int gx, int gy) {
    if (gx <= 0 || gy <= 0) return;
    ProfScope ps(c, "elementwise", (uint64_t)gx * gy);
    hipLaunchKernelGGL((k_copy<3>), dim3(gx, gy, EW_SLICES), dim3(256), 0, c->cur, src, dst);
}
KsArgs ks_args(fheram_ctx* c, GlweRef a, GlweRef b, GlweRef out, const double* key, int64_t gal, int t = 0, int rot_mul = 0, int rot_base = 0) {
    KsArgs ka;
    ka.a = a; ka.b = b; ka.out = out; ka.key = key; ka.tw = c->d_tw;
    ka.g = galois_mod(gal); ka.ginv = galois_inv_mod(ka.g); ka.t = t; ka.rot_mul = rot_mul; ka.rot_base = rot_base; ka.big = c->d_big;
    return ka;
}
const double* trace_key(fheram_ctx* c, int i) { return c->d_atk + (size_t)i * fheram_ctx::ATK; }
bool same(const GlweRef& a, const GlweRef& b) { return a.p == b.p; }

// Runs n dependent out-of-place steps src -> ... -> dst, alternating between dst and tmp so that
// the last step lands in dst.  step(i, in, out) launches step i.  dst may be src.
template <typename F>
void run_chain(fheram_ctx* c, int n, GlweRef src, GlweRef dst, GlweRef tmp, int gx, int gy, F&& step) {
    if (n <= 0) { if (!same(src, dst)) launch_copy(c, src, dst, gx, gy); return; }
    if (same(src, dst) && (n % 2 == 1)) {   // the first step would write what it reads: finish in tmp, copy back
        run_chain(c, n, src, tmp, dst, gx, gy, step);
        launch_copy(c, tmp, dst, gx, gy);
        return;
    }
    GlweRef cur = src;
    for (int i = 0; i < n; i++) {
        GlweRef out = ((n - 1 - i) % 2 == 0) ? dst : tmp;
        step(i, cur, out);
        cur = out;
    }
}
// CoordinatePrepared::product / product_inplace (coordinate_prepared.rs:147-177): d external products.
void ep_chain(fheram_ctx* c, GlweRef src, GlweRef dst, GlweRef tmp, const double* prep, int d, int gx, int gy) {
    run_chain(c, d, src, dst, tmp, gx, gy, [&](int i, GlweRef in, GlweRef out) { launch_ep(c, in, out, prep + (size_t)i * fheram_ctx::GGSW, gx, gy); });
}
// GLWE::trace(start, end): step i = rsh(1) then a += phi_{g_i}(KS(a)).
// The first step may read its input rotated by X^-(x*rot_mul) (write path, ram.rs:621,629).
void trace_steps(fheram_ctx* c, GlweRef src, GlweRef dst, GlweRef tmp, int start, int end, int gx, int gy, int rot_mul = 0, int rot_base = 0) {
    run_chain(c, end - start, src, dst, tmp, gx, gy, [&](int i, GlweRef in, GlweRef out) {
        KsArgs ka = ks_args(c, in, in, out, trace_key(c, start + i), c->gal[start + i], 0, i == 0 ? rot_mul : 0, i == 0 ? rot_base : 0);
        launch_ks<KS_TRACE, 3, 4, 3>(c, ka, gx, gy);
    });
}
// GLWEPacker (ram.rs:425-448), level-synchronous, over `count` leaves per y at
// src(x, y); A and B are ping-pong arenas with the same strides (src may be A).
//   n_alone    : packer levels 0..n_alone-1 in which every leaf is alone (a <- rsh(a); a <- a + phi(a))
//   first_pair : packer level of the first pairing step; level first_pair + m joins x with x + count/2^(m+1)
// Whole RAM: n_alone = first_pair = log N - ceil(log2 rows).  Row-sharded RAM: the shards run the
// levels that stay inside one residue class (same n_alone / first_pair, count = local rows) and the
// root finishes with n_alone = 0, first_pair = log N - log2(n_shards) over the gathered partials.
// Returns the arena that holds the packed result at x = 0.
int32_t* pack_levels(fheram_ctx* c, int32_t* src, int32_t* A, int32_t* B, long sy, long sx, size_t count, int gy,
                     int n_alone, int first_pair) {
    const int k = ilog2_ceil(count);
    int32_t* cur = src;
    auto other = [&](int32_t* x) { return x == A ? B : A; };
    for (int i = 0; i < n_alone; i++) {
        int32_t* nxt = other(cur);
        KsArgs ka = ks_args(c, ref(cur, sy, sx), ref(cur, sy, sx), ref(nxt, sy, sx), trace_key(c, i), c->gal[i]);
        launch_ks<KS_TRACE, 3, 4, 3>(c, ka, (int)count, gy);
        cur = nxt;
    }
    size_t live = count;
    for (int m = 0; m < k; m++) {
        const int i = first_pair + m;
        const long h = (long)1 << (k - 1 - m);
        int32_t* nxt = other(cur);
        const long n_pair = std::max<long>(0, std::min<long>(h, (long)live - h));
        const long n_alone_here = std::min<long>(h, (long)live) - n_pair;
        if (n_pair > 0) {
            KsArgs ka = ks_args(c, ref(cur, sy, sx), ref(cur + h * sx, sy, sx), ref(nxt, sy, sx), trace_key(c, i), c->gal[i], N >> (i + 1));
            launch_ks<KS_PAIR, 3, 4, 3>(c, ka, (int)n_pair, gy);
        }
        if (n_alone_here > 0) {
            KsArgs ka = ks_args(c, ref(cur + n_pair * sx, sy, sx), ref(cur, sy, sx), ref(nxt + n_pair * sx, sy, sx), trace_key(c, i), c->gal[i]);
            launch_ks<KS_TRACE, 3, 4, 3>(c, ka, (int)n_alone_here, gy);
        }
        live = std::min<size_t>(live, (size_t)h);
        cur = nxt;
    }
    return cur;
}
// CoordinatePrepared::prepare (coordinate_prepared.rs:104-116) for coordinate `ci` of addr.
int coord_first_digit(const fheram_ctx* c, int ci) { int s = 0; for (int i = 0; i < ci; i++) s += (int)c->base2d[i].size(); return s; }
void coordinate_prepare(fheram_ctx* c, const fheram_addr* addr, int ci) {
    const int d = (int)c->base2d[ci].size();
    launch_prepare(c, addr->d_ggsw + (size_t)coord_first_digit(c, ci) * fheram_ctx::GGSW, c->d_prep, d * (int)(fheram_ctx::GGSW / N));
}
// CoordinatePrepared::prepare_inv (coordinate_prepared.rs:121-142): GGSW(X^i) -> GGSW(X^-i).
void ggsw_inverse(fheram_ctx* c, const int32_t* in, int32_t* tmp, int d) {
    const long g4 = (long)fheram_ctx::GLWE4;
    int32_t* inp = const_cast<int32_t*>(in);
    // GGSW::automorphism, column 0 of every row: res[r][0] = phi_-1(KS(in[r][0]))
    KsArgs ka = ks_args(c, ref(inp, (long)fheram_ctx::GGSW, 2 * g4), ref(inp, 0, 0), ref(tmp, (long)fheram_ctx::GGSW, 2 * g4), c->d_atk_inv, -1);
    launch_ks<KS_AUTO, 4, 5, 4>(c, ka, fheram_ctx::DNUM_CT, d);
    // row expansion with the tensor key: res[r][1] = KS_tsk(res[r][0].mask) + (0, res[r][0].body)
    KsArgs kt = ks_args(c, ref(tmp, (long)fheram_ctx::GGSW, 2 * g4), ref(tmp, 0, 0), ref(tmp + g4, (long)fheram_ctx::GGSW, 2 * g4), c->d_tsk, 1);
    launch_ks<KS_TENSOR, 4, 5, 4>(c, kt, fheram_ctx::DNUM_CT, d);
}
void coordinate_prepare_inv(fheram_ctx* c, const fheram_addr* addr, int ci, int32_t* tmp, double* prep) {
    const int d = (int)c->base2d[ci].size();
    ggsw_inverse(c, addr->d_ggsw + (size_t)coord_first_digit(c, ci) * fheram_ctx::GGSW, tmp, d);
    launch_prepare(c, tmp, prep, d * (int)(fheram_ctx::GGSW / N));
}

// The launch sequence of an op is a pure function of (context, address, op): with FHERAM_GRAPH=1 it is
// captured once per address into a hipGraph and replayed, instead of being re-enqueued kernel by kernel.
template <typename F>
int run_op(fheram_ctx* c, const fheram_addr* addr, int which, F&& enqueue) {
    if (!c->use_graph || c->profile) return enqueue();
    fheram_addr* a = const_cast<fheram_addr*>(addr);
    if (!a->graph[which]) {
        hipGraph_t g = nullptr;
        HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        const int rc = enqueue();
        const hipError_t e = hipStreamEndCapture(c->stream, &g);
        if (rc != FHERAM_OK || e != hipSuccess) {
            if (g) hipGraphDestroy(g);
            return rc != FHERAM_OK ? rc : fail(c, FHERAM_ERR_DEVICE, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
        }
        const hipError_t e2 = hipGraphInstantiate(&a->graph[which], g, nullptr, nullptr, 0);
        hipGraphDestroy(g);
        if (e2 != hipSuccess) { a->graph[which] = nullptr; return fail(c, FHERAM_ERR_DEVICE, std::string("hipGraphInstantiate: ") + hipGetErrorString(e2)); }
    }
    HIPCHK(c, hipGraphLaunch(a->graph[which], c->stream));
    return FHERAM_OK;
}

int check_common(fheram_ctx* c, const fheram_addr* addr) {
    if (!c) return FHERAM_ERR_INVALID_ARG;
    if (!addr || addr->ctx != c) return fail(c, FHERAM_ERR_INVALID_ARG, "address does not belong to this context (layout mismatch, ram.rs:404)");
    if (!c->initialized) return fail(c, FHERAM_ERR_UNINITIALIZED, "unitialized memory: self.data.len()=0");
    if (!c->keys_loaded) return fail(c, FHERAM_ERR_KEYS, "evaluation keys not loaded");
    return FHERAM_OK;
}

// SubRam::read (ram.rs:382-459) / SubRam::read_prepare_write (ram.rs:461-542) for all sub-RAMs at
// once, in two stages so that a row-sharded RAM can exchange between them.
// Stage 1 (every shard): coordinate-0 products on the local rows + the packing levels that stay
// inside the shard.  Leaves one GLWE per sub-RAM in d_part.
int read_local(fheram_ctx* c, const fheram_addr* addr, bool prepare_write) {
    const long G = (long)fheram_ctx::GLWE;
    const long sy = (long)c->rows * G;
    const int ws = c->ws;
    const int R = (int)c->rows;
    GlweRef data = ref(c->d_data, sy, G), A = ref(c->d_scrA, sy, G), B = ref(c->d_scrB, sy, G);
    GlweRef part = ref(c->d_part, G, 0);
    coordinate_prepare(c, addr, 0);                                                   // ram.rs:416-419 / 496-499
    const int d0 = (int)c->base2d[0].size();
    if (c->n2 == 1) {
        GlweRef row0 = ref(c->d_data, sy, 0);
        if (prepare_write) {
            ep_chain(c, row0, row0, ref(c->d_scrA, sy, 0), c->d_prep, d0, 1, ws);     // ram.rs:502-504 (rows == 1)
            launch_copy(c, row0, part, 1, ws);
        } else {
            ep_chain(c, row0, part, ref(c->d_tmp, G, 0), c->d_prep, d0, 1, ws);       // ram.rs:451
        }
        return FHERAM_OK;
    }
    int32_t* leaves;
    if (prepare_write) {
        ep_chain(c, data, data, A, c->d_prep, d0, R, ws);                             // ram.rs:502-504
        leaves = c->d_data;
    } else {
        ep_chain(c, data, A, B, c->d_prep, d0, R, ws);                                // ram.rs:429-434
        leaves = c->d_scrA;
    }
    const int L0 = LOGN - ilog2_ceil(c->rows_glob);
    int32_t* packed = pack_levels(c, leaves, c->d_scrA, c->d_scrB, sy, G, (size_t)R, ws, L0, L0);   // ram.rs:435-448 / 510-521
    launch_copy(c, ref(packed, sy, 0), part, 1, ws);
    return FHERAM_OK;
}
// Stage 2 (root / unsharded): remaining packing levels over the shards' partials (`gathered`:
// [n_shards][ws] GLWEs, or nullptr when the RAM is not sharded and the packed rows are in d_part),
// coordinate-1 products and the final trace.  Result left in d_res.
int read_top(fheram_ctx* c, const fheram_addr* addr, bool prepare_write, int32_t* gathered) {
    const long G = (long)fheram_ctx::GLWE;
    const int ws = c->ws;
    GlweRef res = ref(c->d_res, G, 0), tmp = ref(c->d_tmp, G, 0), tree = ref(c->d_tree, G, 0);
    GlweRef pk = ref(c->d_part, G, 0);
    if (c->n2 == 2) {
        if (gathered) {
            const int kG = ilog2_ceil((size_t)c->n_shards);
            int32_t* a0 = c->d_gat[1];   // gathered partials live in d_gat[0]
            int32_t* a1 = c->d_gat[2];
            int32_t* packed = pack_levels(c, gathered, a0, a1, G, (long)ws * G, (size_t)c->n_shards, ws, 0, LOGN - kG);
            pk = ref(packed, G, 0);
        }
        coordinate_prepare(c, addr, 1);
        const int d1 = (int)c->base2d[1].size();
        if (prepare_write) {
            launch_copy(c, pk, tree, 1, ws);                                          // ram.rs:525-527
            ep_chain(c, tree, tree, tmp, c->d_prep, d1, 1, ws);                       // ram.rs:502-504 (i = 1)
            launch_copy(c, tree, res, 1, ws);                                         // ram.rs:535
        } else {
            ep_chain(c, pk, res, tmp, c->d_prep, d1, 1, ws);                          // ram.rs:454
        }
    } else {
        launch_copy(c, pk, res, 1, ws);                                               // ram.rs:452 / 537
    }
    trace_steps(c, res, res, tmp, 0, LOGN, 1, ws);                                    // ram.rs:457 / 540
    return FHERAM_OK;
}
int read_impl(fheram_ctx* c, const fheram_addr* addr, bool prepare_write) {
    int rc = read_local(c, addr, prepare_write);
    if (rc != FHERAM_OK) return rc;
    rc = read_top(c, addr, prepare_write, nullptr);
    if (rc == FHERAM_OK && prepare_write) c->state = true;                            // ram.rs:533
    return rc;
}

// Ram::write (ram.rs:226-294) in two stages.
// Stage 1 (root / unsharded): write_first_step on the top of the tree and, for n2 == 2, the inverse
// coordinate-1 products: leaves the un-rotated ct_lo of every sub-RAM in d_part.
int write_top(fheram_ctx* c, const fheram_addr* addr) {
    const long G = (long)fheram_ctx::GLWE;
    const long sy = (long)c->rows * G;
    const int ws = c->ws;
    GlweRef wref = ref(c->d_w, G, 0), tmp = ref(c->d_tmp, G, 0), tmp2 = ref(c->d_tmp2, G, 0), tree = ref(c->d_tree, G, 0);
    // write_first_step (ram.rs:544-577): t <- normalize(t - trace(t) + w)
    GlweRef top = (c->n2 != 1) ? tree : ref(c->d_data, sy, 0);
    trace_steps(c, top, tmp, tmp2, 0, LOGN, 1, ws);
    {
        ProfScope ps(c, "elementwise", ws);
        hipLaunchKernelGGL((k_sub_add_norm<3>), dim3(1, ws, EW_SLICES), dim3(256), 0, c->cur, top, tmp, wref, top);
    }
    if (c->n2 == 2) {
        coordinate_prepare_inv(c, addr, 1, c->d_ggsw_tmp, c->d_prep);                 // ram.rs:260-271
        ep_chain(c, tree, tree, tmp, c->d_prep, (int)c->base2d[1].size(), 1, ws);     // ram.rs:610
        launch_copy(c, tree, ref(c->d_part, G, 0), 1, ws);
        {   // ct_lo ends up rotated `rows` times by X^-1 (ram.rs:629)
            ProfScope ps(c, "elementwise", ws);
            hipLaunchKernelGGL((k_rotate<3>), dim3(1, ws, EW_SLICES), dim3(256), 0, c->cur, tree, tmp, -(int)c->rows_glob);
        }
        launch_copy(c, tmp, tree, 1, ws);
    }
    return FHERAM_OK;
}
// Work of a write that does not depend on stage 1: tmp_a = trace(ct_hi) for every local row
// (ram.rs:616) and the inverse of coordinate 0 (ram.rs:278-289).  It is enqueued on the side stream
// so that it fills the CUs the latency-bound stage 1 (a chain of word_size-ciphertext launches)
// leaves idle.
void write_side_begin(fheram_ctx* c, const fheram_addr* addr) {
    const long G = (long)fheram_ctx::GLWE;
    const long sy = (long)c->rows * G;
    hipEventRecord(c->ev_fork, c->stream);            // everything before this write (rows after rpw)
    hipStreamWaitEvent(c->stream2, c->ev_fork, 0);
    c->cur = c->stream2;
    if (c->n2 == 2) trace_steps(c, ref(c->d_data, sy, G), ref(c->d_scrA, sy, G), ref(c->d_scrC, sy, G), 0, LOGN, (int)c->rows, c->ws);
    coordinate_prepare_inv(c, addr, 0, c->d_ggsw_tmp2, c->d_prep2);
    hipEventRecord(c->ev_join, c->stream2);
    c->cur = c->stream;
}
// Stage 2 (every shard): write_mid_step on the local rows given ct_lo (in d_part), then write_last_step.
int write_rows(fheram_ctx* c, const fheram_addr* addr) {
    const long G = (long)fheram_ctx::GLWE;
    const long sy = (long)c->rows * G;
    const int ws = c->ws, R = (int)c->rows;
    GlweRef data = ref(c->d_data, sy, G), A = ref(c->d_scrA, sy, G), B = ref(c->d_scrB, sy, G), D = ref(c->d_scrD, sy, G);
    if (c->n2 == 2)
        trace_steps(c, ref(c->d_part, G, 0), B, D, 0, LOGN, R, ws, c->n_shards, c->shard);     // tmp_a = trace(ct_lo * X^-row)   ram.rs:621,629
    hipStreamWaitEvent(c->stream, c->ev_join, 0);                                              // side stream: trace(ct_hi), inverse coordinate 0
    if (c->n2 == 2) {
        ProfScope ps(c, "elementwise", (uint64_t)R * ws);
        hipLaunchKernelGGL((k_sub_add_norm<3>), dim3(R, ws, EW_SLICES), dim3(256), 0, c->cur, data, A, B, data);   // ram.rs:617,625-626
    }
    ep_chain(c, data, data, A, c->d_prep2, (int)c->base2d[0].size(), R, ws);                   // ram.rs:644-646
    c->state = false;                                                                          // ram.rs:648
    return FHERAM_OK;
}

}  // namespace

extern "C" {

int fheram_params_default(fheram_params* p) {
    if (!p) return FHERAM_ERR_INVALID_ARG;
    std::memset(p, 0, sizeof(*p));
    p->log_n = 12; p->base2k = 17; p->rank = 1;
    p->k_glwe_pt = 3; p->k_glwe_ct = 51; p->k_ggsw_addr = 68; p->k_evk_trace = 68; p->k_evk_ggsw_inv = 85;
    p->word_size = 4; p->n_decomp = 4;
    p->decomp_n[0] = p->decomp_n[1] = p->decomp_n[2] = p->decomp_n[3] = 3;
    p->max_addr = (uint64_t)1 << 14;
    return FHERAM_OK;
}

const char* fheram_last_error(const fheram_ctx* c) { return c ? c->err.c_str() : g_create_err.c_str(); }

int fheram_ctx_create(const fheram_params* p, int device, fheram_ctx** out) {
    return fheram_ctx_create_sharded(p, device, 0, 1, out);
}

int fheram_ctx_create_sharded(const fheram_params* p, int device, int shard, int n_shards, fheram_ctx** out) {
    if (!p || !out) return fail(nullptr, FHERAM_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    if (n_shards < 1 || shard < 0 || shard >= n_shards || (n_shards & (n_shards - 1)) != 0)
        return fail(nullptr, FHERAM_ERR_INVALID_ARG, "n_shards must be a power of two and 0 <= shard < n_shards");
    // The kernels are built for the reference's cryptographic parameters (parameters.rs:11-18).
    if (p->log_n != 12 || p->base2k != 17 || p->rank != 1 || p->k_glwe_ct != 51 || p->k_ggsw_addr != 68 ||
        p->k_evk_trace != 68 || p->k_evk_ggsw_inv != 85)
        return fail(nullptr, FHERAM_ERR_UNSUPPORTED, "kernels are built for LOG_N=12, BASE2K=17, RANK=1, K_CT=51, K_ADDR=68, K_EVK=68/85");
    if (p->word_size == 0 || p->word_size > 64 || p->n_decomp == 0 || p->n_decomp > 16 || p->max_addr < 2)
        return fail(nullptr, FHERAM_ERR_INVALID_ARG, "bad word_size / decomp_n / max_addr");
    unsigned sum = 0;
    for (uint32_t i = 0; i < p->n_decomp; i++) { if (p->decomp_n[i] == 0) return fail(nullptr, FHERAM_ERR_INVALID_ARG, "zero digit width"); sum += p->decomp_n[i]; }
    if (sum != p->log_n) return fail(nullptr, FHERAM_ERR_INVALID_ARG, "DECOMP_N must sum to LOG_N (parameters.rs:168)");
    if (p->max_addr > ((uint64_t)N * N)) return fail(nullptr, FHERAM_ERR_UNSUPPORTED, "max_addr > N^2 is not supported by the reference either (SURVEY.md 3.1)");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, FHERAM_ERR_DEVICE, "no HIP device: the FHE-RAM evaluator has no CPU path");
    if (device < 0 || device >= ndev) return fail(nullptr, FHERAM_ERR_INVALID_ARG, "bad device index");

    fheram_ctx* c = new fheram_ctx();
    c->p = *p; c->device = device; c->ws = (int)p->word_size;
    c->rows_glob = (size_t)((p->max_addr + N - 1) / N);
    c->shard = shard; c->n_shards = n_shards;
    if (n_shards > 1 && ((c->rows_glob & (c->rows_glob - 1)) != 0 || c->rows_glob < (size_t)n_shards)) {
        delete c;
        return fail(nullptr, FHERAM_ERR_INVALID_ARG, "row sharding needs a power-of-two number of rows >= n_shards (else: replicas only)");
    }
    c->rows = c->rows_glob / (size_t)n_shards;
    {   // get_base_2d (base.rs:84-108)
        uint32_t x = (uint32_t)(p->max_addr - 1), bits = 0;
        while (x) { bits++; x >>= 1; }
        while (bits != 0) {
            std::vector<int> v;
            for (uint32_t i = 0; i < p->n_decomp; i++) {
                const uint32_t b = p->decomp_n[i];
                if (b <= bits) { v.push_back((int)b); bits -= b; }
                else { if (bits != 0) { v.push_back((int)bits); bits = 0; } break; }
            }
            c->base2d.push_back(v);
        }
    }
    c->n2 = (int)c->base2d.size();
    for (auto& v : c->base2d) { c->n_digits += (int)v.size(); c->max_digits = std::max(c->max_digits, (int)v.size()); }
    for (int i = 0; i < LOGN; i++) c->gal[i] = galois_element(i);

#define CCHK(call)                                                                  \
    do {                                                                            \
        hipError_t e_ = (call);                                                     \
        if (e_ != hipSuccess) {                                                     \
            g_create_err = std::string(#call) + ": " + hipGetErrorString(e_);       \
            fheram_ctx_destroy(c);                                                  \
            return FHERAM_ERR_DEVICE;                                               \
        }                                                                           \
    } while (0)
    CCHK(hipSetDevice(device));
    CCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    CCHK(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
    c->cur = c->stream;
    CCHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    CCHK(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    CCHK(hipEventCreate(&c->t0));
    CCHK(hipEventCreate(&c->t1));
    {
        const char* ls = getenv("FHERAM_LIMB_SPLIT");
        c->limb_split = (ls && ls[0] == '0') ? 0 : 1;
        const char* gr = getenv("FHERAM_GRAPH");
        c->use_graph = (gr && gr[0] == '1') ? 1 : 0;
        const char* e = getenv("FHERAM_NCO");
        c->nco = (e && e[0] == '2') ? 2 : ((e && e[0] == '1') ? 1 : 0);
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) c->cus = prop.multiProcessorCount;
    }
#define LDSATTR(k) CCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES))
    LDSATTR(k_prepare);
    LDSATTR((&k_ext_product<3, 4, 1>));
    LDSATTR((&k_ext_product<3, 4, 2>));
    LDSATTR((&k_ext_product<3, 4, 1, 1>));
#define LDSATTR_KS(M, SX, SK, SO) LDSATTR((&k_keyswitch<M, SX, SK, SO, 1>)); LDSATTR((&k_keyswitch<M, SX, SK, SO, 2>)); LDSATTR((&k_keyswitch<M, SX, SK, SO, 1, 1>))
    LDSATTR_KS(KS_AUTO, 3, 4, 3);
    LDSATTR_KS(KS_TRACE, 3, 4, 3);
    LDSATTR_KS(KS_PAIR, 3, 4, 3);
    LDSATTR_KS(KS_ADD, 3, 4, 3);
    LDSATTR_KS(KS_SUBNEG, 3, 4, 3);
    LDSATTR_KS(KS_AUTO, 4, 5, 4);
    LDSATTR_KS(KS_TENSOR, 4, 5, 4);
    LDSATTR((&k_encrypt_sk<1, 1>));
    LDSATTR((&k_encrypt_sk<3, 0>)); LDSATTR((&k_encrypt_sk<3, 1>));
    LDSATTR((&k_encrypt_sk<4, 0>)); LDSATTR((&k_encrypt_sk<4, 1>));
    LDSATTR((&k_encrypt_sk<5, 0>)); LDSATTR((&k_encrypt_sk<5, 1>));
#undef LDSATTR_KS
#undef LDSATTR

    std::vector<double> tw = make_twiddles();
    c->ninv = centred(powmod_u((uint64_t)N, P_U64 - 2));
    CCHK(hipMalloc(&c->d_tw, N * sizeof(double)));
    CCHK(hipMemcpy(c->d_tw, tw.data(), N * sizeof(double), hipMemcpyHostToDevice));
    const size_t G = fheram_ctx::GLWE, nrow = (size_t)c->ws * c->rows;
    CCHK(hipMalloc(&c->d_atk, (size_t)LOGN * fheram_ctx::ATK * sizeof(double)));
    CCHK(hipMalloc(&c->d_atk_inv, fheram_ctx::EVK5 * sizeof(double)));
    CCHK(hipMalloc(&c->d_tsk, fheram_ctx::EVK5 * sizeof(double)));
    CCHK(hipMalloc(&c->d_data, nrow * G * sizeof(int32_t)));
    CCHK(hipMalloc(&c->d_scrA, nrow * G * sizeof(int32_t)));
    CCHK(hipMalloc(&c->d_scrB, nrow * G * sizeof(int32_t)));
    CCHK(hipMalloc(&c->d_scrC, nrow * G * sizeof(int32_t)));
    CCHK(hipMalloc(&c->d_scrD, nrow * G * sizeof(int32_t)));
    CCHK(hipMalloc(&c->d_prep2, (size_t)c->max_digits * fheram_ctx::GGSW * sizeof(double)));
    CCHK(hipMalloc(&c->d_ggsw_tmp2, (size_t)c->max_digits * fheram_ctx::GGSW * sizeof(int32_t)));
    CCHK(hipMalloc(&c->d_tmp2, (size_t)c->ws * G * sizeof(int32_t)));
    CCHK(hipMalloc(&c->d_part, (size_t)c->ws * G * sizeof(int32_t)));
    CCHK(hipMalloc(&c->d_big, (size_t)LIMB_SPLIT_MAX * BIG_STRIDE * sizeof(double)));
    CCHK(hipMalloc(&c->d_big2, (size_t)LIMB_SPLIT_MAX * BIG_STRIDE * sizeof(double)));
    if (n_shards > 1) for (int i = 0; i < 3; i++) CCHK(hipMalloc(&c->d_gat[i], (size_t)n_shards * c->ws * G * sizeof(int32_t)));
    CCHK(hipMalloc(&c->d_tree, (size_t)c->ws * G * sizeof(int32_t)));
    CCHK(hipMalloc(&c->d_res, (size_t)c->ws * G * sizeof(int32_t)));
    CCHK(hipMalloc(&c->d_tmp, (size_t)c->ws * G * sizeof(int32_t)));
    CCHK(hipMalloc(&c->d_w, (size_t)c->ws * G * sizeof(int32_t)));
    CCHK(hipMalloc(&c->d_prep, (size_t)c->max_digits * fheram_ctx::GGSW * sizeof(double)));
    CCHK(hipMalloc(&c->d_ggsw_tmp, (size_t)c->max_digits * fheram_ctx::GGSW * sizeof(int32_t)));
    CCHK(hipMemset(c->d_tree, 0, (size_t)c->ws * G * sizeof(int32_t)));
#undef CCHK
    *out = c;
    return FHERAM_OK;
}

void fheram_ctx_destroy(fheram_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->stream2) hipStreamSynchronize(c->stream2);
    prof_collect(c);
    if (c->ev_fork) hipEventDestroy(c->ev_fork);
    if (c->ev_join) hipEventDestroy(c->ev_join);
    if (c->stream2) hipStreamDestroy(c->stream2);
    for (auto e : c->ev_pool) hipEventDestroy(e);
    if (c->t0) hipEventDestroy(c->t0);
    if (c->t1) hipEventDestroy(c->t1);
    void* bufs[] = {c->d_tw, c->d_atk, c->d_atk_inv, c->d_tsk, c->d_data, c->d_scrA, c->d_scrB, c->d_big, c->d_big2, c->d_scrC, c->d_scrD, c->d_prep2, c->d_ggsw_tmp2, c->d_tmp2, c->d_part, c->d_gat[0], c->d_gat[1], c->d_gat[2], c->d_tree, c->d_res, c->d_tmp, c->d_w, c->d_prep, c->d_ggsw_tmp};
    for (void* b : bufs) if (b) hipFree(b);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

size_t fheram_glwe_len(const fheram_ctx*) { return fheram_ctx::GLWE; }
size_t fheram_ggsw_len(const fheram_ctx*) { return fheram_ctx::GGSW; }
size_t fheram_atk_len(const fheram_ctx*) { return fheram_ctx::ATK; }
size_t fheram_evk_inv_len(const fheram_ctx*) { return fheram_ctx::EVK5; }
size_t fheram_rows(const fheram_ctx* c) { return c ? c->rows : 0; }
int fheram_n_digits(const fheram_ctx* c) { return c ? c->n_digits : 0; }
int fheram_n_coordinates(const fheram_ctx* c) { return c ? c->n2 : 0; }

int fheram_keys_load(fheram_ctx* c, const int64_t* gal_els, int n_gal, const int64_t* const* atk_glwe,
                     const int64_t* atk_ggsw_inv, int64_t atk_ggsw_inv_p, const int64_t* tsk) {
    if (!c) return FHERAM_ERR_INVALID_ARG;
    if (!gal_els || !atk_glwe || !atk_ggsw_inv || !tsk) return fail(c, FHERAM_ERR_INVALID_ARG, "null key pointer");
    if (n_gal != LOGN) return fail(c, FHERAM_ERR_KEYS, "expected log2(N) trace keys (keys.rs:39)");
    if (atk_ggsw_inv_p != -1) return fail(c, FHERAM_ERR_KEYS, "auto_key.p() != -1 (coordinate_prepared.rs:134)");
    HIPCHK(c, hipSetDevice(c->device));
    // keys may come in any order (the reference keeps them in a HashMap, keys.rs:28): sort by Galois element
    int order[LOGN];
    for (int i = 0; i < LOGN; i++) {
        order[i] = -1;
        for (int j = 0; j < n_gal; j++) if (gal_els[j] == c->gal[i]) order[i] = j;
        if (order[i] < 0) return fail(c, FHERAM_ERR_KEYS, "missing trace key for a Galois element of GLWE::trace_galois_elements");
    }
    int32_t* d_stage = nullptr;
    const size_t stage_n = std::max(fheram_ctx::ATK, fheram_ctx::EVK5);
    HIPCHK(c, hipMalloc(&d_stage, stage_n * sizeof(int32_t)));
    int rc = FHERAM_OK;
    for (int i = 0; i < LOGN && rc == FHERAM_OK; i++) {
        rc = upload_i64(c, d_stage, atk_glwe[order[i]], fheram_ctx::ATK);
        if (rc == FHERAM_OK) launch_prepare(c, d_stage, c->d_atk + (size_t)i * fheram_ctx::ATK, (int)(fheram_ctx::ATK / N), c->gal[i]);
        hipStreamSynchronize(c->stream);
    }
    if (rc == FHERAM_OK) rc = upload_i64(c, d_stage, atk_ggsw_inv, fheram_ctx::EVK5);
    if (rc == FHERAM_OK) { launch_prepare(c, d_stage, c->d_atk_inv, (int)(fheram_ctx::EVK5 / N), -1); hipStreamSynchronize(c->stream); }
    if (rc == FHERAM_OK) rc = upload_i64(c, d_stage, tsk, fheram_ctx::EVK5);
    if (rc == FHERAM_OK) { launch_prepare(c, d_stage, c->d_tsk, (int)(fheram_ctx::EVK5 / N)); hipStreamSynchronize(c->stream); }
    hipFree(d_stage);
    if (rc != FHERAM_OK) return rc;
    HIPCHK(c, hipGetLastError());
    c->keys_loaded = true;
    return FHERAM_OK;
}

int fheram_ram_upload(fheram_ctx* c, const int64_t* rows) {
    if (!c) return FHERAM_ERR_INVALID_ARG;
    if (!rows) return fail(c, FHERAM_ERR_INVALID_ARG, "null rows");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = upload_i64(c, c->d_data, rows, (size_t)c->ws * c->rows * fheram_ctx::GLWE);
    if (rc != FHERAM_OK) return rc;
    c->initialized = true; c->state = false;
    return FHERAM_OK;
}
int fheram_ram_download(fheram_ctx* c, int64_t* rows) {
    if (!c || !rows) return FHERAM_ERR_INVALID_ARG;
    if (!c->initialized) return fail(c, FHERAM_ERR_UNINITIALIZED, "unitialized memory: self.data.len()=0");
    HIPCHK(c, hipSetDevice(c->device));
    return download_i64(c, rows, c->d_data, (size_t)c->ws * c->rows * fheram_ctx::GLWE);
}
int fheram_ram_tree_download(fheram_ctx* c, int level, int64_t* out) {
    if (!c || !out) return FHERAM_ERR_INVALID_ARG;
    if (level != 0 || c->n2 < 2) return fail(c, FHERAM_ERR_INVALID_ARG, "tree level does not exist (ram.rs:315-324)");
    HIPCHK(c, hipSetDevice(c->device));
    return download_i64(c, out, c->d_tree, (size_t)c->ws * fheram_ctx::GLWE);
}
int fheram_ram_state(const fheram_ctx* c) { return c ? (int)c->state : 0; }

int fheram_address_create(fheram_ctx* c, const int64_t* const* ggsw, int n_ggsw, fheram_addr** out) {
    if (!c || !out) return FHERAM_ERR_INVALID_ARG;
    *out = nullptr;
    if (!ggsw) return fail(c, FHERAM_ERR_INVALID_ARG, "null ggsw");
    if (n_ggsw != c->n_digits) return fail(c, FHERAM_ERR_INVALID_ARG, "address digit count does not match the context's Base2D (ram.rs:404)");
    HIPCHK(c, hipSetDevice(c->device));
    fheram_addr* a = new fheram_addr{c, nullptr, n_ggsw, c->device};
    hipError_t e = hipMalloc(&a->d_ggsw, (size_t)n_ggsw * fheram_ctx::GGSW * sizeof(int32_t));
    if (e != hipSuccess) { delete a; return fail(c, FHERAM_ERR_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(e)); }
    for (int i = 0; i < n_ggsw; i++) {
        if (!ggsw[i]) { fheram_address_destroy(a); return fail(c, FHERAM_ERR_INVALID_ARG, "null ggsw digit"); }
        int rc = upload_i64(c, a->d_ggsw + (size_t)i * fheram_ctx::GGSW, ggsw[i], fheram_ctx::GGSW);
        if (rc != FHERAM_OK) { fheram_address_destroy(a); return rc; }
    }
    *out = a;
    return FHERAM_OK;
}
void fheram_address_destroy(fheram_addr* a) {
    if (!a) return;
    hipSetDevice(a->device);
    for (auto& g : a->graph) if (g) { hipGraphExecDestroy(g); g = nullptr; }
    if (a->d_ggsw) hipFree(a->d_ggsw);   // hipFree waits for outstanding work on the buffer
    delete a;
}

int fheram_result_download(fheram_ctx* c, int64_t* out) {
    if (!c || !out) return FHERAM_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    return download_i64(c, out, c->d_res, (size_t)c->ws * fheram_ctx::GLWE);
}
int fheram_sync(fheram_ctx* c) {
    if (!c) return FHERAM_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    return FHERAM_OK;
}

int fheram_read(fheram_ctx* c, const fheram_addr* addr, int64_t* out) {
    int rc = check_common(c, addr);
    if (rc != FHERAM_OK) return rc;
    if (c->n_shards != 1) return fail(c, FHERAM_ERR_INVALID_ARG, "row-sharded context: use fheram_read_partial / fheram_read_finish");
    if (c->state) return fail(c, FHERAM_ERR_STATE, "invalid call to Memory.read: internal state is true -> requires calling Memory.write");
    HIPCHK(c, hipSetDevice(c->device));
    rc = run_op(c, addr, 0, [&] { return read_impl(c, addr, false); });
    if (rc != FHERAM_OK) return rc;
    HIPCHK(c, hipGetLastError());
    return out ? fheram_result_download(c, out) : FHERAM_OK;
}
int fheram_read_prepare_write(fheram_ctx* c, const fheram_addr* addr, int64_t* out) {
    int rc = check_common(c, addr);
    if (rc != FHERAM_OK) return rc;
    if (c->n_shards != 1) return fail(c, FHERAM_ERR_INVALID_ARG, "row-sharded context: use fheram_read_partial / fheram_read_finish");
    if (c->state) return fail(c, FHERAM_ERR_STATE, "invalid call to Memory.read: internal state is true -> requires calling Memory.write");
    HIPCHK(c, hipSetDevice(c->device));
    rc = run_op(c, addr, 1, [&] { return read_impl(c, addr, true); });
    if (rc != FHERAM_OK) return rc;
    c->state = true;                                                                  // ram.rs:533
    HIPCHK(c, hipGetLastError());
    return out ? fheram_result_download(c, out) : FHERAM_OK;
}
int fheram_word_stage(fheram_ctx* c, const int64_t* w, int n_w) {
    if (!c || !w) return FHERAM_ERR_INVALID_ARG;
    if (n_w != c->ws) return fail(c, FHERAM_ERR_INVALID_ARG, "w.len() != subrams.len() (ram.rs:243)");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = upload_i64(c, c->d_w, w, (size_t)c->ws * fheram_ctx::GLWE);
    if (rc == FHERAM_OK) c->words_staged = true;
    return rc;
}
// Ram::write, ram.rs:226-294
int fheram_write(fheram_ctx* c, const int64_t* w, int n_w, const fheram_addr* addr) {
    int rc = check_common(c, addr);
    if (rc != FHERAM_OK) return rc;
    if (n_w != c->ws) return fail(c, FHERAM_ERR_INVALID_ARG, "w.len() != subrams.len() (ram.rs:243)");
    if (!c->state) return fail(c, FHERAM_ERR_STATE, "invalid call to Memory.write: internal state is false -> requires calling Memory.read_prepare_write");
    HIPCHK(c, hipSetDevice(c->device));
    if (w) { rc = fheram_word_stage(c, w, n_w); if (rc != FHERAM_OK) return rc; }
    else if (!c->words_staged) return fail(c, FHERAM_ERR_INVALID_ARG, "w == NULL and no staged words");
    if (c->n_shards != 1) return fail(c, FHERAM_ERR_INVALID_ARG, "row-sharded context: use fheram_write_root / fheram_write_shard");
    rc = run_op(c, addr, 2, [&] {
        write_side_begin(c, addr);
        int r2 = write_top(c, addr);
        return r2 == FHERAM_OK ? write_rows(c, addr) : r2;
    });
    if (rc != FHERAM_OK) return rc;
    c->state = false;
    HIPCHK(c, hipGetLastError());
    return FHERAM_OK;
}

// ---- row-sharded RAM ------------------------------------------------------------
namespace {
// copies ws GLWEs between the context and a caller buffer: device int32 [ws][GLWE] or host int64
int export_glwes(fheram_ctx* c, const int32_t* src, void* dst, int on_device, size_t n_glwe) {
    const size_t n = n_glwe * fheram_ctx::GLWE;
    if (on_device) { HIPCHK(c, hipMemcpyAsync(dst, src, n * 4, hipMemcpyDeviceToDevice, c->stream)); return FHERAM_OK; }
    return download_i64(c, (int64_t*)dst, src, n);
}
int import_glwes(fheram_ctx* c, int32_t* dst, const void* src, int on_device, size_t n_glwe) {
    const size_t n = n_glwe * fheram_ctx::GLWE;
    if (on_device) { HIPCHK(c, hipMemcpyAsync(dst, src, n * 4, hipMemcpyDeviceToDevice, c->stream)); return FHERAM_OK; }
    return upload_i64(c, dst, (const int64_t*)src, n);
}
}  // namespace

int fheram_shard_info(const fheram_ctx* c, int* shard, int* n_shards, size_t* local_rows) {
    if (!c) return FHERAM_ERR_INVALID_ARG;
    if (shard) *shard = c->shard;
    if (n_shards) *n_shards = c->n_shards;
    if (local_rows) *local_rows = c->rows;
    return FHERAM_OK;
}
int fheram_read_partial(fheram_ctx* c, const fheram_addr* addr, int prepare_write, void* out, int out_on_device) {
    int rc = check_common(c, addr);
    if (rc != FHERAM_OK) return rc;
    if (!out) return fail(c, FHERAM_ERR_INVALID_ARG, "null output");
    if (c->state) return fail(c, FHERAM_ERR_STATE, "invalid call to Memory.read: internal state is true -> requires calling Memory.write");
    HIPCHK(c, hipSetDevice(c->device));
    rc = read_local(c, addr, prepare_write != 0);
    if (rc != FHERAM_OK) return rc;
    HIPCHK(c, hipGetLastError());
    if (prepare_write) c->state = true;
    rc = export_glwes(c, c->d_part, out, out_on_device, (size_t)c->ws);
    if (rc == FHERAM_OK && out_on_device) HIPCHK(c, hipStreamSynchronize(c->stream));   // the caller's collective runs on another stream
    return rc;
}
int fheram_read_finish(fheram_ctx* c, const fheram_addr* addr, int prepare_write, const void* partials, int partials_on_device, int64_t* out) {
    int rc = check_common(c, addr);
    if (rc != FHERAM_OK) return rc;
    if (!partials) return fail(c, FHERAM_ERR_INVALID_ARG, "null partials");
    HIPCHK(c, hipSetDevice(c->device));
    int32_t* gathered = nullptr;
    if (c->n_shards > 1) {
        rc = import_glwes(c, c->d_gat[0], partials, partials_on_device, (size_t)c->n_shards * c->ws);
        gathered = c->d_gat[0];
    } else {
        rc = import_glwes(c, c->d_part, partials, partials_on_device, (size_t)c->ws);
    }
    if (rc != FHERAM_OK) return rc;
    rc = read_top(c, addr, prepare_write != 0, gathered);
    if (rc != FHERAM_OK) return rc;
    HIPCHK(c, hipGetLastError());
    return out ? fheram_result_download(c, out) : fheram_sync(c);
}
int fheram_write_root(fheram_ctx* c, const int64_t* w, int n_w, const fheram_addr* addr, void* ct_lo_out, int out_on_device) {
    int rc = check_common(c, addr);
    if (rc != FHERAM_OK) return rc;
    if (n_w != c->ws) return fail(c, FHERAM_ERR_INVALID_ARG, "w.len() != subrams.len() (ram.rs:243)");
    if (c->n2 != 2) return fail(c, FHERAM_ERR_INVALID_ARG, "a row-sharded RAM has two coordinates");
    if (!c->state) return fail(c, FHERAM_ERR_STATE, "invalid call to Memory.write: internal state is false -> requires calling Memory.read_prepare_write");
    if (!ct_lo_out) return fail(c, FHERAM_ERR_INVALID_ARG, "null output");
    HIPCHK(c, hipSetDevice(c->device));
    if (w) { rc = fheram_word_stage(c, w, n_w); if (rc != FHERAM_OK) return rc; }
    else if (!c->words_staged) return fail(c, FHERAM_ERR_INVALID_ARG, "w == NULL and no staged words");
    rc = write_top(c, addr);
    if (rc != FHERAM_OK) return rc;
    HIPCHK(c, hipGetLastError());
    rc = export_glwes(c, c->d_part, ct_lo_out, out_on_device, (size_t)c->ws);
    if (rc == FHERAM_OK && out_on_device) HIPCHK(c, hipStreamSynchronize(c->stream));
    return rc;
}
int fheram_write_shard(fheram_ctx* c, const fheram_addr* addr, const void* ct_lo, int on_device) {
    int rc = check_common(c, addr);
    if (rc != FHERAM_OK) return rc;
    if (!ct_lo) return fail(c, FHERAM_ERR_INVALID_ARG, "null ct_lo");
    if (!c->state) return fail(c, FHERAM_ERR_STATE, "invalid call to Memory.write: internal state is false -> requires calling Memory.read_prepare_write");
    HIPCHK(c, hipSetDevice(c->device));
    write_side_begin(c, addr);   // (a shard could start this before the broadcast arrives; kept here for a simple contract)
    rc = import_glwes(c, c->d_part, ct_lo, on_device, (size_t)c->ws);
    if (rc != FHERAM_OK) return rc;
    rc = write_rows(c, addr);
    if (rc != FHERAM_OK) return rc;
    HIPCHK(c, hipGetLastError());
    return FHERAM_OK;
}

// ---- Poulpy-level operations ---------------------------------------------------------------
namespace {
struct DevBuf {
    int32_t* p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
};
}  // namespace

int fheram_glwe_external_product(fheram_ctx* c, const int64_t* a, int batch, const int64_t* ggsw, int64_t* res) {
    if (!c || !a || !ggsw || !res || batch <= 0) return fail(c, FHERAM_ERR_INVALID_ARG, "bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t G = fheram_ctx::GLWE;
    DevBuf da, dg;
    HIPCHK(c, hipMalloc(&da.p, (size_t)batch * G * 4));
    HIPCHK(c, hipMalloc(&dg.p, fheram_ctx::GGSW * 4));
    int rc = upload_i64(c, da.p, a, (size_t)batch * G);
    if (rc == FHERAM_OK) rc = upload_i64(c, dg.p, ggsw, fheram_ctx::GGSW);
    if (rc != FHERAM_OK) return rc;
    DevBuf dout;
    HIPCHK(c, hipMalloc(&dout.p, (size_t)batch * G * 4));
    launch_prepare(c, dg.p, c->d_prep, (int)(fheram_ctx::GGSW / N));
    launch_ep(c, ref(da.p, 0, (long)G), ref(dout.p, 0, (long)G), c->d_prep, batch, 1);
    HIPCHK(c, hipGetLastError());
    return download_i64(c, res, dout.p, (size_t)batch * G);
}
int fheram_glwe_automorphism(fheram_ctx* c, int mode, int64_t gal_el, const int64_t* a, int batch, int64_t* res) {
    if (!c || !a || !res || batch <= 0 || mode < 0 || mode > 2) return fail(c, FHERAM_ERR_INVALID_ARG, "bad argument");
    if (!c->keys_loaded) return fail(c, FHERAM_ERR_KEYS, "evaluation keys not loaded");
    int ki = -1;
    for (int i = 0; i < LOGN; i++) if (c->gal[i] == gal_el) ki = i;
    if (ki < 0) return fail(c, FHERAM_ERR_KEYS, "no key for this Galois element");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t G = fheram_ctx::GLWE;
    DevBuf da, dout;
    HIPCHK(c, hipMalloc(&da.p, (size_t)batch * G * 4));
    HIPCHK(c, hipMalloc(&dout.p, (size_t)batch * G * 4));
    int rc = upload_i64(c, da.p, a, (size_t)batch * G);
    if (rc != FHERAM_OK) return rc;
    KsArgs ka = ks_args(c, ref(da.p, 0, (long)G), ref(da.p, 0, 0), ref(dout.p, 0, (long)G), trace_key(c, ki), gal_el);
    if (mode == 0) launch_ks<KS_AUTO, 3, 4, 3>(c, ka, batch, 1);
    else if (mode == 1) launch_ks<KS_ADD, 3, 4, 3>(c, ka, batch, 1);
    else launch_ks<KS_SUBNEG, 3, 4, 3>(c, ka, batch, 1);
    HIPCHK(c, hipGetLastError());
    return download_i64(c, res, dout.p, (size_t)batch * G);
}
int fheram_glwe_trace(fheram_ctx* c, int start, int end, const int64_t* a, int batch, int64_t* res) {
    if (!c || !a || !res || batch <= 0 || start < 0 || end > LOGN || start > end) return fail(c, FHERAM_ERR_INVALID_ARG, "bad argument");
    if (!c->keys_loaded) return fail(c, FHERAM_ERR_KEYS, "evaluation keys not loaded");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t G = fheram_ctx::GLWE;
    DevBuf da, db, dc;
    HIPCHK(c, hipMalloc(&da.p, (size_t)batch * G * 4));
    HIPCHK(c, hipMalloc(&db.p, (size_t)batch * G * 4));
    HIPCHK(c, hipMalloc(&dc.p, (size_t)batch * G * 4));
    int rc = upload_i64(c, da.p, a, (size_t)batch * G);
    if (rc != FHERAM_OK) return rc;
    trace_steps(c, ref(da.p, 0, (long)G), ref(db.p, 0, (long)G), ref(dc.p, 0, (long)G), start, end, batch, 1);
    HIPCHK(c, hipGetLastError());
    return download_i64(c, res, db.p, (size_t)batch * G);
}
int fheram_glwe_pack(fheram_ctx* c, const int64_t* cts, int count, int64_t* out) {
    if (!c || !cts || !out || count <= 0 || count > N) return fail(c, FHERAM_ERR_INVALID_ARG, "bad argument");
    if (!c->keys_loaded) return fail(c, FHERAM_ERR_KEYS, "evaluation keys not loaded");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t G = fheram_ctx::GLWE;
    DevBuf src, A, B;
    HIPCHK(c, hipMalloc(&src.p, (size_t)count * G * 4));
    HIPCHK(c, hipMalloc(&A.p, (size_t)count * G * 4));
    HIPCHK(c, hipMalloc(&B.p, (size_t)count * G * 4));
    int rc = upload_i64(c, src.p, cts, (size_t)count * G);
    if (rc != FHERAM_OK) return rc;
    const int L0p = LOGN - ilog2_ceil((size_t)count);
    int32_t* packed = pack_levels(c, src.p, A.p, B.p, 0, (long)G, (size_t)count, 1, L0p, L0p);
    HIPCHK(c, hipGetLastError());
    return download_i64(c, out, packed, G);
}
int fheram_ggsw_automorphism_inv(fheram_ctx* c, const int64_t* ggsw_in, int64_t* ggsw_out) {
    if (!c || !ggsw_in || !ggsw_out) return fail(c, FHERAM_ERR_INVALID_ARG, "bad argument");
    if (!c->keys_loaded) return fail(c, FHERAM_ERR_KEYS, "evaluation keys not loaded");
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf din;
    HIPCHK(c, hipMalloc(&din.p, fheram_ctx::GGSW * 4));
    int rc = upload_i64(c, din.p, ggsw_in, fheram_ctx::GGSW);
    if (rc != FHERAM_OK) return rc;
    ggsw_inverse(c, din.p, c->d_ggsw_tmp, 1);
    HIPCHK(c, hipGetLastError());
    return download_i64(c, ggsw_out, c->d_ggsw_tmp, fheram_ctx::GGSW);
}

// ---- measurement hooks ------------------------------------------------------------------------
int fheram_timer_begin(fheram_ctx* c) {
    if (!c) return FHERAM_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipEventRecord(c->t0, c->stream));
    return FHERAM_OK;
}
int fheram_timer_end(fheram_ctx* c, float* ms) {
    if (!c || !ms) return FHERAM_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipEventRecord(c->t1, c->stream));
    HIPCHK(c, hipEventSynchronize(c->t1));
    HIPCHK(c, hipEventElapsedTime(ms, c->t0, c->t1));
    return FHERAM_OK;
}
int fheram_profile_enable(fheram_ctx* c, int on) {
    if (!c) return FHERAM_ERR_INVALID_ARG;
    c->profile = on != 0;
    return FHERAM_OK;
}
int fheram_profile_reset(fheram_ctx* c) {
    if (!c) return FHERAM_ERR_INVALID_ARG;
    hipSetDevice(c->device);
    prof_collect(c);
    c->prof.clear();
    return FHERAM_OK;
}
int fheram_profile_get(fheram_ctx* c, const char* cls, uint64_t* launches, uint64_t* blocks, double* total_ms) {
    if (!c || !cls) return FHERAM_ERR_INVALID_ARG;
    hipSetDevice(c->device);
    prof_collect(c);
    auto it = c->prof.find(cls);
    if (launches) *launches = it == c->prof.end() ? 0 : it->second.launches;
    if (blocks) *blocks = it == c->prof.end() ? 0 : it->second.blocks;
    if (total_ms) *total_ms = it == c->prof.end() ? 0.0 : it->second.ms;
    return FHERAM_OK;
}
#ifdef FK_STAMP
int fheram_debug_ntt_probe(fheram_ctx* c, int blocks) {
    hipSetDevice(c->device);
    double* sink = nullptr;
    hipMalloc(&sink, (size_t)blocks * T * sizeof(double));
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_ntt_probe), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k_ntt_probe, dim3(blocks), dim3(T), LDS_BYTES, c->stream, c->d_tw, sink);
    hipStreamSynchronize(c->stream);
    hipFree(sink);
    return 0;
}
// diagnostic build only: not part of include/fheram.h
int fheram_debug_stamps(fheram_ctx* c, unsigned long long* out, int n) {
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * (n > 64 ? 64 : n)) == hipSuccess ? 0 : 7;
}
#endif
int fheram_device_info(const fheram_ctx* c, char* name, size_t name_len, int* cus) {
    if (!c) return FHERAM_ERR_INVALID_ARG;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c->device) != hipSuccess) return FHERAM_ERR_DEVICE;
    if (name && name_len) { std::snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName); }
    if (cus) *cus = prop.multiProcessorCount;
    return FHERAM_OK;
}

}  // extern "C"

// ---- setup side on the device ------------------------------------------
// Sampling is the caller's: masks and noise arrive as integers; what runs here is the arithmetic.
namespace {

constexpr int64_t NOISE_LIM = (int64_t)1 << 30;
int noise_limb(int k) { return (k + BASE2K - 1) / BASE2K - 1; }

template <int S, int DEC>
void launch_enc(fheram_ctx* c, int32_t* cts, const double* s_hat, const int32_t* pt1, int n) {
    ProfScope ps(c, "encrypt", (uint64_t)n);
    hipLaunchKernelGGL((k_encrypt_sk<S, DEC>), dim3(n), dim3(T), LDS_BYTES, c->cur, cts, s_hat, c->d_tw, pt1);
}
bool launch_enc_dyn(fheram_ctx* c, int S, int dec, int32_t* cts, const double* s_hat, const int32_t* pt1, int n) {
    if (n <= 0) return true;
    switch (S * 2 + dec) {
        case 3: launch_enc<1, 1>(c, cts, s_hat, pt1, n); return true;
        case 6: launch_enc<3, 0>(c, cts, s_hat, pt1, n); return true;
        case 7: launch_enc<3, 1>(c, cts, s_hat, pt1, n); return true;
        case 8: launch_enc<4, 0>(c, cts, s_hat, pt1, n); return true;
        case 9: launch_enc<4, 1>(c, cts, s_hat, pt1, n); return true;
        case 10: launch_enc<5, 0>(c, cts, s_hat, pt1, n); return true;
        case 11: launch_enc<5, 1>(c, cts, s_hat, pt1, n); return true;
    }
    return false;
}
// Adds the caller's draws for one GLWE to a staged pre-ciphertext (int32 [S][2][N], plaintext already in
// place): mask limbs into column 1, the noise polynomial onto its limb of column 0.
bool stage_random(int32_t* pre, int S, int k, const int64_t* mask, const int64_t* noise) {
    int64_t bad = 0;
    for (int j = 0; j < S; j++) {
        int32_t* m = pre + (size_t)(j * 2 + 1) * N;
        const int64_t* src = mask + (size_t)j * N;
        for (int i = 0; i < N; i++) { const int64_t v = src[i]; bad |= (v > 65535) | (v < -65536); m[i] = (int32_t)v; }
    }
    int32_t* b = pre + (size_t)(noise_limb(k) * 2) * N;
    for (int i = 0; i < N; i++) { const int64_t e = noise[i]; bad |= (e >= NOISE_LIM) | (e <= -NOISE_LIM); b[i] += (int32_t)e; }
    return bad == 0;
}
int check_setup_args(fheram_ctx* c, const fheram_secret* sk, int S, int k) {
    if (!sk || sk->ctx != c) return fail(c, FHERAM_ERR_INVALID_ARG, "secret belongs to another context");
    if (k <= 0 || noise_limb(k) >= S) return fail(c, FHERAM_ERR_INVALID_ARG, "precision k does not fit the ciphertext's limbs");
    return FHERAM_OK;
}
// H2D of staged pre-ciphertexts (+ optional mask-column plaintexts), in-place encryption at d_dst.
int encrypt_staged(fheram_ctx* c, const double* s_hat, int32_t* d_dst, const std::vector<int32_t>& pre,
                   const std::vector<int32_t>* pt1, int n, int S) {
    int32_t* d_pt1 = nullptr;
    HIPCHK(c, hipMemcpyAsync(d_dst, pre.data(), (size_t)n * S * 2 * N * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    if (pt1) {
        HIPCHK(c, hipMalloc(&d_pt1, (size_t)n * S * N * sizeof(int32_t)));
        hipError_t e = hipMemcpyAsync(d_pt1, pt1->data(), (size_t)n * S * N * sizeof(int32_t), hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) { hipFree(d_pt1); return fail(c, FHERAM_ERR_DEVICE, std::string("hipMemcpyAsync: ") + hipGetErrorString(e)); }
    }
    c->cur = c->stream;
    const bool ok = launch_enc_dyn(c, S, 0, d_dst, s_hat, d_pt1, n);
    hipError_t e = hipStreamSynchronize(c->stream);
    if (d_pt1) hipFree(d_pt1);
    if (!ok) return fail(c, FHERAM_ERR_UNSUPPORTED, "ciphertext size must be 3, 4 or 5 limbs");
    if (e != hipSuccess) return fail(c, FHERAM_ERR_DEVICE, std::string("k_encrypt_sk: ") + hipGetErrorString(e));
    HIPCHK(c, hipGetLastError());
    return FHERAM_OK;
}
// phi_g on a small polynomial: res(X) = a(X^g)
void host_automorphism(int64_t g, const int32_t* a, int32_t* res) {
    const int64_t m = 2 * N, gg = ((g % m) + m) % m;
    for (int i = 0; i < N; i++) {
        const int64_t j = (int64_t)i * gg % m;
        if (j >= N) res[j - N] = -a[i]; else res[j] = a[i];
    }
}
// encode_vec_i64 at precision k_pt on ceil(k_pt/base2k) limbs: value * 2^-k_pt on the torus
void encode_value(int64_t v, int k_pt, int size_pt, int32_t* limbs /*[size_pt]*/) {
    int64_t x = (int64_t)((uint64_t)v << (size_pt * BASE2K - k_pt)), carry = 0;
    for (int j = size_pt - 1; j >= 0; j--) {
        const int64_t t = (j == size_pt - 1 ? x : 0) + carry;
        const int64_t d = (int64_t)((uint64_t)t << (64 - BASE2K)) >> (64 - BASE2K);
        limbs[j] = (int32_t)d;
        carry = (t - d) >> BASE2K;
    }
}

}  // namespace

extern "C" {

int fheram_secret_create(fheram_ctx* c, const int64_t* sk, fheram_secret** out) {
    if (!c || !out) return FHERAM_ERR_INVALID_ARG;
    *out = nullptr;
    if (!sk) return fail(c, FHERAM_ERR_INVALID_ARG, "null secret");
    HIPCHK(c, hipSetDevice(c->device));
    fheram_secret* s = new fheram_secret{c, c->device, std::vector<int32_t>(N), nullptr};
    for (int i = 0; i < N; i++) {
        if (sk[i] < -1 || sk[i] > 1) { delete s; return fail(c, FHERAM_ERR_RANGE, "secret coefficients must be in {-1, 0, 1}"); }
        s->sk[i] = (int32_t)sk[i];
    }
    int32_t* d_in = nullptr;
    hipError_t e = hipMalloc(&d_in, N * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc(&s->d_hat, N * sizeof(double));
    if (e == hipSuccess) e = hipMemcpyAsync(d_in, s->sk.data(), N * sizeof(int32_t), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) { c->cur = c->stream; launch_prepare(c, d_in, s->d_hat, 1); e = hipStreamSynchronize(c->stream); }
    if (d_in) hipFree(d_in);
    if (e != hipSuccess) { fheram_secret_destroy(s); return fail(c, FHERAM_ERR_DEVICE, std::string("secret prepare: ") + hipGetErrorString(e)); }
    *out = s;
    return FHERAM_OK;
}
void fheram_secret_destroy(fheram_secret* s) {
    if (!s) return;
    hipSetDevice(s->device);
    if (s->d_hat) hipFree(s->d_hat);
    std::fill(s->sk.begin(), s->sk.end(), 0);
    delete s;
}

int fheram_glwe_encrypt_sk(fheram_ctx* c, const fheram_secret* sk, int n_glwe, int size, int k, const int64_t* pt,
                           int pt_size, int pt_col, const int64_t* mask, const int64_t* noise, int64_t* out) {
    if (!c) return FHERAM_ERR_INVALID_ARG;
    if (!mask || !noise || !out || n_glwe <= 0) return fail(c, FHERAM_ERR_INVALID_ARG, "null argument / empty batch");
    if (size < 3 || size > 5) return fail(c, FHERAM_ERR_UNSUPPORTED, "ciphertext size must be 3, 4 or 5 limbs");
    if (pt && (pt_size <= 0 || (pt_col != 0 && pt_col != 1))) return fail(c, FHERAM_ERR_INVALID_ARG, "bad plaintext shape");
    int rc = check_setup_args(c, sk, size, k);
    if (rc != FHERAM_OK) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t glen = (size_t)size * 2 * N;
    std::vector<int32_t> pre((size_t)n_glwe * glen, 0), pt1;
    if (pt && pt_col == 1) pt1.assign((size_t)n_glwe * size * N, 0);
    const int np = pt ? std::min(pt_size, size) : 0;
    for (int g = 0; g < n_glwe; g++) {
        int32_t* pg = pre.data() + (size_t)g * glen;
        for (int j = 0; j < np; j++) {
            const int64_t* src = pt + ((size_t)g * pt_size + j) * N;
            int32_t* dst = pt_col == 0 ? pg + (size_t)(j * 2) * N : pt1.data() + ((size_t)g * size + j) * N;
            for (int i = 0; i < N; i++) {
                if (src[i] > 65536 || src[i] < -65536) return fail(c, FHERAM_ERR_RANGE, "plaintext limb out of the normalised range [-2^16, 2^16]");
                dst[i] = (int32_t)src[i];
            }
        }
        if (!stage_random(pg, size, k, mask + (size_t)g * size * N, noise + (size_t)g * N))
            return fail(c, FHERAM_ERR_RANGE, "mask limb outside [-2^16, 2^16) or |noise| >= 2^30");
    }
    int32_t* d_ct = nullptr;
    HIPCHK(c, hipMalloc(&d_ct, pre.size() * sizeof(int32_t)));
    rc = encrypt_staged(c, sk->d_hat, d_ct, pre, pt1.empty() ? nullptr : &pt1, n_glwe, size);
    if (rc == FHERAM_OK) rc = download_i64(c, out, d_ct, pre.size());
    hipFree(d_ct);
    return rc;
}

int fheram_glwe_decrypt(fheram_ctx* c, const fheram_secret* sk, int n_glwe, int size, const int64_t* ct, int64_t* pt) {
    if (!c) return FHERAM_ERR_INVALID_ARG;
    if (!ct || !pt || n_glwe <= 0) return fail(c, FHERAM_ERR_INVALID_ARG, "null argument / empty batch");
    if (size < 3 || size > 5) return fail(c, FHERAM_ERR_UNSUPPORTED, "ciphertext size must be 3, 4 or 5 limbs");
    if (!sk || sk->ctx != c) return fail(c, FHERAM_ERR_INVALID_ARG, "secret belongs to another context");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t glen = (size_t)size * 2 * N, total = (size_t)n_glwe * glen;
    int32_t* d_ct = nullptr;
    HIPCHK(c, hipMalloc(&d_ct, total * sizeof(int32_t)));
    int rc = upload_i64(c, d_ct, ct, total);
    if (rc == FHERAM_OK) {
        c->cur = c->stream;
        launch_enc_dyn(c, size, 1, d_ct, sk->d_hat, nullptr, n_glwe);
        c->h_i32.resize(total);
        hipError_t e = hipMemcpyAsync(c->h_i32.data(), d_ct, total * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = fail(c, FHERAM_ERR_DEVICE, std::string("decrypt: ") + hipGetErrorString(e));
        else
            for (int g = 0; g < n_glwe; g++)
                for (int j = 0; j < size; j++)
                    widen(c->h_i32.data() + (size_t)g * glen + (size_t)(j * 2) * N, pt + ((size_t)g * size + j) * N, N);
    }
    hipFree(d_ct);
    return rc;
}

int fheram_ram_encrypt_sk(fheram_ctx* c, const fheram_secret* sk, const uint8_t* data, size_t data_len,
                          const int64_t* mask, const int64_t* noise) {
    if (!c) return FHERAM_ERR_INVALID_ARG;
    if (!data || !mask || !noise) return fail(c, FHERAM_ERR_INVALID_ARG, "null argument");
    const size_t ws = (size_t)c->ws;
    if (data_len % ws != 0) return fail(c, FHERAM_ERR_INVALID_ARG, "invalid data: data.len()%ram_chunks != 0");            // ram.rs:144-148
    if (data_len / ws != c->p.max_addr) return fail(c, FHERAM_ERR_INVALID_ARG, "invalid data: data.len()/ram_chunks != max_addr");   // ram.rs:150-155
    const int k = (int)c->p.k_glwe_ct, S = fheram_ctx::S_CT;
    int rc = check_setup_args(c, sk, S, k);
    if (rc != FHERAM_OK) return rc;
    const int k_pt = (int)c->p.k_glwe_pt, size_pt = (k_pt + BASE2K - 1) / BASE2K;
    if (k_pt <= 0 || k_pt > 8 + BASE2K || size_pt > S) return fail(c, FHERAM_ERR_UNSUPPORTED, "k_glwe_pt out of range");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t G = fheram_ctx::GLWE, max_addr = c->p.max_addr;
    const size_t CH = 256;   // rows staged per launch
    std::vector<int32_t> pre(std::min(CH, c->rows) * G);
    int32_t limbs[4];
    for (size_t w = 0; w < ws; w++)                         // de-interleave by word, ram.rs:161-164
        for (size_t x0 = 0; x0 < c->rows; x0 += CH) {       // chunks of N addresses per row, ram.rs:358-379
            const size_t nx = std::min(CH, c->rows - x0);
            std::fill(pre.begin(), pre.begin() + nx * G, 0);
            for (size_t x = 0; x < nx; x++) {
                const size_t r = (size_t)c->shard + (x0 + x) * (size_t)c->n_shards;   // global row
                int32_t* pg = pre.data() + x * G;
                for (size_t q = 0; q < (size_t)N; q++) {
                    const size_t a = r * N + q;
                    if (a >= max_addr) break;                                          // zero padding, ram.rs:363-367
                    encode_value((int64_t)(int8_t)data[a * ws + w], k_pt, size_pt, limbs);
                    for (int j = 0; j < size_pt; j++) pg[(size_t)(j * 2) * N + q] = limbs[j];
                }
                const size_t gi = w * c->rows + x0 + x;
                if (!stage_random(pg, S, k, mask + gi * S * N, noise + gi * N))
                    return fail(c, FHERAM_ERR_RANGE, "mask limb outside [-2^16, 2^16) or |noise| >= 2^30");
            }
            rc = encrypt_staged(c, sk->d_hat, c->d_data + (w * c->rows + x0) * G, pre, nullptr, (int)nx, S);
            if (rc != FHERAM_OK) return rc;
        }
    c->initialized = true; c->state = false;
    return FHERAM_OK;
}

int fheram_address_encrypt_sk(fheram_ctx* c, const fheram_secret* sk, uint32_t value, const int64_t* mask,
                              const int64_t* noise, fheram_addr** out) {
    if (!c || !out) return FHERAM_ERR_INVALID_ARG;
    *out = nullptr;
    if (!mask || !noise) return fail(c, FHERAM_ERR_INVALID_ARG, "null argument");
    const int S = fheram_ctx::S_ADDR, k = (int)c->p.k_ggsw_addr, D = fheram_ctx::DNUM_CT;
    int rc = check_setup_args(c, sk, S, k);
    if (rc != FHERAM_OK) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    {   // debug_assert!(self.base2d.max() > value) (address.rs:98)
        unsigned bits = 0;
        for (auto& b1 : c->base2d) for (int b : b1) bits += (unsigned)b;
        if (bits < 32 && ((uint64_t)value >> bits) != 0)
            return fail(c, FHERAM_ERR_INVALID_ARG, "self.base2d.max() > value (address.rs:98): address does not fit the digit plan");
    }
    const size_t glen = fheram_ctx::GLWE4;
    const int n = c->n_digits * D * 2;
    std::vector<int32_t> pre((size_t)n * glen, 0), pt1((size_t)n * S * N, 0);
    size_t remain = value;
    int gi = 0;
    for (auto& base1d : c->base2d) {                                     // Address::encrypt_sk, address.rs:99-107
        unsigned tot = 0;
        for (int b : base1d) tot += (unsigned)b;
        const size_t max = (size_t)1 << tot;
        size_t rem_c = remain & (max - 1);                               // value of this coordinate; encrypted NEGATED (address.rs:104)
        remain /= max;
        const bool neg = rem_c != 0;
        unsigned tot_base = 0;
        for (int b : base1d) {                                           // Coordinate::encrypt_sk, coordinate.rs:146-176
            const size_t chunk = (rem_c & (((size_t)1 << b) - 1)) << tot_base;
            const size_t pos = (neg && chunk != 0) ? (size_t)N - chunk : chunk;
            const int32_t sgn = (neg && chunk != 0) ? -1 : 1;            // X^-chunk = -X^(N-chunk)
            for (int r = 0; r < D; r++)
                for (int ci = 0; ci < 2; ci++, gi++) {
                    int32_t* pg = pre.data() + (size_t)gi * glen;
                    if (ci == 0) pg[(size_t)(r * 2) * N + pos] = sgn;    // row r: m * 2^-((r+1)*base2k)
                    else pt1[((size_t)gi * S + r) * N + pos] = sgn;      //        m * s * ..., added to the mask column
                    if (!stage_random(pg, S, k, mask + (size_t)gi * S * N, noise + (size_t)gi * N))
                        return fail(c, FHERAM_ERR_RANGE, "mask limb outside [-2^16, 2^16) or |noise| >= 2^30");
                }
            rem_c >>= b;
            tot_base += (unsigned)b;
        }
    }
    fheram_addr* a = new fheram_addr{c, nullptr, c->n_digits, c->device};
    hipError_t e = hipMalloc(&a->d_ggsw, (size_t)c->n_digits * fheram_ctx::GGSW * sizeof(int32_t));
    if (e != hipSuccess) { delete a; return fail(c, FHERAM_ERR_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(e)); }
    rc = encrypt_staged(c, sk->d_hat, a->d_ggsw, pre, &pt1, n, S);
    if (rc != FHERAM_OK) { fheram_address_destroy(a); return rc; }
    *out = a;
    return FHERAM_OK;
}
int fheram_address_download(fheram_ctx* c, const fheram_addr* a, int64_t* out) {
    if (!c || !out) return FHERAM_ERR_INVALID_ARG;
    if (!a || a->ctx != c) return fail(c, FHERAM_ERR_INVALID_ARG, "address belongs to another context");
    HIPCHK(c, hipSetDevice(c->device));
    return download_i64(c, out, a->d_ggsw, (size_t)a->n_digits * fheram_ctx::GGSW);
}

int fheram_keys_encrypt_sk(fheram_ctx* c, const fheram_secret* sk, const int64_t* mask, const int64_t* noise, int64_t* std_out) {
    if (!c) return FHERAM_ERR_INVALID_ARG;
    if (!mask || !noise) return fail(c, FHERAM_ERR_INVALID_ARG, "null argument");
    int rc = check_setup_args(c, sk, fheram_ctx::S_EVK, (int)c->p.k_evk_trace);
    if (rc == FHERAM_OK) rc = check_setup_args(c, sk, fheram_ctx::S_INV, (int)c->p.k_evk_ggsw_inv);
    if (rc != FHERAM_OK) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    c->cur = c->stream;
    int32_t *d_stage = nullptr, *d_small = nullptr;
    double* d_hat = nullptr;
    const size_t stage_n = std::max(fheram_ctx::ATK, fheram_ctx::EVK5);
    HIPCHK(c, hipMalloc(&d_stage, stage_n * sizeof(int32_t)));
    hipError_t e = hipMalloc(&d_small, 2 * N * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc(&d_hat, N * sizeof(double));
    if (e != hipSuccess) { hipFree(d_stage); if (d_small) hipFree(d_small); return fail(c, FHERAM_ERR_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(e)); }
    std::vector<int32_t> sk_out(N), ss(2 * N, 0), pre;
    // GGLWE of `scalar` (placed on limb r of row r) under the secret whose prepared form is `hat`
    auto gglwe = [&](const int32_t* scalar, const double* hat, int rows, int S, int k, double* d_prepared, int64_t* std_dst, int64_t gal) -> int {
        const size_t glen = (size_t)S * 2 * N;
        pre.assign((size_t)rows * glen, 0);
        for (int r = 0; r < rows; r++) {
            int32_t* pg = pre.data() + (size_t)r * glen;
            if (r < S) std::copy(scalar, scalar + N, pg + (size_t)(r * 2) * N);
            if (!stage_random(pg, S, k, mask + (size_t)r * S * N, noise + (size_t)r * N))
                return fail(c, FHERAM_ERR_RANGE, "mask limb outside [-2^16, 2^16) or |noise| >= 2^30");
        }
        mask += (size_t)rows * S * N; noise += (size_t)rows * N;
        int rc2 = encrypt_staged(c, hat, d_stage, pre, nullptr, rows, S);
        if (rc2 == FHERAM_OK && std_dst) rc2 = download_i64(c, std_dst, d_stage, (size_t)rows * glen);
        if (rc2 == FHERAM_OK) { launch_prepare(c, d_stage, d_prepared, (int)((size_t)rows * glen / N), gal); hipStreamSynchronize(c->stream); }
        return rc2;
    };
    // key from s to phi_{p^-1}(s): phi_p(KS(a)) then decrypts under s (keys.rs:158-165,171-173)
    auto automorphism_key = [&](int64_t p, int rows, int S, int k, double* d_prepared, int64_t* std_dst) -> int {
        host_automorphism(galois_inv_mod(galois_mod(p)), sk->sk.data(), sk_out.data());
        hipMemcpyAsync(d_small, sk_out.data(), N * sizeof(int32_t), hipMemcpyHostToDevice, c->stream);
        launch_prepare(c, d_small, d_hat, 1);
        hipStreamSynchronize(c->stream);   // sk_out is reused by the next key
        return gglwe(sk->sk.data(), d_hat, rows, S, k, d_prepared, std_dst, p);
    };
    for (int i = 0; i < LOGN && rc == FHERAM_OK; i++)
        rc = automorphism_key(c->gal[i], fheram_ctx::DNUM_CT, fheram_ctx::S_EVK, (int)c->p.k_evk_trace,
                              c->d_atk + (size_t)i * fheram_ctx::ATK, std_out ? std_out + (size_t)i * fheram_ctx::ATK : nullptr);
    if (rc == FHERAM_OK) {   // tensor key (rank 1): GGLWE of s*s under s (keys.rs:167-169); s*s = phase of (0, s) under s
        std::copy(sk->sk.begin(), sk->sk.end(), ss.begin() + N);
        hipMemcpyAsync(d_small, ss.data(), 2 * N * sizeof(int32_t), hipMemcpyHostToDevice, c->stream);
        launch_enc_dyn(c, 1, 1, d_small, sk->d_hat, nullptr, 1);
        hipMemcpyAsync(ss.data(), d_small, N * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream);
        if (hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(c, FHERAM_ERR_DEVICE, "tensor key: s*s failed");
    }
    if (rc == FHERAM_OK)
        rc = gglwe(ss.data(), sk->d_hat, fheram_ctx::DNUM_GGSW, fheram_ctx::S_INV, (int)c->p.k_evk_ggsw_inv, c->d_tsk,
                   std_out ? std_out + (size_t)LOGN * fheram_ctx::ATK : nullptr, 0);
    if (rc == FHERAM_OK)
        rc = automorphism_key(-1, fheram_ctx::DNUM_GGSW, fheram_ctx::S_INV, (int)c->p.k_evk_ggsw_inv, c->d_atk_inv,
                              std_out ? std_out + (size_t)LOGN * fheram_ctx::ATK + fheram_ctx::EVK5 : nullptr);
    hipFree(d_stage); hipFree(d_small); hipFree(d_hat);
    std::fill(sk_out.begin(), sk_out.end(), 0); std::fill(ss.begin(), ss.end(), 0);
    if (rc != FHERAM_OK) return rc;
    HIPCHK(c, hipGetLastError());
    c->keys_loaded = true;
    return FHERAM_OK;
}

}  // extern "C"
