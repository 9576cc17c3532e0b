// Private to the C-ABI library (one translation unit, fheram.hip): context / address / secret objects,
// error and profiling helpers, the int64 <-> int32 layout conversion at the boundary.
#pragma once
#include "../../include/fheram.h"
#include "kernels.hpp"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

using namespace fk;

namespace {

thread_local std::string g_create_err;

// Twiddle table of the transforms: fk::make_fft_twiddles() (fft_dev.hpp).

struct ProfCls {
    uint64_t launches = 0, blocks = 0;
    double ms = 0.0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

}  // namespace

inline uint64_t next_addr_id() { static std::atomic<uint64_t> n{0}; return ++n; }

struct fheram_ctx {
    fheram_params p;
    int device = 0;
    hipStream_t stream = nullptr;    // main stream: every op is ordered on it
    hipStream_t stream2 = nullptr;   // side stream for work that is independent inside one op (write path)
    hipStream_t cur = nullptr;       // stream the launchers currently enqueue on
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_xout = nullptr, ev_xin = nullptr;   // ordering against a caller's stream (fheram_stream_signal / _wait)
    bool side_begun = false;                          // write_side_begin has been enqueued for the pending write
    // derived
    int ws = 0, n2 = 0, n_digits = 0;
    size_t rows = 0;        // GLWE rows per sub-RAM held by THIS context (all of them unless sharded)
    size_t rows_glob = 0;   // rows per sub-RAM of the whole RAM
    int shard = 0, n_shards = 1;   // row sharding: this context owns rows r = shard (mod n_shards)
    std::vector<std::vector<int>> base2d;
    static constexpr int S_CT = 3, S_ADDR = 4, S_INV = 5, DNUM_CT = 3, DNUM_GGSW = 4;
    int s_evk = 4;          // limbs of a trace / packing key: ceil(k_evk_trace / base2k) = 4 (source constants, parameters.rs:17) or 5 (README.md:17-27: K_EVK = 5 * BASEK)
    size_t atk = 0;         // elements of one trace key: DNUM_CT * s_evk * 2 * N   (evk_glwe_infos, parameters.rs:71-81)
    static constexpr size_t GLWE = (size_t)S_CT * 2 * N;                   // elements of a ct
    static constexpr size_t GLWE4 = (size_t)S_ADDR * 2 * N;                // one GGSW row ct
    static constexpr size_t GGSW = (size_t)DNUM_CT * 2 * GLWE4;            // elements of a GGSW
    static constexpr size_t EVK5 = (size_t)DNUM_GGSW * S_INV * 2 * N;      // inverse / tensor key
    static constexpr size_t GGSW5 = (size_t)DNUM_GGSW * 2 * S_INV * 2 * N; // one bit of an FheUint (N4)
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
    int limb_split = 1;            // FHERAM_LIMB_SPLIT=0 disables the limb-parallel paths
    int chain = 1;                 // FHERAM_CHAIN=0: one launch per step instead of one launch per dependent chain of fused steps
    int chain_y = 3;               // 3: the intermediates of a trace chain as Y = ceil(A/2) with the closed-form normalisation, handed over through LDS and registers (ks_trace_l); 0 (FHERAM_CHAIN_Y=0): as int32 limbs through global memory (ks_run)
    int fine_split = 1;            // FHERAM_FINE_SPLIT=0 disables the fine limb split (one workgroup per input and output limb)
    int use_graph = 0;             // FHERAM_GRAPH=1: replay each op's launch sequence from a hipGraph (per address)
    int32_t* d_part = nullptr;     // [ws]            this shard's partial pack / the un-rotated ct_lo
    int32_t* d_gat[3] = {nullptr, nullptr, nullptr};   // [n_shards][ws] gathered partials + ping-pong (root)
    int nco = 0;                   // output columns per workgroup: 1 = split by column (2 workgroups per
                                   // ciphertext), 2 = one workgroup, 0 = choose per launch from the batch size
    int cus = 256;
    double* d_prep = nullptr;      // [n_digits] prepared GGSW: coordinate ci at its first digit (write: inverse digits at 0)
    bool prep1_ready = false;      // coordinate 1 was prepared together with coordinate 0 (unsharded reads)
    int32_t* d_ggsw_tmp = nullptr; // [max digits per coordinate] std GGSW (inversion result)
    int32_t* d_ggsw_tmp2 = nullptr;
    int max_digits = 0;
    bool initialized = false, state = false, words_staged = false;
    // Results of read_prepare_write that Ram::write recomputes on the unchanged state (FHERAM_MEMO=0 recomputes them):
    //  memo_top  : d_trtop = trace(tree top) — the output of read_prepare_write (ram.rs:540) is the very value
    //              write_first_step computes first (ram.rs:571-572: same ciphertext, same deterministic operations);
    //  memo_alone: arena A holds every local row after the packer levels in which it is alone (ram.rs:514; n steps),
    //              which ARE the first n steps of trace(ct_hi) in write_mid_step (ram.rs:616).
    int memo = 1;
    //  tail: the dependent trace chain at the end of a read as ONE launch with in-kernel hand-offs (k_trace_tail);
    //        FHERAM_TAIL=0: one launch pair per step as before;  FHERAM_TAIL=2 / 3: test hooks, the launch gives up two steps before its end
    //        and the fused fallback launch behind it does the work.
    int tail = 1;
    int tail_ep = 1;                   // coordinate 1's products (two digits or more) run inside that launch, in front of the trace steps (FHERAM_TAIL_EP=0: launches of their own)
    int tail_test = 0;                 // FHERAM_TAIL=2 / 3: every launch gives up late; 3 keeps the watch below active
    unsigned tail_seq = 0;
    int tail_xoff = 0;                 // first XCD of this context's groups (0 or 4, alternating over contexts and processes)
    uint64_t tail_launches = 0;
    //  watch: the fallback launch mirrors its count into a pinned host word; if more than a quarter of the last 64
    //  single-launch chains gave up (a GPU shared so heavily, or partitioned so, that their groups do not fit side by
    //  side: each such launch waited ~5 ms first), the context goes back to one launch pair per step for good.
    unsigned* h_tail_fb = nullptr;     // pinned, device-visible
    unsigned tail_fb_mark = 0;
    uint64_t tail_launch_mark = 0;
    unsigned* d_tail_sync = nullptr;   // [8 groups][32] + abort generation, fallbacks taken
    //  mid: dependent chains on 9..64 ciphertexts as ONE launch with in-kernel hand-offs (k_chain_mid); FHERAM_MID=0: launch
    //       pairs per step as before; FHERAM_MID=2: test hook, every launch gives up late.  One sync block per stream.
    int mid = 2;
    int mid_test = 0;
    unsigned mid_seq = 0;
    uint64_t mid_launches = 0, mid_launch_mark = 0;
    unsigned mid_fb_mark = 0;
    int mid_bad_windows = 0, mid_saved = 0;          // auto-disable of the single-launch mid chains: consecutive bad windows; the setting to come back to
    uint64_t mid_window_cts = 0, mid_disabled_count = 0; // ciphertexts launched in the current window of 64 launches; times the path has been switched off
    unsigned mid_off_ops = 0;                        // ops since then (re-armed after 256)
    unsigned* h_mid_fb = nullptr;      // pinned, device-visible: ciphertexts redone, [0] main stream, [16] side stream
    unsigned* d_mid_sync[2] = {nullptr, nullptr};   // [64 groups][32] + [_, ciphertexts redone]: main / side stream
    double* d_mid_big[2] = {nullptr, nullptr};      // [step parity][ciphertext x RS <= 64] x BIG_STRIDE doubles: k_chain_mid's partial limb polynomials
    double* d_mid_y[2] = {nullptr, nullptr};        // [2][64 groups][2][N] doubles: the chain's intermediates in the one-double form
    //  inv_id[ci]: d_prep_inv holds the prepared INVERSE digits of coordinate ci of the address with that id
    //              (CoordinatePrepared::prepare_inv, ram.rs:260-271,278-289): read_prepare_write — which is told the
    //              address the write will use — starts them on the (low-priority) side stream next to its trace chain,
    //              which is one launch holding 24 CUs on half of the XCDs: the rest of the chip is idle then.  A write
    //              with another address, or after new keys, computes them itself.  FHERAM_PRE_INV=0: always.
    int pair_z = 1;                // FHERAM_PAIR_Z=0: the column-split packer combine with the limb-by-limb normalisation (k_keyswitch<KS_PAIR,...,1>) instead of k_pair_z
    int fuse = 1;                  // FHERAM_FUSE=0: a row's product chain and trace chain as two launches (and the write's elementwise step as a third) instead of k_read_chain / k_write_chain
    bool wide = false;             // the launch being enqueued can never meet the gate wave (Ram::read, Ram::write: only read_prepare_write parks one): the chain kernels' 256-register variants
    int safe = 0;                  // FHERAM_SAFE=1: no in-kernel hand-offs between workgroups, no gate wave (fheram.hip)
    //  Round-off monitor (fft_dev.hpp mon_note / RoMonitor): every rounding of an inverse transform reports |x - rint(x)|; the
    //  context's maximum lives behind its twiddle table (d_tw[N]), and a pinned host word is set once it passed MON_LIMIT.
    int monitor = 1;               // 0: nothing is reported or checked; 1 (default): one coefficient per thread and transform; 2 (`safe`): every coefficient
    unsigned* h_ro_flag = nullptr; // pinned, device-visible: 1 = a round-off above MON_LIMIT was seen (sticky until fheram_roundoff_reset)
    int pre_inv = 1;
    uint64_t inv_id[2] = {0, 0};
    double* d_prep_inv = nullptr;  // [n_digits] prepared GGSW
    int32_t* d_ggsw_inv = nullptr; // [n_digits] std GGSW: the inversion result on its way there (own scratch: runs beside anything)
    hipEvent_t ev_inv[2] = {nullptr, nullptr};
    bool inv_pending[2] = {false, false};   // a precompute of coordinate ci has been enqueued on the side stream and no write has consumed / overwritten it
    hipEvent_t ev_wdone = nullptr;          // recorded at the end of every write (main stream): the next precompute waits for it
    // recorded on the main stream where a read_prepare_write STARTS (free there: nothing of the op has been enqueued yet); its gate wave waits for
    // it on the side stream.  A host that enqueues ops without waiting for them is ahead of the device: without this the gate wave could be
    // parked while the PREVIOUS op's 256-register chain launch is still being placed, and cost that launch a second round on one CU.
    // Only needed while a 256-register launch of an earlier op may still be waiting for its CUs: wide_unsynced = such a launch has been enqueued
    // and the host has not waited for the stream since (a host that waits for every op — the reference's calling pattern — never pays the record).
    hipEvent_t ev_opstart = nullptr;
    bool opstart_valid = false;
    bool wide_unsynced = false;
    bool wdone_pending = false;
    bool memo_top = false;
    int memo_alone = 0;
    int32_t* d_trtop = nullptr;    // [ws]
    int32_t* d_last_res = nullptr; // where the last read / read_prepare_write left its result (d_res or d_trtop)
    bool tree_rotate_pending = false;
    int32_t* d_trhi = nullptr;     // arena that holds trace(ct_hi) of the local rows during a write (A or C)
    int32_t* h_pin[2] = {nullptr, nullptr};   // pinned host staging (hand-over of int64 host buffers)
    hipEvent_t ev_pin[2] = {nullptr, nullptr};
    // the two hand-overs ON the path (result of a read out, words of a write in) have staging of their own:
    //  h_res: pinned, device-visible; an export kernel widens the result into it as int64 in the ABI's layout (no DMA
    //         engine round trip, no host-side widening: the host copies it out, or reads it in place, fheram_result_map)
    //  h_w  : pinned; the host narrows the words into it and an asynchronous copy takes them to d_w — the call does not wait
    int64_t* h_res = nullptr;
    int64_t* d_h_res = nullptr;               // device address of h_res
    int32_t* h_w = nullptr;
    hipEvent_t ev_w = nullptr;
    bool w_busy = false;
    // profiling
    int profile = 0;               // 1: every launch class bracketed by HIP events; 2: only the chain launches themselves (two events per launch: leaves back-to-back submission intact)
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
    uint64_t id = next_addr_id();   // never reused (a freed address may be followed by another one at the same pointer)
    hipGraphExec_t graph[3] = {nullptr, nullptr, nullptr};   // captured launch sequences: read, read_prepare_write, write
    unsigned graph_sig[3] = {0, 0, 0};                       // context state the capture depended on (run_op)
};

struct fheram_fheuint {
    fheram_ctx* ctx;
    int device, n_bits;
    int32_t* d_std;    // [n_bits] std-form GGSW (5 limbs, dnum 4), int32
    double* d_prep;    // the same, prepared
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
// Called wherever the host has just waited for the device: the sticky flag of the round-off monitor turns into a status.
int check_precision(fheram_ctx* c) {
    if (c->h_ro_flag && __atomic_load_n(c->h_ro_flag, __ATOMIC_RELAXED))
        return fail(c, FHERAM_ERR_PRECISION, "FP64 round-off of an inverse transform exceeded 3/8 (fheram_roundoff_max): the rounded integers are no longer trustworthy");
    return FHERAM_OK;
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
    // steps: ciphertext-operation rounds inside the launch (a chain kernel runs several): `launches` counts rounds,
    // so that ms / launches stays the time of ONE round over `blocks / launches` ciphertexts
    ProfScope(fheram_ctx* c_, const char* name, uint64_t blocks, int steps = 1) : c(c_) {
        if (!c->profile) return;
        if (c->profile == 2 && !strstr(name, "chain_launch")) return;
        cls = &c->prof[name];
        cls->launches += steps; cls->blocks += blocks * steps;
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

// The hand-over goes through two pinned staging buffers of PIN_CHUNK elements: the narrowing / widening
// of one chunk overlaps the DMA of the other, and the DMA itself runs at the pinned-memory rate (a
// pageable copy is staged by the runtime a second time).  Both calls return when the data has arrived.
constexpr size_t PIN_CHUNK = (size_t)1 << 21;   // int32 elements per staging buffer (8 MiB)
int pin_init(fheram_ctx* c) {
    if (c->h_pin[0]) return FHERAM_OK;
    for (int b = 0; b < 2; b++) {
        HIPCHK(c, hipHostMalloc((void**)&c->h_pin[b], PIN_CHUNK * sizeof(int32_t), hipHostMallocDefault));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_pin[b], hipEventDisableTiming));
    }
    return FHERAM_OK;
}
int upload_i64(fheram_ctx* c, int32_t* dst, const int64_t* src, size_t n) {
    int rc = pin_init(c);
    if (rc != FHERAM_OK) return rc;
    bool ok = true;
    int k = 0;
    for (size_t off = 0; off < n && ok; off += PIN_CHUNK, k++) {
        const int b = k & 1;
        const size_t m = std::min(PIN_CHUNK, n - off);
        if (k >= 2) HIPCHK(c, hipEventSynchronize(c->ev_pin[b]));   // the copy that last used this buffer is done
        ok = narrow(src + off, c->h_pin[b], m);
        if (!ok) break;
        HIPCHK(c, hipMemcpyAsync(dst + off, c->h_pin[b], m * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipEventRecord(c->ev_pin[b], c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (!ok) return fail(c, FHERAM_ERR_RANGE, "limb out of the normalised range [-2^16, 2^16]");
    return FHERAM_OK;
}
int download_i64(fheram_ctx* c, int64_t* dst, const int32_t* src, size_t n) {
    int rc = pin_init(c);
    if (rc != FHERAM_OK) return rc;
    const size_t n_chunks = (n + PIN_CHUNK - 1) / PIN_CHUNK;
    for (size_t k = 0; k <= n_chunks; k++) {
        if (k < n_chunks) {   // request chunk k
            const size_t off = k * PIN_CHUNK, m = std::min(PIN_CHUNK, n - off);
            HIPCHK(c, hipMemcpyAsync(c->h_pin[k & 1], src + off, m * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipEventRecord(c->ev_pin[k & 1], c->stream));
        }
        if (k > 0) {          // widen chunk k - 1 while chunk k is in flight
            const size_t off = (k - 1) * PIN_CHUNK, m = std::min(PIN_CHUNK, n - off);
            HIPCHK(c, hipEventSynchronize(c->ev_pin[(k - 1) & 1]));
            widen(c->h_pin[(k - 1) & 1], dst + off, m);
        }
    }
    return FHERAM_OK;
}

}  // namespace
