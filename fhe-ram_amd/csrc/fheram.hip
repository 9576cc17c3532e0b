// C-ABI library of the MI355X-native FHE-RAM evaluator (include/fheram.h).
// Host orchestration of Ram::read / read_prepare_write / write (reference: src/ram.rs) over the
// fused HIP kernels in kernels.hpp.  No CPU compute path exists here: every ciphertext
// operation is a kernel launch, and a missing GPU is a hard error.
#include <unistd.h>
#include <atomic>
#include "path.hpp"

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

// Library defaults, then FHERAM_* environment overrides (how the tests force every decomposition and hand-over form).
void fheram_config_default(fheram_config* cfg) {
    if (!cfg) return;
    fheram_config d{};
    d.limb_split = 1; d.fine_split = 1; d.memo = 1; d.pre_inv = 1; d.tail = 1; d.tail_test = 0; d.mid = 2; d.mid_test = 0;
    d.chain = 1; d.chain_y = 3; d.pair_z = 1; d.fuse = 1; d.graph = 0; d.safe = 0; d.nco = 0; d.tail_ep = 1; d.monitor = 1; d.reserved = 0;
    auto env = [](const char* n) { const char* v = getenv(n); return (v && v[0]) ? v[0] : '\0'; };
    if (env("FHERAM_LIMB_SPLIT") == '0') d.limb_split = 0;
    if (env("FHERAM_FINE_SPLIT") == '0') d.fine_split = 0;
    if (env("FHERAM_MEMO") == '0') d.memo = 0;
    if (env("FHERAM_PRE_INV") == '0') d.pre_inv = 0; else if (env("FHERAM_PRE_INV") == '2') d.pre_inv = 2;
    const char tl = env("FHERAM_TAIL");
    if (tl == '0') d.tail = 0;
    d.tail_test = tl == '2' ? 1 : (tl == '3' ? 2 : 0);
    const char md = env("FHERAM_MID");
    d.mid = md == '0' ? 0 : (md == '1' ? 1 : 2);
    d.mid_test = md == '2' ? 1 : 0;
    if (env("FHERAM_CHAIN") == '0') d.chain = 0;
    if (env("FHERAM_CHAIN_Y") == '0') d.chain_y = 0;
    if (env("FHERAM_PAIR_Z") == '0') d.pair_z = 0;
    if (env("FHERAM_FUSE") == '0') d.fuse = 0;
    if (env("FHERAM_GRAPH") == '1') d.graph = 1;
    if (env("FHERAM_SAFE") == '1') d.safe = 1;
    const char nc = env("FHERAM_NCO");
    d.nco = nc == '2' ? 2 : (nc == '1' ? 1 : 0);
    if (env("FHERAM_TAIL_EP") == '0') d.tail_ep = 0;
    const char mo = env("FHERAM_MONITOR");
    if (mo == '0') d.monitor = 0; else if (mo == '2') d.monitor = 2;
    *cfg = d;
}

int fheram_ctx_create(const fheram_params* p, int device, fheram_ctx** out) {
    return fheram_ctx_create_cfg(p, device, 0, 1, nullptr, out);
}

int fheram_ctx_create_sharded(const fheram_params* p, int device, int shard, int n_shards, fheram_ctx** out) {
    return fheram_ctx_create_cfg(p, device, shard, n_shards, nullptr, out);
}

int fheram_ctx_create_cfg(const fheram_params* p, int device, int shard, int n_shards, const fheram_config* user_cfg, fheram_ctx** out) {
    if (!p || !out) return fail(nullptr, FHERAM_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    if (n_shards < 1 || shard < 0 || shard >= n_shards || (n_shards & (n_shards - 1)) != 0)
        return fail(nullptr, FHERAM_ERR_INVALID_ARG, "n_shards must be a power of two and 0 <= shard < n_shards");
    // The kernels are built for the limb counts of the reference's two published parameter blocks: the source constants
    // (parameters.rs:11-18: trace keys at 4 limbs) and the README block its timings were taken with (README.md:17-27:
    // K_EVK = 5 * BASEK for every evaluation key, K_PT = 9).  Precisions are accepted anywhere inside those limb counts.
    auto limbs = [&](uint32_t k) { return (k + p->base2k - 1) / (p->base2k ? p->base2k : 1); };
    if (p->log_n != 12 || p->base2k != 17 || p->rank != 1 || p->k_glwe_ct == 0 || limbs(p->k_glwe_ct) != 3 || limbs(p->k_ggsw_addr) != 4 ||
        (limbs(p->k_evk_trace) != 4 && limbs(p->k_evk_trace) != 5) || limbs(p->k_evk_ggsw_inv) != 5)
        return fail(nullptr, FHERAM_ERR_UNSUPPORTED, "kernels are built for LOG_N=12, BASE2K=17, RANK=1 and limb counts K_CT: 3, K_ADDR: 4, K_EVK_TRACE: 4 or 5, K_EVK_GGSW_INV: 5");
    if (p->k_glwe_pt == 0 || p->k_glwe_pt > p->k_glwe_ct)
        return fail(nullptr, FHERAM_ERR_INVALID_ARG, "k_glwe_pt must be in [1, k_glwe_ct]");
    if (p->word_size == 0 || p->word_size > 64 || p->n_decomp == 0 || p->n_decomp > 16 || p->max_addr < 2)
        return fail(nullptr, FHERAM_ERR_INVALID_ARG, "bad word_size / decomp_n / max_addr");
    unsigned sum = 0;
    for (uint32_t i = 0; i < p->n_decomp; i++) { if (p->decomp_n[i] == 0) return fail(nullptr, FHERAM_ERR_INVALID_ARG, "zero digit width"); sum += p->decomp_n[i]; }
    if (sum != p->log_n) return fail(nullptr, FHERAM_ERR_INVALID_ARG, "DECOMP_N must sum to LOG_N (parameters.rs:168)");
    if (p->max_addr > ((uint64_t)N * N)) return fail(nullptr, FHERAM_ERR_UNSUPPORTED, "max_addr > N^2 is not supported by the reference either (SURVEY.md 3.1)");
    if (user_cfg && user_cfg->reserved != 0)
        return fail(nullptr, FHERAM_ERR_INVALID_ARG, "fheram_config.reserved must be 0 (start from fheram_config_default())");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, FHERAM_ERR_DEVICE, "no HIP device: the FHE-RAM evaluator has no CPU path");
    if (device < 0 || device >= ndev) return fail(nullptr, FHERAM_ERR_INVALID_ARG, "bad device index");

    fheram_ctx* c = new fheram_ctx();
    c->p = *p; c->device = device; c->ws = (int)p->word_size;
    c->s_evk = (int)limbs(p->k_evk_trace);
    c->atk = (size_t)fheram_ctx::DNUM_CT * c->s_evk * 2 * N;
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
    {   // side stream at the lowest priority: when both streams have a launch ready, the main stream's goes first
        int lo = 0, hi = 0;
        CCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        CCHK(hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, lo));
    }
    c->cur = c->stream;
    CCHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    CCHK(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    CCHK(hipEventCreateWithFlags(&c->ev_xout, hipEventDisableTiming));
    CCHK(hipEventCreateWithFlags(&c->ev_xin, hipEventDisableTiming));
    CCHK(hipEventCreate(&c->t0));
    CCHK(hipEventCreate(&c->t1));
    {
        fheram_config cfg;
        if (user_cfg) cfg = *user_cfg; else fheram_config_default(&cfg);
        c->limb_split = cfg.limb_split ? 1 : 0;
        c->fine_split = cfg.fine_split ? 1 : 0;
        c->memo = cfg.memo ? 1 : 0;
        c->pre_inv = c->memo ? (cfg.pre_inv == 2 ? 2 : (cfg.pre_inv ? 1 : 0)) : 0;
        c->tail = cfg.tail ? 1 : 0;
        static std::atomic<int> serial{0};
        c->tail_xoff = ((serial++ + (int)getpid()) & 1) * (TAIL_GROUPS / 2);
        c->tail_test = cfg.tail_test < 0 ? 0 : (cfg.tail_test > 2 ? 2 : cfg.tail_test);
        c->tail_ep = cfg.tail_ep ? 1 : 0;
        c->mid = cfg.mid < 0 ? 0 : (cfg.mid > 2 ? 2 : cfg.mid);   // 1: the <= 16 ciphertext split only
        c->mid_test = cfg.mid_test ? 1 : 0;
        c->chain = cfg.chain ? 1 : 0;
        c->chain_y = cfg.chain_y ? 3 : 0;
        c->pair_z = cfg.pair_z ? 1 : 0;
        c->fuse = cfg.fuse ? 1 : 0;
        c->use_graph = cfg.graph ? 1 : 0;
        // a captured launch sequence must be a pure function of (context, address, op): under replay the write always
        // computes its own inverse digits (whether a precompute matched is state the capture would freeze)
        if (c->use_graph) c->pre_inv = 0;
        // safe: ONE switch for a configuration that stays inside the HIP memory model — no launch with in-kernel hand-offs
        // between workgroups (k_trace_tail, k_chain_mid: relaxed agent-scope atomics + drained stores + L1-bypassing loads on one
        // XCD's L2) and no gate wave (k_tail_gate): dependent steps are kernel boundaries, the side work forks from an event.
        // Same results (tests/test_gpu_golden.py); priced in profiles/r05_bench_safe.json.
        c->safe = cfg.safe ? 1 : 0;
        if (c->safe) { c->tail = 0; c->tail_test = 0; c->mid = 0; c->mid_test = 0; if (c->pre_inv == 1) c->pre_inv = 2; }
        // round-off monitor: one coefficient per thread and transform by default, every coefficient under `safe`
        c->monitor = cfg.monitor < 0 ? 0 : (cfg.monitor > 2 ? 2 : cfg.monitor);
        if (c->safe && c->monitor == 1) c->monitor = 2;
        c->nco = cfg.nco == 2 ? 2 : (cfg.nco == 1 ? 1 : 0);
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) c->cus = prop.multiProcessorCount;
    }
#define LDSATTR(k) CCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES))
    LDSATTR(k_prepare);
    LDSATTR((&k_ext_product<3, 4, 1>));
    LDSATTR((&k_ext_product<3, 4, 2>));
    LDSATTR((&k_ext_product<3, 4, 1, 1>));
    LDSATTR((&k_ext_product_fine<3, 4>));
    LDSATTR((&k_ext_product_fine<4, 5>));
    LDSATTR((&k_ext_product_chain<3, 4>));
    LDSATTR((&k_ext_product_chain_r<4>));
    LDSATTR((&k_pair_z<4>)); LDSATTR((&k_pair_z<5>));
    LDSATTR((&k_read_chain<4, 4>)); LDSATTR((&k_read_chain<5, 4>)); LDSATTR((&k_write_chain<4, 4>)); LDSATTR((&k_write_chain<5, 4>));
    LDSATTR((&k_read_chain_w<4, 4>)); LDSATTR((&k_read_chain_w<5, 4>));
    LDSATTR((&k_keyswitch_chain_w<3, 4, 3, 3>)); LDSATTR((&k_keyswitch_chain_w<3, 5, 3, 3>));
    LDSATTR((&k_keyswitch_chain<3, 4, 3>));
    LDSATTR((&k_keyswitch_chain<3, 4, 3, 3>));
    LDSATTR((&k_keyswitch_chain<3, 5, 3, 3>));
    LDSATTR((&k_trace_tail<3, 4, 3>));
    LDSATTR((&k_chain_mid<false, 4, 3, 2>)); LDSATTR((&k_chain_mid<false, 5, 3, 2>)); LDSATTR((&k_chain_mid<true, 4, 3, 2>));
    LDSATTR((&k_chain_mid<false, 4, 1, 1>)); LDSATTR((&k_chain_mid<false, 5, 1, 1>));
    LDSATTR((&k_chain_mid<false, 4, 1, 2>)); LDSATTR((&k_chain_mid<false, 5, 1, 2>));
#define LDSATTR_KS4(M, SX, SK, SO) LDSATTR((&k_keyswitch<M, SX, SK, SO, 1>)); LDSATTR((&k_keyswitch<M, SX, SK, SO, 1, 1>)); LDSATTR((&k_keyswitch_fine<M, SX, SK>))
#define LDSATTR_KS(M, SX, SK, SO) LDSATTR_KS4(M, SX, SK, SO); LDSATTR((&k_keyswitch<M, SX, SK, SO, 2>))
    LDSATTR_KS(KS_AUTO, 3, 4, 3);
    LDSATTR_KS(KS_TRACE, 3, 4, 3);
    LDSATTR_KS4(KS_PAIR, 3, 4, 3);
    LDSATTR_KS(KS_ADD, 3, 4, 3);
    LDSATTR_KS(KS_SUBNEG, 3, 4, 3);
    LDSATTR((&k_keyswitch_chain<3, 5, 3>));
    LDSATTR((&k_trace_tail<3, 5, 3>));
    LDSATTR_KS(KS_AUTO, 3, 5, 3);
    LDSATTR_KS(KS_TRACE, 3, 5, 3);
    LDSATTR_KS4(KS_PAIR, 3, 5, 3);
    LDSATTR_KS(KS_ADD, 3, 5, 3);
    LDSATTR_KS(KS_SUBNEG, 3, 5, 3);
    LDSATTR_KS4(KS_AUTO, 4, 5, 4);
    LDSATTR_KS4(KS_TENSOR, 4, 5, 4);
    LDSATTR((&k_encrypt_sk<1, 1>));
    LDSATTR((&k_encrypt_sk<3, 0>)); LDSATTR((&k_encrypt_sk<3, 1>));
    LDSATTR((&k_encrypt_sk<4, 0>)); LDSATTR((&k_encrypt_sk<4, 1>));
    LDSATTR((&k_encrypt_sk<5, 0>)); LDSATTR((&k_encrypt_sk<5, 1>));
#undef LDSATTR_KS
#undef LDSATTR_KS4
#undef LDSATTR

    std::vector<double> tw = make_fft_twiddles();
    c->ninv = 1.0 / (double)NC;   // the inverse transform's 1/n, folded into every prepared operand (a power of two: exact)
    {   // the round-off monitor's words ride with the table (fft_dev.hpp TW_GLOBAL): mode in the first dword of slot 0, the maximum
        // at [N], the address of the pinned flag word at [N + 1]
        CCHK(hipHostMalloc((void**)&c->h_ro_flag, 64, hipHostMallocDefault));
        *c->h_ro_flag = 0;
        const uint64_t mode = (uint64_t)c->monitor, flag_addr = (uint64_t)(size_t)c->h_ro_flag;
        std::memcpy(&tw[0], &mode, 8);
        std::memcpy(&tw[N + 1], &flag_addr, 8);
    }
    CCHK(hipMalloc(&c->d_tw, TW_GLOBAL * sizeof(double)));
    CCHK(hipMemcpy(c->d_tw, tw.data(), TW_GLOBAL * sizeof(double), hipMemcpyHostToDevice));
    const size_t G = fheram_ctx::GLWE, nrow = (size_t)c->ws * c->rows;
    CCHK(hipMalloc(&c->d_atk, (size_t)LOGN * c->atk * sizeof(double)));
    CCHK(hipMalloc(&c->d_atk_inv, fheram_ctx::EVK5 * sizeof(double)));
    CCHK(hipMalloc(&c->d_tsk, fheram_ctx::EVK5 * sizeof(double)));
    CCHK(hipMalloc(&c->d_data, nrow * G * sizeof(int32_t)));
    CCHK(hipMalloc(&c->d_scrA, nrow * G * sizeof(int32_t)));
    CCHK(hipMalloc(&c->d_scrB, nrow * G * sizeof(int32_t)));
    CCHK(hipMalloc(&c->d_scrC, nrow * G * sizeof(int32_t)));
    CCHK(hipMalloc(&c->d_scrD, nrow * G * sizeof(int32_t)));
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
    CCHK(hipMalloc(&c->d_trtop, (size_t)c->ws * G * sizeof(int32_t)));
    CCHK(hipMalloc(&c->d_prep, (size_t)std::max(c->n_digits, c->max_digits) * fheram_ctx::GGSW * sizeof(double)));
    CCHK(hipMalloc(&c->d_ggsw_tmp, (size_t)c->max_digits * fheram_ctx::GGSW * sizeof(int32_t)));
    CCHK(hipMemset(c->d_tree, 0, (size_t)c->ws * G * sizeof(int32_t)));
    CCHK(hipMalloc(&c->d_prep_inv, (size_t)c->n_digits * fheram_ctx::GGSW * sizeof(double)));
    CCHK(hipMalloc(&c->d_ggsw_inv, (size_t)c->n_digits * fheram_ctx::GGSW * sizeof(int32_t)));
    for (int i = 0; i < 2; i++) CCHK(hipEventCreateWithFlags(&c->ev_inv[i], hipEventDisableTiming));
    CCHK(hipEventCreateWithFlags(&c->ev_wdone, hipEventDisableTiming));
    CCHK(hipEventCreateWithFlags(&c->ev_opstart, hipEventDisableTiming));
    CCHK(hipMalloc(&c->d_tail_sync, (size_t)(TAIL_GROUPS + 1) * 32 * sizeof(unsigned)));
    CCHK(hipMemset(c->d_tail_sync, 0, (size_t)(TAIL_GROUPS + 1) * 32 * sizeof(unsigned)));
    for (int i = 0; i < 2; i++) {
        CCHK(hipMalloc(&c->d_mid_sync[i], (size_t)(MID_GROUPS_MAX + 1) * 32 * sizeof(unsigned)));
        CCHK(hipMemset(c->d_mid_sync[i], 0, (size_t)(MID_GROUPS_MAX + 1) * 32 * sizeof(unsigned)));
        CCHK(hipMalloc(&c->d_mid_y[i], (size_t)2 * MID_GROUPS_MAX * 2 * N * sizeof(double)));
        CCHK(hipMalloc(&c->d_mid_big[i], (size_t)2 * MID_GROUPS_MAX * BIG_STRIDE * sizeof(double)));
    }
    CCHK(hipHostMalloc((void**)&c->h_tail_fb, 64, hipHostMallocDefault));
    *c->h_tail_fb = 0;
    CCHK(hipHostMalloc((void**)&c->h_mid_fb, 128, hipHostMallocDefault));
    memset(c->h_mid_fb, 0, 128);
    CCHK(hipHostMalloc((void**)&c->h_res, ((size_t)c->ws * G + 1) * sizeof(int64_t), hipHostMallocMapped));   // + the monitor's maximum at export time
    CCHK(hipHostGetDevicePointer((void**)&c->d_h_res, c->h_res, 0));
    CCHK(hipHostMalloc((void**)&c->h_w, (size_t)c->ws * G * sizeof(int32_t), hipHostMallocDefault));
    CCHK(hipEventCreateWithFlags(&c->ev_w, hipEventDisableTiming));
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
    for (int i = 0; i < 2; i++) if (c->ev_inv[i]) hipEventDestroy(c->ev_inv[i]);
    if (c->ev_wdone) hipEventDestroy(c->ev_wdone);
    if (c->ev_opstart) hipEventDestroy(c->ev_opstart);
    if (c->ev_xout) hipEventDestroy(c->ev_xout);
    if (c->ev_xin) hipEventDestroy(c->ev_xin);
    if (c->stream2) hipStreamDestroy(c->stream2);
    for (auto e : c->ev_pool) hipEventDestroy(e);
    if (c->t0) hipEventDestroy(c->t0);
    if (c->t1) hipEventDestroy(c->t1);
    for (int b = 0; b < 2; b++) { if (c->ev_pin[b]) hipEventDestroy(c->ev_pin[b]); if (c->h_pin[b]) hipHostFree(c->h_pin[b]); }
    void* bufs[] = {c->d_tw, c->d_atk, c->d_atk_inv, c->d_tsk, c->d_data, c->d_scrA, c->d_scrB, c->d_big, c->d_big2, c->d_scrC, c->d_scrD, c->d_ggsw_tmp2, c->d_tmp2, c->d_part, c->d_gat[0], c->d_gat[1], c->d_gat[2], c->d_tree, c->d_res, c->d_tmp, c->d_w, c->d_trtop, c->d_prep, c->d_ggsw_tmp, c->d_tail_sync, c->d_prep_inv, c->d_ggsw_inv, c->d_mid_sync[0], c->d_mid_sync[1], c->d_mid_y[0], c->d_mid_y[1], c->d_mid_big[0], c->d_mid_big[1]};
    for (void* b : bufs) if (b) hipFree(b);
    if (c->h_tail_fb) hipHostFree(c->h_tail_fb);
    if (c->h_mid_fb) hipHostFree(c->h_mid_fb);
    if (c->h_res) hipHostFree(c->h_res);
    if (c->h_ro_flag) hipHostFree(c->h_ro_flag);
    if (c->h_w) hipHostFree(c->h_w);
    if (c->ev_w) hipEventDestroy(c->ev_w);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

int fheram_ctx_config(const fheram_ctx* c, fheram_config* out) {
    if (!c || !out) return FHERAM_ERR_INVALID_ARG;
    fheram_config d{};
    d.limb_split = c->limb_split; d.fine_split = c->fine_split; d.memo = c->memo; d.pre_inv = c->pre_inv; d.tail = c->tail; d.tail_test = c->tail_test;
    d.mid = c->mid; d.mid_test = c->mid_test; d.chain = c->chain; d.chain_y = c->chain_y; d.pair_z = c->pair_z; d.fuse = c->fuse;
    d.graph = c->use_graph; d.safe = c->safe; d.nco = c->nco; d.monitor = c->monitor; d.tail_ep = c->tail_ep;
    *out = d;
    return FHERAM_OK;
}

size_t fheram_glwe_len(const fheram_ctx*) { return fheram_ctx::GLWE; }
size_t fheram_ggsw_len(const fheram_ctx*) { return fheram_ctx::GGSW; }
size_t fheram_atk_len(const fheram_ctx* c) { return c ? c->atk : 0; }
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
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream2));   // a precompute started by read_prepare_write may still be reading the old keys
    // keys may come in any order (the reference keeps them in a HashMap, keys.rs:28): sort by Galois element
    int order[LOGN];
    for (int i = 0; i < LOGN; i++) {
        order[i] = -1;
        for (int j = 0; j < n_gal; j++) if (gal_els[j] == c->gal[i]) order[i] = j;
        if (order[i] < 0) return fail(c, FHERAM_ERR_KEYS, "missing trace key for a Galois element of GLWE::trace_galois_elements");
    }
    int32_t* d_stage = nullptr;
    const size_t stage_n = std::max(c->atk, fheram_ctx::EVK5);
    HIPCHK(c, hipMalloc(&d_stage, stage_n * sizeof(int32_t)));
    int rc = FHERAM_OK;
    for (int i = 0; i < LOGN && rc == FHERAM_OK; i++) {
        rc = upload_i64(c, d_stage, atk_glwe[order[i]], c->atk);
        if (rc == FHERAM_OK) launch_prepare(c, d_stage, c->d_atk + (size_t)i * c->atk, (int)(c->atk / N), c->gal[i]);
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
    c->inv_id[0] = c->inv_id[1] = 0;   // inverse digits prepared with the previous keys are void
    c->inv_pending[0] = c->inv_pending[1] = false;
    c->memo_top = false; c->memo_alone = 0;   // ... and so are the traces read_prepare_write kept for the write (ram.rs:572,616 use the keys of the write)
    return FHERAM_OK;
}

int fheram_ram_upload(fheram_ctx* c, const int64_t* rows) {
    if (!c) return FHERAM_ERR_INVALID_ARG;
    if (!rows) return fail(c, FHERAM_ERR_INVALID_ARG, "null rows");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = upload_i64(c, c->d_data, rows, (size_t)c->ws * c->rows * fheram_ctx::GLWE);
    if (rc != FHERAM_OK) return rc;
    c->initialized = true; c->state = false; c->memo_top = false; c->memo_alone = 0;
    return FHERAM_OK;
}
int fheram_ram_download(fheram_ctx* c, int64_t* rows) {
    if (!c || !rows) return FHERAM_ERR_INVALID_ARG;
    if (!c->initialized) return fail(c, FHERAM_ERR_UNINITIALIZED, "unitialized memory: self.data.len()=0");
    HIPCHK(c, hipSetDevice(c->device));
    const int rc = download_i64(c, rows, c->d_data, (size_t)c->ws * c->rows * fheram_ctx::GLWE);
    return rc == FHERAM_OK ? check_precision(c) : rc;
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

int fheram_result_map(fheram_ctx* c, const int64_t** out) {
    if (!c || !out) return FHERAM_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const int n4 = (int)((size_t)c->ws * fheram_ctx::GLWE / 4);
    hipLaunchKernelGGL(k_export_i64, dim3((n4 + 255) / 256), dim3(256), 0, c->stream, c->d_last_res ? c->d_last_res : c->d_res,
                       reinterpret_cast<long long*>(c->d_h_res), n4, reinterpret_cast<const long long*>(c->d_tw + N));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    c->wide_unsynced = false;
    *out = c->h_res;
    {   // the monitor's maximum as it stood when the result was exported (the export kernel copies it behind the result)
        double m;
        std::memcpy(&m, c->h_res + (size_t)c->ws * fheram_ctx::GLWE, 8);
        if (c->monitor && m > MON_LIMIT) __atomic_store_n(c->h_ro_flag, 1u, __ATOMIC_RELAXED);
    }
    return check_precision(c);
}
int fheram_result_download(fheram_ctx* c, int64_t* out) {
    if (!c || !out) return FHERAM_ERR_INVALID_ARG;
    const int64_t* src = nullptr;
    const int rc = fheram_result_map(c, &src);
    if (rc != FHERAM_OK) return rc;
    std::memcpy(out, src, (size_t)c->ws * fheram_ctx::GLWE * sizeof(int64_t));
    return FHERAM_OK;
}
int fheram_sync(fheram_ctx* c) {
    if (!c) return FHERAM_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    c->wide_unsynced = false;
    return check_precision(c);
}
int fheram_roundoff_max(fheram_ctx* c, double* max_out) {
    if (!c || !max_out) return FHERAM_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream2));
    double m = 0.0;
    HIPCHK(c, hipMemcpy(&m, c->d_tw + N, sizeof(double), hipMemcpyDeviceToHost));
    *max_out = m;
    if (c->monitor && m > MON_LIMIT) __atomic_store_n(c->h_ro_flag, 1u, __ATOMIC_RELAXED);
    return check_precision(c);
}
int fheram_roundoff_reset(fheram_ctx* c) {
    if (!c) return FHERAM_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream2));
    HIPCHK(c, hipMemset(c->d_tw + N, 0, sizeof(double)));
    __atomic_store_n(c->h_ro_flag, 0u, __ATOMIC_RELAXED);
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
    // narrowed into a pinned buffer of its own and copied asynchronously: the call does not wait for the copy (the kernels
    // that read d_w are ordered behind it on the stream; the buffer is reused only after its event)
    if (c->w_busy) { HIPCHK(c, hipEventSynchronize(c->ev_w)); c->w_busy = false; }
    const size_t n = (size_t)c->ws * fheram_ctx::GLWE;
    if (!narrow(w, c->h_w, n)) return fail(c, FHERAM_ERR_RANGE, "limb out of the normalised range [-2^16, 2^16]");
    HIPCHK(c, hipMemcpyAsync(c->d_w, c->h_w, n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipEventRecord(c->ev_w, c->stream));
    c->w_busy = true;
    c->words_staged = true;
    return FHERAM_OK;
}
// Ram::write, ram.rs:226-294
int fheram_write(fheram_ctx* c, const int64_t* w, int n_w, const fheram_addr* addr) {
    int rc = check_common(c, addr);
    if (rc != FHERAM_OK) return rc;
    if (n_w != c->ws) return fail(c, FHERAM_ERR_INVALID_ARG, "w.len() != subrams.len() (ram.rs:243)");
    if (!c->state) return fail(c, FHERAM_ERR_STATE, "invalid call to Memory.write: internal state is false -> requires calling Memory.read_prepare_write");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->n_shards != 1) return fail(c, FHERAM_ERR_INVALID_ARG, "row-sharded context: use fheram_write_root / fheram_write_shard");
    if (!w && !c->words_staged) return fail(c, FHERAM_ERR_INVALID_ARG, "w == NULL and no staged words");
    // the part of a write that needs no words (trace(ct_hi) of every row, inverse of coordinate 0) is enqueued BEFORE the host
    // narrows the words: the GPU works while the host converts
    if (w && !c->side_begun && !(c->use_graph && !c->profile)) write_side_begin(c, addr);
    if (w) { rc = fheram_word_stage(c, w, n_w); if (rc != FHERAM_OK) { write_side_abort(c); return rc; } }
    rc = run_op(c, addr, 2, [&] {
        if (!c->side_begun) write_side_begin(c, addr);   // (fheram_write_begin may have started it)
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
    GlweRef packed;
    rc = read_local(c, addr, prepare_write != 0, &packed, true);
    if (rc != FHERAM_OK) return rc;
    HIPCHK(c, hipGetLastError());
    if (prepare_write) c->state = true;
    return export_glwes(c, c->d_part, out, out_on_device, (size_t)c->ws);   // device buffers: asynchronous, see fheram_stream_signal
}
int fheram_stream_signal(fheram_ctx* c, void* hip_stream) {
    if (!c) return FHERAM_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipEventRecord(c->ev_xout, c->stream));
    HIPCHK(c, hipStreamWaitEvent((hipStream_t)hip_stream, c->ev_xout, 0));
    return FHERAM_OK;
}
int fheram_stream_wait(fheram_ctx* c, void* hip_stream) {
    if (!c) return FHERAM_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipEventRecord(c->ev_xin, (hipStream_t)hip_stream));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_xin, 0));
    return FHERAM_OK;
}
int fheram_device_malloc(fheram_ctx* c, size_t bytes, void** out) {
    if (!c || !out || bytes == 0) return fail(c, FHERAM_ERR_INVALID_ARG, "bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMalloc(out, bytes));
    return FHERAM_OK;
}
int fheram_device_free(fheram_ctx* c, void* ptr) {
    if (!c) return FHERAM_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipFree(ptr));   // waits for outstanding work on the buffer
    return FHERAM_OK;
}
int fheram_write_begin(fheram_ctx* c, const fheram_addr* addr) {
    int rc = check_common(c, addr);
    if (rc != FHERAM_OK) return rc;
    if (!c->state) return fail(c, FHERAM_ERR_STATE, "invalid call to Memory.write: internal state is false -> requires calling Memory.read_prepare_write");
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->side_begun) write_side_begin(c, addr);
    HIPCHK(c, hipGetLastError());
    return FHERAM_OK;
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
    rc = read_top(c, addr, prepare_write != 0, gathered, ref(c->d_part, (long)fheram_ctx::GLWE, 0));
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
    if (!c->side_begun) write_side_begin(c, addr);   // the root's own rows: overlaps the latency-bound head below
    rc = write_top(c, addr);
    if (rc != FHERAM_OK) return rc;
    HIPCHK(c, hipGetLastError());
    return export_glwes(c, c->d_part, ct_lo_out, out_on_device, (size_t)c->ws);
}
int fheram_write_shard(fheram_ctx* c, const fheram_addr* addr, const void* ct_lo, int on_device) {
    int rc = check_common(c, addr);
    if (rc != FHERAM_OK) return rc;
    if (!ct_lo) return fail(c, FHERAM_ERR_INVALID_ARG, "null ct_lo");
    if (!c->state) return fail(c, FHERAM_ERR_STATE, "invalid call to Memory.write: internal state is false -> requires calling Memory.read_prepare_write");
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->side_begun) write_side_begin(c, addr);   // normally started earlier by fheram_write_begin / fheram_write_root
    rc = import_glwes(c, c->d_part, ct_lo, on_device, (size_t)c->ws);
    if (rc != FHERAM_OK) { write_side_abort(c); return rc; }
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
    if (mode == 0) launch_ks_tr<KS_AUTO>(c, ka, batch, 1);
    else if (mode == 1) launch_ks_tr<KS_ADD>(c, ka, batch, 1);
    else launch_ks_tr<KS_SUBNEG>(c, ka, batch, 1);
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
    c->wide_unsynced = false;   // (t1 was recorded behind everything enqueued on the main stream)
    return check_precision(c);
}
int fheram_profile_enable(fheram_ctx* c, int on) {
    if (!c) return FHERAM_ERR_INVALID_ARG;
    c->profile = on == 2 ? 2 : (on != 0);   // 2: the chain launches only
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
int fheram_mid_stats(fheram_ctx* c, uint64_t* launches, uint64_t* fallbacks) {
    if (!c) return FHERAM_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    unsigned fb[2] = {0, 0};
    HIPCHK(c, hipStreamSynchronize(c->stream2));
    for (int i = 0; i < 2; i++)
        HIPCHK(c, hipMemcpyAsync(&fb[i], c->d_mid_sync[i] + MID_GROUPS_MAX * 32 + 1, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (launches) *launches = c->mid_launches;
    if (fallbacks) *fallbacks = (uint64_t)fb[0] + fb[1];
    return FHERAM_OK;
}
int fheram_mid_state(const fheram_ctx* c, int* enabled, uint64_t* times_disabled) {
    if (!c) return FHERAM_ERR_INVALID_ARG;
    if (enabled) *enabled = c->mid;
    if (times_disabled) *times_disabled = c->mid_disabled_count;
    return FHERAM_OK;
}
int fheram_tail_stats(fheram_ctx* c, uint64_t* launches, uint64_t* fallbacks) {
    if (!c) return FHERAM_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    unsigned fb = 0;
    HIPCHK(c, hipMemcpyAsync(&fb, c->d_tail_sync + TAIL_GROUPS * 32 + 1, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (launches) *launches = c->tail_launches;
    if (fallbacks) *fallbacks = fb;
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
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * (n > 192 ? 192 : n)) == hipSuccess ? 0 : 7;
}
#endif
int fheram_bench_external_product(fheram_ctx* c, int batch, int iters, float* total_ms) {
    if (!c || !total_ms || batch <= 0 || iters <= 0) return fail(c, FHERAM_ERR_INVALID_ARG, "bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t G = fheram_ctx::GLWE;
    DevBuf da, db, dg;
    HIPCHK(c, hipMalloc(&da.p, (size_t)batch * G * 4));
    HIPCHK(c, hipMalloc(&db.p, (size_t)batch * G * 4));
    HIPCHK(c, hipMalloc(&dg.p, fheram_ctx::GGSW * 4));
    {   // synthetic normalised limbs
        std::vector<int32_t> h(std::max((size_t)batch * G, fheram_ctx::GGSW));
        uint64_t x = 0x9E3779B97F4A7C15ull;
        for (auto& v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (int32_t)(x & 0x1FFFF) - 65536; }
        HIPCHK(c, hipMemcpy(da.p, h.data(), (size_t)batch * G * 4, hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(dg.p, h.data(), fheram_ctx::GGSW * 4, hipMemcpyHostToDevice));
    }
    c->cur = c->stream;
    launch_prepare(c, dg.p, c->d_prep, (int)(fheram_ctx::GGSW / N));
    GlweRef ra = ref(da.p, 0, (long)G), rb = ref(db.p, 0, (long)G);
    for (int i = 0; i < 3; i++) { launch_ep(c, ra, rb, c->d_prep, batch, 1); launch_ep(c, rb, ra, c->d_prep, batch, 1); }   // warm-up
    HIPCHK(c, hipEventRecord(c->t0, c->stream));
    for (int i = 0; i < iters; i++) launch_ep(c, (i & 1) ? rb : ra, (i & 1) ? ra : rb, c->d_prep, batch, 1);
    HIPCHK(c, hipEventRecord(c->t1, c->stream));
    HIPCHK(c, hipEventSynchronize(c->t1));
    HIPCHK(c, hipEventElapsedTime(total_ms, c->t0, c->t1));
    HIPCHK(c, hipGetLastError());
    return FHERAM_OK;
}
int fheram_bench_chain(fheram_ctx* c, int kind, int batch, int n, int iters, float* total_ms) {
    if (!c || !total_ms || batch <= 0 || iters <= 0 || n < 1 || n > CHAIN_MAX || kind < 0 || kind > 1) return fail(c, FHERAM_ERR_INVALID_ARG, "bad argument");
    if (kind == 1 && n > c->n_digits) return fail(c, FHERAM_ERR_INVALID_ARG, "a product chain is at most as long as the address has digits (d_prep holds n_digits prepared GGSWs)");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t G = fheram_ctx::GLWE;
    DevBuf da, db, dc, dg;
    HIPCHK(c, hipMalloc(&da.p, (size_t)batch * G * 4));
    HIPCHK(c, hipMalloc(&db.p, (size_t)batch * G * 4));
    HIPCHK(c, hipMalloc(&dc.p, (size_t)batch * G * 4));
    HIPCHK(c, hipMalloc(&dg.p, (size_t)n * fheram_ctx::GGSW * 4));
    {   // synthetic normalised limbs; the trace keys of the context are used as they are (synthetic ones if none were loaded)
        std::vector<int32_t> h(std::max((size_t)batch * G, (size_t)n * fheram_ctx::GGSW));
        uint64_t x = 0x9E3779B97F4A7C15ull;
        for (auto& v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (int32_t)(x & 0x1FFFF) - 65536; }
        HIPCHK(c, hipMemcpy(da.p, h.data(), (size_t)batch * G * 4, hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(dg.p, h.data(), (size_t)n * fheram_ctx::GGSW * 4, hipMemcpyHostToDevice));
        if (kind == 0 && !c->keys_loaded)
            for (int i = 0; i < LOGN; i++) { c->cur = c->stream; launch_prepare(c, dg.p, c->d_atk + (size_t)i * c->atk, (int)(c->atk / N), c->gal[i]); }
    }
    c->cur = c->stream;
    if (kind == 1) launch_prepare(c, dg.p, c->d_prep, n * (int)(fheram_ctx::GGSW / N));   // (n <= n_digits: checked above)
    GlweRef ra = ref(da.p, 0, (long)G), rb = ref(db.p, 0, (long)G), rc = ref(dc.p, 0, (long)G);
    auto once = [&](int i) {
        GlweRef src = (i & 1) ? rb : ra, dst = (i & 1) ? ra : rb;
        if (kind == 0) trace_steps(c, src, dst, rc, 0, n, batch, 1);
        else ep_chain(c, src, dst, rc, c->d_prep, n, batch, 1);
    };
    for (int i = 0; i < 4; i++) once(i);   // warm-up
    HIPCHK(c, hipEventRecord(c->t0, c->stream));
    for (int i = 0; i < iters; i++) once(i);
    HIPCHK(c, hipEventRecord(c->t1, c->stream));
    HIPCHK(c, hipEventSynchronize(c->t1));
    HIPCHK(c, hipEventElapsedTime(total_ms, c->t0, c->t1));
    HIPCHK(c, hipGetLastError());
    return FHERAM_OK;
}
int fheram_device_info(const fheram_ctx* c, char* name, size_t name_len, int* cus) {
    if (!c) return FHERAM_ERR_INVALID_ARG;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c->device) != hipSuccess) return FHERAM_ERR_DEVICE;
    if (name && name_len) { std::snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName); }
    if (cus) *cus = prop.multiProcessorCount;
    return FHERAM_OK;
}

}  // extern "C"

#include "setup.hpp"
#include "selftest.hpp"
#include "group.hpp"
