// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see oracle/README.md).
//
// CPU restatement of the phantomzone-org/fhe-ram read / read_prepare_write / write path
// (reference @ /root/reference, snapshot 2026-02-13) together with the Poulpy GLWE / GGSW /
// GGLWE operations that path reaches (poulpy-core 0.3.2, poulpy-hal 0.3.2, un-vendored:
// Cargo.toml:7-10, Cargo.lock:376-435).  Each function cites the reference file:line it
// follows; routines whose body lives in Poulpy are marked [UPSTREAM-RECALL] and restate the
// published algorithm.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this code.
#pragma once
#include <atomic>
#include <exception>
#include "znx.hpp"
#include "ntt.hpp"
#include <algorithm>
#include <cmath>
#include <map>
#include <string>

namespace fo {

// =======================================================================================
// base.rs — digit plans (pure integers; the only component with exact KATs, base.rs:110-439)
// =======================================================================================
struct Base1D {
    std::vector<uint8_t> d;                                     // base.rs:3
    size_t size() const { return d.size(); }                    // base.rs:6
    size_t max() const { size_t m = 1; for (auto b : d) m <<= b; return m; }   // base.rs:10-14
    size_t gap(size_t log_n) const { size_t g = log_n; for (auto b : d) g >>= b; return (size_t)1 << g; }  // base.rs:17-21
    std::vector<uint8_t> decomp(uint32_t value) const {         // base.rs:24-33
        std::vector<uint8_t> out; unsigned sum = 0;
        for (auto b : d) { out.push_back((uint8_t)((value >> sum) & ((1u << b) - 1))); sum += b; }
        return out;
    }
    uint32_t recomp(const std::vector<uint8_t>& dec) const {    // base.rs:36-45
        uint32_t v = 0; unsigned sum = 0;
        for (size_t i = 0; i < d.size(); i++) { v |= ((uint32_t)dec[i]) << sum; sum += d[i]; }
        return v;
    }
    bool operator==(const Base1D& o) const { return d == o.d; }
};
struct Base2D {
    std::vector<Base1D> v;                                      // base.rs:49
    size_t max_len() const { size_t m = 0; for (auto& b : v) m = std::max(m, b.size()); return m; }  // base.rs:52-58
    Base1D as_1d() const { Base1D r; for (auto& b : v) for (auto x : b.d) r.d.push_back(x); return r; }  // base.rs:64-71
    size_t max() const { return as_1d().max(); }                // base.rs:60
    std::vector<uint8_t> decomp(uint32_t value) const { return as_1d().decomp(value); }
    uint32_t recomp(const std::vector<uint8_t>& dec) const { return as_1d().recomp(dec); }
};
// base.rs:84-108
static inline Base2D get_base_2d(uint32_t value, const std::vector<uint8_t>& base) {
    Base2D out;
    uint32_t x = value - 1;
    uint32_t bits = 0;
    while (x) { bits++; x >>= 1; }                              // 32 - leading_zeros(value-1)
    while (bits != 0) {
        Base1D v;
        for (auto b : base) {
            if ((uint32_t)b <= bits) { v.d.push_back(b); bits -= b; }
            else { if (bits != 0) { v.d.push_back((uint8_t)bits); bits = 0; } break; }
        }
        out.v.push_back(v);
    }
    return out;
}
// lib.rs:23-26
static inline size_t reverse_bits_msb(size_t x, uint32_t n) {
    size_t r = 0;
    for (uint32_t i = 0; i < n; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}

// =======================================================================================
// parameters.rs — constants and layouts
// =======================================================================================
struct Params {
    int log_n = 12, base2k = 17, rank = 1;                       // parameters.rs:11-13
    int k_glwe_pt = 3, k_glwe_ct = 51, k_ggsw_addr = 68;         // parameters.rs:14-16
    int k_evk_trace = 68, k_evk_ggsw_inv = 85;                   // parameters.rs:17-18
    std::vector<uint8_t> decomp_n = {3, 3, 3, 3};                // parameters.rs:19
    int word_size = 4;                                           // parameters.rs:20
    size_t max_addr = (size_t)1 << 14;                           // parameters.rs:21
    double sigma = 3.2;                                          // README.md:25 (Poulpy SIGMA)

    int n() const { return 1 << log_n; }
    static int ceil_div(int a, int b) { return (a + b - 1) / b; }
    int size_pt() const { return ceil_div(k_glwe_pt, base2k); }
    int size_ct() const { return ceil_div(k_glwe_ct, base2k); }       // glwe_ct_infos parameters.rs:62-69
    int size_addr() const { return ceil_div(k_ggsw_addr, base2k); }   // ggsw_infos parameters.rs:95-104
    int size_evk_trace() const { return ceil_div(k_evk_trace, base2k); }   // evk_glwe_infos :71-81
    int size_evk_inv() const { return ceil_div(k_evk_ggsw_inv, base2k); }  // evk_ggsw_infos :83-93
    int dnum_ct() const { return ceil_div(k_glwe_ct, base2k); }       // parameters.rs:138-140
    int dnum_ggsw() const { return ceil_div(k_ggsw_addr, base2k); }   // parameters.rs:142-144
    Base2D base2d() const { return get_base_2d((uint32_t)max_addr, decomp_n); }  // parameters.rs:285-287
    size_t rows() const { return (max_addr + n() - 1) / n(); }
    // element counts (i64) of the host layouts, SURVEY.md A.2
    size_t glwe_len(int size) const { return (size_t)size * 2 * n(); }
    size_t ggsw_len() const { return (size_t)dnum_ct() * 2 * glwe_len(size_addr()); }
    size_t atk_trace_len() const { return (size_t)dnum_ct() * glwe_len(size_evk_trace()); }
    size_t evk_inv_len() const { return (size_t)dnum_ggsw() * glwe_len(size_evk_inv()); }
};

// GLWE::trace_galois_elements [UPSTREAM-RECALL] (used at keys.rs:39,158): g_0 = -1,
// g_i = 5^(2^(i-1)) mod 2n.
static inline int64_t galois_element(int log_n, int i) {
    if (i == 0) return -1;
    const int64_t two_n = (int64_t)2 << log_n;
    int64_t g = 5, e = (int64_t)1 << (i - 1), r = 1;
    while (e) { if (e & 1) r = r * g % two_n; g = g * g % two_n; e >>= 1; }
    return r;
}
static inline int64_t galois_inverse(int log_n, int64_t p) {
    const int64_t two_n = (int64_t)2 << log_n;
    int64_t a = ((p % two_n) + two_n) % two_n, r = 1, e = two_n / 2 - 1;  // group exponent divides n
    // a^(n-1) = a^-1 since a^n = 1 for odd a mod 2n (n >= 2)
    int64_t b = a;
    while (e) { if (e & 1) r = r * b % two_n; b = b * b % two_n; e >>= 1; }
    return r;
}

// =======================================================================================
// Context: parameters + transform tables + exactness statistics
// =======================================================================================
struct Ctx {
    Params p;
    Ntt ntt;
    // statistics are atomics: the all-core variant (threads > 1, below) updates them from several threads
    std::atomic<int64_t> max_big{0};   // largest |coefficient| of any post-inverse-transform value (A.9)
    std::atomic<uint64_t> n_ep{0}, n_ks{0}, n_prepare{0};
    // threads > 1: the all-core CPU baseline.  Sub-RAMs are independent (ram.rs:187-190 maps
    // over them one after the other) and so are the rows inside the per-row loops (ram.rs:429-434, 502-504,
    // 612-630, 644-646); OpenMP runs those loops in parallel, and the packing runs level by level with the leaves of a
    // level in parallel (pack_level_sync).  Results are identical to threads = 1 (every ciphertext sees the same
    // operations; tests/test_oracle.py, and the committed digests of the full sizes are produced this way).
    int threads = 1;
    explicit Ctx(const Params& p_) : p(p_), ntt(p_.log_n) {}
    int n() const { return p.n(); }
    void note_big(int64_t mx) {
        int64_t cur = max_big.load(std::memory_order_relaxed);
        while (mx > cur && !max_big.compare_exchange_weak(cur, mx, std::memory_order_relaxed)) {}
    }
    int outer_threads() const { return std::max(1, std::min(threads, (int)p.word_size)); }
    int inner_threads() const { return std::max(1, threads / outer_threads()); }
};
static constexpr int64_t BIG_BOUND = (int64_t)1 << 47;   // HIP prime 2^48+57345 needs |big| < p/2

static inline VecView glwe_view(int64_t* p, int n, int size) { return VecView{p, n, 2, size}; }

// Prepared matrix (VmpPMat stand-in): polys indexed [row][col_in][limb][col_out].
struct MatPrepared {
    int rows = 0, cols_in = 0, size = 0;
    std::vector<PolyHat> polys;
    const PolyHat& at(int row, int cin, int limb, int cout) const {
        return polys[(((size_t)row * cols_in + cin) * size + limb) * 2 + cout];
    }
};
struct KeyPrepared { MatPrepared m; int64_t p = 0; };

// GGSWPrepared::prepare / GLWEAutomorphismKeyPrepared::prepare / GGLWEToGGSWKeyPrepared::prepare
// [UPSTREAM-RECALL: vmp_prepare = forward transform of every polynomial]
// (reference call sites: coordinate_prepared.rs:114,139; keys.rs:64-70).
static inline MatPrepared prepare_mat(Ctx& c, const int64_t* mat, int rows, int cols_in, int size) {
    MatPrepared m; m.rows = rows; m.cols_in = cols_in; m.size = size;
    const int n = c.n();
    m.polys.resize((size_t)rows * cols_in * size * 2);
    for (size_t i = 0; i < m.polys.size(); i++) to_hat_prepared(c.ntt, mat + i * n, m.polys[i]);
    c.n_prepare++;
    return m;
}

// vmp_apply_dft_to_dft + idft [UPSTREAM-RECALL]: big[cout][limb] = sum_{cin,row} a^[cin][row] (.) M[row][cin][limb][cout]
// a_polys[cin][row] are small (limb) polynomials.
static inline void vmp_to_big(Ctx& c, Big& big, const std::vector<std::vector<const int64_t*>>& a_polys,
                              const MatPrepared& M) {
    const int n = c.n();
    const int cols_in = (int)a_polys.size();
    assert(cols_in == M.cols_in && big.size == M.size);
    std::vector<std::vector<std::vector<uint64_t>>> ah(cols_in);
    for (int ci = 0; ci < cols_in; ci++) {
        ah[ci].resize(a_polys[ci].size());
        for (size_t r = 0; r < a_polys[ci].size(); r++) to_hat(c.ntt, a_polys[ci][r], ah[ci][r]);
    }
    std::vector<uint64_t> acc(n);
    for (int co = 0; co < 2; co++)
        for (int j = 0; j < M.size; j++) {
            std::fill(acc.begin(), acc.end(), 0);
            for (int ci = 0; ci < cols_in; ci++)
                for (size_t r = 0; r < a_polys[ci].size(); r++) mac_hat(c.ntt, acc, ah[ci][r], M.at((int)r, ci, j, co));
            int64_t mx = from_hat(c.ntt, acc, big.at(co, j));
            c.note_big(mx);
        }
}

static inline void big_normalize_col(Ctx& c, const VecView& res, int col, const Big& big, int bcol) {
    std::vector<int64_t*> rp(res.size);
    std::vector<const int64_t*> ap(big.size);
    for (int j = 0; j < res.size; j++) rp[j] = res.at(col, j);
    for (int j = 0; j < big.size; j++) ap[j] = big.at(bcol, j);
    big_normalize(c.p.base2k, res.n, rp.data(), res.size, ap.data(), big.size);
}

// ---------------------------------------------------------------------------------------
// glwe_external_product(_inplace) [UPSTREAM-RECALL, SURVEY.md A.4]
// reference call sites: coordinate_prepared.rs:156 (out of place), :158,:175 (in place)
// ---------------------------------------------------------------------------------------
static inline void glwe_external_product(Ctx& c, const VecView& res, const VecView& a, const MatPrepared& ggsw) {
    const int rows_used = std::min(a.size, ggsw.rows);
    std::vector<std::vector<const int64_t*>> ap(2);
    for (int ci = 0; ci < 2; ci++) for (int r = 0; r < rows_used; r++) ap[ci].push_back(a.at(ci, r));
    Big big(c.n(), 2, ggsw.size);
    vmp_to_big(c, big, ap, ggsw);
    for (int co = 0; co < 2; co++) big_normalize_col(c, res, co, big, co);
    c.n_ep++;
}

// ---------------------------------------------------------------------------------------
// glwe_keyswitch_internal [UPSTREAM-RECALL, SURVEY.md A.6]: transform the mask column, multiply
// by the GGLWE key, inverse transform, add the body limbs (vec_znx_big_add_small_inplace).
// ---------------------------------------------------------------------------------------
static inline void keyswitch_internal(Ctx& c, Big& big, const VecView& a, const MatPrepared& key) {
    const int rows_used = std::min(a.size, key.rows);
    std::vector<std::vector<const int64_t*>> ap(1);
    for (int r = 0; r < rows_used; r++) ap[0].push_back(a.at(1, r));
    vmp_to_big(c, big, ap, key);
    const int m = std::min(a.size, big.size);
    for (int j = 0; j < m; j++) {
        int64_t* b = big.at(0, j); const int64_t* s = a.at(0, j);
        for (int i = 0; i < a.n; i++) b[i] += s[i];
    }
    c.n_ks++;
}
static inline void big_automorphism_inplace(Ctx& c, Big& big, int64_t p) {
    std::vector<int64_t> tmp(c.n());
    for (int co = 0; co < big.cols; co++) for (int j = 0; j < big.size; j++) {
        poly_automorphism(c.n(), p, tmp.data(), big.at(co, j));
        std::memcpy(big.at(co, j), tmp.data(), sizeof(int64_t) * c.n());
    }
}
// glwe_automorphism: res = normalize(phi_p(KS(a)))          (GLWEPacker combine, ram.rs:435,514)
static inline void glwe_automorphism(Ctx& c, const VecView& res, const VecView& a, const KeyPrepared& key) {
    Big big(c.n(), 2, key.m.size);
    keyswitch_internal(c, big, a, key.m);
    big_automorphism_inplace(c, big, key.p);
    for (int co = 0; co < 2; co++) big_normalize_col(c, res, co, big, co);
}
// glwe_automorphism_add_inplace: res = normalize(phi_p(KS(res)) + res)   (trace, ram.rs:457,540,572,616,621)
static inline void glwe_automorphism_add_inplace(Ctx& c, const VecView& res, const KeyPrepared& key) {
    Big big(c.n(), 2, key.m.size);
    keyswitch_internal(c, big, res, key.m);
    big_automorphism_inplace(c, big, key.p);
    const int m = std::min(res.size, big.size);
    for (int co = 0; co < 2; co++) {
        for (int j = 0; j < m; j++) {
            int64_t* b = big.at(co, j); const int64_t* s = res.at(co, j);
            for (int i = 0; i < res.n; i++) b[i] += s[i];
        }
        big_normalize_col(c, res, co, big, co);
    }
}
// res = normalize(a - phi_p(KS(a)))   (GLWEPacker combine, "accumulator empty" branch)
static inline void glwe_automorphism_sub_negate(Ctx& c, const VecView& res, const VecView& a, const KeyPrepared& key) {
    Big big(c.n(), 2, key.m.size);
    keyswitch_internal(c, big, a, key.m);
    big_automorphism_inplace(c, big, key.p);
    const int m = std::min(a.size, big.size);
    for (int co = 0; co < 2; co++) {
        for (int j = 0; j < big.size; j++) {
            int64_t* b = big.at(co, j);
            for (int i = 0; i < a.n; i++) b[i] = (j < m ? a.at(co, j)[i] : 0) - b[i];
        }
        big_normalize_col(c, res, co, big, co);
    }
}

// ---------------------------------------------------------------------------------------
// Elementwise GLWE ops
// ---------------------------------------------------------------------------------------
static inline void glwe_copy(const VecView& r, const VecView& a) { std::memcpy(r.p, a.p, sizeof(int64_t) * a.len()); }   // ram.rs:526,535,537
static inline void glwe_add_inplace(const VecView& r, const VecView& a) { for (size_t i = 0; i < r.len(); i++) r.p[i] += a.p[i]; }  // ram.rs:575,625
static inline void glwe_sub_inplace(const VecView& r, const VecView& a) { for (size_t i = 0; i < r.len(); i++) r.p[i] -= a.p[i]; }  // ram.rs:574,617
static inline void glwe_sub(const VecView& r, const VecView& a, const VecView& b) { for (size_t i = 0; i < r.len(); i++) r.p[i] = a.p[i] - b.p[i]; }
static inline void glwe_normalize_inplace(Ctx& c, const VecView& r) { for (int co = 0; co < r.cols; co++) normalize_inplace(c.p.base2k, r, co); }  // ram.rs:576,626
static inline void glwe_rsh(Ctx& c, int k, const VecView& r) { for (int co = 0; co < r.cols; co++) rsh_inplace(c.p.base2k, k, r, co); }
static inline void glwe_rotate(Ctx& c, int64_t k, const VecView& r, const VecView& a) {
    for (int j = 0; j < a.size; j++) for (int co = 0; co < a.cols; co++) poly_rotate(c.n(), k, r.at(co, j), a.at(co, j));
}
static inline void glwe_rotate_inplace(Ctx& c, int64_t k, const VecView& r) {   // ram.rs:629
    std::vector<int64_t> tmp(r.len());
    VecView t{tmp.data(), r.n, r.cols, r.size};
    glwe_rotate(c, k, t, r);
    glwe_copy(r, t);
}

// ---------------------------------------------------------------------------------------
// Evaluation keys (keys.rs:27-31) in prepared form.
// ---------------------------------------------------------------------------------------
struct EvaluationKeysPrepared {
    std::map<int64_t, KeyPrepared> atk_glwe;   // keys.rs:28
    KeyPrepared atk_ggsw_inv;                  // keys.rs:29
    MatPrepared tsk_ggsw_inv;                  // keys.rs:30
};

// GLWE::trace_inplace(start, end) [UPSTREAM-RECALL, SURVEY.md A.7]   (ram.rs:457,540; trace: :572,616,621)
static inline void glwe_trace_inplace(Ctx& c, const VecView& res, int start, int end,
                                      const std::map<int64_t, KeyPrepared>& keys) {
    for (int i = start; i < end; i++) {
        glwe_rsh(c, 1, res);
        int64_t p = galois_element(c.p.log_n, i);
        auto it = keys.find(p);
        if (it == keys.end()) throw std::runtime_error("trace: missing automorphism key");
        glwe_automorphism_add_inplace(c, res, it->second);
    }
}

// ---------------------------------------------------------------------------------------
// GLWEPacker(log_batch = 0) [UPSTREAM-RECALL, SURVEY.md A.7]   (ram.rs:329,435-448,514-521)
// ---------------------------------------------------------------------------------------
struct Accumulator { std::vector<int64_t> data; bool value = false, control = false; };
struct Packer {
    std::vector<Accumulator> accs;
    size_t counter = 0;
    int n = 0, size = 0;
    void alloc(const Ctx& c, int size_) {
        n = c.n(); size = size_;
        accs.assign(c.p.log_n, Accumulator());
        for (auto& a : accs) a.data.assign((size_t)size * 2 * n, 0);
    }
    void reset() { for (auto& a : accs) { a.value = false; a.control = false; } counter = 0; }
};
static inline void packer_combine(Ctx& c, Packer& pk, Accumulator& acc, const int64_t* b, int i,
                                  const std::map<int64_t, KeyPrepared>& keys) {
    const int log_n = c.p.log_n, n = c.n();
    VecView a = glwe_view(acc.data.data(), n, pk.size);
    const int64_t gal = galois_element(log_n, i);
    const int64_t t = (int64_t)1 << (log_n - i - 1);
    auto it = keys.find(gal);
    if (acc.value) {
        if (b) {
            if (it == keys.end()) throw std::runtime_error("pack: missing automorphism key");
            VecView bv = glwe_view(const_cast<int64_t*>(b), n, pk.size);
            std::vector<int64_t> tmp(a.len());
            VecView tb = glwe_view(tmp.data(), n, pk.size);
            glwe_rotate_inplace(c, -t, a);          // a = a * X^-t
            glwe_sub(tb, a, bv);                    // tmp_b = a*X^-t - b
            glwe_rsh(c, 1, tb);
            glwe_add_inplace(a, bv);                // a = a*X^-t + b
            glwe_rsh(c, 1, a);
            glwe_normalize_inplace(c, tb);
            glwe_automorphism(c, tb, tb, it->second);   // tmp_b = phi(a*X^-t - b)
            glwe_sub_inplace(a, tb);
            glwe_normalize_inplace(c, a);
            glwe_rotate_inplace(c, t, a);           // a = a + b*X^t + phi(a - b*X^t)
        } else {
            if (it == keys.end()) throw std::runtime_error("pack: missing automorphism key");
            glwe_rsh(c, 1, a);
            glwe_automorphism_add_inplace(c, a, it->second);   // a = a + phi(a)
        }
    } else if (b) {
        if (it == keys.end()) throw std::runtime_error("pack: missing automorphism key");
        VecView bv = glwe_view(const_cast<int64_t*>(b), n, pk.size);
        std::vector<int64_t> tmp(a.len());
        VecView tb = glwe_view(tmp.data(), n, pk.size);
        glwe_rotate(c, t, tb, bv);
        glwe_rsh(c, 1, tb);
        glwe_automorphism_sub_negate(c, a, tb, it->second);    // a = b*X^t - phi(b*X^t)
        acc.value = true;
    }
}
static inline void pack_core(Ctx& c, Packer& pk, const int64_t* a, int i, const std::map<int64_t, KeyPrepared>& keys) {
    if (i == c.p.log_n) return;
    Accumulator& acc = pk.accs[i];
    if (!acc.control) {
        if (a) { std::memcpy(acc.data.data(), a, sizeof(int64_t) * acc.data.size()); acc.value = true; }
        else acc.value = false;
        acc.control = true;
    } else {
        packer_combine(c, pk, acc, a, i, keys);
        acc.control = false;
        if (acc.value) pack_core(c, pk, acc.data.data(), i + 1, keys);
        else pack_core(c, pk, nullptr, i + 1, keys);
    }
}
static inline void packer_add(Ctx& c, Packer& pk, const int64_t* a, const std::map<int64_t, KeyPrepared>& keys) {
    if (pk.counter >= (size_t)c.n()) throw std::runtime_error("packing limit reached");
    pack_core(c, pk, a, 0, keys);
    pk.counter += 1;
}
static inline void packer_flush(Ctx& c, Packer& pk, int64_t* res) {
    if (pk.counter != (size_t)c.n()) throw std::runtime_error("packer flush: counter != n");
    std::memcpy(res, pk.accs[c.p.log_n - 1].data.data(), sizeof(int64_t) * pk.accs[0].data.size());
    pk.reset();
}

// ---------------------------------------------------------------------------------------
// GGSW::automorphism(p = -1) with tensor-key row expansion [UPSTREAM-RECALL, SURVEY.md A.8]
// (coordinate_prepared.rs:138).  in/out: std-form GGSW [row][col_in][limb][col_out][n].
// ---------------------------------------------------------------------------------------
static inline void ggsw_automorphism(Ctx& c, int64_t* res, const int64_t* a, int dnum, int size,
                                     const KeyPrepared& atk, const MatPrepared& tsk) {
    const int n = c.n();
    const size_t glen = (size_t)size * 2 * n;
    for (int row = 0; row < dnum; row++) {
        VecView r0 = glwe_view(res + ((size_t)row * 2 + 0) * glen, n, size);
        VecView a0 = glwe_view(const_cast<int64_t*>(a) + ((size_t)row * 2 + 0) * glen, n, size);
        glwe_automorphism(c, r0, a0, atk);
    }
    // ggsw_expand_row (rank 1): res[row][1] = normalize( vmp(res[row][0].mask, tsk) + (0, res[row][0].body) )
    for (int row = 0; row < dnum; row++) {
        VecView r0 = glwe_view(res + ((size_t)row * 2 + 0) * glen, n, size);
        VecView r1 = glwe_view(res + ((size_t)row * 2 + 1) * glen, n, size);
        const int rows_used = std::min(size, tsk.rows);
        std::vector<std::vector<const int64_t*>> ap(1);
        for (int r = 0; r < rows_used; r++) ap[0].push_back(r0.at(1, r));
        Big big(n, 2, tsk.size);
        vmp_to_big(c, big, ap, tsk);
        const int m = std::min(size, big.size);
        for (int j = 0; j < m; j++) {
            int64_t* b = big.at(1, j); const int64_t* s = r0.at(0, j);
            for (int i = 0; i < n; i++) b[i] += s[i];
        }
        for (int co = 0; co < 2; co++) big_normalize_col(c, r1, co, big, co);
        c.n_ks++;
    }
}

// All-core variant of the packing (Ctx::threads > 1): the same combines as the sequential GLWEPacker fed in
// bit-reversed order, scheduled level by level so that the leaves of a level run in parallel.  With count leaves
// (k = ceil(log2 count), L0 = log N - k): every leaf first goes alone through packer levels 0..L0-1
// (combine(acc, None, i)); then pairing round m joins leaf q with leaf q + 2^(k-1-m) at packer level L0 + m, a leaf
// without partner goes through that level alone.  (This is the schedule of the HIP path, csrc/launch.hpp pack_levels;
// tests/test_oracle.py checks it against the sequential packer, ragged counts included.)  leaves[q] is consumed.
static inline void pack_level_sync(Ctx& c, std::vector<std::vector<int64_t>>& leaves, int size, int64_t* res,
                                   const std::map<int64_t, KeyPrepared>& keys, int nth) {
    const int log_n = c.p.log_n;
    const size_t count = leaves.size();
    int k = 0; while (((size_t)1 << k) < count) k++;
    const int L0 = log_n - k;
    Packer dummy; dummy.n = c.n(); dummy.size = size;
    auto combine = [&](std::vector<int64_t>& a, const int64_t* b, int level) {
        Accumulator acc; acc.data.swap(a); acc.value = true; acc.control = true;
        packer_combine(c, dummy, acc, b, level, keys);
        a.swap(acc.data);
    };
#pragma omp parallel for num_threads(nth) schedule(dynamic)
    for (size_t q = 0; q < count; q++)
        for (int i = 0; i < L0; i++) combine(leaves[q], nullptr, i);
    size_t live = count;
    for (int m = 0; m < k; m++) {
        const int level = L0 + m;
        const size_t h = (size_t)1 << (k - 1 - m);
        const size_t top = std::min(h, live);
#pragma omp parallel for num_threads(nth) schedule(dynamic)
        for (size_t q = 0; q < top; q++) combine(leaves[q], (q + h < live) ? leaves[q + h].data() : nullptr, level);
        live = top;
    }
    std::memcpy(res, leaves[0].data(), sizeof(int64_t) * leaves[0].size());
}

// =======================================================================================
// coordinate.rs / address.rs / coordinate_prepared.rs
// =======================================================================================
struct Coordinate {                       // coordinate.rs:22-25
    std::vector<std::vector<int64_t>> value;   // std-form GGSW per digit
    Base1D base1d;
};
struct Address {                          // address.rs:21-24
    std::vector<Coordinate> coordinates;
    Base2D base2d;
    size_t n2() const { return coordinates.size(); }             // address.rs:113
    const Coordinate& at(size_t i) const { return coordinates[i]; }   // address.rs:117
};
struct CoordinatePrepared {               // coordinate_prepared.rs:16-19
    std::vector<MatPrepared> value;
    Base1D base1d;
};
// coordinate_prepared.rs:104-116
static inline void coordinate_prepare(Ctx& c, CoordinatePrepared& self, const Coordinate& other) {
    self.base1d = other.base1d;
    self.value.clear();
    for (auto& g : other.value) self.value.push_back(prepare_mat(c, g.data(), c.p.dnum_ct(), 2, c.p.size_addr()));
}
// coordinate_prepared.rs:121-142
static inline void coordinate_prepare_inv(Ctx& c, CoordinatePrepared& self, const Coordinate& other,
                                          const KeyPrepared& auto_key, const MatPrepared& tensor_key) {
    if (auto_key.p != -1) throw std::runtime_error("prepare_inv: auto_key.p() != -1");   // :134
    self.base1d = other.base1d;
    self.value.clear();
    std::vector<int64_t> tmp(c.p.ggsw_len());
    for (auto& g : other.value) {
        ggsw_automorphism(c, tmp.data(), g.data(), c.p.dnum_ct(), c.p.size_addr(), auto_key, tensor_key);   // :138
        self.value.push_back(prepare_mat(c, tmp.data(), c.p.dnum_ct(), 2, c.p.size_addr()));               // :139
    }
}
// coordinate_prepared.rs:147-161
static inline void coordinate_product(Ctx& c, const CoordinatePrepared& self, const VecView& res, const VecView& a) {
    for (size_t i = 0; i < self.value.size(); i++) {
        if (i == 0) glwe_external_product(c, res, a, self.value[i]);
        else glwe_external_product(c, res, res, self.value[i]);
    }
}
// coordinate_prepared.rs:164-177
static inline void coordinate_product_inplace(Ctx& c, const CoordinatePrepared& self, const VecView& res) {
    for (auto& g : self.value) glwe_external_product(c, res, res, g);
}

// =======================================================================================
// conversion.rs:18-82 — Address::set_from_fheuint
// SELF-CONSISTENT RESTATEMENT, UNPINNED EVEN FUNCTIONALLY AGAINST UPSTREAM: the reference delegates to
// poulpy-schemes' `scalar_to_ggsw_blind_rotation` (tfhe::bdd_arithmetic, un-vendored, conversion.rs:51-60).  What is
// restated is its contract (conversion.rs:41-65 and the test at :100-220): digit d of the address becomes a GGSW of
//   X^{ +-(((k >> bit_rsh) mod 2^bit_mask) << bit_lsh) }          (sign = true: +, as the reference's test expects)
// derived from the encrypted bits of k without any evaluation key — hence by CMux chains on noiseless rows:
//   row (r, col_in c) starts as the trivial encryption of 2^{-(r+1) base2k} (c = 0: in the body; c = 1: in the mask
//   column, whose phase is m * s), and for bit i of the digit
//       acc <- normalize(acc + normalize(X^{+-2^(i+lsh)} * acc - acc) (x) GGSW(b_i)).
// FheUintPrepared<u32> = one GGSW per bit (layout here: k_evk_ggsw_inv / dnum_ggsw, so that all 4 limbs of an address
// row are decomposed).  Bit order, CMux form and normalisation points are this restatement's choices.
// =======================================================================================
struct FheUint {
    std::vector<MatPrepared> bits;
    int size = 0, dnum = 0;
};
static inline void fheuint_prepare(Ctx& c, FheUint& fu, const int64_t* bits_std, int n_bits) {
    fu.size = c.p.size_evk_inv(); fu.dnum = c.p.dnum_ggsw();
    const size_t glen = (size_t)fu.dnum * 2 * c.p.glwe_len(fu.size);
    fu.bits.clear();
    for (int i = 0; i < n_bits; i++) fu.bits.push_back(prepare_mat(c, bits_std + (size_t)i * glen, fu.dnum, 2, fu.size));
}
// conversion.rs:41-65
static inline void address_set_from_fheuint(Ctx& c, Address& res, const FheUint& fu, bool sign) {
    const int n = c.n(), S = c.p.size_addr(), D = c.p.dnum_ct();
    const size_t glen = c.p.glwe_len(S);
    res.base2d = c.p.base2d();
    res.coordinates.clear();
    size_t bit_rsh = 0;
    for (auto& base1d : res.base2d.v) {                                               // :45
        Coordinate co; co.base1d = base1d;
        size_t bit_lsh = 0;                                                           // :46
        for (uint8_t bit_mask : base1d.d) {                                           // :49
            std::vector<int64_t> ggsw(c.p.ggsw_len(), 0);
            for (int r = 0; r < D; r++) for (int ci = 0; ci < 2; ci++) {
                int64_t* row = ggsw.data() + ((size_t)r * 2 + ci) * glen;
                VecView acc = glwe_view(row, n, S);
                acc.at(ci, r)[0] = 1;                                                 // test_vector = X^0 (:42-43), gadget row r
                std::vector<int64_t> t(glen), e(glen);
                VecView tv = glwe_view(t.data(), n, S), ev = glwe_view(e.data(), n, S);
                for (size_t i = 0; i < bit_mask; i++) {
                    if (bit_rsh + i >= fu.bits.size()) throw std::runtime_error("set_from_fheuint: address wider than the integer");
                    const int64_t step = (int64_t)1 << (i + bit_lsh);
                    glwe_rotate(c, sign ? step : -step, tv, acc);
                    glwe_sub_inplace(tv, acc);
                    glwe_normalize_inplace(c, tv);
                    glwe_external_product(c, ev, tv, fu.bits[bit_rsh + i]);
                    glwe_add_inplace(acc, ev);
                    glwe_normalize_inplace(c, acc);
                }
            }
            co.value.push_back(std::move(ggsw));
            bit_lsh += bit_mask;                                                      // :61
            bit_rsh += bit_mask;                                                      // :62
        }
        res.coordinates.push_back(std::move(co));
    }
}

// =======================================================================================
// ram.rs — SubRam / Ram
// =======================================================================================
struct SubRam {                                                  // ram.rs:298-303
    std::vector<std::vector<int64_t>> data;
    std::vector<std::vector<std::vector<int64_t>>> tree;
    Packer packer;
    bool state = false;
};
struct Ram {                                                     // ram.rs:25-29
    Ctx* c;
    std::vector<SubRam> subrams;
    explicit Ram(Ctx* c_) : c(c_) {
        subrams.resize(c->p.word_size);
        for (auto& s : subrams) alloc(s);
    }
    size_t glen() const { return c->p.glwe_len(c->p.size_ct()); }
    VecView view(std::vector<int64_t>& v) const { return glwe_view(v.data(), c->n(), c->p.size_ct()); }

    void alloc(SubRam& s) {                                      // ram.rs:306-332
        const size_t n = c->n();
        size_t max_addr_split = c->p.max_addr;
        s.tree.clear();
        if (max_addr_split > n) {
            size_t size = (max_addr_split + n - 1) / n;
            while (size != 1) {
                size = (size + n - 1) / n;
                s.tree.push_back(std::vector<std::vector<int64_t>>(size, std::vector<int64_t>(glen(), 0)));
            }
        }
        s.packer.alloc(*c, c->p.size_ct());
        s.state = false;
    }

    // SubRam::read, ram.rs:382-459
    void subram_read(SubRam& s, const Address& address, const EvaluationKeysPrepared& keys, int64_t* out) {
        if (s.state) throw std::runtime_error("invalid call to Memory.read: internal state is true -> requires calling Memory.write");   // :393-396
        const int log_n = c->p.log_n; const size_t n = c->n();
        std::vector<std::vector<int64_t>> results;
        std::vector<int64_t> tmp_ct(glen(), 0);
        for (size_t i = 0; i < address.n2(); i++) {
            const Coordinate& coordinate = address.at(i);
            std::vector<std::vector<int64_t>>& res_prev = (i == 0) ? s.data : results;
            CoordinatePrepared cp;
            coordinate_prepare(*c, cp, coordinate);                                   // :416-419
            if (i < address.n2() - 1) {
                const size_t total = res_prev.size();
                std::vector<std::vector<int64_t>> prod;   // all-core variant: the rows' products, computed in parallel up front
                const int nth = c->inner_threads();
                if (nth > 1) {
                    prod.assign(total, std::vector<int64_t>(glen()));
#pragma omp parallel for num_threads(nth) schedule(dynamic)
                    for (size_t r = 0; r < total; r++) coordinate_product(*c, cp, view(prod[r]), view(res_prev[r]));
                    if (total <= n) {   // one chunk (always, for max_addr <= N^2): pack level by level, in parallel
                        pack_level_sync(*c, prod, c->p.size_ct(), tmp_ct.data(), keys.atk_glwe, nth);
                        results.push_back(tmp_ct);
                        continue;
                    }
                }
                for (size_t base = 0; base < total; base += n) {                      // chunks(n) :424
                    const size_t chunk_len = std::min(n, total - base);
                    for (size_t j = 0; j < n; j++) {
                        size_t j_rev = reverse_bits_msb(j, (uint32_t)log_n);          // :426
                        if (j_rev < chunk_len) {
                            if (nth > 1) tmp_ct = prod[base + j_rev];
                            else coordinate_product(*c, cp, view(tmp_ct), view(res_prev[base + j_rev]));   // :429-434
                            packer_add(*c, s.packer, tmp_ct.data(), keys.atk_glwe);   // :435
                        } else {
                            packer_add(*c, s.packer, nullptr, keys.atk_glwe);         // :437-443
                        }
                    }
                }
                packer_flush(*c, s.packer, tmp_ct.data());                            // :448
                results.push_back(tmp_ct);                                            // :449
            } else if (i == 0) {
                coordinate_product(*c, cp, view(tmp_ct), view(s.data[0]));            // :451
                results.push_back(tmp_ct);
            } else {
                coordinate_product(*c, cp, view(tmp_ct), view(results[0]));           // :454
            }
        }
        glwe_trace_inplace(*c, view(tmp_ct), 0, log_n, keys.atk_glwe);                // :457
        std::memcpy(out, tmp_ct.data(), sizeof(int64_t) * glen());
    }

    // SubRam::read_prepare_write, ram.rs:461-542
    void subram_read_prepare_write(SubRam& s, const Address& address, const EvaluationKeysPrepared& keys, int64_t* out) {
        if (s.state) throw std::runtime_error("invalid call to Memory.read: internal state is true -> requires calling Memory.write");   // :472-475
        const int log_n = c->p.log_n; const size_t n = c->n();
        std::vector<std::vector<int64_t>> results;
        std::vector<int64_t> tmp_ct(glen(), 0);
        for (size_t i = 0; i < address.n2(); i++) {
            const Coordinate& coordinate = address.at(i);
            std::vector<std::vector<int64_t>>& res_prev = (i == 0) ? s.data : s.tree[i - 1];   // :490-494
            CoordinatePrepared cp;
            coordinate_prepare(*c, cp, coordinate);                                   // :496-499
#pragma omp parallel for num_threads(c->inner_threads()) schedule(dynamic)
            for (size_t r = 0; r < res_prev.size(); r++) coordinate_product_inplace(*c, cp, view(res_prev[r]));   // :502-504
            if (i < address.n2() - 1 && c->inner_threads() > 1 && res_prev.size() <= n) {
                std::vector<std::vector<int64_t>> leaves = res_prev;   // the rows stay as they are (ram.rs:514 packs copies)
                pack_level_sync(*c, leaves, c->p.size_ct(), tmp_ct.data(), keys.atk_glwe, c->inner_threads());
                results.push_back(tmp_ct);
                for (size_t k = 0; k < std::min(s.tree[i].size(), results.size()); k++) s.tree[i][k] = results[k];   // :525-527
            } else
            if (i < address.n2() - 1) {
                const size_t total = res_prev.size();
                for (size_t base = 0; base < total; base += n) {                      // :510
                    const size_t chunk_len = std::min(n, total - base);
                    for (size_t j = 0; j < n; j++) {
                        size_t j_rev = reverse_bits_msb(j, (uint32_t)log_n);          // :512
                        if (j_rev < chunk_len) packer_add(*c, s.packer, res_prev[base + j_rev].data(), keys.atk_glwe);   // :514
                        else packer_add(*c, s.packer, nullptr, keys.atk_glwe);        // :516
                    }
                }
                packer_flush(*c, s.packer, tmp_ct.data());                            // :521
                results.push_back(tmp_ct);                                            // :522
                for (size_t k = 0; k < std::min(s.tree[i].size(), results.size()); k++) s.tree[i][k] = results[k];   // :525-527
            }
        }
        std::vector<int64_t> res(glen());
        s.state = true;                                                               // :533
        if (address.n2() != 1) res = s.tree.back()[0];                                // :534-535
        else res = s.data[0];                                                         // :537
        glwe_trace_inplace(*c, view(res), 0, log_n, keys.atk_glwe);                   // :540
        std::memcpy(out, res.data(), sizeof(int64_t) * glen());
    }

    // SubRam::write_first_step, ram.rs:544-577
    void write_first_step(SubRam& s, const int64_t* w, size_t n2, const EvaluationKeysPrepared& keys) {
        if (!s.state) throw std::runtime_error("invalid call to Memory.write: internal state is false -> requires calling Memory.read_prepare_write");   // :555-558
        std::vector<int64_t>& to_write_on = (n2 != 1) ? s.tree.back()[0] : s.data[0];   // :565-569
        std::vector<int64_t> tmp_a = to_write_on;                                     // trace(out of place) :571-572
        glwe_trace_inplace(*c, view(tmp_a), 0, c->p.log_n, keys.atk_glwe);
        glwe_sub_inplace(view(to_write_on), view(tmp_a));                             // :574
        VecView wv = glwe_view(const_cast<int64_t*>(w), c->n(), c->p.size_ct());
        glwe_add_inplace(view(to_write_on), wv);                                      // :575
        glwe_normalize_inplace(*c, view(to_write_on));                                // :576
    }
    // SubRam::write_mid_step, ram.rs:579-632
    void write_mid_step(SubRam& s, size_t step, const CoordinatePrepared& inv_coordinate, const EvaluationKeysPrepared& keys) {
        const size_t n = c->n(); const int log_n = c->p.log_n;
        std::vector<std::vector<int64_t>>& tree_hi = (step == 0) ? s.data : s.tree[step - 1];   // :599-604
        std::vector<std::vector<int64_t>>& tree_lo = s.tree[step];
        for (size_t base = 0, j = 0; base < tree_hi.size(); base += n, j++) {         // :606
            std::vector<int64_t>& ct_lo = tree_lo[j];                                 // :608
            coordinate_product_inplace(*c, inv_coordinate, view(ct_lo));              // :610
            const size_t chunk_len = std::min(n, tree_hi.size() - base);
            if (c->inner_threads() > 1) {
                // all-core variant: iteration q sees ct_lo * X^-q (q rotations by X^-1 compose exactly), so the
                // rows are independent
                const std::vector<int64_t> ct_lo0 = ct_lo;
#pragma omp parallel for num_threads(c->inner_threads()) schedule(dynamic)
                for (size_t q = 0; q < chunk_len; q++) {
                    std::vector<int64_t>& ct_hi = tree_hi[base + q];
                    std::vector<int64_t> tmp_a = ct_hi, lo(ct_lo0.size());
                    glwe_trace_inplace(*c, view(tmp_a), 0, log_n, keys.atk_glwe);
                    glwe_sub_inplace(view(ct_hi), view(tmp_a));
                    glwe_rotate(*c, -(int64_t)q, view(lo), view(const_cast<std::vector<int64_t>&>(ct_lo0)));
                    glwe_trace_inplace(*c, view(lo), 0, log_n, keys.atk_glwe);
                    glwe_add_inplace(view(ct_hi), view(lo));
                    glwe_normalize_inplace(*c, view(ct_hi));
                }
                glwe_rotate(*c, -(int64_t)chunk_len, view(ct_lo), view(const_cast<std::vector<int64_t>&>(ct_lo0)));
                continue;
            }
            for (size_t q = 0; q < chunk_len; q++) {                                  // :612
                std::vector<int64_t>& ct_hi = tree_hi[base + q];
                std::vector<int64_t> tmp_a = ct_hi;
                glwe_trace_inplace(*c, view(tmp_a), 0, log_n, keys.atk_glwe);         // :616
                glwe_sub_inplace(view(ct_hi), view(tmp_a));                           // :617
                tmp_a = ct_lo;
                glwe_trace_inplace(*c, view(tmp_a), 0, log_n, keys.atk_glwe);         // :621
                glwe_add_inplace(view(ct_hi), view(tmp_a));                           // :625
                glwe_normalize_inplace(*c, view(ct_hi));                              // :626
                glwe_rotate_inplace(*c, -1, view(ct_lo));                             // :629
            }
        }
    }
    // SubRam::write_last_step, ram.rs:634-649
    void write_last_step(SubRam& s, const CoordinatePrepared& inv_coordinate) {
#pragma omp parallel for num_threads(c->inner_threads()) schedule(dynamic)
        for (size_t r = 0; r < s.data.size(); r++) coordinate_product_inplace(*c, inv_coordinate, view(s.data[r]));   // :644-646
        s.state = false;                                                              // :648
    }

    // the reference maps over the sub-RAMs one after the other (ram.rs:187-190); with Ctx::threads > 1 they run
    // concurrently (first exception wins, as a panic would)
    template <typename F>
    void par_subrams(F&& f) {
        const int nt = c->outer_threads();
        if (nt <= 1) { for (size_t i = 0; i < subrams.size(); i++) f(i); return; }
        std::exception_ptr err;
#pragma omp parallel for num_threads(nt) schedule(static, 1)
        for (size_t i = 0; i < subrams.size(); i++) {
            try { f(i); } catch (...) {
#pragma omp critical
                if (!err) err = std::current_exception();
            }
        }
        if (err) std::rethrow_exception(err);
    }

    // Ram::read, ram.rs:172-191.  out: word_size GLWEs.
    void read(const Address& address, const EvaluationKeysPrepared& keys, int64_t* out) {
        if (subrams.empty() || subrams[0].data.empty()) throw std::runtime_error("unitialized memory: self.data.len()=0");   // :182-185
        par_subrams([&](size_t i) { subram_read(subrams[i], address, keys, out + i * glen()); });
    }
    // Ram::read_prepare_write, ram.rs:196-222
    void read_prepare_write(const Address& address, const EvaluationKeysPrepared& keys, int64_t* out) {
        if (subrams.empty() || subrams[0].data.empty()) throw std::runtime_error("unitialized memory: self.data.len()=0");   // :206-209
        par_subrams([&](size_t i) { subram_read_prepare_write(subrams[i], address, keys, out + i * glen()); });
    }
    // Ram::write, ram.rs:226-294
    void write(const int64_t* w, size_t n_w, const Address& address, const EvaluationKeysPrepared& keys) {
        if (n_w != subrams.size()) throw std::runtime_error("w.len() != subrams.len()");   // :243
        par_subrams([&](size_t i) { write_first_step(subrams[i], w + i * glen(), address.n2(), keys); });   // :254-256
        for (size_t ii = address.n2() - 1; ii-- > 0;) {                               // (0..n2-1).rev() :258
            const Coordinate& coordinate = address.at(ii + 1);                        // :260
            CoordinatePrepared inv;
            coordinate_prepare_inv(*c, inv, coordinate, keys.atk_ggsw_inv, keys.tsk_ggsw_inv);   // :265-271
            par_subrams([&](size_t i) { write_mid_step(subrams[i], ii, inv, keys); });   // :273-275
        }
        CoordinatePrepared inv0;
        coordinate_prepare_inv(*c, inv0, address.at(0), keys.atk_ggsw_inv, keys.tsk_ggsw_inv);   // :278-289
        par_subrams([&](size_t i) { write_last_step(subrams[i], inv0); });            // :291-293
    }
};

}  // namespace fo
