// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see oracle/README.md).
//
// Setup side of the reference flow (/root/reference/examples/fhe-ram.rs:34-95): secret and
// evaluation-key generation, RAM / address / word encryption, decryption + noise metric.
// Sampling uses this oracle's own seeded PRNG: Poulpy's `Source` (ChaCha) and its samplers
// are un-vendored, and bit-parity of keys with upstream is neither attainable nor needed
//  — the hot path consumes whatever keys/ciphertexts it is handed.
#pragma once
#include "fheram_oracle.hpp"

namespace fo {

// xoshiro256** seeded through splitmix64
struct Source {
    uint64_t s[4];
    explicit Source(uint64_t seed) {
        uint64_t z = seed;
        for (int i = 0; i < 4; i++) {
            z += 0x9E3779B97F4A7C15ULL;
            uint64_t x = z;
            x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
            x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
            s[i] = x ^ (x >> 31);
        }
    }
    static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        const uint64_t r = rotl(s[1] * 5, 7) * 9;
        const uint64_t t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
        s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    uint32_t next_u32() { return (uint32_t)(next() >> 32); }
    double next_f64() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }   // [0,1)
    int64_t uniform_limb(int base2k) { return get_digit(base2k, (int64_t)next()); }
    // round(N(0, sigma^2)) truncated at +-bound (Poulpy add_normal: resample while |x| > bound)
    int64_t gaussian(double sigma, double bound) {
        for (;;) {
            double u1 = next_f64(), u2 = next_f64();
            if (u1 <= 0.0) continue;
            double x = std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586476925 * u2) * sigma;
            if (std::fabs(x) > bound) continue;
            return (int64_t)std::llround(x);
        }
    }
    void fill_bytes(uint8_t* p, size_t n) {
        for (size_t i = 0; i < n; i += 8) {
            uint64_t r = next();
            for (size_t j = 0; j < 8 && i + j < n; j++) p[i + j] = (uint8_t)(r >> (8 * j));
        }
    }
};

// GLWESecret::fill_ternary_prob(0.5) (examples/fhe-ram.rs:49-50): each coefficient is
// nonzero with probability 0.5, sign uniform.
static inline void secret_fill_ternary_prob(const Ctx& c, int64_t* sk, double prob, Source& xs) {
    for (int i = 0; i < c.n(); i++) {
        double u = xs.next_f64();
        uint64_t b = xs.next();
        sk[i] = (u < prob) ? ((b & 1) ? 1 : -1) : 0;
    }
}

// exact negacyclic product of a limb polynomial with a small polynomial
static inline void poly_mul_small(Ctx& c, const int64_t* a, const PolyHat& s_hat, int64_t* out) {
    std::vector<uint64_t> ah, acc(c.n(), 0);
    to_hat(c.ntt, a, ah);
    mac_hat(c.ntt, acc, ah, s_hat);
    from_hat(c.ntt, acc, out);
}

// GLWE::encrypt_sk [UPSTREAM-RECALL, SURVEY.md A.10]: mask uniform per limb,
// body = -a*s + e (+ pt if pt_col == 0); if pt_col == 1 the plaintext is added to the mask
// column after the product with s was taken (GGSW rows, A.2).
// pt: [pt_size][n] (one column) or null.
static inline void glwe_encrypt_sk(Ctx& c, const VecView& ct, const int64_t* pt, int pt_size, int pt_col,
                                   const PolyHat& sk_hat, int k, Source& xa, Source& xe) {
    const int n = c.n(), size = ct.size, b2k = c.p.base2k;
    for (int j = 0; j < size; j++) { int64_t* m = ct.at(1, j); for (int i = 0; i < n; i++) m[i] = xa.uniform_limb(b2k); }
    Big big(n, 1, size);
    std::vector<int64_t> prod(n);
    for (int j = 0; j < size; j++) {
        poly_mul_small(c, ct.at(1, j), sk_hat, prod.data());
        int64_t* b = big.at(0, j);
        for (int i = 0; i < n; i++) b[i] = -prod[i];
    }
    if (pt && pt_col == 0)
        for (int j = 0; j < std::min(pt_size, size); j++) { int64_t* b = big.at(0, j); for (int i = 0; i < n; i++) b[i] += pt[(size_t)j * n + i]; }
    // noise at precision k: limb ceil(k/base2k)-1, scaled by 2^((limb+1)*base2k - k)
    const int limb = (k + b2k - 1) / b2k - 1;
    const double scale = std::ldexp(1.0, (limb + 1) * b2k - k);
    { int64_t* b = big.at(0, limb); for (int i = 0; i < n; i++) b[i] += xe.gaussian(c.p.sigma * scale, 6.0 * c.p.sigma * scale); }
    big_normalize_col(c, ct, 0, big, 0);
    if (pt && pt_col == 1) {
        for (int j = 0; j < std::min(pt_size, size); j++) { int64_t* m = ct.at(1, j); for (int i = 0; i < n; i++) m[i] += pt[(size_t)j * n + i]; }
        normalize_inplace(b2k, ct, 1);
    }
}

// GGSW::encrypt_sk [UPSTREAM-RECALL, A.2]: row r / col_in c encrypts m * (c==0 ? 1 : s) * 2^-((r+1)*base2k).
// out: [dnum][2][size][2][n]; scalar: [n].     (coordinate.rs:161-168)
static inline void ggsw_encrypt_sk(Ctx& c, int64_t* out, int dnum, int size, int k, const int64_t* scalar,
                                   const PolyHat& sk_hat, Source& xa, Source& xe) {
    const int n = c.n();
    const size_t glen = (size_t)size * 2 * n;
    std::vector<int64_t> pt((size_t)size * n);
    for (int r = 0; r < dnum; r++) {
        std::fill(pt.begin(), pt.end(), 0);
        if (r < size) std::memcpy(pt.data() + (size_t)r * n, scalar, sizeof(int64_t) * n);
        for (int ci = 0; ci < 2; ci++)
            glwe_encrypt_sk(c, glwe_view(out + ((size_t)r * 2 + ci) * glen, n, size), pt.data(), size, ci, sk_hat, k, xa, xe);
    }
}
// GGLWE switching key sk_in -> sk_out [UPSTREAM-RECALL, A.2]: row r encrypts sk_in * 2^-((r+1)*base2k)
// under sk_out.   out: [dnum][1][size][2][n]
static inline void gglwe_encrypt_sk(Ctx& c, int64_t* out, int dnum, int size, int k, const int64_t* pt_scalar,
                                    const PolyHat& sk_out_hat, Source& xa, Source& xe) {
    const int n = c.n();
    const size_t glen = (size_t)size * 2 * n;
    std::vector<int64_t> pt((size_t)size * n);
    for (int r = 0; r < dnum; r++) {
        std::fill(pt.begin(), pt.end(), 0);
        if (r < size) std::memcpy(pt.data() + (size_t)r * n, pt_scalar, sizeof(int64_t) * n);
        glwe_encrypt_sk(c, glwe_view(out + (size_t)r * glen, n, size), pt.data(), size, 0, sk_out_hat, k, xa, xe);
    }
}
// GLWEAutomorphismKey::encrypt_sk(p) [UPSTREAM-RECALL]: key from s to phi_{p^-1}(s), so that
// phi_p(KS(a)) decrypts under s to phi_p(m).   (keys.rs:158-165,171-173)
static inline void automorphism_key_encrypt_sk(Ctx& c, int64_t* out, int dnum, int size, int k, int64_t p,
                                               const int64_t* sk, Source& xa, Source& xe) {
    std::vector<int64_t> sk_out(c.n());
    poly_automorphism(c.n(), galois_inverse(c.p.log_n, p), sk_out.data(), sk);
    PolyHat h; to_hat_prepared(c.ntt, sk_out.data(), h);
    gglwe_encrypt_sk(c, out, dnum, size, k, sk, h, xa, xe);
}
// GGLWEToGGSWKey::encrypt_sk (rank 1) [UPSTREAM-RECALL]: GGLWE of s*s under s.   (keys.rs:167-169)
static inline void tensor_key_encrypt_sk(Ctx& c, int64_t* out, int dnum, int size, int k, const int64_t* sk,
                                         Source& xa, Source& xe) {
    PolyHat h; to_hat_prepared(c.ntt, sk, h);
    std::vector<int64_t> ss(c.n());
    poly_mul_small(c, sk, h, ss.data());
    gglwe_encrypt_sk(c, out, dnum, size, k, ss.data(), h, xa, xe);
}

// EvaluationKeys::encrypt_sk, keys.rs:135-180.  Outputs std-form keys in Poulpy host layout.
struct EvaluationKeysStd {
    std::vector<int64_t> gal_els;
    std::vector<std::vector<int64_t>> atk_glwe;    // one per galois element, [dnum_ct][1][size4][2][n]
    std::vector<int64_t> atk_ggsw_inv;             // [dnum_ggsw][1][size5][2][n]
    std::vector<int64_t> tsk;                      // [dnum_ggsw][1][size5][2][n]
};
static inline void evaluation_keys_encrypt_sk(Ctx& c, EvaluationKeysStd& k, const int64_t* sk, Source& xa, Source& xe) {
    const Params& p = c.p;
    k.gal_els.clear(); k.atk_glwe.clear();
    for (int i = 0; i < p.log_n; i++) k.gal_els.push_back(galois_element(p.log_n, i));     // keys.rs:158
    for (auto g : k.gal_els) {                                                             // keys.rs:159-165
        std::vector<int64_t> key(p.atk_trace_len());
        automorphism_key_encrypt_sk(c, key.data(), p.dnum_ct(), p.size_evk_trace(), p.k_evk_trace, g, sk, xa, xe);
        k.atk_glwe.push_back(std::move(key));
    }
    k.tsk.assign(p.evk_inv_len(), 0);                                                      // keys.rs:167-169
    tensor_key_encrypt_sk(c, k.tsk.data(), p.dnum_ggsw(), p.size_evk_inv(), p.k_evk_ggsw_inv, sk, xa, xe);
    k.atk_ggsw_inv.assign(p.evk_inv_len(), 0);                                             // keys.rs:171-173
    automorphism_key_encrypt_sk(c, k.atk_ggsw_inv.data(), p.dnum_ggsw(), p.size_evk_inv(), p.k_evk_ggsw_inv, -1, sk, xa, xe);
}
// EvaluationKeysPrepared::alloc + prepare, keys.rs:34-71
static inline void evaluation_keys_prepare(Ctx& c, EvaluationKeysPrepared& out, const int64_t* gal_els, int n_gal,
                                           const int64_t* const* atk_glwe, const int64_t* atk_ggsw_inv, const int64_t* tsk) {
    const Params& p = c.p;
    out.atk_glwe.clear();
    for (int i = 0; i < n_gal; i++) {
        KeyPrepared kp; kp.p = gal_els[i];
        kp.m = prepare_mat(c, atk_glwe[i], p.dnum_ct(), 1, p.size_evk_trace());
        out.atk_glwe[gal_els[i]] = std::move(kp);
    }
    out.atk_ggsw_inv.p = -1;
    out.atk_ggsw_inv.m = prepare_mat(c, atk_ggsw_inv, p.dnum_ggsw(), 1, p.size_evk_inv());
    out.tsk_ggsw_inv = prepare_mat(c, tsk, p.dnum_ggsw(), 1, p.size_evk_inv());
}

// encode_vec_i64(data, k) on a 1-column plaintext of ceil(k/base2k) limbs [UPSTREAM-RECALL, A.10]:
// value * 2^-k on the torus; bits above k wrap mod 1.
static inline void encode_vec_i64(const Ctx& c, int64_t* pt, int k, const int64_t* data, size_t len) {
    const int n = c.n(), b2k = c.p.base2k;
    const int size = (k + b2k - 1) / b2k;
    std::fill(pt, pt + (size_t)size * n, 0);
    const int sh = size * b2k - k;
    for (size_t i = 0; i < len; i++) pt[(size_t)(size - 1) * n + i] = (int64_t)((uint64_t)data[i] << sh);
    VecView v{pt, n, 1, size};
    normalize_inplace(b2k, v, 0);
}

// Ram::encrypt_sk + SubRam::encrypt_sk, ram.rs:129-167,334-380.
// rows_out: [word_size][rows][size_ct][2][n]
static inline void ram_encrypt_sk(Ctx& c, const uint8_t* data, size_t data_len, const int64_t* sk,
                                  Source& xa, Source& xe, int64_t* rows_out) {
    const Params& p = c.p;
    const size_t n = c.n(), ws = p.word_size, max_addr = p.max_addr;
    if (data_len % ws != 0) throw std::runtime_error("invalid data: data.len()%ram_chunks != 0");          // :144-148
    if (data_len / ws != max_addr) throw std::runtime_error("invalid data: data.len()/ram_chunks != max_addr");   // :150-155
    PolyHat sk_hat; to_hat_prepared(c.ntt, sk, sk_hat);
    const size_t rows = p.rows(), glen = p.glwe_len(p.size_ct());
    std::vector<uint8_t> split(max_addr);
    std::vector<int64_t> data_i64(n), pt((size_t)p.size_pt() * n);
    for (size_t i = 0; i < ws; i++) {
        for (size_t j = 0; j < max_addr; j++) split[j] = data[j * ws + i];                                  // :161-164
        for (size_t r = 0; r < rows; r++) {                                                                 // chunks(n) :358-379
            const size_t len = std::min(n, max_addr - r * n);
            for (size_t q = 0; q < n; q++) data_i64[q] = q < len ? (int64_t)(int8_t)split[r * n + q] : 0;   // :363-367
            encode_vec_i64(c, pt.data(), p.k_glwe_pt, data_i64.data(), n);                                  // :368
            glwe_encrypt_sk(c, glwe_view(rows_out + (i * rows + r) * glen, (int)n, p.size_ct()), pt.data(), p.size_pt(), 0,
                            sk_hat, p.k_glwe_ct, xa, xe);                                                   // :369-376
        }
    }
}

// Coordinate::encrypt_sk, coordinate.rs:121-180
static inline void coordinate_encrypt_sk(Ctx& c, Coordinate& co, int64_t value, const PolyHat& sk_hat, Source& xa, Source& xe) {
    const Params& p = c.p;
    const int64_t n = c.n();
    if (!(std::llabs(value) < n)) throw std::runtime_error("coordinate: |value| >= n");     // :136
    std::vector<int64_t> scalar(n, 0);
    const int64_t sign = (value > 0) - (value < 0);
    const size_t gap = 1;                                                                   // :146
    size_t remain = (size_t)std::llabs(value);
    unsigned tot_base = 0;
    co.value.assign(co.base1d.d.size(), std::vector<int64_t>(p.ggsw_len()));
    for (size_t d = 0; d < co.base1d.d.size(); d++) {
        const unsigned base = co.base1d.d[d];
        const size_t mask = ((size_t)1 << base) - 1;
        const size_t chunk = ((remain & mask) << tot_base) * gap;                           // :154
        if (sign < 0 && chunk != 0) scalar[n - chunk] = -1;                                 // :156-157
        else scalar[chunk] = 1;
        ggsw_encrypt_sk(c, co.value[d].data(), p.dnum_ct(), p.size_addr(), p.k_ggsw_addr, scalar.data(), sk_hat, xa, xe);   // :161-168
        if (sign < 0 && chunk != 0) scalar[n - chunk] = 0;
        else scalar[chunk] = 0;
        remain >>= base;
        tot_base += base;
    }
}
// Address::alloc_from_params + encrypt_sk, address.rs:58-74,86-109
static inline void address_encrypt_sk(Ctx& c, Address& a, uint32_t value, const int64_t* sk, Source& xa, Source& xe) {
    a.base2d = c.p.base2d();
    a.coordinates.clear();
    for (auto& b : a.base2d.v) { Coordinate co; co.base1d = b; a.coordinates.push_back(co); }
    PolyHat sk_hat; to_hat_prepared(c.ntt, sk, sk_hat);
    size_t remain = value;
    for (auto& co : a.coordinates) {
        const size_t max = co.base1d.max();
        const size_t k = remain & (max - 1);                                                // :103
        coordinate_encrypt_sk(c, co, -(int64_t)k, sk_hat, xa, xe);                          // :104
        remain /= max;                                                                      // :105
    }
}

// examples/fhe-ram.rs:25-32
static inline int64_t cast_u8_to_signed(uint8_t value, int bit_length) {
    const int shift = 8 - bit_length;
    return (int64_t)((int8_t)(uint8_t)(value << shift)) >> shift;
}
// encrypt_glwe, examples/fhe-ram.rs:179-210: value placed on coefficient 0 at precision k_pt.
static inline void encrypt_glwe_coeff0(Ctx& c, int64_t* ct, uint8_t value, const int64_t* sk, Source& xa, Source& xe) {
    const Params& p = c.p;
    std::vector<int64_t> pt((size_t)p.size_pt() * c.n());
    int64_t v = (int64_t)value;
    encode_vec_i64(c, pt.data(), p.k_glwe_pt, &v, 1);        // encode_coeff_i64(value, k_pt, 0)  :196
    PolyHat sk_hat; to_hat_prepared(c.ntt, sk, sk_hat);
    glwe_encrypt_sk(c, glwe_view(ct, c.n(), p.size_ct()), pt.data(), p.size_pt(), 0, sk_hat, p.k_glwe_ct, xa, xe);
}
// decrypt_glwe + noise, examples/fhe-ram.rs:212-237
static inline void decrypt_glwe(Ctx& c, const int64_t* ct, int64_t want, const int64_t* sk, int64_t* value, double* noise,
                                int coeff = 0) {
    const Params& p = c.p;
    const int n = c.n(), size = p.size_ct(), b2k = p.base2k, k = p.k_glwe_ct;
    VecView v = glwe_view(const_cast<int64_t*>(ct), n, size);
    PolyHat sk_hat; to_hat_prepared(c.ntt, sk, sk_hat);
    Big big(n, 1, size);
    for (int j = 0; j < size; j++) {
        poly_mul_small(c, v.at(1, j), sk_hat, big.at(0, j));
        int64_t* b = big.at(0, j); const int64_t* body = v.at(0, j);
        for (int i = 0; i < n; i++) b[i] += body[i];
    }
    std::vector<int64_t> pt((size_t)size * n);
    VecView pv{pt.data(), n, 1, size};
    big_normalize_col(c, pv, 0, big, 0);
    // decode_coeff_i64(k, coeff) [UPSTREAM-RECALL]
    const int dsize = (k + b2k - 1) / b2k;
    const int rem = b2k - (k % b2k);
    int64_t res = 0;
    for (int j = 0; j < dsize; j++) {
        int64_t x = pv.at(0, j)[coeff];
        if (j == dsize - 1 && rem != b2k) res = (int64_t)((uint64_t)res << (b2k - rem)) + (x >> rem);
        else res = (int64_t)((uint64_t)res << b2k) + x;
    }
    const int log_scale = k - p.k_glwe_pt;                                                  // :228
    const int64_t diff = res - (int64_t)((uint64_t)want << log_scale);                      // :230
    *noise = std::log2((double)std::llabs(diff)) - (double)k;                               // :231
    *value = (int64_t)std::llround((double)res / std::ldexp(1.0, log_scale));               // :232-233
}

}  // namespace fo
