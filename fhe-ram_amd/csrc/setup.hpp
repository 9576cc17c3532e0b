// Included by fheram.hip (same translation unit): setup side on the device, SURVEY.md §8(f) N2.
#pragma once
#include "launch.hpp"

// ---- setup side on the device ------------------------------------------
// Sampling is the caller's: masks and noise arrive as integers; what runs here is the arithmetic.
namespace {
// Secret-derived buffers are zeroed before they go back to an allocator: the device ones with a memset that
// has completed before hipFree, the host ones through a volatile pointer (a plain std::fill right before
// the delete / return is a dead store the compiler may drop).
void wipe_free(hipStream_t st, void* d, size_t bytes) {
    if (!d) return;
    if (hipMemsetAsync(d, 0, bytes, st) == hipSuccess) hipStreamSynchronize(st);
    hipFree(d);
}
void wipe_host(void* p, size_t bytes) {
    volatile unsigned char* v = static_cast<volatile unsigned char*>(p);
    for (size_t i = 0; i < bytes; i++) v[i] = 0;
}
}  // namespace

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
    wipe_free(c->stream, d_in, N * sizeof(int32_t));
    if (e != hipSuccess) { fheram_secret_destroy(s); return fail(c, FHERAM_ERR_DEVICE, std::string("secret prepare: ") + hipGetErrorString(e)); }
    *out = s;
    return FHERAM_OK;
}
void fheram_secret_destroy(fheram_secret* s) {
    if (!s) return;
    hipSetDevice(s->device);
    wipe_free(nullptr, s->d_hat, N * sizeof(double));
    wipe_host(s->sk.data(), s->sk.size() * sizeof(int32_t));
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
        std::vector<int32_t> h_i32(total);
        hipError_t e = hipMemcpyAsync(h_i32.data(), d_ct, total * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = fail(c, FHERAM_ERR_DEVICE, std::string("decrypt: ") + hipGetErrorString(e));
        else
            for (int g = 0; g < n_glwe; g++)
                for (int j = 0; j < size; j++)
                    widen(h_i32.data() + (size_t)g * glen + (size_t)(j * 2) * N, pt + ((size_t)g * size + j) * N, N);
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
    int rc = check_setup_args(c, sk, c->s_evk, (int)c->p.k_evk_trace);
    if (rc == FHERAM_OK) rc = check_setup_args(c, sk, fheram_ctx::S_INV, (int)c->p.k_evk_ggsw_inv);
    if (rc != FHERAM_OK) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream2));   // a precompute started by read_prepare_write may still be reading the old keys
    c->cur = c->stream;
    int32_t *d_stage = nullptr, *d_small = nullptr;
    double* d_hat = nullptr;
    const size_t stage_n = std::max(c->atk, fheram_ctx::EVK5);
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
        rc = automorphism_key(c->gal[i], fheram_ctx::DNUM_CT, c->s_evk, (int)c->p.k_evk_trace,
                              c->d_atk + (size_t)i * c->atk, std_out ? std_out + (size_t)i * c->atk : nullptr);
    if (rc == FHERAM_OK) {   // tensor key (rank 1): GGLWE of s*s under s (keys.rs:167-169); s*s = phase of (0, s) under s
        std::copy(sk->sk.begin(), sk->sk.end(), ss.begin() + N);
        hipMemcpyAsync(d_small, ss.data(), 2 * N * sizeof(int32_t), hipMemcpyHostToDevice, c->stream);
        launch_enc_dyn(c, 1, 1, d_small, sk->d_hat, nullptr, 1);
        hipMemcpyAsync(ss.data(), d_small, N * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream);
        if (hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(c, FHERAM_ERR_DEVICE, "tensor key: s*s failed");
    }
    if (rc == FHERAM_OK)
        rc = gglwe(ss.data(), sk->d_hat, fheram_ctx::DNUM_GGSW, fheram_ctx::S_INV, (int)c->p.k_evk_ggsw_inv, c->d_tsk,
                   std_out ? std_out + (size_t)LOGN * c->atk : nullptr, 0);
    if (rc == FHERAM_OK)
        rc = automorphism_key(-1, fheram_ctx::DNUM_GGSW, fheram_ctx::S_INV, (int)c->p.k_evk_ggsw_inv, c->d_atk_inv,
                              std_out ? std_out + (size_t)LOGN * c->atk + fheram_ctx::EVK5 : nullptr);
    hipFree(d_stage);
    wipe_free(c->stream, d_small, 2 * N * sizeof(int32_t));   // phi(s), s*s
    wipe_free(c->stream, d_hat, N * sizeof(double));
    wipe_host(sk_out.data(), sk_out.size() * sizeof(sk_out[0])); wipe_host(ss.data(), ss.size() * sizeof(ss[0]));
    if (rc != FHERAM_OK) return rc;
    HIPCHK(c, hipGetLastError());
    c->keys_loaded = true;
    c->inv_id[0] = c->inv_id[1] = 0;
    c->inv_pending[0] = c->inv_pending[1] = false;
    c->memo_top = false; c->memo_alone = 0;
    return FHERAM_OK;
}

// ---- Address::set_from_fheuint (conversion.rs:18-82) ---------------------------------------
size_t fheram_fheuint_ggsw_len(const fheram_ctx*) { return fheram_ctx::GGSW5; }
namespace {
int fheuint_alloc(fheram_ctx* c, int n_bits, fheram_fheuint** out) {
    fheram_fheuint* f = new fheram_fheuint{c, c->device, n_bits, nullptr, nullptr};
    hipError_t e = hipMalloc(&f->d_std, (size_t)n_bits * fheram_ctx::GGSW5 * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc(&f->d_prep, (size_t)n_bits * fheram_ctx::GGSW5 * sizeof(double));
    if (e != hipSuccess) { fheram_fheuint_destroy(f); return fail(c, FHERAM_ERR_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(e)); }
    *out = f;
    return FHERAM_OK;
}
int fheuint_prepare(fheram_ctx* c, fheram_fheuint* f) {
    c->cur = c->stream;
    launch_prepare(c, f->d_std, f->d_prep, f->n_bits * (int)(fheram_ctx::GGSW5 / N));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    return FHERAM_OK;
}
}  // namespace
int fheram_fheuint_create(fheram_ctx* c, const int64_t* bits, int n_bits, fheram_fheuint** out) {
    if (!c || !out) return FHERAM_ERR_INVALID_ARG;
    *out = nullptr;
    if (!bits || n_bits <= 0 || n_bits > 64) return fail(c, FHERAM_ERR_INVALID_ARG, "bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    fheram_fheuint* f = nullptr;
    int rc = fheuint_alloc(c, n_bits, &f);
    if (rc == FHERAM_OK) rc = upload_i64(c, f->d_std, bits, (size_t)n_bits * fheram_ctx::GGSW5);
    if (rc == FHERAM_OK) rc = fheuint_prepare(c, f);
    if (rc != FHERAM_OK) { fheram_fheuint_destroy(f); return rc; }
    *out = f;
    return FHERAM_OK;
}
int fheram_fheuint_encrypt_sk(fheram_ctx* c, const fheram_secret* sk, uint32_t value, int n_bits, const int64_t* mask,
                              const int64_t* noise, fheram_fheuint** out) {
    if (!c || !out) return FHERAM_ERR_INVALID_ARG;
    *out = nullptr;
    if (!mask || !noise || n_bits <= 0 || n_bits > 32) return fail(c, FHERAM_ERR_INVALID_ARG, "bad argument");
    const int S = fheram_ctx::S_INV, k = (int)c->p.k_evk_ggsw_inv, D = fheram_ctx::DNUM_GGSW;
    int rc = check_setup_args(c, sk, S, k);
    if (rc != FHERAM_OK) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t glen = (size_t)S * 2 * N;
    const int n = n_bits * D * 2;
    std::vector<int32_t> pre((size_t)n * glen, 0), pt1((size_t)n * S * N, 0);
    int gi = 0;
    for (int i = 0; i < n_bits; i++) {
        const int32_t bit = (int32_t)((value >> i) & 1);
        for (int r = 0; r < D; r++)
            for (int ci = 0; ci < 2; ci++, gi++) {
                int32_t* pg = pre.data() + (size_t)gi * glen;
                if (ci == 0) pg[(size_t)(r * 2) * N] = bit;            // row r: b * 2^-((r+1)*base2k) in the body
                else pt1[((size_t)gi * S + r) * N] = bit;              //        b * s * ..., added to the mask column
                if (!stage_random(pg, S, k, mask + (size_t)gi * S * N, noise + (size_t)gi * N))
                    return fail(c, FHERAM_ERR_RANGE, "mask limb outside [-2^16, 2^16) or |noise| >= 2^30");
            }
    }
    fheram_fheuint* f = nullptr;
    rc = fheuint_alloc(c, n_bits, &f);
    if (rc == FHERAM_OK) rc = encrypt_staged(c, sk->d_hat, f->d_std, pre, &pt1, n, S);
    if (rc == FHERAM_OK) rc = fheuint_prepare(c, f);
    wipe_host(pre.data(), pre.size() * sizeof(int32_t)); wipe_host(pt1.data(), pt1.size() * sizeof(int32_t));
    if (rc != FHERAM_OK) { fheram_fheuint_destroy(f); return rc; }
    *out = f;
    return FHERAM_OK;
}
int fheram_fheuint_download(fheram_ctx* c, const fheram_fheuint* f, int64_t* out) {
    if (!c || !out) return FHERAM_ERR_INVALID_ARG;
    if (!f || f->ctx != c) return fail(c, FHERAM_ERR_INVALID_ARG, "integer belongs to another context");
    HIPCHK(c, hipSetDevice(c->device));
    return download_i64(c, out, f->d_std, (size_t)f->n_bits * fheram_ctx::GGSW5);
}
void fheram_fheuint_destroy(fheram_fheuint* f) {
    if (!f) return;
    hipSetDevice(f->device);
    if (f->d_std) hipFree(f->d_std);
    if (f->d_prep) hipFree(f->d_prep);
    delete f;
}
// conversion.rs:41-65.  For every row of every digit: acc = trivial encryption of the gadget element, then one CMux per
// bit of the digit: acc <- normalize(acc + normalize(X^(+-2^(i+lsh)) acc - acc) (x) GGSW(b_i)).  The external products run
// on the fine-split kernels (one workgroup per input and output limb: 80 per row), all 6 rows of a digit per launch.
int fheram_address_set_from_fheuint(fheram_ctx* c, const fheram_fheuint* f, int sign, fheram_addr** out) {
    if (!c || !out) return FHERAM_ERR_INVALID_ARG;
    *out = nullptr;
    if (!f || f->ctx != c) return fail(c, FHERAM_ERR_INVALID_ARG, "integer belongs to another context");
    unsigned bits = 0;
    for (auto& b1 : c->base2d) for (int b : b1) bits += (unsigned)b;
    if ((int)bits > f->n_bits) return fail(c, FHERAM_ERR_INVALID_ARG, "the address plan is wider than the encrypted integer");
    HIPCHK(c, hipSetDevice(c->device));
    constexpr int S = fheram_ctx::S_ADDR, SG = fheram_ctx::S_INV, D = fheram_ctx::DNUM_CT;
    const long g4 = (long)fheram_ctx::GLWE4;
    const int rows = D * 2;                                    // GLWEs per digit
    fheram_addr* a = new fheram_addr{c, nullptr, c->n_digits, c->device};
    DevBuf tbuf, ebuf;
    double* big = nullptr;
    hipError_t e = hipMalloc(&a->d_ggsw, (size_t)c->n_digits * fheram_ctx::GGSW * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc(&tbuf.p, (size_t)rows * g4 * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc(&ebuf.p, (size_t)rows * g4 * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc(&big, (size_t)rows * 2 * SG * 2 * S * N * sizeof(double));
    if (e != hipSuccess) { if (big) hipFree(big); fheram_address_destroy(a); return fail(c, FHERAM_ERR_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(e)); }
    {   // test_vector = X^0 (conversion.rs:42-43) on gadget row r: body limb r for col_in 0, mask limb r for col_in 1
        std::vector<int32_t> h((size_t)c->n_digits * fheram_ctx::GGSW, 0);
        for (int d = 0; d < c->n_digits; d++)
            for (int r = 0; r < D; r++)
                for (int ci = 0; ci < 2; ci++) h[((size_t)(d * D + r) * 2 + ci) * g4 + (size_t)(r * 2 + ci) * N] = 1;
        e = hipMemcpy(a->d_ggsw, h.data(), h.size() * sizeof(int32_t), hipMemcpyHostToDevice);
    }
    c->cur = c->stream;
    int digit = 0;
    unsigned bit_rsh = 0;
    for (auto& base1d : c->base2d) {                                                      // :45
        unsigned bit_lsh = 0;                                                             // :46
        for (int bit_mask : base1d) {                                                     // :49
            GlweRef acc = ref(a->d_ggsw + (size_t)digit * fheram_ctx::GGSW, 0, g4);
            GlweRef t = ref(tbuf.p, 0, g4), ev = ref(ebuf.p, 0, g4);
            for (int i = 0; i < bit_mask; i++) {
                const int step = 1 << (i + bit_lsh);
                const double* gb = f->d_prep + (size_t)(bit_rsh + i) * fheram_ctx::GGSW5;
                ProfScope ps(c, "set_from_fheuint", rows);
                hipLaunchKernelGGL((k_cmux_pre<S>), dim3(rows, 1, EW_SLICES), dim3(256), 0, c->cur, acc, t, sign ? step : -step);
                hipLaunchKernelGGL((k_ext_product_fine<S, SG>), dim3(rows, 1, 2 * SG * 2 * S), dim3(T), LDS_BYTES, c->cur, t, gb, c->d_tw, big);
                hipLaunchKernelGGL((k_ext_product_fine_norm<S, SG>), dim3(rows, 1, 2 * (N / 256)), dim3(256), 0, c->cur, ev, big);
                hipLaunchKernelGGL((k_add_norm<S>), dim3(rows, 1, EW_SLICES), dim3(256), 0, c->cur, acc, ev, acc);
            }
            bit_lsh += (unsigned)bit_mask;                                                // :61
            bit_rsh += (unsigned)bit_mask;                                                // :62
            digit++;
        }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipGetLastError();
    hipFree(big);
    if (e != hipSuccess) { fheram_address_destroy(a); return fail(c, FHERAM_ERR_DEVICE, std::string("set_from_fheuint: ") + hipGetErrorString(e)); }
    *out = a;
    return FHERAM_OK;
}

}  // extern "C"
