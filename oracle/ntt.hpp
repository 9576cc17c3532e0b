// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see oracle/README.md).
//
// Exact negacyclic convolution engine for the oracle.  The reference multiplies polynomials
// with Poulpy's FFT64 backend (f64 complex FFT, /root/reference/examples/fhe-ram.rs:3-7;
// SURVEY.md §0.4); every product on the RAM path is an exact integer negacyclic convolution
// bounded by 2^47, so a number-theoretic transform modulo a 62-bit prime
// followed by a centred lift returns exactly the integers a correct FFT64 rounds to.
//
// The modulus here (Q62) is deliberately different from the one the HIP kernels use
// (2^48+57345): agreement of the lifted i64 results is then a real cross-check.
#pragma once
#include <cstdint>
#include <vector>
#include <memory>
#include <map>
#include <cassert>
#include <stdexcept>

namespace fo {

typedef unsigned __int128 u128;

struct Ntt {
    static constexpr uint64_t Q = 4611686018427322369ULL;      // prime, Q = 1 mod 8192, 62 bits
    static constexpr uint64_t PSI_8192 = 3391169269051246823ULL;  // primitive 8192-th root of 1

    int log_n, n;
    std::vector<uint64_t> w, ws;    // psi^brv(i), Shoup companion
    std::vector<uint64_t> iw, iws;  // psi^-brv(i)
    uint64_t n_inv, n_inv_s;

    static uint64_t mulmod(uint64_t a, uint64_t b) { return (uint64_t)((u128)a * b % Q); }
    static uint64_t powmod(uint64_t a, uint64_t e) {
        uint64_t r = 1;
        while (e) { if (e & 1) r = mulmod(r, a); a = mulmod(a, a); e >>= 1; }
        return r;
    }
    // floor(w * 2^64 / Q): estimate in 80-bit floating point, then correct exactly (a 128/64-bit
    // division per coefficient would dominate the cost of preparing an operand).
    static uint64_t shoup(uint64_t w_) {
        static const long double inv = 18446744073709551616.0L / (long double)Q;
        uint64_t q = (uint64_t)((long double)w_ * inv);
        u128 num = (u128)w_ << 64;
        u128 prod = (u128)q * Q;
        while (prod > num) { q--; prod -= Q; }
        while (num - prod >= Q) { q++; prod += Q; }
        return q;
    }
    // x*w mod Q with precomputed ws = floor(w*2^64/Q); x < 2^64 arbitrary, result in [0,Q)
    static inline uint64_t mul_shoup(uint64_t x, uint64_t w_, uint64_t ws_) {
        uint64_t q = (uint64_t)(((u128)x * ws_) >> 64);
        uint64_t r = x * w_ - q * Q;
        return r - (Q & (0 - (uint64_t)(r >= Q)));
    }
    // branch-free: the comparisons are data dependent and would mispredict half of the time
    static inline uint64_t addmod(uint64_t a, uint64_t b) { uint64_t s = a + b; return s - (Q & (0 - (uint64_t)(s >= Q))); }
    static inline uint64_t submod(uint64_t a, uint64_t b) { uint64_t d = a - b; return d + (Q & (0 - (uint64_t)(a < b))); }
    static inline uint64_t to_mod(int64_t x) { return x < 0 ? (uint64_t)(x + (int64_t)Q) : (uint64_t)x; }
    static inline int64_t center(uint64_t u) { return u > Q / 2 ? (int64_t)(u - Q) : (int64_t)u; }

    static unsigned brv(unsigned x, int bits) {
        unsigned r = 0;
        for (int i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
        return r;
    }

    explicit Ntt(int log_n_) : log_n(log_n_), n(1 << log_n_) {
        if (log_n < 1 || log_n > 12) throw std::runtime_error("oracle Ntt: log_n out of range");
        uint64_t psi = powmod(PSI_8192, (uint64_t)(4096 / n));  // primitive 2n-th root
        uint64_t ipsi = powmod(psi, Q - 2);
        w.resize(n); ws.resize(n); iw.resize(n); iws.resize(n);
        for (int i = 0; i < n; i++) {
            unsigned e = brv((unsigned)i, log_n);
            w[i] = powmod(psi, e);  ws[i] = shoup(w[i]);
            iw[i] = powmod(ipsi, e); iws[i] = shoup(iw[i]);
        }
        n_inv = powmod((uint64_t)n, Q - 2);
        n_inv_s = shoup(n_inv);
    }

    // in place, natural order in -> bit-reversed order out (merged-psi Cooley-Tukey)
    void fwd(uint64_t* a) const {
        int t = n;
        for (int m = 1; m < n; m <<= 1) {
            t >>= 1;
            for (int i = 0; i < m; i++) {
                const uint64_t wi = w[m + i], wsi = ws[m + i];
                uint64_t* x = a + 2 * i * t;
                uint64_t* y = x + t;
                for (int j = 0; j < t; j++) {
                    uint64_t u = x[j];
                    uint64_t v = mul_shoup(y[j], wi, wsi);
                    x[j] = addmod(u, v);
                    y[j] = submod(u, v);
                }
            }
        }
    }
    // in place, bit-reversed in -> natural out (Gentleman-Sande), scaled by n^-1
    void inv(uint64_t* a) const {
        int t = 1;
        for (int m = n >> 1; m >= 1; m >>= 1) {
            for (int i = 0; i < m; i++) {
                const uint64_t wi = iw[m + i], wsi = iws[m + i];
                uint64_t* x = a + 2 * i * t;
                uint64_t* y = x + t;
                for (int j = 0; j < t; j++) {
                    uint64_t u = x[j], v = y[j];
                    x[j] = addmod(u, v);
                    y[j] = mul_shoup(submod(u, v), wi, wsi);
                }
            }
            t <<= 1;
        }
        for (int j = 0; j < n; j++) a[j] = mul_shoup(a[j], n_inv, n_inv_s);
    }
};

// A polynomial in the transform domain, with Shoup companions so it can be used as the
// constant operand of a pointwise multiply-accumulate.  This is the oracle's stand-in for
// Poulpy's opaque "prepared" scalar type (SvpPPol / VmpPMat entries).
struct PolyHat {
    std::vector<uint64_t> v, s;
};

static inline void to_hat(const Ntt& ntt, const int64_t* a, std::vector<uint64_t>& out) {
    out.resize(ntt.n);
    for (int i = 0; i < ntt.n; i++) out[i] = Ntt::to_mod(a[i]);
    ntt.fwd(out.data());
}
static inline void to_hat_prepared(const Ntt& ntt, const int64_t* a, PolyHat& out) {
    to_hat(ntt, a, out.v);
    out.s.resize(ntt.n);
    for (int i = 0; i < ntt.n; i++) out.s[i] = Ntt::shoup(out.v[i]);
}
// acc += x (.) g
static inline void mac_hat(const Ntt& ntt, std::vector<uint64_t>& acc, const std::vector<uint64_t>& x, const PolyHat& g) {
    for (int i = 0; i < ntt.n; i++) acc[i] = Ntt::addmod(acc[i], Ntt::mul_shoup(x[i], g.v[i], g.s[i]));
}
// inverse transform + centred lift into i64; returns max |coeff| seen
static inline int64_t from_hat(const Ntt& ntt, std::vector<uint64_t>& acc, int64_t* out) {
    ntt.inv(acc.data());
    int64_t mx = 0;
    for (int i = 0; i < ntt.n; i++) {
        int64_t c = Ntt::center(acc[i]);
        out[i] = c;
        int64_t m = c < 0 ? -c : c;
        if (m > mx) mx = m;
    }
    return mx;
}

// O(n^2) schoolbook negacyclic product in 128-bit integers, for cross-checking the NTT at
// small n (tests only).
static inline void negacyclic_schoolbook(int n, const int64_t* a, const int64_t* b, int64_t* out) {
    for (int k = 0; k < n; k++) {
        __int128 s = 0;
        for (int i = 0; i < n; i++) {
            int j = k - i;
            if (j >= 0) s += (__int128)a[i] * b[j];
            else s -= (__int128)a[i] * b[j + n];
        }
        out[k] = (int64_t)s;
    }
}

}  // namespace fo
