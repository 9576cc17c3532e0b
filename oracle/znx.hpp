// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see oracle/README.md).
//
// Exact-integer restatement of the limb ("Znx") arithmetic that phantomzone-org/fhe-ram
// reaches through Poulpy (poulpy-hal 0.3.2, un-vendored: /root/reference/Cargo.toml:7-10,
// Cargo.lock:376-435).  Poulpy's source is absent from /root/reference, so every routine in
// this file restates Poulpy's published algorithm as recalled ([UPSTREAM-RECALL] in
// SURVEY.md Appendix A) and is anchored on the reference's call sites, cited per function.
//
// Nothing under fhe-ram_amd/ may include, link or call this file.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>
#include <cassert>

namespace fo {

// ---------------------------------------------------------------------------------------
// Digit / carry extraction base 2^base2k.
// digit(x) = sign-extended low base2k bits, in [-2^(base2k-1), 2^(base2k-1)).
// ---------------------------------------------------------------------------------------
static inline int64_t get_digit(int base2k, int64_t x) {
    const int sh = 64 - base2k;
    return (int64_t)((uint64_t)x << sh) >> sh;
}
static inline int64_t get_carry(int base2k, int64_t x, int64_t digit) {
    return (int64_t)((uint64_t)x - (uint64_t)digit) >> base2k;
}

// One normalisation step on one coefficient.  `lsh` is the "shifted normalisation" used by
// rsh (value is implicitly multiplied by 2^lsh before digit extraction).
// c is the running carry.  Returns the output digit.
static inline void norm_first_carry_only(int base2k, int lsh, int64_t x, int64_t& c) {
    const int b = (lsh == 0) ? base2k : base2k - lsh;
    c = get_carry(b, x, get_digit(b, x));
}
static inline int64_t norm_first(int base2k, int lsh, int64_t x, int64_t& c) {
    if (lsh == 0) {
        int64_t d = get_digit(base2k, x);
        c = get_carry(base2k, x, d);
        return d;
    }
    const int b = base2k - lsh;
    int64_t d = get_digit(b, x);
    c = get_carry(b, x, d);
    return (int64_t)((uint64_t)d << lsh);
}
static inline void norm_middle_carry_only(int base2k, int lsh, int64_t x, int64_t& c) {
    const int b = (lsh == 0) ? base2k : base2k - lsh;
    int64_t d = get_digit(b, x);
    int64_t cr = get_carry(b, x, d);
    int64_t dpc = (int64_t)((uint64_t)d << lsh) + c;
    c = cr + get_carry(base2k, dpc, get_digit(base2k, dpc));
}
static inline int64_t norm_middle(int base2k, int lsh, int64_t x, int64_t& c) {
    const int b = (lsh == 0) ? base2k : base2k - lsh;
    int64_t d = get_digit(b, x);
    int64_t cr = get_carry(b, x, d);
    int64_t dpc = (int64_t)((uint64_t)d << lsh) + c;
    int64_t out = get_digit(base2k, dpc);
    c = cr + get_carry(base2k, dpc, out);
    return out;
}
static inline int64_t norm_final(int base2k, int lsh, int64_t x, int64_t c) {
    const int b = (lsh == 0) ? base2k : base2k - lsh;
    return get_digit(base2k, (int64_t)((uint64_t)get_digit(b, x) << lsh) + c);
}

// ---------------------------------------------------------------------------------------
// VecZnx view: (n, cols, size) with element (col i, limb j, coeff c) at n*(j*cols+i)+c
// (SURVEY.md A.2).  Non-owning.
// ---------------------------------------------------------------------------------------
struct VecView {
    int64_t* p;
    int n, cols, size;
    int64_t* at(int col, int limb) const { return p + (size_t)n * ((size_t)limb * cols + col); }
    size_t len() const { return (size_t)n * cols * size; }
};

// "Big" (post-inverse-transform) accumulator: [col][limb][n] i64, owning.
struct Big {
    int n, cols, size;
    std::vector<int64_t> d;
    Big(int n_, int cols_, int size_) : n(n_), cols(cols_), size(size_), d((size_t)n_ * cols_ * size_, 0) {}
    int64_t* at(int col, int limb) { return d.data() + ((size_t)col * size + limb) * n; }
    const int64_t* at(int col, int limb) const { return d.data() + ((size_t)col * size + limb) * n; }
};

// vec_znx_big_normalize, same base in and out: res column <- normalised a column.
// Limbs j >= res_size of `a` contribute only through the carry (round to nearest at the cut);
// the final carry out of limb 0 is dropped (values live on the torus, mod 1).
// Reached from glwe_external_product / keyswitch / automorphism
// (reference call sites: coordinate_prepared.rs:156-158,175; ram.rs:435,457).
static inline void big_normalize(int base2k, int n, int64_t* const* res, int res_size,
                                 const int64_t* const* a, int a_size) {
    for (int i = 0; i < n; i++) {
        int64_t c = 0;
        if (a_size > res_size) {
            for (int j = a_size - 1; j >= res_size; j--) {
                if (j == a_size - 1) norm_first_carry_only(base2k, 0, a[j][i], c);
                else norm_middle_carry_only(base2k, 0, a[j][i], c);
            }
            for (int j = res_size - 1; j >= 1; j--) res[j][i] = norm_middle(base2k, 0, a[j][i], c);
            res[0][i] = norm_final(base2k, 0, a[0][i], c);
        } else {
            for (int j = a_size - 1; j >= 0; j--) {
                if (j == a_size - 1 && j != 0) res[j][i] = norm_first(base2k, 0, a[j][i], c);
                else if (j == a_size - 1 && j == 0) res[j][i] = get_digit(base2k, a[j][i]);
                else if (j == 0) res[j][i] = norm_final(base2k, 0, a[j][i], c);
                else res[j][i] = norm_middle(base2k, 0, a[j][i], c);
            }
            for (int j = a_size; j < res_size; j++) res[j][i] = 0;
        }
    }
}

// vec_znx_normalize_inplace on one column (glwe_normalize_inplace: ram.rs:576,626).
static inline void normalize_inplace(int base2k, const VecView& v, int col) {
    std::vector<int64_t*> ptr(v.size);
    for (int j = 0; j < v.size; j++) ptr[j] = v.at(col, j);
    big_normalize(base2k, v.n, ptr.data(), v.size, ptr.data(), v.size);
}

// vec_znx_rsh_inplace(base2k, k) on one column: divides the torus value by 2^k.
// Implemented, as in Poulpy's reference backend, as a shift by whole limbs followed by a
// "left-shifted normalisation" (lsh = base2k - k%base2k), so the output is normalised and
// the bits shifted out of the last limb are rounded away through the carry
// (x_last = 2c + d, d in {0,-1}  =>  c = ceil(x_last/2) for k = 1).
// Reached from GLWE::trace_inplace and GLWEPacker combine (ram.rs:435,457,540,572,616,621).
// [UPSTREAM-RECALL; exact rounding rule frozen here — SURVEY.md A.5]
static inline void rsh_inplace(int base2k, int k, const VecView& v, int col) {
    if (k == 0) return;
    const int n = v.n, size = v.size;
    int steps = k / base2k;
    const int k_rem = k % base2k;
    if (steps >= size) {
        for (int j = 0; j < size; j++) std::memset(v.at(col, j), 0, sizeof(int64_t) * n);
        return;
    }
    if (k_rem != 0) {
        steps += 1;
        const int lsh = base2k - k_rem;
        for (int i = 0; i < n; i++) {
            int64_t c = 0;
            // limbs that fall off the end: carry only
            for (int j = size - 1; j >= size - steps; j--) {
                if (j == size - 1) norm_first_carry_only(base2k, lsh, v.at(col, j)[i], c);
                else norm_middle_carry_only(base2k, lsh, v.at(col, j)[i], c);
            }
            // shifted normalisation: out[j] from in[j-steps]
            for (int j = size - 1; j >= steps; j--) {
                v.at(col, j)[i] = norm_middle(base2k, lsh, v.at(col, j - steps)[i], c);
            }
            // propagate the carry into the vacated top limbs
            for (int j = steps - 1; j >= 0; j--) {
                if (j == 0) v.at(col, j)[i] = norm_final(base2k, lsh, 0, c);
                else {
                    int64_t z = 0;
                    v.at(col, j)[i] = norm_middle(base2k, lsh, z, c);
                }
            }
        }
    } else {
        for (int j = size - 1; j >= steps; j--)
            std::memcpy(v.at(col, j), v.at(col, j - steps), sizeof(int64_t) * n);
        for (int j = 0; j < steps; j++) std::memset(v.at(col, j), 0, sizeof(int64_t) * n);
    }
}

// res = a * X^k in Z[X]/(X^n+1), one polynomial (vec_znx_rotate; glwe_rotate_inplace ram.rs:629).
static inline void poly_rotate(int n, int64_t k, int64_t* res, const int64_t* a) {
    const int64_t two_n = 2 * (int64_t)n;
    int64_t kk = ((k % two_n) + two_n) % two_n;
    for (int i = 0; i < n; i++) {
        int64_t j = i + kk;
        bool neg = false;
        if (j >= two_n) j -= two_n;
        if (j >= n) { j -= n; neg = true; }
        res[j] = neg ? -a[i] : a[i];
    }
}

// res(X) = a(X^g), g odd (vec_znx_automorphism): coefficient i goes to i*g mod 2n, negated
// when the image index is >= n.
static inline void poly_automorphism(int n, int64_t g, int64_t* res, const int64_t* a) {
    const int64_t two_n = 2 * (int64_t)n;
    int64_t gg = ((g % two_n) + two_n) % two_n;
    for (int i = 0; i < n; i++) {
        int64_t j = ((int64_t)i * gg) % two_n;
        if (j >= n) res[j - n] = -a[i];
        else res[j] = a[i];
    }
}

}  // namespace fo
