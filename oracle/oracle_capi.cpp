// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see oracle/README.md).
// C entry points (ctypes) over the C++ restatement.  Loaded only by tests/, smoke() and
// bench.py's cpu_baseline leg through oracle/pyoracle.py.
#include <omp.h>
#include "setup.hpp"
#include <string>

using namespace fo;

static thread_local std::string g_err;
#define FO_TRY try {
#define FO_CATCH } catch (const std::exception& e) { g_err = e.what(); return 1; } return 0;

struct KeysHandle { EvaluationKeysPrepared k; };
struct AddrHandle { Address a; };

extern "C" {

const char* fo_last_error() { return g_err.c_str(); }

void* fo_ctx_new(int log_n, int base2k, int k_pt, int k_ct, int k_addr, int k_evk_trace, int k_evk_inv,
                 const uint8_t* decomp_n, int n_decomp, int word_size, uint64_t max_addr) {
    try {
        Params p;
        p.log_n = log_n; p.base2k = base2k; p.k_glwe_pt = k_pt; p.k_glwe_ct = k_ct; p.k_ggsw_addr = k_addr;
        p.k_evk_trace = k_evk_trace; p.k_evk_ggsw_inv = k_evk_inv;
        p.decomp_n.assign(decomp_n, decomp_n + n_decomp);
        p.word_size = word_size; p.max_addr = (size_t)max_addr;
        return new Ctx(p);
    } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}
void fo_ctx_free(void* c) { delete (Ctx*)c; }
int64_t fo_ctx_max_big(void* c) { return ((Ctx*)c)->max_big; }
void fo_ctx_reset_stats(void* c) { Ctx* x = (Ctx*)c; x->max_big = 0; x->n_ep = x->n_ks = x->n_prepare = 0; }
// all-core variant: sub-RAMs and the rows of the per-row loops run on `threads` OpenMP threads
void fo_ctx_set_threads(void* c, int threads) { ((Ctx*)c)->threads = threads < 1 ? 1 : threads; omp_set_max_active_levels(2); }
int fo_ctx_threads(void* c) { return ((Ctx*)c)->threads; }
void fo_ctx_counters(void* c, uint64_t* out) { Ctx* x = (Ctx*)c; out[0] = x->n_ep; out[1] = x->n_ks; out[2] = x->n_prepare; }

// ---- base.rs / lib.rs -------------------------------------------------------------
uint64_t fo_base1d_max(const uint8_t* d, int n) { Base1D b; b.d.assign(d, d + n); return b.max(); }
uint64_t fo_base1d_gap(const uint8_t* d, int n, uint64_t log_n) { Base1D b; b.d.assign(d, d + n); return b.gap(log_n); }
void fo_base1d_decomp(const uint8_t* d, int n, uint32_t value, uint8_t* out) {
    Base1D b; b.d.assign(d, d + n); auto r = b.decomp(value); for (int i = 0; i < n; i++) out[i] = r[i];
}
uint32_t fo_base1d_recomp(const uint8_t* d, int n, const uint8_t* dec) {
    Base1D b; b.d.assign(d, d + n); std::vector<uint8_t> v(dec, dec + n); return b.recomp(v);
}
// returns number of Base1D; digits written flat into out_digits, lengths into out_lens
int fo_get_base_2d(uint32_t value, const uint8_t* base, int n_base, uint8_t* out_digits, int* out_lens, int max_out) {
    std::vector<uint8_t> b(base, base + n_base);
    Base2D r = get_base_2d(value, b);
    int k = 0, q = 0;
    for (auto& v : r.v) {
        if (q >= max_out) return -1;
        out_lens[q++] = (int)v.d.size();
        for (auto x : v.d) out_digits[k++] = x;
    }
    return q;
}
uint64_t fo_reverse_bits_msb(uint64_t x, uint32_t n) { return reverse_bits_msb((size_t)x, n); }
int64_t fo_galois_element(int log_n, int i) { return galois_element(log_n, i); }
int64_t fo_cast_u8_to_signed(uint8_t v, int bits) { return cast_u8_to_signed(v, bits); }

// ---- znx ---------------------------------------------------------------------------
void fo_big_normalize(int base2k, int n, int res_size, int a_size, const int64_t* a, int64_t* res) {
    std::vector<int64_t*> rp(res_size); std::vector<const int64_t*> ap(a_size);
    for (int j = 0; j < res_size; j++) rp[j] = res + (size_t)j * n;
    for (int j = 0; j < a_size; j++) ap[j] = a + (size_t)j * n;
    big_normalize(base2k, n, rp.data(), res_size, ap.data(), a_size);
}
void fo_glwe_rsh(void* c, int k, int64_t* glwe, int size) { Ctx* x = (Ctx*)c; glwe_rsh(*x, k, glwe_view(glwe, x->n(), size)); }
void fo_glwe_normalize(void* c, int64_t* glwe, int size) { Ctx* x = (Ctx*)c; glwe_normalize_inplace(*x, glwe_view(glwe, x->n(), size)); }
void fo_glwe_rotate(void* c, int64_t k, const int64_t* in, int64_t* out, int size) {
    Ctx* x = (Ctx*)c; glwe_rotate(*x, k, glwe_view(out, x->n(), size), glwe_view(const_cast<int64_t*>(in), x->n(), size));
}
void fo_poly_automorphism(int n, int64_t g, const int64_t* in, int64_t* out) { poly_automorphism(n, g, out, in); }
void fo_negacyclic_ntt(void* c, const int64_t* a, const int64_t* b, int64_t* out) {
    Ctx* x = (Ctx*)c; PolyHat h; to_hat_prepared(x->ntt, b, h); poly_mul_small(*x, a, h, out);
}
void fo_negacyclic_schoolbook(int n, const int64_t* a, const int64_t* b, int64_t* out) { negacyclic_schoolbook(n, a, b, out); }

// ---- setup -------------------------------------------------------------------------
void fo_secret_gen(void* c, uint64_t seed, int64_t* sk) { Source xs(seed); secret_fill_ternary_prob(*(Ctx*)c, sk, 0.5, xs); }
int fo_evk_gen(void* c, const int64_t* sk, uint64_t seed_a, uint64_t seed_e, int64_t* gal_els, int64_t* atk_glwe,
               int64_t* atk_inv, int64_t* tsk) {
    FO_TRY
    Ctx* x = (Ctx*)c; Source xa(seed_a), xe(seed_e); EvaluationKeysStd k;
    evaluation_keys_encrypt_sk(*x, k, sk, xa, xe);
    for (size_t i = 0; i < k.gal_els.size(); i++) {
        gal_els[i] = k.gal_els[i];
        std::memcpy(atk_glwe + i * x->p.atk_trace_len(), k.atk_glwe[i].data(), sizeof(int64_t) * x->p.atk_trace_len());
    }
    std::memcpy(atk_inv, k.atk_ggsw_inv.data(), sizeof(int64_t) * k.atk_ggsw_inv.size());
    std::memcpy(tsk, k.tsk.data(), sizeof(int64_t) * k.tsk.size());
    FO_CATCH
}
int fo_ram_encrypt(void* c, const uint8_t* data, uint64_t len, const int64_t* sk, uint64_t seed_a, uint64_t seed_e, int64_t* rows_out) {
    FO_TRY
    Source xa(seed_a), xe(seed_e);
    ram_encrypt_sk(*(Ctx*)c, data, (size_t)len, sk, xa, xe, rows_out);
    FO_CATCH
}
// total number of GGSW digits of an address for this ctx's parameters
int fo_address_n_digits(void* c) { return (int)((Ctx*)c)->p.base2d().as_1d().size(); }
int fo_address_encrypt(void* c, uint32_t value, const int64_t* sk, uint64_t seed_a, uint64_t seed_e, int64_t* ggsw_out) {
    FO_TRY
    Ctx* x = (Ctx*)c; Source xa(seed_a), xe(seed_e); Address a;
    address_encrypt_sk(*x, a, value, sk, xa, xe);
    size_t k = 0;
    for (auto& co : a.coordinates) for (auto& g : co.value) { std::memcpy(ggsw_out + k * x->p.ggsw_len(), g.data(), sizeof(int64_t) * g.size()); k++; }
    FO_CATCH
}
int fo_glwe_encrypt_coeff0(void* c, uint8_t value, const int64_t* sk, uint64_t seed_a, uint64_t seed_e, int64_t* ct) {
    FO_TRY
    Source xa(seed_a), xe(seed_e);
    encrypt_glwe_coeff0(*(Ctx*)c, ct, value, sk, xa, xe);
    FO_CATCH
}
int fo_glwe_decrypt(void* c, const int64_t* ct, int64_t want, const int64_t* sk, int coeff, int64_t* value, double* noise) {
    FO_TRY
    decrypt_glwe(*(Ctx*)c, ct, want, sk, value, noise, coeff);
    FO_CATCH
}
// The oracle's sampler as a handle, so that a test can replay exactly the draws `fo_*_encrypt` make
// from a seed (masks: one uniform limb per call; noise: Box-Muller with rejection) and hand them to
// the device-side setup entry points (include/fheram.h, "Setup side on the device").
void* fo_source_new(uint64_t seed) { return new Source(seed); }
void fo_source_free(void* s) { delete (Source*)s; }
void fo_source_uniform_limbs(void* s, int base2k, uint64_t count, int64_t* out) {
    Source* x = (Source*)s;
    for (uint64_t i = 0; i < count; i++) out[i] = x->uniform_limb(base2k);
}
void fo_source_gaussian(void* s, double sigma, double bound, uint64_t count, int64_t* out) {
    Source* x = (Source*)s;
    for (uint64_t i = 0; i < count; i++) out[i] = x->gaussian(sigma, bound);
}
double fo_sigma(void* c) { return ((Ctx*)c)->p.sigma; }
// generic GLWE::encrypt_sk of `size` limbs at precision k; pt [pt_size][n] or null
int fo_glwe_encrypt_sk(void* c, int size, int k, const int64_t* pt, int pt_size, int pt_col, const int64_t* sk,
                       uint64_t seed_a, uint64_t seed_e, int64_t* ct) {
    FO_TRY
    Ctx* x = (Ctx*)c; Source xa(seed_a), xe(seed_e); PolyHat h; to_hat_prepared(x->ntt, sk, h);
    glwe_encrypt_sk(*x, glwe_view(ct, x->n(), size), pt, pt_size, pt_col, h, k, xa, xe);
    FO_CATCH
}
// phase: pt = normalise(body + mask * s), [size][n]
int fo_glwe_phase(void* c, int size, const int64_t* ct, const int64_t* sk, int64_t* pt) {
    FO_TRY
    Ctx* x = (Ctx*)c; const int n = x->n();
    VecView v = glwe_view(const_cast<int64_t*>(ct), n, size);
    PolyHat h; to_hat_prepared(x->ntt, sk, h);
    Big big(n, 1, size);
    for (int j = 0; j < size; j++) {
        poly_mul_small(*x, v.at(1, j), h, big.at(0, j));
        int64_t* b = big.at(0, j); const int64_t* body = v.at(0, j);
        for (int i = 0; i < n; i++) b[i] += body[i];
    }
    VecView pv{pt, n, 1, size};
    big_normalize_col(*x, pv, 0, big, 0);
    FO_CATCH
}
// generic GGSW encryption of a scalar polynomial (for op-level tests)
int fo_ggsw_encrypt(void* c, const int64_t* scalar, const int64_t* sk, uint64_t seed_a, uint64_t seed_e, int64_t* out) {
    FO_TRY
    Ctx* x = (Ctx*)c; Source xa(seed_a), xe(seed_e); PolyHat h; to_hat_prepared(x->ntt, sk, h);
    ggsw_encrypt_sk(*x, out, x->p.dnum_ct(), x->p.size_addr(), x->p.k_ggsw_addr, scalar, h, xa, xe);
    FO_CATCH
}

// ---- N4: Address::set_from_fheuint (conversion.rs:18-82), self-consistent restatement ------------
size_t fo_fheuint_ggsw_len(void* c) { Ctx* x = (Ctx*)c; return (size_t)x->p.dnum_ggsw() * 2 * x->p.glwe_len(x->p.size_evk_inv()); }
// FheUintPrepared::encrypt_sk stand-in: one GGSW per bit (LSB first), out [n_bits][fo_fheuint_ggsw_len]
int fo_fheuint_encrypt(void* c, uint32_t value, int n_bits, const int64_t* sk, uint64_t seed_a, uint64_t seed_e, int64_t* out) {
    FO_TRY
    Ctx* x = (Ctx*)c; Source xa(seed_a), xe(seed_e); PolyHat h; to_hat_prepared(x->ntt, sk, h);
    std::vector<int64_t> scalar(x->n(), 0);
    for (int i = 0; i < n_bits; i++) {
        scalar[0] = (value >> i) & 1;
        ggsw_encrypt_sk(*x, out + (size_t)i * fo_fheuint_ggsw_len(c), x->p.dnum_ggsw(), x->p.size_evk_inv(), x->p.k_evk_ggsw_inv, scalar.data(), h, xa, xe);
    }
    FO_CATCH
}
int fo_address_from_fheuint(void* c, const int64_t* bits_std, int n_bits, int sign, int64_t* out_digits) {
    FO_TRY
    Ctx* x = (Ctx*)c;
    FheUint fu; fheuint_prepare(*x, fu, bits_std, n_bits);
    Address a; address_set_from_fheuint(*x, a, fu, sign != 0);
    size_t k = 0;
    for (auto& co : a.coordinates) for (auto& g : co.value) { std::memcpy(out_digits + k * x->p.ggsw_len(), g.data(), sizeof(int64_t) * g.size()); k++; }
    FO_CATCH
}

// ---- prepared keys / address handles --------------------------------------------------
void* fo_keys_prepare(void* c, const int64_t* gal_els, int n_gal, const int64_t* atk_glwe, const int64_t* atk_inv, const int64_t* tsk) {
    try {
        Ctx* x = (Ctx*)c; KeysHandle* h = new KeysHandle();
        std::vector<const int64_t*> ptr(n_gal);
        for (int i = 0; i < n_gal; i++) ptr[i] = atk_glwe + (size_t)i * x->p.atk_trace_len();
        evaluation_keys_prepare(*x, h->k, gal_els, n_gal, ptr.data(), atk_inv, tsk);
        return h;
    } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}
void fo_keys_free(void* k) { delete (KeysHandle*)k; }
void* fo_address_new(void* c, const int64_t* ggsw_all) {
    try {
        Ctx* x = (Ctx*)c; AddrHandle* h = new AddrHandle();
        h->a.base2d = x->p.base2d();
        size_t k = 0;
        for (auto& b : h->a.base2d.v) {
            Coordinate co; co.base1d = b;
            for (size_t d = 0; d < b.d.size(); d++) { co.value.emplace_back(ggsw_all + k * x->p.ggsw_len(), ggsw_all + (k + 1) * x->p.ggsw_len()); k++; }
            h->a.coordinates.push_back(std::move(co));
        }
        return h;
    } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}
void fo_address_free(void* a) { delete (AddrHandle*)a; }

// ---- Poulpy-level ops ----------------------------------------------------------------
int fo_glwe_external_product(void* c, const int64_t* a, const int64_t* ggsw, int64_t* res) {
    FO_TRY
    Ctx* x = (Ctx*)c; const int n = x->n(), size = x->p.size_ct();
    MatPrepared m = prepare_mat(*x, ggsw, x->p.dnum_ct(), 2, x->p.size_addr());
    std::vector<int64_t> tmp(a, a + x->p.glwe_len(size));
    glwe_external_product(*x, glwe_view(res, n, size), glwe_view(tmp.data(), n, size), m);
    FO_CATCH
}
// mode 0: res = phi(KS(a)); 1: res = a + phi(KS(a)) (automorphism_add_inplace); 2: res = a - phi(KS(a))
int fo_glwe_automorphism(void* c, void* keys, int64_t gal_el, int mode, const int64_t* a, int64_t* res) {
    FO_TRY
    Ctx* x = (Ctx*)c; KeysHandle* k = (KeysHandle*)keys; const int n = x->n(), size = x->p.size_ct();
    auto it = k->k.atk_glwe.find(gal_el);
    if (it == k->k.atk_glwe.end()) throw std::runtime_error("missing automorphism key");
    std::vector<int64_t> tmp(a, a + x->p.glwe_len(size));
    if (mode == 0) glwe_automorphism(*x, glwe_view(res, n, size), glwe_view(tmp.data(), n, size), it->second);
    else if (mode == 1) { std::memcpy(res, a, sizeof(int64_t) * tmp.size()); glwe_automorphism_add_inplace(*x, glwe_view(res, n, size), it->second); }
    else glwe_automorphism_sub_negate(*x, glwe_view(res, n, size), glwe_view(tmp.data(), n, size), it->second);
    FO_CATCH
}
int fo_glwe_trace(void* c, void* keys, int start, int end, int64_t* a) {
    FO_TRY
    Ctx* x = (Ctx*)c; KeysHandle* k = (KeysHandle*)keys;
    glwe_trace_inplace(*x, glwe_view(a, x->n(), x->p.size_ct()), start, end, k->k.atk_glwe);
    FO_CATCH
}
// Feed n items (present[i] != 0 => cts + idx*glen where idx counts present items) then flush.
int fo_glwe_pack(void* c, void* keys, const int64_t* cts, const uint8_t* present, int64_t* out) {
    FO_TRY
    Ctx* x = (Ctx*)c; KeysHandle* k = (KeysHandle*)keys;
    Packer pk; pk.alloc(*x, x->p.size_ct());
    const size_t glen = x->p.glwe_len(x->p.size_ct());
    size_t idx = 0;
    for (int i = 0; i < x->n(); i++) {
        if (present[i]) { packer_add(*x, pk, cts + idx * glen, k->k.atk_glwe); idx++; }
        else packer_add(*x, pk, nullptr, k->k.atk_glwe);
    }
    packer_flush(*x, pk, out);
    FO_CATCH
}
// One GLWEPacker combine at level i on an accumulator that holds a value (acc.value == true):
// a <- combine(a, b) with b == NULL meaning "no ciphertext" (SURVEY.md A.7).  Used by the tests to
// replay the packer level by level (row-sharded schedule).
int fo_packer_combine(void* c, void* keys, int64_t* a, const int64_t* b, int i) {
    FO_TRY
    Ctx* x = (Ctx*)c; KeysHandle* k = (KeysHandle*)keys;
    Packer pk; pk.n = x->n(); pk.size = x->p.size_ct();
    Accumulator acc; acc.value = true; acc.control = true;
    acc.data.assign(a, a + x->p.glwe_len(pk.size));
    packer_combine(*x, pk, acc, b, i, k->k.atk_glwe);
    std::memcpy(a, acc.data.data(), sizeof(int64_t) * acc.data.size());
    FO_CATCH
}
int fo_ggsw_automorphism_inv(void* c, void* keys, const int64_t* in, int64_t* out) {
    FO_TRY
    Ctx* x = (Ctx*)c; KeysHandle* k = (KeysHandle*)keys;
    ggsw_automorphism(*x, out, in, x->p.dnum_ct(), x->p.size_addr(), k->k.atk_ggsw_inv, k->k.tsk_ggsw_inv);
    FO_CATCH
}

// ---- Ram ------------------------------------------------------------------------------
void* fo_ram_new(void* c) { try { return new Ram((Ctx*)c); } catch (const std::exception& e) { g_err = e.what(); return nullptr; } }
void fo_ram_free(void* r) { delete (Ram*)r; }
// rows: [word_size][rows][size_ct][2][n]
void fo_ram_load(void* r, const int64_t* rows) {
    Ram* ram = (Ram*)r; const size_t nrows = ram->c->p.rows(), glen = ram->glen();
    for (size_t i = 0; i < ram->subrams.size(); i++) {
        ram->subrams[i].data.assign(nrows, std::vector<int64_t>(glen));
        for (size_t q = 0; q < nrows; q++) std::memcpy(ram->subrams[i].data[q].data(), rows + (i * nrows + q) * glen, sizeof(int64_t) * glen);
    }
}
void fo_ram_store(void* r, int64_t* rows) {
    Ram* ram = (Ram*)r; const size_t nrows = ram->c->p.rows(), glen = ram->glen();
    for (size_t i = 0; i < ram->subrams.size(); i++)
        for (size_t q = 0; q < nrows; q++) std::memcpy(rows + (i * nrows + q) * glen, ram->subrams[i].data[q].data(), sizeof(int64_t) * glen);
}
// tree level `lvl`, entry 0 of every sub-RAM: [word_size][glen]; returns 1 if the level does not exist
int fo_ram_tree(void* r, int lvl, int64_t* out) {
    Ram* ram = (Ram*)r; const size_t glen = ram->glen();
    for (size_t i = 0; i < ram->subrams.size(); i++) {
        if ((size_t)lvl >= ram->subrams[i].tree.size()) return 1;
        std::memcpy(out + i * glen, ram->subrams[i].tree[lvl][0].data(), sizeof(int64_t) * glen);
    }
    return 0;
}
int fo_ram_state(void* r) { Ram* ram = (Ram*)r; return ram->subrams.empty() ? 0 : (int)ram->subrams[0].state; }
int fo_ram_read(void* r, void* addr, void* keys, int64_t* out) {
    FO_TRY ((Ram*)r)->read(((AddrHandle*)addr)->a, ((KeysHandle*)keys)->k, out); FO_CATCH
}
int fo_ram_read_prepare_write(void* r, void* addr, void* keys, int64_t* out) {
    FO_TRY ((Ram*)r)->read_prepare_write(((AddrHandle*)addr)->a, ((KeysHandle*)keys)->k, out); FO_CATCH
}
int fo_ram_write(void* r, const int64_t* w, int n_w, void* addr, void* keys) {
    FO_TRY ((Ram*)r)->write(w, (size_t)n_w, ((AddrHandle*)addr)->a, ((KeysHandle*)keys)->k); FO_CATCH
}

}  // extern "C"
