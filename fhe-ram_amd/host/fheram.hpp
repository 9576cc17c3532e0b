// fheram.hpp — host-side mirror of the reference crate's public interface for the RAM path, in
// C++ (the reference is compiled Rust; this image has no Rust toolchain, see INTEGRATION.md for the
// Rust `extern "C"` binding a maintainer would add).  Header-only, over the C ABI of
// include/fheram.h.  Same names, argument meaning and error behaviour as
//   Parameters              /root/reference/src/parameters.rs:147-287
//   EvaluationKeysPrepared  /root/reference/src/keys.rs:27-71
//   Address                 /root/reference/src/address.rs:21-119
//   Ram                     /root/reference/src/ram.rs:25-294
// The reference panics on misuse (assert!); these classes throw fheram::Error with the same text.
#pragma once
#include "../../include/fheram.h"

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace fheram {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

// GLWE<Vec<u8>> stand-in: int64 limbs in Poulpy's host layout [limb][col][N].
using Glwe = std::vector<int64_t>;

// parameters.rs:147-176
struct Parameters {
    fheram_params p;
    Parameters() { fheram_params_default(&p); }                                   // Parameters::new()
    Parameters(size_t word_size, const std::vector<uint8_t>& decomp_n, size_t max_addr) : Parameters() {   // ram.rs:72-78
        p.word_size = (uint32_t)word_size;
        p.max_addr = max_addr;
        p.n_decomp = (uint32_t)decomp_n.size();
        for (size_t i = 0; i < decomp_n.size() && i < 16; i++) p.decomp_n[i] = decomp_n[i];
    }
    size_t max_addr() const { return p.max_addr; }            // parameters.rs:237
    size_t word_size() const { return p.word_size; }          // parameters.rs:261
    uint32_t basek() const { return p.base2k; }
    uint32_t rank() const { return p.rank; }
    uint32_t k_glwe_ct() const { return p.k_glwe_ct; }
    uint32_t k_glwe_pt() const { return p.k_glwe_pt; }
    size_t n() const { return (size_t)1 << p.log_n; }
};

// keys.rs:27-31 — std-form keys; `prepare` (keys.rs:57-71) happens on the device at first use.
struct EvaluationKeysPrepared {
    std::vector<int64_t> gal_els;                 // GLWE::trace_galois_elements (keys.rs:39)
    std::vector<std::vector<int64_t>> atk_glwe;   // one GGLWE per Galois element (keys.rs:28)
    std::vector<int64_t> atk_ggsw_inv;            // keys.rs:29, p() must be -1
    std::vector<int64_t> tsk_ggsw_inv;            // keys.rs:30
    int64_t atk_ggsw_inv_p = -1;
};

class Ram;

// address.rs:21-24 — the GGSW digits of all coordinates, coordinate-major.
class Address {
public:
    std::vector<std::vector<int64_t>> digits;
    Address() = default;
    explicit Address(std::vector<std::vector<int64_t>> d) : digits(std::move(d)) {}
    ~Address() { if (h_) fheram_address_destroy(h_); }
    Address(const Address&) = delete;
    Address& operator=(const Address&) = delete;

private:
    friend class Ram;
    fheram_addr* h_ = nullptr;
    const void* owner_ = nullptr;
};

// ram.rs:25-29
class Ram {
public:
    Parameters params;
    explicit Ram(int device = 0) : Ram(Parameters(), device) {}                   // Ram::new, ram.rs:59
    Ram(const Parameters& prm, int device = 0) : params(prm) {
        int rc = fheram_ctx_create(&params.p, device, &ctx_);
        if (rc != FHERAM_OK) throw Error(rc, fheram_last_error(nullptr));
    }
    static Ram new_from_ram_params(size_t word_size, const std::vector<uint8_t>& decomp_n, size_t max_addr, int device = 0) {   // ram.rs:72
        return Ram(Parameters(word_size, decomp_n, max_addr), device);
    }
    Ram(Ram&& o) noexcept : params(o.params), ctx_(o.ctx_), keys_(o.keys_) { o.ctx_ = nullptr; }
    Ram(const Ram&) = delete;
    ~Ram() { if (ctx_) fheram_ctx_destroy(ctx_); }

    size_t glwe_len() const { return fheram_glwe_len(ctx_); }

    // Ram::encrypt_sk hand-over (ram.rs:129-167): rows = [word_size][rows][GLWE], already encrypted.
    void load_encrypted(const std::vector<int64_t>& rows) {
        if (rows.size() != params.word_size() * fheram_rows(ctx_) * glwe_len())
            throw Error(FHERAM_ERR_INVALID_ARG, "invalid data: data.len()/ram_chunks != max_addr (ram.rs:150-155)");
        chk(fheram_ram_upload(ctx_, rows.data()));
    }
    // Ram::read, ram.rs:172-191
    std::vector<Glwe> read(Address& address, const EvaluationKeysPrepared& keys) {
        use(keys);
        std::vector<int64_t> out(params.word_size() * glwe_len());
        chk(fheram_read(ctx_, dev(address), out.data()));
        return split(out);
    }
    // Ram::read_prepare_write, ram.rs:196-222
    std::vector<Glwe> read_prepare_write(Address& address, const EvaluationKeysPrepared& keys) {
        use(keys);
        std::vector<int64_t> out(params.word_size() * glwe_len());
        chk(fheram_read_prepare_write(ctx_, dev(address), out.data()));
        return split(out);
    }
    // Ram::write, ram.rs:226-294.  Each w[i] must encrypt [w,0,...,0] (ram.rs:228).
    void write(const std::vector<Glwe>& w, Address& address, const EvaluationKeysPrepared& keys) {
        use(keys);
        std::vector<int64_t> flat;
        for (auto& g : w) flat.insert(flat.end(), g.begin(), g.end());
        chk(fheram_write(ctx_, flat.data(), (int)w.size(), dev(address)));
    }
    bool state() const { return fheram_ram_state(ctx_) != 0; }                    // SubRam::state, ram.rs:302

private:
    fheram_ctx* ctx_ = nullptr;
    const EvaluationKeysPrepared* keys_ = nullptr;
    void chk(int rc) { if (rc != FHERAM_OK) throw Error(rc, fheram_last_error(ctx_)); }
    void use(const EvaluationKeysPrepared& k) {
        if (keys_ == &k) return;
        std::vector<const int64_t*> ptr;
        for (auto& a : k.atk_glwe) ptr.push_back(a.data());
        chk(fheram_keys_load(ctx_, k.gal_els.data(), (int)k.gal_els.size(), ptr.data(), k.atk_ggsw_inv.data(), k.atk_ggsw_inv_p,
                             k.tsk_ggsw_inv.data()));
        keys_ = &k;
    }
    fheram_addr* dev(Address& a) {
        if (a.h_ && a.owner_ == this) return a.h_;
        if (a.h_) { fheram_address_destroy(a.h_); a.h_ = nullptr; }
        std::vector<const int64_t*> ptr;
        for (auto& d : a.digits) ptr.push_back(d.data());
        chk(fheram_address_create(ctx_, ptr.data(), (int)ptr.size(), &a.h_));     // layout check of ram.rs:404
        a.owner_ = this;
        return a.h_;
    }
    std::vector<Glwe> split(const std::vector<int64_t>& flat) const {
        std::vector<Glwe> out;
        const size_t g = glwe_len();
        for (size_t i = 0; i < params.word_size(); i++) out.emplace_back(flat.begin() + i * g, flat.begin() + (i + 1) * g);
        return out;
    }
};

}  // namespace fheram
