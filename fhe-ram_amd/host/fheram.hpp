// fheram.hpp — host-side mirror of the reference crate's public interface for the RAM path, in
// C++ (the reference is compiled Rust; this image has no Rust toolchain, see INTEGRATION.md for the
// Rust `extern "C"` binding a maintainer would add).  Header-only, over the C ABI of
// include/fheram.h.  Same names, argument meaning and error behaviour as
//   Parameters              /root/reference/src/parameters.rs:147-287
//   EvaluationKeysPrepared  /root/reference/src/keys.rs:27-71
//   Address                 /root/reference/src/address.rs:21-119
//   Ram                     /root/reference/src/ram.rs:25-294
//   GLWESecret, Source      setup side: the encrypt_sk methods take the caller's samplers (source_xa, source_xe)
// The reference panics on misuse (assert!); these classes throw fheram::Error with the same text.
#pragma once
#include "../../include/fheram.h"

#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
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
    // the block README.md:36's timings were taken with (README.md:20-33): K_GLWE_PT 9, 5-limb trace keys, MAX_ADDR 2^18
    static Parameters readme() {
        Parameters q(4, {4, 4, 4, 4}, (size_t)1 << 18);
        q.p.k_glwe_pt = 9;
        q.p.k_evk_trace = 5 * q.p.base2k;
        return q;
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

// What the encrypt_sk methods need from a sampler (Poulpy's `Source` in the reference: source_xa for the
// uniform masks, source_xe for the noise).  Sampling stays on the host; the device does the arithmetic.
struct Source {
    virtual ~Source() = default;
    // `count` uniform limbs in [-2^(base2k-1), 2^(base2k-1))
    virtual void uniform_limbs(int64_t* out, size_t count) = 0;
    // `count` samples of round(N(0, (sigma*scale)^2)), truncated at 6 sigma*scale (Poulpy add_normal)
    virtual void gaussian(int64_t* out, size_t count, double scale) = 0;
};
// noise of a ciphertext at precision k sits on limb ceil(k/base2k)-1, scaled by 2^((limb+1)*base2k - k)
inline double noise_scale(uint32_t k, uint32_t base2k) { return (double)((uint64_t)1 << (((k + base2k - 1) / base2k) * base2k - k)); }

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
    // explicit execution switches (include/fheram.h: fheram_config; start from default_config() and change what differs)
    Ram(const Parameters& prm, const fheram_config& cfg, int device = 0) : params(prm) {
        int rc = fheram_ctx_create_cfg(&params.p, device, 0, 1, &cfg, &ctx_);
        if (rc != FHERAM_OK) throw Error(rc, fheram_last_error(nullptr));
    }
    static fheram_config default_config() { fheram_config c; fheram_config_default(&c); return c; }
    fheram_config config() const {                                                  // the switches in effect
        fheram_config c;
        int rc = fheram_ctx_config(ctx_, &c);
        if (rc != FHERAM_OK) throw Error(rc, "fheram_ctx_config");
        return c;
    }
    static Ram new_from_ram_params(size_t word_size, const std::vector<uint8_t>& decomp_n, size_t max_addr, int device = 0) {   // ram.rs:72
        return Ram(Parameters(word_size, decomp_n, max_addr), device);
    }
    Ram(Ram&& o) noexcept : params(o.params), ctx_(o.ctx_), keys_(o.keys_) { o.ctx_ = nullptr; }
    Ram(const Ram&) = delete;
    ~Ram() { if (ctx_) fheram_ctx_destroy(ctx_); }

    size_t glwe_len() const { return fheram_glwe_len(ctx_); }
    // The exactness contract of the FFT64 arithmetic, checked (include/fheram.h: fheram_roundoff_max): the largest |x - rint(x)| any rounding
    // has seen on this context; throws Error(FHERAM_ERR_PRECISION) once it has passed 3/8, like every call that waits for the device.
    double roundoff_max() {
        double m = 0.0;
        const int rc = fheram_roundoff_max(ctx_, &m);
        if (rc != FHERAM_OK) throw Error(rc, fheram_last_error(ctx_));
        return m;
    }
    void roundoff_reset() { const int rc = fheram_roundoff_reset(ctx_); if (rc != FHERAM_OK) throw Error(rc, fheram_last_error(ctx_)); }

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
    // Health of the single-launch chains (in-kernel hand-offs): how many were launched and how many gave up and were redone by
    // the launch behind them (0 on a GPU this context has to itself; INTEGRATION.md, "Sharing a GPU")
    struct ChainStats { uint64_t launches = 0, redone = 0; };
    ChainStats tail_stats() { ChainStats s; chk(fheram_tail_stats(ctx_, &s.launches, &s.redone)); return s; }
    ChainStats mid_stats() { ChainStats s; chk(fheram_mid_stats(ctx_, &s.launches, &s.redone)); return s; }
    // The result of the last read / read_prepare_write where the device left it: [word_size][GLWE] int64 in the context's
    // pinned host buffer (no copy on the host); valid until the next operation on this Ram.
    const int64_t* result_view() {
        const int64_t* v = nullptr;
        chk(fheram_result_map(ctx_, &v));
        return v;
    }

    // ---- setup side on the device (include/fheram.h, "Setup side on the device") -------------------
    // GLWESecret + GLWESecretPrepared, examples/fhe-ram.rs:49-59
    class Secret {
    public:
        Secret(Ram& ram, const std::vector<int64_t>& sk) : ram_(ram) { ram.chk(fheram_secret_create(ram.ctx_, sk.data(), &h_)); }
        ~Secret() { if (h_) fheram_secret_destroy(h_); }
        Secret(const Secret&) = delete;
    private:
        friend class Ram;
        Ram& ram_;
        fheram_secret* h_ = nullptr;
    };
    // Ram::encrypt_sk, ram.rs:129-167
    void encrypt_sk(const std::vector<uint8_t>& data, const Secret& sk, Source& source_xa, Source& source_xe) {
        const size_t n = params.n(), glwes = params.word_size() * fheram_rows(ctx_);
        const size_t size = (params.k_glwe_ct() + params.basek() - 1) / params.basek();
        std::vector<int64_t> mask(glwes * size * n), noise(glwes * n);
        source_xa.uniform_limbs(mask.data(), mask.size());
        source_xe.gaussian(noise.data(), noise.size(), noise_scale(params.k_glwe_ct(), params.basek()));
        chk(fheram_ram_encrypt_sk(ctx_, sk.h_, data.data(), data.size(), mask.data(), noise.data()));
    }
    // Address::encrypt_sk, address.rs:86-109: the digits are created on the device
    void encrypt_address(Address& a, uint32_t value, const Secret& sk, Source& source_xa, Source& source_xe) {
        const size_t n = params.n(), size = (params.p.k_ggsw_addr + params.basek() - 1) / params.basek();
        const size_t glwes = (size_t)fheram_n_digits(ctx_) * ((params.k_glwe_ct() + params.basek() - 1) / params.basek()) * 2;   // dnum_ct rows x 2
        std::vector<int64_t> mask(glwes * size * n), noise(glwes * n);
        source_xa.uniform_limbs(mask.data(), mask.size());
        source_xe.gaussian(noise.data(), noise.size(), noise_scale(params.p.k_ggsw_addr, params.basek()));
        if (a.h_) { fheram_address_destroy(a.h_); a.h_ = nullptr; }
        chk(fheram_address_encrypt_sk(ctx_, sk.h_, value, mask.data(), noise.data(), &a.h_));
        a.owner_ = this;
    }
    // poulpy-schemes' FheUintPrepared<u32> as the path needs it (conversion.rs:68-82): one GGSW per bit, on this device
    class FheUintPrepared {
    public:
        // FheUintPrepared::encrypt_sk (conversion.rs:160-168) with the caller's samplers
        FheUintPrepared(Ram& ram, uint32_t value, const Secret& sk, Source& source_xa, Source& source_xe, int n_bits = 32) {
            const size_t n = ram.params.n(), b = ram.params.basek(), k = ram.params.p.k_evk_ggsw_inv;
            const size_t size = (k + b - 1) / b, dnum = (ram.params.p.k_ggsw_addr + b - 1) / b, glwes = (size_t)n_bits * dnum * 2;
            std::vector<int64_t> mask(glwes * size * n), noise(glwes * n);
            source_xa.uniform_limbs(mask.data(), mask.size());
            source_xe.gaussian(noise.data(), noise.size(), noise_scale((uint32_t)k, (uint32_t)b));
            ram.chk(fheram_fheuint_encrypt_sk(ram.ctx_, sk.h_, value, n_bits, mask.data(), noise.data(), &h_));
        }
        // from host ciphertexts: [n_bits][fheram_fheuint_ggsw_len]
        FheUintPrepared(Ram& ram, const std::vector<int64_t>& bits, int n_bits) { ram.chk(fheram_fheuint_create(ram.ctx_, bits.data(), n_bits, &h_)); }
        ~FheUintPrepared() { if (h_) fheram_fheuint_destroy(h_); }
        FheUintPrepared(const FheUintPrepared&) = delete;
    private:
        friend class Ram;
        fheram_fheuint* h_ = nullptr;
    };
    // Address::set_from_fheuint, conversion.rs:68-82 (sign = false: the convention of Address::encrypt_sk, i.e. an
    // address Ram::read accepts; true: what the reference's test decrypts to)
    void set_from_fheuint(Address& a, const FheUintPrepared& fheuint, bool sign = false) {
        if (a.h_) { fheram_address_destroy(a.h_); a.h_ = nullptr; }
        chk(fheram_address_set_from_fheuint(ctx_, fheuint.h_, sign ? 1 : 0, &a.h_));
        a.owner_ = this;
    }
    // EvaluationKeys::encrypt_sk + EvaluationKeysPrepared::prepare, keys.rs:135-180,57-71: generated and
    // prepared on this context; `keys` is marked as the set in use (its std forms stay empty).
    void encrypt_keys(EvaluationKeysPrepared& keys, const Secret& sk, Source& source_xa, Source& source_xe) {
        const size_t n = params.n(), b = params.basek();
        const size_t s4 = (params.p.k_evk_trace + b - 1) / b, s5 = (params.p.k_evk_ggsw_inv + b - 1) / b;
        const size_t dnum_ct = (params.k_glwe_ct() + b - 1) / b, dnum_ggsw = (params.p.k_ggsw_addr + b - 1) / b;   // parameters.rs:273-279
        const size_t n4 = params.p.log_n * dnum_ct, n5 = 2 * dnum_ggsw;
        std::vector<int64_t> mask((n4 * s4 + n5 * s5) * n), noise((n4 + n5) * n);
        source_xa.uniform_limbs(mask.data(), n4 * s4 * n);
        source_xa.uniform_limbs(mask.data() + n4 * s4 * n, n5 * s5 * n);
        source_xe.gaussian(noise.data(), n4 * n, noise_scale(params.p.k_evk_trace, params.basek()));
        source_xe.gaussian(noise.data() + n4 * n, n5 * n, noise_scale(params.p.k_evk_ggsw_inv, params.basek()));
        chk(fheram_keys_encrypt_sk(ctx_, sk.h_, mask.data(), noise.data(), nullptr));
        keys_ = &keys;
    }
    // GLWE::decrypt (examples/fhe-ram.rs:217-222): normalised plaintext limbs [n][size][N]
    std::vector<int64_t> decrypt(const std::vector<Glwe>& cts, const Secret& sk) {
        std::vector<int64_t> flat, pt(cts.size() * glwe_len() / 2);
        for (auto& g : cts) flat.insert(flat.end(), g.begin(), g.end());
        chk(fheram_glwe_decrypt(ctx_, sk.h_, (int)cts.size(), (int)(glwe_len() / (2 * params.n())), flat.data(), pt.data()));
        return pt;
    }

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

// Ram (ram.rs:25-29) over several GPUs of one node behind ONE handle: the native row-sharded path (fheram_group_*) for a
// single-process host.  Same three calls, one per op; `devices` a power of two of HIP device indices.
class GroupRam {
public:
    Parameters params;
    GroupRam(const Parameters& prm, const std::vector<int>& devices) : params(prm) {
        int rc = fheram_group_create(&params.p, devices.data(), (int)devices.size(), &grp_);
        if (rc != FHERAM_OK) throw Error(rc, fheram_group_last_error(nullptr));
    }
    GroupRam(const GroupRam&) = delete;
    ~GroupRam() {
        for (auto& kv : addr_) fheram_group_address_destroy(kv.second);
        if (grp_) fheram_group_destroy(grp_);
    }
    int size() const { return fheram_group_size(grp_); }
    size_t glwe_len() const { return fheram_glwe_len(fheram_group_ctx(grp_, 0)); }
    // the whole RAM [word_size][rows][GLWE] (Ram::encrypt_sk output, ram.rs:129-167); scattered by shard
    void load_encrypted(const std::vector<int64_t>& rows) { chk(fheram_group_ram_upload(grp_, rows.data())); }
    std::vector<Glwe> read(Address& address, const EvaluationKeysPrepared& keys) {                   // ram.rs:172-191
        use(keys);
        std::vector<int64_t> out(params.word_size() * glwe_len());
        chk(fheram_group_read(grp_, dev(address), out.data()));
        return split(out);
    }
    std::vector<Glwe> read_prepare_write(Address& address, const EvaluationKeysPrepared& keys) {     // ram.rs:196-222
        use(keys);
        std::vector<int64_t> out(params.word_size() * glwe_len());
        chk(fheram_group_read_prepare_write(grp_, dev(address), out.data()));
        return split(out);
    }
    void write(const std::vector<Glwe>& w, Address& address, const EvaluationKeysPrepared& keys) {   // ram.rs:226-294
        use(keys);
        std::vector<int64_t> flat;
        for (auto& g : w) flat.insert(flat.end(), g.begin(), g.end());
        chk(fheram_group_write(grp_, flat.data(), (int)w.size(), dev(address)));
    }
    bool state() const { return fheram_group_ram_state(grp_) != 0; }

private:
    fheram_group* grp_ = nullptr;
    const EvaluationKeysPrepared* keys_ = nullptr;
    std::vector<std::pair<const Address*, fheram_group_addr*>> addr_;   // replicated device copies, by host address object
    void chk(int rc) { if (rc != FHERAM_OK) throw Error(rc, fheram_group_last_error(grp_)); }
    void use(const EvaluationKeysPrepared& k) {
        if (keys_ == &k) return;
        std::vector<const int64_t*> ptr;
        for (auto& a : k.atk_glwe) ptr.push_back(a.data());
        chk(fheram_group_keys_load(grp_, k.gal_els.data(), (int)k.gal_els.size(), ptr.data(), k.atk_ggsw_inv.data(), k.atk_ggsw_inv_p,
                                   k.tsk_ggsw_inv.data()));
        keys_ = &k;
    }
    fheram_group_addr* dev(const Address& a) {
        for (auto& kv : addr_) if (kv.first == &a) return kv.second;
        std::vector<const int64_t*> ptr;
        for (auto& d : a.digits) ptr.push_back(d.data());
        fheram_group_addr* h = nullptr;
        chk(fheram_group_address_create(grp_, ptr.data(), (int)ptr.size(), &h));
        addr_.emplace_back(&a, h);
        return h;
    }
    std::vector<Glwe> split(const std::vector<int64_t>& flat) const {
        std::vector<Glwe> out;
        const size_t g = glwe_len();
        for (size_t i = 0; i < params.word_size(); i++) out.emplace_back(flat.begin() + i * g, flat.begin() + (i + 1) * g);
        return out;
    }
};

}  // namespace fheram
