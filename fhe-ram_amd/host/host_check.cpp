// Build + link check of the C++ host mirror (no GPU needed: without a device Ram::Ram throws the
// C ABI's FHERAM_ERR_DEVICE, which is the behaviour the check asserts on a CPU-only box).
#include "fheram.hpp"
#include <cstdio>

// a sampler with the interface the encrypt_sk methods take (a real host plugs in its own CSPRNG)
struct ZeroSource : fheram::Source {
    void uniform_limbs(int64_t* out, size_t count) override { for (size_t i = 0; i < count; i++) out[i] = 0; }
    void gaussian(int64_t* out, size_t count, double) override { for (size_t i = 0; i < count; i++) out[i] = 0; }
};

int main() {
    fheram::Parameters p;
    if (p.max_addr() != (1u << 14) || p.word_size() != 4 || p.basek() != 17) return 2;   // parameters.rs:11-21
    fheram::Parameters rd = fheram::Parameters::readme();                                    // README.md:20-33
    if (rd.max_addr() != (1u << 18) || rd.k_glwe_pt() != 9 || rd.p.k_evk_trace != 5 * 17) return 2;
    {   // execution switches: library defaults (pure host code: runs without a GPU)
        const fheram_config c = fheram::Ram::default_config();
        if (!c.fuse || !c.tail || c.chain_y != 3 || c.reserved != 0) return 6;
    }
    try {
        fheram::Ram ram = fheram::Ram::new_from_ram_params(4, {3, 3, 3, 3}, 1 << 12);
        if (ram.config().fuse != fheram::Ram::default_config().fuse) return 6;
        fheram::EvaluationKeysPrepared keys;
        fheram::Address addr;
        try { ram.read(addr, keys); return 3; }                      // no keys, empty RAM: must throw
        catch (const fheram::Error& e) { std::printf("refused as the reference would: %s\n", e.what()); }
        // setup side on the device: secret, keys, RAM and address, then one read and its decryption
        ZeroSource xa, xe;
        std::vector<int64_t> sk(p.n(), 0);
        sk[1] = 1; sk[5] = -1;
        fheram::Ram::Secret dsk(ram, sk);
        ram.encrypt_keys(keys, dsk, xa, xe);
        std::vector<uint8_t> data((size_t)(1 << 12) * 4);
        for (size_t i = 0; i < data.size(); i++) data[i] = (uint8_t)(i * 7 + 3);
        ram.encrypt_sk(data, dsk, xa, xe);
        const uint32_t idx = 1234;
        ram.encrypt_address(addr, idx, dsk, xa, xe);
        std::vector<int64_t> pt = ram.decrypt(ram.read(addr, keys), dsk);
        for (size_t i = 0; i < 4; i++) {                             // noiseless: limb 0 of coefficient 0 is the byte's low 3 bits << 14
            const int64_t want = (int64_t)((int8_t)(uint8_t)(data[idx * 4 + i] << 5) >> 5) * (1 << 14);
            const int64_t got = pt[i * 3 * p.n()];
            if (((got - want) & ((1 << 17) - 1)) != 0) { std::printf("word %zu: got %lld want %lld\n", i, (long long)got, (long long)want); return 5; }
        }
        std::printf("device-side setup + read + decrypt: ok\n");
        {   // the result in place (pinned host buffer the device wrote): the same limbs as the copy returned by read
            std::vector<fheram::Glwe> r = ram.read(addr, keys);
            const int64_t* v = ram.result_view();
            for (size_t i = 0; i < r.size(); i++)
                for (size_t k = 0; k < r[i].size(); k++)
                    if (v[i * r[i].size() + k] != r[i][k]) { std::printf("result_view differs\n"); return 7; }
            std::printf("result in place == result copied: ok\n");
            const auto ts = ram.tail_stats();
            if (ts.launches == 0 || ts.redone != 0) { std::printf("tail stats: %llu launches, %llu redone\n", (unsigned long long)ts.launches, (unsigned long long)ts.redone); return 10; }
        }
        // Address::set_from_fheuint (conversion.rs:68-82): the same word through an address derived from an encrypted integer
        fheram::Ram::FheUintPrepared fu(ram, idx, dsk, xa, xe, 12);
        fheram::Address derived;
        ram.set_from_fheuint(derived, fu);
        std::vector<int64_t> pt2 = ram.decrypt(ram.read(derived, keys), dsk);
        for (size_t i = 0; i < 4; i++)
            if (((pt2[i * 3 * p.n()] - pt[i * 3 * p.n()]) & ((1 << 17) - 1)) != 0) { std::printf("derived address: word %zu differs\n", i); return 6; }
        std::printf("address derived from an encrypted integer reads the same word: ok\n");
        // one handle over "two GPUs" (the same device twice: rehearsal): a group must refuse an operation before its keys
        // and RAM are loaded exactly as the single context does
        fheram::GroupRam grp(fheram::Parameters(4, {3, 3, 3, 3}, 1 << 13), {0, 0});   // two rows per sub-RAM: one per shard
        if (grp.size() != 2) return 8;
        try { fheram::Address a2; grp.read(a2, fheram::EvaluationKeysPrepared()); return 9; }
        catch (const fheram::Error& e) { std::printf("group refused as the reference would: %s\n", e.what()); }
    } catch (const fheram::Error& e) {
        if (e.code != FHERAM_ERR_DEVICE) { std::printf("unexpected error %d: %s\n", e.code, e.what()); return 4; }
        std::printf("no GPU: %s\n", e.what());
    }
    return 0;
}
