// Build + link check of the C++ host mirror (no GPU needed: without a device Ram::Ram throws the
// C ABI's FHERAM_ERR_DEVICE, which is the behaviour the check asserts on a CPU-only box).
#include "fheram.hpp"
#include <cstdio>

int main() {
    fheram::Parameters p;
    if (p.max_addr() != (1u << 14) || p.word_size() != 4 || p.basek() != 17) return 2;   // parameters.rs:11-21
    try {
        fheram::Ram ram = fheram::Ram::new_from_ram_params(4, {3, 3, 3, 3}, 1 << 12);
        fheram::EvaluationKeysPrepared keys;
        fheram::Address addr;
        try { ram.read(addr, keys); return 3; }                      // no keys, empty RAM: must throw
        catch (const fheram::Error& e) { std::printf("refused as the reference would: %s\n", e.what()); }
    } catch (const fheram::Error& e) {
        if (e.code != FHERAM_ERR_DEVICE) return 4;
        std::printf("no GPU: %s\n", e.what());
    }
    return 0;
}
