// poulpy_kat.rs — known-answer-vector exporter for the FHE-RAM hot path (see README.md next to this file).
//
// NEVER COMPILED in the repository that ships it (no Rust toolchain there).  Drop into `examples/` of a
// phantomzone-org/fhe-ram checkout (snapshot 2026-02-13) next to a Poulpy 0.3.2-era checkout in ../poulpy and run
//     cargo run --release --example poulpy_kat -- <output directory>
// It follows examples/fhe-ram.rs:34-177 statement by statement (same seeds), and adds the Poulpy-level operations
// the path reaches (call sites: coordinate_prepared.rs:114,138,156; ram.rs:435,457).
//
// OPTIONAL one-line patch for whole-flow vectors (Ram's rows are private, ram.rs:298-303) — add to `impl SubRam`:
//     pub fn data(&self) -> &Vec<GLWE<Vec<u8>>> { &self.data }
// and build with `--features kat_rows` (or just delete the cfg below).
//
// ADAPT: the four `raw_*` helpers are the only places that touch Poulpy's layout internals (limb-major i64 buffers,
// SURVEY.md A.2).  At the pinned commit they are expected to be `obj.data().raw()` (poulpy_hal::layouts::ZnxView);
// if the accessor is named differently there, change these four lines only.

use std::{collections::HashMap, env, fs, io::Write, path::PathBuf};

use poulpy_backend::FFT64Ref as BackendImpl; // the portable backend: what the crate's unit tests use (parameters.rs:292)

use fhe_ram::{Address, EvaluationKeys, EvaluationKeysPrepared, Parameters, Ram};
use poulpy_core::{
    GGSWAutomorphism, GLWEAutomorphism, GLWEEncryptSk, GLWEExternalProduct, GLWEPacker, GLWETrace,
    layouts::{
        GGLWEToGGSWKeyPrepared, GGLWEToGGSWKeyPreparedFactory, GGSW, GGSWPrepared, GLWE, GLWEAutomorphismKeyPrepared,
        GLWEAutomorphismKeyPreparedFactory, GLWEInfos, GLWEPlaintext, GLWESecret, prepared::GLWESecretPrepared,
    },
};
use poulpy_hal::{
    api::{ScratchOwnedAlloc, ScratchOwnedBorrow},
    layouts::{Module, ScratchOwned, ZnxView},
    source::Source,
};
use rand_core::RngCore;

// ---- ADAPT (see header) --------------------------------------------------------------------------------------
fn raw_glwe(ct: &GLWE<Vec<u8>>) -> &[i64] { ct.data().raw() }
fn raw_ggsw(g: &GGSW<Vec<u8>>) -> &[i64] { g.data().raw() }
fn raw_atk(k: &poulpy_core::layouts::GLWEAutomorphismKey<Vec<u8>>) -> &[i64] { k.data().raw() }
fn raw_tsk(k: &poulpy_core::layouts::GGLWEToGGSWKey<Vec<u8>>) -> &[i64] { k.data().raw() }
// ---------------------------------------------------------------------------------------------------------------

struct Out { dir: PathBuf, manifest: Vec<String> }
impl Out {
    fn put(&mut self, name: &str, shape: &[usize], v: &[i64]) {
        assert_eq!(shape.iter().product::<usize>(), v.len(), "{name}: shape does not match the buffer");
        let mut f = fs::File::create(self.dir.join(format!("{name}.i64"))).unwrap();
        for x in v { f.write_all(&x.to_le_bytes()).unwrap(); }
        self.manifest.push(format!("  \"{name}\": {:?}", shape));
    }
    fn put_u8(&mut self, name: &str, v: &[u8]) {
        fs::write(self.dir.join(format!("{name}.u8")), v).unwrap();
        self.manifest.push(format!("  \"{name}\": [{}]", v.len()));
    }
}

fn main() {
    let dir = PathBuf::from(env::args().nth(1).expect("usage: poulpy_kat <output directory>"));
    fs::create_dir_all(&dir).unwrap();
    let mut out = Out { dir, manifest: Vec::new() };

    // ---- setup exactly as examples/fhe-ram.rs:37-95 -------------------------------------------------------------
    let mut source_xs = Source::new([0u8; 32]);
    let mut source_xa = Source::new([0u8; 32]);
    let mut source_xe = Source::new([0u8; 32]);
    let params: Parameters<BackendImpl> = Parameters::<BackendImpl>::new();          // MAX_ADDR = 2^14 (parameters.rs:21)
    let module: &Module<BackendImpl> = params.module();
    let n = module.n();
    let mut sk: GLWESecret<Vec<u8>> = GLWESecret::alloc_from_infos(&params.glwe_ct_infos());
    sk.fill_ternary_prob(0.5, &mut source_xs);
    let keys: EvaluationKeys<Vec<u8>> = EvaluationKeys::encrypt_sk(&params, &sk, &mut source_xa, &mut source_xe);
    let mut scratch: ScratchOwned<BackendImpl> = ScratchOwned::alloc(1 << 26);
    let mut sk_prep: GLWESecretPrepared<Vec<u8>, BackendImpl> = GLWESecretPrepared::alloc(module, sk.rank());
    sk_prep.prepare(module, &sk);
    let mut keys_prepared: EvaluationKeysPrepared<Vec<u8>, BackendImpl> = EvaluationKeysPrepared::alloc(&params);
    keys_prepared.prepare(module, &keys, scratch.borrow());
    // EvaluationKeysPrepared's fields are pub(crate) (keys.rs:27-31): for the Poulpy-level calls below the std-form keys
    // are prepared once more here, with the very calls keys.rs:34-71 makes
    let gal_els: Vec<i64> = GLWE::trace_galois_elements(module);
    let mut atk_prep: HashMap<i64, GLWEAutomorphismKeyPrepared<Vec<u8>, BackendImpl>> = HashMap::new();
    for g in gal_els.iter() {
        let mut p = GLWEAutomorphismKeyPrepared::alloc_from_infos(module, &params.evk_glwe_infos());
        p.prepare(module, keys.atk_glwe().get(g).unwrap(), scratch.borrow());
        atk_prep.insert(*g, p);
    }
    let mut atk_inv_prep = GLWEAutomorphismKeyPrepared::alloc_from_infos(module, &params.evk_ggsw_infos());
    atk_inv_prep.prepare(module, keys.atk_ggsw_inv(), scratch.borrow());
    let mut tsk_prep = GGLWEToGGSWKeyPrepared::alloc_from_infos(module, &params.evk_ggsw_infos());
    tsk_prep.prepare(module, keys.tsk_ggsw_inv(), scratch.borrow());

    out.put("sk", &[n], sk.data().raw());                                              // ADAPT: ScalarZnx of {-1,0,1}
    out.put("gal_els", &[gal_els.len()], &gal_els);
    for (i, g) in gal_els.iter().enumerate() {
        let k = keys.atk_glwe().get(g).unwrap();
        out.put(&format!("atk_{i}"), &[raw_atk(k).len()], raw_atk(k));
    }
    out.put("atk_inv", &[raw_atk(keys.atk_ggsw_inv()).len()], raw_atk(keys.atk_ggsw_inv()));
    out.put("tsk", &[raw_tsk(keys.tsk_ggsw_inv()).len()], raw_tsk(keys.tsk_ggsw_inv()));

    let mut source = Source::new([5u8; 32]);
    let ws = params.word_size();
    let mut data: Vec<u8> = vec![0u8; params.max_addr() * ws];
    source.fill_bytes(data.as_mut_slice());
    out.put_u8("data", &data);
    let mut ram: Ram<BackendImpl> = Ram::new();
    ram.encrypt_sk(&data, &sk, &mut source_xa, &mut source_xe);
    let mut addr: Address<Vec<u8>> = Address::alloc_from_params(&params);
    let idx: u32 = source.next_u32() % params.max_addr() as u32;
    addr.encrypt_sk(&params, idx, &sk, &mut source_xa, &mut source_xe, scratch.borrow());
    out.put("idx", &[1], &[idx as i64]);
    let mut digits: Vec<i64> = Vec::new();
    let mut n_digits = 0usize;
    for c in addr.coordinates.iter() { for g in c.value.iter() { digits.extend_from_slice(raw_ggsw(g)); n_digits += 1; } }
    out.put("addr", &[n_digits, digits.len() / n_digits], &digits);

    // ---- Poulpy-level operations on ciphertexts this program makes itself ------------------------------------------
    let glwe_infos = params.glwe_ct_infos();
    let fresh = |value: i64, seed: u8| -> GLWE<Vec<u8>> {                               // encrypt_glwe, examples/fhe-ram.rs:179-210
        let mut ct: GLWE<Vec<u8>> = GLWE::alloc_from_infos(&glwe_infos);
        let mut pt: GLWEPlaintext<Vec<u8>> = GLWEPlaintext::alloc_from_infos(&params.glwe_pt_infos());
        pt.encode_coeff_i64(value, params.glwe_pt_infos().k(), 0);
        let mut sc: ScratchOwned<BackendImpl> = ScratchOwned::alloc(GLWE::encrypt_sk_tmp_bytes(module, &glwe_infos));
        ct.encrypt_sk(module, &pt, &sk_prep, &mut Source::new([seed; 32]), &mut Source::new([seed + 100; 32]), sc.borrow());
        ct
    };
    // (1) external product with address digit 0 (coordinate_prepared.rs:156)
    let a = fresh(3, 11);
    let digit0: &GGSW<Vec<u8>> = &addr.coordinates[0].value[0];
    let mut digit0_prep: GGSWPrepared<Vec<u8>, BackendImpl> = GGSWPrepared::alloc_from_infos(module, digit0);
    digit0_prep.prepare(module, digit0, scratch.borrow());
    let mut res: GLWE<Vec<u8>> = GLWE::alloc_from_infos(&glwe_infos);
    module.glwe_external_product(&mut res, &a, &digit0_prep, scratch.borrow());
    out.put("ep_a", &[raw_glwe(&a).len()], raw_glwe(&a));
    out.put("ep_ggsw", &[raw_ggsw(digit0).len()], raw_ggsw(digit0));
    out.put("ep_res", &[raw_glwe(&res).len()], raw_glwe(&res));
    // (2) automorphism with two trace keys
    for g in [-1i64, 5] {
        let x = fresh(2, 12);
        let mut y: GLWE<Vec<u8>> = GLWE::alloc_from_infos(&glwe_infos);
        y.automorphism(module, &x, atk_prep.get(&g).unwrap(), scratch.borrow());
        out.put(&format!("auto_{}_in", if g < 0 { "m1".to_string() } else { g.to_string() }), &[raw_glwe(&x).len()], raw_glwe(&x));
        out.put(&format!("auto_{}_out", if g < 0 { "m1".to_string() } else { g.to_string() }), &[raw_glwe(&y).len()], raw_glwe(&y));
    }
    // (3) trace (ram.rs:457)
    let x = fresh(1, 13);
    let mut t: GLWE<Vec<u8>> = GLWE::alloc_from_infos(&glwe_infos);
    t.trace(module, 0, module.log_n(), &x, &atk_prep, scratch.borrow());
    out.put("trace_in", &[raw_glwe(&x).len()], raw_glwe(&x));
    out.put("trace_out", &[raw_glwe(&t).len()], raw_glwe(&t));
    // (4) packer over 4 ciphertexts, fed as SubRam::read does (ram.rs:425-448): j = 0..N, leaf reverse_bits_msb(j)
    let leaves: Vec<GLWE<Vec<u8>>> = (0..4).map(|i| fresh(i as i64 - 2, 20 + i as u8)).collect();
    let mut packer = GLWEPacker::alloc(&glwe_infos, 0);
    for j in 0..n {
        let j_rev = fhe_ram::reverse_bits_msb(j, module.log_n() as u32);
        if j_rev < leaves.len() { packer.add(module, Some(&leaves[j_rev]), &atk_prep, scratch.borrow()); }
        else { packer.add(module, None::<&GLWE<Vec<u8>>>, &atk_prep, scratch.borrow()); }
    }
    let mut packed: GLWE<Vec<u8>> = GLWE::alloc_from_infos(&glwe_infos);
    packer.flush(module, &mut packed);
    let mut flat: Vec<i64> = Vec::new();
    for l in leaves.iter() { flat.extend_from_slice(raw_glwe(l)); }
    out.put("pack_in", &[4, flat.len() / 4], &flat);
    out.put("pack_out", &[raw_glwe(&packed).len()], raw_glwe(&packed));
    // (5) GGSW inversion (coordinate_prepared.rs:138)
    let mut inv: GGSW<Vec<u8>> = GGSW::alloc_from_infos(digit0);
    inv.automorphism(module, digit0, &atk_inv_prep, &tsk_prep, scratch.borrow());
    out.put("ggsw_inv_in", &[raw_ggsw(digit0).len()], raw_ggsw(digit0));
    out.put("ggsw_inv_out", &[raw_ggsw(&inv).len()], raw_ggsw(&inv));

    // (6) N4: Address::set_from_fheuint (conversion.rs:68-82) — needs poulpy-schemes (Cargo feature `kat_fheuint`).  The evaluator's
    //     fheram_fheuint_* entry points are CONTRACT-compatible only: their FheUintPrepared is an invented layout (one 5-limb GGSW per
    //     bit, LSB first, k = K_EVK_GGSW_INV, dnum 4: include/fheram.h) and their digits come from CMux chains, not from
    //     scalar_to_ggsw_blind_rotation.  These vectors show how far that is from the real thing: (a) the raw buffer and layout
    //     infos of a real FheUintPrepared<u32> (is it one GGSW per bit at all? which k / dnum / bit order?), (b) ONE address digit
    //     derived by the reference from it (does the digit the evaluator derives for the same integer decrypt to the same
    //     X^{...}? — bit parity cannot be expected: the two algorithms accumulate different noise).
    #[cfg(feature = "kat_fheuint")]
    {
        use poulpy_schemes::tfhe::bdd_arithmetic::FheUintPrepared;
        let k_value: u32 = 0x1234_5678 % params.max_addr() as u32;
        let ggsw_k_infos = params.ggsw_infos();                                              // ADAPT: the layout conversion.rs' test gives its fheuint (:70-83)
        let mut fheuint: FheUintPrepared<Vec<u8>, u32, BackendImpl> = FheUintPrepared::alloc_from_infos(module, &ggsw_k_infos);
        fheuint.encrypt_sk(module, k_value, &sk_prep, &mut Source::new([61u8; 32]), &mut Source::new([62u8; 32]), scratch.borrow());   // conversion.rs:160-168
        out.put("fheuint_value", &[1], &[k_value as i64]);
        out.put("fheuint_raw", &[fheuint.raw_len()], fheuint.raw());                          // ADAPT: every limb of every block, in memory order
        out.put("fheuint_infos", &[6], &[fheuint.n() as i64, fheuint.base2k() as i64, fheuint.k() as i64, fheuint.dnum() as i64,
                                         fheuint.rank() as i64, fheuint.blocks() as i64]);    // ADAPT: whatever the type exposes
        let mut derived: Address<Vec<u8>> = Address::alloc_from_params(&params);
        derived.set_from_fheuint(module, &fheuint, scratch.borrow());                         // conversion.rs:68-82
        let d0: &GGSW<Vec<u8>> = &derived.coordinates[0].value[0];
        out.put("fheuint_digit0", &[raw_ggsw(d0).len()], raw_ggsw(d0));
    }

    // ---- whole flow (needs the SubRam::data accessor patch, see header) ------------------------------------------------
    #[cfg(feature = "kat_rows")]
    {
        let mut rows: Vec<i64> = Vec::new();
        for s in ram.subrams.iter() { for ct in s.data().iter() { rows.extend_from_slice(raw_glwe(ct)); } }
        out.put("rows", &[ws, ram.subrams[0].data().len(), rows.len() / ws / ram.subrams[0].data().len()], &rows);
        let dump = |name: &str, cts: &Vec<GLWE<Vec<u8>>>, out: &mut Out| {
            let mut v: Vec<i64> = Vec::new();
            for c in cts.iter() { v.extend_from_slice(raw_glwe(c)); }
            out.put(name, &[cts.len(), v.len() / cts.len()], &v);
        };
        let r = ram.read(&addr, &keys_prepared);
        dump("read", &r, &mut out);
        let r = ram.read_prepare_write(&addr, &keys_prepared);
        dump("rpw", &r, &mut out);
        let mut value: Vec<u8> = vec![0u8; ws];
        source.fill_bytes(value.as_mut_slice());
        let ct_w: Vec<GLWE<Vec<u8>>> = value.iter().enumerate().map(|(i, w)| fresh(*w as i64, 40 + i as u8)).collect();
        dump("w", &ct_w, &mut out);
        ram.write(&ct_w, &addr, &keys_prepared);
        let r = ram.read(&addr, &keys_prepared);
        dump("readback", &r, &mut out);
    }
    let _ = &mut ram;

    let body = out.manifest.join(",\n");
    fs::write(out.dir.join("manifest.json"), format!(
        "{{\n \"poulpy\": \"FILL IN: git rev-parse HEAD of ../poulpy\",\n \"backend\": \"FFT64Ref\",\n \"n\": {n}, \"base2k\": 17, \"max_addr\": {}, \"word_size\": {ws},\n \"files\": {{\n{body}\n }}\n}}\n",
        params.max_addr())).unwrap();
    println!("wrote {} vectors", out.manifest.len());
}
