// Private to fheram.hip: kernel launchers (choice between the fused / column-split / limb-parallel
// decompositions), dependent chains, the packing tree, coordinate preparation.
#pragma once
#include "ctx.hpp"

namespace {

// ---- kernel launchers ---------------------------------------------------------------------
constexpr int LIMB_SPLIT_MAX = 64;   // ciphertexts per launch the limb-parallel path is used for (at most)
constexpr int EW_SLICES = 8;   // workgroups per ciphertext of the elementwise kernels (blockIdx.z)
// One workgroup per ciphertext does the least work (no repeated forward transforms); splitting by
// output column doubles the number of workgroups, which pays while the batch cannot fill the CUs.
int pick_nco(const fheram_ctx* c, int gx, int gy) {
    if (c->nco != 0) return c->nco;
    return ((long)gx * gy * 2 <= c->cus) ? 1 : 2;
}
// gal != 0: automorphism key of Galois element gal, prepared as FFT(phi_gal(K)) (see k_prepare)
void launch_prepare(fheram_ctx* c, const int32_t* in, double* out, int npoly, int64_t gal = 0) {
    ProfScope ps(c, "prepare", npoly);
    const int ginv = gal == 0 ? 0 : galois_inv_mod(galois_mod(gal));
    hipLaunchKernelGGL(k_prepare, dim3((npoly + 1) / 2), dim3(T), LDS_PREPARE_BYTES, c->cur, in, out, c->d_tw, c->ninv, ginv, npoly);
}
// res = a (x) ggsw over a (gx, gy) grid of ciphertexts; res must not alias a
// Limb-parallel path: 2*SK workgroups per ciphertext + a normalisation pass, chosen while even the
// column split leaves most CUs idle.
double* big_of(const fheram_ctx* c) { return c->cur == c->stream2 ? c->d_big2 : c->d_big; }
bool use_limb_split(const fheram_ctx* c, int gx, int gy, int sk) {
    return c->limb_split && (long)gx * gy <= LIMB_SPLIT_MAX && (long)gx * gy * 2 * sk <= c->cus;
}
// Fine limb split (k_keyswitch_fine / k_ext_product_fine): wgs workgroups per ciphertext, one forward and one
// inverse transform each, while the whole launch still fits the chip in one wave of workgroups.
bool use_fine_split(const fheram_ctx* c, int gx, int gy, int wgs) {
    return c->limb_split && c->fine_split && (long)gx * gy * wgs <= c->cus && (long)gx * gy * wgs * N * 8 <= (long)LIMB_SPLIT_MAX * BIG_STRIDE * 8;
}
void launch_ep(fheram_ctx* c, GlweRef a, GlweRef res, const double* ggsw, int gx, int gy) {
    if (gx <= 0 || gy <= 0) return;
    ProfScope ps(c, "ext_product", (uint64_t)gx * gy);
    if (use_fine_split(c, gx, gy, 2 * 4 * 2 * 3)) {
        hipLaunchKernelGGL((k_ext_product_fine<3, 4>), dim3(gx, gy, 2 * 4 * 2 * 3), dim3(T), LDS_BYTES, c->cur, a, ggsw, c->d_tw, big_of(c));
        hipLaunchKernelGGL((k_ext_product_fine_norm<3, 4>), dim3(gx, gy, 2 * (N / 256)), dim3(256), 0, c->cur, res, big_of(c));
        return;
    }
    if (use_limb_split(c, gx, gy, 4)) {
        hipLaunchKernelGGL((k_ext_product<3, 4, 1, 1>), dim3(gx, gy, 8), dim3(T), LDS_BYTES, c->cur, a, res, ggsw, c->d_tw, big_of(c));
        hipLaunchKernelGGL((k_ext_product<3, 4, 1, 2>), dim3(gx, gy, 2), dim3(T), 0, c->cur, a, res, ggsw, c->d_tw, big_of(c));
        return;
    }
    if (pick_nco(c, gx, gy) == 1) hipLaunchKernelGGL((k_ext_product<3, 4, 1>), dim3(gx, gy, 2), dim3(T), LDS_BYTES, c->cur, a, res, ggsw, c->d_tw, big_of(c));
    else {
        ProfScope pf(c, "ext_product_fused", (uint64_t)gx * gy);   // the dominant launch shape of its class, timed on its own
        hipLaunchKernelGGL((k_ext_product<3, 4, 2>), dim3(gx, gy, 1), dim3(T), LDS_BYTES, c->cur, a, res, ggsw, c->d_tw, big_of(c));
    }
}
template <int MODE, int SX, int SK, int SO>
void launch_ks(fheram_ctx* c, const KsArgs& ka, int gx, int gy) {
    if (gx <= 0 || gy <= 0) return;
    ProfScope ps(c, "keyswitch", (uint64_t)gx * gy);
    if (use_fine_split(c, gx, gy, 2 * SK * SX)) {
        KsArgs kb = ka;
        kb.big = big_of(c);
        hipLaunchKernelGGL((k_keyswitch_fine<MODE, SX, SK>), dim3(gx, gy, 2 * SK * SX), dim3(T), LDS_BYTES, c->cur, kb);
        hipLaunchKernelGGL((k_keyswitch_norm<MODE, SX, SK, SO, SX>), dim3(gx, gy, 2 * (N / 256)), dim3(256), 0, c->cur, kb);
        return;
    }
    if (use_limb_split(c, gx, gy, SK)) {
        KsArgs kb = ka;
        kb.big = big_of(c);
        hipLaunchKernelGGL((k_keyswitch<MODE, SX, SK, SO, 1, 1>), dim3(gx, gy, 2 * SK), dim3(T), LDS_BYTES, c->cur, kb);
        hipLaunchKernelGGL((k_keyswitch_norm<MODE, SX, SK, SO>), dim3(gx, gy, 2 * (N / 256)), dim3(256), 0, c->cur, kb);
        return;
    }
    // The two-column fused form exists for the 3-limb automorphism family only: the 4-limb GGSW-inversion steps and
    // the packer combine always run split by column (their fused forms spill registers; at 2^21, where pair levels
    // have up to 1024 pairs, the split combine is 3 % faster per read than the spilling fused one was).
    constexpr bool FUSABLE = (SX == 3) && (MODE != KS_PAIR);
    if constexpr (MODE == KS_PAIR && SX == 3 && SO == 3) {
        if (c->pair_z) {   // the column-split combine in closed form (k_pair_z)
            hipLaunchKernelGGL((k_pair_z<SK>), dim3(gx, gy, 2), dim3(T), LDS_BYTES, c->cur, ka);
            return;
        }
    }
    if (!FUSABLE || pick_nco(c, gx, gy) == 1) {
        hipLaunchKernelGGL((k_keyswitch<MODE, SX, SK, SO, 1>), dim3(gx, gy, 2), dim3(T), LDS_BYTES, c->cur, ka);
    } else if constexpr (FUSABLE) {
        ProfScope pf(c, "keyswitch_fused", (uint64_t)gx * gy);     // the dominant launch shape (one workgroup per ciphertext)
        hipLaunchKernelGGL((k_keyswitch<MODE, SX, SK, SO, 2>), dim3(gx, gy, 1), dim3(T), LDS_BYTES, c->cur, ka);
    }
}
void launch_copy(fheram_ctx* c, GlweRef src, GlweRef dst, int gx, int gy) {
    if (gx <= 0 || gy <= 0) return;
    ProfScope ps(c, "elementwise", (uint64_t)gx * gy);
    hipLaunchKernelGGL((k_copy<3>), dim3(gx, gy, EW_SLICES), dim3(256), 0, c->cur, src, dst);
}
KsArgs ks_args(fheram_ctx* c, GlweRef a, GlweRef b, GlweRef out, const double* key, int64_t gal, int t = 0, int rot_mul = 0, int rot_base = 0) {
    KsArgs ka;
    ka.a = a; ka.b = b; ka.out = out; ka.key = key; ka.tw = c->d_tw;
    ka.g = galois_mod(gal); ka.ginv = galois_inv_mod(ka.g); ka.t = t; ka.rot_mul = rot_mul; ka.rot_base = rot_base; ka.big = c->d_big;
    return ka;
}
const double* trace_key(fheram_ctx* c, int i) { return c->d_atk + (size_t)i * c->atk; }
// the automorphism family on RAM ciphertexts (3 limbs) with a trace key of the context's size (4 or 5 limbs)
template <int MODE>
void launch_ks_tr(fheram_ctx* c, const KsArgs& ka, int gx, int gy) {
    if (c->s_evk == 5) launch_ks<MODE, 3, 5, 3>(c, ka, gx, gy);
    else launch_ks<MODE, 3, 4, 3>(c, ka, gx, gy);
}
bool same(const GlweRef& a, const GlweRef& b) { return a.p == b.p; }

// Runs n dependent out-of-place steps src -> ... -> dst, alternating between dst and tmp so that
// the last step lands in dst.  step(i, in, out) launches step i.  dst may be src.
template <typename F>
void run_chain(fheram_ctx* c, int n, GlweRef src, GlweRef dst, GlweRef tmp, int gx, int gy, F&& step) {
    if (n <= 0) { if (!same(src, dst)) launch_copy(c, src, dst, gx, gy); return; }
    if (same(src, dst) && (n % 2 == 1)) {   // the first step would write what it reads: finish in tmp, copy back
        run_chain(c, n, src, tmp, dst, gx, gy, step);
        launch_copy(c, tmp, dst, gx, gy);
        return;
    }
    GlweRef cur = src;
    for (int i = 0; i < n; i++) {
        GlweRef out = ((n - 1 - i) % 2 == 0) ? dst : tmp;
        step(i, cur, out);
        cur = out;
    }
}
// Dependent chains on 9..64 ciphertexts (MAX_ADDR = 2^13 .. 2^16: the alone packer levels, the products of coordinate 0,
// write_mid_step's traces): ONE launch with in-kernel hand-offs (k_chain_mid), followed by the fused chain launch that redoes
// the ciphertexts whose group gave up (normally none).  Returns the split (0: not applicable; 1: <3,2>, 2: <1,1>, 3: <1,2>).
int use_mid(const fheram_ctx* c, int n, int gx, int gy, int sk, bool ep = false) {
    const long batch = (long)gx * gy;
    if (!(c->mid && c->limb_split && n >= 2 && n <= CHAIN_MAX && batch > TAIL_GROUPS &&
          c->cus >= TAIL_GROUPS * 32 &&      // the whole chip (8 XCDs x 32 CUs)
          !(c->use_graph && !c->profile)))   // a captured launch would replay its generation number
        return 0;
    if (batch <= 16) return 1;
    if (ep || c->mid < 2) return 0;          // the coarser splits: trace chains only
    const int m1 = 2 * sk, m2 = sk;          // members of <1,1>, <1,2>
    if (batch <= 8 * (32 / m1)) return 2;
    if (batch <= 8 * (32 / m2)) return 3;
    return 0;
}
template <bool EP>
void fill_mid(fheram_ctx* c, MidArgs& ma, GlweRef src, GlweRef dst, int n, int gx, int gy) {
    const int side = c->cur == c->stream2 ? 1 : 0;
    ma.src = src; ma.dst = dst; ma.tw = c->d_tw; ma.big = c->d_mid_big[side]; ma.sync = c->d_mid_sync[side]; ma.y = c->d_mid_y[side];
    if (++c->mid_seq == 0) ++c->mid_seq;
    c->mid_launches++;
    // A context that keeps losing its CUs to others stops asking — for a while.  The fallback launch mirrors the number of
    // ciphertexts it had to redo into a pinned host word (read without a synchronisation: a stale value only delays the
    // decision by a window).  A WINDOW is 64 launches; it is bad when more than a quarter of the ciphertexts launched in it
    // (counted, both streams) were redone.  One contended launch does not switch the path off, two bad windows in a row do
    // (fheram_mid_state); 256 ops later (path.hpp) the single-launch form is tried again.
    c->mid_window_cts += (uint64_t)(gx * gy);
    if (!c->mid_test && c->mid_launches - c->mid_launch_mark >= 64) {
        const unsigned fb = __atomic_load_n(c->h_mid_fb, __ATOMIC_RELAXED) + __atomic_load_n(c->h_mid_fb + 16, __ATOMIC_RELAXED);
        const unsigned redone = fb - c->mid_fb_mark;
        const bool bad = (uint64_t)redone * 4u > c->mid_window_cts;
        c->mid_bad_windows = bad ? c->mid_bad_windows + 1 : 0;
        if (c->mid && c->mid_bad_windows >= 2) { c->mid_saved = c->mid; c->mid = 0; c->mid_disabled_count++; }
        c->mid_fb_mark = fb;
        c->mid_launch_mark = c->mid_launches;
        c->mid_window_cts = 0;
    }
    ma.seq = c->mid_seq; ma.n = n; ma.n_ct = gx * gy; ma.gx = gx; ma.rot_mul = 0; ma.rot_base = 0;
    ma.give_up_at = c->mid_test ? n - 2 : -1;
}
// b: the buffers of the fallback chain (step i writes b[i & 1]; b[0] != src); the result lands in b[(n - 1) & 1], which may be src
template <int SK, int RS, int LPM>
void launch_k_mid_trace(fheram_ctx* c, const MidArgs& ma) {
    constexpr int members = RS * 2 * SK / LPM;
    hipLaunchKernelGGL((k_chain_mid<false, SK, RS, LPM>), dim3(8 * (32 / members) * members), dim3(T), LDS_BYTES, c->cur, ma);
}
void launch_mid_trace(fheram_ctx* c, GlweRef src, const GlweRef (&b)[2], int start, int n, int gx, int gy, int rot_mul, int rot_base, int split) {
    ProfScope ps(c, "keyswitch", (uint64_t)gx * gy, n);
    ProfScope pm(c, "keyswitch_mid_launch", (uint64_t)gx * gy * n, 1);
    MidArgs ma;
    fill_mid<false>(c, ma, src, b[(n - 1) & 1], n, gx, gy);
    ma.rot_mul = rot_mul; ma.rot_base = rot_base;
    KsChainArgs ca;
    ca.base = ks_args(c, src, src, b[0], trace_key(c, start), c->gal[start], 0, rot_mul, rot_base);
    ca.buf[0] = b[0]; ca.buf[1] = b[1]; ca.n = n;
    ca.done = ma.sync; ca.done_seq = ma.seq; ca.host_count = c->h_mid_fb + (c->cur == c->stream2 ? 16 : 0);
    for (int i = 0; i < n; i++) { ma.opnd[i] = ca.key[i] = trace_key(c, start + i); ma.ginv[i] = ca.ginv[i] = galois_inv_mod(galois_mod(c->gal[start + i])); }
    if (c->s_evk == 5) {
        if (split == 1) launch_k_mid_trace<5, 3, 2>(c, ma); else if (split == 2) launch_k_mid_trace<5, 1, 1>(c, ma); else launch_k_mid_trace<5, 1, 2>(c, ma);
        hipLaunchKernelGGL((k_keyswitch_chain<3, 5, 3>), dim3(gx, gy, 1), dim3(T), LDS_BYTES, c->cur, ca);
    } else {
        if (split == 1) launch_k_mid_trace<4, 3, 2>(c, ma); else if (split == 2) launch_k_mid_trace<4, 1, 1>(c, ma); else launch_k_mid_trace<4, 1, 2>(c, ma);
        hipLaunchKernelGGL((k_keyswitch_chain<3, 4, 3>), dim3(gx, gy, 1), dim3(T), LDS_BYTES, c->cur, ca);
    }
}
void launch_mid_ep(fheram_ctx* c, GlweRef src, const GlweRef (&b)[2], const double* prep, int d, int gx, int gy) {
    ProfScope ps(c, "ext_product", (uint64_t)gx * gy, d);
    ProfScope pm(c, "ext_product_mid_launch", (uint64_t)gx * gy * d, 1);
    MidArgs ma;
    fill_mid<true>(c, ma, src, b[(d - 1) & 1], d, gx, gy);
    EpChainArgs ca;
    ca.src = src; ca.buf[0] = b[0]; ca.buf[1] = b[1]; ca.tw = c->d_tw; ca.n = d;
    ca.done = ma.sync; ca.done_seq = ma.seq; ca.host_count = c->h_mid_fb + (c->cur == c->stream2 ? 16 : 0);
    for (int i = 0; i < d; i++) { ma.opnd[i] = ca.ggsw[i] = prep + (size_t)i * fheram_ctx::GGSW; ma.ginv[i] = 1; }
    hipLaunchKernelGGL((k_chain_mid<true, 4, 3, 2>), dim3(8 * 2 * 12), dim3(T), LDS_BYTES, c->cur, ma);
    hipLaunchKernelGGL((k_ext_product_chain<3, 4>), dim3(gx, gy, 1), dim3(T), LDS_BYTES, c->cur, ca);
}
// A dependent chain of n fused steps on the same ciphertexts runs as ONE launch when the batch is large enough
// for the fused decomposition (one workgroup per ciphertext): the workgroup ping-pongs between its own slots
// of two buffers.  bufs: step i writes b[i & 1]; b[0] must not be the source.
bool use_chain(const fheram_ctx* c, int n, int gx, int gy, int sk) {
    return c->chain && n >= 2 && n <= CHAIN_MAX && c->nco != 1 && !use_limb_split(c, gx, gy, sk) && pick_nco(c, gx, gy) == 2;
}
// picks (b0, b1) for a chain src -> dst with scratch tmp such that the last step lands in dst; false when the
// first step would have to write what it reads (src == dst and n odd)
bool chain_bufs(int n, GlweRef src, GlweRef dst, GlweRef tmp, GlweRef (&b)[2]) {
    if (n % 2 == 1) { if (same(src, dst)) return false; b[0] = dst; b[1] = tmp; }
    else { b[0] = tmp; b[1] = dst; }
    return !same(b[0], src);
}
void launch_ep_chain(fheram_ctx* c, GlweRef src, const GlweRef (&b)[2], const double* prep, int d, int gx, int gy) {
    ProfScope ps(c, "ext_product", (uint64_t)gx * gy, d);
    ProfScope pf(c, "ext_product_fused", (uint64_t)gx * gy, d);
    EpChainArgs ca;
    ca.src = src; ca.buf[0] = b[0]; ca.buf[1] = b[1]; ca.tw = c->d_tw; ca.n = d;
    for (int i = 0; i < d; i++) ca.ggsw[i] = prep + (size_t)i * fheram_ctx::GGSW;
    if (d >= 2) hipLaunchKernelGGL((k_ext_product_chain_r<4>), dim3(gx, gy, 1), dim3(T), LDS_BYTES, c->cur, ca);   // closed-form normalisation, products handed over in registers / LDS
    else hipLaunchKernelGGL((k_ext_product_chain<3, 4>), dim3(gx, gy, 1), dim3(T), LDS_BYTES, c->cur, ca);
}
// CoordinatePrepared::product / product_inplace (coordinate_prepared.rs:147-177): d external products.
void ep_chain(fheram_ctx* c, GlweRef src, GlweRef dst, GlweRef tmp, const double* prep, int d, int gx, int gy) {
    if (gx <= 0 || gy <= 0) return;
    if (use_mid(c, d, gx, gy, 4, true)) {
        GlweRef b[2];   // in place too (read_prepare_write): only the last step writes the destination
        if (chain_bufs(d, src, dst, tmp, b)) { launch_mid_ep(c, src, b, prep, d, gx, gy); return; }
    }
    if (use_chain(c, d, gx, gy, 4) && !use_fine_split(c, gx, gy, 2 * 4 * 2 * 3)) {
        GlweRef b[2];
        if (chain_bufs(d, src, dst, tmp, b)) { launch_ep_chain(c, src, b, prep, d, gx, gy); return; }
        if (chain_bufs(d, src, tmp, dst, b)) {   // src == dst, d odd: finish in tmp, copy back
            launch_ep_chain(c, src, b, prep, d, gx, gy);
            launch_copy(c, tmp, dst, gx, gy);
            return;
        }
    }
    run_chain(c, d, src, dst, tmp, gx, gy, [&](int i, GlweRef in, GlweRef out) { launch_ep(c, in, out, prep + (size_t)i * fheram_ctx::GGSW, gx, gy); });
}
void launch_trace_chain(fheram_ctx* c, GlweRef src, const GlweRef (&b)[2], int start, int n, int gx, int gy, int rot_mul, int rot_base) {
    ProfScope ps(c, "keyswitch", (uint64_t)gx * gy, n);
    ProfScope pf(c, "keyswitch_fused", (uint64_t)gx * gy, n);
    ProfScope pl(c, "keyswitch_chain_launch", (uint64_t)gx * gy * n, 1);   // the launch itself, as rocprofv3 sees it
    KsChainArgs ca;
    ca.base = ks_args(c, src, src, b[0], trace_key(c, start), c->gal[start], 0, rot_mul, rot_base);
    ca.buf[0] = b[0]; ca.buf[1] = b[1]; ca.n = n;
    for (int i = 0; i < n; i++) { ca.key[i] = trace_key(c, start + i); ca.ginv[i] = galois_inv_mod(galois_mod(c->gal[start + i])); }
    const int yf = n >= 2 ? c->chain_y : 0;   // intermediates handed over as Y = ceil(A/2) through LDS and registers (ks_trace_l); 0: int32 limbs (ks_run)
    if (c->s_evk == 5) {
        if (yf && c->wide) { c->wide_unsynced = true; hipLaunchKernelGGL((k_keyswitch_chain_w<3, 5, 3, 3>), dim3(gx, gy, 1), dim3(T), LDS_BYTES, c->cur, ca); }
        else if (yf) hipLaunchKernelGGL((k_keyswitch_chain<3, 5, 3, 3>), dim3(gx, gy, 1), dim3(T), LDS_BYTES, c->cur, ca);
        else hipLaunchKernelGGL((k_keyswitch_chain<3, 5, 3>), dim3(gx, gy, 1), dim3(T), LDS_BYTES, c->cur, ca);
    } else {
        if (yf && c->wide) { c->wide_unsynced = true; hipLaunchKernelGGL((k_keyswitch_chain_w<3, 4, 3, 3>), dim3(gx, gy, 1), dim3(T), LDS_BYTES, c->cur, ca); }
        else if (yf) hipLaunchKernelGGL((k_keyswitch_chain<3, 4, 3, 3>), dim3(gx, gy, 1), dim3(T), LDS_BYTES, c->cur, ca);
        else hipLaunchKernelGGL((k_keyswitch_chain<3, 4, 3>), dim3(gx, gy, 1), dim3(T), LDS_BYTES, c->cur, ca);
    }
}
// The two chains a row goes through back to back as ONE launch (k_read_chain / k_write_chain): both must be in the fused,
// one-workgroup-per-ciphertext regime, in the forms that hand over through LDS and registers.
bool use_row_fuse(const fheram_ctx* c, int d, int n_tr, int gx, int gy) {
    return c->fuse && c->chain_y == 3 && d >= 2 && d <= CHAIN_MAX && n_tr >= 2 && n_tr <= CHAIN_MAX &&
           use_chain(c, d, gx, gy, 4) && !use_mid(c, d, gx, gy, 4, true) && !use_fine_split(c, gx, gy, 2 * 4 * 2 * 3) &&
           use_chain(c, n_tr, gx, gy, c->s_evk) && !use_mid(c, n_tr, gx, gy, c->s_evk) && !use_fine_split(c, gx, gy, 2 * c->s_evk * 3) &&
           !(c->use_graph && !c->profile);
}
void fill_row_chain(fheram_ctx* c, RowChainArgs& ra, const double* prep, int d, int start, int n_tr) {
    ra.ep.tw = c->d_tw; ra.ep.n = d;
    for (int i = 0; i < d; i++) ra.ep.ggsw[i] = prep + (size_t)i * fheram_ctx::GGSW;
    ra.ks.n = n_tr;
    for (int i = 0; i < n_tr; i++) { ra.ks.key[i] = trace_key(c, start + i); ra.ks.ginv[i] = galois_inv_mod(galois_mod(c->gal[start + i])); }
}
// read / read_prepare_write: d products of `src` with the prepared digits, then trace steps 0 .. n_tr-1 (the alone packer levels);
// the result lands in dst; ep_store != nullptr: the products' result is also written there (in-place products of read_prepare_write)
void launch_read_chain(fheram_ctx* c, GlweRef src, const GlweRef* ep_store, GlweRef dst, const double* prep, int d, int n_tr, int gx, int gy) {
    ProfScope ps(c, "read_chain_launch", (uint64_t)gx * gy, 1);
    RowChainArgs ra;
    fill_row_chain(c, ra, prep, d, 0, n_tr);
    ra.ep.src = src;
    ra.ep.buf[0] = ra.ep.buf[1] = ep_store ? *ep_store : dst;     // only the last product stores, and only when asked to
    ra.store_ep = ep_store ? 1 : 0;
    ra.ks.base = ks_args(c, dst, dst, dst, trace_key(c, 0), c->gal[0]);
    ra.ks.buf[0] = ra.ks.buf[1] = dst;                            // only the last step stores
    ra.hi = dst; ra.trhi = dst;
    if (c->s_evk == 5) {
        if (c->wide) { c->wide_unsynced = true; hipLaunchKernelGGL((k_read_chain_w<5, 4>), dim3(gx, gy, 1), dim3(T), LDS_BYTES, c->cur, ra); }
        else hipLaunchKernelGGL((k_read_chain<5, 4>), dim3(gx, gy, 1), dim3(T), LDS_BYTES, c->cur, ra);
    } else {
        if (c->wide) { c->wide_unsynced = true; hipLaunchKernelGGL((k_read_chain_w<4, 4>), dim3(gx, gy, 1), dim3(T), LDS_BYTES, c->cur, ra); }
        else hipLaunchKernelGGL((k_read_chain<4, 4>), dim3(gx, gy, 1), dim3(T), LDS_BYTES, c->cur, ra);
    }
}
// write: trace steps 0 .. n_tr-1 of ct_lo * X^-row (src, read rotated), data <- normalize(data - trhi + that), d products in place
void launch_write_chain(fheram_ctx* c, GlweRef src, int rot_mul, int rot_base, GlweRef data, GlweRef trhi, const double* prep, int d, int n_tr, int gx, int gy) {
    ProfScope ps(c, "write_chain_launch", (uint64_t)gx * gy, 1);
    RowChainArgs ra;
    fill_row_chain(c, ra, prep, d, 0, n_tr);
    ra.ks.base = ks_args(c, src, src, data, trace_key(c, 0), c->gal[0], 0, rot_mul, rot_base);
    ra.ks.buf[0] = ra.ks.buf[1] = data;                           // (no trace step stores)
    ra.hi = data; ra.trhi = trhi;
    ra.ep.src = data; ra.ep.buf[0] = ra.ep.buf[1] = data;         // only the last product stores: in place on the rows
    c->wide_unsynced = true;                                      // (k_write_chain takes the whole register file)
    if (c->s_evk == 5) hipLaunchKernelGGL((k_write_chain<5, 4>), dim3(gx, gy, 1), dim3(T), LDS_BYTES, c->cur, ra);
    else hipLaunchKernelGGL((k_write_chain<4, 4>), dim3(gx, gy, 1), dim3(T), LDS_BYTES, c->cur, ra);
}
// The latency-bound end of the path (at most 8 ciphertexts, one per XCD): n trace steps as ONE launch with in-kernel
// hand-offs (k_trace_tail), followed by the fused chain launch that only runs if that one gave up.
bool use_tail(const fheram_ctx* c, int n, int gx, int gy) {
    return c->tail && c->limb_split && c->fine_split && n >= 2 && n <= CHAIN_MAX && (long)gx * gy <= TAIL_GROUPS &&
           c->cur == c->stream &&            // every launch of a context shares d_tail_sync: main stream only
           2 * c->s_evk * 3 <= 32 &&         // the workgroups of a ciphertext fit the 32 CUs of one XCD (24 with 4-limb keys, 30 with 5)
           c->cus >= TAIL_GROUPS * 32 &&     // the whole chip (8 XCDs x 32 CUs): a partition could not hold the groups side by side
           !(c->use_graph && !c->profile);   // a captured launch would replay its generation number
}
// prep != nullptr (round 6): the d external products with the prepared digits at `prep` run in front of the trace chain in the SAME launch
// (coordinate 1's products, ram.rs:454 / 525-527): src -> products -> ep_out -> trace -> b[(n - 1) & 1]; store_ep: the caller needs ep_out
// afterwards (read_prepare_write's tree[0]).  The fallback launch is then the fused row chain (k_read_chain), predicated likewise.
void launch_trace_tail(fheram_ctx* c, GlweRef src, const GlweRef (&b)[2], int start, int n, int gx, int gy,
                       const double* prep = nullptr, int d = 0, GlweRef ep_out = GlweRef{nullptr, 0, 0}, bool store_ep = false) {
    ProfScope ps(c, "keyswitch", (uint64_t)gx * gy, n);
    ProfScope pt(c, "keyswitch_tail_launch", (uint64_t)gx * gy * n, 1);
    TailArgs ta;
    ta.src = src; ta.buf[0] = b[0]; ta.buf[1] = b[1]; ta.tw = c->d_tw; ta.big = big_of(c); ta.sync = c->d_tail_sync;
    ta.n_ep = prep ? d : 0; ta.ep_out = ep_out;
    for (int i = 0; i < TAIL_EP_MAX; i++) ta.ggsw[i] = (prep && i < d) ? prep + (size_t)i * fheram_ctx::GGSW : nullptr;
    if (++c->tail_seq == 0) ++c->tail_seq;
    c->tail_launches++;
    if (c->tail_test != 1 && c->tail_launches - c->tail_launch_mark >= 64) {
        const unsigned fb = *(volatile unsigned*)c->h_tail_fb;
        if (fb - c->tail_fb_mark > 16) c->tail = 0;      // takes effect from the next chain on
        c->tail_fb_mark = fb;
        c->tail_launch_mark = c->tail_launches;
    }
    ta.seq = c->tail_seq; ta.n = n; ta.n_ct = gx * gy; ta.gx = gx; ta.xoff = c->tail_xoff; ta.give_up_at = c->tail_test ? ta.n_ep + n - 2 : -1;   // late: every buffer but the source has been overwritten by then
    KsChainArgs ca;
    ca.base = ks_args(c, src, src, b[0], trace_key(c, start), c->gal[start], 0, 0, 0);
    ca.buf[0] = b[0]; ca.buf[1] = b[1]; ca.n = n;
    ca.pred = c->d_tail_sync + TAIL_GROUPS * 32; ca.pred_seq = ta.seq; ca.host_count = c->h_tail_fb;
    for (int i = 0; i < n; i++) { ta.key[i] = ca.key[i] = trace_key(c, start + i); ta.ginv[i] = ca.ginv[i] = galois_inv_mod(galois_mod(c->gal[start + i])); }
    RowChainArgs ra;   // (the fallback of the products + trace form)
    if (ta.n_ep) {
        fill_row_chain(c, ra, prep, d, start, n);
        ra.ep.src = src; ra.ep.buf[0] = ra.ep.buf[1] = ep_out; ra.store_ep = store_ep ? 1 : 0;
        ra.ks.base = ks_args(c, b[(n - 1) & 1], b[(n - 1) & 1], b[(n - 1) & 1], trace_key(c, start), c->gal[start]);
        ra.ks.buf[0] = ra.ks.buf[1] = b[(n - 1) & 1];      // only the last step stores
        ra.hi = ra.trhi = b[(n - 1) & 1];
        ra.ks.pred = ca.pred; ra.ks.pred_seq = ca.pred_seq; ra.ks.host_count = ca.host_count;
    }
    if (c->s_evk == 5) {
        hipLaunchKernelGGL((k_trace_tail<3, 5, 3>), dim3(TAIL_GROUPS * 2 * 5 * 3), dim3(T), LDS_BYTES, c->cur, ta);
        if (ta.n_ep) hipLaunchKernelGGL((k_read_chain<5, 4>), dim3(gx, gy, 1), dim3(T), LDS_BYTES, c->cur, ra);
        else hipLaunchKernelGGL((k_keyswitch_chain<3, 5, 3>), dim3(gx, gy, 1), dim3(T), LDS_BYTES, c->cur, ca);
    } else {
        hipLaunchKernelGGL((k_trace_tail<3, 4, 3>), dim3(TAIL_GROUPS * 2 * 4 * 3), dim3(T), LDS_BYTES, c->cur, ta);
        if (ta.n_ep) hipLaunchKernelGGL((k_read_chain<4, 4>), dim3(gx, gy, 1), dim3(T), LDS_BYTES, c->cur, ra);
        else hipLaunchKernelGGL((k_keyswitch_chain<3, 4, 3>), dim3(gx, gy, 1), dim3(T), LDS_BYTES, c->cur, ca);
    }
}
// GLWE::trace(start, end): step i = rsh(1) then a += phi_{g_i}(KS(a)).
// The first step may read its input rotated by X^-(x*rot_mul) (write path, ram.rs:621,629).
void trace_steps(fheram_ctx* c, GlweRef src, GlweRef dst, GlweRef tmp, int start, int end, int gx, int gy, int rot_mul = 0, int rot_base = 0) {
    if (gx <= 0 || gy <= 0) return;
    const int n = end - start;
    if (rot_mul == 0 && rot_base == 0 && use_tail(c, n, gx, gy)) {
        // the source must survive the launch (its fallback restarts from it): out of place only
        GlweRef b[2];
        if (chain_bufs(n, src, dst, tmp, b) && !same(b[1], src)) { launch_trace_tail(c, src, b, start, n, gx, gy); return; }
    }
    if (const int split = use_mid(c, n, gx, gy, c->s_evk)) {
        GlweRef b[2];
        if (chain_bufs(n, src, dst, tmp, b)) { launch_mid_trace(c, src, b, start, n, gx, gy, rot_mul, rot_base, split); return; }
    }
    if (use_chain(c, n, gx, gy, c->s_evk) && !use_fine_split(c, gx, gy, 2 * c->s_evk * 3)) {
        GlweRef b[2];
        if (chain_bufs(n, src, dst, tmp, b)) { launch_trace_chain(c, src, b, start, n, gx, gy, rot_mul, rot_base); return; }
        if (chain_bufs(n, src, tmp, dst, b)) {
            launch_trace_chain(c, src, b, start, n, gx, gy, rot_mul, rot_base);
            launch_copy(c, tmp, dst, gx, gy);
            return;
        }
    }
    run_chain(c, end - start, src, dst, tmp, gx, gy, [&](int i, GlweRef in, GlweRef out) {
        KsArgs ka = ks_args(c, in, in, out, trace_key(c, start + i), c->gal[start + i], 0, i == 0 ? rot_mul : 0, i == 0 ? rot_base : 0);
        launch_ks_tr<KS_TRACE>(c, ka, gx, gy);
    });
}
// GLWEPacker (ram.rs:425-448), level-synchronous, over `count` leaves per y at
// src(x, y); A and B are ping-pong arenas with the same strides (src may be A).
//   n_alone    : packer levels 0..n_alone-1 in which every leaf is alone (a <- rsh(a); a <- a + phi(a))
//   first_pair : packer level of the first pairing step; level first_pair + m joins x with x + count/2^(m+1)
// Whole RAM: n_alone = first_pair = log N - ceil(log2 rows).  Row-sharded RAM: the shards run the
// levels that stay inside one residue class (same n_alone / first_pair, count = local rows) and the
// root finishes with n_alone = 0, first_pair = log N - log2(n_shards) over the gathered partials.
// Returns the arena that holds the packed result at x = 0.
// keep_alone (read_prepare_write): the rows after their alone levels are left in arena A, untouched by the
// pairing levels, which then ping-pong between P0 and P1 (Ram::write resumes trace(ct_hi) from them).
int32_t* pack_levels(fheram_ctx* c, int32_t* src, int32_t* A, int32_t* B, long sy, long sx, size_t count, int gy,
                     int n_alone, int first_pair, bool keep_alone = false, int32_t* P0 = nullptr, int32_t* P1 = nullptr) {
    const int k = ilog2_ceil(count);
    int32_t* cur = src;
    auto other = [&](int32_t* x) { return x == A ? B : A; };
    if (keep_alone && n_alone > 0 && count > 0) {
        // src -> ... -> A in n_alone out-of-place steps between A and B (src is neither)
        trace_steps(c, ref(src, sy, sx), ref(A, sy, sx), ref(B, sy, sx), 0, n_alone, (int)count, gy);
        cur = A;
    } else
    if (count > 0 && use_tail(c, n_alone, (int)count, gy) && (other(other(cur)) != cur || (P0 && P0 != cur))) {
        // at most 8 leaves (MAX_ADDR = 2^13): the single-launch trace chain of the tail.  Its source must survive the launch
        // (the fallback restarts from it): when the leaves sit in one of the arenas the third one (P0) stands in for it
        int32_t* b0 = other(cur);
        int32_t* b1 = other(b0) != cur ? other(b0) : P0;
        const GlweRef b[2] = {ref(b0, sy, sx), ref(b1, sy, sx)};
        launch_trace_tail(c, ref(cur, sy, sx), b, 0, n_alone, (int)count, gy);
        cur = ((n_alone - 1) & 1) ? b1 : b0;
    } else
    if (count > 0 && use_mid(c, n_alone, (int)count, gy, c->s_evk)) {
        int32_t* b0 = other(cur);
        int32_t* b1 = other(b0);
        const GlweRef b[2] = {ref(b0, sy, sx), ref(b1, sy, sx)};
        launch_mid_trace(c, ref(cur, sy, sx), b, 0, n_alone, (int)count, gy, 0, 0, use_mid(c, n_alone, (int)count, gy, c->s_evk));
        cur = ((n_alone - 1) & 1) ? b1 : b0;
    } else
    if (count > 0 && use_chain(c, n_alone, (int)count, gy, c->s_evk) && !use_fine_split(c, (int)count, gy, 2 * c->s_evk * 3)) {
        int32_t* b0 = other(cur);
        int32_t* b1 = other(b0);
        const GlweRef b[2] = {ref(b0, sy, sx), ref(b1, sy, sx)};
        launch_trace_chain(c, ref(cur, sy, sx), b, 0, n_alone, (int)count, gy, 0, 0);
        cur = ((n_alone - 1) & 1) ? b1 : b0;
    } else
    for (int i = 0; i < n_alone; i++) {
        int32_t* nxt = other(cur);
        KsArgs ka = ks_args(c, ref(cur, sy, sx), ref(cur, sy, sx), ref(nxt, sy, sx), trace_key(c, i), c->gal[i]);
        launch_ks_tr<KS_TRACE>(c, ka, (int)count, gy);
        cur = nxt;
    }
    size_t live = count;
    for (int m = 0; m < k; m++) {
        const int i = first_pair + m;
        const long h = (long)1 << (k - 1 - m);
        int32_t* nxt = (keep_alone && P0) ? (cur == P0 ? P1 : P0) : other(cur);
        const long n_pair = std::max<long>(0, std::min<long>(h, (long)live - h));
        const long n_alone_here = std::min<long>(h, (long)live) - n_pair;
        if (n_pair > 0) {
            KsArgs ka = ks_args(c, ref(cur, sy, sx), ref(cur + h * sx, sy, sx), ref(nxt, sy, sx), trace_key(c, i), c->gal[i], N >> (i + 1));
            launch_ks_tr<KS_PAIR>(c, ka, (int)n_pair, gy);
        }
        if (n_alone_here > 0) {
            KsArgs ka = ks_args(c, ref(cur + n_pair * sx, sy, sx), ref(cur, sy, sx), ref(nxt + n_pair * sx, sy, sx), trace_key(c, i), c->gal[i]);
            launch_ks_tr<KS_TRACE>(c, ka, (int)n_alone_here, gy);
        }
        live = std::min<size_t>(live, (size_t)h);
        cur = nxt;
    }
    return cur;
}
// CoordinatePrepared::prepare (coordinate_prepared.rs:104-116) for coordinate `ci` of addr.
int coord_first_digit(const fheram_ctx* c, int ci) { int s = 0; for (int i = 0; i < ci; i++) s += (int)c->base2d[i].size(); return s; }
// The prepared digits of coordinate ci live at prep_of(c, ci) inside d_prep ([n_digits] prepared GGSW).
double* prep_of(const fheram_ctx* c, int ci) { return c->d_prep + (size_t)coord_first_digit(c, ci) * fheram_ctx::GGSW; }
void coordinate_prepare(fheram_ctx* c, const fheram_addr* addr, int ci) {
    const int d = (int)c->base2d[ci].size();
    launch_prepare(c, addr->d_ggsw + (size_t)coord_first_digit(c, ci) * fheram_ctx::GGSW, prep_of(c, ci), d * (int)(fheram_ctx::GGSW / N));
}
// every coordinate of the address in ONE launch (the reference prepares coordinate i at the top of loop iteration i,
// ram.rs:416-419; nothing in between depends on the order)
void coordinate_prepare_all(fheram_ctx* c, const fheram_addr* addr) {
    launch_prepare(c, addr->d_ggsw, c->d_prep, c->n_digits * (int)(fheram_ctx::GGSW / N));
}
// CoordinatePrepared::prepare_inv (coordinate_prepared.rs:121-142): GGSW(X^i) -> GGSW(X^-i).
void ggsw_inverse(fheram_ctx* c, const int32_t* in, int32_t* tmp, int d) {
    const long g4 = (long)fheram_ctx::GLWE4;
    int32_t* inp = const_cast<int32_t*>(in);
    // GGSW::automorphism, column 0 of every row: res[r][0] = phi_-1(KS(in[r][0]))
    KsArgs ka = ks_args(c, ref(inp, (long)fheram_ctx::GGSW, 2 * g4), ref(inp, 0, 0), ref(tmp, (long)fheram_ctx::GGSW, 2 * g4), c->d_atk_inv, -1);
    launch_ks<KS_AUTO, 4, 5, 4>(c, ka, fheram_ctx::DNUM_CT, d);
    // row expansion with the tensor key: res[r][1] = KS_tsk(res[r][0].mask) + (0, res[r][0].body)
    KsArgs kt = ks_args(c, ref(tmp, (long)fheram_ctx::GGSW, 2 * g4), ref(tmp, 0, 0), ref(tmp + g4, (long)fheram_ctx::GGSW, 2 * g4), c->d_tsk, 1);
    launch_ks<KS_TENSOR, 4, 5, 4>(c, kt, fheram_ctx::DNUM_CT, d);
}
double* prep_inv_of(const fheram_ctx* c, int ci) { return c->d_prep_inv + (size_t)coord_first_digit(c, ci) * fheram_ctx::GGSW; }
void coordinate_prepare_inv(fheram_ctx* c, const fheram_addr* addr, int ci, int32_t* tmp, double* prep) {
    const int d = (int)c->base2d[ci].size();
    ggsw_inverse(c, addr->d_ggsw + (size_t)coord_first_digit(c, ci) * fheram_ctx::GGSW, tmp, d);
    launch_prepare(c, tmp, prep, d * (int)(fheram_ctx::GGSW / N));
}

// Under FHERAM_GRAPH=1 the enqueue functions run inside a stream capture: forked work has to be joined before the
// capture ends, and an event recorded inside one captured op cannot be waited on from another.
bool capturing(const fheram_ctx* c) { return c->use_graph && !c->profile; }
// read_prepare_write: start the inverse digits of coordinate ci on the side stream (they depend on the address and the
// keys only), behind everything enqueued on the main stream so far; Ram::write picks them up through ev_inv[ci].
// fork == false: behind what the side stream already holds (the other coordinate): one event record on the main stream
// per op, not two (a record between two dependent launches delays the second by ~13 us).
// (Starting the side work from a signal word the trace chain's launch writes when its workgroups are placed —
// hipStreamWaitValue32, no event on the main stream — was measured: the command processor polling that word for the
// ~600 us until then slows every dispatch of the main stream, read_prepare_write 0.78 -> 0.95 ms at 2^18.)
// gate_seq != 0: no event; a one-wave gate launch on the side stream waits for the trace chain launch of that generation.
void precompute_inverse(fheram_ctx* c, const fheram_addr* addr, int ci, bool fork, unsigned gate_seq = 0) {
    if (fork && c->wdone_pending) {   // behind the last write's readers of d_prep_inv (recorded at the end of that write: free)
        hipStreamWaitEvent(c->stream2, c->ev_wdone, 0);
        c->wdone_pending = false;
    }
    if (fork && gate_seq) {
        if (c->opstart_valid) { hipStreamWaitEvent(c->stream2, c->ev_opstart, 0); c->opstart_valid = false; }   // not before the op that parks it has started
        hipLaunchKernelGGL(k_tail_gate, dim3(1), dim3(64), 0, c->stream2, c->d_tail_sync + TAIL_GROUPS * 32 + 2, gate_seq);
    } else if (fork) {
        hipEventRecord(c->ev_fork, c->stream);
        hipStreamWaitEvent(c->stream2, c->ev_fork, 0);
    }
    hipStream_t keep = c->cur;
    c->cur = c->stream2;
    coordinate_prepare_inv(c, addr, ci, c->d_ggsw_inv + (size_t)coord_first_digit(c, ci) * fheram_ctx::GGSW, prep_inv_of(c, ci));
    hipEventRecord(c->ev_inv[ci], c->stream2);
    c->cur = keep;
    c->inv_id[ci] = addr->id;
    c->inv_pending[ci] = true;
}
void wait_inverse(fheram_ctx* c, hipStream_t s, int ci) { if (!capturing(c)) hipStreamWaitEvent(s, c->ev_inv[ci], 0); }

}  // namespace
