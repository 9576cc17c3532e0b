/*
 * fheram.h — C ABI of the MI355X-native FHE-RAM evaluator.
 *
 * Drop-in boundary for the hot path of phantomzone-org/fhe-ram (reference snapshot
 * 2026-02-13): Ram::read / Ram::read_prepare_write / Ram::write and the types either side of
 * them.  The reference has no FFI of its own (pure Rust over Poulpy); these entry points are
 * what a Rust `extern "C"` shim behind the crate's public API (src/lib.rs:12-21) binds — see
 * INTEGRATION.md for the binding.  Each entry cites the reference interface it replaces.
 *
 * Host data interchange = Poulpy's host layouts, little-endian int64:
 *   GLWE  (size limbs)                : [limb][col 2][N]
 *   GGSW  (dnum rows, size limbs)     : [row][col_in 2][limb][col_out 2][N]
 *   GGLWE key (dnum rows, size limbs) : [row][col_in 1][limb][col_out 2][N]
 * All uploaded limbs must be normalised: -2^(base2k-1) <= x < 2^(base2k-1).
 *
 * Threading contract = the reference's: every op takes `&mut self` (ram.rs:172,196,226), so
 * one op in flight per context; a context is not thread-safe.
 * Errors: the reference panics (assert!); this ABI returns a status and keeps a message.
 */
#ifndef FHERAM_H
#define FHERAM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fheram_ctx fheram_ctx;   /* Ram<B> + Parameters<B> + EvaluationKeysPrepared (ram.rs:25-29) */
typedef struct fheram_addr fheram_addr; /* Address (address.rs:21-24), device resident              */

enum fheram_status {
    FHERAM_OK = 0,
    FHERAM_ERR_INVALID_ARG = 1,   /* size/layout asserts: ram.rs:144-155,243,404,482          */
    FHERAM_ERR_STATE = 2,         /* state-machine asserts: ram.rs:393-396,472-475,555-558   */
    FHERAM_ERR_UNINITIALIZED = 3, /* "unitialized memory": ram.rs:182-185,206-209            */
    FHERAM_ERR_KEYS = 4,          /* missing key / auto_key.p() != -1: coordinate_prepared.rs:134 */
    FHERAM_ERR_UNSUPPORTED = 5,   /* parameter set the kernels are not built for             */
    FHERAM_ERR_RANGE = 6,         /* a limb is not normalised                                */
    FHERAM_ERR_DEVICE = 7,        /* HIP runtime error                                       */
    FHERAM_ERR_PRECISION = 8      /* the FP64 round-off monitor saw |x - rint(x)| > 3/8 (no reference counterpart: Poulpy's FFT64
                                     backend, examples/fhe-ram.rs:3-7, rounds unchecked)     */
};

/* Parameters (parameters.rs:11-21,147-152). */
typedef struct fheram_params {
    uint32_t log_n;          /* LOG_N = 12                     parameters.rs:11 */
    uint32_t base2k;         /* BASE2K = 17                    parameters.rs:12 */
    uint32_t rank;           /* RANK = 1                       parameters.rs:13 */
    uint32_t k_glwe_pt;      /* K_GLWE_PT = 3                  parameters.rs:14 */
    uint32_t k_glwe_ct;      /* K_GLWE_CT = 3*BASE2K           parameters.rs:15 */
    uint32_t k_ggsw_addr;    /* K_GGSW_ADDR = 4*BASE2K         parameters.rs:16 */
    uint32_t k_evk_trace;    /* K_EVK_TRACE = 4*BASE2K         parameters.rs:17 */
    uint32_t k_evk_ggsw_inv; /* K_EVK_GGSW_INV = 5*BASE2K      parameters.rs:18 */
    uint32_t word_size;      /* WORDSIZE = 4                   parameters.rs:20 */
    uint32_t n_decomp;       /* len(DECOMP_N)                  parameters.rs:19 */
    uint8_t decomp_n[16];    /* DECOMP_N = [3,3,3,3]                            */
    uint64_t max_addr;       /* MAX_ADDR = 1<<14               parameters.rs:21 */
} fheram_params;

/* Parameters::new() defaults (parameters.rs:167-176). */
int fheram_params_default(fheram_params* out);

/* Ram::new / Ram::new_from_ram_params (ram.rs:59-87): allocates every device buffer the
 * context will ever need (rows, tree, packing scratch, prepared operands) on HIP device
 * `device`.  Mirrors Ram owning `subrams` + `scratch`. */
int fheram_ctx_create(const fheram_params* params, int device, fheram_ctx** out);
void fheram_ctx_destroy(fheram_ctx* ctx);
/* Message of the last failing call on ctx (ctx == NULL: last fheram_ctx_create failure). */
const char* fheram_last_error(const fheram_ctx* ctx);

/* Layout sizes in int64 elements (GLWELayout/GGSWLayout/GGLWELayout, parameters.rs:53-104). */
size_t fheram_glwe_len(const fheram_ctx* ctx);      /* glwe_ct_infos                    */
size_t fheram_ggsw_len(const fheram_ctx* ctx);      /* ggsw_infos                       */
size_t fheram_atk_len(const fheram_ctx* ctx);       /* evk_glwe_infos (one key)         */
size_t fheram_evk_inv_len(const fheram_ctx* ctx);   /* evk_ggsw_infos (one key)         */
size_t fheram_rows(const fheram_ctx* ctx);          /* GLWE rows per sub-RAM            */
int fheram_n_digits(const fheram_ctx* ctx);         /* GGSW digits of one Address       */
int fheram_n_coordinates(const fheram_ctx* ctx);    /* Address::n2 (address.rs:113)     */

/* EvaluationKeysPrepared::alloc + prepare (keys.rs:34-71): std-form keys in, device-prepared
 * (transform-domain) keys kept in the context.  gal_els must be GLWE::trace_galois_elements
 * (-1, 5^(2^(i-1)) mod 2N); atk_ggsw_inv_p must be -1 (coordinate_prepared.rs:134). */
int fheram_keys_load(fheram_ctx* ctx, const int64_t* gal_els, int n_gal, const int64_t* const* atk_glwe,
                     const int64_t* atk_ggsw_inv, int64_t atk_ggsw_inv_p, const int64_t* tsk_ggsw_inv);

/* Ram::encrypt_sk result hand-over (ram.rs:129-167): rows = [word_size][rows][GLWE]. The
 * encryption itself is the caller's (host) business.  Resets state to "readable". */
int fheram_ram_upload(fheram_ctx* ctx, const int64_t* rows);
int fheram_ram_download(fheram_ctx* ctx, int64_t* rows);
/* SubRam::tree level `level`, entry 0 of every sub-RAM: [word_size][GLWE] (ram.rs:300). */
int fheram_ram_tree_download(fheram_ctx* ctx, int level, int64_t* out);
/* SubRam::state (ram.rs:302): 1 after read_prepare_write, 0 after write. */
int fheram_ram_state(const fheram_ctx* ctx);

/* Address::alloc_from_params + encrypt_sk result hand-over (address.rs:58,86-109): n_ggsw
 * std-form GGSW digits, coordinate-major (coordinate 0 digits first). */
int fheram_address_create(fheram_ctx* ctx, const int64_t* const* ggsw, int n_ggsw, fheram_addr** out);
void fheram_address_destroy(fheram_addr* addr);

/* Ram::read (ram.rs:172-191).  out: [word_size][GLWE] int64, or NULL to leave the result on
 * the device (fetch it later with fheram_result_download). */
int fheram_read(fheram_ctx* ctx, const fheram_addr* addr, int64_t* out);
/* Ram::read_prepare_write (ram.rs:196-222). */
int fheram_read_prepare_write(fheram_ctx* ctx, const fheram_addr* addr, int64_t* out);
/* Ram::write (ram.rs:226-294).  w: n_w GLWEs, each encrypting [w,0,...,0] (ram.rs:228);
 * w == NULL uses the words staged by fheram_word_stage. */
int fheram_write(fheram_ctx* ctx, const int64_t* w, int n_w, const fheram_addr* addr);
int fheram_word_stage(fheram_ctx* ctx, const int64_t* w, int n_w);
int fheram_result_download(fheram_ctx* ctx, int64_t* out);
/* The result of the last read / read_prepare_write in place: [word_size][GLWE] int64 in the context's pinned host buffer
 * (the device widens it straight into host memory; no copy on the host).  Valid until the next operation on ctx.  A Rust
 * shim builds its Vec<GLWE<Vec<u8>>> (ram.rs:176) from it with one copy instead of two. */
int fheram_result_map(fheram_ctx* ctx, const int64_t** out);
/* Block until every queued operation of ctx has finished. */
int fheram_sync(fheram_ctx* ctx);

/* ---- The exactness contract, checked.  The reference multiplies with Poulpy's FFT64 backend and rounds (examples/fhe-ram.rs:3-7);
 * so does csrc/fft_dev.hpp: the rounded output of an inverse transform is the exact integer only while the accumulated FP64
 * round-off stays below 1/2.  No a-priori bound below 1/2 is known for six accumulated terms of extreme limbs (measured worst
 * case found by search 0.25: DESIGN.md §2), so every rounding on the path reports |x - rint(x)| to a per-context monitor
 * (fheram_config.monitor: one coefficient per thread and transform by default, every coefficient under `safe`).
 *   fheram_roundoff_max: waits for the context's streams; *max_out = the largest round-off seen since creation / the last reset;
 *                        returns FHERAM_ERR_PRECISION if it exceeds 3/8 — halfway between the largest round-off a search over
 *                        extreme operands has produced (1/4, tools/fft_search.hip) and failure (1/2); max_out is still written.
 *   Once a round-off above 3/8 was seen, every call that waits for the device (fheram_sync, the downloads of fheram_read /
 *   fheram_read_prepare_write / fheram_result_map / fheram_ram_download, fheram_timer_end) returns FHERAM_ERR_PRECISION until
 *   fheram_roundoff_reset.  No reference counterpart. */
int fheram_roundoff_max(fheram_ctx* ctx, double* max_out);
int fheram_roundoff_reset(fheram_ctx* ctx);

/* ---- Row-sharded RAM across the GPUs of a node.  The reference is single
 * threaded and has no counterpart; the split follows its control flow: the rows of one residue class
 * mod n_shards form a complete sub-tree of the GLWEPacker (bit-reversed feed, ram.rs:425-444), so a
 * shard that owns rows r = shard (mod n_shards) of every sub-RAM runs SubRam::read up to and
 * including its own packing levels; one exchange (all-gather of word_size GLWEs per shard) later the
 * root finishes the top log2(n_shards) levels, the coordinate-1 products and the trace.  Every
 * combine sees the same operands as in the sequential packer, so results are bit-identical.
 * Buffers are either host int64 ([..][GLWE], *_on_device = 0) or device int32 in the same
 * [limb][col][N] order (*_on_device = 1; the pointer must be valid on the context's device —
 * this is what an RCCL collective moves). */
int fheram_ctx_create_sharded(const fheram_params* params, int device, int shard, int n_shards, fheram_ctx** out);

/* Execution switches of a context: which decomposition / hand-over form the launchers of csrc/launch.hpp choose.  Every setting
 * computes the same results (tests/test_gpu_parity.py and test_gpu_golden.py force each one); the defaults are the fastest
 * measured forms.  fheram_config_default() fills the library defaults and then applies the FHERAM_* environment overrides of
 * the same names (FHERAM_LIMB_SPLIT, _FINE_SPLIT, _MEMO, _PRE_INV, _TAIL, _TAIL_EP, _MID, _CHAIN, _CHAIN_Y, _PAIR_Z, _FUSE, _GRAPH,
 * _SAFE, _NCO, _MONITOR); fheram_ctx_create / _sharded use exactly that.  No reference counterpart (the reference has one code path). */
typedef struct fheram_config {
    int32_t limb_split;   /* 1: limb-parallel launches for batches far smaller than the chip */
    int32_t fine_split;   /* 1: one forward + one inverse transform per workgroup for <= 10 key-switches / <= 5 products */
    int32_t memo;         /* 1: Ram::write resumes from what read_prepare_write computed on the unchanged state */
    int32_t pre_inv;      /* 1: read_prepare_write starts the write's inverse digits behind a gate wave; 2: behind an event; 0: never */
    int32_t tail;         /* 1: the trace chain at the end of a read as one launch with in-kernel hand-offs (k_trace_tail) */
    int32_t tail_test;    /* test hook: 1 / 2 = every such launch gives up late (2 keeps the watch active) */
    int32_t mid;          /* 2: dependent chains on 9..64 ciphertexts as one launch (k_chain_mid); 1: the <= 16 split only; 0: off */
    int32_t mid_test;     /* test hook: one member of every such launch gives up */
    int32_t chain;        /* 1: a dependent chain of fused steps as one launch */
    int32_t chain_y;      /* 3: trace chains hand over through LDS and registers (ks_trace_l); 0: int32 limbs through global memory */
    int32_t pair_z;       /* 1: the column-split packer combine in closed form (k_pair_z) */
    int32_t fuse;         /* 1: a row's product chain and trace chain as one launch (k_read_chain / k_write_chain) */
    int32_t graph;        /* 1: replay each op's launch sequence from a hipGraph */
    int32_t safe;         /* 1: no in-kernel hand-offs between workgroups, no gate wave: stays inside the HIP memory model */
    int32_t nco;          /* output columns per workgroup: 1, 2, or 0 = chosen per launch */
    int32_t tail_ep;      /* 1 (default): coordinate 1's external products (>= 2 digits) run inside that launch, in front of the trace steps; 0: launches of their own */
    int32_t monitor;      /* round-off monitor (fheram_roundoff_max): 1 (default) every rounding of an inverse transform reports one of the thread's eight coefficients (which one differs between call sites); 2 all eight (what `safe` selects); 0 nothing is reported or checked */
    int32_t reserved;     /* must be 0 (a later version / size field): fheram_ctx_create_cfg refuses anything else */
} fheram_config;
/* ALWAYS start from fheram_config_default(): a zero-initialised struct is a valid configuration, but it selects every slow
 * path (no chains, no fused launches, no memo) and switches the round-off monitor off. */
void fheram_config_default(fheram_config* cfg);
/* fheram_ctx_create_sharded with explicit switches (cfg == NULL: fheram_config_default) */
int fheram_ctx_create_cfg(const fheram_params* p, int device, int shard, int n_shards, const fheram_config* cfg, fheram_ctx** out);
/* the switches IN EFFECT on this context (after `safe`, `graph` and `memo` have been applied to the others) */
int fheram_ctx_config(const fheram_ctx* ctx, fheram_config* out);

/* Stream ordering for DEVICE buffers (*_on_device = 1).  The evaluator enqueues on its own HIP stream;
 * a collective (RCCL) runs on the caller's.  Hand-overs of device buffers are asynchronous: no call
 * below blocks the host for them.  The caller orders the two streams with events instead:
 *   fheram_stream_signal(ctx, s): work enqueued on s AFTER this call waits for everything the context
 *                                 has enqueued so far (call it after fheram_read_partial / fheram_write_root,
 *                                 before the collective);
 *   fheram_stream_wait(ctx, s):   work the context enqueues AFTER this call waits for everything enqueued
 *                                 on s so far (call it after the collective, before fheram_read_finish /
 *                                 fheram_write_shard).
 * s is a hipStream_t (NULL = the legacy default stream).  Host buffers (*_on_device = 0) need neither. */
int fheram_stream_signal(fheram_ctx* ctx, void* hip_stream);
int fheram_stream_wait(fheram_ctx* ctx, void* hip_stream);
/* Exchange buffers on the context's device for a host that has no HIP binding of its own (hipMalloc /
 * hipFree): `bytes` of device memory, e.g. n_shards * word_size * fheram_glwe_len * 4 for the gathered
 * partials.  A host that already owns device memory (an RCCL / torch buffer) passes that instead. */
int fheram_device_malloc(fheram_ctx* ctx, size_t bytes, void** out);
int fheram_device_free(fheram_ctx* ctx, void* ptr);
int fheram_shard_info(const fheram_ctx* ctx, int* shard, int* n_shards, size_t* local_rows);
/* Every shard.  out: word_size partial GLWEs.  prepare_write != 0 keeps the rotated rows (ram.rs:502-504). */
int fheram_read_partial(fheram_ctx* ctx, const fheram_addr* addr, int prepare_write, void* out, int out_on_device);
/* Root.  partials: [n_shards][word_size] GLWEs in shard order.  out as in fheram_read (may be NULL). */
int fheram_read_finish(fheram_ctx* ctx, const fheram_addr* addr, int prepare_write, const void* partials,
                       int partials_on_device, int64_t* out);
/* Root: write_first_step + inverse coordinate-1 products (ram.rs:254-256,260-271,610).  ct_lo_out:
 * word_size GLWEs to broadcast. */
int fheram_write_root(fheram_ctx* ctx, const int64_t* w, int n_w, const fheram_addr* addr, void* ct_lo_out, int out_on_device);
/* Every shard, optional: start the part of a write that does not depend on ct_lo — trace(ct_hi) of the
 * local rows (ram.rs:616) and the inversion of coordinate 0 (ram.rs:278-289) — so that it runs while the
 * root computes ct_lo and the broadcast is in flight.  fheram_write_root and fheram_write_shard start
 * it themselves when it has not been started. */
int fheram_write_begin(fheram_ctx* ctx, const fheram_addr* addr);
/* Every shard: write_mid_step on its rows with the broadcast ct_lo, then write_last_step (ram.rs:612-630,644-648). */
int fheram_write_shard(fheram_ctx* ctx, const fheram_addr* addr, const void* ct_lo, int on_device);

/* ---- ONE RAM over the GPUs of a node behind ONE handle (reference API kept: one call per op,
 * ram.rs:172-176,196-200,226-231).  For a host that is a single process (a Rust `Ram` wrapper): the group owns one
 * row-shard context per device (rows r = shard (mod n) of every sub-RAM, as above), one host thread per device drives
 * it, and the two exchange steps — every shard's partial pack to the root for a read, the root's ct_lo to every shard
 * for a write — are peer-to-peer device copies (xGMI) ordered by events: no host staging, no collective library.
 * fhe-ram_amd/sharded.py drives the per-shard entry points above from one PROCESS per GPU over RCCL instead; both give
 * the results of the unsharded path bit for bit.  n_devices must be a power of two <= rows per sub-RAM (1 = plain
 * context); the same device may be named several times (rehearsal on one GPU).  Group ops are synchronous: they return
 * when the op is complete on every shard.  One op in flight per group (the reference's `&mut self`). */
typedef struct fheram_group fheram_group;
typedef struct fheram_group_addr fheram_group_addr;
int fheram_group_create(const fheram_params* params, const int* devices, int n_devices, fheram_group** out);   /* Ram::new, ram.rs:59-87 */
void fheram_group_destroy(fheram_group* grp);
const char* fheram_group_last_error(const fheram_group* grp);   /* grp == NULL: last fheram_group_create failure */
int fheram_group_size(const fheram_group* grp);
/* the shard context behind the group (setup-side calls per shard: fheram_ram_encrypt_sk, fheram_keys_encrypt_sk, ...);
 * not to be used while a group op is in flight */
fheram_ctx* fheram_group_ctx(fheram_group* grp, int shard);
int fheram_group_keys_load(fheram_group* grp, const int64_t* gal_els, int n_gal, const int64_t* const* atk_glwe,
                           const int64_t* atk_ggsw_inv, int64_t atk_ggsw_inv_p, const int64_t* tsk_ggsw_inv);   /* keys.rs:57-71, replicated */
/* rows: the WHOLE RAM [word_size][rows][GLWE] (Ram::encrypt_sk output, ram.rs:129-167); scattered / gathered by shard */
int fheram_group_ram_upload(fheram_group* grp, const int64_t* rows);
int fheram_group_ram_download(fheram_group* grp, int64_t* rows);
int fheram_group_ram_tree_download(fheram_group* grp, int level, int64_t* out);
int fheram_group_ram_state(const fheram_group* grp);
int fheram_group_address_create(fheram_group* grp, const int64_t* const* ggsw, int n_ggsw, fheram_group_addr** out);   /* address.rs:58,86-109, replicated */
void fheram_group_address_destroy(fheram_group_addr* addr);
int fheram_group_read(fheram_group* grp, const fheram_group_addr* addr, int64_t* out);                 /* Ram::read, ram.rs:172-191 */
int fheram_group_read_prepare_write(fheram_group* grp, const fheram_group_addr* addr, int64_t* out);   /* ram.rs:196-222 */
int fheram_group_write(fheram_group* grp, const int64_t* w, int n_w, const fheram_group_addr* addr);   /* ram.rs:226-294; w == NULL: staged words */
int fheram_group_word_stage(fheram_group* grp, const int64_t* w, int n_w);
int fheram_group_result_download(fheram_group* grp, int64_t* out);
/* Exchange path of the group (round 4).  fheram_group_create runs one self-test copy per shard in both directions (shard -> root as a
 * read's partial pack, root -> shard as a write's ct_lo) and fails with FHERAM_ERR_DEVICE, naming the pair, if the bytes do not arrive.
 * direct[i] = 1: the runtime reports a peer-to-peer path between shard i's device and the root's (or they are one device); 0: staged. */
int fheram_group_peer_info(const fheram_group* grp, int* direct, int n);
/* 1 after a group op failed part-way (work had been enqueued on some shard): rows and state flags of the shards are inconsistent,
 * every further op returns FHERAM_ERR_STATE until fheram_group_ram_upload has replaced the rows.  Calls the reference would refuse
 * with an assert (wrong state, foreign address, ram.rs:393-396,404,472-475,555-558) are refused before anything is enqueued and
 * poison nothing. */
int fheram_group_poisoned(const fheram_group* grp);
/* the largest round-off over the shards' monitors (fheram_roundoff_max on every shard) */
int fheram_group_roundoff_max(fheram_group* grp, double* max_out);

/* ---- Poulpy-level operations reached from the path, exposed for
 * parity tests and micro-benchmarks.  Inputs/outputs are host buffers in the layouts above. */

/* glwe_external_product (coordinate_prepared.rs:156): res[i] = a[i] (x) ggsw, i < batch. */
int fheram_glwe_external_product(fheram_ctx* ctx, const int64_t* a, int batch, const int64_t* ggsw, int64_t* res);
/* glwe_automorphism family with the loaded trace key of Galois element gal_el:
 * mode 0: res = phi(KS(a));  1: res = a + phi(KS(a));  2: res = a - phi(KS(a)). */
int fheram_glwe_automorphism(fheram_ctx* ctx, int mode, int64_t gal_el, const int64_t* a, int batch, int64_t* res);
/* GLWE::trace(start, end) (ram.rs:457,540,572,616,621). */
int fheram_glwe_trace(fheram_ctx* ctx, int start, int end, const int64_t* a, int batch, int64_t* res);
/* GLWEPacker (ram.rs:425-448): packs coefficient 0 of cts[0..count) into coefficients
 * 0..count of one GLWE, feeding them in the RAM's bit-reversed order. */
int fheram_glwe_pack(fheram_ctx* ctx, const int64_t* cts, int count, int64_t* out);
/* GGSW::automorphism(p=-1) + tensor-key row expansion (coordinate_prepared.rs:138). */
int fheram_ggsw_automorphism_inv(fheram_ctx* ctx, const int64_t* ggsw_in, int64_t* ggsw_out);

/* ---- Setup side on the device: secret-key encryption of the RAM, of
 * addresses and of the evaluation keys, and decryption.  SAMPLING STAYS ON THE HOST: the caller
 * draws the uniform mask limbs and the rounded-Gaussian noise polynomials from its own sources
 * (Poulpy's `Source` in a Rust host: source_xa / source_xe of the calls cited below) and hands them
 * over; the device does the arithmetic (products with the secret, limb normalisation, key
 * preparation).  The same draws therefore give the same ciphertexts as the host implementation.
 * Per GLWE of `size` limbs, in the order the reference encrypts them: mask = size*N limbs in
 * [-2^16, 2^16) (limb-major), noise = N integers already scaled to the ciphertext precision k
 * (added on limb ceil(k/base2k)-1; |e| < 2^30). */
typedef struct fheram_secret fheram_secret;
/* GLWESecret + GLWESecretPrepared (examples/fhe-ram.rs:49-59): sk = N coefficients in {-1,0,1}. */
int fheram_secret_create(fheram_ctx* ctx, const int64_t* sk, fheram_secret** out);
void fheram_secret_destroy(fheram_secret* sk);
/* GLWE::encrypt_sk (Poulpy; call sites ram.rs:369-376, examples/fhe-ram.rs:179-210) on n_glwe
 * ciphertexts of `size` limbs (3, 4 or 5) at precision k.  pt: [n_glwe][pt_size][N] normalised
 * limbs or NULL (encryption of zero); pt_col 0 adds it to the body, 1 to the mask column after
 * the product with the secret (GGSW rows).  out: [n_glwe][size][2][N]. */
int fheram_glwe_encrypt_sk(fheram_ctx* ctx, const fheram_secret* sk, int n_glwe, int size, int k, const int64_t* pt,
                           int pt_size, int pt_col, const int64_t* mask, const int64_t* noise, int64_t* out);
/* GLWE::decrypt (examples/fhe-ram.rs:217-222): pt[i] = normalise(body + mask*s), [n_glwe][size][N]. */
int fheram_glwe_decrypt(fheram_ctx* ctx, const fheram_secret* sk, int n_glwe, int size, const int64_t* ct, int64_t* pt);
/* Ram::encrypt_sk (ram.rs:129-167, SubRam::encrypt_sk ram.rs:334-380): data = max_addr*word_size
 * bytes, interleaved by word as in the reference.  mask [word_size][rows][3][N], noise
 * [word_size][rows][N] for the rows THIS context holds (all of them unless sharded; a shard owns
 * global rows shard + x*n_shards).  The rows are encrypted straight into device memory. */
int fheram_ram_encrypt_sk(fheram_ctx* ctx, const fheram_secret* sk, const uint8_t* data, size_t data_len,
                          const int64_t* mask, const int64_t* noise);
/* Address::encrypt_sk (address.rs:86-109) -> Coordinate::encrypt_sk (coordinate.rs:121-180): GGSW
 * digits of -value per coordinate.  mask [n_digits][3 rows][2 col_in][4][N], noise
 * [n_digits][3][2][N], digits coordinate-major.  The address is created on the device. */
int fheram_address_encrypt_sk(fheram_ctx* ctx, const fheram_secret* sk, uint32_t value, const int64_t* mask,
                              const int64_t* noise, fheram_addr** out);
/* The std-form GGSW digits of an address, [n_digits][fheram_ggsw_len] (parity tests, hand-over to a host). */
int fheram_address_download(fheram_ctx* ctx, const fheram_addr* addr, int64_t* out);
/* EvaluationKeys::encrypt_sk (keys.rs:135-180) + EvaluationKeysPrepared::prepare (keys.rs:57-71):
 * the log2(N) trace keys in GLWE::trace_galois_elements order (3 rows of 4 limbs each), then the
 * tensor key, then the p = -1 key (4 rows of 5 limbs each): mask = (36*4 + 8*5)*N limbs, noise =
 * 44*N.  Keys are generated and prepared on the device; std_out (optional, may be NULL) receives
 * the std forms [12][fheram_atk_len] ++ tensor [fheram_evk_inv_len] ++ inverse [fheram_evk_inv_len]. */
int fheram_keys_encrypt_sk(fheram_ctx* ctx, const fheram_secret* sk, const int64_t* mask, const int64_t* noise,
                           int64_t* std_out);

/* ---- Address::set_from_fheuint (conversion.rs:18-82).
 * *** CONTRACT-COMPATIBLE ONLY — THE INPUT LAYOUT IS NOT poulpy-schemes' ***  A Rust host cannot hand its own FheUintPrepared
 * to fheram_fheuint_create: poulpy-schemes 0.3.3 is un-vendored (Cargo.toml:10), so neither the memory layout of
 * FheUintPrepared nor the algorithm of scalar_to_ggsw_blind_rotation could be read; the type below is this library's own.  What
 * these entry points guarantee is the reference test's contract (conversion.rs:100-220): digit d decrypts to
 * X^{((k >> bit_rsh) mod 2^bit_mask) << bit_lsh} on the gadget within the test's noise bound.  tools/poulpy_kat (feature
 * kat_fheuint) exports a real FheUintPrepared and one derived digit for the first Rust-equipped run to compare against.
 * The reference derives every address
 * digit from an encrypted integer with poulpy-schemes' scalar_to_ggsw_blind_rotation (un-vendored); what is built
 * here is its CONTRACT (conversion.rs:41-65 and the test at :100-220), by CMux chains on noiseless GGSW rows with
 * the external-product kernels — a self-consistent restatement, bit-exact against the in-repo oracle only.
 * FheUintPrepared<u32> = one GGSW per bit, LSB first, in the layout of the GGSW-inversion keys
 * (k = k_evk_ggsw_inv, 5 limbs, dnum 4): [n_bits][row 4][col_in 2][limb 5][col_out 2][N] int64. */
typedef struct fheram_fheuint fheram_fheuint;
size_t fheram_fheuint_ggsw_len(const fheram_ctx* ctx);
/* FheUintPrepared::alloc + prepare from host ciphertexts. */
int fheram_fheuint_create(fheram_ctx* ctx, const int64_t* bits, int n_bits, fheram_fheuint** out);
/* FheUintPrepared::encrypt_sk (conversion.rs:160-168) on the device with host-drawn randomness (as the other
 * encrypt_sk entry points): mask [n_bits][4][2][5][N], noise [n_bits][4][2][N]. */
int fheram_fheuint_encrypt_sk(fheram_ctx* ctx, const fheram_secret* sk, uint32_t value, int n_bits, const int64_t* mask,
                              const int64_t* noise, fheram_fheuint** out);
int fheram_fheuint_download(fheram_ctx* ctx, const fheram_fheuint* fu, int64_t* out);
void fheram_fheuint_destroy(fheram_fheuint* fu);
/* Address::set_from_fheuint (conversion.rs:68-82): digit d = GGSW of X^{+-(((k >> bit_rsh) mod 2^bit_mask) << bit_lsh)}
 * over the context's digit plan; sign != 0: + (what the reference's test decrypts to), 0: - (the convention of
 * Address::encrypt_sk, address.rs:102-108, i.e. an address Ram::read accepts). */
int fheram_address_set_from_fheuint(fheram_ctx* ctx, const fheram_fheuint* fu, int sign, fheram_addr** out);

/* ---- Measurement hooks (bench.py).  HIP events recorded on the context's own stream. */
int fheram_timer_begin(fheram_ctx* ctx);
int fheram_timer_end(fheram_ctx* ctx, float* elapsed_ms);
/* Per-kernel-class timing: when enabled, every launch of the hot kernels is bracketed by HIP
 * events on the launch stream; totals are read back per class name
 * ("ext_product", "keyswitch", "prepare", "elementwise").  on = 2: only the chain launches themselves
 * ("keyswitch_chain_launch": two events per launch of the dominant kernel, nothing else — back-to-back submission of
 * the other launches stays as in an unprofiled run). */
int fheram_profile_enable(fheram_ctx* ctx, int on);
int fheram_profile_get(fheram_ctx* ctx, const char* kernel_class, uint64_t* launches, uint64_t* blocks, double* total_ms);
int fheram_profile_reset(fheram_ctx* ctx);
/* The trace chain at the end of a read (ram.rs:457,540) runs as one launch whose workgroups hand over inside the
 * kernel; a launch that could not assemble its workgroup groups (CUs held by another context) gives up and the fused
 * launch enqueued behind it redoes the chain, with the same result.  launches = such launches since the context was
 * created, fallbacks = how many of them gave up.  Waits for the context's stream. */
int fheram_tail_stats(fheram_ctx* ctx, uint64_t* launches, uint64_t* fallbacks);
/* The same for the dependent chains on 9..64 ciphertexts (the alone packer levels, CoordinatePrepared::product[_inplace]
 * and write_mid_step's traces at MAX_ADDR = 2^14 — the source default — to 2^16), which run as one launch with in-kernel
 * hand-offs too (k_chain_mid).  Each ciphertext's workgroups stand alone there: fallbacks = CIPHERTEXTS redone by the fused
 * launch behind (0 on a GPU the context has to itself). */
int fheram_mid_stats(fheram_ctx* ctx, uint64_t* launches, uint64_t* fallbacks);
/* enabled: the FHERAM_MID setting in effect (0 = the path has switched the single-launch mid chains off because their launches kept
 * giving up — two windows of 64 launches in a row with more than a quarter of the ciphertexts redone — and will try them again
 * 256 ops later); times_disabled: how often that has happened on this context */
int fheram_mid_state(const fheram_ctx* ctx, int* enabled, uint64_t* times_disabled);
/* BASELINE.json configs[1]: one GLWE x GGSW external product at N = 4096 on device-resident synthetic
 * operands (normalised limbs; the work is data independent).  Runs `iters` launches of `batch` products
 * back to back on the context's stream, each launch consuming the previous one's output (a dependent
 * chain, as CoordinatePrepared::product is, coordinate_prepared.rs:147-161), and returns the HIP-event
 * time of the whole chain in ms.  batch = 1 gives the latency of one product, batch >= #CUs the throughput. */
int fheram_bench_external_product(fheram_ctx* ctx, int batch, int iters, float* total_ms);
/* The two dependent chains that carry the path's work, on device-resident synthetic operands, `iters` times back to back:
 * kind 0 = GLWE::trace(0, n) on `batch` ciphertexts (the packer levels in which a row is alone, ram.rs:514, and
 * write_mid_step's traces, ram.rs:616,621), kind 1 = CoordinatePrepared::product over n digits (coordinate_prepared.rs:147-177).
 * Returns the HIP-event time of all iterations in ms. */
int fheram_bench_chain(fheram_ctx* ctx, int kind, int batch, int n, int iters, float* total_ms);
/* Device properties of the context's GPU (name, CU count) for the bench report. */
int fheram_device_info(const fheram_ctx* ctx, char* name, size_t name_len, int* compute_units);

/* ---- Self-test of the arithmetic the kernels are built on (no reference counterpart: the reference delegates its
 * arithmetic to Poulpy's FFT64 backend, examples/fhe-ram.rs:3-7, whose contract — the FP64 round-off of a product of
 * normalised limbs stays below 1/2, so rounding gives the exact integer — is the contract of csrc/fft_dev.hpp too).
 * n_terms (even, <= 8) pairs of polynomials of N int32 coefficients in; out[0][N] = sum_r a_r * g_r and
 * out[1][N] = sum_r a_r * g_{r ^ 1} (negacyclic) as RAW doubles, BEFORE the rounding the path applies, through the
 * transforms, the prepared-operand scaling and the multiply-accumulate exactly as the fused kernels call them
 * (singles != 0: every transform on its own instead of two at a time).  The caller compares with exact integer arithmetic
 * (tests/test_gpu_fft.py).  Not on the RAM path. */
int fheram_selftest_convolve(fheram_ctx* ctx, int n_terms, const int32_t* a, const int32_t* g, double* out, int singles);
/* The same products with the ROUNDING of the path (nat_out) and therefore through the round-off monitor: out = the rounded
 * sums.  operand_scale multiplies the prepared operands (1.0: the path's arithmetic; 0.5 makes every odd sum a half-integer,
 * which is how tests/test_gpu_fft.py drives the monitor over its limit and checks FHERAM_ERR_PRECISION). */
int fheram_selftest_convolve_rounded(fheram_ctx* ctx, int n_terms, const int32_t* a, const int32_t* g, double* out, double operand_scale);

#ifdef __cplusplus
}
#endif
#endif /* FHERAM_H */
