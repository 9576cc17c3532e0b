// Private to fheram.hip: Ram::read / read_prepare_write / write as launch sequences (reference: src/ram.rs).
#pragma once
#include "launch.hpp"

namespace {

// The launch sequence of an op is a function of (context, address, op) and of a few bits of the context's state (what a
// write may resume from): with FHERAM_GRAPH=1 it is captured once per address and state signature into a hipGraph and
// replayed, instead of being re-enqueued kernel by kernel.
template <typename F>
int run_op(fheram_ctx* c, const fheram_addr* addr, int which, F&& enqueue) {
    if (!c->use_graph || c->profile) return enqueue();
    fheram_addr* a = const_cast<fheram_addr*>(addr);
    // what the enqueue function reads of the context's mutable state (a write resumes from what read_prepare_write kept —
    // or not, after a key load or with another address): a capture taken under another signature is not replayed
    const unsigned sig = 1u | (c->memo_top ? 2u : 0u) | ((unsigned)c->memo_alone << 2) | (c->side_begun ? 64u : 0u) |
                         (c->inv_id[0] == addr->id ? 128u : 0u) | (c->inv_id[1] == addr->id ? 256u : 0u);
    if (a->graph[which] && a->graph_sig[which] != sig) { hipGraphExecDestroy(a->graph[which]); a->graph[which] = nullptr; }
    if (!a->graph[which]) {
        hipGraph_t g = nullptr;
        HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        const int rc = enqueue();
        const hipError_t e = hipStreamEndCapture(c->stream, &g);
        if (rc != FHERAM_OK || e != hipSuccess) {
            if (g) hipGraphDestroy(g);
            return rc != FHERAM_OK ? rc : fail(c, FHERAM_ERR_DEVICE, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
        }
        const hipError_t e2 = hipGraphInstantiate(&a->graph[which], g, nullptr, nullptr, 0);
        hipGraphDestroy(g);
        if (e2 != hipSuccess) { a->graph[which] = nullptr; return fail(c, FHERAM_ERR_DEVICE, std::string("hipGraphInstantiate: ") + hipGetErrorString(e2)); }
        a->graph_sig[which] = sig;
    }
    HIPCHK(c, hipGraphLaunch(a->graph[which], c->stream));
    // the host-side bookkeeping the enqueue functions do (a replay does not run them)
    const int L0 = LOGN - ilog2_ceil(c->rows_glob);
    if (which == 0) { c->memo_top = false; c->d_last_res = c->d_res; }
    else if (which == 1) {
        c->memo_top = c->memo != 0;
        c->d_last_res = c->memo_top ? c->d_trtop : c->d_res;
        c->memo_alone = (c->memo && c->n2 == 2 && L0 > 0) ? L0 : 0;
        if (c->pre_inv && (long)c->rows * c->ws <= c->cus) for (int ci = 0; ci < c->n2; ci++) c->inv_id[ci] = addr->id;
    } else { c->memo_top = false; c->memo_alone = 0; c->side_begun = false; c->tree_rotate_pending = false; c->inv_id[0] = c->inv_id[1] = 0; }
    c->prep1_ready = false;
    return FHERAM_OK;
}

int check_common(fheram_ctx* c, const fheram_addr* addr) {
    // the single-launch mid chains, switched off because their launches kept giving up (launch.hpp fill_mid), are tried again
    // 256 ops later: a neighbour that held the CUs for a while does not cost the path its faster form for the context's life
    if (c && !c->mid && c->mid_saved && ++c->mid_off_ops >= 256) { c->mid = c->mid_saved; c->mid_saved = 0; c->mid_bad_windows = 0; c->mid_off_ops = 0; }
    if (!c) return FHERAM_ERR_INVALID_ARG;
    if (!addr || addr->ctx != c) return fail(c, FHERAM_ERR_INVALID_ARG, "address does not belong to this context (layout mismatch, ram.rs:404)");
    if (!c->initialized) return fail(c, FHERAM_ERR_UNINITIALIZED, "unitialized memory: self.data.len()=0");
    if (!c->keys_loaded) return fail(c, FHERAM_ERR_KEYS, "evaluation keys not loaded");
    return FHERAM_OK;
}

// SubRam::read (ram.rs:382-459) / SubRam::read_prepare_write (ram.rs:461-542) for all sub-RAMs at
// once, in two stages so that a row-sharded RAM can exchange between them.
// Stage 1 (every shard): coordinate-0 products on the local rows + the packing levels that stay
// inside the shard.  The packed GLWE of every sub-RAM is left where the last launch wrote it (*packed_out,
// indexed by sub-RAM); to_part also copies it into d_part (the buffer a sharded RAM exchanges).
int read_local(fheram_ctx* c, const fheram_addr* addr, bool prepare_write, GlweRef* packed_out, bool to_part) {
    c->wide = !prepare_write;   // read_prepare_write parks the gate wave beside its launches (read_top): its chain kernels keep a wave slot free
    if (prepare_write && c->pre_inv == 1 && !capturing(c) && c->wide_unsynced) {   // the gate wave may not be parked before this op's own launches start (ctx.hpp: ev_opstart)
        hipEventRecord(c->ev_opstart, c->stream);
        c->opstart_valid = true;
    }
    const long G = (long)fheram_ctx::GLWE;
    const long sy = (long)c->rows * G;
    const int ws = c->ws;
    const int R = (int)c->rows;
    GlweRef data = ref(c->d_data, sy, G), A = ref(c->d_scrA, sy, G), B = ref(c->d_scrB, sy, G);
    GlweRef part = ref(c->d_part, G, 0);
    const bool all = (c->n_shards == 1 && c->n2 == 2);   // unsharded: both coordinates now, one launch
    if (all) coordinate_prepare_all(c, addr); else coordinate_prepare(c, addr, 0);    // ram.rs:416-419 / 496-499
    c->prep1_ready = all;
    const int d0 = (int)c->base2d[0].size();
    if (c->n2 == 1) {
        GlweRef row0 = ref(c->d_data, sy, 0);
        if (prepare_write) {
            ep_chain(c, row0, row0, ref(c->d_scrA, sy, 0), prep_of(c, 0), d0, 1, ws);     // ram.rs:502-504 (rows == 1)
            *packed_out = row0;
            if (to_part) launch_copy(c, row0, part, 1, ws);
        } else {
            ep_chain(c, row0, part, ref(c->d_tmp, G, 0), prep_of(c, 0), d0, 1, ws);       // ram.rs:451
            *packed_out = part;
        }
        return FHERAM_OK;
    }
    int32_t* leaves;
    const int L0 = LOGN - ilog2_ceil(c->rows_glob);
    const bool keep = prepare_write && c->memo && L0 > 0;   // leaves = d_data then: arena A keeps the rows after their alone levels
    int32_t* packed;
    if (use_row_fuse(c, d0, L0, R, ws) && !(prepare_write && (d0 & 1))) {
        // the products and the alone packer levels as ONE launch (k_read_chain): rows after the alone levels in arena A
        // (read_prepare_write: the products' result also lands in the rows, ram.rs:502-504)
        launch_read_chain(c, data, prepare_write ? &data : nullptr, A, prep_of(c, 0), d0, L0, R, ws);
        packed = pack_levels(c, c->d_scrA, c->d_scrA, c->d_scrB, sy, G, (size_t)R, ws, 0, L0, keep, c->d_scrC, c->d_scrD);   // ram.rs:435-448 / 510-521: the pairing levels
    } else {
    if (prepare_write) {
        ep_chain(c, data, data, A, prep_of(c, 0), d0, R, ws);                             // ram.rs:502-504
        leaves = c->d_data;
    } else {
        ep_chain(c, data, A, B, prep_of(c, 0), d0, R, ws);                                // ram.rs:429-434
        leaves = c->d_scrA;
    }
    packed = pack_levels(c, leaves, c->d_scrA, c->d_scrB, sy, G, (size_t)R, ws, L0, L0, keep, c->d_scrC, c->d_scrD);   // ram.rs:435-448 / 510-521
    }
    c->memo_alone = keep ? L0 : 0;
    *packed_out = ref(packed, sy, 0);
    if (to_part) launch_copy(c, *packed_out, part, 1, ws);
    return FHERAM_OK;
}
// Stage 2 (root / unsharded): remaining packing levels over the shards' partials (`gathered`:
// [n_shards][ws] GLWEs, or nullptr when the RAM is not sharded and the packed rows are at `pk`),
// coordinate-1 products and the final trace.  Result left in d_res.  Every step is out of place, so
// nothing has to be copied between them.
int read_top(fheram_ctx* c, const fheram_addr* addr, bool prepare_write, int32_t* gathered, GlweRef pk) {
    c->wide = !prepare_write;
    const long G = (long)fheram_ctx::GLWE;
    const int ws = c->ws;
    GlweRef tmp = ref(c->d_tmp, G, 0), tree = ref(c->d_tree, G, 0);
    GlweRef last = pk;
    // coordinate 1's products inside the trace chain's launch (k_trace_tail's product steps): from two digits on (the fallback is the fused row chain)
    const int d1q = c->n2 == 2 ? (int)c->base2d[1].size() : 0;
    const bool fuse_ep = c->n2 == 2 && c->tail_ep && d1q >= 2 && d1q <= TAIL_EP_MAX && use_tail(c, LOGN, 1, ws);
    GlweRef ep_out = prepare_write ? tree : ref(c->d_tmp2, G, 0);
    if (c->n2 == 2) {
        if (gathered) {
            const int kG = ilog2_ceil((size_t)c->n_shards);
            int32_t* a0 = c->d_gat[1];   // gathered partials live in d_gat[0]
            int32_t* a1 = c->d_gat[2];
            int32_t* packed = pack_levels(c, gathered, a0, a1, G, (long)ws * G, (size_t)c->n_shards, ws, 0, LOGN - kG);
            pk = ref(packed, G, 0);
        }
        if (!c->prep1_ready) coordinate_prepare(c, addr, 1);
        c->prep1_ready = false;
        const int d1 = (int)c->base2d[1].size();
        if (fuse_ep) {
            last = ep_out;                                                            // (enqueued below, with the trace chain)
        } else if (prepare_write) {
            ep_chain(c, pk, tree, tmp, prep_of(c, 1), d1, 1, ws);                         // ram.rs:525-527 + 502-504 (i = 1): tree[0] <- rotated packed row
            last = tree;                                                              // ram.rs:535 (res <- tree[0])
        } else {
            GlweRef tmp2 = ref(c->d_tmp2, G, 0);
            ep_chain(c, pk, tmp2, tmp, prep_of(c, 1), d1, 1, ws);                         // ram.rs:454 (not into res: the trace below runs out of place)
            last = tmp2;
        }
    }                                                                                 // n2 == 1: res <- packed row (ram.rs:452 / 537)
    // read_prepare_write: the result is also what write_first_step computes first (trace of the same ciphertext,
    // ram.rs:571-572): it lands in d_trtop, which no read overwrites, and stays there for the write
    // the write's inverse digits, next to the trace chain below (one launch on half of the XCDs; the side stream has the
    // lowest priority, so that launch is placed first)
    // (only while the write's chains are one workgroup round on the chip: with several rounds — 2^21 on one GPU — the
    // earlier start of the write's main chain interleaves it with the side chain less favourably, write 8.57 -> 8.74 ms)
    const bool pre = prepare_write && c->pre_inv && (long)c->rows * c->ws <= c->cus;
    const bool gated = pre && c->pre_inv == 1 && !capturing(c) && use_tail(c, LOGN, 1, ws);   // FHERAM_PRE_INV=2: event fork (A/B switch)
    if (pre && !gated)
        for (int ci = c->n2 - 1; ci >= 0; ci--) precompute_inverse(c, addr, ci, ci == c->n2 - 1);   // coordinate 1 first: the write's head needs it first
    c->memo_top = prepare_write && c->memo;
    c->d_last_res = c->memo_top ? c->d_trtop : c->d_res;
    const uint64_t tl0 = c->tail_launches;
    GlweRef tb[2];
    if (fuse_ep && chain_bufs(LOGN, last, ref(c->d_last_res, G, 0), tmp, tb))          // ram.rs:454 / 525-527 + 457 / 540 as ONE launch
        launch_trace_tail(c, pk, tb, 0, LOGN, 1, ws, prep_of(c, 1), d1q, ep_out, prepare_write);
    else {
        if (fuse_ep) ep_chain(c, pk, ep_out, tmp, prep_of(c, 1), d1q, 1, ws);          // (cannot happen with these buffers; kept for safety)
        trace_steps(c, last, ref(c->d_last_res, G, 0), tmp, 0, LOGN, 1, ws);          // ram.rs:457 / 540
    }
    if (gated && !c->opstart_valid && c->wide_unsynced) {   // (a root's read_finish: no read_local of this op ran on this context)
        hipEventRecord(c->ev_opstart, c->stream);
        c->opstart_valid = true;
    }
    if (gated) {   // behind a gate that opens when the trace chain's launch is placed (no event on the main stream); host order is irrelevant
        const unsigned seq = c->tail_launches != tl0 ? c->tail_seq : 0;                // 0: no such launch after all -> event fork
        for (int ci = c->n2 - 1; ci >= 0; ci--) precompute_inverse(c, addr, ci, ci == c->n2 - 1, seq);
    }
    return FHERAM_OK;
}
int read_impl(fheram_ctx* c, const fheram_addr* addr, bool prepare_write) {
    GlweRef packed;
    int rc = read_local(c, addr, prepare_write, &packed, false);
    if (rc != FHERAM_OK) return rc;
    rc = read_top(c, addr, prepare_write, nullptr, packed);
    if (prepare_write && c->inv_id[0] == addr->id && capturing(c))                    // a capture ends with every fork joined
        for (int ci = 0; ci < c->n2; ci++) hipStreamWaitEvent(c->stream, c->ev_inv[ci], 0);
    if (rc == FHERAM_OK && prepare_write) c->state = true;                            // ram.rs:533
    return rc;
}

// Ram::write (ram.rs:226-294) in two stages.
// Stage 1 (root / unsharded): write_first_step on the top of the tree and, for n2 == 2, the inverse
// coordinate-1 products: leaves the un-rotated ct_lo of every sub-RAM in d_part.
int write_top(fheram_ctx* c, const fheram_addr* addr) {
    c->wide = true;             // (everything a write enqueues runs behind read_prepare_write's trace chain, whose placement releases the gate wave)
    const long G = (long)fheram_ctx::GLWE;
    const long sy = (long)c->rows * G;
    const int ws = c->ws;
    GlweRef wref = ref(c->d_w, G, 0), tmp = ref(c->d_tmp, G, 0), tmp2 = ref(c->d_tmp2, G, 0), tree = ref(c->d_tree, G, 0);
    // write_first_step (ram.rs:544-577): t <- normalize(t - trace(t) + w)
    GlweRef top = (c->n2 != 1) ? tree : ref(c->d_data, sy, 0);
    GlweRef tr = tmp;
    if (c->memo_top) tr = ref(c->d_trtop, G, 0);      // = trace(top), computed by read_prepare_write on this very ciphertext
    else trace_steps(c, top, tmp, tmp2, 0, LOGN, 1, ws);
    {
        ProfScope ps(c, "elementwise", ws);
        hipLaunchKernelGGL((k_sub_add_norm<3>), dim3(1, ws, EW_SLICES), dim3(256), 0, c->cur, top, tr, wref, top);
    }
    c->memo_top = false;
    if (c->n2 == 2) {
        if (c->inv_id[1] == addr->id) wait_inverse(c, c->stream, 1);                   // started by read_prepare_write
        else {
            // another address (or new keys): a precompute that read_prepare_write started for ITS address may still be
            // writing prep_inv_of(c, 1) on the side stream — the main stream must not overtake it
            if (c->inv_pending[1]) wait_inverse(c, c->stream, 1);
            coordinate_prepare_inv(c, addr, 1, c->d_ggsw_tmp, prep_inv_of(c, 1));     // ram.rs:260-271
        }
        c->inv_pending[1] = false;
        c->inv_id[1] = 0;
        ep_chain(c, tree, ref(c->d_part, G, 0), tmp, prep_inv_of(c, 1), (int)c->base2d[1].size(), 1, ws);   // ram.rs:610: the un-rotated ct_lo, in d_part
        // tree[0] <- ct_lo * X^-rows (ram.rs:629, `rows` rotations by X^-1): nothing in this write reads it again, so
        // the rotation is enqueued behind the rows' work (write_rows) instead of in front of it
        c->tree_rotate_pending = true;
    }
    return FHERAM_OK;
}
// Work of a write that does not depend on stage 1: tmp_a = trace(ct_hi) for every local row
// (ram.rs:616) and the inverse of coordinate 0 (ram.rs:278-289).  It is enqueued on the side stream
// so that it fills the CUs the latency-bound stage 1 (a chain of word_size-ciphertext launches)
// leaves idle.
void write_side_begin(fheram_ctx* c, const fheram_addr* addr) {
    c->wide = true;
    const long G = (long)fheram_ctx::GLWE;
    const long sy = (long)c->rows * G;
    hipEventRecord(c->ev_fork, c->stream);            // everything before this write (rows after rpw)
    hipStreamWaitEvent(c->stream2, c->ev_fork, 0);
    c->cur = c->stream2;
    if (c->n2 == 2) {
        c->d_trhi = c->d_scrA;
        if (c->memo_alone > 0) {   // arena A = the rows after trace steps 0 .. memo_alone-1 (left there by read_prepare_write)
            // ping-pong A <-> C; an odd number of remaining steps ends in C
            if ((LOGN - c->memo_alone) % 2 == 1) c->d_trhi = c->d_scrC;
            int32_t* tmp = c->d_trhi == c->d_scrA ? c->d_scrC : c->d_scrA;
            trace_steps(c, ref(c->d_scrA, sy, G), ref(c->d_trhi, sy, G), ref(tmp, sy, G), c->memo_alone, LOGN, (int)c->rows, c->ws);
        } else {
            trace_steps(c, ref(c->d_data, sy, G), ref(c->d_scrA, sy, G), ref(c->d_scrC, sy, G), 0, LOGN, (int)c->rows, c->ws);
        }
        c->memo_alone = 0;
    }
    if (c->inv_id[0] == addr->id) wait_inverse(c, c->stream2, 0);   // started by read_prepare_write (on this very stream)
    else coordinate_prepare_inv(c, addr, 0, c->d_ggsw_tmp2, prep_inv_of(c, 0));   // (a precompute for another address sits on this very stream: ordered)
    c->inv_id[0] = 0;
    c->inv_pending[0] = false;
    hipEventRecord(c->ev_join, c->stream2);
    c->cur = c->stream;
    c->side_begun = true;
}
// Error path of a write whose side-stream work was already enqueued: rejoin the side stream, so that no
// later operation on the main stream can overtake it.
void write_side_abort(fheram_ctx* c) {
    if (!c->side_begun) return;
    hipStreamWaitEvent(c->stream, c->ev_join, 0);
    c->side_begun = false;
}
// Stage 2 (every shard): write_mid_step on the local rows given ct_lo (in d_part), then write_last_step.
int write_rows(fheram_ctx* c, const fheram_addr* addr) {
    c->wide = true;
    const long G = (long)fheram_ctx::GLWE;
    const long sy = (long)c->rows * G;
    const int ws = c->ws, R = (int)c->rows;
    GlweRef data = ref(c->d_data, sy, G), A = ref(c->d_scrA, sy, G), B = ref(c->d_scrB, sy, G), D = ref(c->d_scrD, sy, G);
    GlweRef trhi = ref(c->d_trhi ? c->d_trhi : c->d_scrA, sy, G);
    const int d0 = (int)c->base2d[0].size();
    if (c->n2 == 2 && use_row_fuse(c, d0, LOGN, R, ws)) {
        // trace(ct_lo * X^-row), normalize(ct_hi - trace(ct_hi) + that) and write_last_step's products as ONE launch (k_write_chain);
        // it needs trace(ct_hi) and the inverse digits of coordinate 0 from the side stream at its start (that stream's chain holds
        // every CU until then anyway)
        hipStreamWaitEvent(c->stream, c->ev_join, 0);
        launch_write_chain(c, ref(c->d_part, G, 0), c->n_shards, c->shard, data, trhi, prep_inv_of(c, 0), d0, LOGN, R, ws);   // ram.rs:612-646
        if (c->tree_rotate_pending) {   // root / unsharded: the tree's copy of ct_lo, rotated (see write_top)
            ProfScope ps(c, "elementwise", ws);
            hipLaunchKernelGGL((k_rotate<3>), dim3(1, ws, EW_SLICES), dim3(256), 0, c->cur, ref(c->d_part, G, 0), ref(c->d_tree, G, 0), -(int)c->rows_glob);
            c->tree_rotate_pending = false;
        }
    } else {
    if (c->n2 == 2)
        trace_steps(c, ref(c->d_part, G, 0), B, D, 0, LOGN, R, ws, c->n_shards, c->shard);     // tmp_a = trace(ct_lo * X^-row)   ram.rs:621,629
    if (c->tree_rotate_pending) {   // root / unsharded: the tree's copy of ct_lo, rotated (see write_top)
        ProfScope ps(c, "elementwise", ws);
        hipLaunchKernelGGL((k_rotate<3>), dim3(1, ws, EW_SLICES), dim3(256), 0, c->cur, ref(c->d_part, G, 0), ref(c->d_tree, G, 0), -(int)c->rows_glob);
        c->tree_rotate_pending = false;
    }
    hipStreamWaitEvent(c->stream, c->ev_join, 0);                                              // side stream: trace(ct_hi), inverse coordinate 0
    if (c->n2 == 2) {
        ProfScope ps(c, "elementwise", (uint64_t)R * ws);
        hipLaunchKernelGGL((k_sub_add_norm<3>), dim3(R, ws, EW_SLICES), dim3(256), 0, c->cur, data, trhi, B, data);   // ram.rs:617,625-626
    }
    ep_chain(c, data, data, A, prep_inv_of(c, 0), d0, R, ws);                                         // ram.rs:644-646
    }
    // the next read_prepare_write's side work overwrites d_prep_inv: it is ordered behind this write by an event (the gate
    // launch in front of that work is time-bounded, so it cannot be the only ordering)
    if (!capturing(c)) { hipEventRecord(c->ev_wdone, c->stream); c->wdone_pending = true; }
    c->state = false;                                                                          // ram.rs:648
    c->side_begun = false;
    return FHERAM_OK;
}

}  // namespace
