// Included by fheram.hip (same translation unit): ONE RAM over the GPUs of a node behind ONE handle — the native form of
// the row-sharded path for a host that is a single process (the reference's Ram is one object with one
// call per op: ram.rs:172-176,196-200,226-231).  fhe-ram_amd/sharded.py drives the same per-shard entry points from one
// process per GPU over RCCL; here one host thread per device drives its shard context, and the two exchange steps
//   read  : every shard's partial pack (word_size GLWEs, 96 KiB each as int32)  ->  the root's gather buffer
//   write : the root's un-rotated ct_lo (word_size GLWEs)                       ->  every shard
// are peer-to-peer copies on the producing context's stream (hipMemcpyPeerAsync over xGMI; 384 KiB per rank and op:
// latency-bound, nothing to bucket), ordered on the device by events.  The host threads only agree on WHEN an event has
// been recorded (a wait on an event that has not been recorded yet would be a no-op), through two atomics per op.
// Results are bit-identical to the unsharded path: every packer combine sees the operands of the sequential packer.
#pragma once
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include "path.hpp"

struct fheram_group_addr {
    fheram_group* grp;
    std::vector<fheram_addr*> a;   // one device address per shard context
};

struct fheram_group {
    struct Worker {
        std::thread th;
        std::mutex m;
        std::condition_variable cv;
        std::function<int()> job;
        bool has_job = false, done = true, quit = false;
        int rc = FHERAM_OK;
    };
    std::vector<fheram_ctx*> ctx;
    std::vector<int> dev;
    std::vector<std::unique_ptr<Worker>> w;
    std::vector<hipEvent_t> ev_part;   // [shard] recorded on the shard's stream after its partial has been copied to the root
    std::vector<hipEvent_t> ev_ctlo;   // [shard] recorded on the root's stream after ct_lo has been copied to the shard
    std::atomic<int> parts_recorded{0}, ctlo_recorded{0}, ready{0};
    std::atomic<bool> failed{false};
    // A group op that fails AFTER work has been enqueued leaves the shards inconsistent (rows half written, state flags and kept
    // by-products of read_prepare_write out of step): the group is poisoned and every further op fails fast until the rows are
    // uploaded again (fheram_group_ram_upload resets every shard).  Errors found BEFORE anything is enqueued (the reference's
    // asserts: wrong state, foreign address, missing keys) are checked on the calling thread for every shard and poison nothing.
    bool poisoned = false;
    std::vector<char> peer_direct;     // [shard] 1: peer access ENABLED in both directions between the shard's device and the root's (or the same device); the self-test copy went through either way (fheram_group_peer_info)
    std::vector<char> peer_enabled;    // [shard] result of hipDeviceEnablePeerAccess in both directions
    int root = 0;
    std::string err;

    int n() const { return (int)ctx.size(); }
};

namespace {

thread_local std::string g_group_err;

int gfail(fheram_group* g, int code, const std::string& msg) {
    if (g) g->err = msg; else g_group_err = msg;
    return code;
}
void worker_main(fheram_group::Worker* w, int device) {
    hipSetDevice(device);
    std::unique_lock<std::mutex> lk(w->m);
    for (;;) {
        w->cv.wait(lk, [&] { return w->has_job || w->quit; });
        if (w->quit) return;
        std::function<int()> job = std::move(w->job);
        w->has_job = false;
        lk.unlock();
        int rc;
        try { rc = job(); }                       // a job must never take the process down (bad_alloc in a staging vector, ...)
        catch (const std::bad_alloc&) { rc = FHERAM_ERR_DEVICE; }
        catch (...) { rc = FHERAM_ERR_DEVICE; }
        lk.lock();
        w->rc = rc;
        w->done = true;
        w->cv.notify_all();
    }
}
void post(fheram_group::Worker* w, std::function<int()> job) {
    std::lock_guard<std::mutex> lk(w->m);
    w->job = std::move(job);
    w->has_job = true;
    w->done = false;
    w->cv.notify_all();
}
int wait(fheram_group::Worker* w) {
    std::unique_lock<std::mutex> lk(w->m);
    w->cv.wait(lk, [&] { return w->done; });
    return w->rc;
}
// runs job(shard) on every worker and returns the first failure; the message of the failing context becomes the group's
int run_all(fheram_group* g, const std::function<int(int)>& job) {
    for (int i = 0; i < g->n(); i++) post(g->w[i].get(), [&job, i] { return job(i); });
    int rc = FHERAM_OK, who = -1;
    for (int i = 0; i < g->n(); i++) { const int r = wait(g->w[i].get()); if (r != FHERAM_OK && rc == FHERAM_OK) { rc = r; who = i; } }
    if (rc != FHERAM_OK) g->err = "shard " + std::to_string(who) + ": " + (g->ctx[who]->err.empty() ? std::string("job failed (exception / out of memory)") : g->ctx[who]->err);
    return rc;
}
// bounded host-side wait for `want` recordings (the other workers are enqueueing at this very moment); gives up when a
// peer failed, so that nobody waits for an event that will never be recorded
constexpr int GROUP_WAIT_S = 30;   // the peers only ENQUEUE before they count (milliseconds); 30 s means one of them is gone
bool await_count(fheram_group* g, std::atomic<int>& ctr, int want) {
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(GROUP_WAIT_S);
    for (long spin = 0; ctr.load(std::memory_order_acquire) < want; spin++) {
        if (g->failed.load(std::memory_order_acquire)) return false;
        if (spin > 64) std::this_thread::yield();
        if ((spin & 1023) == 1023 && std::chrono::steady_clock::now() > deadline) { g->failed.store(true, std::memory_order_release); return false; }
    }
    return !g->failed.load(std::memory_order_acquire);
}
// the reference's asserts, for every shard, on the calling thread, BEFORE anything is enqueued: a refused call changes nothing
int group_precheck(fheram_group* g, const fheram_group_addr* ga, bool want_state, const char* state_msg) {
    if (g->poisoned) return gfail(g, FHERAM_ERR_STATE, "the group is poisoned: an earlier op failed part-way and left the shards inconsistent; upload the rows again (fheram_group_ram_upload)");
    for (int i = 0; i < g->n(); i++) {
        fheram_ctx* c = g->ctx[i];
        int rc = check_common(c, ga->a[i]);
        if (rc == FHERAM_OK && c->state != want_state) rc = fail(c, FHERAM_ERR_STATE, state_msg);
        if (rc != FHERAM_OK) { g->err = "shard " + std::to_string(i) + ": " + c->err; return rc; }
    }
    return FHERAM_OK;
}
// an op failed after its jobs had started: see fheram_group::poisoned
int group_poison(fheram_group* g, int rc) {
    if (rc == FHERAM_OK) return rc;
    g->poisoned = true;
    int dev_before = 0;
    (void)hipGetDevice(&dev_before);
    for (fheram_ctx* c : g->ctx) { hipSetDevice(c->device); write_side_abort(c); }
    (void)hipSetDevice(dev_before);   // the caller's current device is the caller's
    g->err += " [group poisoned: upload the rows again]";
    return rc;
}
int check_group_addr(fheram_group* g, const fheram_group_addr* a) {
    if (!g) return FHERAM_ERR_INVALID_ARG;
    if (!a || a->grp != g || (int)a->a.size() != g->n()) return gfail(g, FHERAM_ERR_INVALID_ARG, "address does not belong to this group (layout mismatch, ram.rs:404)");
    return FHERAM_OK;
}

// SubRam::read / read_prepare_write over the group (ram.rs:382-542), one job per shard
int group_read_job(fheram_group* g, const fheram_group_addr* ga, bool prepare_write, int64_t* out, int i) {
    fheram_ctx* c = g->ctx[i];
    fheram_ctx* r = g->ctx[g->root];
    const fheram_addr* addr = ga->a[i];
    const size_t part = (size_t)c->ws * fheram_ctx::GLWE;
    auto bail = [&](int rc) { g->failed.store(true, std::memory_order_release); return rc; };
    int rc;
    if (hipSetDevice(c->device) != hipSuccess) return bail(fail(c, FHERAM_ERR_DEVICE, "hipSetDevice"));
    GlweRef packed;
    rc = read_local(c, addr, prepare_write, &packed, true);            // ... -> d_part
    if (rc != FHERAM_OK) return bail(rc);
    // (ram.rs:533: the state flag is committed by the caller once EVERY shard has come through)
    hipError_t e = hipMemcpyPeerAsync(r->d_gat[0] + (size_t)i * part, r->device, c->d_part, c->device, part * sizeof(int32_t), c->stream);
    if (e == hipSuccess) e = hipEventRecord(g->ev_part[i], c->stream);
    if (e != hipSuccess) return bail(fail(c, FHERAM_ERR_DEVICE, std::string("partial -> root: ") + hipGetErrorString(e)));
    g->parts_recorded.fetch_add(1, std::memory_order_release);
    if (i != g->root) return FHERAM_OK;
    // root: the one exchange step of a read has been enqueued by everybody -> finish (top packer levels, coordinate 1, trace)
    if (!await_count(g, g->parts_recorded, g->n())) return fail(c, FHERAM_ERR_DEVICE, "a shard failed (or did not arrive within 30 s) before the exchange");
    for (int k = 0; k < g->n(); k++)
        if (k != i && hipStreamWaitEvent(c->stream, g->ev_part[k], 0) != hipSuccess) return bail(fail(c, FHERAM_ERR_DEVICE, "hipStreamWaitEvent"));
    rc = read_top(c, addr, prepare_write, c->d_gat[0], ref(c->d_part, (long)fheram_ctx::GLWE, 0));
    if (rc != FHERAM_OK) return bail(rc);
    if (hipGetLastError() != hipSuccess) return bail(fail(c, FHERAM_ERR_DEVICE, "launch failure in read_top"));
    return out ? fheram_result_download(c, out) : fheram_sync(c);
}

// Ram::write over the group (ram.rs:226-294), one job per shard
int group_write_job(fheram_group* g, const fheram_group_addr* ga, int i) {
    fheram_ctx* c = g->ctx[i];
    fheram_ctx* r = g->ctx[g->root];
    const fheram_addr* addr = ga->a[i];
    const size_t part = (size_t)c->ws * fheram_ctx::GLWE;
    auto bail = [&](int rc) { g->failed.store(true, std::memory_order_release); return rc; };
    int rc;
    if (hipSetDevice(c->device) != hipSuccess) return bail(fail(c, FHERAM_ERR_DEVICE, "hipSetDevice"));
    if (!c->side_begun) write_side_begin(c, addr);                        // trace(ct_hi) of the local rows, inverse of coordinate 0: no ct_lo needed
    if (i == g->root) {
        rc = write_top(c, addr);                                          // write_first_step + inverse coordinate-1 products -> ct_lo in d_part
        if (rc != FHERAM_OK) return bail(rc);
        for (int k = 0; k < g->n(); k++) {                                // the one exchange step of a write
            if (k == i) continue;
            fheram_ctx* s = g->ctx[k];
            hipError_t e = hipMemcpyPeerAsync(s->d_part, s->device, c->d_part, c->device, part * sizeof(int32_t), c->stream);
            if (e == hipSuccess) e = hipEventRecord(g->ev_ctlo[k], c->stream);
            if (e != hipSuccess) return bail(fail(c, FHERAM_ERR_DEVICE, std::string("ct_lo -> shard: ") + hipGetErrorString(e)));
        }
        g->ctlo_recorded.store(1, std::memory_order_release);
    } else {
        if (!await_count(g, g->ctlo_recorded, 1)) { write_side_abort(c); return fail(c, FHERAM_ERR_DEVICE, "the root failed (or did not arrive within 30 s) before the exchange"); }
        if (hipStreamWaitEvent(c->stream, g->ev_ctlo[i], 0) != hipSuccess) { write_side_abort(c); return bail(fail(c, FHERAM_ERR_DEVICE, "hipStreamWaitEvent")); }
    }
    (void)r;
    // second rendezvous: nobody touches its rows before EVERY shard has its ct_lo on the way — a shard that failed up to here
    // leaves all rows as they were (the root included)
    g->ready.fetch_add(1, std::memory_order_release);
    if (!await_count(g, g->ready, g->n())) { write_side_abort(c); return fail(c, FHERAM_ERR_DEVICE, "a shard failed before the rows were written: the rows are unchanged, but the tree top and the write state have been consumed — upload the RAM again"); }
    rc = write_rows(c, addr);                                             // write_mid_step on the local rows, write_last_step
    if (rc != FHERAM_OK) return bail(rc);
    if (hipGetLastError() != hipSuccess) return bail(fail(c, FHERAM_ERR_DEVICE, "launch failure in write_rows"));
    return fheram_sync(c);                                                // the op is complete when every shard's rows are
}

}  // namespace

extern "C" {

const char* fheram_group_last_error(const fheram_group* g) { return g ? g->err.c_str() : g_group_err.c_str(); }
int fheram_group_size(const fheram_group* g) { return g ? g->n() : 0; }
fheram_ctx* fheram_group_ctx(fheram_group* g, int shard) { return (g && shard >= 0 && shard < g->n()) ? g->ctx[shard] : nullptr; }

void fheram_group_destroy(fheram_group* g) {
    if (!g) return;
    for (auto& w : g->w) {
        if (!w) continue;
        { std::lock_guard<std::mutex> lk(w->m); w->quit = true; w->cv.notify_all(); }
        if (w->th.joinable()) w->th.join();
    }
    for (size_t i = 0; i < g->ctx.size(); i++) {
        if (g->ctx[i]) hipSetDevice(g->ctx[i]->device);
        if (i < g->ev_part.size() && g->ev_part[i]) hipEventDestroy(g->ev_part[i]);
        if (i < g->ev_ctlo.size() && g->ev_ctlo[i]) hipEventDestroy(g->ev_ctlo[i]);
        fheram_ctx_destroy(g->ctx[i]);
    }
    delete g;
}

int fheram_group_create(const fheram_params* p, const int* devices, int n_devices, fheram_group** out) {
    if (!p || !devices || !out || n_devices < 1) return gfail(nullptr, FHERAM_ERR_INVALID_ARG, "null argument / no device");
    *out = nullptr;
    // context creation, event creation, peer enabling and the self-test all move the calling thread between devices: the caller's
    // device is read FIRST and restored on every return path
    struct DeviceRestore {
        int dev = -1;
        DeviceRestore() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
        ~DeviceRestore() { if (dev >= 0) (void)hipSetDevice(dev); }
    } restore_device;
    fheram_group* g = new fheram_group();
    for (int i = 0; i < n_devices; i++) {
        fheram_ctx* c = nullptr;
        const int rc = fheram_ctx_create_sharded(p, devices[i], i, n_devices, &c);
        if (rc != FHERAM_OK) { g_group_err = std::string("shard ") + std::to_string(i) + ": " + fheram_last_error(nullptr); fheram_group_destroy(g); return rc; }
        g->ctx.push_back(c);
        g->dev.push_back(devices[i]);
    }
    g->ev_part.assign(n_devices, nullptr);
    g->ev_ctlo.assign(n_devices, nullptr);
    g->peer_enabled.assign(n_devices, 0);
    for (int i = 0; i < n_devices; i++) {
        // ev_part[i] lives on shard i's device (recorded there), ev_ctlo[i] on the root's (recorded on the root's stream)
        hipError_t e = hipSetDevice(devices[i]);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&g->ev_part[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipSetDevice(devices[g->root]);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&g->ev_ctlo[i], hipEventDisableTiming);
        if (e != hipSuccess) { g_group_err = std::string("hipEventCreate: ") + hipGetErrorString(e); fheram_group_destroy(g); return FHERAM_ERR_DEVICE; }
        // direct peer access root <-> shard where the topology offers it (xGMI); without it hipMemcpyPeerAsync stages
        if (devices[i] != devices[g->root]) {
            int can = 0, can_back = 0;
            auto enabled = [](hipError_t e) { return e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled; };
            if (hipDeviceCanAccessPeer(&can, devices[g->root], devices[i]) == hipSuccess && can &&
                hipDeviceCanAccessPeer(&can_back, devices[i], devices[g->root]) == hipSuccess && can_back) {
                hipSetDevice(devices[g->root]); const bool a = enabled(hipDeviceEnablePeerAccess(devices[i], 0));
                hipSetDevice(devices[i]); const bool b = enabled(hipDeviceEnablePeerAccess(devices[g->root], 0));
                (void)hipGetLastError();
                g->peer_enabled[i] = (a && b) ? 1 : 0;   // what fheram_group_peer_info reports: access ENABLED in both directions
            }
        }
    }
    // self-test of the exchange path: one copy shard -> root and one root -> shard per shard, checked word for word.  A pair that
    // cannot copy (no peer path, IOMMU / IPC restrictions) fails HERE, with the pair named, not in the middle of the first read.
    g->peer_direct.assign(n_devices, 1);
    {
        fheram_ctx* r = g->ctx[g->root];
        const size_t n_probe = 256;
        std::vector<int32_t> pat(n_probe), back(n_probe);
        for (int i = 0; i < n_devices; i++) {
            fheram_ctx* c = g->ctx[i];
            if (c == r) continue;
            g->peer_direct[i] = (devices[i] == devices[g->root]) ? 1 : g->peer_enabled[i];
            for (size_t k = 0; k < n_probe; k++) pat[k] = (int32_t)(0x5EED0000u + (unsigned)i * 4096u + (unsigned)k);
            hipError_t e = hipSetDevice(c->device);
            if (e == hipSuccess) e = hipMemcpy(c->d_part, pat.data(), n_probe * 4, hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpyPeerAsync(r->d_gat[0], r->device, c->d_part, c->device, n_probe * 4, c->stream);      // shard -> root (read)
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e == hipSuccess) e = hipSetDevice(r->device);
            if (e == hipSuccess) e = hipMemcpyPeerAsync(c->d_part + n_probe, c->device, r->d_gat[0], r->device, n_probe * 4, r->stream);  // root -> shard (write)
            if (e == hipSuccess) e = hipStreamSynchronize(r->stream);
            if (e == hipSuccess) e = hipSetDevice(c->device);
            if (e == hipSuccess) e = hipMemcpy(back.data(), c->d_part + n_probe, n_probe * 4, hipMemcpyDeviceToHost);
            if (e != hipSuccess || back != pat) {
                g_group_err = "peer copy self-test failed between device " + std::to_string(devices[i]) + " (shard " + std::to_string(i) + ") and the root's device " +
                              std::to_string(devices[g->root]) + (e != hipSuccess ? std::string(": ") + hipGetErrorString(e) : std::string(": data mismatch"));
                fheram_group_destroy(g);
                return FHERAM_ERR_DEVICE;
            }
        }
    }
    for (int i = 0; i < n_devices; i++) {
        g->w.emplace_back(new fheram_group::Worker());
        g->w.back()->th = std::thread(worker_main, g->w.back().get(), devices[i]);
    }
    *out = g;
    return FHERAM_OK;
}
/* per shard: 1 = the exchange copies between the shard's device and the root's go peer to peer (or are on one device), 0 = the
 * runtime stages them (hipDeviceCanAccessPeer says no); the self-test copy of fheram_group_create went through either way */
int fheram_group_peer_info(const fheram_group* g, int* direct, int n) {
    if (!g || !direct || n < g->n()) return FHERAM_ERR_INVALID_ARG;
    for (int i = 0; i < g->n(); i++) direct[i] = g->peer_direct[i];
    return FHERAM_OK;
}
int fheram_group_poisoned(const fheram_group* g) { return (g && g->poisoned) ? 1 : 0; }
/* max over the shards' round-off monitors (fheram_roundoff_max); FHERAM_ERR_PRECISION if any shard is above the limit */
int fheram_group_roundoff_max(fheram_group* g, double* out) {
    if (!g || !out) return FHERAM_ERR_INVALID_ARG;
    double m = 0.0;
    int rc = FHERAM_OK;
    for (int i = 0; i < g->n(); i++) {
        double v = 0.0;
        const int r = fheram_roundoff_max(g->ctx[i], &v);
        if (r != FHERAM_OK && r != FHERAM_ERR_PRECISION) return gfail(g, r, std::string("shard ") + std::to_string(i) + ": " + fheram_last_error(g->ctx[i]));
        if (r == FHERAM_ERR_PRECISION) rc = r;
        m = v > m ? v : m;
    }
    *out = m;
    return rc == FHERAM_OK ? rc : gfail(g, rc, "FP64 round-off above the limit on a shard: results are not trustworthy");
}

/* EvaluationKeysPrepared::prepare on every shard (keys are replicated; keys.rs:57-71) */
int fheram_group_keys_load(fheram_group* g, const int64_t* gal_els, int n_gal, const int64_t* const* atk_glwe,
                           const int64_t* atk_ggsw_inv, int64_t atk_ggsw_inv_p, const int64_t* tsk) {
    if (!g) return FHERAM_ERR_INVALID_ARG;
    return run_all(g, [&](int i) { return fheram_keys_load(g->ctx[i], gal_els, n_gal, atk_glwe, atk_ggsw_inv, atk_ggsw_inv_p, tsk); });
}

/* rows: the WHOLE RAM, [word_size][rows][GLWE]; shard i takes rows i, i + n, i + 2n, ... of every sub-RAM */
int fheram_group_ram_upload(fheram_group* g, const int64_t* rows) {
    if (!g) return FHERAM_ERR_INVALID_ARG;
    if (!rows) return gfail(g, FHERAM_ERR_INVALID_ARG, "null rows");
    const size_t G = fheram_ctx::GLWE;
    const int rc = run_all(g, [&](int i) -> int {
        fheram_ctx* c = g->ctx[i];
        std::vector<int64_t> mine((size_t)c->ws * c->rows * G);
        for (int y = 0; y < c->ws; y++)
            for (size_t x = 0; x < c->rows; x++)
                std::memcpy(&mine[((size_t)y * c->rows + x) * G], rows + ((size_t)y * c->rows_glob + (size_t)i + x * g->n()) * G, G * sizeof(int64_t));
        write_side_abort(c);
        return fheram_ram_upload(c, mine.data());     // resets state and the kept by-products of read_prepare_write
    });
    if (rc == FHERAM_OK) g->poisoned = false;         // every shard holds consistent rows again
    return rc;
}
int fheram_group_ram_download(fheram_group* g, int64_t* rows) {
    if (!g) return FHERAM_ERR_INVALID_ARG;
    if (!rows) return gfail(g, FHERAM_ERR_INVALID_ARG, "null rows");
    const size_t G = fheram_ctx::GLWE;
    return run_all(g, [&](int i) -> int {
        fheram_ctx* c = g->ctx[i];
        std::vector<int64_t> mine((size_t)c->ws * c->rows * G);
        const int rc = fheram_ram_download(c, mine.data());
        if (rc != FHERAM_OK) return rc;
        for (int y = 0; y < c->ws; y++)
            for (size_t x = 0; x < c->rows; x++)
                std::memcpy(rows + ((size_t)y * c->rows_glob + (size_t)i + x * g->n()) * G, &mine[((size_t)y * c->rows + x) * G], G * sizeof(int64_t));
        return FHERAM_OK;
    });
}
int fheram_group_ram_tree_download(fheram_group* g, int level, int64_t* out) {
    if (!g) return FHERAM_ERR_INVALID_ARG;
    const int rc = fheram_ram_tree_download(g->ctx[g->root], level, out);
    if (rc != FHERAM_OK) g->err = g->ctx[g->root]->err;
    return rc;
}
int fheram_group_ram_state(const fheram_group* g) { return g ? fheram_ram_state(g->ctx[g->root]) : 0; }

/* Address digits are replicated on every shard (address.rs:21-24) */
int fheram_group_address_create(fheram_group* g, const int64_t* const* ggsw, int n_ggsw, fheram_group_addr** out) {
    if (!g || !out) return FHERAM_ERR_INVALID_ARG;
    *out = nullptr;
    fheram_group_addr* ga = new fheram_group_addr{g, std::vector<fheram_addr*>(g->n(), nullptr)};
    const int rc = run_all(g, [&](int i) { return fheram_address_create(g->ctx[i], ggsw, n_ggsw, &ga->a[i]); });
    if (rc != FHERAM_OK) { fheram_group_address_destroy(ga); return rc; }
    *out = ga;
    return FHERAM_OK;
}
void fheram_group_address_destroy(fheram_group_addr* ga) {
    if (!ga) return;
    for (fheram_addr* a : ga->a) fheram_address_destroy(a);
    delete ga;
}

int fheram_group_word_stage(fheram_group* g, const int64_t* w, int n_w) {
    if (!g) return FHERAM_ERR_INVALID_ARG;
    const int rc = fheram_word_stage(g->ctx[g->root], w, n_w);
    if (rc != FHERAM_OK) g->err = g->ctx[g->root]->err;
    return rc;
}

/* Ram::read (ram.rs:172-191) */
int fheram_group_read(fheram_group* g, const fheram_group_addr* addr, int64_t* out) {
    int rc = check_group_addr(g, addr);
    if (rc != FHERAM_OK) return rc;
    if (g->n() == 1) { rc = fheram_read(g->ctx[0], addr->a[0], out); if (rc != FHERAM_OK) g->err = g->ctx[0]->err; else if (!out) rc = fheram_sync(g->ctx[0]); return rc; }
    rc = group_precheck(g, addr, false, "invalid call to Memory.read: internal state is true -> requires calling Memory.write");
    if (rc != FHERAM_OK) return rc;
    g->parts_recorded.store(0); g->failed.store(false);
    return group_poison(g, run_all(g, [&](int i) { return group_read_job(g, addr, false, out, i); }));
}
/* Ram::read_prepare_write (ram.rs:196-222) */
int fheram_group_read_prepare_write(fheram_group* g, const fheram_group_addr* addr, int64_t* out) {
    int rc = check_group_addr(g, addr);
    if (rc != FHERAM_OK) return rc;
    if (g->n() == 1) { rc = fheram_read_prepare_write(g->ctx[0], addr->a[0], out); if (rc != FHERAM_OK) g->err = g->ctx[0]->err; else if (!out) rc = fheram_sync(g->ctx[0]); return rc; }
    rc = group_precheck(g, addr, false, "invalid call to Memory.read: internal state is true -> requires calling Memory.write");
    if (rc != FHERAM_OK) return rc;
    g->parts_recorded.store(0); g->failed.store(false);
    rc = group_poison(g, run_all(g, [&](int i) { return group_read_job(g, addr, true, out, i); }));
    if (rc == FHERAM_OK) for (fheram_ctx* c : g->ctx) c->state = true;      // ram.rs:533, on every shard or on none
    return rc;
}
/* Ram::write (ram.rs:226-294); w == NULL uses the words staged by fheram_group_word_stage */
int fheram_group_write(fheram_group* g, const int64_t* w, int n_w, const fheram_group_addr* addr) {
    int rc = check_group_addr(g, addr);
    if (rc != FHERAM_OK) return rc;
    fheram_ctx* r = g->ctx[g->root];
    if (n_w != r->ws) return gfail(g, FHERAM_ERR_INVALID_ARG, "w.len() != subrams.len() (ram.rs:243)");
    if (g->n() == 1) { rc = fheram_write(r, w, n_w, addr->a[0]); if (rc != FHERAM_OK) g->err = r->err; else rc = fheram_sync(r); return rc; }
    rc = group_precheck(g, addr, true, "invalid call to Memory.write: internal state is false -> requires calling Memory.read_prepare_write");
    if (rc != FHERAM_OK) return rc;
    if (w) { rc = fheram_word_stage(r, w, n_w); if (rc != FHERAM_OK) { g->err = r->err; return rc; } }
    else if (!r->words_staged) return gfail(g, FHERAM_ERR_INVALID_ARG, "w == NULL and no staged words");
    g->ctlo_recorded.store(0); g->ready.store(0); g->failed.store(false);
    return group_poison(g, run_all(g, [&](int i) { return group_write_job(g, addr, i); }));
}
int fheram_group_result_download(fheram_group* g, int64_t* out) {
    if (!g) return FHERAM_ERR_INVALID_ARG;
    const int rc = fheram_result_download(g->ctx[g->root], out);
    if (rc != FHERAM_OK) g->err = g->ctx[g->root]->err;
    return rc;
}

}  // extern "C"
