// sla_xfer.cpp -- large host <-> device copies from / to the caller's PAGEABLE arrays (the drop-in boundary hands over plain host
// pointers: `fromListSM`'s triples, `SpVector`s, Data/Sparse/SpMatrix.hs:218-224).
//
// hipMemcpy on pageable memory stages through the runtime's own pinned buffers on the calling thread: 6-8 GB/s measured here
// (0.84 GB of canonical CSR arrays at 216^3: 0.11 s; the 4 GB of BASELINE config 3a: 0.5 s; an 80 MB vector: 10 ms) on a link that
// carries several times that.  The staging copy is the bottleneck and it is single-threaded, so this file does the staging itself, on
// several host threads: the copy is cut into 8 MiB chunks dealt round-robin to `lanes` workers, each with its own stream and two pinned
// slots (memcpy of chunk k + 2 overlaps the DMA of chunk k).  Synchronous semantics like hipMemcpy: on return the data is there.
//
// A copy can be CALLED OFF (`stop`): the workers stop taking chunks and the call reports the length of the prefix that is complete --
// the canonical CSR upload is abandoned that way when the matrix turns out to be value-indexed (sla_lower.cpp).
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <pthread.h>
#include <system_error>
#include <thread>
#include <vector>

#include "sla_internal.hpp"

namespace sla {

namespace {

constexpr size_t kXferSlot = (size_t)8 << 20;    // bytes per chunk / pinned slot
constexpr size_t kXferMin = (size_t)24 << 20;    // smaller copies: plain hipMemcpy
constexpr int kXferMaxLanes = 8;

struct XferLane {
    hipStream_t st = nullptr;
    void *buf[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool busy = false;
};
struct XferPool {
    std::mutex mu;
    std::condition_variable cv;
    XferLane lane[kXferMaxLanes];
    int ready = 0;          // lanes created so far
    bool broken = false;    // a lane could not be created: plain copies from now on
};

// one pool per device, created on first use and never torn down (a static destructor would run after the HIP runtime's own)
XferPool *pool_of(int device) {
    static std::mutex mu;
    static XferPool *pools[64] = {};
    if (device < 0 || device >= 64) return nullptr;
    std::lock_guard<std::mutex> g(mu);
    if (!pools[device]) pools[device] = new XferPool();
    return pools[device];
}

bool lane_create(XferLane &l) {
    bool ok = hipStreamCreateWithFlags(&l.st, hipStreamNonBlocking) == hipSuccess;
    if (!ok) l.st = nullptr;
    for (int s = 0; s < 2 && ok; ++s) {
        ok = hipHostMalloc(&l.buf[s], kXferSlot, hipHostMallocDefault) == hipSuccess;
        if (!ok) l.buf[s] = nullptr;
        if (ok) {
            ok = hipEventCreateWithFlags(&l.ev[s], hipEventDisableTiming) == hipSuccess;
            if (!ok) l.ev[s] = nullptr;
        }
    }
    if (!ok) {   // a lane that failed part-way gives back what it got: the next attempt starts from an empty slot
        for (int s = 0; s < 2; ++s) {
            if (l.ev[s]) { (void)hipEventDestroy(l.ev[s]); l.ev[s] = nullptr; }
            if (l.buf[s]) { (void)hipHostFree(l.buf[s]); l.buf[s] = nullptr; }
        }
        if (l.st) { (void)hipStreamDestroy(l.st); l.st = nullptr; }
    }
    return ok;
}

// up to `want` free lanes (at least one: waits for one if all are taken by other callers)
int lanes_acquire(XferPool *p, int want, int *ids) {
    std::unique_lock<std::mutex> g(p->mu);
    for (;;) {
        if (p->broken) return 0;
        int got = 0;
        for (int i = 0; i < p->ready && got < want; ++i)
            if (!p->lane[i].busy) ids[got++] = i;
        while (got < want && p->ready < kXferMaxLanes) {   // grow the pool on demand
            XferLane &l = p->lane[p->ready];
            if (!lane_create(l)) {
                (void)hipGetLastError();
                p->broken = got == 0 && p->ready == 0;      // (nothing works: give up for good; otherwise live with what there is)
                break;
            }
            ids[got++] = p->ready++;
        }
        if (got > 0) {
            for (int i = 0; i < got; ++i) p->lane[ids[i]].busy = true;
            return got;
        }
        if (p->broken) return 0;
        p->cv.wait(g);
    }
}
void lanes_release(XferPool *p, int n, const int *ids) {
    {
        std::lock_guard<std::mutex> g(p->mu);
        for (int i = 0; i < n; ++i) p->lane[ids[i]].busy = false;
    }
    p->cv.notify_all();
}

}  // namespace

// kind: hipMemcpyHostToDevice or hipMemcpyDeviceToHost.  *done (optional) = bytes of the leading part that is complete (= bytes unless
// the copy was called off).  Device-side ordering is the caller's: the buffers must not be in use by work still queued on a stream.
// stage (host-to-device only, optional): fills a pinned slot with the bytes [off, off + len) of the DEVICE image instead of a memcpy from
// src -- e.g. the narrowing of the caller's int64 column indices to the device's int32, done on the way instead of in a pass of its own.
// ordered: the buffers may still be in use by work queued on the context's stream (a new vector's zero fill, the kernels that wrote a
// vector): the copy goes behind it.  Lowering uploads into fresh allocations pass false and do not meet on that one stream.
hipError_t xfer_copy(sla_ctx *c, void *dst, const void *src, size_t bytes, hipMemcpyKind kind, const std::atomic<int> *stop, size_t *done,
                     xfer_stage_fn stage, const void *stage_ctx, bool ordered) {
    if (done) *done = 0;
    if (bytes == 0) return hipSuccess;
    auto stopped = [&] { return stop && stop->load(std::memory_order_relaxed) != 0; };
    XferPool *pool = (c->xfer && bytes >= kXferMin) ? pool_of(c->device) : nullptr;
    int ids[kXferMaxLanes];
    const size_t nchunks = (bytes + kXferSlot - 1) / kXferSlot;
    const int want = (int)std::min<size_t>((size_t)std::max(1, std::min(c->xfer_lanes, kXferMaxLanes)), nchunks / 2);
    const int L = pool ? lanes_acquire(pool, std::max(1, want), ids) : 0;
    if (L == 0) {   // plain copies, chunked when they can be called off (or have to be staged)
        const size_t step = (stop || stage) ? kXferSlot : bytes;
        std::vector<char> tmp(stage ? std::min(step, bytes) : 0);
        size_t off = 0;
        for (; off < bytes && !stopped(); off += step) {
            const size_t len = std::min(step, bytes - off);
            if (stage) stage(tmp.data(), off, len, stage_ctx);
            const void *from = stage ? (const void *)tmp.data() : (const void *)((const char *)src + off);
            hipError_t e;
            if (ordered) {   // on the context's stream, behind what it has queued for these buffers
                e = hipMemcpyAsync((char *)dst + off, from, len, kind, stream_of(c));
                if (e == hipSuccess) e = hipStreamSynchronize(stream_of(c));
            } else {
                e = hipMemcpy((char *)dst + off, from, len, kind);
            }
            if (e != hipSuccess) return e;
        }
        if (done) *done = std::min(off, bytes);
        return hipSuccess;
    }
    {   // the lanes have streams of their own: what the context's stream still has queued for these buffers (a vector's zero fill, the
        // kernels that wrote it) comes first
        const hipError_t e0 = ordered ? hipStreamSynchronize(stream_of(c)) : hipSuccess;
        if (e0 != hipSuccess) {
            lanes_release(pool, L, ids);
            return e0;
        }
    }
    std::vector<size_t> next((size_t)L, 0);          // first chunk index a lane has NOT completed
    std::vector<hipError_t> err((size_t)L, hipSuccess);
    auto work = [&](int li) {
        XferLane &l = pool->lane[ids[li]];
        hipError_t e = hipSetDevice(c->device);
        size_t k = 0;                                // chunks this lane has issued
        size_t pend[2] = {0, 0};                     // D2H: the chunk waiting in slot s
        bool has[2] = {false, false};
        size_t i = (size_t)li;
        for (; i < nchunks && e == hipSuccess && !stopped(); i += (size_t)L, ++k) {
            const int s = (int)(k & 1);
            const size_t off = i * kXferSlot, len = std::min(kXferSlot, bytes - off);
            if (kind == hipMemcpyHostToDevice) {
                if (k >= 2) e = hipEventSynchronize(l.ev[s]);                 // the DMA out of this slot two chunks ago
                if (e != hipSuccess) break;
                if (stage) stage(l.buf[s], off, len, stage_ctx);
                else memcpy(l.buf[s], (const char *)src + off, len);
                e = hipMemcpyAsync((char *)dst + off, l.buf[s], len, hipMemcpyHostToDevice, l.st);
                if (e == hipSuccess) e = hipEventRecord(l.ev[s], l.st);
            } else {
                if (has[s]) {                                                  // drain what sits in this slot first
                    e = hipEventSynchronize(l.ev[s]);
                    if (e != hipSuccess) break;
                    const size_t poff = pend[s] * kXferSlot;
                    memcpy((char *)dst + poff, l.buf[s], std::min(kXferSlot, bytes - poff));
                    has[s] = false;
                }
                e = hipMemcpyAsync(l.buf[s], (const char *)src + off, len, hipMemcpyDeviceToHost, l.st);
                if (e == hipSuccess) e = hipEventRecord(l.ev[s], l.st);
                pend[s] = i;
                has[s] = true;
            }
        }
        if (e == hipSuccess) e = hipStreamSynchronize(l.st);
        if (kind == hipMemcpyDeviceToHost && e == hipSuccess)
            for (int s = 0; s < 2; ++s)
                if (has[s]) {
                    const size_t poff = pend[s] * kXferSlot;
                    memcpy((char *)dst + poff, l.buf[s], std::min(kXferSlot, bytes - poff));
                }
        next[(size_t)li] = i;                        // (every chunk below i of this lane's residue class is complete)
        err[(size_t)li] = e;
    };
    std::vector<std::thread> th;
    th.reserve((size_t)L);
    int started = 1;                                 // lanes with a thread of their own (lane 0 is the caller's)
    try {
        for (; started < L; ++started) th.emplace_back(work, started);
    } catch (const std::system_error &) {            // thread limit: the lanes without a thread run here, one after the other
    }
    work(0);
    for (int li = started; li < L; ++li) work(li);
    for (auto &t : th) t.join();
    lanes_release(pool, L, ids);
    (void)hipSetDevice(c->device);
    size_t first_open = nchunks;                     // smallest chunk index some lane has not completed
    for (int li = 0; li < L; ++li) {
        if (err[(size_t)li] != hipSuccess) return err[(size_t)li];
        first_open = std::min(first_open, next[(size_t)li]);
    }
    if (done) *done = std::min(bytes, first_open * kXferSlot);
    return hipSuccess;
}

// The first large copy of a process paid for the lanes (streams, 8 x 8 MiB of pinned memory: ~20 ms); a new context builds them behind
// the caller's back instead (sla_api.cpp: ctx_create_common keeps the future).
void xfer_warm(int device, int lanes) {
    XferPool *pool = pool_of(device);
    if (!pool || hipSetDevice(device) != hipSuccess) return;
    int ids[kXferMaxLanes];
    const int L = lanes_acquire(pool, std::max(1, std::min(lanes, kXferMaxLanes)), ids);
    if (L > 0) lanes_release(pool, L, ids);
}

namespace {
struct BgGate {
    std::mutex mu;
    std::condition_variable cv;
    int inflight = 0;
};
BgGate *bg_gate() {
    static BgGate *g = new BgGate();   // never destroyed: threads may still touch it while statics are being torn down
    return g;
}
void bg_wait_all() {
    BgGate *g = bg_gate();
    std::unique_lock<std::mutex> lk(g->mu);
    g->cv.wait(lk, [g] { return g->inflight == 0; });
}
}  // namespace

void bg_begin() {
    BgGate *g = bg_gate();
    std::lock_guard<std::mutex> lk(g->mu);
    ++g->inflight;
}
void bg_end() {
    BgGate *g = bg_gate();
    {
        std::lock_guard<std::mutex> lk(g->mu);
        --g->inflight;
    }
    g->cv.notify_all();
}
// Exit handlers run in reverse order of registration: this one is registered after the HIP runtime initialised (and registered its own),
// so it runs first.
void bg_exit_handler_once() {
    static std::once_flag once;
    std::call_once(once, [] {
        (void)atexit(bg_wait_all);
        // a forked child has the count but not the threads: it would wait for ever at its own exit
        (void)pthread_atfork(nullptr, nullptr, [] { new (bg_gate()) BgGate(); });
    });
}

}  // namespace sla
