// mdc_stream.cpp -- a list of files (plain or gzip) through one parser, with the host work of the
// next files overlapped with the GPU work on the current one.
//
//   worker threads     read a file, inflate it if it is gzip, into a pinned host buffer
//   copy stream        uploads file k+1 (hipMemcpyAsync from the pinned buffer) ...
//   parser's stream    ... while file k is being parsed
//
// Replaces the per-file loop of the reference's ingest (tools/xmltodb.cpp:147-218, one file at a
// time: open, read, parse) for inputs that tools/gzipfiles.cpp:59 has compressed.  Built on the
// public entry points of mdc_parse.h plus the HIP runtime.  Kernel launches, copies and stream
// waits are made on the thread that calls mdc_stream_next; the worker threads bind the parser's
// device (hipSetDevice) only to allocate their pinned buffers (hipHostMalloc; kept for the next
// stream when this one is closed) and touch no stream.
#include <hip/hip_runtime.h>
#include <zlib.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "mdc_parse.h"

extern "C" int mdc_internal_fail(int code, const char *msg);  // sets the calling thread's last error
extern "C" int mdc_internal_parser_device(const mdc_parser *p);

namespace {

enum SlotState { EMPTY, FILLING, READY };

struct Slot {
    SlotState state = EMPTY;
    int index = -1;
    uint8_t *h = nullptr;  // pinned
    size_t cap = 0, n = 0;
    int rc = MDC_OK;
    std::string err;
    bool uploading = false;
};

// Pinning host memory costs about as much as reading a file into it, so the buffers of a closed
// stream are kept (up to POOL_MAX_BYTES per process) for the next one: an ingest opens a stream per
// directory scan (tools/xmltodb.cpp:147-218 runs that loop every few seconds).
struct PinnedBuf {
    uint8_t *p;
    size_t cap;
};
std::mutex g_pool_mu;
std::vector<PinnedBuf> g_pool;
size_t g_pool_bytes = 0;
constexpr size_t POOL_MAX_BYTES = (size_t)2 << 30;

// a pinned buffer of at least `want` bytes: the smallest kept one that fits, else a new one
bool pinned_get(size_t want, PinnedBuf &out) {
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        size_t best = g_pool.size();
        for (size_t i = 0; i < g_pool.size(); ++i)
            if (g_pool[i].cap >= want && (best == g_pool.size() || g_pool[i].cap < g_pool[best].cap)) best = i;
        if (best < g_pool.size() && g_pool[best].cap <= 2 * want + ((size_t)8 << 20)) {
            out = g_pool[best];
            g_pool_bytes -= out.cap;
            g_pool.erase(g_pool.begin() + (long)best);
            return true;
        }
    }
    const size_t cap = (want + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    uint8_t *q = nullptr;
    if (hipHostMalloc((void **)&q, cap, hipHostMallocPortable) != hipSuccess) return false;
    out.p = q;
    out.cap = cap;
    return true;
}

void pinned_put(uint8_t *p, size_t cap) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        if (g_pool_bytes + cap <= POOL_MAX_BYTES) {
            g_pool.push_back({p, cap});
            g_pool_bytes += cap;
            return;
        }
    }
    (void)hipHostFree(p);
}

// grows a pinned buffer, keeping its first `keep` bytes
bool pinned_reserve(Slot &s, size_t want, size_t keep) {
    if (want <= s.cap) return true;
    if (s.cap && want < s.cap + s.cap / 2) want = s.cap + s.cap / 2;  // (a file that grows while it is read)
    PinnedBuf b;
    if (!pinned_get(want, b)) return false;
    if (keep) memcpy(b.p, s.h, keep);
    pinned_put(s.h, s.cap);
    s.h = b.p;
    s.cap = b.cap;
    return true;
}

// Reads `path` into the slot; a gzip file (magic 1f 8b) is inflated, concatenated members
// included.  Sets rc/err on failure.
void fill_slot(Slot &s, const char *path) {
    s.n = 0;
    s.rc = MDC_OK;
    s.err.clear();
    FILE *f = fopen(path, "rb");
    if (!f) {
        s.rc = MDC_FILE_OPEN_FAILED;
        s.err = std::string("open(") + path + ") failed";
        return;
    }
    unsigned char magic[2] = {0, 0};
    size_t got = fread(magic, 1, 2, f);
    const bool gz = got == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
    fseek(f, 0, SEEK_END);
    const long fsize = ftell(f);
    fseek(f, 0, SEEK_SET);
    auto io_error = [&](const char *what) {
        s.rc = MDC_ERR_IO;
        s.err = std::string(what) + "(" + path + ") failed";
        fclose(f);
    };
    if (!gz) {
        if (!pinned_reserve(s, (size_t)(fsize > 0 ? fsize : 0) + 64, 0)) return io_error("pinned allocation for");
        for (;;) {
            if (s.n == s.cap && !pinned_reserve(s, s.cap * 2, s.n)) return io_error("pinned allocation for");
            got = fread(s.h + s.n, 1, s.cap - s.n, f);
            if (got == 0) break;
            s.n += got;
        }
        if (ferror(f)) return io_error("read");
        fclose(f);
        return;
    }
    // gzip: the last four bytes hold the uncompressed size modulo 2^32 (of the last member)
    size_t guess = (size_t)(fsize > 0 ? fsize : 0) * 4;
    if (fsize >= 18) {
        unsigned char tail[4];
        fseek(f, -4, SEEK_END);
        if (fread(tail, 1, 4, f) == 4) {
            const size_t isize = (size_t)tail[0] | ((size_t)tail[1] << 8) | ((size_t)tail[2] << 16) | ((size_t)tail[3] << 24);
            if (isize > guess / 8) guess = isize;
        }
        fseek(f, 0, SEEK_SET);
    }
    if (!pinned_reserve(s, guess + 64, 0)) return io_error("pinned allocation for");
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, 15 + 16) != Z_OK) return io_error("inflateInit");
    std::vector<unsigned char> in((size_t)1 << 20);
    bool member_done = false;  // the stream stands behind a complete member
    for (;;) {
        zs.avail_in = (uInt)fread(in.data(), 1, in.size(), f);
        if (ferror(f)) {
            inflateEnd(&zs);
            return io_error("read");
        }
        if (zs.avail_in == 0) break;
        zs.next_in = in.data();
        while (zs.avail_in > 0) {
            if (member_done) {  // another member follows
                if (inflateReset(&zs) != Z_OK) {
                    inflateEnd(&zs);
                    return io_error("inflateReset");
                }
                member_done = false;
            }
            if (s.n == s.cap && !pinned_reserve(s, s.cap * 2, s.n)) {
                inflateEnd(&zs);
                return io_error("pinned allocation for");
            }
            const size_t room = s.cap - s.n;
            zs.next_out = s.h + s.n;
            zs.avail_out = (uInt)(room > 0x40000000u ? 0x40000000u : room);
            const uInt before = zs.avail_out;
            const int zr = inflate(&zs, Z_NO_FLUSH);
            s.n += before - zs.avail_out;
            if (zr == Z_STREAM_END) {
                member_done = true;
            } else if (zr != Z_OK && zr != Z_BUF_ERROR) {
                inflateEnd(&zs);
                return io_error("inflate");
            }
        }
    }
    inflateEnd(&zs);
    if (!member_done) return io_error("inflate (truncated file)");
    fclose(f);
}

}  // namespace

struct mdc_stream {
    mdc_parser *parser = nullptr;
    int device = 0;
    std::vector<std::string> paths;
    std::vector<Slot> slots;
    std::mutex mu;
    std::condition_variable cv;
    int next_fill = 0;      // next file a worker takes
    int next_deliver = 0;   // next file mdc_stream_next hands out
    bool stop = false;
    std::vector<std::thread> workers;
    hipStream_t copy = nullptr;
    void *dbuf[2] = {nullptr, nullptr};
    size_t dcap[2] = {0, 0};
    hipEvent_t up[2] = {nullptr, nullptr};
    int uploaded = -1;      // highest file index whose upload has been enqueued
    uint64_t bytes = 0;     // uncompressed bytes handed to the parser so far
};

static void worker_main(mdc_stream *s) {
    (void)hipSetDevice(s->device);
    for (;;) {
        int k;
        Slot *slot;
        {
            std::unique_lock<std::mutex> lk(s->mu);
            // files are taken in order; file k needs the slot that file k - depth has released
            s->cv.wait(lk, [&] {
                return s->stop || s->next_fill >= (int)s->paths.size() ||
                       s->slots[(size_t)s->next_fill % s->slots.size()].state == EMPTY;
            });
            if (s->stop || s->next_fill >= (int)s->paths.size()) return;
            k = s->next_fill++;
            slot = &s->slots[(size_t)k % s->slots.size()];
            slot->state = FILLING;
            slot->index = k;
        }
        fill_slot(*slot, s->paths[(size_t)k].c_str());
        {
            std::lock_guard<std::mutex> lk(s->mu);
            slot->state = READY;
        }
        s->cv.notify_all();
    }
}

extern "C" int mdc_stream_open(mdc_parser *p, const char *const *paths, int nfiles, int threads, mdc_stream **out) {
    if (!p || !out || nfiles < 0 || (nfiles > 0 && !paths)) return mdc_internal_fail(MDC_ERR_ARG, "NULL argument");
    *out = nullptr;
    for (int i = 0; i < nfiles; ++i)
        if (!paths[i]) return mdc_internal_fail(MDC_ERR_ARG, "NULL path");
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    mdc_stream *s = new mdc_stream();
    s->parser = p;
    s->device = mdc_internal_parser_device(p);
    for (int i = 0; i < nfiles; ++i) s->paths.emplace_back(paths[i]);
    s->slots.resize((size_t)threads + 1);
    if (hipSetDevice(s->device) != hipSuccess || hipStreamCreateWithFlags(&s->copy, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&s->up[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s->up[1], hipEventDisableTiming) != hipSuccess) {
        if (s->copy) (void)hipStreamDestroy(s->copy);
        for (auto e : s->up)
            if (e) (void)hipEventDestroy(e);
        delete s;
        return mdc_internal_fail(MDC_ERR_HIP, "cannot create the copy stream");
    }
    for (int t = 0; t < threads; ++t) s->workers.emplace_back(worker_main, s);
    *out = s;
    return MDC_OK;
}

// enqueues the upload of file k (its slot is READY and holds no error)
static int start_upload(mdc_stream *s, int k) {
    Slot &slot = s->slots[(size_t)k % s->slots.size()];
    const int b = k & 1;
    if (slot.n + 64 > s->dcap[b]) {
        if (s->dbuf[b]) (void)mdc_device_free(s->device, s->dbuf[b]);
        s->dbuf[b] = nullptr;
        s->dcap[b] = 0;
        size_t want = slot.n + slot.n / 8 + 4096;
        int rc = mdc_device_alloc(s->device, want, &s->dbuf[b]);
        if (rc) return rc;
        s->dcap[b] = want;
    }
    if (slot.n) {
        hipError_t e = hipMemcpyAsync(s->dbuf[b], slot.h, slot.n, hipMemcpyHostToDevice, s->copy);
        if (e != hipSuccess) return mdc_internal_fail(MDC_ERR_HIP, hipGetErrorString(e));
    }
    if (hipEventRecord(s->up[b], s->copy) != hipSuccess) return mdc_internal_fail(MDC_ERR_HIP, "hipEventRecord");
    slot.uploading = true;
    s->uploaded = k;
    return MDC_OK;
}

extern "C" int mdc_stream_next(mdc_stream *s, mdc_batch **out, int *index) {
    if (!s || !out) return mdc_internal_fail(MDC_ERR_ARG, "NULL argument");
    *out = nullptr;
    const int k = s->next_deliver;
    if (index) *index = k;
    if (k >= (int)s->paths.size()) return MDC_STREAM_END;
    if (hipSetDevice(s->device) != hipSuccess) return mdc_internal_fail(MDC_ERR_HIP, "hipSetDevice");
    Slot &slot = s->slots[(size_t)k % s->slots.size()];
    {
        std::unique_lock<std::mutex> lk(s->mu);
        s->cv.wait(lk, [&] { return slot.state == READY && slot.index == k; });
    }
    auto release = [&] {
        {
            std::lock_guard<std::mutex> lk(s->mu);
            slot.state = EMPTY;
            slot.uploading = false;
            slot.index = -1;
        }
        s->next_deliver = k + 1;
        s->cv.notify_all();
    };
    if (slot.rc != MDC_OK) {
        const int rc = slot.rc;
        const std::string msg = slot.err;
        release();
        return mdc_internal_fail(rc, msg.c_str());
    }
    int rc = MDC_OK;
    if (s->uploaded < k) rc = start_upload(s, k);
    if (rc) {
        release();
        return rc;
    }
    // the next file's upload runs beside this file's parse, if its bytes are there already
    if (k + 1 < (int)s->paths.size() && s->slots.size() > 1) {
        Slot &nx = s->slots[(size_t)(k + 1) % s->slots.size()];
        bool ready;
        {
            std::lock_guard<std::mutex> lk(s->mu);
            ready = nx.state == READY && nx.index == k + 1 && nx.rc == MDC_OK;
        }
        if (ready) (void)start_upload(s, k + 1);  // on failure it is tried again in its own turn
    }
    if (hipEventSynchronize(s->up[k & 1]) != hipSuccess) {
        release();
        return mdc_internal_fail(MDC_ERR_HIP, "upload failed");
    }
    const size_t n = slot.n;
    rc = mdc_parse_device(s->parser, s->dbuf[k & 1], n, out);
    s->bytes += n;
    release();
    return rc;
}

extern "C" uint64_t mdc_stream_bytes(const mdc_stream *s) { return s ? s->bytes : 0; }

extern "C" void mdc_stream_close(mdc_stream *s) {
    if (!s) return;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        s->stop = true;
    }
    s->cv.notify_all();
    for (auto &t : s->workers) t.join();
    (void)hipSetDevice(s->device);
    if (s->copy) {
        (void)hipStreamSynchronize(s->copy);
        (void)hipStreamDestroy(s->copy);
    }
    for (auto e : s->up)
        if (e) (void)hipEventDestroy(e);
    for (auto &b : s->dbuf)
        if (b) (void)mdc_device_free(s->device, b);
    for (auto &slot : s->slots) pinned_put(slot.h, slot.cap);
    delete s;
}

// (for mdc_parse_host, which stages a pageable buffer through the same kept pinned buffers)
extern "C" int mdc_internal_pinned_get(size_t want, uint8_t **p, size_t *cap) {
    PinnedBuf b;
    if (!pinned_get(want, b)) return 0;
    *p = b.p;
    *cap = b.cap;
    return 1;
}
extern "C" void mdc_internal_pinned_put(uint8_t *p, size_t cap) { pinned_put(p, cap); }

extern "C" int mdc_host_buffer_get(size_t want, void **h_buf, size_t *cap) {
    if (!h_buf || !cap) return mdc_internal_fail(MDC_ERR_ARG, "NULL argument");
    PinnedBuf b;
    if (!pinned_get(want ? want : 1, b)) return mdc_internal_fail(MDC_ERR_NOMEM, "hipHostMalloc failed");
    *h_buf = b.p;
    *cap = b.cap;
    return MDC_OK;
}
extern "C" void mdc_host_buffer_put(void *h_buf, size_t cap) { pinned_put((uint8_t *)h_buf, cap); }

extern "C" void mdc_stream_release_pinned(void) {
    std::vector<PinnedBuf> drop;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        drop.swap(g_pool);
        g_pool_bytes = 0;
    }
    for (auto &b : drop) (void)hipHostFree(b.p);
}
