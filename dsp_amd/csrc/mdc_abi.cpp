// mdc_abi.cpp -- host side of libmdc_parse.so: the C ABI declared in include/mdc_parse.h.
//
// Holds no parsing logic: it validates arguments, turns a schema into the blob the kernels
// read (column descriptors + tag hash table), sizes LDS regions and output buffers, launches
// the gfx950 kernels of mdc_kernels.hip and mirrors results to the host on request.  There is
// no CPU implementation of the path here: without a usable HIP device every parse call fails
// with MDC_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include <ctime>
#include <sys/stat.h>
#include <unistd.h>

#include "mdc_device.h"
#include "mdc_format.h"
#include "mdc_parse.h"

namespace mdc {
hipError_t launch_tile_kernel(int mode, const TileParams &P, const OutPtrs &O, hipStream_t stream);
hipError_t launch_emit_kernel(const TileParams &P, const OutPtrs &O, hipStream_t stream);
hipError_t launch_template_kernel(const TileParams &P, hipStream_t stream);
hipError_t launch_long_records(bool write, const TileParams &P, const OutPtrs &O, uint32_t count, hipStream_t stream);
hipError_t launch_mark_overlap(const TileParams &P, hipStream_t stream);
uint32_t scan_segments(uint32_t ntiles);
hipError_t launch_scan_tiles(const uint32_t *counts, uint64_t *base, uint32_t ntiles, uint32_t ncnt, uint64_t *report,
                             const uint64_t *status, uint32_t report_status, hipStream_t stream);
hipError_t launch_long_float(const uint8_t *schema, uint32_t *const *off, uint8_t *const *bytes,
                             uint64_t *const *val, uint32_t *valid, const uint32_t *found, uint64_t rows,
                             uint32_t ncols, hipStream_t stream);
hipError_t launch_fixed_stride(const uint32_t *off, const uint8_t *bytes, uint8_t *slab, uint64_t rows,
                               uint32_t stride, hipStream_t stream);
hipError_t launch_format_prep(const uint32_t *const *off, const uint64_t *nbytes, uint32_t ncols, uint64_t rows,
                              uint64_t *info, hipStream_t stream);
hipError_t launch_format(bool exact, const FormatParams &P, hipStream_t stream);
hipError_t launch_timetostr(const int64_t *tv, uint64_t rows, const TimeFmt &F, int32_t tz_offset, uint32_t *off,
                            uint8_t *bytes, uint32_t *status, hipStream_t stream);
uint32_t format_scan_parts(uint64_t rows);
hipError_t noalloc_scratch_init(int device);
void *noalloc_scratch_addr(int device);
hipError_t launch_format_measure(const FormatParams &P, uint32_t *elen, uint64_t *reclen, uint64_t *part,
                                 uint64_t *recpos, hipStream_t stream);
}  // namespace mdc

using namespace mdc;

// ------------------------------------------------------------------------------------------
// errors

static thread_local std::string g_err;

static int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return fail(MDC_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));    \
    } while (0)

// ------------------------------------------------------------------------------------------
// devices and a caching device allocator (hipMalloc of GB-sized buffers costs milliseconds;
// a daemon parses file after file of the same shape)

// d_status: two status words, the diagnostic builds' counters from byte 16 on, and from u64
// index REPORT_TOTALS on the totals that the scan writes for the host
static const size_t REPORT_TOTALS = 32;
static const size_t REPORT_FLAGS = 8;  // (word index: 32 flag bytes of the write pass)
static_assert(NOALLOC_SCRATCH_BYTES == MDC_NOALLOC_SCRATCH_BYTES, "the header states the size of the static scratch array");
static const size_t STATUS_BYTES = (REPORT_TOTALS + MAX_COLS + 8) * 8;

constexpr int UPLOAD_THREADS = 4;  // host threads staging a pageable buffer (mdc_parse_host)

struct Device {
    int id = -1;
    hipStream_t stream = nullptr;
    hipStream_t up_stream[UPLOAD_THREADS] = {};     // created on the first large host buffer
    hipEvent_t up_event[2 * UPLOAD_THREADS] = {};
    std::mutex up_mu;  // one staged upload at a time per device (they would share the link anyway)
    std::mutex noalloc_mu;  // one call at a time per device in the static scratch array of mdc_parse_device_noalloc
    bool scratch_ready = false;  // (under noalloc_mu) the array's address on this device has been looked up
    std::mutex mu;
    std::multimap<size_t, void *> free_blocks;
    std::map<void *, size_t> live;
};

static std::mutex g_dev_mu;
static std::map<int, Device *> g_devices;

static int get_device(int id, Device **out) {
    if (id < 0) id = 0;
    std::lock_guard<std::mutex> lk(g_dev_mu);
    auto it = g_devices.find(id);
    if (it != g_devices.end()) {
        *out = it->second;
        return MDC_OK;
    }
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(MDC_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU path",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (id >= count) return fail(MDC_ERR_ARG, "device %d out of range (%d devices)", id, count);
    HIP_TRY(hipSetDevice(id));
    Device *d = new Device();
    d->id = id;
    const hipError_t se = hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking);
    if (se != hipSuccess) {
        delete d;
        return fail(MDC_ERR_HIP, "hipStreamCreateWithFlags: %s", hipGetErrorString(se));
    }
    g_devices[id] = d;
    *out = d;
    return MDC_OK;
}

static size_t size_class(size_t n) {
    if (n < 4096) return 4096;
    // round up to 1/8 steps of the enclosing power of two
    size_t p = 1;
    while (p < n) p <<= 1;
    size_t step = p / 16;
    return (n + step - 1) / step * step;
}

static int dev_alloc(Device *d, size_t n, void **out) {
    size_t want = size_class(n ? n : 1);
    {
        std::lock_guard<std::mutex> lk(d->mu);
        auto it = d->free_blocks.lower_bound(want);
        if (it != d->free_blocks.end() && it->first <= want + want / 4 + 65536) {
            *out = it->second;
            d->live[it->second] = it->first;
            d->free_blocks.erase(it);
            return MDC_OK;
        }
    }
    HIP_TRY(hipSetDevice(d->id));
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
        // drop the cache and retry once
        std::lock_guard<std::mutex> lk(d->mu);
        for (auto &kv : d->free_blocks) (void)hipFree(kv.second);
        d->free_blocks.clear();
        e = hipMalloc(&p, want);
        if (e != hipSuccess) return fail(MDC_ERR_NOMEM, "hipMalloc(%zu): %s", want, hipGetErrorString(e));
    }
    std::lock_guard<std::mutex> lk(d->mu);
    d->live[p] = want;
    *out = p;
    return MDC_OK;
}

static void dev_free(Device *d, void *p) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(d->mu);
    auto it = d->live.find(p);
    if (it == d->live.end()) return;
    d->free_blocks.emplace(it->second, p);
    d->live.erase(it);
}

// ------------------------------------------------------------------------------------------
// parser

struct mdc_parser {
    Device *dev = nullptr;
    uint32_t mode = MODE_XML;
    int ncols = 0;
    uint32_t words = 1;
    unsigned flags = 0;
    std::vector<mdc_col> cols;
    std::vector<int> typed;
    std::vector<uint8_t> blob;
    void *d_blob = nullptr;
    LdsLayout Ls[3]{};  // one layout per look-behind level
    LdsLayout LEs[3]{}; // ... and the emit kernel's
    LdsLayout LEd[3]{}; // ... for inputs of short records
    LdsLayout LSs[3]{}; // ... and the sizing pass's
    LdsLayout LF{};     // the one-pass parse's (first look-behind level only)
    // one-pass parse: counts published from tile to tile ((ncols + 1) x ws_tiles words) and what the output
    // arrays hold; the output sizes are guessed from the previous parse's bytes per input byte
    uint64_t *d_pub = nullptr;
    uint64_t *d_caps = nullptr;
    std::vector<double> est_bpb;
    bool est_valid = false;
    int fused_hold = 0;          // parses that stay with two passes after a one-pass parse had to be repeated
    bool one_pass = false;       // MDC_F_ONE_PASS, or MDC_ONE_PASS in the environment
    bool mark_nl = false;        // the end mark holds a newline: no template, and a pass that looks for overlapping marks
    uint32_t spin_limit = FUSED_SPIN_LIMIT;  // (test hook mdc_internal_spin_limit: a tiny limit makes waiting tiles give up)
    int lb_level = 0;   // level that fitted the previous parse
    // workspace
    uint32_t *d_counts = nullptr;
    uint64_t *d_base = nullptr;
    uint32_t *d_status = nullptr;
    uint32_t *d_tmpl = nullptr;     // tag template (see TMPL_MAX): built from the first input, kept while the
    uint32_t tmpl_sample = 0;       // the record the next template is taken from (cycles when tiles do not follow)
    int tmpl_fail = 0, tmpl_rest = 0;  // parses in a row whose tiles mostly left the template; parses to go without one
    bool tmpl_valid = false;        // inputs follow it (any template of this schema gives right results: a tile
                                    // whose records differ takes the table lookup), rebuilt when many tiles do not
    uint32_t *d_hdr = nullptr;      // per tile: cell map slot, first record start
    void *d_long = nullptr;         // records longer than the look-behind: list (one LongRec per tile at most), then their cells' lengths
    uint32_t *d_cellmap = nullptr;  // (ncols + 1) x map_cap words
    size_t map_cap = 0, map_stride = 0;
    double rows_per_byte = 0;       // of the previous parse; sizes the cell map
    size_t ws_tiles = 0;
    // pinned: h_report mirrors the head of d_status (status words, then from REPORT_TOTALS on the
    // ncols+1 totals, where h_totals points); behind it the batch's pointer table on its way to the device
    uint64_t *h_report = nullptr;
    uint64_t *h_report_dev = nullptr;  // the same memory as the device addresses it
    uint64_t *h_totals = nullptr;
    void **h_ptrs = nullptr;
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // sizing begins, ends, scan ends, write ends, write begins
    mdc_timing timing{};
    // test hook (mdc_internal_skew_counts): the next parse changes one counter of the sizing pass
    // before the scan, so that the write pass meets a tile that was sized wrongly
    int skew_tile = -1, skew_col = 0, skew_delta = 0;
};

struct mdc_batch {
    Device *dev = nullptr;
    int64_t rows = 0;
    int ncols = 0;
    uint32_t words = 1;
    std::vector<int> typed;
    std::vector<int> collen;
    uint64_t *d_rec_off = nullptr;
    std::vector<uint32_t *> d_off;
    std::vector<uint8_t *> d_bytes;
    std::vector<uint64_t> nbytes;
    std::vector<uint64_t *> d_val;
    uint32_t *d_valid = nullptr, *d_found = nullptr;
    void *d_ptrs = nullptr;  // device arrays of the column pointers
    // host mirrors
    std::vector<uint64_t> h_rec_off;
    std::vector<std::vector<uint32_t>> h_off;
    std::vector<std::vector<uint8_t>> h_bytes;
    std::vector<std::vector<uint64_t>> h_val;
    std::vector<uint32_t> h_valid, h_found;
    std::vector<bool> have_col, have_val;
    bool have_masks = false, have_recoff = false;
    std::vector<std::vector<char>> h_fixed;
    std::vector<size_t> fixed_stride;
};

static uint32_t round16(uint32_t v) { return (v + 15) & ~15u; }

static const OutPtrs g_no_out{};  // the sizing pass writes to no output array

// five blocks per CU fit when a block asks for at most this much LDS (measured: 31.6 KB does, 32.2 KB does not)
static const uint32_t LDS_FIVE_BLOCKS = 31 * 1024 + 256;
static const uint32_t LDS_PER_CU = 160 * 1024;  // what one block may ask for at most

// lean = the sizing pass of tagged input: it keeps no event list, stages no cells and needs no
// output pointers, which is worth a sixth block per CU
static const uint32_t LDS_SIX_BLOCKS = 25 * 1024 + 512;
// sizing = the sizing pass (it needs no output pointers and no table of powers of ten; blob_bytes is
// the schema without that table); of delimited input it takes batches of up to 128 records where
// they fit, since such records are short (a list of 2,568 instead of 3,072 separators pays for that).
static void compute_layout(uint32_t ncols, uint32_t npat, uint32_t words, uint32_t blob_bytes, uint32_t lbmax,
                           LdsLayout &L, bool lean = false, bool sizing = false) {
    // (a write kernel that follows the tag template needs the row masks and the template side by side)
    const bool own_tmpl = npat && !sizing;
    uint32_t RB = sizing && !npat ? 128 : 112;
    for (;;) {
        uint32_t o = 0;
        L.lbmax = lbmax;
        L.data = o;
        o += round16(TILE + lbmax + DATA_SLACK);
        L.evlist = o;
        L.evcap = lean ? 0 : npat ? 1280 : sizing ? 1712 : 2048;  // staging area of the full write kernel; separators of a tile
        L.stage_cap = L.evcap * 3;
        o += L.evcap * 3;
        L.evt = o;
        // XML: two 16-bit first positions per word; delimited: one descriptor word per field, plus the separator index
        o += round16(npat ? (RB * npat + 1) / 2 * 4 : RB * (ncols + 2) * 4);
        L.recend = o;
        o += round16((RB + 1) * 4);
        L.vmask = o;
        o += round16(RB * words * 4);
        L.fmask = o;
        o += round16(RB * words * 4);
        L.coltot = o;
        L.tmpl = L.vmask;
        L.tmpl_bytes = L.coltot - L.vmask;
        o += round16(2 * (ncols + 1) * 4);
        L.colout = o;
        if (!lean && !sizing) o += round16(ncols * 32);
        if (own_tmpl) {
            L.tmpl = o;
            L.tmpl_bytes = round16((RB + 2) * 2) + round16(ncols * 8) + 16 * 40;
            o += L.tmpl_bytes;
        }
        L.schema = o;
        o += round16(blob_bytes);
        L.misc = o;
        o += 64;
        L.total = o;
        L.RB = RB;
        if (o <= (lean ? LDS_SIX_BLOCKS : LDS_FIVE_BLOCKS) || RB <= 8) break;
        if (RB > 64 && lbmax > 1536 && o <= LDS_PER_CU) break;  // the larger look-behind configurations trade occupancy
        RB = RB > 16 ? RB - 8 : RB / 2;
    }
}

// The emit kernel's LDS: tile image, map slice (in `evt`), staging (in `evlist`), masks,
// schema; small enough for five blocks per CU at the first look-behind level.
// dense = for inputs of short records (more than a hundred to a tile): batches of 128 records, paid
// for with a smaller staging area (their cells are short as well).
static void compute_emit_layout(uint32_t ncols, uint32_t words, uint32_t lbmax, LdsLayout &L, bool dense = false) {
    uint32_t RB = dense ? 128 : 112;
    for (;;) {
        uint32_t o = 0;
        L = LdsLayout{};
        L.lbmax = lbmax;
        L.data = o;
        o += round16(TILE + lbmax + DATA_SLACK);
        L.evlist = o;
        L.evcap = 0;
        L.stage_cap = dense ? 5120 : 6144;
        o += L.stage_cap;
        L.evt = o;
        o += round16(RB * (ncols + 1) * 4);
        L.vmask = o;
        o += round16(RB * words * 4);
        L.fmask = o;
        o += round16(RB * words * 4);
        L.coltot = o;
        o += round16(2 * (ncols + 1) * 4);
        L.colout = o;
        o += round16(ncols * 32);
        L.schema = o;
        o += (uint32_t)sizeof(SchemaHdr) + round16(ncols * (uint32_t)sizeof(DevCol)) + P10_TABLE_DOUBLES * 8;  // header, columns, 10^k
        L.total = o;
        L.RB = RB;
        if (o <= LDS_FIVE_BLOCKS || RB <= 8) break;
        if (RB > 64 && lbmax > 1536 && o <= LDS_PER_CU) break;
        RB = RB > 16 ? RB - 8 : RB / 2;
    }
}

// The one-pass parse's LDS (tagged input): the sizing pass's regions with the row masks and the template side
// by side, the ends of the cells within their columns, and what is left of a four-blocks-per-CU share as
// the staging area of the whole tile's cell bytes.
static const uint32_t LDS_FOUR_BLOCKS = 40 * 1024 - 768;
static void compute_fused_layout(uint32_t ncols, uint32_t npat, uint32_t words, uint32_t blob_bytes, uint32_t lbmax,
                                 LdsLayout &L) {
    uint32_t RB = 112;
    for (;;) {
        uint32_t o = 0;
        L = LdsLayout{};
        L.lbmax = lbmax;
        L.data = o;
        o += round16(TILE + lbmax + DATA_SLACK);
        L.evt = o;
        o += round16(npat ? (RB * npat + 1) / 2 * 4 : RB * (ncols + 2) * 4);
        L.recend = o;
        o += round16((RB + 1) * 4);
        L.vmask = o;
        o += round16(RB * words * 4);
        L.fmask = o;
        o += round16(RB * words * 4);
        L.coltot = o;
        o += round16(2 * (ncols + 1) * 4);
        L.colout = o;
        L.endl = o;
        o += round16(RB * ncols * 2);
        L.tmpl = o;
        L.tmpl_bytes = round16((RB + 2) * 2) + round16(ncols * 8) + 16 * 40;
        o += L.tmpl_bytes;
        L.schema = o;
        o += round16(blob_bytes);
        L.misc = o;
        o += 128;
        L.evlist = o;
        L.evcap = 0;
        const uint32_t room = o < LDS_FOUR_BLOCKS ? (LDS_FOUR_BLOCKS - o) & ~15u : 0;
        L.stage_cap = room > 16384 ? 16384 : room;
        o += L.stage_cap;
        L.total = o;
        L.RB = RB;
        if (L.stage_cap >= 64 * RB || RB <= 8) break;  // (about 64 bytes of cells a record)
        RB = RB > 16 ? RB - 8 : RB / 2;
    }
}

// look-behind capacities tried in turn when a record does not fit (fewer blocks per CU each)
static const uint32_t LB_LEVELS[3] = {1536, 8192, LBMAX};

static int build_blob(mdc_parser *p, const char *endbz, int carry, const char *sep, int bdelspace, int tz_offset) {
    const uint32_t ncols = (uint32_t)p->ncols;
    const uint32_t npat = p->mode == MODE_XML ? 2 * ncols : 0;
    size_t el = strlen(endbz);
    if (el > MAX_LONG_TEXT) return fail(MDC_ERR_ARG, "endbz longer than %d bytes", MAX_LONG_TEXT);
    // (An end mark that holds a newline may overlap the terminator in front of it, and whether it counts then
    // depends on where the record began: cifile::readline matches the mark within the record it has gathered
    // so far.  Inputs in which no two occurrences of the mark overlap need no such left-to-right decision; a
    // pass over the input looks for an overlap and the parse is refused where there is one.)
    p->mark_nl = memchr(endbz, '\n', el) != nullptr;
    size_t sl = 0;
    if (p->mode == MODE_CSV) {
        sl = strlen(sep);
        if (sl == 0 || sl > MAX_LONG_TEXT) return fail(MDC_ERR_ARG, "separator must be 1..%d bytes", MAX_LONG_TEXT);
    }
    std::vector<DevCol> dc(ncols);
    uint32_t any_typed = 0;
    for (uint32_t c = 0; c < ncols; ++c) {
        const mdc_col &col = p->cols[c];
        memset(&dc[c], 0, sizeof(DevCol));
        size_t nl = strnlen(col.colname, 32);
        if (nl > 31) return fail(MDC_ERR_ARG, "column %u: name not NUL-terminated", c);
        if (memchr(col.colname, '>', nl) || memchr(col.colname, '\n', nl))
            return fail(MDC_ERR_ARG, "column %u: name must not contain '>' or a newline", c);
        memcpy(dc[c].name, col.colname, nl);
        dc[c].namelen = (uint8_t)nl;
        if (strcmp(col.datatype, "number") == 0)
            dc[c].level = LVL_NUMBER;
        else if (strcmp(col.datatype, "date") == 0)
            dc[c].level = LVL_DATE;
        else
            dc[c].level = LVL_CHAR;  // splitbuffer leaves every other type untouched
        int t = p->typed[c];
        if (t == MDC_T_DEFAULT) t = dc[c].level == LVL_NUMBER ? MDC_T_F64 : dc[c].level == LVL_DATE ? MDC_T_I64 : MDC_T_NONE;
        if (t < MDC_T_NONE || t > MDC_T_TIME) return fail(MDC_ERR_ARG, "column %u: bad typed code %d", c, t);
        p->typed[c] = t;
        dc[c].typed = (uint8_t)t;
        if (t != MDC_T_NONE) any_typed = 1;
        dc[c].collen = col.collen;
    }

    // tag texts -> perfect hash
    struct Pat {
        uint8_t text[48];  // text + '>' padded with zeros
        uint32_t tlen, col, kind, lo, hi;
    };
    std::vector<Pat> pats;
    for (uint32_t c = 0; c < ncols && p->mode == MODE_XML; ++c)
        for (uint32_t kind = 0; kind < 2; ++kind) {
            Pat q;
            memset(&q, 0, sizeof q);
            uint32_t tlen = 0;
            if (kind) q.text[tlen++] = '/';
            memcpy(q.text + tlen, dc[c].name, dc[c].namelen);
            tlen += dc[c].namelen;
            q.text[tlen] = '>';
            q.tlen = tlen;
            q.col = c;
            q.kind = kind;
            uint8_t key[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            memcpy(key, q.text, tlen + 1 < 8 ? tlen + 1 : 8);
            memcpy(&q.lo, key, 4);
            memcpy(&q.hi, key + 4, 4);
            pats.push_back(q);
        }
    // The smallest table (a power of two, 1.4 slots per pattern at least) for which a seed puts
    // all distinct keys into slots of their own: every 32-byte slot is LDS that the kernels would
    // rather spend on resident blocks.  Twice that size is tried as well; where no such seed
    // exists (many similar names) the seed with the fewest shared slots is taken and the keys
    // that share a slot are chained.
    uint32_t ht_size = 16, seed = 0;
    while (ht_size * 5 < npat * 7) ht_size <<= 1;
    std::vector<int> head;  // slot -> first pattern index that lands there, or -1
    {
        uint32_t best_size = ht_size, best_seed = 0, best_shared = 0xFFFFFFFFu;
        for (uint32_t size = ht_size; size <= 2 * ht_size && best_shared; size <<= 1)
            for (uint32_t sd = 0; sd < 16384 && best_shared; ++sd) {
                head.assign(size, -1);
                uint32_t shared = 0;
                for (size_t i = 0; i < pats.size() && shared < best_shared; ++i) {
                    const uint32_t slot = ht_hash(pats[i].lo, pats[i].hi, sd * 0x9E3779B9u) & (size - 1);
                    if (head[slot] < 0)
                        head[slot] = (int)i;
                    else if (pats[head[slot]].lo != pats[i].lo || pats[head[slot]].hi != pats[i].hi)
                        ++shared;  // (another key than the slot's first: counted once per pattern)
                }
                if (shared < best_shared) {
                    best_shared = shared;
                    best_size = size;
                    best_seed = sd;
                }
            }
        ht_size = best_size;
        seed = best_seed;
        head.assign(ht_size, -1);
        for (size_t i = 0; i < pats.size(); ++i) {
            const uint32_t slot = ht_hash(pats[i].lo, pats[i].hi, seed * 0x9E3779B9u) & (ht_size - 1);
            if (head[slot] < 0) head[slot] = (int)i;
        }
    }
    seed *= 0x9E3779B9u;
    // chained entries (same key) go behind the hashed slots
    std::vector<HtEntry> ht(ht_size);
    memset(ht.data(), 0, ht.size() * sizeof(HtEntry));
    auto fill = [&](HtEntry &e, const Pat &q) {
        e.klo = q.lo;
        e.khi = q.hi;
        e.meta = 0x80000000u | (q.col << 16) | (q.kind << 8) | q.tlen;
        if (q.tlen + 1 > 16) e.meta |= 0x40000000u;
        uint8_t k2[8] = {0}, m2[8] = {0};
        for (uint32_t k = 8; k < 16 && k <= q.tlen; ++k) {
            k2[k - 8] = q.text[k];
            m2[k - 8] = 0xFF;
        }
        memcpy(&e.k2lo, k2, 4);
        memcpy(&e.k2hi, k2 + 4, 4);
        memcpy(&e.m2lo, m2, 4);
        memcpy(&e.m2hi, m2 + 4, 4);
        e.next = 0;
    };
    for (size_t i = 0; i < pats.size(); ++i) {
        uint32_t slot = ht_hash(pats[i].lo, pats[i].hi, seed) & (ht_size - 1);
        if (head[slot] == (int)i) {
            fill(ht[slot], pats[i]);
        } else {
            HtEntry e;
            fill(e, pats[i]);
            uint32_t at = slot;
            while (ht[at].next) at = ht[at].next;
            ht.push_back(e);
            ht[at].next = (uint32_t)ht.size() - 1;
        }
    }

    const size_t cols_bytes = (ncols * sizeof(DevCol) + 15) & ~(size_t)15;
    // (an end mark or a separator too long for the header lies in front of the table of powers of ten)
    const size_t text_off = sizeof(SchemaHdr) + cols_bytes + ht.size() * sizeof(HtEntry);
    const size_t p10_off = text_off + long_text_room((uint32_t)el, MAX_ENDBZ) + long_text_room((uint32_t)sl, MAX_SEP);
    p->blob.assign(p10_off + P10_TABLE_DOUBLES * sizeof(double), 0);
    p->blob.resize((p->blob.size() + 15) & ~(size_t)15);
    SchemaHdr *H = (SchemaHdr *)p->blob.data();
    H->ncols = ncols;
    H->npat = npat;
    H->ht_mask = ht_size - 1;
    H->ht_seed = seed;
    H->p10_off = (uint32_t)p10_off;
    H->carry = carry ? 1 : 0;
    H->mode = p->mode;
    H->words = p->words;
    H->bdelspace = bdelspace ? 1 : 0;
    H->any_typed = any_typed;
    H->tz_offset = tz_offset;
    H->endbz_len = (uint32_t)el;
    memcpy(H->endbz, endbz, el < MAX_ENDBZ ? el : MAX_ENDBZ);
    if (el <= 8) memcpy(H->endbz_tail + 8 - el, endbz, el);
    H->sep_len = (uint32_t)sl;
    if (sl) memcpy(H->sep, sep, sl < MAX_SEP ? sl : MAX_SEP);
    if (el > MAX_ENDBZ) memcpy(p->blob.data() + text_off, endbz, el);
    if (sl > MAX_SEP) memcpy(p->blob.data() + text_off + long_text_room((uint32_t)el, MAX_ENDBZ), sep, sl);
    H->sep_fast = sl >= 1 && sl <= 4;
    for (size_t k = 0; k < sl && k < 4; ++k)
        if ((uint8_t)sep[k] >= 0x80 || sep[k] == '\n') H->sep_fast = 0;
    memcpy(p->blob.data() + sizeof(SchemaHdr), dc.data(), ncols * sizeof(DevCol));
    memcpy(p->blob.data() + sizeof(SchemaHdr) + cols_bytes, ht.data(), ht.size() * sizeof(HtEntry));
    double p10 = 1.0;
    for (int k = 0; k < 23; ++k) {
        memcpy(p->blob.data() + p10_off + 8 * k, &p10, 8);
        p10 *= 10.0;  // exact up to 1e22
    }
    // behind them (16-byte aligned): the pairs {10^k, 1 / 10^k rounded to nearest} for k < P10_PAIRS
    p10 = 1.0;
    for (int k = 0; k < P10_PAIRS; ++k) {
        const double r = 1.0 / p10;
        memcpy(p->blob.data() + p10_off + 8 * (P10_PAIR_AT + 2 * k), &p10, 8);
        memcpy(p->blob.data() + p10_off + 8 * (P10_PAIR_AT + 2 * k + 1), &r, 8);
        p10 *= 10.0;
    }
    for (int lv = 0; lv < 3; ++lv) {
        compute_layout(ncols, npat, p->words, (uint32_t)p->blob.size(), LB_LEVELS[lv], p->Ls[lv]);
        compute_emit_layout(ncols, p->words, LB_LEVELS[lv], p->LEs[lv]);
        compute_emit_layout(ncols, p->words, LB_LEVELS[lv], p->LEd[lv], true);
        compute_layout(ncols, npat, p->words, ((const SchemaHdr *)p->blob.data())->p10_off, LB_LEVELS[lv], p->LSs[lv], npat != 0, true);
        for (const LdsLayout *L : {&p->Ls[lv], &p->LEs[lv], &p->LEd[lv], &p->LSs[lv]})
            if (L->total > LDS_PER_CU) return fail(MDC_ERR_ARG, "schema too large for LDS (%u bytes)", L->total);
    }
    compute_fused_layout(ncols, npat, p->words, (uint32_t)p->blob.size(), LB_LEVELS[0], p->LF);
    return MDC_OK;
}

static int parser_finish(mdc_parser *p, const mdc_opts *opts) {
    int rc = get_device(opts ? opts->device : -1, &p->dev);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(p->dev->id));
    rc = dev_alloc(p->dev, p->blob.size(), &p->d_blob);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(p->d_blob, p->blob.data(), p->blob.size(), hipMemcpyHostToDevice));
    void *st = nullptr;
    rc = dev_alloc(p->dev, STATUS_BYTES, &st);
    if (rc) return rc;
    p->d_status = (uint32_t *)st;
    HIP_TRY(hipMemset(p->d_status, 0, STATUS_BYTES));
    rc = dev_alloc(p->dev, (1 + TMPL_MAX) * 16 + MAX_COLS * 8 + 16, &st);
    if (rc) return rc;
    p->d_tmpl = (uint32_t *)st;
    HIP_TRY(hipMemset(p->d_tmpl, 0, (1 + TMPL_MAX) * 16 + MAX_COLS * 8 + 16));
    rc = dev_alloc(p->dev, (MAX_COLS + 1) * 8, &st);
    if (rc) return rc;
    p->d_caps = (uint64_t *)st;
    p->est_bpb.assign(p->ncols, 0.0);
    p->one_pass = (p->flags & MDC_F_ONE_PASS) != 0 || getenv("MDC_ONE_PASS") != nullptr;
    // (status words and flag bytes | totals | pointer table | capacities of the output arrays)
    HIP_TRY(hipHostMalloc((void **)&p->h_report, (REPORT_TOTALS + MAX_COLS + 8 + 3 * MAX_COLS + MAX_COLS + 8) * sizeof(uint64_t),
                          hipHostMallocDefault));
    HIP_TRY(hipHostGetDevicePointer((void **)&p->h_report_dev, p->h_report, 0));
    p->h_totals = p->h_report + REPORT_TOTALS;
    p->h_ptrs = (void **)(p->h_totals + MAX_COLS + 8);
    for (int i = 0; i < 5; ++i) HIP_TRY(hipEventCreate(&p->ev[i]));
    return MDC_OK;
}

extern "C" int mdc_parser_create_xml(const mdc_col *cols, int ncols, const mdc_opts *opts, mdc_parser **out) {
    if (!out) return fail(MDC_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (ncols < 1 || ncols > MAX_COLS || !cols) return fail(MDC_ERR_ARG, "ncols must be 1..%d", MAX_COLS);
    mdc_parser *p = new mdc_parser();
    p->mode = MODE_XML;
    p->ncols = ncols;
    p->words = (uint32_t)(ncols + 31) / 32;
    p->flags = opts ? opts->flags : 0;
    p->cols.assign(cols, cols + ncols);
    p->typed.assign(ncols, MDC_T_DEFAULT);
    if (opts && opts->typed) p->typed.assign(opts->typed, opts->typed + ncols);
    const char *endbz = (opts && opts->endbz) ? opts->endbz : "<endl/>";
    int carry = opts ? opts->carry : 1;
    int rc = build_blob(p, endbz, carry, "", 0, opts ? opts->tz_offset : 0);
    if (!rc) rc = parser_finish(p, opts);
    if (rc) {
        mdc_parser_destroy(p);
        return rc;
    }
    *out = p;
    return MDC_OK;
}

extern "C" int mdc_parser_create_csv(const char *sep, int bdelspace, int nfields, const int *ilen,
                                     const mdc_opts *opts, mdc_parser **out) {
    if (!out) return fail(MDC_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (nfields < 1 || nfields > MAX_COLS || !sep) return fail(MDC_ERR_ARG, "nfields must be 1..%d", MAX_COLS);
    mdc_parser *p = new mdc_parser();
    p->mode = MODE_CSV;
    p->ncols = nfields;
    p->words = (uint32_t)(nfields + 31) / 32;
    p->flags = opts ? opts->flags : 0;
    p->cols.resize(nfields);
    for (int c = 0; c < nfields; ++c) {
        memset(&p->cols[c], 0, sizeof(mdc_col));
        snprintf(p->cols[c].colname, 32, "f%d", c);
        strcpy(p->cols[c].datatype, "char");
        p->cols[c].collen = ilen ? ilen[c] : 0;
    }
    p->typed.assign(nfields, MDC_T_NONE);
    if (opts && opts->typed) p->typed.assign(opts->typed, opts->typed + nfields);
    for (int c = 0; c < nfields; ++c)
        if (p->typed[c] == MDC_T_DEFAULT) p->typed[c] = MDC_T_NONE;
    const char *endbz = (opts && opts->endbz) ? opts->endbz : "";
    int rc = build_blob(p, endbz, 0, sep, bdelspace, opts ? opts->tz_offset : 0);
    if (!rc) rc = parser_finish(p, opts);
    if (rc) {
        mdc_parser_destroy(p);
        return rc;
    }
    *out = p;
    return MDC_OK;
}

extern "C" void mdc_parser_destroy(mdc_parser *p) {
    if (!p) return;
    if (p->dev) {
        (void)hipSetDevice(p->dev->id);
        dev_free(p->dev, p->d_blob);
        dev_free(p->dev, p->d_counts);
        dev_free(p->dev, p->d_base);
        dev_free(p->dev, p->d_status);
        dev_free(p->dev, p->d_tmpl);
        dev_free(p->dev, p->d_hdr);
        dev_free(p->dev, p->d_long);
        dev_free(p->dev, p->d_cellmap);
        dev_free(p->dev, p->d_pub);
        dev_free(p->dev, p->d_caps);
        if (p->h_report) (void)hipHostFree(p->h_report);
        for (int i = 0; i < 5; ++i)
            if (p->ev[i]) (void)hipEventDestroy(p->ev[i]);
    }
    delete p;
}

// ------------------------------------------------------------------------------------------
// parsing

static void batch_release(mdc_batch *b) {
    if (!b) return;
    if (b->dev) {
        dev_free(b->dev, b->d_rec_off);
        for (auto q : b->d_off) dev_free(b->dev, q);
        for (auto q : b->d_bytes) dev_free(b->dev, q);
        for (auto q : b->d_val) dev_free(b->dev, q);
        dev_free(b->dev, b->d_valid);
        dev_free(b->dev, b->d_found);
        dev_free(b->dev, b->d_ptrs);
    }
    delete b;
}

extern "C" void mdc_batch_free(mdc_batch *b) { batch_release(b); }

// a tile may hand its first record to the long-record kernel: a list entry and the lengths of its cells
static size_t long_bytes_per_tile(int ncols) { return sizeof(LongRec) + (size_t)ncols * 4; }

static int ensure_workspace(mdc_parser *p, size_t ntiles) {
    if (ntiles <= p->ws_tiles && p->d_counts) return MDC_OK;
    dev_free(p->dev, p->d_counts);
    dev_free(p->dev, p->d_base);
    dev_free(p->dev, p->d_hdr);
    dev_free(p->dev, p->d_long);
    dev_free(p->dev, p->d_pub);
    p->d_counts = nullptr;
    p->d_base = nullptr;
    p->d_hdr = nullptr;
    p->d_long = nullptr;
    p->d_pub = nullptr;
    p->ws_tiles = 0;
    size_t ncnt = (size_t)p->ncols + 1;
    void *a = nullptr, *b = nullptr, *h = nullptr;
    int rc = dev_alloc(p->dev, ntiles * 8 + 16, &h);
    if (rc) return rc;
    p->d_hdr = (uint32_t *)h;
    rc = dev_alloc(p->dev, ntiles * long_bytes_per_tile(p->ncols), &p->d_long);
    if (rc) return rc;
    rc = dev_alloc(p->dev, ntiles * ncnt * 4, &a);
    if (rc) return rc;
    rc = dev_alloc(p->dev, (ntiles + 1 + scan_segments((uint32_t)ntiles)) * ncnt * 8, &b);
    if (rc) {
        dev_free(p->dev, a);
        return rc;
    }
    p->d_counts = (uint32_t *)a;
    p->d_base = (uint64_t *)b;
    if (p->mode == MODE_XML && p->one_pass) {
        void *q = nullptr;
        if (dev_alloc(p->dev, ntiles * ncnt * 8, &q) == MDC_OK) p->d_pub = (uint64_t *)q;
    }
    p->ws_tiles = ntiles;
    return MDC_OK;
}

// Room for the cell map: every tile owns map_stride record slots (one word per column and
// record, plus the record ends).  The number of records per tile is not known before the
// sizing pass, so the stride is a guess: the previous parse's records per tile with a margin,
// else one record per 32 bytes; a tile with more records than that is written by the full
// write kernel instead.
static int ensure_cellmap(mdc_parser *p, size_t n, size_t ntiles) {
    const size_t per_slot = ((size_t)p->ncols + 1) * 4;
    size_t stride = TILE / 32;
    if (p->rows_per_byte > 0) {
        stride = (size_t)(p->rows_per_byte * TILE * 1.25) + 16;
        stride = (stride + 15) & ~(size_t)15;
    }
    if (ntiles == 0) ntiles = 1;
    // the map is scratch: keep it within twice the input (64 MiB at least)
    const size_t budget = n > ((size_t)32 << 20) ? 2 * n : (size_t)64 << 20;
    if (stride * ntiles * per_slot > budget) stride = (budget / (ntiles * per_slot)) & ~(size_t)15;
    size_t want = stride * ntiles;
    if (stride == 0 || want > 0xFFFF0000u) {  // no map: every tile takes the full write kernel
        p->map_stride = 0;
        return MDC_OK;
    }
    p->map_stride = stride;
    if (p->d_cellmap && want <= p->map_cap && p->map_cap <= 4 * want) return MDC_OK;
    dev_free(p->dev, p->d_cellmap);
    p->d_cellmap = nullptr;
    p->map_cap = 0;
    void *m = nullptr;
    if (dev_alloc(p->dev, per_slot * want + 64, &m) != MDC_OK) {
        p->map_stride = 0;  // no memory to spare for the map: do without
        return MDC_OK;
    }
    p->d_cellmap = (uint32_t *)m;
    p->map_cap = want;
    return MDC_OK;
}

static int status_to_error(uint32_t st) {
    if (st & ST_RECORD_TOO_LONG)
        return fail(MDC_ERR_RECORD_TOO_LONG, "a record of 2 GiB or more (the reference's own positions are ints)");
    if (st & ST_MARK_OVERLAP)
        return fail(MDC_ERR_ARG, "two occurrences of the end mark overlap in this input (a newline inside one follows another): "
                                 "which of them ends a record is a left-to-right decision this parser does not take");
    if (st) return fail(MDC_ERR_INTERNAL, "kernel status %#x", st);
    return MDC_OK;
}

// A new batch with room for rows_cap rows and bytes_cap[c] bytes of column c (the two-pass parse asks for
// exactly what its sizing pass counted, the one-pass parse for a guess with a margin).
static int alloc_batch(mdc_parser *p, uint64_t rows_cap, const uint64_t *bytes_cap, mdc_batch **out) {
    Device *d = p->dev;
    const uint32_t ncols = (uint32_t)p->ncols;
    mdc_batch *b = new mdc_batch();
    b->dev = d;
    b->ncols = (int)ncols;
    b->words = p->words;
    b->typed = p->typed;
    for (auto &c : p->cols) b->collen.push_back(c.collen);
    b->d_off.assign(ncols, nullptr);
    b->d_bytes.assign(ncols, nullptr);
    b->d_val.assign(ncols, nullptr);
    b->nbytes.assign(ncols, 0);
    b->h_off.resize(ncols);
    b->h_bytes.resize(ncols);
    b->h_val.resize(ncols);
    b->have_col.assign(ncols, false);
    b->have_val.assign(ncols, false);
    b->h_fixed.resize(ncols);
    b->fixed_stride.assign(ncols, 0);
    int rc = MDC_OK;
    void *q = nullptr;
#define ALLOC_OR_FAIL(bytes, dst, T)     \
    do {                                 \
        rc = dev_alloc(d, (bytes), &q);  \
        if (rc) {                        \
            batch_release(b);            \
            return rc;                   \
        }                                \
        dst = (T)q;                      \
    } while (0)
    if (!(p->flags & MDC_F_NO_RECOFF)) ALLOC_OR_FAIL((rows_cap + 1) * 8, b->d_rec_off, uint64_t *);
    for (uint32_t c = 0; c < ncols; ++c) {
        ALLOC_OR_FAIL((rows_cap + 1) * 4, b->d_off[c], uint32_t *);
        ALLOC_OR_FAIL(bytes_cap[c] + 16, b->d_bytes[c], uint8_t *);
        if (p->typed[c] != MDC_T_NONE) ALLOC_OR_FAIL(rows_cap * 8 + 8, b->d_val[c], uint64_t *);
    }
    ALLOC_OR_FAIL(rows_cap * p->words * 4 + 4, b->d_valid, uint32_t *);
    ALLOC_OR_FAIL(rows_cap * p->words * 4 + 4, b->d_found, uint32_t *);
    ALLOC_OR_FAIL(3 * (size_t)ncols * sizeof(void *), b->d_ptrs, void *);
#undef ALLOC_OR_FAIL
    *out = b;
    return MDC_OK;
}

// algorithmic output bytes of a parse: offsets, cells, values, the two row masks, record offsets
static uint64_t output_bytes(const mdc_parser *p, uint64_t rows, const uint64_t *totals, bool recoff) {
    uint64_t out = 2 * rows * p->words * 4 + (recoff ? (rows + 1) * 8 : 0);
    for (int c = 0; c < p->ncols; ++c) out += (rows + 1) * 4 + totals[c] + (p->typed[c] != MDC_T_NONE ? rows * 8 : 0);
    return out;
}

// what the next parse of this parser may expect: rows and bytes of every column per input byte
static void remember_sizes(mdc_parser *p, size_t n, uint64_t rows, const uint64_t *totals) {
    if (!n) return;
    p->rows_per_byte = (double)rows / (double)n;
    for (int c = 0; c < p->ncols; ++c) p->est_bpb[c] = (double)totals[c] / (double)n;
    p->est_valid = true;
}

static const int FUSED_REDO = 1000;  // (private) the one-pass parse was given up: parse in two passes

// The one-pass parse: one launch of the tile kernel that sizes and writes.  The output arrays must exist
// before anything is counted: the caller's (ext), or a new batch sized from the previous parse's bytes per
// input byte with a margin.  A tile whose place lies beyond an array stores nothing and raises ST_OVERFLOW;
// then, after a long record (whose cells this pass cannot count in time) or a tile that gave up waiting,
// the result is dropped and FUSED_REDO returned.  Tiles that do not take the straight-line write are left,
// with their places, to the full write kernel.
static int parse_fused(mdc_parser *p, TileParams P, const mdc_outputs *ext, void *d_ptrs_ext, mdc_batch **out,
                       int64_t *rows_out, uint64_t *col_bytes) {
    Device *d = p->dev;
    hipStream_t s = d->stream;
    const uint32_t ncols = (uint32_t)p->ncols, ncnt = ncols + 1;
    const size_t n = (size_t)P.n, ntiles = P.ntiles;
    mdc_timing &t = p->timing;
    uint64_t *caps = p->h_totals + MAX_COLS + 8 + 3 * MAX_COLS;  // pinned, behind the pointer table
    mdc_batch *b = nullptr;
    void **hp = p->h_ptrs;
    int rc;
    if (ext) {
        for (uint32_t c = 0; c < ncols; ++c) {
            caps[c] = ext->cols[c].bytes_cap;
            hp[c] = ext->cols[c].d_off;
            hp[ncols + c] = ext->cols[c].d_bytes;
            hp[2 * ncols + c] = p->typed[c] != MDC_T_NONE ? ext->cols[c].d_val : nullptr;
        }
        caps[ncols] = (uint64_t)ext->rows_cap;
    } else {
        for (uint32_t c = 0; c < ncols; ++c) caps[c] = (uint64_t)(p->est_bpb[c] * (double)n * 1.03) + 65536;
        caps[ncols] = (uint64_t)(p->rows_per_byte * (double)n * 1.03) + 4096;
        for (uint32_t c = 0; c < ncols; ++c)
            if (caps[c] > 0xFFFFFFFFull) return FUSED_REDO;  // (the two-pass parse reports a column beyond 4 GiB)
        rc = alloc_batch(p, caps[ncols], caps, &b);
        if (rc) return rc;
        for (uint32_t c = 0; c < ncols; ++c) {
            hp[c] = b->d_off[c];
            hp[ncols + c] = b->d_bytes[c];
            hp[2 * ncols + c] = b->d_val[c];
        }
    }
    void *d_ptrs = ext ? d_ptrs_ext : b->d_ptrs;
    OutPtrs O{};
    for (uint32_t c = 0; c < ncols; ++c) {
        O.off[c] = (uint32_t *)hp[c];
        O.bytes[c] = (uint8_t *)hp[ncols + c];
        O.val[c] = (uint64_t *)hp[2 * ncols + c];
    }
    P.rec_off = (p->flags & MDC_F_NO_RECOFF) ? nullptr : ext ? ext->d_rec_off : b->d_rec_off;
    P.off = (uint32_t *const *)d_ptrs;
    P.bytes = (uint8_t *const *)((void **)d_ptrs + ncols);
    P.val = (uint64_t *const *)((void **)d_ptrs + 2 * ncols);
    P.valid = ext ? ext->d_valid : b->d_valid;
    P.found = ext ? ext->d_found : b->d_found;
    P.cellmap = nullptr;
    P.map_cap = P.map_stride = 0;
    P.fused_totals = p->h_report_dev + REPORT_TOTALS;
    P.spin_limit = p->spin_limit;
    volatile uint8_t *host_flags = (volatile uint8_t *)(p->h_report + REPORT_FLAGS);
    for (uint32_t i = 0; i < 32; ++i) host_flags[i] = 0;
    for (uint32_t j = 0; j < ncnt; ++j) p->h_totals[j] = 0;
    P.host_flags = (uint8_t *)(p->h_report_dev + REPORT_FLAGS);
    P.L = p->LF;
#define HIP_TRY_B(expr)                                                                \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) {                                                        \
            batch_release(b);                                                          \
            return fail(MDC_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));          \
        }                                                                              \
    } while (0)
    HIP_TRY_B(hipMemsetAsync(p->d_status, 0, 16, s));
    HIP_TRY_B(hipMemsetAsync(P.pub, 0, ntiles * ncnt * 8, s));
    HIP_TRY_B(hipMemcpyAsync((void *)P.caps, caps, ncnt * 8, hipMemcpyHostToDevice, s));
    HIP_TRY_B(hipEventRecord(p->ev[0], s));
    HIP_TRY_B(launch_tile_kernel(2, P, O, s));
    HIP_TRY_B(hipEventRecord(p->ev[3], s));
    // status word, tiles that left the template, long records, tiles left to the full write kernel
    uint32_t *hst = (uint32_t *)p->h_report;
    HIP_TRY_B(hipMemcpyAsync(hst, p->d_status, 16, hipMemcpyDeviceToHost, s));
    HIP_TRY_B(hipStreamSynchronize(s));
    uint32_t st = 0;
    for (uint32_t i = 0; i < 32; ++i)
        if (host_flags[i]) st |= 1u << i;
    const uint32_t left = hst[1], nlong = hst[2], odd = hst[3];
    t.lookup_tiles = (int)left;
    if ((size_t)left * 8 > ntiles) {  // few tiles follow the template: the next (two-pass) parse samples another record
        p->tmpl_valid = false;
        p->tmpl_sample = (p->tmpl_sample + 1) % 4;
    }
    if ((st & (ST_OVERFLOW | ST_GIVEUP)) || nlong) {
        batch_release(b);
        p->fused_hold = nlong ? 16 : 2;
        return FUSED_REDO;
    }
    if ((size_t)odd * 8 > ntiles) p->fused_hold = 16;  // (mostly the full write kernel: two passes do that better)
    const uint64_t rows = p->h_totals[ncols];
    if (odd) {
        // tiles that did not take the straight-line write, at the places this pass found for them
        for (uint32_t i = 0; i < 32; ++i) host_flags[i] = 0;
        P.L = p->Ls[0];
        P.only_unmapped = 1;
        HIP_TRY_B(launch_tile_kernel(1, P, O, s));
        HIP_TRY_B(hipEventRecord(p->ev[3], s));
        HIP_TRY_B(hipStreamSynchronize(s));
        for (uint32_t i = 0; i < 32; ++i)
            if (host_flags[i]) st |= 1u << i;
    }
    st &= ~ST_UNMAPPED;
    if (st & ST_NUMERIC_RANGE) {
        // some floating-point texts need the multi-precision stage
        hipError_t le = hipMemcpyAsync(d_ptrs, hp, 3 * (size_t)ncols * sizeof(void *), hipMemcpyHostToDevice, s);
        if (le == hipSuccess)
            le = launch_long_float((const uint8_t *)p->d_blob, P.off, P.bytes, P.val, P.valid, P.found, rows, ncols, s);
        if (le == hipSuccess) le = hipStreamSynchronize(s);
        if (le != hipSuccess) {
            batch_release(b);
            return fail(MDC_ERR_HIP, "long_float: %s", hipGetErrorString(le));
        }
        st &= ~ST_NUMERIC_RANGE;
    }
#undef HIP_TRY_B
    rc = status_to_error(st);
    if (rc) {
        batch_release(b);
        return rc;
    }
    remember_sizes(p, n, rows, p->h_totals);
    if (b) {
        b->rows = (int64_t)rows;
        for (uint32_t c = 0; c < ncols; ++c) b->nbytes[c] = p->h_totals[c];
    }
    t.count_ms = t.scan_ms = 0;
    (void)hipEventElapsedTime(&t.total_ms, p->ev[0], p->ev[3]);
    t.write_ms = t.total_ms;
    t.bytes_in = n;
    t.bytes_out = output_bytes(p, rows, p->h_totals, P.rec_off != nullptr);
    t.rows = (int64_t)rows;
    t.tiles = (int)ntiles;
    t.full_write = odd ? 1 : 0;
    t.long_records = 0;
    t.one_pass = 1;
    t.one_pass_redone = 0;
    if (out) *out = b;
    if (rows_out) *rows_out = (int64_t)rows;
    if (col_bytes)
        for (uint32_t c = 0; c < ncols; ++c) col_bytes[c] = p->h_totals[c];
    return MDC_OK;
}

// One parse of n device-resident bytes.  `ext` == nullptr: the outputs become a new batch (*out)
// and the workspace is the parser's, grown on demand.  Otherwise (the entry point that allocates
// nothing) the outputs are the caller's arrays, the workspace is carved out of this device's static
// scratch array, no cell map is used (every tile takes the full write kernel), and the row count
// and the columns' byte counts are returned through *rows_out / col_bytes.
static int parse_impl(mdc_parser *p, const void *d_data, size_t n, const mdc_outputs *ext, mdc_batch **out,
                      int64_t *rows_out, uint64_t *col_bytes) {
    if ((uintptr_t)d_data & 15) return fail(MDC_ERR_ARG, "d_data must be 16-byte aligned");
    if (n >= ((size_t)1 << 40)) return fail(MDC_ERR_ARG, "input larger than 1 TiB");
    Device *d = p->dev;
    HIP_TRY(hipSetDevice(d->id));
    const uint32_t ncols = (uint32_t)p->ncols, ncnt = ncols + 1;
    const size_t ntiles = (n + TILE - 1) / TILE;
    uint32_t *d_counts, *d_hdr, *d_cellmap = nullptr;
    uint64_t *d_base, *d_pub = nullptr;
    uint8_t *d_long;
    void *d_ptrs_ext = nullptr;
    size_t map_cap = 0, map_stride = 0;
    int rc;
    if (ext) {
        // tile headers | pointer table | prefixes, totals, segment totals | counters | long records | published counts
        const size_t nt = ntiles ? ntiles : 1;
        size_t o = 0;
        const size_t o_hdr = o;
        o += (nt * 8 + 16 + 255) & ~(size_t)255;
        const size_t o_ptrs = o;
        o += (3 * (size_t)ncols * sizeof(void *) + 255) & ~(size_t)255;
        const size_t o_base = o;
        o += ((nt + 1 + scan_segments((uint32_t)nt)) * ncnt * 8 + 255) & ~(size_t)255;
        const size_t o_counts = o;
        o += (nt * ncnt * 4 + 255) & ~(size_t)255;
        const size_t o_long = o;
        o += (nt * long_bytes_per_tile(p->ncols) + 255) & ~(size_t)255;
        if (o > NOALLOC_SCRATCH_BYTES)
            return fail(MDC_ERR_TOO_LARGE,
                        "an input of %zu bytes with %u columns needs %zu bytes of scratch; the static array holds %zu "
                        "(use mdc_parse_device, or parse the input in parts)",
                        n, ncols, o, (size_t)NOALLOC_SCRATCH_BYTES);
        uint8_t *sc = (uint8_t *)noalloc_scratch_addr(d->id);
        if (!sc) return fail(MDC_ERR_INTERNAL, "the static scratch array of device %d was never set up", d->id);
        d_hdr = (uint32_t *)(sc + o_hdr);
        d_ptrs_ext = sc + o_ptrs;
        d_base = (uint64_t *)(sc + o_base);
        d_counts = (uint32_t *)(sc + o_counts);
        d_long = sc + o_long;
        if (p->one_pass && o + nt * ncnt * 8 <= NOALLOC_SCRATCH_BYTES) d_pub = (uint64_t *)(sc + o);  // (else: two passes)
    } else {
        rc = ensure_workspace(p, ntiles ? ntiles : 1);
        if (rc) return rc;
        rc = ensure_cellmap(p, n, ntiles);
        if (rc) return rc;
        d_counts = p->d_counts;
        d_base = p->d_base;
        d_hdr = p->d_hdr;
        d_long = (uint8_t *)p->d_long;
        d_cellmap = p->d_cellmap;
        d_pub = p->d_pub;
        map_cap = p->map_cap;
        map_stride = p->map_stride;
#ifdef MDC_ABLATE
        if (getenv("MDC_NO_MAP")) map_stride = 0;  // every tile through the full write kernel
#endif
    }

    TileParams P{};
    P.data = (const uint8_t *)d_data;
    P.n = n;
    P.schema = (const uint8_t *)p->d_blob;
    P.schema_bytes = (uint32_t)p->blob.size();
    P.ncols = (uint32_t)p->ncols;
    P.words = p->words;
    P.p10_off = ((const SchemaHdr *)p->blob.data())->p10_off;
    P.ntiles = (uint32_t)ntiles;
    P.tile_counts = d_counts;
    P.tile_base = d_base;
    P.status = p->d_status;
    P.cellmap = map_stride ? d_cellmap : nullptr;
    P.map_cap = (uint32_t)map_cap;
    P.map_stride = (uint32_t)map_stride;
    P.tile_hdr = d_hdr;
    P.long_list = (LongRec *)d_long;
    P.long_len = (uint32_t *)(d_long + (ntiles ? ntiles : 1) * sizeof(LongRec));
    P.tmpl = p->d_tmpl;
    P.use_tmpl = p->mode == MODE_XML ? 1u : 0u;
    P.tmpl_sample = p->tmpl_sample;
    P.pub = d_pub;
    P.caps = p->d_caps;
    if (p->flags & MDC_F_RECORDS_ONLY) {  // the record split alone: no template, no cell map (the full write kernel writes the rows)
        P.records_only = 1;
        P.use_tmpl = 0;
        P.cellmap = nullptr;
        P.map_cap = P.map_stride = 0;
    }
    if (p->mark_nl) P.use_tmpl = 0;  // (a template's tag texts hold no newline; every tile takes the table lookup)
    // Records of several shapes (a field that is there in some and not in others): most tiles then hold a record
    // that leaves any template, and the compare they all try first is lost time (a third of the sizing pass).
    // After two such parses in a row the next sixteen go without a template; then it is tried again.
    if (P.use_tmpl && p->tmpl_rest > 0) {
        --p->tmpl_rest;
        P.use_tmpl = 0;
    }
    for (int t : p->typed)
        if (t == MDC_T_TIME || t == MDC_T_F32) P.special_cols = 1;
#ifdef MDC_ABLATE
    if (getenv("MDC_NO_TEMPLATE")) P.use_tmpl = 0;
    if (const char *dbg = getenv("MDC_DEBUG_SKIP"))
        if (atoi(dbg) & 1) P.use_tmpl = 0;  // (no tag search: nothing would follow the template)
    if (const char *dbg = getenv("MDC_DEBUG_SKIP")) P.debug = (uint32_t)atoi(dbg);
#endif
    mdc_timing &t = p->timing;
    t.one_pass = t.one_pass_redone = 0;
    t.lookup_tiles = p->mode == MODE_XML && !P.use_tmpl ? (int)ntiles : 0;  // (without a template every tile takes the lookup)

    // One pass where the previous parse left what it needs: a template the input followed, the first
    // look-behind level, and (for a new batch) the sizes to expect.
    if (p->mode == MODE_XML && p->one_pass && d_pub && ntiles && P.use_tmpl && p->tmpl_valid && p->lb_level == 0 &&
        p->skew_tile < 0 && (ext || p->est_valid)) {
        if (p->fused_hold > 0) {
            --p->fused_hold;
        } else {
            rc = parse_fused(p, P, ext, d_ptrs_ext, out, rows_out, col_bytes);
            if (rc != FUSED_REDO) return rc;
            t.one_pass_redone = 1;
        }
    }

    hipStream_t s = d->stream;
    // Sizing pass.  The look-behind capacity starts at the level that fitted the previous parse.  A
    // record that starts in front of it goes to the long-record kernel; when more than one tile in
    // 64 met such a record, the pass is repeated with the next larger capacity (which costs blocks
    // per CU, so it is not the default) and far fewer records take that slower way.
    uint32_t nlong = 0;
    for (;;) {
        P.L = p->LSs[p->lb_level];
        HIP_TRY(hipMemsetAsync(p->d_status, 0, 16, s));  // status word, tiles that left the template, long records
        HIP_TRY(hipEventRecord(p->ev[0], s));
        if (p->mark_nl) HIP_TRY(launch_mark_overlap(P, s));
        const bool build_tmpl = P.use_tmpl && !p->tmpl_valid && ntiles;
        if (build_tmpl) HIP_TRY(launch_template_kernel(P, s));
        P.schema_bytes = P.p10_off;  // (the sizing pass converts nothing: the schema without the powers of ten)
        HIP_TRY(launch_tile_kernel(0, P, g_no_out, s));
        P.schema_bytes = (uint32_t)p->blob.size();
        if (p->skew_tile >= 0 && (size_t)p->skew_tile < ntiles && p->skew_col >= 0 && p->skew_col < (int)ncols) {
            uint32_t *w = d_counts + (size_t)p->skew_tile * ncnt + p->skew_col, v = 0;
            HIP_TRY(hipStreamSynchronize(s));
            HIP_TRY(hipMemcpy(&v, w, 4, hipMemcpyDeviceToHost));
            v += (uint32_t)p->skew_delta;
            HIP_TRY(hipMemcpy(w, &v, 4, hipMemcpyHostToDevice));
            p->skew_tile = -1;
        }
        HIP_TRY(hipEventRecord(p->ev[1], s));
        // (the scan's last kernel writes the totals and the status words into h_report itself)
        HIP_TRY(launch_scan_tiles(d_counts, d_base, (uint32_t)ntiles, ncnt, p->h_report_dev + REPORT_TOTALS,
                                  (const uint64_t *)p->d_status, (uint32_t)REPORT_TOTALS, s));
        HIP_TRY(hipEventRecord(p->ev[2], s));
        if (build_tmpl) HIP_TRY(hipMemcpyAsync(p->h_totals + ncnt + 1, p->d_tmpl, 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (P.use_tmpl && ntiles) {
            // keep a template that exists and that most tiles followed
            const uint64_t left = p->h_report[0] >> 32;
            if (build_tmpl) p->tmpl_valid = (uint32_t)p->h_totals[ncnt + 1] != 0;
            if (left * 8 > ntiles || (build_tmpl && !p->tmpl_valid)) {
                // few tiles followed (or the sampled record gave no template): the next parse samples
                // another record, in case this one was the odd one out
                p->tmpl_valid = false;
                p->tmpl_sample = (p->tmpl_sample + 1) % 4;
            }
            p->timing.lookup_tiles = (int)left;
            if (ntiles >= 64) {  // (a small input says little: its first tile, with the file's head, never follows)
                p->tmpl_fail = left * 2 > ntiles ? p->tmpl_fail + 1 : 0;
                if (p->tmpl_fail >= 2) {
                    p->tmpl_fail = 0;
                    p->tmpl_rest = 16;
                }
            }
        }
        nlong = (uint32_t)p->h_report[1];
        if ((size_t)nlong * 64 > ntiles && p->lb_level < 2) {
            ++p->lb_level;
            continue;
        }
        if (nlong) {
            // their cells' lengths enter the tiles' counters; then the prefixes and totals again
            HIP_TRY(launch_long_records(false, P, g_no_out, nlong, s));
            HIP_TRY(launch_scan_tiles(d_counts, d_base, (uint32_t)ntiles, ncnt, p->h_report_dev + REPORT_TOTALS,
                                      (const uint64_t *)p->d_status, (uint32_t)REPORT_TOTALS, s));
            HIP_TRY(hipEventRecord(p->ev[2], s));
            HIP_TRY(hipStreamSynchronize(s));
        }
        break;
    }
    const bool unmapped = ((uint32_t)p->h_report[0] & ST_UNMAPPED) != 0;
    rc = status_to_error((uint32_t)p->h_report[0] & ~(ST_NUMERIC_RANGE | ST_UNMAPPED));
    if (rc) return rc;
    const uint64_t rows = p->h_totals[ncols];
    if (n) p->rows_per_byte = (double)rows / (double)n;
    for (uint32_t c = 0; c < ncols; ++c)
        if (p->h_totals[c] > 0xFFFFFFFFull)
            return fail(MDC_ERR_TOO_LARGE, "column %u holds %llu bytes; split the input", c,
                        (unsigned long long)p->h_totals[c]);
    remember_sizes(p, n, rows, p->h_totals);

    mdc_batch *b = nullptr;
    uint64_t *o_rec_off = nullptr;
    uint32_t *o_valid = nullptr, *o_found = nullptr;
    void *d_ptrs = nullptr;
    void **hp = p->h_ptrs;  // the pointer table on its way to the device: off[], bytes[], val[]
    if (ext) {
        // the caller's arrays must hold what the sizing pass counted
        if ((int64_t)rows > ext->rows_cap)
            return fail(MDC_ERR_TOO_LARGE, "%llu records, the output arrays hold %lld", (unsigned long long)rows,
                        (long long)ext->rows_cap);
        for (uint32_t c = 0; c < ncols; ++c) {
            if (p->h_totals[c] > ext->cols[c].bytes_cap)
                return fail(MDC_ERR_TOO_LARGE, "column %u holds %llu bytes, its output array %llu", c,
                            (unsigned long long)p->h_totals[c], (unsigned long long)ext->cols[c].bytes_cap);
            hp[c] = ext->cols[c].d_off;
            hp[ncols + c] = ext->cols[c].d_bytes;
            hp[2 * ncols + c] = p->typed[c] != MDC_T_NONE ? ext->cols[c].d_val : nullptr;
        }
        o_rec_off = (p->flags & MDC_F_NO_RECOFF) ? nullptr : ext->d_rec_off;
        o_valid = ext->d_valid;
        o_found = ext->d_found;
        d_ptrs = d_ptrs_ext;
    } else {
        rc = alloc_batch(p, rows, p->h_totals, &b);
        if (rc) return rc;
        b->rows = (int64_t)rows;
        for (uint32_t c = 0; c < ncols; ++c) {
            b->nbytes[c] = p->h_totals[c];
            hp[c] = b->d_off[c];
            hp[ncols + c] = b->d_bytes[c];
            hp[2 * ncols + c] = b->d_val[c];
        }
        o_rec_off = b->d_rec_off;
        o_valid = b->d_valid;
        o_found = b->d_found;
        d_ptrs = b->d_ptrs;
    }
    const uint64_t bytes_out = output_bytes(p, rows, p->h_totals, o_rec_off != nullptr);
    // from here on a failed runtime call gives the half-built batch back before it returns
#define HIP_TRY_B(expr)                                                                \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) {                                                        \
            batch_release(b);                                                          \
            return fail(MDC_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));          \
        }                                                                              \
    } while (0)
    // the write pass's kernels get the pointers as a kernel argument (a copy of the table to the device
    // would stand between the passes; the second conversion stage, if it runs at all, uploads it)
    OutPtrs O{};
    for (uint32_t c = 0; c < ncols; ++c) {
        O.off[c] = (uint32_t *)hp[c];
        O.bytes[c] = (uint8_t *)hp[ncols + c];
        O.val[c] = (uint64_t *)hp[2 * ncols + c];
    }
    if (ntiles == 0) {
        // nothing to launch: every offset array is the single entry 0
        for (uint32_t c = 0; c < ncols; ++c) HIP_TRY_B(hipMemsetAsync(hp[c], 0, 4, s));
        if (o_rec_off) HIP_TRY_B(hipMemsetAsync(o_rec_off, 0, 8, s));
    }
    P.rec_off = o_rec_off;
    P.off = (uint32_t *const *)d_ptrs;
    P.bytes = (uint8_t *const *)((void **)d_ptrs + ncols);
    P.val = (uint64_t *const *)((void **)d_ptrs + 2 * ncols);
    P.valid = o_valid;
    P.found = o_found;
    // the write pass reports through 32 bytes of the pinned report, one per status bit
    volatile uint8_t *host_flags = (volatile uint8_t *)(p->h_report + REPORT_FLAGS);
    for (uint32_t i = 0; i < 32; ++i) host_flags[i] = 0;
    P.host_flags = (uint8_t *)(p->h_report_dev + REPORT_FLAGS);
    HIP_TRY_B(hipEventRecord(p->ev[4], s));
    // tiles with a complete cell map take the short write kernel, the others the full one
    {
        const LdsLayout &d = p->LEd[p->lb_level], &n = p->LEs[p->lb_level];
        // batches per tile at the input's mean record count: the layout that needs fewer of them
        const uint64_t per_tile = ntiles ? (rows + ntiles - 1) / ntiles : 0;
        const bool dense = d.RB > n.RB && (per_tile + d.RB - 1) / d.RB < (per_tile + n.RB - 1) / n.RB;
        P.L = dense ? d : n;
    }
    hipError_t le = launch_emit_kernel(P, O, s);
    P.L = p->Ls[p->lb_level];
    if (le == hipSuccess && unmapped) {
        P.only_unmapped = 1;
        le = launch_tile_kernel(1, P, O, s);
    }
    if (le == hipSuccess && nlong) le = launch_long_records(true, P, O, nlong, s);
    if (le != hipSuccess) {
        batch_release(b);
        return fail(MDC_ERR_HIP, "launch: %s", hipGetErrorString(le));
    }
    HIP_TRY_B(hipEventRecord(p->ev[3], s));
    // (asked for here, while the write pass runs, not between the passes where the GPU would wait)
    (void)hipEventElapsedTime(&t.count_ms, p->ev[0], p->ev[1]);
    (void)hipEventElapsedTime(&t.scan_ms, p->ev[1], p->ev[2]);
    HIP_TRY_B(hipStreamSynchronize(s));
#undef HIP_TRY_B
    // what the sizing pass reported, and what the write pass raised in the flag bytes
    uint32_t st = (uint32_t)p->h_report[0] & ~ST_UNMAPPED;
    for (uint32_t i = 0; i < 32; ++i)
        if (host_flags[i]) st |= 1u << i;
    if (st & ST_NUMERIC_RANGE) {
        // some floating-point texts need the multi-precision stage
        // (the table from pinned memory that lives until the parser's next call)
        le = hipMemcpyAsync(d_ptrs, hp, 3 * (size_t)ncols * sizeof(void *), hipMemcpyHostToDevice, s);
        if (le == hipSuccess)
            le = launch_long_float((const uint8_t *)p->d_blob, P.off, P.bytes, P.val, o_valid, o_found, rows, ncols, s);
        if (le == hipSuccess) le = hipStreamSynchronize(s);
        if (le != hipSuccess) {
            batch_release(b);
            return fail(MDC_ERR_HIP, "long_float: %s", hipGetErrorString(le));
        }
        st &= ~ST_NUMERIC_RANGE;
    }
    rc = status_to_error(st);
    if (rc) {
        batch_release(b);
        return rc;
    }
    (void)hipEventElapsedTime(&t.write_ms, p->ev[4], p->ev[3]);
    (void)hipEventElapsedTime(&t.total_ms, p->ev[0], p->ev[3]);
    t.bytes_in = n;
    t.bytes_out = bytes_out;
    t.rows = (int64_t)rows;
    t.tiles = (int)ntiles;
    t.full_write = unmapped ? 1 : 0;
    t.long_records = (int)nlong;
    if (out) *out = b;
    if (rows_out) *rows_out = (int64_t)rows;
    if (col_bytes)
        for (uint32_t c = 0; c < ncols; ++c) col_bytes[c] = p->h_totals[c];
    return MDC_OK;
}

extern "C" int mdc_parse_device(mdc_parser *p, const void *d_data, size_t n, mdc_batch **out) {
    if (!p || !out) return fail(MDC_ERR_ARG, "NULL argument");
    *out = nullptr;
    if (n && !d_data) return fail(MDC_ERR_ARG, "d_data is NULL");
    return parse_impl(p, d_data, n, nullptr, out, nullptr, nullptr);
}

extern "C" int mdc_parse_device_noalloc(mdc_parser *p, const void *d_data, size_t n, const mdc_outputs *o, int64_t *rows,
                                        uint64_t *col_bytes) {
    if (!p || !o || !rows) return fail(MDC_ERR_ARG, "NULL argument");
    if (n && !d_data) return fail(MDC_ERR_ARG, "d_data is NULL");
    if (o->rows_cap < 0 || !o->cols || !o->d_valid || !o->d_found) return fail(MDC_ERR_ARG, "incomplete output description");
    for (int c = 0; c < p->ncols; ++c) {
        if (!o->cols[c].d_off || !o->cols[c].d_bytes) return fail(MDC_ERR_ARG, "column %d: NULL output array", c);
        if (p->typed[c] != MDC_T_NONE && !o->cols[c].d_val) return fail(MDC_ERR_ARG, "column %d: typed, but no value array", c);
        // (the cells leave in 16-byte stores at the arena's own 16-byte phase)
        if (((uintptr_t)o->cols[c].d_off & 3) || ((uintptr_t)o->cols[c].d_val & 7) || ((uintptr_t)o->cols[c].d_bytes & 15))
            return fail(MDC_ERR_ARG, "column %d: misaligned output array", c);
    }
    if (((uintptr_t)o->d_valid & 3) || ((uintptr_t)o->d_found & 3) || ((uintptr_t)o->d_rec_off & 7))
        return fail(MDC_ERR_ARG, "misaligned output array");
    // Every call on a GPU works in that GPU's one scratch array: a second call that arrives while one is
    // under way (another thread, another parser) is refused, not queued, since its kernels would overwrite
    // the counters and places the first call's write pass is about to read.
    Device *d = p->dev;
    std::unique_lock<std::mutex> lk(d->noalloc_mu, std::try_to_lock);
    if (!lk.owns_lock())
        return fail(MDC_ERR_BUSY, "another mdc_parse_device_noalloc call is using the static scratch array of device %d", d->id);
    if (!d->scratch_ready) {
        // this device's copy of the array: its address is looked up once, with the device current
        HIP_TRY(hipSetDevice(d->id));
        const hipError_t e = noalloc_scratch_init(d->id);
        if (e != hipSuccess) return fail(MDC_ERR_INTERNAL, "hipGetSymbolAddress of the static scratch array: %s", hipGetErrorString(e));
        d->scratch_ready = true;
    }
    return parse_impl(p, d_data, n, o, nullptr, rows, col_bytes);
}

extern "C" int mdc_internal_pinned_get(size_t want, uint8_t **p, size_t *cap);
extern "C" void mdc_internal_pinned_put(uint8_t *p, size_t cap);

// A host buffer into device memory.  The runtime takes a pageable source through one staging copy
// of its own (about 28 GB/s here); a large buffer is instead cut into 8 MiB pieces that a few host
// threads copy into pinned buffers (two a thread, kept from call to call) and upload on a copy stream
// each, the copy of one piece beside the upload of the one before.
static hipError_t upload_host(Device *d, void *dbuf, const void *h_data, size_t n) {
    constexpr size_t PIECE = (size_t)8 << 20;
    const size_t npieces = (n + PIECE - 1) / PIECE;
    unsigned hw = std::thread::hardware_concurrency();
    size_t T = std::min<size_t>({(size_t)UPLOAD_THREADS, npieces / 2, hw ? hw : 1});
    if (T >= 2) {  // a source that is pinned already goes up as it is
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, h_data) == hipSuccess) {
            if (at.type == hipMemoryTypeHost) T = 1;
        } else {
            (void)hipGetLastError();  // (plain pageable memory is unknown to the runtime: not an error here)
        }
    }
    if (T < 2) {
        hipError_t e = hipMemcpyAsync(dbuf, h_data, n, hipMemcpyHostToDevice, d->stream);
        return e == hipSuccess ? hipStreamSynchronize(d->stream) : e;
    }
    std::lock_guard<std::mutex> one(d->up_mu);
    {
        std::lock_guard<std::mutex> lk(d->mu);
        for (int t = 0; t < UPLOAD_THREADS; ++t) {
            if (!d->up_stream[t]) {
                hipError_t e = hipStreamCreateWithFlags(&d->up_stream[t], hipStreamNonBlocking);
                if (e != hipSuccess) return e;
            }
            for (int b = 0; b < 2; ++b)
                if (!d->up_event[2 * t + b]) {
                    hipError_t e = hipEventCreateWithFlags(&d->up_event[2 * t + b], hipEventDisableTiming);
                    if (e != hipSuccess) return e;
                }
        }
    }
    hipError_t errs[UPLOAD_THREADS];
    auto work = [&](size_t t) {
        hipError_t e = hipSetDevice(d->id);
        uint8_t *buf[2] = {nullptr, nullptr};
        size_t cap[2] = {0, 0};
        for (int b = 0; b < 2 && e == hipSuccess; ++b)
            if (!mdc_internal_pinned_get(PIECE, &buf[b], &cap[b])) e = hipErrorOutOfMemory;
        size_t j = 0;
        for (size_t k = t; k < npieces && e == hipSuccess; k += T, ++j) {
            const int b = (int)(j & 1);
            if (j >= 2) e = hipEventSynchronize(d->up_event[2 * t + b]);  // the piece this buffer held is up
            if (e != hipSuccess) break;
            const size_t off = k * PIECE, len = n - off < PIECE ? n - off : PIECE;
            memcpy(buf[b], (const uint8_t *)h_data + off, len);
            e = hipMemcpyAsync((uint8_t *)dbuf + off, buf[b], len, hipMemcpyHostToDevice, d->up_stream[t]);
            if (e == hipSuccess) e = hipEventRecord(d->up_event[2 * t + b], d->up_stream[t]);
        }
        const hipError_t e2 = hipStreamSynchronize(d->up_stream[t]);  // (also before the buffers go back)
        for (int b = 0; b < 2; ++b)
            if (buf[b]) mdc_internal_pinned_put(buf[b], cap[b]);
        errs[t] = e != hipSuccess ? e : e2;
    };
    std::vector<std::thread> th;
    size_t started = 1;
    try {  // (a thread that cannot be started must not throw through the C boundary)
        for (size_t t = 1; t < T; ++t, ++started) th.emplace_back(work, t);
    } catch (const std::system_error &) {
    }
    work(0);
    for (auto &x : th) x.join();
    for (size_t t = 0; t < started; ++t)
        if (errs[t] != hipSuccess) return errs[t];
    if (started < T) {
        // the pieces of the threads that never ran: one plain copy each (piece k belongs to thread k % T)
        for (size_t k = 0; k < npieces; ++k) {
            if (k % T < started) continue;
            const size_t off = k * PIECE, len = n - off < PIECE ? n - off : PIECE;
            hipError_t e = hipMemcpyAsync((uint8_t *)dbuf + off, (const uint8_t *)h_data + off, len, hipMemcpyHostToDevice, d->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
            if (e != hipSuccess) return e;
        }
    }
    return hipSuccess;
}

extern "C" int mdc_parse_host(mdc_parser *p, const void *h_data, size_t n, mdc_batch **out) {
    if (!p || !out) return fail(MDC_ERR_ARG, "NULL argument");
    *out = nullptr;
    if (n && !h_data) return fail(MDC_ERR_ARG, "h_data is NULL");
    Device *d = p->dev;
    HIP_TRY(hipSetDevice(d->id));
    void *dbuf = nullptr;
    int rc = dev_alloc(d, n + 64, &dbuf);
    if (rc) return rc;
    if (n) {
        const hipError_t e = upload_host(d, dbuf, h_data, n);
        if (e != hipSuccess) {
            dev_free(d, dbuf);
            return fail(MDC_ERR_HIP, "H2D copy: %s", hipGetErrorString(e));
        }
    }
    rc = mdc_parse_device(p, dbuf, n, out);
    dev_free(d, dbuf);
    return rc;
}

// used by mdc_stream.cpp, which is built on the public entry points
extern "C" int mdc_internal_fail(int code, const char *msg) { return fail(code, "%s", msg); }
extern "C" int mdc_internal_parser_device(const mdc_parser *p) { return p->dev->id; }
// Test hook: the next parse on `p` adds `delta` to what its sizing pass counted for column `col` of
// tile `tile`, as a sizing pass that disagreed with the write pass would have left it.  The write
// pass must answer with MDC_ERR_INTERNAL and store nothing outside what was sized.
// (Refused unless MDC_TEST_HOOKS is set in the environment when a hook is first called.)
static bool test_hooks_on() {
    static const bool on = getenv("MDC_TEST_HOOKS") != nullptr;
    return on;
}
// Test hook: how many polls a tile of the one-pass parse may spend waiting for one tile in front of it.  With 0 a
// tile that would have to wait gives up at once: the parse is then repeated in two passes, and no wave waits.
extern "C" int mdc_internal_spin_limit(mdc_parser *p, unsigned polls) {
    if (!test_hooks_on()) return fail(MDC_ERR_ARG, "test hooks are off (MDC_TEST_HOOKS is not set)");
    if (!p) return fail(MDC_ERR_ARG, "NULL argument");
    p->spin_limit = polls;
    return MDC_OK;
}
extern "C" int mdc_internal_skew_counts(mdc_parser *p, int tile, int col, int delta) {
    if (!test_hooks_on()) return fail(MDC_ERR_ARG, "test hooks are off (MDC_TEST_HOOKS is not set)");
    if (!p) return fail(MDC_ERR_ARG, "NULL argument");
    p->skew_tile = tile;
    p->skew_col = col;
    p->skew_delta = delta;
    return MDC_OK;
}

extern "C" int mdc_parse_file(mdc_parser *p, const char *path, mdc_batch **out) {
    if (!p || !out || !path) return fail(MDC_ERR_ARG, "NULL argument");
    *out = nullptr;
    // a stream of one file: same reader (plain or gzip), same pinned upload
    mdc_stream *st = nullptr;
    int rc = mdc_stream_open(p, &path, 1, 1, &st);
    if (rc) return rc;
    rc = mdc_stream_next(st, out, nullptr);
    mdc_stream_close(st);
    return rc;
}

#ifdef MDC_PHASE_TIMING
// diagnostic build only: cycle counters behind the status word (see STAMP in the kernels)
extern "C" int mdc_debug_phase_counters(mdc_parser *p, unsigned long long *out, int n, int reset) {
    HIP_TRY(hipSetDevice(p->dev->id));
    HIP_TRY(hipMemcpy(out, (unsigned long long *)p->d_status + 2, (size_t)n * 8, hipMemcpyDeviceToHost));
    if (reset) HIP_TRY(hipMemset((unsigned long long *)p->d_status + 2, 0, (size_t)n * 8));
    return MDC_OK;
}
#endif

extern "C" int mdc_parser_last_timing(const mdc_parser *p, mdc_timing *t) {
    if (!p || !t) return fail(MDC_ERR_ARG, "NULL argument");
    *t = p->timing;
    return MDC_OK;
}

extern "C" int mdc_parse_xml_buffer(const void *h_data, size_t n, const mdc_col *cols, int ncols,
                                    const mdc_opts *opts, mdc_batch **out) {
    mdc_parser *p = nullptr;
    int rc = mdc_parser_create_xml(cols, ncols, opts, &p);
    if (rc) return rc;
    rc = mdc_parse_host(p, h_data, n, out);
    mdc_parser_destroy(p);
    return rc;
}

extern "C" int mdc_parse_xml_file(const char *path, const mdc_col *cols, int ncols, const mdc_opts *opts,
                                  mdc_batch **out) {
    mdc_parser *p = nullptr;
    int rc = mdc_parser_create_xml(cols, ncols, opts, &p);
    if (rc) return rc;
    rc = mdc_parse_file(p, path, out);
    mdc_parser_destroy(p);
    return rc;
}

extern "C" int mdc_parse_csv_buffer(const void *h_data, size_t n, const char *sep, int bdelspace, int nfields,
                                    const int *ilen, const mdc_opts *opts, mdc_batch **out) {
    mdc_parser *p = nullptr;
    int rc = mdc_parser_create_csv(sep, bdelspace, nfields, ilen, opts, &p);
    if (rc) return rc;
    rc = mdc_parse_host(p, h_data, n, out);
    mdc_parser_destroy(p);
    return rc;
}

// ------------------------------------------------------------------------------------------
// batch accessors

extern "C" int64_t mdc_batch_rows(const mdc_batch *b) { return b ? b->rows : -1; }
extern "C" int mdc_batch_ncols(const mdc_batch *b) { return b ? b->ncols : -1; }
extern "C" int mdc_batch_words_per_row(const mdc_batch *b) { return b ? (int)b->words : -1; }

#define CHECK_COL(b, col)                                                   \
    if (!(b)) return fail(MDC_ERR_ARG, "batch is NULL");                    \
    if ((col) < 0 || (col) >= (b)->ncols) return fail(MDC_ERR_ARG, "column %d out of range", (col))

extern "C" int mdc_batch_device_column(const mdc_batch *b, int col, const uint32_t **d_off,
                                       const uint8_t **d_bytes, uint64_t *nbytes) {
    CHECK_COL(b, col);
    if (d_off) *d_off = b->d_off[col];
    if (d_bytes) *d_bytes = b->d_bytes[col];
    if (nbytes) *nbytes = b->nbytes[col];
    return MDC_OK;
}

extern "C" int mdc_batch_device_values(const mdc_batch *b, int col, const uint64_t **d_val) {
    CHECK_COL(b, col);
    if (!b->d_val[col]) return fail(MDC_ERR_ARG, "column %d has no typed values", col);
    if (d_val) *d_val = b->d_val[col];
    return MDC_OK;
}

extern "C" int mdc_batch_device_masks(const mdc_batch *b, const uint32_t **d_valid, const uint32_t **d_found) {
    if (!b) return fail(MDC_ERR_ARG, "batch is NULL");
    if (d_valid) *d_valid = b->d_valid;
    if (d_found) *d_found = b->d_found;
    return MDC_OK;
}

extern "C" int mdc_batch_device_recoff(const mdc_batch *b, const uint64_t **d_rec_off) {
    if (!b) return fail(MDC_ERR_ARG, "batch is NULL");
    if (!b->d_rec_off) return fail(MDC_ERR_ARG, "record offsets were not requested");
    if (d_rec_off) *d_rec_off = b->d_rec_off;
    return MDC_OK;
}

static int d2h(Device *d, void *dst, const void *src, size_t n) {
    if (!n) return MDC_OK;
    HIP_TRY(hipSetDevice(d->id));
    HIP_TRY(hipMemcpy(dst, src, n, hipMemcpyDeviceToHost));
    return MDC_OK;
}

extern "C" int mdc_batch_host_column(mdc_batch *b, int col, const uint32_t **off, const uint8_t **bytes,
                                     uint64_t *nbytes) {
    CHECK_COL(b, col);
    if (!b->have_col[col]) {
        b->h_off[col].resize((size_t)b->rows + 1);
        b->h_bytes[col].resize((size_t)b->nbytes[col] + 1);
        int rc = d2h(b->dev, b->h_off[col].data(), b->d_off[col], ((size_t)b->rows + 1) * 4);
        if (!rc) rc = d2h(b->dev, b->h_bytes[col].data(), b->d_bytes[col], (size_t)b->nbytes[col]);
        if (rc) return rc;
        b->have_col[col] = true;
    }
    if (off) *off = b->h_off[col].data();
    if (bytes) *bytes = b->h_bytes[col].data();
    if (nbytes) *nbytes = b->nbytes[col];
    return MDC_OK;
}

extern "C" int mdc_batch_host_values(mdc_batch *b, int col, const uint64_t **val) {
    CHECK_COL(b, col);
    if (!b->d_val[col]) return fail(MDC_ERR_ARG, "column %d has no typed values", col);
    if (!b->have_val[col]) {
        b->h_val[col].resize((size_t)b->rows + 1);
        int rc = d2h(b->dev, b->h_val[col].data(), b->d_val[col], (size_t)b->rows * 8);
        if (rc) return rc;
        b->have_val[col] = true;
    }
    if (val) *val = b->h_val[col].data();
    return MDC_OK;
}

extern "C" int mdc_batch_host_masks(mdc_batch *b, const uint32_t **valid, const uint32_t **found) {
    if (!b) return fail(MDC_ERR_ARG, "batch is NULL");
    if (!b->have_masks) {
        size_t cnt = (size_t)b->rows * b->words;
        b->h_valid.resize(cnt + 1);
        b->h_found.resize(cnt + 1);
        int rc = d2h(b->dev, b->h_valid.data(), b->d_valid, cnt * 4);
        if (!rc) rc = d2h(b->dev, b->h_found.data(), b->d_found, cnt * 4);
        if (rc) return rc;
        b->have_masks = true;
    }
    if (valid) *valid = b->h_valid.data();
    if (found) *found = b->h_found.data();
    return MDC_OK;
}

extern "C" int mdc_batch_host_recoff(mdc_batch *b, const uint64_t **rec_off) {
    if (!b) return fail(MDC_ERR_ARG, "batch is NULL");
    if (!b->d_rec_off) return fail(MDC_ERR_ARG, "record offsets were not requested");
    if (!b->have_recoff) {
        b->h_rec_off.resize((size_t)b->rows + 1);
        int rc = d2h(b->dev, b->h_rec_off.data(), b->d_rec_off, ((size_t)b->rows + 1) * 8);
        if (rc) return rc;
        b->have_recoff = true;
    }
    if (rec_off) *rec_off = b->h_rec_off.data();
    return MDC_OK;
}

extern "C" int mdc_batch_cell(mdc_batch *b, int64_t row, int col, const char **p, uint32_t *len) {
    CHECK_COL(b, col);
    if (row < 0 || row >= b->rows) return fail(MDC_ERR_ARG, "row out of range");
    const uint32_t *off;
    const uint8_t *bytes;
    int rc = mdc_batch_host_column(b, col, &off, &bytes, nullptr);
    if (rc) return rc;
    if (p) *p = (const char *)bytes + off[row];
    if (len) *len = off[row + 1] - off[row];
    return MDC_OK;
}

extern "C" int mdc_batch_column_fixed(mdc_batch *b, int col, const char **base, size_t *stride) {
    CHECK_COL(b, col);
    if (b->h_fixed[col].empty()) {
        const uint32_t *off;
        int rc = mdc_batch_host_column(b, col, &off, nullptr, nullptr);
        if (rc) return rc;
        size_t st;
        if (b->collen[col] > 0) {
            st = (size_t)b->collen[col] + 1;
        } else {
            uint32_t mx = 0;
            for (int64_t r = 0; r < b->rows; ++r) mx = off[r + 1] - off[r] > mx ? off[r + 1] - off[r] : mx;
            st = (size_t)mx + 1;
        }
        if (st > 0xFFFFFFFFull) return fail(MDC_ERR_TOO_LARGE, "stride too large");
        size_t total = (size_t)b->rows * st;
        b->h_fixed[col].assign(total + 1, 0);
        if (total) {
            void *slab = nullptr;
            rc = dev_alloc(b->dev, total, &slab);
            if (rc) return rc;
            hipError_t e = launch_fixed_stride(b->d_off[col], b->d_bytes[col], (uint8_t *)slab, (uint64_t)b->rows,
                                               (uint32_t)st, b->dev->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(b->dev->stream);
            if (e == hipSuccess) e = hipMemcpy(b->h_fixed[col].data(), slab, total, hipMemcpyDeviceToHost);
            dev_free(b->dev, slab);
            if (e != hipSuccess) return fail(MDC_ERR_HIP, "fixed-stride expansion: %s", hipGetErrorString(e));
        }
        b->fixed_stride[col] = st;
    }
    if (base) *base = b->h_fixed[col].data();
    if (stride) *stride = b->fixed_stride[col];
    return MDC_OK;
}

// ------------------------------------------------------------------------------------------
// writer side: columns -> text (include/mdc_format.h)

struct mdc_formatter {
    Device *dev = nullptr;
    int ncols = 0;
    std::vector<uint8_t> blob;
    uint32_t pre_plain_off = 0, pre_open_off = 0;
    uint32_t fixed = 0;          // bytes per record that are not values
    uint32_t lds_budget = 22 * 1024 + 512; // seven blocks of the formatting kernel per compute unit
    void *d_blob = nullptr;
    void *d_tab = nullptr;       // ncols off pointers, ncols byte pointers, ncols arena sizes
    void *h_tab = nullptr;       // pinned image of d_tab
    uint64_t *d_info = nullptr;  // prep sums (3 words), status word
    uint64_t *h_info = nullptr;  // pinned
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    mdc_format_timing timing{};
};

struct mdc_text {
    Device *dev = nullptr;
    uint8_t *d_data = nullptr;
    uint64_t nbytes = 0;
    int64_t maxcount = 0;
    std::vector<uint64_t> file_off; // files + 1 entries
    std::vector<uint8_t> h_data;
    bool have_host = false;
};

extern "C" int mdc_formatter_create(const char *const *fieldname, const int *fieldlen, int nfields, int device,
                                    mdc_formatter **out) {
    if (!out) return fail(MDC_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (!fieldname || nfields < 1 || nfields > 256) return fail(MDC_ERR_ARG, "1..256 fields required");
    size_t text = 0;
    for (int c = 0; c < nfields; ++c) {
        if (!fieldname[c]) return fail(MDC_ERR_ARG, "field %d: name is NULL", c);
        size_t nl = strlen(fieldname[c]);
        if (nl < 1 || nl > 255) return fail(MDC_ERR_ARG, "field %d: name of %zu bytes (1..255 allowed)", c, nl);
        if (fieldlen && fieldlen[c] < 0) return fail(MDC_ERR_ARG, "field %d: negative fieldlen", c);
        text += 2 * nl + 5;
    }
    if (text > 12 * 1024) return fail(MDC_ERR_ARG, "%zu bytes of tag text (12 KiB allowed)", text);
    Device *d;
    int rc = get_device(device, &d);
    if (rc) return rc;

    mdc_formatter *f = new mdc_formatter();
    f->dev = d;
    f->ncols = nfields;
    f->fixed = (uint32_t)text + 8;
    f->blob.assign((sizeof(FormatField) * nfields + 15) & ~size_t(15), 0);
    auto add_text = [&](const std::string &t) { // 16-aligned, zero padded
        uint32_t off = (uint32_t)f->blob.size();
        f->blob.insert(f->blob.end(), t.begin(), t.end());
        f->blob.resize((f->blob.size() + 15) & ~size_t(15), 0);
        return off;
    };
    for (int c = 0; c < nfields; ++c) {
        FormatField ff{};
        ff.name_len = (uint32_t)strlen(fieldname[c]);
        ff.fieldlen = fieldlen ? (uint32_t)fieldlen[c] : 0;
        std::string t = std::string("</") + fieldname[c] + ">";
        t += c + 1 < nfields ? std::string("<") + fieldname[c + 1] + ">" : std::string("<endl/>\n");
        ff.join_len = (uint32_t)t.size();
        if (c + 1 == nfields) t += "</data>\n";
        ff.text_off = add_text(t);
        memcpy(f->blob.data() + sizeof(FormatField) * c, &ff, sizeof ff);
    }
    f->pre_plain_off = add_text(std::string("<") + fieldname[0] + ">");
    f->pre_open_off = add_text(std::string("<data>\n<") + fieldname[0] + ">");
    f->blob.resize(f->blob.size() + 16, 0);
    auto bail = [&](int code) {
        mdc_formatter_destroy(f);
        return code;
    };
    if (hipSetDevice(d->id) != hipSuccess) return bail(fail(MDC_ERR_HIP, "hipSetDevice failed"));
    if ((rc = dev_alloc(d, f->blob.size(), &f->d_blob))) return bail(rc);
    if ((rc = dev_alloc(d, (size_t)nfields * 24, &f->d_tab))) return bail(rc);
    if ((rc = dev_alloc(d, 64, (void **)&f->d_info))) return bail(rc);
    hipError_t e = hipHostMalloc(&f->h_tab, (size_t)nfields * 24);
    if (e == hipSuccess) e = hipHostMalloc((void **)&f->h_info, 64);
    for (int i = 0; i < 3 && e == hipSuccess; ++i) e = hipEventCreate(&f->ev[i]);
    if (e == hipSuccess) e = hipMemcpy(f->d_blob, f->blob.data(), f->blob.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) return bail(fail(MDC_ERR_HIP, "formatter set-up: %s", hipGetErrorString(e)));
    *out = f;
    return MDC_OK;
}

extern "C" void mdc_formatter_destroy(mdc_formatter *f) {
    if (!f) return;
    if (f->dev) {
        (void)hipSetDevice(f->dev->id);
        dev_free(f->dev, f->d_blob);
        dev_free(f->dev, f->d_tab);
        dev_free(f->dev, f->d_info);
    }
    if (f->h_tab) (void)hipHostFree(f->h_tab);
    if (f->h_info) (void)hipHostFree(f->h_info);
    for (auto &e : f->ev)
        if (e) (void)hipEventDestroy(e);
    delete f;
}

// tuning hook for the measurement tools: LDS bytes a block of the formatting kernel may take
extern "C" int mdc_internal_format_budget(mdc_formatter *f, int bytes) {
    if (!test_hooks_on()) return fail(MDC_ERR_ARG, "test hooks are off (MDC_TEST_HOOKS is not set)");
    if (!f || bytes < 8 * 1024 || bytes > 60 * 1024) return fail(MDC_ERR_ARG, "budget out of range");
    f->lds_budget = (uint32_t)bytes;
    return MDC_OK;
}

// Rows per block and the LDS regions: the largest power of two whose rows' text is expected to
// fit one staging window within the budget (longer rows go through several windows).
static void format_layout(const mdc_formatter *f, uint64_t rows, uint64_t total, FormatParams &P) {
    const uint32_t nc = (uint32_t)f->ncols;
    const uint32_t head = (uint32_t)f->blob.size() + 16 * nc;
    const uint32_t budget = std::max(f->lds_budget, head + 12 * 1024);
    const uint64_t avg = rows ? (total + rows - 1) / rows : 1;
    uint32_t R = FMT_BLOCK, shift = 8;
    auto tables = [&](uint32_t r) { return round16((2 * nc + 1) * r * 4) + round16(r * 9) + 32; };
    while (R > 1) {
        uint64_t want = (avg * R * 9 / 8 + 48 + 15) & ~uint64_t(15);
        if (head + tables(R) + want <= budget) break;
        R >>= 1;
        --shift;
    }
    P.R = R;
    P.Rshift = shift;
    P.lds_ptrs = (uint32_t)f->blob.size();
    P.lds_cpos = head;
    P.lds_src = P.lds_cpos + (nc + 1) * R * 4;
    P.lds_rstart = round16(P.lds_src + nc * R * 4);
    P.lds_rflag = P.lds_rstart + R * 8;
    P.lds_misc = round16(P.lds_rflag + R);
    P.lds_stage = P.lds_misc + 32;
    uint32_t room = budget > P.lds_stage ? (budget - P.lds_stage) & ~15u : 0;
    P.stage_cap = std::max(room, 4096u);
    P.lds_total = P.lds_stage + P.stage_cap;
}

static int format_columns(mdc_formatter *f, const uint32_t *const *d_off, const uint8_t *const *d_bytes,
                          const uint64_t *nbytes, int64_t rows, int64_t maxcount, mdc_text **out) {
    Device *d = f->dev;
    const uint32_t nc = (uint32_t)f->ncols;
    HIP_TRY(hipSetDevice(d->id));
    mdc_text *t = new mdc_text();
    t->dev = d;
    t->maxcount = maxcount;
    t->file_off.assign(1, 0);
    f->timing = mdc_format_timing{};
    if (rows == 0) {
        *out = t;
        return MDC_OK;
    }
    auto bail = [&](int code) {
        mdc_text_free(t);
        return code;
    };
#define FMT_TRY(expr)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return bail(fail(MDC_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_))); \
    } while (0)

    // the columns' pointers, then the sums of their first and last offsets
    const void **tab = (const void **)f->h_tab;
    for (uint32_t c = 0; c < nc; ++c) {
        tab[c] = d_off[c];
        tab[nc + c] = d_bytes[c];
        ((uint64_t *)f->h_tab)[2 * nc + c] = nbytes ? nbytes[c] : 0;
    }
    hipStream_t st = d->stream;
    FMT_TRY(hipEventRecord(f->ev[0], st));
    FMT_TRY(hipMemcpyAsync(f->d_tab, f->h_tab, (size_t)nc * 24, hipMemcpyHostToDevice, st));
    FMT_TRY(hipMemsetAsync(f->d_info, 0, 64, st));
    const uint32_t *const *t_off = (const uint32_t *const *)f->d_tab;
    const uint8_t *const *t_bytes = (const uint8_t *const *)((const void **)f->d_tab + nc);
    const uint64_t *t_nbytes = nbytes ? (const uint64_t *)f->d_tab + 2 * nc : nullptr;
    FMT_TRY(launch_format_prep(t_off, t_nbytes, nc, (uint64_t)rows, f->d_info, st));
    FMT_TRY(hipMemcpyAsync(f->h_info, f->d_info, 32, hipMemcpyDeviceToHost, st));
    FMT_TRY(hipStreamSynchronize(st));
    if (f->h_info[2])
        return bail(fail(MDC_ERR_ARG, "a column's offsets descend or pass the end of its arena"));
    const uint64_t cells = f->h_info[0] - f->h_info[1];
    const uint64_t nfiles = maxcount > 0 ? ((uint64_t)rows + maxcount - 1) / (uint64_t)maxcount : 1;
    uint64_t total = (uint64_t)rows * f->fixed + cells + 15 * nfiles;

    FormatParams P{};
    P.off = t_off;
    P.bytes = t_bytes;
    P.blob = (const uint8_t *)f->d_blob;
    P.blob_bytes = (uint32_t)f->blob.size();
    P.pre_plain_off = f->pre_plain_off;
    P.pre_open_off = f->pre_open_off;
    P.ncols = nc;
    P.fixed = f->fixed;
    P.rows = (uint64_t)rows;
    P.maxcount = (uint64_t)maxcount;
    P.sumbase = f->h_info[1];
    P.total = total;
    P.status = (uint32_t *)(f->d_info + 3);
#ifdef MDC_ABLATE
    if (const char *dbg = getenv("MDC_FORMAT_SKIP")) P.debug = (uint32_t)atoi(dbg);
#endif
    format_layout(f, P.rows, total, P);
    if (((P.rows + P.R - 1) >> P.Rshift) > 0x7fffffffull)
        return bail(fail(MDC_ERR_TOO_LARGE, "%lld rows in one call", (long long)rows));

    int rc;
    void *d_fo = nullptr, *d_elen = nullptr, *d_len = nullptr, *d_pos = nullptr, *d_part = nullptr;
    if ((rc = dev_alloc(d, total + 64, (void **)&t->d_data))) return bail(rc);
    if ((rc = dev_alloc(d, (nfiles + 1) * 8, &d_fo))) return bail(rc);
    auto release = [&]() {
        dev_free(d, d_fo);
        dev_free(d, d_elen);
        dev_free(d, d_len);
        dev_free(d, d_pos);
        dev_free(d, d_part);
    };
    auto bail2 = [&](int code) {
        release();
        return bail(code);
    };
#define FMT_TRY2(expr)                                                                            \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return bail2(fail(MDC_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_))); \
    } while (0)
    P.out = t->d_data;
    P.file_off = (uint64_t *)d_fo;

    FMT_TRY2(hipEventRecord(f->ev[1], st));
    FMT_TRY2(launch_format(false, P, st));
    FMT_TRY2(hipEventRecord(f->ev[2], st));
    FMT_TRY2(hipMemcpyAsync(f->h_info, f->d_info, 32, hipMemcpyDeviceToHost, st));
    FMT_TRY2(hipStreamSynchronize(st));
    uint32_t status = (uint32_t)f->h_info[3];
    int exact = 0;
#ifdef MDC_ABLATE
    if (P.debug) status &= ~FS_NEED_EXACT; // an ablated kernel writes zeros where it skipped the values
#endif
    if (!(status & FS_BAD_OFFSETS) && (status & FS_NEED_EXACT)) {
        // some value is cut: measure every value, scan the record lengths, format again
        exact = 1;
        const uint32_t np = format_scan_parts(P.rows);
        if ((rc = dev_alloc(d, (size_t)nc * P.rows * 4, &d_elen))) return bail2(rc);
        if ((rc = dev_alloc(d, P.rows * 8, &d_len))) return bail2(rc);
        if ((rc = dev_alloc(d, (P.rows + 1) * 8, &d_pos))) return bail2(rc);
        if ((rc = dev_alloc(d, (size_t)np * 8, &d_part))) return bail2(rc);
        FMT_TRY2(hipMemsetAsync(f->d_info, 0, 64, st));
        FMT_TRY2(launch_format_measure(P, (uint32_t *)d_elen, (uint64_t *)d_len, (uint64_t *)d_part,
                                       (uint64_t *)d_pos, st));
        FMT_TRY2(hipMemcpyAsync(f->h_info, (uint64_t *)d_pos + P.rows, 8, hipMemcpyDeviceToHost, st));
        FMT_TRY2(hipStreamSynchronize(st));
        total = f->h_info[0] + 15 * nfiles;
        if (total > P.total) return bail2(fail(MDC_ERR_INTERNAL, "measured text longer than its bound"));
        P.total = total;
        P.elen = (const uint32_t *)d_elen;
        P.recpos = (const uint64_t *)d_pos;
        FMT_TRY2(hipEventRecord(f->ev[1], st));
        FMT_TRY2(launch_format(true, P, st));
        FMT_TRY2(hipEventRecord(f->ev[2], st));
        FMT_TRY2(hipMemcpyAsync(f->h_info, f->d_info, 32, hipMemcpyDeviceToHost, st));
        FMT_TRY2(hipStreamSynchronize(st));
        status = (uint32_t)f->h_info[3];
    }
    if (status & FS_BAD_OFFSETS) return bail2(fail(MDC_ERR_ARG, "a column's offsets descend"));
    t->nbytes = total;
    t->file_off.assign(nfiles + 1, 0);
    FMT_TRY2(hipMemcpy(t->file_off.data(), d_fo, nfiles * 8, hipMemcpyDeviceToHost));
    t->file_off[nfiles] = total;
    release();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, f->ev[1], f->ev[2]);
    f->timing.format_ms = ms;
    (void)hipEventElapsedTime(&ms, f->ev[0], f->ev[2]);
    f->timing.total_ms = ms;
    f->timing.bytes_in = cells + (uint64_t)nc * ((uint64_t)rows + 1) * 4;
    f->timing.bytes_out = total;
    f->timing.exact_pass = exact;
    *out = t;
    return MDC_OK;
#undef FMT_TRY
#undef FMT_TRY2
}

extern "C" int mdc_format_device(mdc_formatter *f, const uint32_t *const *d_off, const uint8_t *const *d_bytes,
                                 int64_t rows, int64_t maxcount, mdc_text **out) {
    if (!f || !out) return fail(MDC_ERR_ARG, "NULL argument");
    *out = nullptr;
    if (rows < 0 || maxcount < 0) return fail(MDC_ERR_ARG, "negative rows or maxcount");
    if (rows && (!d_off || !d_bytes)) return fail(MDC_ERR_ARG, "column arrays are NULL");
    for (int c = 0; rows && c < f->ncols; ++c)
        if (!d_off[c] || !d_bytes[c]) return fail(MDC_ERR_ARG, "column %d: NULL pointer", c);
    return format_columns(f, d_off, d_bytes, nullptr, rows, maxcount, out);
}

extern "C" int mdc_format_batch(mdc_formatter *f, const mdc_batch *b, int64_t maxcount, mdc_text **out) {
    if (!f || !b || !out) return fail(MDC_ERR_ARG, "NULL argument");
    *out = nullptr;
    if (maxcount < 0) return fail(MDC_ERR_ARG, "negative maxcount");
    if (b->ncols != f->ncols)
        return fail(MDC_ERR_ARG, "batch has %d columns, formatter %d fields", b->ncols, f->ncols);
    if (b->dev != f->dev) return fail(MDC_ERR_ARG, "batch and formatter live on different devices");
    std::vector<const uint32_t *> off(b->d_off.begin(), b->d_off.end());
    std::vector<const uint8_t *> bytes(b->d_bytes.begin(), b->d_bytes.end());
    return format_columns(f, off.data(), bytes.data(), b->nbytes.data(), b->rows, maxcount, out);
}

extern "C" int mdc_format_host(mdc_formatter *f, const uint32_t *const *off, const uint8_t *const *bytes,
                               int64_t rows, int64_t maxcount, mdc_text **out) {
    if (!f || !out) return fail(MDC_ERR_ARG, "NULL argument");
    *out = nullptr;
    if (rows < 0 || maxcount < 0) return fail(MDC_ERR_ARG, "negative rows or maxcount");
    if (rows && (!off || !bytes)) return fail(MDC_ERR_ARG, "column arrays are NULL");
    Device *d = f->dev;
    HIP_TRY(hipSetDevice(d->id));
    const int nc = f->ncols;
    std::vector<void *> d_off(nc, nullptr), d_bytes(nc, nullptr);
    std::vector<uint64_t> nbytes(nc, 0);
    auto release = [&]() {
        for (int c = 0; c < nc; ++c) {
            dev_free(d, d_off[c]);
            dev_free(d, d_bytes[c]);
        }
    };
    int rc = MDC_OK;
    for (int c = 0; rows && c < nc && rc == MDC_OK; ++c) {
        if (!off[c] || (!bytes[c] && off[c][rows])) {
            rc = fail(MDC_ERR_ARG, "column %d: NULL pointer", c);
            break;
        }
        nbytes[c] = off[c][rows];
        if ((rc = dev_alloc(d, ((size_t)rows + 1) * 4, &d_off[c]))) break;
        if ((rc = dev_alloc(d, nbytes[c] + 64, &d_bytes[c]))) break;
        hipError_t e = hipMemcpyAsync(d_off[c], off[c], ((size_t)rows + 1) * 4, hipMemcpyHostToDevice, d->stream);
        if (e == hipSuccess && nbytes[c])
            e = hipMemcpyAsync(d_bytes[c], bytes[c], nbytes[c], hipMemcpyHostToDevice, d->stream);
        if (e != hipSuccess) rc = fail(MDC_ERR_HIP, "H2D copy: %s", hipGetErrorString(e));
    }
    if (rc == MDC_OK && hipStreamSynchronize(d->stream) != hipSuccess) rc = fail(MDC_ERR_HIP, "H2D copy failed");
    if (rc == MDC_OK)
        rc = format_columns(f, (const uint32_t *const *)d_off.data(), (const uint8_t *const *)d_bytes.data(),
                            nbytes.data(), rows, maxcount, out);
    release();
    return rc;
}

// ---- timetostr in bulk ------------------------------------------------------------------
// the formats of the reference's timetostr (public/_public.cpp:639-719), each as the bytes it
// takes from "yyyy-mm-dd hh:mi:ss"
static bool time_format(const char *fmt, mdc::TimeFmt *F) {
    static const uint8_t dashed[19] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18};
    static const uint8_t packed[14] = {0, 1, 2, 3, 5, 6, 8, 9, 11, 12, 14, 15, 17, 18};
    static const struct {
        const char *name;
        const uint8_t *idx;
        uint32_t from, width;
    } table[] = {
        {"", dashed, 0, 19},
        {"yyyy-mm-dd hh24:mi:ss", dashed, 0, 19},
        {"yyyy-mm-dd hh24:mi", dashed, 0, 16},
        {"yyyy-mm-dd hh24", dashed, 0, 13},
        {"yyyy-mm-dd", dashed, 0, 10},
        {"yyyy-mm", dashed, 0, 7},
        {"yyyymmddhh24miss", packed, 0, 14},
        {"yyyymmddhh24mi", packed, 0, 12},
        {"yyyymmddhh24", packed, 0, 10},
        {"yyyymmdd", packed, 0, 8},
        {"hh24miss", packed, 8, 6},
        {"hh24mi", packed, 8, 4},
        {"hh24", packed, 8, 2},
        {"mi", packed, 10, 2},
    };
    for (const auto &e : table)
        if (strcmp(fmt, e.name) == 0) {
            memset(F, 0, sizeof *F);
            F->width = e.width;
            for (uint32_t k = 0; k < e.width; ++k) F->idx[k] = e.idx[e.from + k];
            return true;
        }
    return false;
}

extern "C" int mdc_timetostr_width(const char *fmt) {
    mdc::TimeFmt F;
    if (!fmt || !time_format(fmt, &F)) return fail(MDC_ERR_ARG, "not one of timetostr's formats");
    return (int)F.width;
}

extern "C" int mdc_timetostr_device(int device, const int64_t *d_time, int64_t rows, const char *fmt, int tz_offset,
                                    uint32_t *d_off, uint8_t *d_bytes) {
    mdc::TimeFmt F;
    if (!fmt || !time_format(fmt, &F))
        return fail(MDC_ERR_ARG, "not one of timetostr's formats (the reference leaves the string as it is)");
    if (rows < 0) return fail(MDC_ERR_ARG, "negative rows");
    if (tz_offset <= -86400 || tz_offset >= 86400) return fail(MDC_ERR_ARG, "tz_offset %d is no zone", tz_offset);
    if (!d_off || (rows && (!d_time || !d_bytes))) return fail(MDC_ERR_ARG, "NULL pointer");
    if ((uint64_t)rows * F.width > 0xFFFFFFFFull) return fail(MDC_ERR_TOO_LARGE, "the column would exceed 4 GiB: split the rows");
    Device *d;
    int rc = get_device(device, &d);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(d->id));
    if (rows == 0) {
        const uint32_t zero = 0;
        HIP_TRY(hipMemcpy(d_off, &zero, 4, hipMemcpyHostToDevice));
        return MDC_OK;
    }
    void *d_status = nullptr;
    if ((rc = dev_alloc(d, 64, &d_status))) return rc;
    uint32_t status = 0;
    hipError_t e = hipMemcpyAsync(d_status, &status, 4, hipMemcpyHostToDevice, d->stream);
    if (e == hipSuccess)
        e = mdc::launch_timetostr(d_time, (uint64_t)rows, F, tz_offset, d_off, d_bytes, (uint32_t *)d_status, d->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&status, d_status, 4, hipMemcpyDeviceToHost, d->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
    dev_free(d, d_status);
    if (e != hipSuccess) return fail(MDC_ERR_HIP, "timetostr: %s", hipGetErrorString(e));
    if (status)
        return fail(MDC_ERR_NUMERIC, "a time lies outside the years 0000 to 9999 (its text would not have the format's width)");
    return MDC_OK;
}

extern "C" int mdc_formatter_last_timing(const mdc_formatter *f, mdc_format_timing *t) {
    if (!f || !t) return fail(MDC_ERR_ARG, "NULL argument");
    *t = f->timing;
    return MDC_OK;
}

extern "C" int64_t mdc_text_files(const mdc_text *t) { return t ? (int64_t)t->file_off.size() - 1 : -1; }
extern "C" uint64_t mdc_text_bytes(const mdc_text *t) { return t ? t->nbytes : 0; }

extern "C" int mdc_text_file_range(const mdc_text *t, int64_t k, uint64_t *begin, uint64_t *end) {
    if (!t || !begin || !end) return fail(MDC_ERR_ARG, "NULL argument");
    if (k < 0 || k + 1 >= (int64_t)t->file_off.size()) return fail(MDC_ERR_ARG, "file %lld out of range", (long long)k);
    *begin = t->file_off[k];
    *end = t->file_off[k + 1];
    return MDC_OK;
}

extern "C" int mdc_text_device(const mdc_text *t, const uint8_t **d_data, uint64_t *nbytes) {
    if (!t || !d_data || !nbytes) return fail(MDC_ERR_ARG, "NULL argument");
    *d_data = t->d_data;
    *nbytes = t->nbytes;
    return MDC_OK;
}

extern "C" int mdc_text_host(mdc_text *t, const uint8_t **data, uint64_t *nbytes) {
    if (!t || !data || !nbytes) return fail(MDC_ERR_ARG, "NULL argument");
    if (!t->have_host) {
        t->h_data.resize(t->nbytes);
        if (t->nbytes) {
            int rc = d2h(t->dev, t->h_data.data(), t->d_data, t->nbytes);
            if (rc) return rc;
        }
        t->have_host = true;
    }
    *data = t->h_data.data();
    *nbytes = t->nbytes;
    return MDC_OK;
}

// create the directories above a file name, as cofile::open does through newdir(name,true)
// (public/_public.cpp:952-977)
static bool make_parent_dirs(const std::string &name) {
    for (size_t pos = name.find('/', 1); pos != std::string::npos; pos = name.find('/', pos + 1)) {
        std::string dir = name.substr(0, pos);
        if (access(dir.c_str(), F_OK) != 0 && mkdir(dir.c_str(), 0755) != 0) return false;
    }
    return true;
}

extern "C" int mdc_text_write_files(mdc_text *t, const char *outpath, const char *bfilename, const char *timestamp,
                                    const char *efilename, int first_seq, int *nwritten) {
    if (!t || !outpath || !bfilename || !efilename) return fail(MDC_ERR_ARG, "NULL argument");
    if (nwritten) *nwritten = 0;
    const int64_t nfiles = (int64_t)t->file_off.size() - 1;
    if (nfiles == 0) {
        if (t->maxcount == 0) return fail(MDC_ERR_IO, "no rows and maxcount 0: no file is open to close");
        return MDC_OK;
    }
    char stamp[32];
    if (!timestamp) {
        time_t now = time(nullptr);
        struct tm tmv;
        localtime_r(&now, &tmv);
        strftime(stamp, sizeof stamp, "%Y%m%d%H%M%S", &tmv);
        timestamp = stamp;
    }
    const uint8_t *data;
    uint64_t n;
    int rc = mdc_text_host(t, &data, &n);
    if (rc) return rc;
    for (int64_t k = 0; k < nfiles; ++k) {
        std::string name = std::string(outpath) + "/" + bfilename + "_" + timestamp + "_" + efilename + "_" +
                           std::to_string(first_seq + (int)k) + ".xml";
        std::string tmp = name + ".tmp";
        make_parent_dirs(name);
        FILE *fp = fopen(tmp.c_str(), "wb");
        if (!fp) return fail(MDC_ERR_IO, "cannot open %s", tmp.c_str());
        const size_t len = (size_t)(t->file_off[k + 1] - t->file_off[k]);
        bool ok = fwrite(data + t->file_off[k], 1, len, fp) == len;
        ok = (fclose(fp) == 0) && ok;
        if (!ok) return fail(MDC_ERR_IO, "write to %s failed", tmp.c_str());
        if (rename(tmp.c_str(), name.c_str()) != 0) return fail(MDC_ERR_IO, "rename of %s failed", tmp.c_str());
        if (nwritten) *nwritten = (int)(k + 1);
    }
    return MDC_OK;
}

extern "C" void mdc_text_free(mdc_text *t) {
    if (!t) return;
    if (t->dev) dev_free(t->dev, t->d_data);
    delete t;
}

// ------------------------------------------------------------------------------------------
// library / device helpers

extern "C" const char *mdc_version(void) { return "mdc_parse 0.1 (gfx950)"; }

extern "C" int mdc_device_count(void) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess) return fail(MDC_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return count;
}

extern "C" const char *mdc_strerror(int code) {
    switch (code) {
    case MDC_OK: return "ok";
    case MDC_FILE_OPEN_FAILED: return "open file failed";
    case MDC_ERR_ARG: return "bad argument";
    case MDC_ERR_NO_DEVICE: return "no HIP device (no CPU fallback exists)";
    case MDC_ERR_HIP: return "HIP runtime error";
    case MDC_ERR_NOMEM: return "out of device memory";
    case MDC_ERR_RECORD_TOO_LONG: return "record of 2 GiB or more";
    case MDC_ERR_TOO_LARGE: return "column larger than 4 GiB";
    case MDC_ERR_NUMERIC: return "numeric text beyond the exact-conversion range";
    case MDC_ERR_IO: return "I/O error";
    case MDC_ERR_INTERNAL: return "internal error";
    case MDC_ERR_BUSY: return "the static scratch array is in use by another call";
    default: return "unknown error";
    }
}

extern "C" const char *mdc_last_error(void) { return g_err.c_str(); }

extern "C" int mdc_device_alloc(int device, size_t n, void **d_ptr) {
    if (!d_ptr) return fail(MDC_ERR_ARG, "d_ptr is NULL");
    Device *d;
    int rc = get_device(device, &d);
    if (rc) return rc;
    return dev_alloc(d, n + 64, d_ptr);
}

extern "C" int mdc_device_free(int device, void *d_ptr) {
    Device *d;
    int rc = get_device(device, &d);
    if (rc) return rc;
    dev_free(d, d_ptr);
    return MDC_OK;
}

extern "C" int mdc_device_upload(int device, void *d_dst, const void *h_src, size_t n) {
    Device *d;
    int rc = get_device(device, &d);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(d->id));
    if (n) HIP_TRY(hipMemcpy(d_dst, h_src, n, hipMemcpyHostToDevice));
    return MDC_OK;
}

extern "C" int mdc_device_download(int device, void *h_dst, const void *d_src, size_t n) {
    Device *d;
    int rc = get_device(device, &d);
    if (rc) return rc;
    return d2h(d, h_dst, d_src, n);
}

extern "C" int mdc_device_synchronize(int device) {
    Device *d;
    int rc = get_device(device, &d);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(d->id));
    HIP_TRY(hipDeviceSynchronize());
    return MDC_OK;
}
