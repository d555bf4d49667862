// mdc_kernels.hip -- gfx950 kernels of the bulk record parser.
//
// One workgroup (256 threads = 4 wave64) per TILE input bytes.  A tile owns the records
// whose terminating '\n' lies inside it.  Per tile:
//
//   A. every lane loads 64 consecutive tile bytes (4 x dwordx4) into registers and LDS; a
//      512-byte look-behind window in front of the tile is staged with them and grows
//      backwards only when the terminator in front of the tile is not inside it
//   B. SWAR compares on the registers give 64-bit masks of '\n' and '<'; a '\n' preceded by
//      endbz is a record terminator (cifile::readline, reference public/_public.cpp:866-892);
//      one packed block-wide prefix sum ranks every byte by record and numbers the '<' events
//   C. the '<' positions are compacted into a dense LDS list (position | record rank)
//   D. one lane per event: an unaligned 8-byte LDS read is the key into a hash table of the
//      schema's tag texts; a hit does an LDS atomicMin on first_pos[pattern][record]: the
//      independent first-open / first-close search of getxmlbuffer (public/_public.cpp:466,469)
//   E. one lane per cell: clip, negative length and carry-forward (tools/xmltodb.cpp:492-522),
//      then picknumber's filter (public/_public.cpp:178-204), the cell length and the typed
//      value (mdc_numparse.h), on 16 source bytes held in registers
//   F. per-column prefix sums give every cell its offset
//   G. cells are compacted into an LDS staging area that mirrors the destination's 16-byte
//      phase and leave with 16-byte stores; delimited mode resolves its fields with one lane
//      per record (ccmdstr::splittocmd, public/_public.cpp:274-302) in place of C/D
//
// The kernel runs twice per parse: COUNT mode stops after step 4 and reports per-tile byte
// and row counts; a scan turns them into tile base offsets; WRITE mode repeats the (fully
// deterministic) analysis and stores the columns.
#include <hip/hip_runtime.h>

#include "mdc_device.h"
#include <cstdlib>

#include "mdc_numparse.h"
#include "mdc_bignum.h"

namespace mdc {

// ---------------------------------------------------------------------------------------
// small device helpers

// 0x80 in every byte of w equal to the byte replicated in pat4 (exact, no borrow artefacts)
__device__ __forceinline__ uint32_t bytes_eq(uint32_t w, uint32_t pat4) {
    uint32_t x = w ^ pat4;
    uint32_t t = ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x;
    return ~(t | 0x7F7F7F7Fu);
}
// the same for a pattern byte below 0x80: with the high bits set aside, 0x80 - (byte ^ pat) keeps
// bit 7 only for a zero difference and never borrows from the next byte.  The two terms that do
// not depend on the pattern are shared by all masks made from one word.
__device__ __forceinline__ uint32_t bytes_eq7(uint32_t w, uint32_t pat4) {
    return (0x80808080u - ((w & 0x7F7F7F7Fu) ^ pat4)) & ~w & 0x80808080u;
}
// gathers the four 0x80 flags of m into bits 0..3
__device__ __forceinline__ uint32_t pack4(uint32_t m) { return (((m >> 7) * 0x00204081u) >> 21) & 0xFu; }

// Packs the 0x80 flags of eight words into one 32-bit mask, bit 4k+b = byte b of word k: a
// byte-wise dot product with the weights 1,2,4,8 (16..128 for the odd word of a pair) turns two
// words into 128 * (their eight bits); the four pairs are then shifted into place.
__device__ __forceinline__ uint32_t pack32(uint32_t f0, uint32_t f1, uint32_t f2, uint32_t f3, uint32_t f4, uint32_t f5,
                                           uint32_t f6, uint32_t f7) {
    // a flag is 0x80 = 128, so every pair's sum is 128 * (its eight bits): bits 7..14
    const uint32_t a0 = __builtin_amdgcn_udot4(f1, 0x80402010u, __builtin_amdgcn_udot4(f0, 0x08040201u, 0u, false), false);
    const uint32_t a1 = __builtin_amdgcn_udot4(f3, 0x80402010u, __builtin_amdgcn_udot4(f2, 0x08040201u, 0u, false), false);
    const uint32_t a2 = __builtin_amdgcn_udot4(f5, 0x80402010u, __builtin_amdgcn_udot4(f4, 0x08040201u, 0u, false), false);
    const uint32_t a3 = __builtin_amdgcn_udot4(f7, 0x80402010u, __builtin_amdgcn_udot4(f6, 0x08040201u, 0u, false), false);
    return (a0 >> 7) | (a1 << 1) | (a2 << 9) | (a3 << 17);
}

__device__ __forceinline__ uint32_t mask16_eq(const uint4 &v, uint32_t pat4) {
    return pack4(bytes_eq(v.x, pat4)) | (pack4(bytes_eq(v.y, pat4)) << 4) | (pack4(bytes_eq(v.z, pat4)) << 8) |
           (pack4(bytes_eq(v.w, pat4)) << 12);
}

__device__ __forceinline__ uint32_t rep4(uint32_t b) { return b * 0x01010101u; }

// inclusive prefix sum across the 64 lanes of a wave, in the data-parallel-primitive form of the
// vector adds (no LDS traffic): four shifts inside every row of 16 lanes, then the last lane of
// row 0 and row 2 is added to the row above, then lane 31 to the upper half.  A lane without a
// source (shifted in from outside its row, or in a row the mask leaves out) adds 0.
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
    return v;
}
// the value of the wave's last lane
__device__ __forceinline__ uint32_t wave_last(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, 63); }

// Walks the bytes of one cell's source through the cell's filter.  EMIT=false measures the
// cell (and feeds the number parser); EMIT=true writes the surviving bytes to dst.
template <bool EMIT>
__device__ __forceinline__ uint32_t cell_walk(const uint8_t *src, uint32_t len, uint32_t lvl, bool nulcut,
                                             uint32_t typed, uint8_t *dst, uint64_t &value, bool &valid,
                                             bool &range) {
    uint32_t out = 0;
    NumAcc acc;
    bool boolok = true;
    const bool isfloat = (typed == T_F64 || typed == T_F32);
    const bool issigned = (typed == T_I32 || typed == T_I64 || isfloat);
    if (!EMIT) num_init(acc);
    for (uint32_t k = 0; k < len; ++k) {
        const uint32_t b = src[k];
        const bool dig = (b - '0') < 10u;
        const bool sgn = (b == '+') | (b == '-');
        const bool dot = (b == '.');
        bool pass;
        if (lvl == LVL_CHAR) {
            if (nulcut && b == 0) break;
            pass = true;
        } else if (lvl == LVL_NUMBER) {
            pass = dig | sgn | dot;
        } else {
            pass = dig;
        }
        if (!pass) continue;
        if (EMIT) {
            dst[out] = (uint8_t)b;
        } else if (typed != T_NONE) {
            if (typed == T_BOOL) {
                const uint32_t u = (b >= 'a' && b <= 'z') ? b - 32 : b;
                const uint32_t want = out == 0 ? 'T' : out == 1 ? 'R' : out == 2 ? 'U' : out == 3 ? 'E' : 0x100;
                boolok = boolok && (u == want);
            } else if (dig | (issigned & sgn) | (isfloat & dot)) {
                num_feed(acc, b, isfloat);
            }
        }
        ++out;
    }
    if (!EMIT && typed != T_NONE) {
        if (typed == T_BOOL) {
            valid = true;
            value = (boolok && out == 4) ? 1 : 0;
        } else {
            uint64_t v = 0;
            valid = num_finish(acc, typed, v, range);
            value = valid ? v : 0;
        }
    }
    return out;
}

// Diagnostic build only (-DMDC_PHASE_TIMING): thread 0 of every block adds the cycles spent
// since the previous stamp to a per-phase counter that sits behind the status word.  The
// counters are read by nothing in the kernel.
#ifdef MDC_ABLATE
#define SKIP(bit) (P.debug & (bit))
#else
#define SKIP(bit) false
#endif
#ifdef MDC_PHASE_TIMING
#define STAMP(i)                                                                                   \
    do {                                                                                           \
        if (threadIdx.x == 0) {                                                                    \
            const unsigned long long now_ = __builtin_readcyclecounter();                          \
            atomicAdd(((unsigned long long *)P.status) + 2 + (WRITE ? 12 : 0) + (i), now_ - stamp_t_);                \
            stamp_t_ = now_;                                                                       \
        }                                                                                          \
    } while (0)
#define STAMP_INIT unsigned long long stamp_t_ = __builtin_readcyclecounter()
#else
#define STAMP(i) \
    do {         \
    } while (0)
#define STAMP_INIT \
    do {           \
    } while (0)
#endif

// ---------------------------------------------------------------------------------------
// indices of the block-wide scalars kept in LDS
enum {
    MS_LAST = 0,    // look-behind: LDS index of the last terminator in front of the tile
    MS_HASNUL = 1,
    MS_STAGED = 2,
    MS_LBCNT = 3,   // delimited mode: separators of the first record that lie in the look-behind
    MS_WTOT = 4,    // 4 words: per-wave totals of the rank/event scan
    MS_NOMAP = 9,   // sizing pass: a cell does not fit a map word
    MS_IRR = 10,    // sizing pass: a record of the tile does not follow the tag template
    MS_T0 = 11,     // long first record: LDS index of the first terminator of the tile
    MS_LONGE = 12,  // long first record: its entry in the list of long records
    MS_SEPOV = 13,  // delimited mode: two separator matches of the tile overlap
    MS_COUNT = 16,
    // one-pass parse only (its layout has room for MS_XCOUNT words)
    MS_ROW0 = 16,   // 2 words: the tile's first row
    MS_STAGE = 18,  // bytes of the staging area handed out so far
    MS_ODD = 19,    // the tile does not take the straight-line write: left to the full write kernel
    MS_ROWOK = 20,  // the tile's rows fit the output arrays
    MS_FWORD = 24,  // up to 8 words: found bits of a row that follows the template
    MS_XCOUNT = 32
};

// unaligned LDS reads (gfx950 honours byte-unaligned ds_read_b64/b128)
__device__ __forceinline__ uint64_t lds_u64(const uint8_t *p) {
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
__device__ __forceinline__ uint4 lds_u128(const uint8_t *p) {
    uint4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}

// The same bytes through aligned dword reads.  An LDS access that is not aligned to its own size
// is taken lane by lane (measured: 66 cycles of the LDS pipe per wave instruction against 7-8
// for an aligned one), so a window at an arbitrary address is read as the aligned dwords around
// it (ds_read2_b32: two dwords, 4-byte alignment is enough) and shifted into place with
// v_alignbyte.  The reads are written out because the compiler would merge dword loads of a
// pointer into one wide, misaligned access.
typedef uint32_t lds_u32x2 __attribute__((ext_vector_type(2)));
// bytes [p, p + 12)
__device__ __forceinline__ void lds_bytes12(const uint8_t *p, uint32_t &w0, uint32_t &w1, uint32_t &w2) {
    const uint32_t addr = (uint32_t)(size_t)p;  // (the low half of a pointer into LDS is the LDS address)
    const uint32_t base = addr & ~3u, sh = addr & 3u;
    lds_u32x2 a, b;
    asm volatile("ds_read2_b32 %0, %2 offset1:1\n\tds_read2_b32 %1, %2 offset0:2 offset1:3\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(a), "=&v"(b)
                 : "v"(base)
                 : "memory");
    w0 = __builtin_amdgcn_alignbyte(a[1], a[0], sh);
    w1 = __builtin_amdgcn_alignbyte(b[0], a[1], sh);
    w2 = __builtin_amdgcn_alignbyte(b[1], b[0], sh);
}
// bytes [p, p + 8)
__device__ __forceinline__ void lds_bytes8(const uint8_t *p, uint32_t &w0, uint32_t &w1) {
    const uint32_t addr = (uint32_t)(size_t)p;
    const uint32_t base = addr & ~3u, sh = addr & 3u;
    lds_u32x2 a;
    uint32_t b;
    asm volatile("ds_read2_b32 %0, %2 offset1:1\n\tds_read_b32 %1, %2 offset:8\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(a), "=&v"(b)
                 : "v"(base)
                 : "memory");
    w0 = __builtin_amdgcn_alignbyte(a[1], a[0], sh);
    w1 = __builtin_amdgcn_alignbyte(b, a[1], sh);
}
// bytes [p, p + 16)
__device__ __forceinline__ uint4 lds_bytes16(const uint8_t *p) {
    const uint32_t addr = (uint32_t)(size_t)p;
    const uint32_t base = addr & ~3u, sh = addr & 3u;
    lds_u32x2 a, b;
    uint32_t c;
    asm volatile("ds_read2_b32 %0, %3 offset1:1\n\tds_read2_b32 %1, %3 offset0:2 offset1:3\n\tds_read_b32 %2, %3 offset:16\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c)
                 : "v"(base)
                 : "memory");
    return make_uint4(__builtin_amdgcn_alignbyte(a[1], a[0], sh), __builtin_amdgcn_alignbyte(b[0], a[1], sh),
                      __builtin_amdgcn_alignbyte(b[1], b[0], sh), __builtin_amdgcn_alignbyte(c, b[1], sh));
}

// 64-bit byte-equality mask of the 64 bytes held in four dwordx4 registers
__device__ __forceinline__ uint64_t mask64_eq(const uint4 (&v)[4], uint32_t pat4) {
    const uint32_t lo = pack32(bytes_eq7(v[0].x, pat4), bytes_eq7(v[0].y, pat4), bytes_eq7(v[0].z, pat4),
                               bytes_eq7(v[0].w, pat4), bytes_eq7(v[1].x, pat4), bytes_eq7(v[1].y, pat4),
                               bytes_eq7(v[1].z, pat4), bytes_eq7(v[1].w, pat4));
    const uint32_t hi = pack32(bytes_eq7(v[2].x, pat4), bytes_eq7(v[2].y, pat4), bytes_eq7(v[2].z, pat4),
                               bytes_eq7(v[2].w, pat4), bytes_eq7(v[3].x, pat4), bytes_eq7(v[3].y, pat4),
                               bytes_eq7(v[3].z, pat4), bytes_eq7(v[3].w, pat4));
    return (uint64_t)lo | ((uint64_t)hi << 32);
}
// does any of the 64 bytes equal zero?
__device__ __forceinline__ bool any_zero64(const uint4 (&v)[4]) {
    uint32_t z = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        z |= (v[k].x - 0x01010101u) & ~v[k].x;
        z |= (v[k].y - 0x01010101u) & ~v[k].y;
        z |= (v[k].z - 0x01010101u) & ~v[k].z;
        z |= (v[k].w - 0x01010101u) & ~v[k].w;
    }
    return (z & 0x80808080u) != 0;
}

// is the '\n' at LDS index p a record terminator (preceded by endbz)?  All bytes involved are
// staged: the look-behind always covers at least MAX_ENDBZ bytes in front of the tile.
__device__ __forceinline__ bool is_term(const uint8_t *sdata, const SchemaHdr *H, int64_t base_pos, int p) {
    const int el = (int)H->endbz_len;
    if (el == 0) return true;
    if (base_pos + p - el < 0) return false;
    if (el <= 8) {
        const uint64_t w = lds_u64(sdata + p - 8), want = *(const uint64_t *)H->endbz_tail;
        const uint64_t m = ~0ull << (8 * (8 - el));
        return ((w ^ want) & m) == 0;
    }
    const uint8_t *mark = endbz_text(H);
    for (int k = 0; k < el; ++k)
        if (sdata[p - el + k] != mark[k]) return false;
    return true;
}

// first_pos[pattern][record] is kept in 16 bits (positions in the LDS image are below 2^15), two
// entries per LDS word; LDS atomics are 32 bits wide, so the minimum is taken with a
// compare-and-swap on the word (two events rarely meet in one word: one round, as a rule)
constexpr uint32_t NONE16 = 0xFFFFu;
__device__ __forceinline__ void evt_min(uint32_t *evt, uint32_t idx, uint32_t q) {
    uint32_t *w = evt + (idx >> 1);
    const uint32_t sh = (idx & 1) * 16;
    uint32_t old = 0xFFFFFFFFu;  // guess: untouched (a wrong guess costs one round and returns the truth)
    for (;;) {
        if (((old >> sh) & 0xFFFFu) <= q) return;
        const uint32_t want = (old & ~(0xFFFFu << sh)) | (q << sh);
        const uint32_t seen = atomicCAS(w, old, want);
        if (seen == old) return;
        old = seen;
    }
}
__device__ __forceinline__ uint32_t evt_get(const uint32_t *evt, uint32_t idx) {
    const uint32_t v = ((const uint16_t *)evt)[idx];
    return v == NONE16 ? NONE : v;
}

// Tag lookup for the '<' at LDS index q, in three steps so that a lane can keep several events
// in flight (every step starts with an LDS access whose latency the others hide):
//   tag_key    one unaligned 8-byte read: the key = text up to and including '>' (at most 8 bytes)
//   tag_probe  ONE probe of the perfect-hash table (32-byte entry)
//   tag_apply  a hit lowers first_pos[pattern][record]; entries that share a key are chained
//              and told apart by the rest of their text
struct TagKey {
    uint32_t lo, hi;
};
__device__ __forceinline__ TagKey tag_key(const uint8_t *sdata, uint32_t q) {
    uint32_t lo, hi, w2;
    lds_bytes12(sdata + q + 1, lo, hi, w2);
    const uint32_t glo = bytes_eq7(lo, 0x3E3E3E3Eu), ghi = bytes_eq7(hi, 0x3E3E3E3Eu);
    if (glo) {
        const uint32_t g = (uint32_t)__builtin_ctz(glo) >> 3;
        lo &= 0xFFFFFFFFu >> (8 * (3 - g));
        hi = 0;
    } else if (ghi) {
        const uint32_t g = (uint32_t)__builtin_ctz(ghi) >> 3;
        hi &= 0xFFFFFFFFu >> (8 * (3 - g));
    }
    TagKey k;
    k.lo = lo;
    k.hi = hi;
    return k;
}
__device__ __forceinline__ uint32_t tag_slot(const SchemaHdr *H, const TagKey &k) {
    return ht_hash(k.lo, k.hi, H->ht_seed) & H->ht_mask;
}
__device__ __forceinline__ void tag_apply(const uint8_t *sdata, const DevCol *cols, const HtEntry *ht, uint32_t q,
                                          const TagKey &k, uint32_t slot, const uint4 &e0, uint32_t *evt, uint32_t rl,
                                          uint32_t RB) {
    if (!(e0.z >> 31)) return;                               // nothing hashes to this slot
    if ((e0.x != k.lo || e0.y != k.hi) && !e0.w) return;    // (the common miss: another key, no chain)
    const uint32_t a = q + 1;
    for (;;) {
        const HtEntry &e = ht[slot];
        const uint32_t meta = e.meta;
        const uint32_t tlen = meta & 0xFFu, kind = (meta >> 8) & 1u, col = (meta >> 16) & 0x3FFFu;
        bool ok = e.klo == k.lo && e.khi == k.hi;
        if (ok && tlen >= 8) {  // the key covers eight bytes; compare the rest of text+'>'
            const uint64_t w2 = lds_u64(sdata + a + 8);
            ok = (((uint32_t)w2 ^ e.k2lo) & e.m2lo) == 0 && (((uint32_t)(w2 >> 32) ^ e.k2hi) & e.m2hi) == 0;
            if (ok && (meta & 0x40000000u)) {
                const uint8_t *nm = cols[col].name;
                for (uint32_t k2 = 16; k2 < tlen; ++k2)
                    if (sdata[a + k2] != nm[k2 - kind]) {
                        ok = false;
                        break;
                    }
                if (ok && sdata[a + tlen] != '>') ok = false;
            }
        }
        if (ok) evt_min(evt, (2 * col + kind) * RB + rl, q);
        slot = e.next;
        if (!slot) break;
    }
}
__device__ __forceinline__ void match_tag(const uint8_t *sdata, const SchemaHdr *H, const DevCol *cols,
                                          const HtEntry *ht, uint32_t q, uint32_t *evt, uint32_t rl, uint32_t RB) {
    const TagKey k = tag_key(sdata, q);
    const uint32_t slot = tag_slot(H, k);
    const uint4 e0 = *(const uint4 *)&ht[slot];
    tag_apply(sdata, cols, ht, q, k, slot, e0, evt, rl, RB);
}

// The same lookup without side effects: how many patterns the '<' at LDS index q matches, and for
// the last of them idx = 2 * column + kind and L = length of the tag text through '>'.
__device__ __forceinline__ uint32_t tag_lookup(const uint8_t *sdata, const SchemaHdr *H, const DevCol *cols,
                                               const HtEntry *ht, uint32_t q, uint32_t &idx, uint32_t &L) {
    const TagKey k = tag_key(sdata, q);
    uint32_t slot = tag_slot(H, k);
    const HtEntry &f = ht[slot];
    if (!(f.meta >> 31)) return 0;
    const uint32_t a = q + 1;
    uint32_t n = 0;
    for (;;) {
        const HtEntry &e = ht[slot];
        const uint32_t meta = e.meta;
        const uint32_t tlen = meta & 0xFFu, kind = (meta >> 8) & 1u, col = (meta >> 16) & 0x3FFFu;
        bool ok = e.klo == k.lo && e.khi == k.hi;
        if (ok && tlen >= 8) {
            const uint64_t w2 = lds_u64(sdata + a + 8);
            ok = (((uint32_t)w2 ^ e.k2lo) & e.m2lo) == 0 && (((uint32_t)(w2 >> 32) ^ e.k2hi) & e.m2hi) == 0;
            if (ok && (meta & 0x40000000u)) {
                const uint8_t *nm = cols[col].name;
                for (uint32_t k2 = 16; k2 < tlen; ++k2)
                    if (sdata[a + k2] != nm[k2 - kind]) {
                        ok = false;
                        break;
                    }
                if (ok && sdata[a + tlen] != '>') ok = false;
            }
        }
        if (ok) {
            ++n;
            idx = 2 * col + kind;
            L = tlen + 1;
        }
        slot = e.next;
        if (!slot) break;
    }
    return n;
}

// Event k of a record against entry k of the tag template: the tag text through '>' must repeat
// the sample's byte for byte; then the event matches what the sample's did, and its position goes
// to first_pos[pattern][record] with a plain store (in a record that repeats the sample no
// pattern occurs twice).  false = the record does not follow the template.
// (tl: the entries as the tile kernel keeps them in LDS, the pattern index replaced by the pattern's row of
// first_pos, index x records per batch, so that no multiplication stands in the walk)
__device__ __forceinline__ bool tmpl_event(const uint8_t *sdata, const uint4 *tl, uint32_t nT, uint32_t q, uint32_t k,
                                           uint32_t *evt, uint32_t rl) {
    if (k >= nT) return false;
    const uint4 t = tl[k];
    uint32_t w0, w1, w2;
    lds_bytes12(sdata + q + 1, w0, w1, w2);
    const uint64_t w = (uint64_t)w0 | ((uint64_t)w1 << 32);
    // the bytes that do not belong to the text are shifted out (the shift counts come with the entry)
    const uint32_t sh1 = t.w & 0xFFu, sh2 = (t.w >> 8) & 0xFFu, row = t.w >> 16;
    const uint64_t key = (uint64_t)t.x | ((uint64_t)t.y << 32);
    const uint64_t d1 = (w ^ key) << sh1;
    const uint32_t d2 = (uint32_t)((uint64_t)(w2 ^ t.z) << sh2);
    const bool ok = (d1 == 0) & (d2 == 0);
    if (ok && row != TMPL_NOPAT) ((uint16_t *)evt)[row + rl] = (uint16_t)q;
    return ok;
}

// Builds the tag template from the second record of the input (the first one that starts right
// behind a terminator), or a later one when the host asks for it (the records of an input need not
// look like its second one).  One wave; lane l looks at bytes [64 l, 64 l + 64) of the first
// TMPL_WINDOW bytes.  No template (word 0 = 0) when the window holds no two terminators, the
// record has no or too many '<', a tag text is longer than TMPL_MAXLEN or holds a newline, an
// event matches two patterns or two events match the same one.
__global__ __launch_bounds__(64) void template_kernel(const TileParams P) {
    extern __shared__ __align__(16) uint8_t smem[];
    __shared__ int tpos[2];
    __shared__ uint32_t seen[2 * MAX_COLS / 32], bad;
    const int lane = threadIdx.x;
    const uint32_t sb = (P.schema_bytes + 15) & ~15u;
    uint8_t *win = smem + sb;
    for (uint32_t i = lane; i < P.schema_bytes / 16; i += 64) ((uint4 *)smem)[i] = ((const uint4 *)P.schema)[i];
    const uint32_t nwin = P.n < TMPL_WINDOW ? (uint32_t)P.n : TMPL_WINDOW;
    constexpr uint32_t WCH = (TMPL_WINDOW + 64) / 16;  // 16-byte chunks of the window; zeros behind the input
    if (P.n >= TMPL_WINDOW + 64) {                     // (all loads in flight at once)
        uint4 tv[(WCH + 63) / 64];
#pragma unroll
        for (uint32_t k = 0; k < (WCH + 63) / 64; ++k) {
            const uint32_t i = k * 64 + lane;
            tv[k] = ((const uint4 *)P.data)[i < WCH ? i : 0];
        }
#pragma unroll
        for (uint32_t k = 0; k < (WCH + 63) / 64; ++k) {
            const uint32_t i = k * 64 + lane;
            if (i < WCH) ((uint4 *)win)[i] = tv[k];
        }
    } else {
        for (uint32_t i = lane; i < WCH; i += 64) {
            uint32_t w[4] = {0, 0, 0, 0};
            for (uint32_t k = 0; k < 16; ++k)
                if ((uint64_t)i * 16 + k < P.n) w[k >> 2] |= (uint32_t)P.data[(uint64_t)i * 16 + k] << (8 * (k & 3));
            ((uint4 *)win)[i] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
    if (lane < 2) tpos[lane] = -1;
    if (lane < 2 * MAX_COLS / 32) seen[lane] = 0;
    if (lane == 0) bad = 0;
    __syncthreads();
    const SchemaHdr *H = (const SchemaHdr *)smem;
    const DevCol *cols = (const DevCol *)(H + 1);
    const HtEntry *ht = (const HtEntry *)((const uint8_t *)cols + ((H->ncols * sizeof(DevCol) + 15) & ~15u));
    const int el = (int)H->endbz_len;
    uint64_t tm = 0, em = 0;
    {
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = ((const uint4 *)win)[lane * 4 + k];
        uint64_t nlm = mask64_eq(v, 0x0A0A0A0Au);
        em = mask64_eq(v, 0x3C3C3C3Cu);
        const int lim = (int)nwin - lane * 64;  // bytes of this lane that belong to the window
        const uint64_t keep = lim >= 64 ? ~0ull : lim <= 0 ? 0ull : (~0ull >> (64 - lim));
        nlm &= keep;
        em &= keep;
        while (nlm) {
            const int b = __builtin_ctzll(nlm);
            nlm &= nlm - 1;
            const int p = lane * 64 + b;
            bool t = p - el >= 0;
            for (int k = 0; t && k < el; ++k) t = win[p - el + k] == endbz_text(H)[k];
            if (t) tm |= 1ull << b;
        }
    }
    {
        const uint32_t cnt = (uint32_t)__builtin_popcountll(tm);
        uint32_t rank = wave_incl_scan(cnt) - cnt;
        uint64_t t = tm;
        // the sampled record lies between terminators P.tmpl_sample and P.tmpl_sample + 1
        while (t && rank < P.tmpl_sample + 2) {
            const int b = __builtin_ctzll(t);
            t &= t - 1;
            if (rank >= P.tmpl_sample) tpos[rank - P.tmpl_sample] = lane * 64 + b;
            ++rank;
        }
    }
    __syncthreads();
    const int t0 = tpos[0], t1 = tpos[1];
    if (t1 < 0) {
        if (lane == 0) P.tmpl[0] = 0;
        return;
    }
    // the record's events: '<' at positions in (t0, t1)
    {
        const int lo = t0 + 1 - lane * 64, hi = t1 - lane * 64;  // window of this lane's bits: [lo, hi)
        uint64_t keep = ~0ull;
        if (lo >= 64 || hi <= 0)
            keep = 0;
        else {
            if (lo > 0) keep &= ~0ull << lo;
            if (hi < 64) keep &= ~(~0ull << hi);
        }
        em &= keep;
    }
    const uint32_t cnt = (uint32_t)__builtin_popcountll(em);
    const uint32_t incl = wave_incl_scan(cnt);
    const uint32_t nT = wave_last(incl);
    uint32_t k = incl - cnt;
    if (nT == 0 || nT > TMPL_MAX) {
        if (lane == 0) P.tmpl[0] = 0;
        return;
    }
    while (em) {
        const int b = __builtin_ctzll(em);
        em &= em - 1;
        const uint32_t q = (uint32_t)(lane * 64 + b);
        uint32_t idx = TMPL_NOPAT, L = 0;
        const uint32_t nm = tag_lookup(win, H, cols, ht, q, idx, L);
        if (nm > 1) bad = 1;
        if (nm == 0) {  // no pattern: the text through the first '>' of the eight bytes behind '<'
            idx = TMPL_NOPAT;
            L = 0;
            for (uint32_t i = 0; i < 8 && !L; ++i)
                if (win[q + 1 + i] == '>') L = i + 1;
        } else {
            const uint32_t old = atomicOr(&seen[idx >> 5], 1u << (idx & 31));
            if (old & (1u << (idx & 31))) bad = 1;
        }
        if (L == 0 || L > TMPL_MAXLEN) {
            bad = 1;
            L = 1;
        }
        uint32_t w[3] = {0, 0, 0};
        for (uint32_t i = 0; i < L; ++i) {
            const uint32_t c = win[q + 1 + i];
            if (c == '\n') bad = 1;
            w[i >> 2] |= c << (8 * (i & 3));
        }
        uint32_t *e = P.tmpl + 4 + 4 * k;
        e[0] = w[0];
        e[1] = w[1];
        e[2] = w[2];
        // shift counts that leave the text's bytes only: of the first 8 bytes, of bytes 8..11
        const uint32_t sh1 = 64 - 8 * (L < 8 ? L : 8), sh2 = L > 8 ? 32 - 8 * (L - 8) : 32;
        e[3] = sh1 | (sh2 << 8) | (idx << 16);
        ++k;
    }
    __syncthreads();
    // the column plan: what splitbuffer's column loop does for a record with the sample's tags
    for (uint32_t c = lane; c < H->ncols; c += 64) {
        auto has = [&](uint32_t col) {
            return ((seen[(2 * col) >> 5] >> ((2 * col) & 31)) & 1u) && ((seen[(2 * col + 1) >> 5] >> ((2 * col + 1) & 31)) & 1u);
        };
        const bool found = has(c);
        uint32_t cs = c, lvl = cols[c].level;
        bool have = found;
        if (!found && H->carry)
            while (cs > 0) {
                --cs;
                if (cols[cs].level > lvl) lvl = cols[cs].level;
                if (has(cs)) {
                    have = true;
                    break;
                }
            }
        P.tmpl[TMPL_PLAN_WORD + 2 * c] = cs | (lvl << 8) | ((uint32_t)found << 10) | ((uint32_t)have << 11) |
                                         (((uint32_t)cols[cs].namelen + 2) << 16);
        P.tmpl[TMPL_PLAN_WORD + 2 * c + 1] = (uint32_t)cols[cs].collen;
    }
    if (lane == 0) P.tmpl[0] = bad ? 0 : nT;
}

// Cells longer than 16 source bytes: filter only (length, or the surviving bytes).  Their
// typed conversion is left to deferred_convert_kernel, which keeps the number parser for
// arbitrary lengths out of this kernel.
template <bool EMIT>
__device__ __forceinline__ uint32_t cell_filter_long(const uint8_t *src, uint32_t len, uint32_t lvl, bool nulcut,
                                                    uint8_t *dst) {
    uint32_t out = 0;
    for (uint32_t k = 0; k < len; ++k) {
        const uint32_t b = src[k];
        const bool dig = (b - '0') < 10u;
        bool pass;
        if (lvl == LVL_CHAR) {
            if (nulcut && b == 0) break;
            pass = true;
        } else if (lvl == LVL_NUMBER) {
            pass = dig | (b == '+') | (b == '-') | (b == '.');
        } else {
            pass = dig;
        }
        if (pass) {
            if (EMIT) dst[out] = (uint8_t)b;
            ++out;
        }
    }
    return out;
}

// Cells of at most 16 source bytes, held in registers (lo = bytes 0..7, hi = bytes 8..15).
// One predicated pass per byte does picknumber's filter, the NUL cut, the cell length and the
// whole [+-]digits[.digits] prefix grammar of std::sto* (mdc_numparse.h) without a branch:
// a 16-byte text has at most 16 digits, so the digits fit one u64 and nothing can overflow.
// EMIT=true writes the surviving bytes to dst instead.
template <bool EMIT>
__device__ __forceinline__ uint32_t cell_scan16(uint64_t lo, uint64_t hi, uint32_t len, uint32_t lvl, bool nulcut,
                                               uint32_t typed, const double *p10, uint8_t *dst, uint64_t &value,
                                               bool &valid, bool &later) {
    uint32_t out = 0;
    const bool isfloat = (typed == T_F64) | (typed == T_F32);
    const bool issigned = (typed == T_I32) | (typed == T_I64) | isfloat;
    const bool cut = nulcut & (lvl == LVL_CHAR);
    uint64_t m = 0;
    uint32_t k10 = 0;
    bool seenq = false, stop = false, neg = false, seendot = false, any = false, boolok = true, dead = false;
    for (uint32_t k = 0; k < len; ++k) {
        const uint32_t b = (uint32_t)lo & 0xFFu;
        lo = (lo >> 8) | (hi << 56);
        hi >>= 8;
        const uint32_t d = b - '0';
        const bool dig = d < 10u;
        const bool sgn = (b == '+') | (b == '-');
        const bool dot = (b == '.');
        dead |= cut & (b == 0);
        const bool pass = !dead & (lvl == LVL_CHAR ? true : lvl == LVL_NUMBER ? (dig | sgn | dot) : dig);
        if (EMIT) {
            if (pass) dst[out] = (uint8_t)b;
        } else {
            if (typed == T_BOOL) {
                const uint32_t u = (b >= 'a' && b <= 'z') ? b - 32 : b;
                const uint32_t want = out == 0 ? 'T' : out == 1 ? 'R' : out == 2 ? 'U' : out == 3 ? 'E' : 0x100;
                boolok &= !pass | (u == want);
            } else {
                const bool q = pass & (dig | (issigned & sgn) | (isfloat & dot));
                const bool first = q & !seenq;
                stop |= (q & sgn & !first) | (q & dot & seendot);
                const bool live = q & !stop;
                neg |= live & first & (b == '-');
                const bool ld = live & dig;
                m = ld ? m * 10 + d : m;
                k10 += (ld & seendot) ? 1 : 0;
                any |= ld;
                seendot |= live & dot;
                seenq |= q;
            }
        }
        out += pass ? 1 : 0;
    }
    if (EMIT || typed == T_NONE) return out;
    valid = false;
    value = 0;
    if (typed == T_BOOL) {
        valid = true;
        value = (boolok && out == 4) ? 1 : 0;
        return out;
    }
    if (!any) return out;
    switch (typed) {
    case T_I32:
        if (m <= (neg ? 0x80000000ull : 0x7fffffffull)) {
            valid = true;
            value = neg ? (uint64_t)(-(int64_t)m) : m;
        }
        break;
    case T_U32:
        if (m <= 0x7fffffffull) {
            valid = true;
            value = m;
        }
        break;
    case T_I64:
    case T_U64:  // at most 16 digits: always in range
        valid = true;
        value = neg ? (uint64_t)0 - m : m;
        break;
    case T_F64:
        valid = true;
        if (m == 0) {
            value = neg ? 1ull << 63 : 0;
        } else if (m < (1ull << 53)) {
            double x = (double)m / p10[k10];  // both exact (k10 <= 16): one correctly rounded division
            value = (uint64_t)__double_as_longlong(neg ? -x : x);
        } else {  // more than 53 bits of digits: the exact division is the second stage's business
            valid = false;
            later = true;
        }
        break;
    default:  // T_F32: one rounding, directly from the decimal (second stage), except for zero
        if (m == 0) {
            valid = true;
            value = neg ? 1ull << 31 : 0;
        } else {
            later = true;
        }
    }
    return out;
}

// ---------------------------------------------------------------------------------------
// Fast cell path.  For a cell of up to 16 source bytes a 16-bit mask says which bytes are
// digits.  Cells with at most two other bytes (a sign, a decimal point, one stray byte: all
// well-formed numbers and dates) are measured, filtered and converted without any per-byte
// loop; everything else takes cell_scan16 / cell_filter_long.

// exact per-byte digit flags (0x80 where the byte is '0'..'9')
__device__ __forceinline__ uint32_t digit_flags(uint32_t w) {
    // (byte & 0x7F) ^ 0x30 is 0..9 for a digit; adding 0x76 sets bit 7 for everything above
    const uint32_t s = ((w & 0x7F7F7F7Fu) ^ 0x30303030u) + 0x76767676u;
    return ~s & ~w & 0x80808080u;
}
__device__ __forceinline__ uint32_t digit_mask16(const uint4 &v) {
    const uint32_t a0 = __builtin_amdgcn_udot4(digit_flags(v.y), 0x80402010u,
                                               __builtin_amdgcn_udot4(digit_flags(v.x), 0x08040201u, 0u, false), false);
    const uint32_t a1 = __builtin_amdgcn_udot4(digit_flags(v.w), 0x80402010u,
                                               __builtin_amdgcn_udot4(digit_flags(v.z), 0x08040201u, 0u, false), false);
    return (a0 >> 7) | (a1 << 1);
}

// byte p (0..15) of the 16-byte window (lo, hi)
__device__ __forceinline__ uint32_t win_byte(uint64_t lo, uint64_t hi, uint32_t p) {
    const uint64_t w = p < 8 ? lo : hi;
    return (uint32_t)(w >> (8 * (p & 7))) & 0xFFu;
}
// removes byte p from the window (bytes above it move down by one)
__device__ __forceinline__ void win_remove(uint64_t &lo, uint64_t &hi, uint32_t p) {
    const uint64_t slo = (lo >> 8) | (hi << 56), shi = hi >> 8;
    const uint64_t blo = p < 8 ? ((1ull << (8 * p)) - 1) : ~0ull;
    const uint64_t bhi = p < 8 ? 0ull : ((1ull << (8 * (p - 8))) - 1);
    lo = (lo & blo) | (slo & ~blo);
    hi = (hi & bhi) | (shi & ~bhi);
}
// keeps bytes 0..n-1 of the window (n <= 16), zeroes the rest
__device__ __forceinline__ void win_keep(uint64_t &lo, uint64_t &hi, uint32_t n) {
    lo &= n >= 8 ? ~0ull : ((1ull << (8 * n)) - 1);
    hi &= n >= 16 ? ~0ull : n <= 8 ? 0ull : ((1ull << (8 * (n - 8))) - 1);
}
// four decimal digit values in the bytes of w (byte 0 most significant) -> 0..9999
__device__ __forceinline__ uint32_t parse4(uint32_t w) {
    // byte-wise dot products: tens and units of the two pairs
    const uint32_t d = w & 0x0F0F0F0Fu;
    return __umul24(__builtin_amdgcn_udot4(d, 0x0000010Au, 0u, false), 100u) + __builtin_amdgcn_udot4(d, 0x010A0000u, 0u, false);
}

// strtotime's value (see time_convert) of fourteen digits that lie right-aligned in the 16-byte
// pair (ylo, yhi): two zero bytes, then yyyymmddhhmiss.  Without a branch or a table: the month
// is folded with a multiplication for the division by 12, the days before it come from
// floor(275 M / 9) less one or two from March on, and the leap days are counted from a year
// 400 earlier, which keeps every dividend positive (month 00 of year 0000 is a year -1).
__device__ __forceinline__ uint64_t time_from_digits(uint64_t ylo, uint64_t yhi, int32_t tz_offset) {
    const uint32_t w0 = (uint32_t)ylo & 0x0F0F0F0Fu, w1 = (uint32_t)(ylo >> 32) & 0x0F0F0F0Fu;
    const uint32_t w2 = (uint32_t)yhi & 0x0F0F0F0Fu, w3 = (uint32_t)(yhi >> 32) & 0x0F0F0F0Fu;
    const uint32_t year = __umul24(__builtin_amdgcn_udot4(w0, 0x010A0000u, 0u, false), 100u) +
                          __builtin_amdgcn_udot4(w1, 0x0000010Au, 0u, false);
    const uint32_t mon1 = __builtin_amdgcn_udot4(w1, 0x010A0000u, 0u, false);  // 0..99: mktime's month + 1
    const uint32_t mday = __builtin_amdgcn_udot4(w2, 0x0000010Au, 0u, false), hour = __builtin_amdgcn_udot4(w2, 0x010A0000u, 0u, false);
    const uint32_t mi = __builtin_amdgcn_udot4(w3, 0x0000010Au, 0u, false), sec = __builtin_amdgcn_udot4(w3, 0x010A0000u, 0u, false);
    const uint32_t m12 = mon1 + 11;                          // month + 12, 11..110
    const uint32_t q = __umul24(m12, 21846u) >> 18;          // m12 / 12
    const uint32_t M = m12 - 12u * q + 1;                    // month of the year, 1..12
    const uint32_t Y = year + q + 399;                       // the year (year + q - 1) plus 400
    const bool leap = ((Y & 3u) == 0) & ((Y % 100u != 0) | (Y % 400u == 0));
    const uint32_t t9 = __umul24(275u * M, 7282u) >> 16;     // 275 M / 9
    const int32_t yday = (int32_t)t9 - (M >= 3 ? (leap ? 1 : 2) : 0) + (int32_t)mday - 31;
    const uint32_t u = Y - 1;
    const int32_t leaps = (int32_t)(u / 4 - u / 100 + u / 400) - 97 - 477;  // leap days from 1970 to the year
    const int32_t days = 365 * ((int32_t)Y - 400 - 1970) + leaps + yday;
    const int32_t secs = (int32_t)(hour * 3600 + mi * 60 + sec) - tz_offset;
    return (uint64_t)((int64_t)days * 86400 + secs);
}

// Returns false when the cell needs the general path.  On success: flen = cell length, (elo,
// ehi) = the cell's bytes (window with the filtered-out bytes removed), and for typed != NONE
// the converted value / validity.  D16 = digit bits of the window's bytes.
__device__ __forceinline__ bool cell_fast(uint64_t lo, uint64_t hi, uint32_t len, uint32_t D16, uint32_t lvl,
                                          uint32_t typed, const double *p10, uint32_t &flen, uint64_t &value,
                                          bool &valid, uint64_t &elo, uint64_t &ehi, bool &later) {
    const uint32_t Lm = (1u << len) - 1;  // len <= 16
    const uint32_t D = D16 & Lm, N = ~D16 & Lm;
    const uint32_t N2 = N & (N - 1);
    const bool e1 = N != 0, e2 = N2 != 0;
    const uint32_t p1 = e1 ? (uint32_t)__builtin_ctz(N) : 16u, p2 = e2 ? (uint32_t)__builtin_ctz(N2) : 16u;
    const uint32_t c1 = win_byte(lo, hi, p1 & 15), c2 = win_byte(lo, hi, p2 & 15);
    const bool s1 = (c1 == '+') | (c1 == '-'), s2 = (c2 == '+') | (c2 == '-');
    const bool d1 = c1 == '.', d2 = c2 == '.';
    // picknumber's filter for the cell's level
    const bool P1 = e1 & (lvl == LVL_CHAR ? true : lvl == LVL_NUMBER ? (s1 | d1) : false);
    const bool P2 = e2 & (lvl == LVL_CHAR ? true : lvl == LVL_NUMBER ? (s2 | d2) : false);
    flen = len - ((e1 & !P1) ? 1 : 0) - ((e2 & !P2) ? 1 : 0);
    elo = lo;
    ehi = hi;
    if (e2 & !P2) win_remove(elo, ehi, p2);
    if (e1 & !P1) win_remove(elo, ehi, p1);
    const bool ok = (N2 & (N2 - 1)) == 0;  // at most two non-digit bytes
    valid = false;
    value = 0;
    if (typed == T_NONE) return ok;
    // the conversion's own filter, then the [+-]digits[.digits] prefix grammar
    const bool isfloat = typed == T_F64;
    const bool issigned = (typed == T_I32) | (typed == T_I64) | isfloat;
    const bool Q1 = P1 & ((s1 & issigned) | (d1 & isfloat));
    const bool Q2 = P2 & ((s2 & issigned) | (d2 & isfloat));
    uint32_t stop = 16, dotp = 16;
    bool neg = false;
    if (Q1) {
        if (s1) {
            if ((D & ((1u << p1) - 1)) == 0)
                neg = c1 == '-';
            else
                stop = p1;
        } else {
            dotp = p1;
        }
    }
    if (Q2 & (p2 < stop)) {
        if (s2) {
            if (!Q1 && (D & ((1u << p2) - 1)) == 0)
                neg = c2 == '-';
            else
                stop = p2;
        } else {
            if (dotp == 16)
                dotp = p2;
            else
                stop = p2;
        }
    }
    const uint32_t DL = D & ((1u << stop) - 1);  // stop == 16 keeps all 16 bits
    if (DL == 0) return ok;                        // no digit: the reference overload returns false
    const uint32_t nd = (uint32_t)__builtin_popcount(DL);
    const uint32_t k10 = dotp < 16 ? (uint32_t)__builtin_popcount(DL >> (dotp + 1)) : 0u;
    // the digits, contiguous from byte 0 (what follows them leaves the window in the shift below)
    uint64_t xlo = lo, xhi = hi;
    if (e2 & (p2 < stop)) win_remove(xlo, xhi, p2);
    if (e1 & (p1 < stop)) win_remove(xlo, xhi, p1);
    // right-align the nd digits in a 16-digit field and convert four at a time
    const uint32_t sh = 8 * (16 - nd);  // 0..120
    uint64_t ylo, yhi;
    if (sh >= 64) {
        yhi = xlo << (sh - 64);
        ylo = 0;
    } else {
        yhi = sh ? (xhi << sh) | (xlo >> (64 - sh)) : xhi;
        ylo = xlo << sh;
    }
    const uint32_t hi8 = __umul24(parse4((uint32_t)ylo), 10000u) + parse4((uint32_t)(ylo >> 32));
    const uint32_t lo8 = __umul24(parse4((uint32_t)yhi), 10000u) + parse4((uint32_t)(yhi >> 32));
    const uint64_t m = (uint64_t)hi8 * 100000000ull + lo8;
    switch (typed) {
    case T_I32:
        if (m <= (neg ? 0x80000000ull : 0x7fffffffull)) {
            valid = true;
            value = neg ? (uint64_t)(-(int64_t)m) : m;
        }
        break;
    case T_U32:
        if (m <= 0x7fffffffull) {
            valid = true;
            value = m;
        }
        break;
    case T_I64:
    case T_U64:  // at most 16 digits: always in range
        valid = true;
        value = neg ? (uint64_t)0 - m : m;
        break;
    default:  // T_F64
        valid = true;
        if (m == 0) {
            value = neg ? 1ull << 63 : 0;
        } else if (m < (1ull << 53)) {
            const double x = (double)m / p10[k10];  // both exact (k10 <= 16): one correctly rounded division
            value = (uint64_t)__double_as_longlong(neg ? -x : x);
        } else {  // more than 53 bits of digits: left to the second stage
            valid = false;
            later = true;
        }
        break;
    }
    return ok;
}

// The straight-line path for a whole wave of plain cells.  A cell is plain when it has at most 16
// source bytes that are digits, one sign in front and one decimal point (every well-formed number
// and date), or when it is an untyped text cell of that length.  Everything is computed without a
// per-lane branch; the caller takes the results only when all lanes of the pass report true, and
// runs the general code otherwise.  `typed` is the same for the whole wave.
// SHORT: every cell of the wave has at most 8 source bytes (most numbers): half the window, half
// the work.
template <bool SHORT, bool TIME = false, bool F32 = false>
__device__ __forceinline__ bool cell_simple(const uint4 &w4, uint32_t len, uint32_t lvl, uint32_t typed, bool convert,
                                            const double *p10, int32_t tz_offset, uint32_t &flen, uint64_t &elo,
                                            uint64_t &ehi, uint64_t &value, bool &valid) {
    constexpr uint32_t CAP = SHORT ? 8 : 16;
    const uint64_t lo = (uint64_t)w4.x | ((uint64_t)w4.y << 32);
    const uint64_t hi = SHORT ? 0ull : (uint64_t)w4.z | ((uint64_t)w4.w << 32);
    const uint32_t lenc = len < CAP ? len : CAP;
    const uint32_t Lm = (1u << lenc) - 1;
    const uint32_t D16 = SHORT ? __builtin_amdgcn_udot4(digit_flags(w4.y), 0x80402010u,
                                                        __builtin_amdgcn_udot4(digit_flags(w4.x), 0x08040201u, 0u, false), false) >> 7
                               : digit_mask16(w4);
    const uint32_t D = D16 & Lm, N = ~D16 & Lm;
    const uint32_t c0 = (uint32_t)lo & 0xFFu;
    const uint32_t sg = N & (uint32_t)((c0 == '+') | (c0 == '-'));  // 1: a sign in front
    const uint32_t N1 = N & ~sg;                                     // the other bytes that are no digits
    const uint32_t p = (uint32_t)__builtin_ctz(N1 | (1u << CAP));    // the first of them (CAP: none)
    const uint32_t cp = SHORT ? (uint32_t)(lo >> (8 * (p & 7))) & 0xFFu : win_byte(lo, hi, p & 15);
    const uint32_t nd = (uint32_t)__builtin_popcount(D);
    const bool isdate = lvl == LVL_DATE;
    bool simple = (len <= CAP) & ((N1 & (N1 - 1)) == 0) & ((N1 == 0) | (cp == '.')) & !(isdate & (sg != 0));
    if (!SHORT && (F32 || typed == T_F64)) simple &= nd <= 15;  // (more digits may pass 2^53: the second stage divides exactly)
    if (typed == T_NONE) simple |= (lvl == LVL_CHAR) & (len <= CAP);
    // the window without the decimal point (p == CAP: as it is)
    const uint64_t slo = SHORT ? lo >> 8 : (lo >> 8) | (hi << 56), shi = hi >> 8;
    const uint32_t pl = p < 8 ? p : 8, ph = p < 8 ? 0 : p - 8;  // bytes kept in place in either half
    const uint64_t klo = pl < 8 ? ~(~0ull << (8 * pl)) : ~0ull;
    const uint64_t khi = ph < 8 ? ~(~0ull << (8 * ph)) : ~0ull;
    const uint64_t xlo = (lo & klo) | (slo & ~klo), xhi = SHORT ? 0ull : (hi & khi) | (shi & ~khi);
    // (plain: apart from the digits there are only the sign and the point, which the number level
    // keeps; the date level keeps the digits)
    flen = isdate ? nd : len;
    elo = isdate ? xlo : lo;
    ehi = isdate ? xhi : hi;
    value = 0;
    valid = false;
    if (typed == T_NONE) return simple;
    // the digits right-aligned (the sign's byte cleared, what follows the digits leaves at the
    // top), four at a time to a number
    const bool keep = !isdate;  // the date level has taken sign and point away before the conversion
    const uint64_t zlo = xlo & ~(uint64_t)(0xFFu & (0u - sg));
    uint64_t m;
    uint32_t m32 = 0;
    uint64_t tval = 0;
    if (SHORT) {
        const uint32_t sh = 8 * (8 - nd - sg);
        const uint64_t y = (sh & 64) ? 0ull : zlo << sh;
        m32 = __umul24(parse4((uint32_t)y), 10000u) + parse4((uint32_t)(y >> 32));
        m = m32;
    } else {
        const uint32_t sh = 8 * (16 - nd - sg);
        const uint64_t up = (sh & 64) ? zlo : xhi, dn = (sh & 64) ? 0ull : zlo;
        const uint32_t t = sh & 63;
        const uint64_t yhi = (sh & 128) ? 0ull : (up << t) | (t ? dn >> (64 - t) : 0ull);
        const uint64_t ylo = (sh & 128) ? 0ull : dn << t;
        const uint32_t hi8 = __umul24(parse4((uint32_t)ylo), 10000u) + parse4((uint32_t)(ylo >> 32));
        const uint32_t lo8 = __umul24(parse4((uint32_t)yhi), 10000u) + parse4((uint32_t)(yhi >> 32));
        m = (uint64_t)hi8 * 100000000ull + lo8;
        if (TIME) tval = time_from_digits(ylo, yhi, tz_offset);
    }
    if (TIME) {  // fourteen digits, whatever else the text holds
        valid = (nd == 14) & convert;
        value = valid ? tval : 0;
        return simple;
    }
    if (F32) {
        // binary32: the quotient rounded to binary64 and that to binary32 is the decimal rounded
        // once, unless the binary64 value lies exactly half way between two binary32 values (then
        // the first rounding may have hidden on which side the decimal lies: second stage)
        const bool neg32 = keep & (sg != 0) & (c0 == '-');
        const uint32_t k10 = keep ? (uint32_t)__builtin_popcount(D >> (p + 1)) : 0u;
        const double x = (SHORT ? (double)m32 : (double)m) / p10[k10];
        const bool half = ((uint32_t)__double_as_longlong(x) & 0x1FFFFFFFu) == 0x10000000u;
        valid = (nd != 0) & convert;
        value = valid ? (uint64_t)(__float_as_uint((float)x) | (neg32 ? 0x80000000u : 0u)) : 0;
        return simple & !half;
    }
    const bool issigned = (typed == T_I32) | (typed == T_I64) | (typed == T_F64);
    const bool neg = keep & issigned & (sg != 0) & (c0 == '-');
    bool ok = nd != 0;
    uint64_t v;
    switch (typed) {
    case T_I32:
        ok &= m <= (neg ? 0x80000000ull : 0x7fffffffull);
        v = neg ? (uint64_t)(-(int64_t)m) : m;
        break;
    case T_U32:
        ok &= m <= 0x7fffffffull;
        v = m;
        break;
    case T_I64:
    case T_U64:  // at most 16 digits: always in range
        v = neg ? (uint64_t)0 - m : m;
        break;
    default: {  // T_F64: both operands exact (fewer than 16 digits, at most 15 behind the point): one rounding
        const uint32_t k10 = keep ? (uint32_t)__builtin_popcount(D >> (p + 1)) : 0u;
        double x;
        if (SHORT) {
            // at most eight digits: the quotient by a multiplication with the rounded reciprocal and one
            // correction step, q = x r, q' = q + (x - q 10^k) r.  For every integer below 10^8 and every
            // k <= 8 that is the correctly rounded quotient (all 9 x 10^8 cases compared with the
            // division on the host), and it takes three instructions where the division takes a dozen.
            typedef double f64x2 __attribute__((ext_vector_type(2)));
            const f64x2 pr = ((const f64x2 *)(p10 + P10_PAIR_AT))[k10];  // {10^k, 1 / 10^k}
            const double xm = (double)m32;
            const double q = xm * pr.y;
            x = __builtin_fma(__builtin_fma(-q, pr.x, xm), pr.y, q);
        } else {
            x = (double)m / p10[k10];
        }
        v = (uint64_t)__double_as_longlong(x) | (neg ? 1ull << 63 : 0ull);
        break;
    }
    }
    valid = ok & convert;
    value = valid ? v : 0;
    return simple;
}

// stores the first n (<= 16) bytes of the window at dst with at most five unaligned stores
__device__ __forceinline__ void win_store(uint8_t *dst, uint64_t lo, uint64_t hi, uint32_t n) {
    if (n & 16) {
        __builtin_memcpy(dst, &lo, 8);
        __builtin_memcpy(dst + 8, &hi, 8);
        return;
    }
    if (n & 8) {
        __builtin_memcpy(dst, &lo, 8);
        dst += 8;
        lo = hi;
    }
    if (n & 4) {
        const uint32_t w = (uint32_t)lo;
        __builtin_memcpy(dst, &w, 4);
        dst += 4;
        lo >>= 32;
    }
    if (n & 2) {
        const uint16_t w = (uint16_t)lo;
        __builtin_memcpy(dst, &w, 2);
        dst += 2;
        lo >>= 16;
    }
    if (n & 1) *dst = (uint8_t)lo;
}

// ORs the first n (<= 16; WIDE: some cell of the wave has more than 8) bytes of the window into a
// cleared LDS area at byte x: the bytes are shifted to x's place within aligned words (one byte
// permute per word) and go in with aligned word ORs, which neighbouring cells sharing a word
// may do in any order.  No branch on n, no unaligned access.
template <bool WIDE>
__device__ __forceinline__ void win_or(uint8_t *area, uint32_t x, uint64_t lo, uint64_t hi, uint32_t n) {
    const uint32_t nlo = n < 8 ? n : 8;
    lo &= nlo < 8 ? ~(~0ull << (8 * nlo)) : ~0ull;
    const uint32_t w0 = (uint32_t)lo, w1 = (uint32_t)(lo >> 32);
    const uint32_t sel = 0x07060504u - 0x01010101u * (x & 3);  // byte i of a word: byte 4 - (x & 3) + i of (word, word before)
    uint32_t *p = (uint32_t *)(area + (x & ~3u));
    atomicOr(p, __builtin_amdgcn_perm(w0, 0u, sel));
    atomicOr(p + 1, __builtin_amdgcn_perm(w1, w0, sel));
    if (!WIDE) {
        atomicOr(p + 2, __builtin_amdgcn_perm(0u, w1, sel));
    } else {
        const uint32_t nhi = n - nlo;
        hi &= nhi < 8 ? ~(~0ull << (8 * nhi)) : ~0ull;
        const uint32_t w2 = (uint32_t)hi, w3 = (uint32_t)(hi >> 32);
        atomicOr(p + 2, __builtin_amdgcn_perm(w2, w1, sel));
        atomicOr(p + 3, __builtin_amdgcn_perm(w3, w2, sel));
        atomicOr(p + 4, __builtin_amdgcn_perm(0u, w3, sel));
    }
}

// Raises a status bit: in the device's status word and, for the write pass, by the first thread that
// raises it, in the flag bytes that lie in memory of the host (which reads them after the wait for
// the stream, without a copy).
__device__ __forceinline__ void raise_status(const TileParams &P, uint32_t bit) {
    const uint32_t old = atomicOr(P.status, bit);
    if (!(old & bit) && P.host_flags) P.host_flags[__builtin_ctz(bit)] = 1;
}

constexpr int RB_PASSES = 2;  // records per batch <= 64 * RB_PASSES

// descriptor of one cell: source start (15 bits) | source length (15 bits) | filter level (2)
__device__ __forceinline__ uint32_t desc_pack(uint32_t start, uint32_t len, uint32_t lvl) {
    return start | (len << 15) | (lvl << 30);
}

// cell map word: source start (15 bits) | source length (14) | filter level (2) | found (1)
constexpr uint32_t MAP_MAXLEN = 1u << 14;
__device__ __forceinline__ uint32_t map_pack(uint32_t start, uint32_t len, uint32_t lvl, bool found) {
    return start | (len << 15) | (lvl << 29) | ((uint32_t)found << 31);
}

// what a wave needs, besides the cells' sources, to turn one column of a batch into output
struct CellEnv {
    const uint8_t *sdata;  // LDS image of the tile and its look-behind
    bool has_nul, nulcut;
    const double *p10;
    int32_t tz_offset = 0;  // time columns: seconds east of UTC
    uint32_t W;
    uint32_t *vmask;
    uint8_t *wstage;  // this wave's staging area for cell bytes
    uint32_t wstage_cap;
    bool plain_ok = false; // no NUL cut in this tile: passes of plain cells may take the straight-line path
    bool nostore = false;  // diagnostic builds: measure without the column's stores
    uint32_t dbg = 0;      // diagnostic builds: 64 no conversion, 128 cells not staged, 256 staged but not copied out,
                           // 512 no value/offset stores
};

// pointers known to address global memory (a pointer that comes out of a table in memory is
// generic to the compiler, which then emits the slower flat stores)
typedef __attribute__((address_space(1))) uint8_t gmem_u8;
typedef __attribute__((address_space(1))) uint32_t gmem_u32;
typedef __attribute__((address_space(1))) uint64_t gmem_u64;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) u32x4 gmem_u128;

// where one column's output goes (copied to LDS at kernel start: plain loads from the pointer
// tables in global memory would queue behind the wave's outstanding stores)
struct ColOut {
    uint32_t *off;
    uint8_t *bytes;
    uint64_t *val;
    uint64_t base;  // byte position of the tile's first cell in the column arena
};

// One column of one batch, handled by one wave: lane l has records l and l + 64.  From the
// cells' sources (start, length, filter level, found) it measures, filters and converts them
// and, when WRITE, stores values, end offsets and the filtered bytes (staged in LDS, then
// 16-byte stores).  gb = global byte position of the batch's first cell in the column arena,
// rowbase = global row of the batch's first record.  Returns the column's bytes in this batch.
// room = bytes the sizing pass counted for this column in this tile that are not yet written.
// SPECIAL: the straight-line path also converts time and binary32 columns (instances of their
// own; a kernel for schemas without such columns does not carry their code).
template <bool WRITE, bool SPECIAL>
__device__ __forceinline__ uint32_t column_cells(const CellEnv &E, const ColOut &O, uint32_t c, uint32_t typed,
                                                 int lane, uint32_t RBn, uint64_t rowbase, uint64_t gb,
                                                 const uint32_t (&cs_)[RB_PASSES], const uint32_t (&cl_)[RB_PASSES],
                                                 const uint32_t (&cv_)[RB_PASSES], const bool (&fd_)[RB_PASSES],
                                                 bool &deferred, uint32_t room, bool &overrun) {
    const uint8_t *const sdata = E.sdata;
    gmem_u64 *const valp = (WRITE && typed != T_NONE) ? (gmem_u64 *)O.val : nullptr;
    uint32_t cf_[RB_PASSES], cx_[RB_PASSES];
    uint64_t clo_[RB_PASSES], chi_[RB_PASSES];  // fast cells: their bytes, ready to store
    bool cfast_[RB_PASSES];
    uint32_t total = 0;
#pragma unroll
    for (int ps = 0; ps < RB_PASSES; ++ps) {
        const uint32_t r = ps * 64 + lane;
        uint32_t flen = 0;
        uint64_t elo = 0, ehi = 0;
        bool fast = false;
        uint64_t value = 0;
        bool valid = false;
        // a whole pass of plain cells takes the straight-line path
        bool plain = false;
        if (WRITE && E.plain_ok) {
            uint32_t tyu = (typed == T_BOOL || (!SPECIAL && (typed == T_F32 || typed == T_TIME))) ? (uint32_t)T_NONE : typed;
#ifdef MDC_ABLATE
            if (E.dbg & 64) tyu = T_NONE;
#endif
            tyu = (uint32_t)__builtin_amdgcn_readfirstlane((int)tyu);  // (the wave's, in a scalar register)
            const bool act = r < RBn;
            const uint4 w4 = lds_u128(sdata + cs_[ps]);  // (a lane without a record reads the image's first bytes)
            const uint32_t lenp = act ? cl_[ps] : 0u;
            const bool conv = tyu != T_NONE && (fd_[ps] || E.nulcut);
            // (a time column has an instance of its own: the other types' code stays as it is)
            const bool sc = SPECIAL && tyu == T_TIME
                                ? cell_simple<false, true>(w4, lenp, cv_[ps], tyu, conv, E.p10, E.tz_offset, flen, elo, ehi, value, valid)
                            : SPECIAL && tyu == T_F32
                                ? cell_simple<false, false, true>(w4, lenp, cv_[ps], tyu, conv, E.p10, E.tz_offset, flen, elo, ehi, value, valid)
                            : __builtin_amdgcn_ballot_w64(lenp > 8) == 0
                                ? cell_simple<true>(w4, lenp, cv_[ps], tyu, conv, E.p10, E.tz_offset, flen, elo, ehi, value, valid)
                                : cell_simple<false>(w4, lenp, cv_[ps], tyu, conv, E.p10, E.tz_offset, flen, elo, ehi, value, valid);
            plain = __builtin_amdgcn_ballot_w64(!sc) == 0 && (typed == tyu || typed == T_NONE);
            if (plain) {
                fast = true;
                if (!act) {
                    flen = 0;
                    valid = false;
                }
            } else {
                flen = 0;
                elo = ehi = value = 0;
                valid = false;
            }
        }
        if (!plain && r < RBn) {
            const uint32_t s = cs_[ps], len = cl_[ps], lvl = cv_[ps];
            const bool found = fd_[ps];
            const bool convert = typed != T_NONE && (found || E.nulcut);
            uint32_t ty = (WRITE && convert) ? typed : (uint32_t)T_NONE;  // sizing needs lengths only
#ifdef MDC_ABLATE
            if (E.dbg & 64) ty = T_NONE;
#endif
            bool later = false;
            if (ty == T_TIME) {  // calendar arithmetic is the second stage's: here the cell is text
                ty = T_NONE;
                deferred = true;
            }
            if (len <= 16) {
                const uint4 w4 = lds_u128(sdata + s);
                const uint64_t lo = (uint64_t)w4.x | ((uint64_t)w4.y << 32);
                const uint64_t hi = (uint64_t)w4.z | ((uint64_t)w4.w << 32);
                const bool cutting = lvl == LVL_CHAR && E.nulcut && E.has_nul;
                if (lvl == LVL_CHAR && ty == T_NONE && !cutting) {
                    flen = len;
                    elo = lo;
                    ehi = hi;
                    fast = true;
                } else if (!cutting && ty != T_F32 && ty != T_BOOL) {
                    fast = cell_fast(lo, hi, len, digit_mask16(w4), lvl, ty, E.p10, flen, value, valid, elo, ehi, later);
                }
                if (!fast) flen = cell_scan16<false>(lo, hi, len, lvl, E.nulcut, ty, E.p10, nullptr, value, valid, later);
                if (later) deferred = true;
            } else {
                flen = cell_filter_long<false>(sdata + s, len, lvl, E.nulcut, nullptr);
                if (ty != T_NONE) deferred = true;  // value and valid bit come from the second stage
            }
            valid = valid && convert;
        }
        if (r < RBn) {
            if (WRITE && valid) atomicOr(&E.vmask[r * E.W + (c >> 5)], 1u << (c & 31));
#ifdef MDC_ABLATE
            if (valp && !(E.dbg & 512)) valp[rowbase + r] = value;
#else
            if (valp) valp[rowbase + r] = value;
#endif
        }
        const uint32_t incl = wave_incl_scan(flen);
        clo_[ps] = elo;
        chi_[ps] = ehi;
        cfast_[ps] = fast;
        cf_[ps] = flen;
        cx_[ps] = total + incl - flen;
        total += wave_last(incl);
    }
    // The arenas are sized by the sizing pass, which measures every cell with code of its own: a
    // batch that would write more than that pass left for this column in this tile stores nothing
    // (the host then reports an internal error instead of the kernel writing out of bounds).
    if (WRITE && total > room) overrun = true;
    if (WRITE && !E.nostore && total <= room) {
        const uint32_t a = (uint32_t)(gb & 15);
        const bool staged = a + total <= E.wstage_cap;
        gmem_u32 *const offp = (gmem_u32 *)O.off;
        gmem_u8 *const bytesp = (gmem_u8 *)O.bytes;
#pragma unroll
        for (int ps = 0; ps < RB_PASSES; ++ps) {
            const uint32_t r = ps * 64 + lane;
            if (r < RBn) offp[rowbase + r + 1] = (uint32_t)(gb + cx_[ps] + cf_[ps]);
        }
        // the cells' bytes go to `base` + their offset in the batch; called once per destination
        // so that LDS and global stores are told apart at compile time
        auto put_cells = [&](uint8_t *base, bool others) {  // others: only the cells that are not fast ones
#pragma unroll
            for (int ps = 0; ps < RB_PASSES; ++ps) {
                const uint32_t r = ps * 64 + lane;
                if (r < RBn && cf_[ps] && !(others && cfast_[ps])) {
                    uint8_t *dst = base + cx_[ps];
                    uint64_t dv;
                    bool b1;
                    if (cfast_[ps]) {
                        win_store(dst, clo_[ps], chi_[ps], cf_[ps]);
                    } else if (cl_[ps] <= 16) {
                        const uint4 w4 = lds_u128(sdata + cs_[ps]);
                        cell_scan16<true>((uint64_t)w4.x | ((uint64_t)w4.y << 32), (uint64_t)w4.z | ((uint64_t)w4.w << 32),
                                          cl_[ps], cv_[ps], E.nulcut, T_NONE, nullptr, dst, dv, b1, b1);
                    } else {
                        cell_filter_long<true>(sdata + cs_[ps], cl_[ps], cv_[ps], E.nulcut, dst);
                    }
                }
            }
        };
#ifdef MDC_ABLATE
        if (E.dbg & 128) return total;
#endif
        if (staged) {
            // fast cells (their bytes are in registers) are ORed into the cleared area, the others
            // are stored byte by byte behind them (LDS takes a wave's instructions in order)
            for (uint32_t u = lane; u < (a + total + 15) / 16; u += 64) ((uint4 *)E.wstage)[u] = make_uint4(0, 0, 0, 0);
            bool wide = false, others = false;
            uint32_t nf_[RB_PASSES];
#pragma unroll
            for (int ps = 0; ps < RB_PASSES; ++ps) {
                const bool act = ps * 64 + (uint32_t)lane < RBn && cf_[ps];
                nf_[ps] = act && cfast_[ps] ? cf_[ps] : 0u;
                others |= act && !cfast_[ps];
                wide |= nf_[ps] > 8;
            }
            if (__builtin_amdgcn_ballot_w64(wide) == 0) {
#pragma unroll
                for (int ps = 0; ps < RB_PASSES; ++ps)
                    if (nf_[ps]) win_or<false>(E.wstage, a + cx_[ps], clo_[ps], chi_[ps], nf_[ps]);
            } else {
#pragma unroll
                for (int ps = 0; ps < RB_PASSES; ++ps)
                    if (nf_[ps]) win_or<true>(E.wstage, a + cx_[ps], clo_[ps], chi_[ps], nf_[ps]);
            }
            if (__builtin_amdgcn_ballot_w64(others)) put_cells(E.wstage + a, true);
#ifdef MDC_ABLATE
            if (E.dbg & 256) return total;
#endif
            if (total) {
                // the wave's own LDS writes are visible to it in program order
                gmem_u8 *dst = bytesp + gb;
                const uint8_t *src = E.wstage + a;
                uint32_t head = (16 - a) & 15;
                if (head > total) head = total;
                if ((uint32_t)lane < head) dst[lane] = src[lane];
                const uint32_t body = (total - head) / 16;
                for (uint32_t u = lane; u < body; u += 64)
                    *(gmem_u128 *)(dst + head + u * 16) = *(const u32x4 *)(src + head + u * 16);
                const uint32_t done = head + body * 16;
                if ((uint32_t)lane < total - done) dst[done + lane] = src[done + lane];
            }
        } else {
            put_cells((uint8_t *)(bytesp + gb), false);
        }
    }
    return total;
}

// ---------------------------------------------------------------------------------------
// One-pass parse: the sizing pass's analysis and the write in one kernel.  A tile cannot know where
// its cells go before the tiles in front of it have counted theirs, so every tile publishes its counts
// (bytes per column, rows) in one 64-bit word each and adds up what the tiles in front published:
// a word is written by ONE store and read by ONE load of device scope, kind and value together, so it
// needs no fence and no flag beside it.
__device__ __forceinline__ void pub_store(uint64_t *p, uint64_t w) {
    __hip_atomic_store((gmem_u64 *)p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t pub_load(const uint64_t *p) {
    return __hip_atomic_load((const gmem_u64 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The count of all tiles in front of `tile` for one counter (`pubj`: that counter's published words, one
// per tile), by one wave: lane l looks at tile hi-1-l of the window [hi-64, hi).  The nearest inclusive
// prefix that has every nearer tile's aggregate in front of it ends the search; a window of aggregates
// only is added and the window moves on; a tile that has published nothing yet is waited for (the window
// then begins at it).  Blocks start in index order on each of the chip's dispatchers, so the first tile
// that has not finished is always resident and the wait ends; should it not, the tile gives up after
// TileParams::spin_limit polls (at once when another tile has given up) and the host parses the input in two passes.
__device__ __forceinline__ uint64_t fused_lookback(const uint64_t *pubj, uint32_t tile, int lane, const uint32_t *status, uint32_t spin_limit,
                                                   bool &giveup) {
    uint64_t excl = 0;
    int64_t hi = (int64_t)tile;
    uint32_t spins = 0;
    while (hi > 0) {
        const int64_t idx = hi - 1 - lane;
        const uint64_t w = idx >= 0 ? pub_load(pubj + idx) : PUB_INCL;  // (in front of tile 0: an inclusive prefix of 0)
        const uint32_t kind = (uint32_t)(w >> 62);
        const uint64_t pm = __builtin_amdgcn_ballot_w64(kind != 0), im = __builtin_amdgcn_ballot_w64(kind == 2);
        const uint32_t np = ~pm ? (uint32_t)__builtin_ctzll(~pm) : 64u;  // lanes [0, np) have published
        const uint32_t fi = im ? (uint32_t)__builtin_ctzll(im) : 64u;   // the nearest inclusive prefix
        const uint32_t take = fi < np ? fi : np;                          // aggregates of lanes [0, take)
        const uint32_t agg = (uint32_t)lane < take ? (uint32_t)w : 0u;  // (a tile's own count fits 32 bits)
        excl += wave_last(wave_incl_scan(agg));
        if (fi < np) {
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)w, (int)fi);
            const uint32_t hi32 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(w >> 32), (int)fi);
            return excl + ((((uint64_t)hi32 << 32) | lo) & PUB_VALUE);
        }
        hi -= take;
        if (np < 64) {  // tile hi-1 has published nothing yet
            // (once one tile has given up the result is dropped anyway: the others need not wait their turn out)
            if (++spins > spin_limit ||
                ((spins & 255u) == 0 && (__hip_atomic_load((const gmem_u32 *)status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & ST_GIVEUP))) {
                giveup = true;
                return excl;
            }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    return excl;
}

// One column of a tile whose records all follow the tag template, by one wave: lane l has records l and
// l + 64.  Only the straight-line cell routine (cell_simple) is there: false = some cell is not plain,
// the column's type has no such routine, or the staging area is full, and the tile is left to the full
// write kernel.  Values go to their rows at once (valp: this tile's first row of the column's values, or
// null); the cells' bytes are ORed into a cleared piece of the staging area, 16 bytes behind the piece's
// start `area_out`, and the end of every cell within the column's bytes of the tile goes to endl[record].
template <bool SPECIAL>
__device__ __forceinline__ bool fused_column(const CellEnv &E, uint32_t c, uint32_t typed, int lane, uint32_t RBn, gmem_u64 *valp,
                                             const uint32_t (&cs_)[RB_PASSES], const uint32_t (&cl_)[RB_PASSES], uint32_t lvl,
                                             bool found, uint16_t *endl, uint8_t *stage, uint32_t stage_cap, uint32_t *stage_top,
                                             uint32_t &total_out, uint32_t &area_out) {
    const uint32_t tyu = (uint32_t)__builtin_amdgcn_readfirstlane((int)typed);
    if (tyu == T_BOOL || (!SPECIAL && (tyu == T_F32 || tyu == T_TIME))) return false;
    const bool conv = tyu != T_NONE && (found || E.nulcut);
    uint32_t cf_[RB_PASSES], cx_[RB_PASSES];
    uint64_t clo_[RB_PASSES], chi_[RB_PASSES];
    uint32_t total = 0;
    bool ok = true;
#pragma unroll
    for (int ps = 0; ps < RB_PASSES; ++ps) {
        const uint32_t r = ps * 64 + lane;
        const bool act = r < RBn;
        const uint4 w4 = lds_u128(E.sdata + cs_[ps]);  // (a lane without a record reads the image's first bytes)
        const uint32_t lenp = act ? cl_[ps] : 0u;
        uint32_t flen = 0;
        uint64_t elo = 0, ehi = 0, value = 0;
        bool valid = false;
        const bool sc = SPECIAL && tyu == T_TIME
                            ? cell_simple<false, true>(w4, lenp, lvl, tyu, conv, E.p10, E.tz_offset, flen, elo, ehi, value, valid)
                        : SPECIAL && tyu == T_F32
                            ? cell_simple<false, false, true>(w4, lenp, lvl, tyu, conv, E.p10, E.tz_offset, flen, elo, ehi, value, valid)
                        : __builtin_amdgcn_ballot_w64(lenp > 8) == 0
                            ? cell_simple<true>(w4, lenp, lvl, tyu, conv, E.p10, E.tz_offset, flen, elo, ehi, value, valid)
                            : cell_simple<false>(w4, lenp, lvl, tyu, conv, E.p10, E.tz_offset, flen, elo, ehi, value, valid);
        ok &= __builtin_amdgcn_ballot_w64(!sc) == 0;
        if (!act) {
            flen = 0;
            valid = false;
        }
        if (act) {
            if (valid) atomicOr(&E.vmask[r * E.W + (c >> 5)], 1u << (c & 31));
            if (valp) valp[r] = value;
        }
        const uint32_t incl = wave_incl_scan(flen);
        clo_[ps] = elo;
        chi_[ps] = ehi;
        cf_[ps] = flen;
        cx_[ps] = total + incl - flen;
        total += wave_last(incl);
    }
    if (!ok || total > 0xFFFFu) return false;
    // a piece of the staging area: 16 bytes in front (the copy-out reads from the destination's 16-byte
    // phase), the bytes, and what the word-wise ORs and reads may touch behind them
    const uint32_t need = (16 + total + 20 + 15) & ~15u;
    uint32_t top = 0;
    if (lane == 0) top = atomicAdd(stage_top, need);
    top = (uint32_t)__builtin_amdgcn_readfirstlane((int)top);
    if (top + need > stage_cap) return false;
    uint8_t *area = stage + top;
    for (uint32_t u = lane; u < need / 16; u += 64) ((uint4 *)area)[u] = make_uint4(0, 0, 0, 0);
    bool wide = false;
#pragma unroll
    for (int ps = 0; ps < RB_PASSES; ++ps) wide |= cf_[ps] > 8;
    if (__builtin_amdgcn_ballot_w64(wide) == 0) {
#pragma unroll
        for (int ps = 0; ps < RB_PASSES; ++ps)
            if (cf_[ps]) win_or<false>(area, 16 + cx_[ps], clo_[ps], chi_[ps], cf_[ps]);
    } else {
#pragma unroll
        for (int ps = 0; ps < RB_PASSES; ++ps)
            if (cf_[ps]) win_or<true>(area, 16 + cx_[ps], clo_[ps], chi_[ps], cf_[ps]);
    }
#pragma unroll
    for (int ps = 0; ps < RB_PASSES; ++ps) {
        const uint32_t r = ps * 64 + lane;
        if (r < RBn) endl[r] = (uint16_t)(cx_[ps] + cf_[ps]);
    }
    total_out = total;
    area_out = top;
    return true;
}

// The staged bytes of one column of a tile (src: LDS, 16-byte aligned) to their place in the column's
// arena, which is known by now: 16-byte stores at the destination's phase, the LDS side read as aligned
// words shifted into place.
__device__ __forceinline__ void fused_copy_out(const uint8_t *src, uint32_t total, gmem_u8 *dst, int lane) {
    uint32_t head = (16 - (uint32_t)((size_t)dst & 15)) & 15;
    if (head > total) head = total;
    if ((uint32_t)lane < head) dst[lane] = src[lane];
    const uint32_t body = (total - head) / 16;
    for (uint32_t u = lane; u < body; u += 64) {
        const uint4 v = lds_bytes16(src + head + u * 16);
        u32x4 o;
        o[0] = v.x;
        o[1] = v.y;
        o[2] = v.z;
        o[3] = v.w;
        *(gmem_u128 *)(dst + head + u * 16) = o;
    }
    const uint32_t done = head + body * 16;
    if ((uint32_t)lane < total - done) dst[done + lane] = src[done + lane];
}

// MODE: the sizing pass, the full write pass, or the one-pass parse (the sizing pass's analysis, then for a
// tile whose records all follow the tag template the straight-line write; output places come from the
// tiles in front through published counts, see fused_* above).
enum : int { PASS_SIZING = 0, PASS_WRITE = 1, PASS_FUSED = 2, PASS_FUSED_SPECIAL = 3 };  // (SPECIAL: see column_cells)
template <int MODE>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(4))) void tile_kernel(const TileParams P, const OutPtrs O) {
    constexpr bool WRITE = MODE == PASS_WRITE;
    constexpr bool FUSED = MODE == PASS_FUSED || MODE == PASS_FUSED_SPECIAL;
    extern __shared__ __align__(16) uint8_t smem[];
    const LdsLayout &L = P.L;
    uint8_t *sdata = smem + L.data;
    uint16_t *evpos = (uint16_t *)(smem + L.evlist);         // dense list of '<' events: LDS index
    uint32_t *evt = (uint32_t *)(smem + L.evt);
    uint32_t *desc = (uint32_t *)(smem + L.evt);  // delimited mode has no tag events: same space
    uint32_t *recend = (uint32_t *)(smem + L.recend);
    uint32_t *vmask = (uint32_t *)(smem + L.vmask);
    uint32_t *fmask = (uint32_t *)(smem + L.fmask);
    uint8_t *stage = smem + L.evlist;  // the event list is dead when cells are staged
    uint32_t *ms = (uint32_t *)(smem + L.misc);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (uniform: scalar loads for per-column data)
    STAMP_INIT;

    if (WRITE && P.only_unmapped && P.tile_hdr[2 * blockIdx.x] < HDR_LONGREC) return;  // the emit kernel has it
    const int64_t n = (int64_t)P.n;
    const int64_t tile_start = (int64_t)blockIdx.x * TILE;
    const int64_t tile_end = (tile_start + TILE < n) ? tile_start + TILE : n;
    const int LB = (int)L.lbmax;                // look-behind capacity of this launch
    const int64_t base_pos = tile_start - LB;  // global position of LDS index 0
    const int ts_li = LB;
    const int te_li = LB + (int)(tile_end - tile_start);
    const int li_min = base_pos < 0 ? (int)(-base_pos) : 0;

    auto load_chunk = [&](int j) -> uint4 {  // 16-byte chunk j of the LDS image, from global memory
        const int64_t pos = base_pos + (int64_t)j * CHUNK;
        if (pos + CHUNK <= n) return *(const uint4 *)(P.data + pos);
        uint32_t w[4] = {0, 0, 0, 0};
        for (int k = 0; k < CHUNK; ++k)
            if (pos + k < n) w[k >> 2] |= (uint32_t)P.data[pos + k] << (8 * (k & 3));
        return make_uint4(w[0], w[1], w[2], w[3]);
    };

    // event index at the start of every record of the batch (entry 0: minus the events of the
    // tile's first record that lie in the look-behind), then the template's entries
    int16_t *e0t = (int16_t *)(smem + L.tmpl);
    const uint32_t e0_bytes = ((L.RB + 2) * 2 + 15) & ~15u;
    const uint32_t plan_bytes = (P.ncols * 8 + 15) & ~15u;
    uint2 *plan_l = (uint2 *)(smem + L.tmpl + e0_bytes);  // the template's column plan
    uint4 *tmpl_l = (uint4 *)(smem + L.tmpl + e0_bytes + plan_bytes);
    const uint32_t tmpl_room = L.tmpl_bytes > e0_bytes + plan_bytes ? (L.tmpl_bytes - e0_bytes - plan_bytes) / 16 : 0;
    const uint32_t tmpl_cap = tmpl_room < TMPL_MAX ? tmpl_room : TMPL_MAX;
    const uint32_t nT_in = P.tmpl[0];

    // ---- A. coalesced 16-byte loads of the tile, the first look-behind window and the schema.
    // Every load is requested before the first one is waited for; for that no load may stand
    // behind a branch (the compiler waits at the end of a branch for what was loaded in it), so
    // a tile that lies whole in the input takes plain loads, and a lane without a look-behind or
    // schema chunk of its own loads one it then drops.
    int loaded_lo = ts_li, jlb = 0;
    bool lbmine = false;
    if (tile_start > 0) {
        loaded_lo = ts_li - LB_FIRST < li_min ? li_min : ts_li - LB_FIRST;
        jlb = loaded_lo / CHUNK + tid;
        lbmine = jlb < ts_li / CHUNK;
    }
    {
        const uint4 *g = (const uint4 *)P.schema;
        uint4 *s = (uint4 *)(smem + L.schema);
        const uint32_t sch16 = P.schema_bytes / 16;
        uint4 tv[4], lbv;
        if (tile_start + TILE <= n) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                tv[k] = *(const uint4 *)(P.data + (base_pos + (int64_t)(ts_li / CHUNK + k * BLOCK + tid) * CHUNK));
            lbv = *(const uint4 *)(P.data + (base_pos + (int64_t)(lbmine ? jlb : ts_li / CHUNK) * CHUNK));
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int j = ts_li / CHUNK + k * BLOCK + tid;
                tv[k] = j * CHUNK < te_li ? load_chunk(j) : make_uint4(0, 0, 0, 0);
            }
            lbv = lbmine ? load_chunk(jlb) : make_uint4(0, 0, 0, 0);
        }
        const uint4 sv = g[(uint32_t)tid < sch16 ? tid : 0];
        // tag template (sizing pass of tagged input): entry k sits behind the event-index array
        // in the space of the two row masks, which that pass does not use
        const uint32_t plan16 = (P.ncols + 1) / 2;  // 16-byte pieces of the plan, behind the entries
        const uint4 tq = ((const uint4 *)P.tmpl)[(uint32_t)tid <= TMPL_MAX + plan16 ? tid : 0];
#pragma unroll
        for (int k = 0; k < 4; ++k) ((uint4 *)sdata)[ts_li / CHUNK + k * BLOCK + tid] = tv[k];
        if (lbmine) ((uint4 *)sdata)[jlb] = lbv;
        if ((uint32_t)tid < sch16) s[tid] = sv;
        if (P.use_tmpl && tid >= 1 && (uint32_t)tid <= tmpl_cap) {
            // (pattern index -> the pattern's row of first_pos: at most 255 x 128 entries, below TMPL_NOPAT)
            uint4 e = tq;
            const uint32_t idx = e.w >> 16;
            e.w = (e.w & 0xFFFFu) | ((idx == TMPL_NOPAT ? TMPL_NOPAT : idx * L.RB) << 16);
            tmpl_l[tid - 1] = e;
        }
        if (P.use_tmpl && tmpl_room && (uint32_t)tid > TMPL_MAX && (uint32_t)tid <= TMPL_MAX + plan16)
            ((uint4 *)plan_l)[tid - 1 - TMPL_MAX] = tq;
        for (uint32_t i = tid + BLOCK; i < sch16; i += BLOCK) s[i] = g[i];
        if (tid < MS_COUNT) ms[tid] = tid == MS_LAST ? NONE : 0;
        if (FUSED && tid >= MS_COUNT && tid < MS_XCOUNT) ms[tid] = 0;
    }
    __syncthreads();
    const SchemaHdr *H = (const SchemaHdr *)(smem + L.schema);
    const DevCol *cols = (const DevCol *)(H + 1);
    const uint32_t ncols = H->ncols, npat = H->npat, W = H->words;
    const HtEntry *ht = (const HtEntry *)((const uint8_t *)cols + ((ncols * sizeof(DevCol) + 15) & ~15u));
    const bool csv = H->mode == MODE_CSV;
    const bool carry = H->carry != 0;
    // delimited mode, sizing pass, a separator of at most four bytes: separators are collected like tag
    // events and the field descriptors are made in parallel (otherwise: one lane walks each record)
    const bool sepfast = csv && !WRITE && H->sep_fast;
    // tagged input, sizing pass: records that repeat the sample record's tags take the template
    const bool tmode = !csv && P.use_tmpl && nT_in != 0 && nT_in <= tmpl_cap;
    const uint32_t nT = nT_in;
    uint32_t *coltot = (uint32_t *)(smem + L.coltot);  // write pass: the sizing pass's bytes per column of this tile
    uint32_t *tiletot = coltot + (ncols + 1);          // totals of the batches done so far
    for (uint32_t c = tid; c < ncols + 1; c += BLOCK) tiletot[c] = 0;
    STAMP(0);  // loads
    if (MODE == PASS_SIZING && SKIP(8)) return;  // (diagnostic builds: the kernel up to here)

    // searches the staged chunks [lo, hi) for record terminators; keeps the last one
    auto search_terminators = [&](int lo_li, int hi_li) {
        for (int j = lo_li / CHUNK + tid; j < hi_li / CHUNK; j += BLOCK) {
            const uint4 c4 = ((const uint4 *)sdata)[j];
            uint32_t nl = mask16_eq(c4, 0x0A0A0A0Au);
            while (nl) {
                const int b = __builtin_ctz(nl);
                nl &= nl - 1;
                const int p = j * CHUNK + b;
                bool t;
                if (p >= 8 && p - (int)H->endbz_len >= lo_li) {  // (is_term reads the eight bytes in front of p)
                    t = is_term(sdata, H, base_pos, p);
                } else {  // endbz bytes below the staged region are read from global memory
                    const int el = (int)H->endbz_len;
                    t = base_pos + p - el >= 0;
                    for (int k = 0; t && k < el; ++k) {
                        const int li = p - el + k;
                        const uint32_t bb = li >= lo_li ? sdata[li] : P.data[base_pos + li];
                        t = bb == endbz_text(H)[k];
                    }
                }
                if (t) atomicMax((int *)&ms[MS_LAST], p);
            }
        }
    };

    // ---- B. classify this lane's 64 bytes: terminators, '<', NUL; first look-behind search
    const int my_li = ts_li + tid * 64;
    uint64_t tm = 0;  // record terminators
    uint64_t em;      // event candidates: '<' in XML mode (delimited mode does not use events)
    {
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = ((const uint4 *)sdata)[my_li / CHUNK + k];
        uint64_t nlm = mask64_eq(v, 0x0A0A0A0Au);
        em = !csv ? mask64_eq(v, 0x3C3C3C3Cu) : sepfast ? mask64_eq(v, rep4(H->sep[0])) : 0;
        bool nul = any_zero64(v);
        const int lim = te_li - my_li;  // bytes of this lane that belong to the input
        if (lim < 64) {
            const uint64_t keep = lim <= 0 ? 0 : (~0ull >> (64 - lim));
            nlm &= keep;
            em &= keep;
            nul = (mask64_eq(v, 0u) & keep) != 0;  // (the zeros behind the input's end are not the input's)
        }
        if (nul) ms[MS_HASNUL] = 1;
        while (nlm) {
            const int b = __builtin_ctzll(nlm);
            nlm &= nlm - 1;
            if (is_term(sdata, H, base_pos, my_li + b)) tm |= 1ull << b;
        }
        if (sepfast && H->sep_len > 1) {
            // A separator of several bytes: every match of its first byte is compared in full (a match
            // that would pass the tile's end belongs to a record of the next tile).  The greedy search
            // of ccmdstr::splittocmd (public/_public.cpp:274-302) goes on behind a separator it found,
            // so of two matches that overlap ("~!~!~") it sees the first only: a tile with such a
            // pair is left to the walk that does the same, all others may take every match.
            const uint32_t sl = H->sep_len;
            const uint32_t want = (uint32_t)H->sep[0] | ((uint32_t)H->sep[1] << 8) | ((uint32_t)H->sep[2] << 16) |
                                  ((uint32_t)H->sep[3] << 24);
            const uint32_t care = sl >= 4 ? ~0u : (1u << (8 * sl)) - 1;
            auto match_at = [&](int li) {
                uint32_t w0, w1;
                lds_bytes8(sdata + li, w0, w1);
                return ((w0 ^ want) & care) == 0;
            };
            uint64_t t = em, full = 0;
            while (t) {
                const int b = __builtin_ctzll(t);
                t &= t - 1;
                if (match_at(my_li + b)) full |= 1ull << b;
            }
            em = full;
            uint64_t close = em << 1;
            if (sl > 2) close |= em << 2;
            if (sl > 3) close |= em << 3;
            bool overlap = (em & close) != 0;
            // ... and a match in the last bytes in front of this lane's that reaches one of this lane's
            const uint32_t low = (uint32_t)em & ((1u << (sl - 1)) - 1);
            if (low)
                for (int d = 1; d < (int)sl; ++d)
                    if (my_li - d >= li_min && my_li - d >= loaded_lo && (low & ((1u << (sl - d)) - 1)) && match_at(my_li - d))
                        overlap = true;
            if (overlap) ms[MS_SEPOV] = 1;
        }
    }
    if (tile_start > 0) search_terminators(loaded_lo, ts_li);
    uint32_t rank_base, ev_base, nrec, nev_tile;
    // ranks every byte by record and numbers the events: packed block-wide prefix sum of this lane's
    // terminators and event candidates
    auto rank_scan = [&]() {
        const uint32_t mine = ((uint32_t)__builtin_popcountll(tm) << 16) | (uint32_t)__builtin_popcountll(em);
        const uint32_t incl = wave_incl_scan(mine);
        if (lane == 63) ms[MS_WTOT + wave] = incl;
        __syncthreads();
        uint32_t base = incl - mine, tot = 0;
#pragma unroll
        for (int w = 0; w < BLOCK / 64; ++w) {
            const uint32_t t = ms[MS_WTOT + w];
            if (w < wave) base += t;
            tot += t;
        }
        rank_base = base >> 16;
        ev_base = base & 0xFFFFu;
        nrec = tot >> 16;
        nev_tile = tot & 0xFFFFu;
    };
    rank_scan();
    const uint32_t nrec_all = nrec;  // records the tile owns (one more than it goes through when the first is a long one)
    // one-pass parse: the tiles behind learn this tile's rows as early as they are known
    if (FUSED && tid == 0) pub_store(P.pub + (uint64_t)ncols * P.ntiles + blockIdx.x, PUB_AGG | nrec_all);
    // start of the first owned record: behind the last terminator in front of the tile (a tile
    // that ends no record owns none and has nothing to look for)
    int s0_li = ts_li;
    bool longrec = false;  // (block-uniform) the first record starts in front of the look-behind capacity
    if (tile_start > 0 && nrec > 0) {
        for (;;) {
            const int last = (int)ms[MS_LAST];
            if (last >= 0) {
                s0_li = last + 1;
                break;
            }
            if (base_pos + loaded_lo == 0) {  // reached the start of the input
                s0_li = loaded_lo;
                break;
            }
            if (loaded_lo == 0) {
                longrec = true;
                break;
            }
            // rare: extend the staged region backwards and search it
            const int hi_li = loaded_lo;
            loaded_lo = hi_li - LB_STEP < li_min ? li_min : hi_li - LB_STEP;
            for (int j = loaded_lo / CHUNK + tid; j < hi_li / CHUNK; j += BLOCK) ((uint4 *)sdata)[j] = load_chunk(j);
            __syncthreads();
            search_terminators(loaded_lo, hi_li);
            __syncthreads();
        }
    }
    uint32_t long_entry = NONE;
    if (longrec) {
        // The record that ends at the tile's first terminator is longer than the look-behind holds.
        // It is left to long_record_kernel (one block per such record, straight from global memory);
        // this tile goes on as if it began behind that terminator, with one record less.
        if (rank_base == 0 && tm) ms[MS_T0] = (uint32_t)(my_li + __builtin_ctzll(tm));  // (one lane: the first with a terminator)
        __syncthreads();
        const int t0 = (int)ms[MS_T0];
        if (!WRITE) {
            if (tid == 0) {
                const uint32_t e = atomicAdd(P.status + 2, 1u);
                LongRec R;
                R.gend = (uint64_t)(base_pos + t0);
                R.from = (uint64_t)base_pos;  // (no terminator between here and the tile)
                R.tile = blockIdx.x;
                R.pad = 0;
                P.long_list[e] = R;
                ms[MS_LONGE] = e;
            }
        } else if (tid == 0) {
            ms[MS_LONGE] = P.tile_hdr[2 * blockIdx.x + 1];
        }
        if (my_li + 63 < t0) {  // (a lane whose bytes all lie in front of that terminator)
            em = 0;
        } else if (my_li <= t0) {
            const int b = t0 - my_li;
            tm &= ~(1ull << b);
            em = b == 63 ? 0 : em & (~0ull << (b + 1));
        }
        __syncthreads();  // (the wave totals of the first scan have been read; MS_LONGE is written)
        long_entry = ms[MS_LONGE];
        rank_scan();
        s0_li = t0 + 1;
    }
    // Delimited input whose separators are found with byte masks: the separators of the first record's part
    // in the look-behind, one chunk per lane of wave 0, each with its place in the tile's list of separators
    // (they come first in it).  The record is then measured like every other; otherwise one lane walks it
    // while the block waits, and nearly every tile's first record begins in the look-behind.
    uint32_t lbsep_m = 0, lbsep_k = 0;
    bool lbfast = false;  // (block-uniform)
    if (sepfast && tile_start > 0 && s0_li < ts_li && ts_li / CHUNK - s0_li / CHUNK <= 64) {
        lbfast = true;
        if (wave == 0) {
            const int j = s0_li / CHUNK + lane;
            if (j < ts_li / CHUNK) {
                lbsep_m = mask16_eq(((const uint4 *)sdata)[j], rep4(H->sep[0]));
                if (j * CHUNK < s0_li) lbsep_m &= ~0u << (s0_li - j * CHUNK);
            }
            const uint32_t sl = H->sep_len;
            if (sl > 1 && j < ts_li / CHUNK) {
                // several bytes: every match of the first byte compared in full; two matches that overlap
                // leave the tile to the walk, as for the tile's own bytes (classification above)
                const uint32_t want = (uint32_t)H->sep[0] | ((uint32_t)H->sep[1] << 8) | ((uint32_t)H->sep[2] << 16) |
                                      ((uint32_t)H->sep[3] << 24);
                const uint32_t care = sl >= 4 ? ~0u : (1u << (8 * sl)) - 1;
                auto match_at = [&](int li) {
                    uint32_t w0, w1;
                    lds_bytes8(sdata + li, w0, w1);
                    return ((w0 ^ want) & care) == 0;
                };
                uint32_t t = lbsep_m, full = 0;
                while (t) {
                    const int b = __builtin_ctz(t);
                    t &= t - 1;
                    if (match_at(j * CHUNK + b)) full |= 1u << b;
                }
                lbsep_m = full;
                uint32_t close = full << 1;
                if (sl > 2) close |= full << 2;
                if (sl > 3) close |= full << 3;
                bool overlap = (full & close) != 0;
                const uint32_t low = full & ((1u << (sl - 1)) - 1);
                if (low)
                    for (int d = 1; d < (int)sl; ++d)
                        if (j * CHUNK - d >= s0_li && (low & ((1u << (sl - d)) - 1)) && match_at(j * CHUNK - d)) overlap = true;
                if (overlap) ms[MS_SEPOV] = 1;
            }
            const uint32_t cnt = (uint32_t)__builtin_popcount(lbsep_m);
            const uint32_t incl = wave_incl_scan(cnt);
            lbsep_k = incl - cnt;
            if (lane == 63) ms[MS_LBCNT] = incl;
        }
    }
    // NUL bytes of the first record's part in the look-behind count as well (a chunk that begins in
    // front of the record may add the record before: the flag only has to be set when a NUL is there)
    if (tile_start > 0) {
        bool nul = false;
        for (int j = s0_li / CHUNK + tid; j < ts_li / CHUNK; j += BLOCK) {
            const uint4 c4 = ((const uint4 *)sdata)[j];
            nul |= ((((c4.x - 0x01010101u) & ~c4.x) | ((c4.y - 0x01010101u) & ~c4.y) | ((c4.z - 0x01010101u) & ~c4.z) |
                     ((c4.w - 0x01010101u) & ~c4.w)) & 0x80808080u) != 0;
        }
        if (nul) ms[MS_HASNUL] = 1;
        __syncthreads();
    }
    const bool has_nul = ms[MS_HASNUL] != 0;
    // the tile's place in the cell map: map_stride record slots per tile, no allocation needed
    const uint32_t map_slot = (!WRITE && P.cellmap && nrec <= P.map_stride && !longrec) ? blockIdx.x * P.map_stride : NONE;
    const uint32_t RB = L.RB, EVCAP = L.evcap;
    const uint64_t *tb = WRITE ? P.tile_base + (uint64_t)blockIdx.x * (ncols + 1) : nullptr;
    const uint64_t row0 = WRITE ? tb[ncols] + (longrec ? 1 : 0) : 0;  // (a long first record has the tile's first row)
    ColOut *colout = (ColOut *)(smem + L.colout);
    if (WRITE) {
        // what the sizing pass counted for this tile: the rows must agree before anything is stored
        const uint32_t *tc = P.tile_counts + (uint64_t)blockIdx.x * (ncols + 1);
        if (tc[ncols] != nrec_all) {
            if (tid == 0) raise_status(P, ST_INTERNAL);
            return;
        }
        for (uint32_t c = tid; c < ncols; c += BLOCK) {
            ColOut o;
            o.off = O.off[c];
            o.bytes = O.bytes[c];
            o.val = O.val[c];
            o.base = tb[c];
            colout[c] = o;
            coltot[c] = tc[c];
            if (longrec) tiletot[c] = P.long_len[(uint64_t)long_entry * ncols + c];  // bytes the long record's cells take
        }
    }
    if ((WRITE || FUSED) && blockIdx.x == 0) {
        if (tid == 0 && P.rec_off) P.rec_off[0] = 0;
        for (uint32_t c = tid; c < ncols; c += BLOCK) O.off[c][0] = 0;
    }
    if (tid == 0) recend[0] = (uint32_t)(s0_li - 1);
    const uint32_t seplen = H->sep_len;
    // all separators of the tile, in input order (the staging area is not used by the sizing pass)
    const uint32_t lbtot = lbfast ? ms[MS_LBCNT] : 0u;  // separators in front of the tile's own in the list
    const bool fastcsv = sepfast && nev_tile + lbtot <= EVCAP + EVCAP / 2 && !ms[MS_SEPOV];
    // the first record starts in the look-behind and its separators there were not collected: walked by one lane
    const bool slow0 = fastcsv && s0_li < ts_li && !lbfast;
    uint32_t *e0 = desc + ncols * RB;             // first separator of every record of the batch (RB + 1 entries)
    if (fastcsv) {
        uint64_t m = em;
        uint32_t idx = ev_base + lbtot;
        while (m) {
            const int b = __builtin_ctzll(m);
            m &= m - 1;
            evpos[idx++] = (uint16_t)(my_li + b);
        }
        uint32_t m2 = lbsep_m, k = lbsep_k;  // (zero outside wave 0 and without the look-behind part)
        while (m2) {
            const int b = __builtin_ctz(m2);
            m2 &= m2 - 1;
            evpos[k++] = (uint16_t)((s0_li / CHUNK + lane) * CHUNK + b);
        }
    }
    const int jlb0 = s0_li / CHUNK, jlb1 = ts_li / CHUNK;  // look-behind chunks of the first record
    // template mode: the first record's '<' in the look-behind, one chunk per lane of wave 0, counted
    // so that every one of them knows its place among the record's events
    uint32_t lbm = 0, lbk = 0;
    bool irregular = false;  // (block-uniform) a record of the tile does not follow the template
    if (tmode) {
        if (jlb1 - jlb0 > 64) {
            irregular = true;
        } else if (wave == 0) {
            const int j = jlb0 + lane;
            if (j < jlb1) {
                lbm = mask16_eq(((const uint4 *)sdata)[j], 0x3C3C3C3Cu);
                if (j * CHUNK < s0_li) lbm &= ~0u << (s0_li - j * CHUNK);
            }
            const uint32_t cnt = (uint32_t)__builtin_popcount(lbm);
            const uint32_t incl = wave_incl_scan(cnt);
            lbk = incl - cnt;
            if (lane == 63) e0t[0] = (int16_t)(-(int)incl);
        }
    }
    STAMP(1);  // classify + rank scan
    if (MODE == PASS_SIZING && SKIP(4)) return;  // (diagnostic builds: the kernel up to here)

    // one-pass parse: a tile takes the straight-line write when all its records follow the template, fit
    // one batch and no text cell may need the NUL cut; its first row comes from the tiles in front
    const bool fz_try = FUSED && tmode && !longrec && nrec <= RB && !(carry && !csv && has_nul);
    bool fz = false;         // (block-uniform) the tile has taken it
    bool rows_done = false;  // (block-uniform) the rows in front of the tile have been added up
    auto rows_lookback = [&]() {
        if (FUSED && !rows_done && wave == BLOCK / 64 - 1) {
            bool gu = false;
            uint64_t *pr = P.pub + (uint64_t)ncols * P.ntiles;
            const uint64_t r0 = SKIP(2048) ? (uint64_t)blockIdx.x * 90 : fused_lookback(pr, blockIdx.x, lane, P.status, P.spin_limit, gu);
            if (lane == 0) {
                pub_store(pr + blockIdx.x, PUB_INCL | (r0 + nrec_all));
                const bool fits = r0 + nrec_all <= P.caps[ncols];
                ms[MS_ROW0] = (uint32_t)r0;
                ms[MS_ROW0 + 1] = (uint32_t)(r0 >> 32);
                ms[MS_ROWOK] = fits && !gu ? 1u : 0u;
                if (gu) raise_status(P, ST_GIVEUP);
                if (!fits) raise_status(P, ST_OVERFLOW);
                if (blockIdx.x == P.ntiles - 1) P.fused_totals[ncols] = r0 + nrec_all;
            }
        }
        rows_done = true;
    };
    if (FUSED && tmode && (uint32_t)tid < W) {  // found bits of a row that follows the template: the plan's
        uint32_t word = 0;
        for (uint32_t c = tid * 32; c < ncols && c < (uint32_t)tid * 32 + 32; ++c) word |= ((plan_l[c].x >> 10) & 1u) << (c & 31);
        ms[MS_FWORD + tid] = word;
    }

    // ---- batches of RB records
    for (uint32_t rb0 = 0; rb0 < nrec; rb0 += RB) {
        const uint32_t RBn = (nrec - rb0 < RB) ? nrec - rb0 : RB;
        for (uint32_t i = tid; i < (RB * npat + 1) / 2; i += BLOCK) evt[i] = NONE;
        if (WRITE || FUSED)  // (the sizing pass keeps the tag template there; its delimited walk stores whole words)
            for (uint32_t i = tid; i < RB * W; i += BLOCK) {
                vmask[i] = 0;
                fmask[i] = 0;
            }
        // terminator positions of this batch
        {
            uint64_t t = tm;
            uint32_t r = rank_base;
            while (t) {
                const int b = __builtin_ctzll(t);
                t &= t - 1;
                if (r >= rb0 && r < rb0 + RBn) recend[r - rb0 + 1] = (uint32_t)(my_li + b);
                // the record behind this terminator starts with the separators that follow it
                if (fastcsv && r + 1 >= rb0 && r + 1 <= rb0 + RBn)
                    e0[r + 1 - rb0] = lbtot + ev_base + (uint32_t)__builtin_popcountll(em & ((1ull << b) - 1));
                if (tmode && r + 1 >= rb0 && r + 1 <= rb0 + RBn)
                    e0t[r + 1 - rb0] = (int16_t)(ev_base + (uint32_t)__builtin_popcountll(em & ((1ull << b) - 1)));
                ++r;
            }
            if (fastcsv && rb0 == 0 && tid == 0) e0[0] = 0;
        }
        __syncthreads();
        STAMP(2);  // clears

        bool general = !csv && !SKIP(1) && !P.records_only;
        if (tmode && general && !irregular) {
            // ---- C/D with the tag template.  Every record must have as many '<' as the sample ...
            if ((uint32_t)tid < RBn && (uint32_t)((int)e0t[tid + 1] - (int)e0t[tid]) != nT) ms[MS_IRR] = 1;
            bool ok = true;
            // ... and event k must repeat the sample's event k: the first record's events in the
            // look-behind (first batch only) ...
            if (rb0 == 0 && wave == 0 && !SKIP(128)) {
                uint32_t m = lbm, k = lbk;
                while (m) {
                    const uint32_t b = __builtin_ctz(m);
                    m &= m - 1;
                    ok &= tmpl_event(sdata, tmpl_l, nT, (uint32_t)((jlb0 + lane) * CHUNK) + b, k++, evt, 0);
                }
            }
            // ... and this lane's own: one walk over its '<' and terminators (32 bytes at a time:
            // the bit searches stay in one register) keeps the record and the event's place in it
            {
                uint32_t rl = rank_base - rb0;  // record within the batch (wraps below it: never < RBn then)
                uint32_t k = rl <= RBn ? (uint32_t)((int)ev_base - (int)e0t[rl]) : 0u;
                auto walk = [&](uint32_t xe, uint32_t xt, uint32_t li) {
                    uint32_t x = xe | xt;
                    while (x) {
                        const uint32_t b = (uint32_t)__builtin_ctz(x);
                        x &= x - 1;
                        if ((xt >> b) & 1) {
                            ++rl;
                            k = 0;
                        } else {
                            if (rl < RBn) ok &= tmpl_event(sdata, tmpl_l, nT, li + b, k, evt, rl);
                            ++k;
                        }
                    }
                };
                walk((uint32_t)em, (uint32_t)tm, (uint32_t)my_li);
                walk((uint32_t)(em >> 32), (uint32_t)(tm >> 32), (uint32_t)my_li + 32);
            }
            if (!ok) ms[MS_IRR] = 1;
            rows_lookback();
            __syncthreads();
            irregular = ms[MS_IRR] != 0;  // (nobody writes the flag again before the next barrier)
            general = irregular;
            if (general) {  // some record differs: the whole tile goes through the table lookup
                for (uint32_t i = tid; i < (RB * npat + 1) / 2; i += BLOCK) evt[i] = NONE;
                __syncthreads();
            }
        }
        if (general) {
            // ---- C. the first record's events in the look-behind region (first batch only)
            if (rb0 == 0) {
                for (int j = jlb0 + tid; j < jlb1; j += BLOCK) {
                    const uint4 c4 = ((const uint4 *)sdata)[j];
                    uint32_t m = mask16_eq(c4, 0x3C3C3C3Cu);
                    while (m) {
                        const uint32_t b = __builtin_ctz(m);
                        m &= m - 1;
                        const uint32_t q = (uint32_t)(j * CHUNK) + b;
                        if ((int)q >= s0_li) match_tag(sdata, H, cols, ht, q, evt, 0, RB);
                    }
                }
            }
            // ---- C/D. every lane looks up the tags of its own '<' bytes (no list, no barrier: a
            // balanced event list costs more to build than the idle lanes it saves)
            {
                uint64_t m = em;
                while (m) {
                    const int b = __builtin_ctzll(m);
                    m &= m - 1;
                    const uint32_t r = rank_base + (uint32_t)__builtin_popcountll(tm & ((1ull << b) - 1));
                    if (r >= rb0 && r < rb0 + RBn) match_tag(sdata, H, cols, ht, (uint32_t)(my_li + b), evt, r - rb0, RB);
                }
            }
            rows_lookback();
            __syncthreads();
        }
        STAMP(3);  // events

        // ---- E. cells: descriptor, length, typed value
        if (csv && (!fastcsv || (slow0 && rb0 == 0))) {
            // greedy left-to-right separator search, one lane per record (ccmdstr::splittocmd)
            if ((uint32_t)tid < (fastcsv ? 1u : RBn) && !SKIP(64)) {
                const uint32_t r = tid;
                const uint32_t rs = recend[r] + 1, re = recend[r + 1];
                uint32_t fmw = 0;
                uint32_t pos = rs;
                bool done = false;
                const uint32_t s0b = H->sep[0];
                const uint8_t *sepp = sep_text(H);
                for (uint32_t c = 0; c < ncols; ++c) {
                    uint32_t fs = pos, fl = 0;
                    if (!done) {
                        uint32_t p = pos;
                        bool hit = false;
                        while (p + seplen <= re) {
                            if (sdata[p] == s0b) {
                                bool eq = true;
                                for (uint32_t k = 1; k < seplen; ++k)
                                    if (sdata[p + k] != sepp[k]) {
                                        eq = false;
                                        break;
                                    }
                                if (eq) {
                                    hit = true;
                                    break;
                                }
                            }
                            ++p;
                        }
                        if (hit) {
                            fl = p - pos;
                            pos = p + seplen;
                        } else {
                            fl = re - pos;
                            done = true;
                        }
                        if (H->bdelspace) {
                            // deletelchr then deleterchr with its "pos != 0" test
                            // (public/_public.cpp:24-31, :52-59)
                            while (fl && sdata[fs] == ' ') {
                                ++fs;
                                --fl;
                            }
                            if (fl) {
                                uint32_t lastp = fl - 1;
                                while (sdata[fs + lastp] == ' ') --lastp;  // fs itself is not a blank
                                if (lastp != 0) fl = lastp + 1;
                            }
                        }
                        const DevCol &col = cols[c];
                        if (col.collen > 0 && (uint32_t)col.collen < fl) fl = (uint32_t)col.collen;
                        fmw |= 1u << (c & 31);
                    }
                    desc[c * RB + r] = desc_pack(fs, fl, LVL_CHAR);
                    if ((c & 31) == 31 || c + 1 == ncols) {
                        fmask[r * W + (c >> 5)] = fmw;
                        fmw = 0;
                    }
                }
            }
        }
        if (csv && (!fastcsv || (slow0 && rb0 == 0))) __syncthreads();
        STAMP(4);  // delimited descriptors

        const bool nulcut = carry && !csv;
        const double *p10 = (const double *)(smem + L.schema + H->p10_off);
        // ---- E-H. one wave per column, one lane per record; everything about a column (cell
        // descriptors, lengths, typed values, offsets, staging, copy-out) stays inside the wave,
        // so no block barrier is needed until the row outputs
        bool deferred = false, overrun = false;
        CellEnv E;
        E.sdata = sdata;
        E.has_nul = has_nul;
        E.nulcut = nulcut;
        E.p10 = p10;
        E.tz_offset = H->tz_offset;
        E.W = W;
        E.vmask = vmask;
        E.wstage = stage + wave * (L.stage_cap / (BLOCK / 64));
        E.wstage_cap = L.stage_cap / (BLOCK / 64);
        E.plain_ok = !(nulcut && has_nul);
        const uint32_t slot = map_slot;
        const bool mapping = !WRITE && slot != NONE;
        // the source of cell (c, r): start, length, filter level, found
        auto fetch_cell = [&](uint32_t c, uint32_t r, uint32_t &s, uint32_t &len, uint32_t &lvl, bool &found) {
            s = 0;
            len = 0;
            lvl = LVL_CHAR;
            if (csv) {
                const uint32_t d = desc[c * RB + r];
                s = d & 0x7FFFu;
                len = (d >> 15) & 0x7FFFu;
                found = (fmask[r * W + (c >> 5)] >> (c & 31)) & 1u;
                return;
            }
            // getxmlbuffer's independent first-open / first-close search, then splitbuffer's
            // carry-forward: a missing tag keeps the nearest earlier column's value, filtered by
            // every column in between
            const uint32_t re = recend[r + 1];
            uint32_t o = evt_get(evt, (2 * c) * RB + r), e = evt_get(evt, (2 * c + 1) * RB + r);
            found = (o != NONE) & (e != NONE);
            uint32_t cs = c;
            lvl = cols[c].level;
            bool have = found;
            if (!found && carry) {
                while (cs > 0) {
                    --cs;
                    o = evt_get(evt, (2 * cs) * RB + r);
                    e = evt_get(evt, (2 * cs + 1) * RB + r);
                    const uint32_t l2 = cols[cs].level;
                    if (l2 > lvl) lvl = l2;
                    if ((o != NONE) & (e != NONE)) {
                        have = true;
                        break;
                    }
                }
            }
            if (have) {
                const uint32_t nml = cols[cs].namelen;
                const int cl = cols[cs].collen;
                s = o + nml + 2;
                int l = (int)e - (int)s;
                if (l < 0)
                    l = (int)(re - s);  // close tag in front of the open tag: rest of record
                else if (cl > 0 && cl < l)
                    l = cl;
                len = (uint32_t)l;
            }
        };
        // One-pass parse, a tile that follows the template: the straight-line write, one wave per column.
        // Values go to their rows, the cells' bytes to the staging area; a wave that meets a cell which is
        // not plain raises MS_ODD and the tile is measured below like any other.
        bool fzb = FUSED && fz_try && !irregular;
        if (FUSED && fzb) {
            const uint64_t frow0 = (uint64_t)ms[MS_ROW0] | ((uint64_t)ms[MS_ROW0 + 1] << 32);
            const bool rowok = ms[MS_ROWOK] != 0;
            uint16_t *endl = (uint16_t *)(smem + L.endl);
            for (uint32_t c = wave; c < ncols; c += BLOCK / 64) {
                const uint32_t typed = (uint32_t)__builtin_amdgcn_readfirstlane((int)cols[c].typed);
                const uint2 pl = plan_l[c];
                const uint32_t cs = pl.x & 0xFFu, nml2 = pl.x >> 16, lvl = (pl.x >> 8) & 3u;
                const int cl = (int)pl.y;
                const bool found = (pl.x >> 10) & 1u, hassrc = (pl.x >> 11) & 1u;
                uint32_t cs_[RB_PASSES], cl_[RB_PASSES];
#pragma unroll
                for (int ps = 0; ps < RB_PASSES; ++ps) {
                    const uint32_t r = ps * 64 + lane;
                    uint32_t s = 0, len = 0;
                    if (r < RBn) {
                        const uint32_t o = ((const uint16_t *)evt)[(2 * cs) * RB + r], e = ((const uint16_t *)evt)[(2 * cs + 1) * RB + r];
                        const uint32_t re = recend[r + 1];
                        const uint32_t s1 = o + nml2;
                        int l = (int)e - (int)s1;
                        l = l < 0 ? (int)(re - s1) : (cl > 0 && cl < l) ? cl : l;
                        const bool have = hassrc & (o != NONE16) & (e != NONE16);
                        s = have ? s1 : 0u;
                        len = have ? (uint32_t)l : 0u;
                    }
                    cs_[ps] = s;
                    cl_[ps] = len;
                }
                gmem_u64 *valp = (typed != T_NONE && rowok) ? (gmem_u64 *)O.val[c] + frow0 : nullptr;
                uint32_t total = 0, area = 0;
                const bool ok = fused_column<MODE == PASS_FUSED_SPECIAL>(E, c, typed, lane, RBn, valp, cs_, cl_, lvl, found, endl + c * RB,
                                                                         stage, L.stage_cap, &ms[MS_STAGE], total, area);
                if (!ok) {
                    if (lane == 0) ms[MS_ODD] = 1;
                    break;
                }
                if (lane == 0) {
                    tiletot[c] = total;
                    coltot[c] = area;  // (the sizing pass has no use for this half of the array)
                }
            }
            __syncthreads();
            if (ms[MS_ODD]) {  // measured again, the general way
                fzb = false;
                for (uint32_t c = tid; c < ncols + 1; c += BLOCK) tiletot[c] = 0;
                __syncthreads();
            } else {
                fz = true;
                if ((uint32_t)tid < RBn && rowok) {
                    const uint64_t row = frow0 + tid;
                    if (P.rec_off) P.rec_off[row + 1] = (uint64_t)(base_pos + (int64_t)recend[tid + 1] + 1);
#pragma unroll 1
                    for (uint32_t w = 0; w < W; ++w) {
                        P.valid[row * W + w] = vmask[tid * W + w];
                        P.found[row * W + w] = ms[MS_FWORD + w];
                    }
                }
            }
        }
        if (fzb) {
        } else if (P.records_only) {
            // only the records' places are wanted (cifile::readline without a field schema): no tag search,
            // no cells; every cell of the write pass is empty
            if (WRITE && (uint32_t)tid < RBn)
                for (uint32_t c = 0; c < ncols; ++c) O.off[c][row0 + rb0 + tid + 1] = 0;
        } else if (!WRITE && fastcsv) {
            // Delimited input whose separators were collected: one lane per (field, record).  Field c
            // lies between the record's separators c-1 and c; a field is text, so its length is all the
            // sizing pass needs (no descriptor in LDS, no second loop).  Only a first record that starts
            // in the look-behind was walked by one lane, which left its descriptors behind.
            const uint32_t nitems = ncols * RBn;
            const uint32_t magic = 0xFFFFFFFFu / RBn + 1;  // i / RBn == umulhi(i, magic) for i < 2^15, RBn <= 256
            for (uint32_t i0 = (uint32_t)wave * 64; i0 < nitems && !SKIP(2); i0 += BLOCK) {
                const uint32_t i = i0 + lane;
                const bool act = i < nitems;
                const uint32_t c = !act ? ncols - 1 : RBn == 1 ? i : __umulhi(i, magic);
                const uint32_t r = act ? i - c * RBn : 0;
                uint32_t fs = 0, fl = 0;
                bool found = false;
                if (act && r == 0 && rb0 == 0 && slow0) {
                    uint32_t lvl;
                    fetch_cell(c, 0, fs, fl, lvl, found);
                } else if (act) {
                    const uint32_t rs = recend[r] + 1, re = recend[r + 1];
                    const uint32_t first = e0[r], nsep = e0[r + 1] - first;
                    fs = re;
                    if (c <= nsep) {
                        fs = c ? (uint32_t)evpos[first + c - 1] + seplen : rs;
                        fl = (c < nsep ? (uint32_t)evpos[first + c] : re) - fs;
                        if (H->bdelspace) {
                            // deletelchr then deleterchr with its "pos != 0" test
                            // (public/_public.cpp:24-31, :52-59)
                            while (fl && sdata[fs] == ' ') {
                                ++fs;
                                --fl;
                            }
                            if (fl) {
                                uint32_t lastp = fl - 1;
                                while (sdata[fs + lastp] == ' ') --lastp;  // fs itself is not a blank
                                if (lastp != 0) fl = lastp + 1;
                            }
                        }
                        const int cl = cols[c].collen;
                        if (cl > 0 && (uint32_t)cl < fl) fl = (uint32_t)cl;
                        found = true;
                    }
                }
                if (mapping && act) {
                    if (fl < MAP_MAXLEN)
                        P.cellmap[(uint64_t)c * P.map_cap + slot + rb0 + r] = map_pack(fs, fl, LVL_CHAR, found);
                    else
                        ms[MS_NOMAP] = 1;
                }
                const uint32_t incl = wave_incl_scan(fl);
                // this column's run in the wave: lanes [first, last]
                const uint32_t cstart = c * RBn, cend = cstart + RBn - 1;
                const uint32_t first = cstart > i0 ? cstart - i0 : 0;
                const uint32_t last = cend - i0 < 63 ? cend - i0 : 63;
                const uint32_t before = __shfl(incl, first ? first - 1 : 0, 64);
                if (act && (uint32_t)lane == last) atomicAdd(&tiletot[c], incl - (first ? before : 0));
            }
        } else if (!WRITE) {
            // Sizing needs the cells' lengths and the columns' totals only, no order among the
            // cells: one lane per (column, record), all lanes busy.  A wave's lanes hold
            // consecutive records of (usually) one or two columns; the last lane of each column's
            // run adds the run's bytes to the column total.
            const uint32_t nitems = ncols * RBn;
            const uint32_t magic = 0xFFFFFFFFu / RBn + 1;  // i / RBn == umulhi(i, magic) for i < 2^15, RBn <= 256
            for (uint32_t i0 = (uint32_t)wave * 64; i0 < nitems && !SKIP(2); i0 += BLOCK) {
                const uint32_t i = i0 + lane;
                const bool act = i < nitems;
                const uint32_t c = !act ? ncols - 1 : RBn == 1 ? i : __umulhi(i, magic);
                const uint32_t r = act ? i - c * RBn : 0;
                uint32_t flen = 0;
                bool measured = false;
                if (tmode && !irregular && !(nulcut && has_nul)) {
                    // A tile that follows the template, a whole wave of short cells at once and
                    // without a per-lane branch: source from the column plan, 16 bytes of it, digit
                    // mask.  The date level keeps the digits, text keeps everything, and so does a
                    // number that has only a sign in front and one point besides its digits.
                    uint2 pl = plan_l[c];
                    uint32_t re = recend[r + 1];
                    // (both words of the plan and the record's end are fetched here, not behind the
                    // branches the compiler would make of the selects below)
                    asm volatile("" : "+v"(pl.x), "+v"(pl.y), "+v"(re));
                    const uint32_t cs = pl.x & 0xFFu, nml2 = pl.x >> 16;
                    const int cl = (int)pl.y;
                    const uint32_t lvl = (pl.x >> 8) & 3u;
                    const uint32_t o = ((const uint16_t *)evt)[(2 * cs) * RB + r], e = ((const uint16_t *)evt)[(2 * cs + 1) * RB + r];
                    // (a source column's tags are there in every record that follows the template;
                    // the test on the positions only keeps a length in bounds whatever happens)
                    const bool have = ((pl.x >> 11) & 1u) & act & (o != NONE16) & (e != NONE16);
                    const uint32_t s1 = o + nml2;
                    int l = (int)e - (int)s1;
                    l = l < 0 ? (int)(re - s1) : (cl > 0 && cl < l) ? cl : l;
                    const uint32_t s = have ? s1 : 0u, len = have ? (uint32_t)l : 0u;
                    uint32_t D16, c0, wp, N1, pp;
                    const uint32_t Lm = (1u << (len < 16 ? len : 16)) - 1;
                    if (__builtin_amdgcn_ballot_w64(len > 8) == 0) {  // (most numbers: half the window)
                        uint32_t w0, w1;
                        lds_bytes8(sdata + s, w0, w1);
                        D16 = __builtin_amdgcn_udot4(digit_flags(w1), 0x80402010u,
                                                     __builtin_amdgcn_udot4(digit_flags(w0), 0x08040201u, 0u, false), false) >> 7;
                        c0 = w0 & 0xFFu;
                        N1 = ~D16 & Lm & ~(uint32_t)((c0 == '+') | (c0 == '-'));
                        pp = (uint32_t)__builtin_ctz(N1 | 0x100u) & 7u;
                        wp = pp < 4 ? w0 : w1;
                    } else {
                        const uint4 w4 = lds_bytes16(sdata + s);
                        D16 = digit_mask16(w4);
                        c0 = w4.x & 0xFFu;
                        N1 = ~D16 & Lm & ~(uint32_t)((c0 == '+') | (c0 == '-'));
                        pp = (uint32_t)__builtin_ctz(N1 | 0x10000u) & 15u;
                        wp = pp < 4 ? w4.x : pp < 8 ? w4.y : pp < 12 ? w4.z : w4.w;
                    }
                    const bool numplain = ((N1 & (N1 - 1)) == 0) & ((N1 == 0) | (((wp >> (8 * (pp & 3))) & 0xFFu) == '.'));
                    const bool ok = (len <= 16) & ((lvl != LVL_NUMBER) | numplain);
                    if (__builtin_amdgcn_ballot_w64(!ok) == 0) {
                        measured = true;
                        flen = lvl == LVL_DATE ? (uint32_t)__builtin_popcount(D16 & Lm) : len;
                        if (mapping && act) P.cellmap[(uint64_t)c * P.map_cap + slot + rb0 + r] = map_pack(s, len, lvl, (pl.x >> 10) & 1u);
                    }
                }
                if (act && !measured) {
                    uint32_t s, len, lvl;
                    bool found;
                    if (tmode && !irregular) {
                        // every record of the tile has the sample's tags: which column a cell comes
                        // from, its level and the clip are the plan's, only the positions are the record's
                        const uint2 pl = plan_l[c];
                        const uint32_t cs = pl.x & 0xFFu, nml2 = pl.x >> 16;
                        const int cl = (int)pl.y;
                        lvl = (pl.x >> 8) & 3u;
                        found = (pl.x >> 10) & 1u;
                        const uint32_t o = ((const uint16_t *)evt)[(2 * cs) * RB + r], e = ((const uint16_t *)evt)[(2 * cs + 1) * RB + r];
                        const uint32_t re = recend[r + 1];
                        s = o + nml2;
                        int l = (int)e - (int)s;
                        l = l < 0 ? (int)(re - s) : (cl > 0 && cl < l) ? cl : l;
                        const bool have = ((pl.x >> 11) & 1u) & (o != NONE16) & (e != NONE16);
                        s = have ? s : 0u;
                        len = have ? (uint32_t)l : 0u;
                    } else {
                        fetch_cell(c, r, s, len, lvl, found);
                    }
                    if (mapping) {
                        if (len < MAP_MAXLEN)
                            P.cellmap[(uint64_t)c * P.map_cap + slot + rb0 + r] = map_pack(s, len, lvl, found);
                        else
                            ms[MS_NOMAP] = 1;
                    }
                    if (len <= 16) {
                        const bool cutting = lvl == LVL_CHAR && nulcut && has_nul;
                        if (lvl == LVL_CHAR && !cutting) {
                            flen = len;  // (a text cell is not looked at)
                        } else {
                            const uint4 w4 = lds_bytes16(sdata + s);
                            const uint64_t lo = (uint64_t)w4.x | ((uint64_t)w4.y << 32);
                            const uint64_t hi = (uint64_t)w4.z | ((uint64_t)w4.w << 32);
                            uint64_t value, elo, ehi;
                            bool valid, later = false, fast = false;
                            // the length alone: the date level keeps the digits; a number that has
                            // only a sign in front and one point besides its digits keeps everything
                            const uint32_t D16 = digit_mask16(w4);
                            const uint32_t Lm = (1u << len) - 1;
                            const uint32_t N = ~D16 & Lm;
                            const uint32_t c0 = w4.x & 0xFFu;
                            const uint32_t N1 = N & ~(uint32_t)((c0 == '+') | (c0 == '-'));
                            const uint32_t pp = (uint32_t)__builtin_ctz(N1 | 0x10000u);
                            if (lvl == LVL_DATE) {
                                flen = (uint32_t)__builtin_popcount(D16 & Lm);
                                fast = true;
                            } else if (lvl == LVL_NUMBER && (N1 & (N1 - 1)) == 0 && (N1 == 0 || win_byte(lo, hi, pp & 15) == '.')) {
                                flen = len;
                                fast = true;
                            } else if (!cutting) {
                                fast = cell_fast(lo, hi, len, D16, lvl, T_NONE, p10, flen, value, valid, elo, ehi, later);
                            }
                            if (!fast) flen = cell_scan16<false>(lo, hi, len, lvl, nulcut, T_NONE, p10, nullptr, value, valid, later);
                        }
                    } else {
                        flen = cell_filter_long<false>(sdata + s, len, lvl, nulcut, nullptr);
                    }
                }
                const uint32_t incl = wave_incl_scan(flen);
                // this column's run in the wave: lanes [first, last]
                const uint32_t cstart = c * RBn, cend = cstart + RBn - 1;
                const uint32_t first = cstart > i0 ? cstart - i0 : 0;
                const uint32_t last = cend - i0 < 63 ? cend - i0 : 63;
                const uint32_t before = __shfl(incl, first ? first - 1 : 0, 64);
                if (act && (uint32_t)lane == last) atomicAdd(&tiletot[c], incl - (first ? before : 0));
            }
        } else
        for (uint32_t c = wave; c < ncols && !SKIP(2); c += BLOCK / 64) {
            const uint32_t typed = (uint32_t)__builtin_amdgcn_readfirstlane((int)cols[c].typed);
            uint32_t cs_[RB_PASSES], cl_[RB_PASSES], cv_[RB_PASSES];
            bool fd_[RB_PASSES];
#pragma unroll
            for (int ps = 0; ps < RB_PASSES; ++ps) {
                const uint32_t r = ps * 64 + lane;
                uint32_t s = 0, len = 0, lvl = LVL_CHAR;
                bool found = false;
                if (r < RBn) {
                    if (tmode && !irregular) {
                        // every record of the tile has the sample's tags: which column a cell comes
                        // from, its level and the clip are the plan's, only the positions are the record's
                        const uint2 pl = plan_l[c];
                        const uint32_t cs = pl.x & 0xFFu, nml2 = pl.x >> 16;
                        const int cl = (int)pl.y;
                        lvl = (pl.x >> 8) & 3u;
                        found = (pl.x >> 10) & 1u;
                        const uint32_t o = ((const uint16_t *)evt)[(2 * cs) * RB + r], e = ((const uint16_t *)evt)[(2 * cs + 1) * RB + r];
                        const uint32_t re = recend[r + 1];
                        s = o + nml2;
                        int l = (int)e - (int)s;
                        l = l < 0 ? (int)(re - s) : (cl > 0 && cl < l) ? cl : l;
                        const bool have = ((pl.x >> 11) & 1u) & (o != NONE16) & (e != NONE16);
                        s = have ? s : 0u;
                        len = have ? (uint32_t)l : 0u;
                    } else {
                        fetch_cell(c, r, s, len, lvl, found);
                    }
                    if (found && !csv) atomicOr(&fmask[r * W + (c >> 5)], 1u << (c & 31));
                }
                cs_[ps] = s;
                cl_[ps] = len;
                cv_[ps] = lvl;
                fd_[ps] = found;
            }
            const ColOut O = colout[c];
            const uint64_t gb = O.base + tiletot[c];
            const uint32_t done = (uint32_t)(gb - O.base);
            const uint32_t total = column_cells<WRITE, true>(E, O, c, typed, lane, RBn, row0 + rb0, gb, cs_, cl_, cv_, fd_, deferred,
                                                             coltot[c] - done, overrun);
            if (lane == 0) tiletot[c] = done + total;  // this wave owns column c
        }
        if (WRITE && deferred) raise_status(P, ST_NUMERIC_RANGE);
        if (WRITE && overrun) raise_status(P, ST_INTERNAL);
        __syncthreads();
        STAMP(5);  // cells: descriptors, lengths, values, staging, copy-out
        if (mapping && (uint32_t)tid < RBn)
            P.cellmap[(uint64_t)ncols * P.map_cap + slot + rb0 + tid] = recend[tid + 1];
        if (WRITE && (uint32_t)tid < RBn) {
            const uint64_t row = row0 + rb0 + tid;
            if (P.rec_off) P.rec_off[row + 1] = (uint64_t)(base_pos + (int64_t)recend[tid + 1] + 1);
#pragma unroll 1  // (one word for up to 32 columns: an unrolled loop only costs scalar registers)
            for (uint32_t w = 0; w < W; ++w) {
                P.valid[row * W + w] = vmask[tid * W + w];
                P.found[row * W + w] = fmask[tid * W + w];
            }
        }
        if (rb0 + RB >= nrec) break;  // (the totals below are read behind the barrier that follows the cells)
        // (entry 0 is read by nobody in this phase: the row outputs read entries 1 to RBn)
        if (tid == 0) {
            recend[0] = recend[RBn];
            if (tmode) e0t[0] = e0t[RBn];
        }
        __syncthreads();
        STAMP(9);  // row outputs
    }

    if (WRITE) {
        // (the batch loop ends with a barrier) every column must have come out as long as it was sized
        for (uint32_t c = tid; c < ncols; c += BLOCK)
            if (tiletot[c] != coltot[c]) raise_status(P, ST_INTERNAL);
    }
    bool odd = false;  // one-pass parse: (block-uniform) the tile is left to the full write kernel
    if (FUSED) {
        // (tiles that went through no tag search: tiles without records, delimited input)
        rows_lookback();
        __syncthreads();  // the tile's counts, its first row and its staging pieces are final
        odd = !fz;
        const uint32_t tile = blockIdx.x;
        // what this tile counted, for the tiles behind ...
        for (uint32_t c = tid; c < ncols; c += BLOCK) pub_store(P.pub + (uint64_t)c * P.ntiles + tile, PUB_AGG | tiletot[c]);
        // ... then, one wave per column, what the tiles in front counted: the column's place in its arena
        const uint64_t frow0 = (uint64_t)ms[MS_ROW0] | ((uint64_t)ms[MS_ROW0 + 1] << 32);
        const bool rowok = ms[MS_ROWOK] != 0;
        const uint16_t *endl = (const uint16_t *)(smem + L.endl);
        uint64_t *tbw = const_cast<uint64_t *>(P.tile_base) + (uint64_t)tile * (ncols + 1);
        for (uint32_t c = wave; c < ncols; c += BLOCK / 64) {
            bool gu = false;
            uint64_t *pc = P.pub + (uint64_t)c * P.ntiles;
            const uint64_t base = SKIP(1024) ? (uint64_t)tile * (tiletot[c] - tiletot[c] / 8) : fused_lookback(pc, tile, lane, P.status, P.spin_limit, gu);
            const uint32_t tot = tiletot[c];
            const bool fits = base + tot <= P.caps[c];
            if (lane == 0) {
                pub_store(pc + tile, PUB_INCL | (base + tot));
                if (gu) raise_status(P, ST_GIVEUP);
                if (!fits) raise_status(P, ST_OVERFLOW);
                if (tile == P.ntiles - 1) P.fused_totals[c] = base + tot;
                if (odd) tbw[c] = base;
            }
            if (!odd && fits && rowok && !gu) {
                fused_copy_out(stage + coltot[c] + 16, tot, (gmem_u8 *)O.bytes[c] + base, lane);
                gmem_u32 *offp = (gmem_u32 *)O.off[c] + frow0 + 1;
#pragma unroll
                for (int ps = 0; ps < RB_PASSES; ++ps) {
                    const uint32_t r = ps * 64 + lane;
                    if (r < nrec) offp[r] = (uint32_t)(base + endl[c * RB + r]);
                }
            }
        }
        if (tid == 0 && odd) {
            tbw[ncols] = frow0;
            atomicAdd(P.status + 3, 1u);
        }
    }
    if (!WRITE) {
        if (tmode && irregular && tid == 0) atomicAdd(P.status + 1, 1u);  // the host renews a template few tiles follow
        if (!FUSED || odd) {
            uint32_t *tc = P.tile_counts + (uint64_t)blockIdx.x * (ncols + 1);
            for (uint32_t c = tid; c < ncols; c += BLOCK) tc[c] = tiletot[c];
            if (tid == 0) tc[ncols] = nrec_all;
        }
        if (tid == 0 && P.tile_hdr) {
            // all batches are done (a barrier follows the last batch's cells): is the tile's map complete?
            const uint32_t slot = map_slot;
            const bool mapped = FUSED ? !odd : (slot != NONE && !ms[MS_NOMAP]);
            P.tile_hdr[2 * blockIdx.x] = mapped ? (FUSED ? 0u : slot) : longrec ? HDR_LONGREC : NONE;
            P.tile_hdr[2 * blockIdx.x + 1] = longrec ? long_entry : (uint32_t)s0_li | (has_nul ? 0x80000000u : 0u);
            if (!mapped) raise_status(P, ST_UNMAPPED);
        }
    }
}

// Write pass over tiles whose cell map the sizing pass completed: no record split, no tag
// search.  A block loads its tile (and the head of its first record from the look-behind),
// rebuilds the digit bitmap, reads each cell's source from the map and runs the same column
// routine as the tile kernel.  Tiles without a map are left to the full write pass of the tile kernel.
// MANY: the instance for inputs whose tiles take several batches (batches of a multiple of 64 records):
// later batches' map words come straight into LDS, see below.
template <bool SPECIAL, bool MANY>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(5, 5))) void emit_kernel(const TileParams P, const OutPtrs O) {
    extern __shared__ __align__(16) uint8_t smem[];
    const LdsLayout &L = P.L;
    uint8_t *sdata = smem + L.data;
    uint32_t *vmask = (uint32_t *)(smem + L.vmask);
    uint32_t *fmask = (uint32_t *)(smem + L.fmask);
    uint8_t *stage = smem + L.evlist;
    uint32_t *mapl = (uint32_t *)(smem + L.evt);  // the batch's slice of the cell map: word (c, r) at c * RB + r
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (uniform: scalar loads for per-column data)
    const uint32_t tile = blockIdx.x;
    const uint32_t ncols = P.ncols, W = P.words;
    if (tile == 0) {
        if (tid == 0 && P.rec_off) P.rec_off[0] = 0;
        for (uint32_t c = tid; c < ncols; c += BLOCK) O.off[c][0] = 0;
    }
    // Everything the block needs is requested before anything is waited for: the tile's header
    // words, the tile itself, the first look-behind window and the tile's slice of the cell map
    // (whose place, tile * map_stride, does not depend on the header).
    const uint32_t slot = tile * P.map_stride;
    // The tile's header words are the same for every lane, and the compiler would fetch them into
    // scalar registers at once, which is a wait in front of all the other loads: a zero it
    // cannot see through keeps them per-lane values until they are needed.
    uint32_t vz;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vz));
    const uint32_t hslot_v = P.tile_hdr[2 * tile + vz];
    const uint32_t hw_v = P.tile_hdr[2 * tile + 1 + vz];
    const uint32_t nrec_v = P.tile_counts[(uint64_t)tile * (ncols + 1) + ncols + vz];
    const uint64_t row0_v = P.tile_base[(uint64_t)tile * (ncols + 1) + ncols + vz];

    const int64_t n = (int64_t)P.n;
    const int64_t tile_start = (int64_t)tile * TILE;
    const int64_t tile_end = (tile_start + TILE < n) ? tile_start + TILE : n;
    const int LB = (int)L.lbmax;
    const int64_t base_pos = tile_start - LB;
    const int ts_li = LB;
    const int te_li = LB + (int)(tile_end - tile_start);

    auto load_chunk = [&](int j) -> uint4 {
        const int64_t pos = base_pos + (int64_t)j * CHUNK;
        if (pos + CHUNK <= n) return *(const uint4 *)(P.data + pos);
        uint32_t w[4] = {0, 0, 0, 0};
        for (int k = 0; k < CHUNK; ++k)
            if (pos + k < n) w[k >> 2] |= (uint32_t)P.data[pos + k] << (8 * (k & 3));
        return make_uint4(w[0], w[1], w[2], w[3]);
    };
    // (no load may stand behind a branch: the compiler waits at the end of a branch for what was
    // loaded in it.  A tile that lies whole in the input takes plain loads; a lane without a chunk,
    // map word or table entry of its own loads one that it then drops)
    uint4 tv[4];
    const int jw = (ts_li - LB_FIRST) / CHUNK + tid;  // this thread's chunk of the first look-behind window
    const bool lbmine = tid < LB_FIRST / CHUNK && base_pos + (int64_t)jw * CHUNK >= 0;
    uint4 lbv;
    if (tile_start + TILE <= n) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            tv[k] = *(const uint4 *)(P.data + (base_pos + (int64_t)(ts_li / CHUNK + k * BLOCK + tid) * CHUNK));
        lbv = *(const uint4 *)(P.data + (base_pos + (int64_t)(lbmine ? jw : ts_li / CHUNK) * CHUNK));
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = ts_li / CHUNK + k * BLOCK + tid;
            tv[k] = j * CHUNK < te_li ? load_chunk(j) : make_uint4(0, 0, 0, 0);
        }
        lbv = lbmine ? load_chunk(jw) : make_uint4(0, 0, 0, 0);
    }
    const uint32_t RB = L.RB;
    const uint32_t RBs = P.map_stride < RB ? P.map_stride : RB;  // records of the first batch, at most
    // The first batch's map words: every wave fetches the rows it reads itself (its columns' rows, two
    // words a lane, and its 64 record ends), which needs no division to find a word's place.  Three
    // columns a wave are fetched here, the rows of wider schemas behind the first wait.
    constexpr int MAPCOLS = 3;
    uint32_t mv[2 * MAPCOLS + 1];
    const uint32_t r_lo = (uint32_t)lane < RBs ? lane : RBs - 1, r_hi = (uint32_t)lane + 64 < RBs ? lane + 64 : RBs - 1;
    const uint32_t *const mbase = P.cellmap + slot;
#pragma unroll
    for (int j = 0; j < MAPCOLS; ++j) {
        const uint32_t c0 = (uint32_t)wave + (uint32_t)j * (BLOCK / 64);
        const uint32_t c = c0 < ncols ? c0 : 0;
        mv[2 * j] = mbase[(uint64_t)c * P.map_cap + r_lo];
        mv[2 * j + 1] = mbase[(uint64_t)c * P.map_cap + r_hi];
    }
    {
        const uint32_t r = (uint32_t)tid < RBs ? tid : RBs - 1;  // (RBs <= 128: waves 0 and 1 hold the record ends)
        mv[2 * MAPCOLS] = mbase[(uint64_t)ncols * P.map_cap + r];
    }
    // of the schema only the header, the columns and the powers of ten (the tag table stays out)
    const uint32_t sch_head = (uint32_t)sizeof(SchemaHdr) + ((ncols * (uint32_t)sizeof(DevCol) + 15) & ~15u);
    uint32_t *tiletot = (uint32_t *)(smem + L.coltot);
    uint32_t *sized = tiletot + (ncols + 1);  // the sizing pass's bytes per column of this tile
    ColOut *colout = (ColOut *)(smem + L.colout);
    const uint64_t *tb = P.tile_base + (uint64_t)tile * (ncols + 1);
    if (wave == 0) {
        // One wave fetches the block's tables, as the last loads before the wait that all loads share.
        // (A lane beyond a table's end copies entry 0 once more: the same bytes to the same place, and no
        // branch around a load.)
        const uint4 *g = (const uint4 *)P.schema;
        uint4 *s = (uint4 *)(smem + L.schema);
        const uint64_t *gp = (const uint64_t *)(P.schema + P.p10_off);
        uint64_t *sp = (uint64_t *)(smem + L.schema + sch_head);
        const uint32_t sch16 = sch_head / 16;
        const uint32_t si = (uint32_t)lane < sch16 ? lane : 0;
        const uint32_t pi = lane < P10_TABLE_DOUBLES ? lane : 0;
        const uint32_t cc = (uint32_t)lane < ncols ? lane : 0;
        const uint4 sv = g[si];
        const uint64_t pv = gp[pi];
        ColOut o;
        o.off = O.off[cc];
        o.bytes = O.bytes[cc];
        o.val = O.val[cc];
        o.base = tb[cc];
        const uint32_t szv = P.tile_counts[(uint64_t)tile * (ncols + 1) + cc];
        s[si] = sv;
        for (uint32_t i = lane + 64; i < sch16; i += 64) s[i] = g[i];
        sp[pi] = pv;
        tiletot[cc] = 0;
        sized[cc] = szv;
        colout[cc] = o;
        for (uint32_t c = lane + 64; c < ncols; c += 64) {  // schemas with more than 64 columns
            ColOut o2;
            o2.off = O.off[c];
            o2.bytes = O.bytes[c];
            o2.val = O.val[c];
            o2.base = tb[c];
            tiletot[c] = 0;
            sized[c] = P.tile_counts[(uint64_t)tile * (ncols + 1) + c];
            colout[c] = o2;
        }
    }
    const uint32_t hslot = __builtin_amdgcn_readfirstlane(hslot_v);
    const uint32_t hw = __builtin_amdgcn_readfirstlane(hw_v);
    const uint32_t nrec = __builtin_amdgcn_readfirstlane(nrec_v);
    const uint64_t row0 = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(row0_v >> 32)) << 32) |
                          __builtin_amdgcn_readfirstlane((uint32_t)row0_v);

    if (hslot >= HDR_LONGREC || nrec == 0) return;  // no map (the full write kernel has the tile), or nothing to write
    const int s0_li = (int)(hw & 0x7FFFFFFFu);
    const bool has_nul = (hw >> 31) != 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) ((uint4 *)sdata)[ts_li / CHUNK + k * BLOCK + tid] = tv[k];
    if (lbmine) ((uint4 *)sdata)[jw] = lbv;
    // (rare) the first record starts in front of the first look-behind window
    for (int j = s0_li / CHUNK + tid; j < (ts_li - LB_FIRST) / CHUNK; j += BLOCK) ((uint4 *)sdata)[j] = load_chunk(j);
#pragma unroll
    for (int j = 0; j < MAPCOLS; ++j) {
        const uint32_t c = (uint32_t)wave + (uint32_t)j * (BLOCK / 64);
        if (c < ncols) {
            if ((uint32_t)lane < RBs) mapl[c * RB + lane] = mv[2 * j];
            if ((uint32_t)lane + 64 < RBs) mapl[c * RB + lane + 64] = mv[2 * j + 1];
        }
    }
    if ((uint32_t)tid < RBs) mapl[ncols * RB + tid] = mv[2 * MAPCOLS];
    for (uint32_t c = (uint32_t)wave + MAPCOLS * (BLOCK / 64); c < ncols; c += BLOCK / 64)  // schemas with many columns
        for (uint32_t r = lane; r < RBs; r += 64) mapl[c * RB + r] = mbase[(uint64_t)c * P.map_cap + r];
    // the map words of a later batch
    auto load_map = [&](uint32_t rb0, uint32_t RBn) {
        for (uint32_t c = wave; c <= ncols; c += BLOCK / 64) {
            const uint32_t *mp = P.cellmap + (uint64_t)c * P.map_cap + slot + rb0;
            for (uint32_t r = lane; r < RBn; r += 64) mapl[c * RB + r] = mp[r];
        }
    };
    // With rows of a multiple of 64 slots a later batch's map words come straight into LDS (no
    // register is held while they are on their way): every wave fetches the rows it reads itself, its
    // columns' when it begins its last column of the batch and its 64 record ends behind the row
    // outputs.  The words are wave-private, so no barrier stands between their arrival and their use,
    // only the wave's own wait.  (A lane beyond the next batch's records fetches the last record's word.)
    constexpr bool dma = MANY;
    auto dma_row = [&](uint32_t c, uint32_t h, uint32_t nrb0, uint32_t nRBn) {
        const uint32_t r = h + lane;
        const uint32_t *src = P.cellmap + (uint64_t)c * P.map_cap + slot + nrb0 + (r < nRBn ? r : nRBn - 1);
        // (written out: the compiler, told of such a load, waits for every outstanding store in front of each
        // later LDS read; the wait that counts is the one at the head of the next batch)
        const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(mapl + c * RB + h));
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" : : "v"(src), "s"(dst) : "memory", "m0");
    };
    auto clear_masks = [&]() {
        for (uint32_t i = tid; i < RB * W; i += BLOCK) {
            vmask[i] = 0;
            fmask[i] = 0;
        }
    };
    clear_masks();
    __syncthreads();
    const SchemaHdr *H = (const SchemaHdr *)(smem + L.schema);
    const DevCol *cols = (const DevCol *)(H + 1);
    CellEnv E;
    E.sdata = sdata;
    E.has_nul = has_nul;
    E.nulcut = H->carry != 0 && H->mode != MODE_CSV;
    E.p10 = (const double *)(smem + L.schema + sch_head);
    E.tz_offset = H->tz_offset;
    E.W = W;
    E.vmask = vmask;
    E.wstage = stage + wave * (L.stage_cap / (BLOCK / 64));
    E.wstage_cap = L.stage_cap / (BLOCK / 64);
    E.plain_ok = !(E.nulcut && has_nul);
    E.nostore = SKIP(32);
#ifdef MDC_ABLATE
    E.dbg = P.debug;
#endif
    bool deferred = false, overrun = false;
    for (uint32_t rb0 = 0; rb0 < nrec; rb0 += RB) {
        const uint32_t RBn = (nrec - rb0 < RB) ? nrec - rb0 : RB;
        const bool more = rb0 + RB < nrec;
        const uint32_t nRBn = more ? (nrec - rb0 - RB < RB ? nrec - rb0 - RB : RB) : 0;  // records of the next batch
        if (rb0) {  // (the previous batch ended with a barrier; the first batch was set up above)
            if (dma) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's rows have arrived
            } else {
                load_map(rb0, RBn);
                clear_masks();
                __syncthreads();
            }
        }
        for (uint32_t c = wave; c < ncols && !SKIP(16); c += BLOCK / 64) {
            const uint32_t *mp = mapl + c * RB;
            uint32_t cs_[RB_PASSES], cl_[RB_PASSES], cv_[RB_PASSES];
            bool fd_[RB_PASSES];
#pragma unroll
            for (int ps = 0; ps < RB_PASSES; ++ps) {
                const uint32_t r = ps * 64 + lane;
                const uint32_t d = r < RBn ? mp[r] : 0u;
                cs_[ps] = d & 0x7FFFu;
                cl_[ps] = (d >> 15) & (MAP_MAXLEN - 1);
                cv_[ps] = (d >> 29) & 3u;
                fd_[ps] = (d >> 31) != 0;
                if (fd_[ps]) atomicOr(&fmask[r * W + (c >> 5)], 1u << (c & 31));
            }
            if (dma && more && c + BLOCK / 64 >= ncols)  // the wave's last column: its rows of this batch are read
                for (uint32_t cc = wave; cc < ncols; cc += BLOCK / 64)
                    for (uint32_t h = 0; h < RB; h += 64) dma_row(cc, h, rb0 + RB, nRBn);
            const ColOut O = colout[c];
            const uint32_t done = tiletot[c];
            const uint64_t gb = O.base + done;
            const uint32_t total = column_cells<true, SPECIAL>(E, O, c, cols[c].typed, lane, RBn, row0 + rb0, gb, cs_, cl_, cv_,
                                                               fd_, deferred, sized[c] - done, overrun);
            if (lane == 0) tiletot[c] = done + total;  // this wave owns column c
        }
        __syncthreads();
        if ((uint32_t)tid < RBn) {
            const uint64_t row = row0 + rb0 + tid;
            if (P.rec_off)
                P.rec_off[row + 1] =
                    (uint64_t)(base_pos + (int64_t)mapl[ncols * RB + tid] + 1);
#pragma unroll 1  // (one word for up to 32 columns: an unrolled loop only costs scalar registers)
            for (uint32_t w = 0; w < W; ++w) {
                P.valid[row * W + w] = vmask[tid * W + w];
                P.found[row * W + w] = fmask[tid * W + w];
                if (dma) {  // (for the next batch)
                    vmask[tid * W + w] = 0;
                    fmask[tid * W + w] = 0;
                }
            }
        }
        if (more) {
            if (dma && (uint32_t)wave * 64 < nRBn) dma_row(ncols, wave * 64, rb0 + RB, nRBn);  // the record ends this wave reads
            __syncthreads();  // the next batch reuses the bitmaps (and, fetched by all waves, the map)
        }
    }
    if (deferred) raise_status(P, ST_NUMERIC_RANGE);
    // (behind the barrier that follows the last batch's columns) every column must have come out
    // as long as it was sized
    bool differ = overrun;
    for (uint32_t c = tid; c < ncols; c += BLOCK) differ |= tiletot[c] != sized[c];
    if (differ && !SKIP(16)) raise_status(P, ST_INTERNAL);
}

// ---------------------------------------------------------------------------------------
// Records longer than the look-behind capacity of the tile that ends them.  cifile::readline
// (reference public/_public.cpp:866-892) puts no bound on a record, so neither does the parser:
// the sizing pass lists such records (they are always the first record of their tile) and this
// kernel does what the tile kernel does for a record, with one block per record and every byte
// read straight from global memory: record start, getxmlbuffer's two first-occurrence searches per
// column (or splittocmd's greedy separator search), clip, carry-forward, filter, NUL cut.
// WRITE = false adds the cells' lengths to the owning tile's counters (and keeps them for the
// write pass); WRITE = true stores the record's row.  Typed values are left to the second stage
// (deferred_convert_kernel), as for every cell of more than 16 source bytes.  Speed does not
// matter here; a record of 2 GiB or more is refused (the reference's own positions are ints).
__device__ __forceinline__ uint32_t gbyte(const uint8_t *data, int64_t n, int64_t pos) {
    return pos >= 0 && pos < n ? data[pos] : 0u;
}
__device__ __forceinline__ uint4 gchunk(const uint8_t *data, int64_t n, int64_t pos) {  // pos is a multiple of 16
    if (pos >= 0 && pos + 16 <= n) return *(const uint4 *)(data + pos);
    uint32_t w[4] = {0, 0, 0, 0};
    for (int k = 0; k < 16; ++k) w[k >> 2] |= gbyte(data, n, pos + k) << (8 * (k & 3));
    return make_uint4(w[0], w[1], w[2], w[3]);
}
__device__ __forceinline__ bool gterm(const uint8_t *data, const SchemaHdr *H, int64_t p) {  // is the '\n' at p a terminator?
    const int el = (int)H->endbz_len;
    if (p - el < 0) return el == 0;
    const uint8_t *mark = endbz_text(H);
    for (int k = 0; k < el; ++k)
        if (data[p - el + k] != mark[k]) return false;
    return true;
}
// does byte k of a cell pass the filter of its level?
__device__ __forceinline__ bool level_pass(uint32_t b, uint32_t lvl) {
    const bool dig = (b - '0') < 10u;
    return lvl == LVL_CHAR ? true : lvl == LVL_NUMBER ? (dig | (b == '+') | (b == '-') | (b == '.')) : dig;
}

template <bool WRITE>
__global__ __launch_bounds__(BLOCK) void long_record_kernel(const TileParams P, const OutPtrs O) {
    extern __shared__ __align__(16) uint8_t smem[];
    __shared__ uint32_t sh[8];
    __shared__ uint32_t wsum[BLOCK / 64];
    __shared__ uint16_t cand[BLOCK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const LongRec R = P.long_list[blockIdx.x];
    const uint8_t *data = P.data;
    const int64_t n = (int64_t)P.n;
    for (uint32_t i = tid; i < P.schema_bytes / 16; i += BLOCK) ((uint4 *)smem)[i] = ((const uint4 *)P.schema)[i];
    if (tid < 8) sh[tid] = 0;
    __syncthreads();
    const SchemaHdr *H = (const SchemaHdr *)smem;
    const DevCol *cols = (const DevCol *)(H + 1);
    const uint32_t ncols = H->ncols, W = H->words;
    const HtEntry *ht = (const HtEntry *)((const uint8_t *)cols + ((ncols * sizeof(DevCol) + 15) & ~15u));
    const bool csv = H->mode == MODE_CSV;
    const bool nulcut = H->carry != 0 && !csv;
    uint32_t *work = (uint32_t *)(smem + ((P.schema_bytes + 15) & ~15u));
    uint32_t *first = work;                 // XML: first position of every pattern; delimited: the separators
    uint32_t *dstart = work + 2 * ncols;    // the cells' sources ...
    uint32_t *dlen = dstart + ncols;
    uint32_t *dlvl = dlen + ncols;          // ... level | found << 8
    const int64_t gend = (int64_t)R.gend;

    // 1. the record's start: behind the last terminator in front of `from` (multiples of 16 down to 0)
    int64_t gstart = 0;
    for (int64_t hi = (int64_t)R.from; hi > 0;) {
        const int64_t lo = hi - BLOCK * CHUNK > 0 ? hi - BLOCK * CHUNK : 0;
        const int64_t pos = lo + (int64_t)tid * CHUNK;
        if (pos < hi) {
            uint32_t nl = mask16_eq(gchunk(data, n, pos), 0x0A0A0A0Au);
            while (nl) {
                const int b = 31 - __builtin_clz(nl);  // the last one first
                nl &= ~(1u << b);
                if (gterm(data, H, pos + b)) {
                    atomicMax(&sh[0], (uint32_t)(pos + b - lo) + 1);
                    break;
                }
            }
        }
        __syncthreads();
        const uint32_t hit = sh[0];
        __syncthreads();
        if (hit) {
            gstart = lo + hit;  // (hit - 1 is the terminator's place in the window)
            break;
        }
        hi = lo;
    }
    if (gend - gstart >= ((int64_t)1 << 31)) {
        if (tid == 0) raise_status(P, ST_RECORD_TOO_LONG);
        return;
    }
    const uint32_t reclen = (uint32_t)(gend - gstart);  // the record's text: the '\n' is not part of it

    // 2. where the fields are
    const uint32_t seplen = H->sep_len;
    for (uint32_t i = tid; i < 2 * ncols; i += BLOCK) first[i] = NONE;
    __syncthreads();
    if (!csv) {
        for (int64_t cb = gstart & ~(int64_t)15; cb < gend; cb += BLOCK * CHUNK) {
            const int64_t pos = cb + (int64_t)tid * CHUNK;
            uint32_t m = pos < gend ? mask16_eq(gchunk(data, n, pos), 0x3C3C3C3Cu) : 0u;
            while (m) {
                const int b = __builtin_ctz(m);
                m &= m - 1;
                const int64_t q = pos + b;
                if (q < gstart || q >= gend) continue;
                // the key: up to eight bytes behind '<', through the first '>' (as tag_key)
                uint32_t kw[2] = {0, 0};
                for (int k = 0; k < 8; ++k) {
                    const uint32_t c = gbyte(data, n, q + 1 + k);
                    kw[k >> 2] |= c << (8 * (k & 3));
                    if (c == '>') break;
                }
                TagKey key;
                key.lo = kw[0];
                key.hi = kw[1];
                uint32_t slot = tag_slot(H, key);
                if (!(ht[slot].meta >> 31)) continue;
                for (;;) {
                    const HtEntry &e = ht[slot];
                    const uint32_t tlen = e.meta & 0xFFu, kind = (e.meta >> 8) & 1u, col = (e.meta >> 16) & 0x3FFFu;
                    bool ok = e.klo == key.lo && e.khi == key.hi;
                    if (ok && tlen >= 8) {  // the key covers eight bytes of the text; the rest, then '>'
                        for (uint32_t k = 8; ok && k < tlen; ++k) ok = gbyte(data, n, q + 1 + k) == cols[col].name[k - kind];
                        ok = ok && gbyte(data, n, q + 1 + tlen) == '>';
                    }
                    if (ok) atomicMin(&first[2 * col + kind], (uint32_t)(q - gstart));
                    slot = e.next;
                    if (!slot) break;
                }
            }
            // Once every pattern has been seen the rest of the record cannot lower a first position:
            // a record with all its tags near its head is not walked to its end.
            __syncthreads();
            bool missing = false;
            for (uint32_t i = tid; i < 2 * ncols; i += BLOCK) missing |= first[i] == NONE;
            if (!__syncthreads_or(missing)) break;
        }
    } else {
        // the first ncols separators, left to right without overlap (ccmdstr::splittocmd): all lanes
        // mark where the separator's text stands, one lane walks the marks in order
        uint32_t nsep = 0, allowed = 0;  // (thread 0's) separators found, first position a match may start at
        for (int64_t cb = gstart & ~(int64_t)15; cb < gend; cb += BLOCK * CHUNK) {
            const int64_t pos = cb + (int64_t)tid * CHUNK;
            uint32_t m = 0;
            if (pos < gend) {
                m = mask16_eq(gchunk(data, n, pos), rep4(H->sep[0]));
                uint32_t keep = 0;
                while (m) {
                    const int b = __builtin_ctz(m);
                    m &= m - 1;
                    const int64_t q = pos + b;
                    bool ok = q >= gstart && q + seplen <= gend;
                    for (uint32_t k = 1; ok && k < seplen; ++k) ok = data[q + k] == sep_text(H)[k];
                    if (ok) keep |= 1u << b;
                }
                m = keep;
            }
            cand[tid] = (uint16_t)m;
            // (a window without a mark, as every window of one very long field is, costs no walk)
            if (!__syncthreads_or(m != 0)) continue;
            if (tid == 0) {
                for (int t = 0; t < BLOCK && nsep < ncols; ++t) {
                    uint32_t x = cand[t];
                    while (x && nsep < ncols) {
                        const int b = __builtin_ctz(x);
                        x &= x - 1;
                        const uint32_t rel = (uint32_t)(cb + t * CHUNK + b - gstart);
                        if (rel >= allowed) {
                            first[nsep++] = rel;
                            allowed = rel + seplen;
                        }
                    }
                }
                sh[1] = nsep;
            }
            __syncthreads();
            if (sh[1] >= ncols) break;
        }
    }
    __syncthreads();

    // 3. the cells' sources (as fetch_cell of the tile kernel, with positions relative to the record)
    for (uint32_t c = tid; c < ncols; c += BLOCK) {
        uint32_t s = 0, len = 0, lvl = LVL_CHAR;
        bool found = false;
        if (csv) {
            const uint32_t nsep = sh[1];
            if (c <= nsep) {
                found = true;
                s = c ? first[c - 1] + seplen : 0;
                len = (c < nsep ? first[c] : reclen) - s;
                if (H->bdelspace) {  // deletelchr, then deleterchr with its "pos != 0" test (public/_public.cpp:24-31, :52-59)
                    while (len && data[gstart + s] == ' ') {
                        ++s;
                        --len;
                    }
                    if (len) {
                        uint32_t lastp = len - 1;
                        while (data[gstart + s + lastp] == ' ') --lastp;
                        if (lastp != 0) len = lastp + 1;
                    }
                }
                const int cl = cols[c].collen;
                if (cl > 0 && (uint32_t)cl < len) len = (uint32_t)cl;
            }
        } else {
            uint32_t o = first[2 * c], e = first[2 * c + 1];
            found = (o != NONE) & (e != NONE);
            uint32_t cs = c;
            lvl = cols[c].level;
            bool have = found;
            if (!found && H->carry)
                while (cs > 0) {
                    --cs;
                    o = first[2 * cs];
                    e = first[2 * cs + 1];
                    if (cols[cs].level > lvl) lvl = cols[cs].level;
                    if ((o != NONE) & (e != NONE)) {
                        have = true;
                        break;
                    }
                }
            if (have) {
                s = o + cols[cs].namelen + 2;
                const int cl = cols[cs].collen;
                int64_t l = (int64_t)e - (int64_t)s;
                if (l < 0)
                    l = (int64_t)reclen - (int64_t)s;  // close tag in front of the open tag: rest of the record
                else if (cl > 0 && cl < l)
                    l = cl;
                len = (uint32_t)l;
            }
        }
        dstart[c] = s;
        dlen[c] = len;
        dlvl[c] = lvl | ((uint32_t)found << 8);
    }
    __syncthreads();

    // 4. column by column: NUL cut, filtered length, and the bytes
    const uint64_t *tb = WRITE ? P.tile_base + (uint64_t)R.tile * (ncols + 1) : nullptr;
    const uint64_t row = WRITE ? tb[ncols] : 0;
    bool deferred = false;
    for (uint32_t c = 0; c < ncols; ++c) {
        const int64_t src = gstart + dstart[c];
        uint32_t len = dlen[c];
        const uint32_t lvl = dlvl[c] & 3u;
        const bool found = (dlvl[c] >> 8) & 1u;
        if (lvl == LVL_CHAR && nulcut) {  // vcolvalue[i] = temp.c_str(): the text ends at its first NUL
            if (tid == 0) sh[2] = len;
            __syncthreads();
            uint32_t cut = len;
            const int64_t send = src + len;
            for (int64_t cb = src & ~(int64_t)15; cb < send && cut == len; cb += BLOCK * CHUNK) {
                const int64_t pos = cb + (int64_t)tid * CHUNK;
                uint32_t z = pos < send ? mask16_eq(gchunk(data, n, pos), 0u) : 0u;
                while (z) {
                    const int64_t q = pos + __builtin_ctz(z);
                    z &= z - 1;
                    if (q >= src && q < send) {
                        atomicMin(&sh[2], (uint32_t)(q - src));
                        break;
                    }
                }
                __syncthreads();
                cut = sh[2];
                __syncthreads();  // (everyone has read the word before the next window may lower it)
            }
            len = cut;
        }
        // the filtered length, and with WRITE the surviving bytes in order
        uint8_t *dst = WRITE ? O.bytes[c] + tb[c] : nullptr;
        uint32_t flen = 0;  // (block-uniform running total)
        if (lvl == LVL_CHAR) {
            // text keeps every byte (the only cells that grow without bound: numbers and dates are clipped to
            // their column's length before the filter): nothing to count, a plain copy
            if (WRITE)
                for (uint32_t k = tid; k < len; k += BLOCK) dst[k] = data[src + k];
            flen = len;
        }
        for (uint32_t k0 = 0; k0 < len && lvl != LVL_CHAR; k0 += BLOCK * CHUNK) {
            const uint32_t kb = k0 + tid * CHUNK;
            uint32_t cnt = 0;
            uint8_t keep[CHUNK];
            for (uint32_t k = 0; k < CHUNK; ++k)
                if (kb + k < len) {
                    const uint32_t b = data[src + kb + k];
                    if (level_pass(b, lvl)) keep[cnt++] = (uint8_t)b;
                }
            const uint32_t incl = wave_incl_scan(cnt);
            if (lane == 63) wsum[wave] = incl;
            __syncthreads();
            uint32_t before = incl - cnt, tot = 0;
            for (int w = 0; w < BLOCK / 64; ++w) {
                if (w < wave) before += wsum[w];
                tot += wsum[w];
            }
            if (WRITE)
                for (uint32_t k = 0; k < cnt; ++k) dst[flen + before + k] = keep[k];
            flen += tot;
            __syncthreads();
        }
        const uint32_t typed = cols[c].typed;
        if (typed != T_NONE && (found || nulcut)) deferred = true;
        if (tid == 0) {
            if (!WRITE) {
                P.long_len[(uint64_t)blockIdx.x * ncols + c] = flen;
                P.tile_counts[(uint64_t)R.tile * (ncols + 1) + c] += flen;  // (this record is no other block's)
            } else {
                O.off[c][row + 1] = (uint32_t)(tb[c] + flen);
                if (typed != T_NONE) O.val[c][row] = 0;
            }
        }
    }
    if (WRITE) {
        if ((uint32_t)tid < W) {
            uint32_t fm = 0;
            for (uint32_t c = tid * 32; c < ncols && c < (uint32_t)tid * 32 + 32; ++c) fm |= ((dlvl[c] >> 8) & 1u) << (c & 31);
            P.found[row * W + tid] = fm;
            P.valid[row * W + tid] = 0;
        }
        if (tid == 0 && P.rec_off) P.rec_off[row + 1] = (uint64_t)gend + 1;
        if (deferred && tid == 0) raise_status(P, ST_NUMERIC_RANGE);
    }
}

// An end mark that holds a newline.  cifile::readline (reference public/_public.cpp:878-888) looks for the mark at
// the end of the record gathered so far, so a newline that follows the mark ends a record only if the whole mark
// lies behind the previous terminator.  The kernels take every newline that follows the mark's text; that is the
// same thing unless two such places lie within one mark's length of each other, i.e. unless a newline inside one
// occurrence of the mark itself follows another occurrence.  This pass looks for such a pair (one read of the
// input, only for such marks) and raises ST_MARK_OVERLAP: the parse is then refused instead of split wrongly.
__global__ __launch_bounds__(BLOCK) void mark_overlap_kernel(const uint8_t *data, uint64_t nbytes, const uint8_t *schema,
                                                             uint32_t *status) {
    const SchemaHdr *H = (const SchemaHdr *)schema;
    const int64_t n = (int64_t)nbytes;
    const int el = (int)H->endbz_len;
    const uint8_t *mark = endbz_text(H);
    bool found = false;
    for (int64_t pos = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) * CHUNK; pos < n; pos += (int64_t)gridDim.x * BLOCK * CHUNK) {
        uint32_t nl = mask16_eq(gchunk(data, n, pos), 0x0A0A0A0Au);
        while (nl) {
            const int64_t p = pos + __builtin_ctz(nl);
            nl &= nl - 1;
            if (p >= n || !gterm(data, H, p)) continue;
            for (int j = 0; j < el; ++j)  // the newlines inside this occurrence: does one of them follow the mark as well?
                if (mark[j] == '\n' && gterm(data, H, p - el + j)) found = true;
        }
    }
    if (found) atomicOr(status, ST_MARK_OVERLAP);
}

hipError_t launch_mark_overlap(const TileParams &P, hipStream_t stream) {
    if (P.n == 0) return hipSuccess;
    const uint64_t chunks = (P.n + CHUNK - 1) / CHUNK;
    uint64_t blocks = (chunks + BLOCK - 1) / BLOCK;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(mark_overlap_kernel, dim3((uint32_t)blocks), dim3(BLOCK), 0, stream, P.data, (uint64_t)P.n, P.schema, P.status);
    return hipGetLastError();
}

hipError_t launch_long_records(bool write, const TileParams &P, const OutPtrs &O, uint32_t count, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    const uint32_t lds = ((P.schema_bytes + 15) & ~15u) + 5 * P.ncols * 4 + 64;
    if (lds > 48 * 1024) {  // (beside the kernel's few static words)
        const void *fn = write ? (const void *)long_record_kernel<true> : (const void *)long_record_kernel<false>;
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    if (write)
        hipLaunchKernelGGL(long_record_kernel<true>, dim3(count), dim3(BLOCK), lds, stream, P, O);
    else
        hipLaunchKernelGGL(long_record_kernel<false>, dim3(count), dim3(BLOCK), lds, stream, P, O);
    return hipGetLastError();
}

// Exclusive prefix over tiles of every counter, in two small kernels.  (1) One block per
// (segment of SCAN_SEG tiles, counter): prefix inside the segment, segment total.  (2) The same
// grid: every block sums the totals of the segments in front of its own and adds that to its
// prefixes; the last segment's blocks also write the grand totals (row ntiles of `base`).
constexpr uint32_t SCAN_SEG = 4096;
__device__ __forceinline__ uint64_t block_incl_scan1024(uint64_t v, uint64_t *wtot, uint64_t &total) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint64_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint64_t u = __shfl_up(incl, d, 64);
        if (lane >= d) incl += u;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    uint64_t before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const uint64_t t = wtot[w];
        if (w < wave) before += t;
        all += t;
    }
    total = all;
    return before + incl;
}

__global__ __launch_bounds__(1024) void scan_local_kernel(const uint32_t *counts, uint64_t *base, uint64_t *segtot,
                                                          uint32_t ntiles, uint32_t ncnt) {
    __shared__ uint64_t wtot[16];
    const uint32_t seg = blockIdx.x, j = blockIdx.y;
    const uint32_t t0 = seg * SCAN_SEG + threadIdx.x * 4;
    uint32_t v[4];
    uint64_t sum = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[i] = t0 + i < ntiles ? counts[(uint64_t)(t0 + i) * ncnt + j] : 0u;
        sum += v[i];
    }
    uint64_t total;
    uint64_t run = block_incl_scan1024(sum, wtot, total) - sum;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (t0 + i < ntiles) base[(uint64_t)(t0 + i) * ncnt + j] = run;
        run += v[i];
    }
    if (threadIdx.x == 0) segtot[(uint64_t)seg * ncnt + j] = total;
}

// report (may be null): where the host reads the totals, memory of the host that the device writes
// to directly, and in front of them (report_status words back) the two status words of the sizing
// pass, which this kernel passes on: the host needs no copy between the passes, only the wait.
__global__ __launch_bounds__(1024) void scan_add_kernel(uint64_t *base, const uint64_t *segtot, uint32_t ntiles,
                                                        uint32_t ncnt, uint64_t *report, const uint64_t *status,
                                                        uint32_t report_status) {
    __shared__ uint64_t wtot[16];
    const uint32_t seg = blockIdx.x, j = blockIdx.y;
    uint64_t part = 0;
    for (uint32_t s = threadIdx.x; s < seg; s += 1024) part += segtot[(uint64_t)s * ncnt + j];
    uint64_t pre;
    (void)block_incl_scan1024(part, wtot, pre);
    const uint32_t t0 = seg * SCAN_SEG + threadIdx.x * 4;
    if (seg)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (t0 + i < ntiles) base[(uint64_t)(t0 + i) * ncnt + j] += pre;
    if (seg == gridDim.x - 1 && threadIdx.x == 0) {
        const uint64_t tot = pre + segtot[(uint64_t)seg * ncnt + j];
        base[(uint64_t)ntiles * ncnt + j] = tot;
        if (report) report[j] = tot;
        if (report && status && j == 0) {
            report[-(int)report_status] = status[0];
            report[1 - (int)report_status] = status[1];
        }
    }
}

// Expands a compact column into fixed-stride NUL-terminated slots (the array-bind layout).
__global__ __launch_bounds__(256) void fixed_stride_kernel(const uint32_t *off, const uint8_t *bytes, uint8_t *slab,
                                                           uint64_t rows, uint32_t stride) {
    const uint64_t total = rows * stride;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / stride;
        const uint32_t k = (uint32_t)(i - r * stride);
        const uint32_t o = off[r], len = off[r + 1] - o;
        slab[i] = (k < len && k + 1 < stride) ? bytes[o + k] : 0;
    }
}

// Second stage of the typed conversion.  The tile kernel converts cells of up to 16 source
// bytes itself and leaves the others invalid (raising ST_NUMERIC_RANGE).  This kernel visits
// every typed cell whose valid bit is clear, reads the cell text back from the column arena and
// runs the general parser: 128-bit arithmetic up to 38 digits (mdc_numparse.h), multi-precision
// beyond (mdc_bignum.h).  Re-converting a cell that was legitimately invalid yields invalid
// again, so the criterion needs no extra bookkeeping.
__global__ __launch_bounds__(256) void deferred_convert_kernel(const uint8_t *schema, uint32_t *const *off,
                                                               uint8_t *const *bytes, uint64_t *const *val,
                                                               uint32_t *valid, const uint32_t *found, uint64_t rows) {
    const SchemaHdr *H = (const SchemaHdr *)schema;
    const DevCol *cols = (const DevCol *)(H + 1);
    const uint32_t ncols = H->ncols, W = H->words;
    const bool always = H->carry != 0 && H->mode == MODE_XML;
    const uint64_t total = rows * ncols;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / ncols;
        const uint32_t c = (uint32_t)(i - r * ncols);
        const uint32_t typed = cols[c].typed;
        if (typed == T_NONE) continue;
        const uint32_t bit = 1u << (c & 31);
        if (valid[r * W + (c >> 5)] & bit) continue;
        if (!always && !(found[r * W + (c >> 5)] & bit)) continue;
        const uint32_t o = off[c][r], len = off[c][r + 1] - o;
        uint64_t v = 0;
        bool ok = false, range = false;
        if (typed == T_TIME) {
            ok = time_convert(bytes[c] + o, len, H->tz_offset, v);
        } else {
            cell_walk<false>(bytes[c] + o, len, LVL_CHAR, false, typed, nullptr, v, ok, range);
            if (range) ok = big_convert(bytes[c] + o, len, typed == T_F64, v);
        }
        if (ok) {
            val[c][r] = v;
            atomicOr(&valid[r * W + (c >> 5)], bit);
        }
    }
}

// ---------------------------------------------------------------------------------------
// launchers (called from mdc_abi.cpp)

hipError_t launch_long_float(const uint8_t *schema, uint32_t *const *off, uint8_t *const *bytes,
                             uint64_t *const *val, uint32_t *valid, const uint32_t *found, uint64_t rows,
                             uint32_t ncols, hipStream_t stream) {
    if (rows == 0) return hipSuccess;
    uint64_t total = rows * ncols;
    uint32_t blocks = (uint32_t)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(deferred_convert_kernel, dim3(blocks), dim3(256), 0, stream, schema, off, bytes, val, valid, found,
                       rows);
    return hipGetLastError();
}

hipError_t launch_emit_kernel(const TileParams &P, const OutPtrs &O, hipStream_t stream) {
    if (P.ntiles == 0) return hipSuccess;
    if (!P.cellmap || !P.map_stride) return hipSuccess;  // no map at all: the full write kernel does every tile
    // (schemas with time or binary32 columns take the instance that converts them on the spot)
    const bool many = (P.L.RB & 63u) == 0;  // (the layout for inputs of short records)
    void (*const fn)(const TileParams, const OutPtrs) = P.special_cols ? (many ? emit_kernel<true, true> : emit_kernel<true, false>)
                                                        : (many ? emit_kernel<false, true> : emit_kernel<false, false>);
    hipError_t e = hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    uint32_t lds = P.L.total;
#ifdef MDC_ABLATE
    if (const char *e = getenv("MDC_DEBUG_LDS")) lds = (uint32_t)atoi(e);  // occupancy experiments
#endif
    hipLaunchKernelGGL(fn, dim3(P.ntiles), dim3(BLOCK), lds, stream, P, O);
    return hipGetLastError();
}

hipError_t launch_template_kernel(const TileParams &P, hipStream_t stream) {
    if (P.ntiles == 0 || !P.use_tmpl) return hipSuccess;
    const uint32_t lds = ((P.schema_bytes + 15) & ~15u) + TMPL_WINDOW + 64;
    hipLaunchKernelGGL(template_kernel, dim3(1), dim3(64), lds, stream, P);
    return hipGetLastError();
}

// mode: 0 = sizing pass, 1 = full write pass, 2 = one-pass parse
hipError_t launch_tile_kernel(int mode, const TileParams &P, const OutPtrs &O, hipStream_t stream) {
    if (P.ntiles == 0) return hipSuccess;
    void (*const fn)(const TileParams, const OutPtrs) = mode == 0   ? tile_kernel<PASS_SIZING>
                                                        : mode == 1 ? tile_kernel<PASS_WRITE>
                                                        : P.special_cols ? tile_kernel<PASS_FUSED_SPECIAL>
                                                                         : tile_kernel<PASS_FUSED>;
    // per device and cheap: set on every launch so that each GPU of a process gets it
    hipError_t e = hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    uint32_t lds = P.L.total;
#ifdef MDC_ABLATE
    if (const char *e = getenv("MDC_DEBUG_LDS")) lds = (uint32_t)atoi(e);  // occupancy experiments
#endif
    hipLaunchKernelGGL(fn, dim3(P.ntiles), dim3(BLOCK), lds, stream, P, O);
    return hipGetLastError();
}

// base holds (ntiles + 1 + segments) rows of ncnt counters: prefixes, totals, segment totals
uint32_t scan_segments(uint32_t ntiles) { return ntiles ? (ntiles + SCAN_SEG - 1) / SCAN_SEG : 1; }

hipError_t launch_scan_tiles(const uint32_t *counts, uint64_t *base, uint32_t ntiles, uint32_t ncnt, uint64_t *report,
                             const uint64_t *status, uint32_t report_status, hipStream_t stream) {
    const uint32_t nseg = scan_segments(ntiles);
    uint64_t *segtot = base + (uint64_t)(ntiles + 1) * ncnt;
    hipLaunchKernelGGL(scan_local_kernel, dim3(nseg, ncnt), dim3(1024), 0, stream, counts, base, segtot, ntiles, ncnt);
    hipLaunchKernelGGL(scan_add_kernel, dim3(nseg, ncnt), dim3(1024), 0, stream, base, segtot, ntiles, ncnt,
                       report, status, report_status);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Scratch of the entry point that allocates nothing (mdc_parse_device_noalloc): the per-tile
// counters, their prefixes, the tile headers and the table of output pointers, which the
// allocating entry points keep in device memory of the parser.  Every GPU the library runs on
// holds its own copy of the array until the process ends, at an address of its own: the array's
// name is only a handle for hipGetSymbolAddress, and host code uses the address that call gave
// for the device in question, never the name.
static __device__ __align__(256) uint8_t scratch[NOALLOC_SCRATCH_BYTES];
constexpr int SCRATCH_MAX_DEVICES = 64;
static void *g_scratch_addr[SCRATCH_MAX_DEVICES];

// Once per device, with that device current (the library does it when it first uses a device).
hipError_t noalloc_scratch_init(int device) {
    if (device < 0 || device >= SCRATCH_MAX_DEVICES) return hipErrorInvalidDevice;
    void *ptr = nullptr;
    const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    g_scratch_addr[device] = ptr;
    return hipSuccess;
}
// this device's copy of the array (nullptr: the device was never set up)
void *noalloc_scratch_addr(int device) {
    return device >= 0 && device < SCRATCH_MAX_DEVICES ? g_scratch_addr[device] : nullptr;
}

hipError_t launch_fixed_stride(const uint32_t *off, const uint8_t *bytes, uint8_t *slab, uint64_t rows,
                               uint32_t stride, hipStream_t stream) {
    if (rows == 0) return hipSuccess;
    uint64_t total = rows * stride;
    uint32_t blocks = (uint32_t)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(fixed_stride_kernel, dim3(blocks), dim3(256), 0, stream, off, bytes, slab, rows, stride);
    return hipGetLastError();
}

}  // namespace mdc
