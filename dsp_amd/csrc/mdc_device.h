// mdc_device.h -- structures shared by the host side (mdc_abi.cpp) and the gfx950 kernels
// (mdc_kernels.hip).  Everything here is plain data laid out identically on both sides.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace mdc {

// ---- tile geometry ---------------------------------------------------------------------
// A tile is TILE consecutive input bytes.  A tile owns every record whose terminating '\n'
// lies inside it; the first owned record may start up to LBMAX bytes in front of the tile
// (look-behind; the capacity is chosen per launch, LdsLayout::lbmax <= LBMAX).  The block keeps
// [tile_start-lbmax, tile_start+TILE) in LDS, byte `pos` at LDS index pos - tile_start + lbmax.
constexpr int BLOCK = 256;
constexpr int TILE = 16384;
constexpr int LBMAX = 16320;                             // largest look-behind capacity
constexpr int CHUNK = 16;                                // bytes per lane per step (one dwordx4)
constexpr int DATA_SLACK = 64;                           // readable bytes past the image
constexpr int LB_FIRST = 512;                            // first look-behind window
constexpr int LB_STEP = BLOCK * CHUNK;                   // later windows
constexpr int MAX_COLS = 256;
constexpr int MAX_ENDBZ = 32;       // end marks and separators up to these lengths lie in the schema header ...
constexpr int MAX_SEP = 16;
constexpr int MAX_LONG_TEXT = 255;  // ... longer ones (up to this) in front of the table of powers of ten, see long_text
constexpr uint32_t NONE = 0xFFFFFFFFu;
// The table of powers of ten at SchemaHdr::p10_off (16-byte aligned): double[23] = 10^0 .. 10^22, one
// double of padding, then P10_PAIRS pairs {10^k, 1 / 10^k} (the quotient of an integer below 10^8 by
// 10^k, k <= 8, is taken with one multiplication and one correction step: see cell_simple).
constexpr int P10_PAIR_AT = 24;
constexpr int P10_PAIRS = 9;
constexpr int P10_TABLE_DOUBLES = P10_PAIR_AT + 2 * P10_PAIRS;  // 42: a multiple of two doubles
constexpr size_t NOALLOC_SCRATCH_BYTES = (size_t)2 << 20;  // static device scratch of mdc_parse_device_noalloc
// Tag template (tagged input): the '<' events of one sample record, in order.  Event k of a
// record whose events all repeat the sample's tag texts byte for byte matches what sample event
// k matched, so such a record needs no table lookup.  In memory: 4 words of header (word 0 =
// number of events, 0 = no template), then one 4-word entry per event: the first 8 bytes of the
// tag text through '>' (zero beyond it), bytes 8..11, and a word of two shift counts and the
// pattern: 64 - 8 x (text bytes among the first 8) | (32 - 8 x (text bytes among bytes 8..11)) << 8
// | pattern index << 16 (TMPL_NOPAT: the sample's event matched no pattern).  A 64-bit shift to
// the left by a count leaves the text's bytes of a window, and nothing of the second word when the
// text ends before it.
// Behind the TMPL_MAX entries: the column plan, two words per column, for records that follow the
// template: source column of the cell (itself, or the nearest earlier column whose tags the sample
// has, when values carry over) | filter level << 8 | found << 10 | has a source << 11 |
// (name length + 2) << 16, and the source column's clip length.
constexpr uint32_t TMPL_MAX = 64;
constexpr uint32_t TMPL_PLAN_WORD = 4 + 4 * 64;  // word index of the plan (= 4 + 4 * TMPL_MAX)
constexpr uint32_t TMPL_MAXLEN = 12;
constexpr uint32_t TMPL_NOPAT = 0xFFFFu;
constexpr uint32_t TMPL_WINDOW = 4096;  // the sample record is the second record of the input and must end in this window

static_assert(TILE == BLOCK * 64, "every lane owns 64 tile bytes");
static_assert(TILE + LBMAX + 64 <= 32768, "cell descriptors hold 15-bit LDS indices");

// filter level of a cell: nested projections char (everything) > number > date
enum : uint8_t { LVL_CHAR = 0, LVL_NUMBER = 1, LVL_DATE = 2 };

enum : uint32_t { MODE_XML = 0, MODE_CSV = 1 };

// status bits raised by the kernels (OR-ed into one device word)
enum : uint32_t {
    ST_RECORD_TOO_LONG = 1u,
    ST_NUMERIC_RANGE = 2u,
    ST_INTERNAL = 4u,
    ST_UNMAPPED = 8u,  // a tile of the sizing pass left no cell map: the write pass re-derives it
    ST_OVERFLOW = 16u, // one-pass parse: an output array is smaller than what the input holds
    ST_GIVEUP = 32u,   // one-pass parse: a tile waited too long for the tiles in front of it
    ST_MARK_OVERLAP = 64u,  // an end mark that holds a newline: two of its occurrences overlap in the input
};

// One-pass parse: what a tile publishes for the tiles behind it, one 64-bit word per counter (bytes of
// every column, then rows), written with one store and read with one load: the kind in the two top
// bits (0: nothing yet), the value below.  An aggregate is the tile's own count, an inclusive prefix
// the count of the tile and of every tile in front of it.
constexpr uint64_t PUB_AGG = 1ull << 62, PUB_INCL = 2ull << 62, PUB_VALUE = (1ull << 62) - 1;
// status word 3 of a one-pass parse: tiles it left to the full write kernel
constexpr uint32_t FUSED_SPIN_LIMIT = 1u << 16;  // polls before a tile gives up (the host then parses in two passes)

struct DevCol {
    uint8_t name[32];
    uint8_t namelen;
    uint8_t level;   // LVL_*
    uint8_t typed;   // MDC_T_*
    uint8_t pad;
    int32_t collen;  // clip length, <=0: none
};
static_assert(sizeof(DevCol) == 40, "DevCol layout");

#ifdef __HIPCC__
#define MDC_HD __host__ __device__
#else
#define MDC_HD
#endif

// hash of a tag key (the first eight bytes of text+'>' as two little-endian words)
MDC_HD inline uint32_t ht_hash(uint32_t lo, uint32_t hi, uint32_t seed) {
    // (tag names of one table tend to differ in one or two bytes only: two multiplications with a
    // fold between them spread those over the low bits the slot index is taken from)
    uint32_t h = (lo ^ seed) * 0x9E3779B1u;
    h ^= h >> 15;
    h = (h + hi) * 0x85EBCA77u;
    return h ^ (h >> 13);
}

// Hash table of tag texts.  The host looks for a seed under which all distinct keys fall into
// distinct slots, so that a lookup is ONE probe; where it finds none (wide schemas of similar
// names) it takes the seed with the fewest shared slots.  key = first min(8, tlen+1) bytes of
// text+'>'; bytes 8..15 of text+'>' are compared through (k2 & m2).  Entries that share a slot
// (another key, duplicate column names, names equal in their first eight bytes) are chained
// through `next` (index into the same array, 0 = end; chain entries live behind the hashed
// slots); a lookup compares the key of every entry of the chain.
struct HtEntry {
    uint32_t klo, khi;
    uint32_t meta;  // bit31 used | bit30 text longer than 16 bytes | col<<16 | kind<<8 | tlen
    uint32_t next;
    uint32_t k2lo, k2hi, m2lo, m2hi;
};
static_assert(sizeof(HtEntry) == 32, "HtEntry layout");

struct SchemaHdr {
    uint32_t ncols, npat, ht_mask, endbz_len;
    uint32_t carry, mode, words, sep_len;
    uint32_t bdelspace, any_typed, ht_seed, p10_off;  // p10_off: byte offset of double[23] = 10^k
    uint8_t endbz[MAX_ENDBZ];
    uint8_t sep[MAX_SEP];
    uint8_t endbz_tail[8];  // endbz right-aligned in 8 bytes (when endbz_len <= 8)
    int32_t tz_offset;      // MDC_T_TIME columns: seconds east of UTC of the zone mktime would run in
    uint8_t sep_fast;       // delimited mode: the separator (1 to 4 bytes below 0x80, no '\n') is found with byte masks
    uint8_t pad3[3];
    // followed by DevCol[ncols], HtEntry[ht_mask+1+chained] (16-byte aligned), an end mark longer than
    // MAX_ENDBZ and a separator longer than MAX_SEP (each padded to 16 bytes; the header's arrays then hold
    // their first bytes), double[23]
};
static_assert(sizeof(SchemaHdr) % 16 == 0, "SchemaHdr must keep 16-byte alignment");

// The end mark and the separator of a schema blob that begins at H (in LDS or in global memory).
MDC_HD inline uint32_t long_text_room(uint32_t len, uint32_t in_header) { return len > in_header ? (len + 15) & ~15u : 0; }
MDC_HD inline const uint8_t *sep_text(const SchemaHdr *H) {
    return H->sep_len <= (uint32_t)MAX_SEP ? H->sep : (const uint8_t *)H + H->p10_off - long_text_room(H->sep_len, MAX_SEP);
}
MDC_HD inline const uint8_t *endbz_text(const SchemaHdr *H) {
    return H->endbz_len <= (uint32_t)MAX_ENDBZ
               ? H->endbz
               : (const uint8_t *)H + H->p10_off - long_text_room(H->sep_len, MAX_SEP) - long_text_room(H->endbz_len, MAX_ENDBZ);
}

// byte offsets of the LDS regions (computed on the host from the schema)
struct LdsLayout {
    uint32_t data;    // TILE+LBMAX+DATA_SLACK bytes
    uint32_t evlist;  // evcap x (u16 position + u8 batch rank) of the '<' events; later the cell staging area
    uint32_t evt;     // RB*npat u16: first position of every tag pattern per record
                      // (delimited mode: the field descriptors)
    uint32_t recend;  // (RB+1) u32
    uint32_t vmask;   // RB*words u32
    uint32_t fmask;   // RB*words u32
    uint32_t coltot;  // 2*(ncols+1) u32: batch totals, tile totals
    uint32_t colout;  // ncols x 32 bytes: output pointers and tile base of every column
    uint32_t schema;  // schema blob
    uint32_t misc;    // block-wide scalars
    uint32_t total;
    uint32_t RB;         // records per batch (<= 256)
    uint32_t evcap;      // events per compaction window
    uint32_t stage_cap;  // = evcap*3
    uint32_t lbmax;      // look-behind capacity in bytes (multiple of 16)
    uint32_t endl;       // one-pass parse: u16 [ncols][RB] end of every cell within its column's bytes of the tile
    uint32_t tmpl;       // tagged input: event index per record, the template's column plan and its entries
    uint32_t tmpl_bytes; // (the sizing pass keeps them in the space of the two row masks, which it does not use)
};

// A record longer than the look-behind of the tile that ends it (see long_record_kernel).
struct LongRec {
    uint64_t gend;  // position of the record's terminating '\n'
    uint64_t from;  // no record terminator lies in [from, gend)
    uint32_t tile;  // the tile that owns the record (it is that tile's first)
    uint32_t pad;
};
constexpr uint32_t HDR_LONGREC = 0xFFFFFFFEu;  // tile header word 0: not mapped, and the first record is a long one

struct TileParams {
    const uint8_t *data;
    uint64_t n;
    const uint8_t *schema;
    uint32_t schema_bytes;
    uint32_t ncols, words, p10_off;  // copies of the schema header's fields (kernel arguments need no load)
    uint32_t ntiles;
    LdsLayout L;
    // count mode output / write mode input
    uint32_t *tile_counts;       // [ntiles][ncols+1]: bytes per column, then rows
    const uint64_t *tile_base;   // [ntiles+1][ncols+1] exclusive prefix (row ntiles = totals)
    // write mode outputs
    uint64_t *rec_off;           // rows+1 or null
    uint32_t *const *off;        // [ncols] -> rows+1
    uint8_t *const *bytes;       // [ncols]
    uint64_t *const *val;        // [ncols] (null when untyped)
    uint32_t *valid;             // rows*words
    uint32_t *found;             // rows*words
    uint32_t *status;
    // cell map: the sizing pass leaves, for every record of a tile, one word per column (source
    // start | length | filter level | found) and the record's end, so that the write pass needs
    // neither the tag search nor the record split again.  Column-major: word (c, slot + r) at
    // cellmap[c * map_cap + slot + r], c == ncols holds the record ends.
    uint32_t *cellmap;
    uint32_t map_cap;            // record slots of the map = ntiles * map_stride
    uint32_t map_stride;         // record slots per tile: tile t owns [t * map_stride, (t + 1) * map_stride)
    uint32_t *tile_hdr;          // [ntiles][2]: slot (NONE = not mapped), first record start | has_nul << 31;
                                 // HDR_LONGREC, entry in long_list: not mapped and the first record is a long one
    LongRec *long_list;          // [ntiles] long records met by the sizing pass (their number: status word 2)
    uint32_t *long_len;          // [ntiles][ncols] bytes of each long record's cells
    uint32_t only_unmapped;      // write kernel: leave the mapped tiles to the emit kernel
    uint32_t *tmpl;              // tag template (always a valid pointer; word 0 == 0: none)
    uint8_t *host_flags;         // write pass: one byte per status bit in memory of the host (null: sizing pass)
    uint32_t use_tmpl;           // tagged input: the sizing pass may use the template
    uint32_t special_cols;       // the schema has time or binary32 columns (picks the write kernel's instance)
    uint32_t tmpl_sample;        // which record the template kernel samples: 0 = the second of the input, 1 = the third ...
    uint32_t debug;  // diagnostic builds only: phases to skip (ablation timing)
    uint32_t records_only;  // the record split alone: no tag search, no cells (every cell comes out empty)
    // one-pass parse
    uint64_t *pub;          // [ncols+1][ntiles] published counts (PUB_*), zeroed before the launch
    const uint64_t *caps;   // [ncols+1] what the output arrays hold: bytes per column, then rows
    uint64_t *fused_totals; // [ncols+1] totals, written by the last tile (memory of the host)
    uint32_t spin_limit;    // polls a tile may spend on one tile in front of it before it gives up (FUSED_SPIN_LIMIT)
};

// ------------------------------------------------------------------------------------------
// The output arrays of a write pass, handed to its kernels as a kernel argument: no copy to the device
// stands between the two passes for them.  (The second conversion stage, which rarely runs, reads the
// same pointers from a table in device memory: TileParams::off, bytes, val.)
struct OutPtrs {
    uint32_t *off[MAX_COLS];    // -> rows+1
    uint8_t *bytes[MAX_COLS];
    uint64_t *val[MAX_COLS];    // (null when untyped)
};

// writer side (mdc_format.hip): columns -> "<name>value</name>...<endl/>\n" text

// status bits of the formatting kernels
constexpr uint32_t FS_NEED_EXACT = 1;  // a value holds a NUL byte or exceeds its fieldlen
constexpr uint32_t FS_BAD_OFFSETS = 2; // a column's offsets descend or pass the end of its arena

constexpr int FMT_BLOCK = 256;

// Field table at the head of the formatter's blob, one entry per field; texts follow, each
// 16-aligned and readable up to the next multiple of 16.  A field's text is what stands between
// its value and the next one: "</name><next>", after the last field "</name><endl/>\n</data>\n"
// (join_len stops before "</data>\n").  Two more texts open a row: "<first>" at
// FormatParams::pre_plain_off and "<data>\n<first>" at pre_open_off.
struct FormatField {
    uint32_t text_off;
    uint32_t name_len;
    uint32_t fieldlen; // 0 = no cut
    uint32_t join_len;
};

struct FormatParams {
    const uint32_t *const *off;  // device array [ncols] -> rows+1 offsets
    const uint8_t *const *bytes; // device array [ncols] -> arenas
    const uint8_t *blob;
    uint32_t blob_bytes;         // multiple of 16
    uint32_t pre_plain_off, pre_open_off;
    uint32_t ncols;
    uint32_t R, Rshift;          // records per block, a power of two <= FMT_BLOCK
    uint32_t fixed;              // bytes of a record that are not values: tags and "<endl/>\n"
    uint64_t rows, maxcount;
    uint64_t sumbase;            // sum over columns of off[c][0]
    uint64_t total;              // length of the whole text
    uint8_t *out;
    uint64_t *file_off;          // start of every file
    uint32_t *status;
    // exact mode: lengths measured beforehand
    const uint32_t *elen;        // [ncols][rows] value lengths after the cuts
    const uint64_t *recpos;      // [rows+1] record starts, file framing not counted
    // LDS regions (byte offsets)
    uint32_t debug;              // diagnostic builds only: parts to skip (ablation timing)
    uint32_t lds_ptrs, lds_cpos, lds_src, lds_rstart, lds_rflag, lds_misc, lds_stage, stage_cap, lds_total;
};

// One of timetostr's formats (reference public/_public.cpp:631-722) as a selection of bytes of
// "yyyy-mm-dd hh:mi:ss": byte k of the text is byte idx[k] of that string.
constexpr int TT_BLOCK = 256;
struct TimeFmt {
    uint32_t width;
    uint8_t idx[20];
};

}  // namespace mdc
