// mdc_format.hip -- writer-side kernels of libmdc_parse.so (gfx950).
//
// Columns (u32 offsets + byte arena each) -> the text of the reference's extractor,
// tools/dminingoracle.cpp:196-256: per row "<name>value</name>" for every field, "<endl/>\n",
// files framed by "<data>\n" and "</data>\n" and closed after every maxcount rows.
//
// Where no value needs cutting, the start of every record follows in closed form from the
// column offsets (row index x tag bytes + sum of the columns' offsets + file framing), so one
// kernel does all the work: a block takes R consecutive rows, lays their text out in LDS and
// copies it to HBM in 16-byte stores.  Values with a NUL byte, or longer than their fieldlen,
// are found on the way and reported; the host then measures every value (format_measure_kernel),
// scans the record lengths and runs the same kernel in its exact mode.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "mdc_device.h"

namespace mdc {

#ifdef MDC_ABLATE
#define FSKIP(bit) (P.debug & (bit)) // timing experiments only: 1 value words, 2 stores, 4 pieces
#else
#define FSKIP(bit) false
#endif

typedef __attribute__((address_space(1))) const uint8_t fg_u8;
typedef __attribute__((address_space(1))) const uint32_t fg_u32;
typedef uint32_t fu32x4 __attribute__((ext_vector_type(4)));

// Where row r falls among the files of maxcount rows each (maxcount 0: one file).
struct FilePos {
    uint64_t file;    // index of the row's file
    uint64_t framing; // bytes of "<data>\n" (7) and "</data>\n" (8) written before the row's own first byte
    bool opens, closes;
};

__device__ __forceinline__ FilePos file_pos(uint64_t r, uint64_t m, uint64_t rows) {
    FilePos f;
    uint64_t q = 0, rem = r;
    if (m) {
        if ((r | m) >> 32) {
            q = r / m;
            rem = r % m;
        } else {
            q = (uint32_t)r / (uint32_t)m;
            rem = (uint32_t)r % (uint32_t)m;
        }
    }
    f.file = q;
    f.opens = rem == 0;
    f.closes = (m && rem + 1 == m) || r + 1 == rows;
    f.framing = m ? 7 * (q + (rem ? 1 : 0)) + 8 * q : (r ? 7 : 0);
    return f;
}

constexpr int UPL = 4; // (field, row) units per lane whose value words are fetched ahead
constexpr int PW = 5;  // words fetched ahead per unit: values of up to 4 * PW - 3 bytes

// A row's text is cut into one piece per field: the value and the constant text up to the next
// value, "</name><next>" (after the last field "</name><endl/>\n", and "</data>\n" where the row
// closes a file); the first field's piece starts with "<name>" (after "<data>\n" where the row
// opens a file).  A piece belongs to one lane, which may therefore write whole words and let a
// later write of its own cover the overrun; only the piece's last word is cut to the byte.
//
// Phases of a block (R rows, a power of two; unit u = (field u >> Rshift, row u & (R - 1))):
//   1  every lane, for its units: the value's offsets, hence its length; the words of the arena that
//      hold the value are requested at once and stay in registers until phase 3
//   2  one lane per row: prefix of the field lengths -> where each field starts in the row, and
//      the row's start in the text (closed form, or the measured starts in exact mode)
//   3  every lane, for its units: the piece into the staging window in LDS
//   4  the window to HBM in 16-byte stores
// Rows whose text passes the window repeat 3-4 per window, byte by byte; units beyond UPL per lane
// and values longer than the words fetched ahead read the arena in phase 3.
template <bool EXACT>
__global__ __launch_bounds__(FMT_BLOCK) void format_kernel(const FormatParams P) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t lane = threadIdx.x;
    const uint32_t R = P.R, ncols = P.ncols, Rshift = P.Rshift;
    const uint64_t r0 = (uint64_t)blockIdx.x << Rshift;
    const uint32_t nrec = (uint32_t)(P.rows - r0 < R ? P.rows - r0 : R);

    const FormatField *fld = (const FormatField *)lds;
    const uint32_t *const *offp = (const uint32_t *const *)(lds + P.lds_ptrs);
    const uint8_t *const *bytp = (const uint8_t *const *)(lds + P.lds_ptrs + 8 * ncols);
    uint32_t *cpos = (uint32_t *)(lds + P.lds_cpos);
    uint32_t *srcs = (uint32_t *)(lds + P.lds_src);
    uint64_t *rstart = (uint64_t *)(lds + P.lds_rstart);
    uint8_t *rflag = lds + P.lds_rflag;
    uint64_t *misc = (uint64_t *)(lds + P.lds_misc); // [0] block start, [1] block end, [2] bad
    uint8_t *stage = lds + P.lds_stage;
    const uint32_t STAGE = P.stage_cap;

    for (uint32_t i = lane * 16; i < P.blob_bytes; i += FMT_BLOCK * 16)
        *(fu32x4 *)(lds + i) = *(const fu32x4 *)(P.blob + i);
    for (uint32_t c = lane; c < ncols; c += FMT_BLOCK) {
        ((const uint32_t **)(lds + P.lds_ptrs))[c] = P.off[c];
        ((const uint8_t **)(lds + P.lds_ptrs + 8 * ncols))[c] = P.bytes[c];
    }
    if (lane == 0) misc[2] = 0;
    __syncthreads();

    // ---- phase 1
    const uint32_t cells = ncols << Rshift;
    uint32_t ua[UPL], ub[UPL], ulen[UPL], uw[UPL][PW];
    uint32_t bad = 0, cut = 0;
    // units beyond the ones kept in registers: lengths only
    for (uint32_t u = lane + UPL * FMT_BLOCK; u < cells; u += FMT_BLOCK) {
        const uint32_t i = u & (R - 1), c = u >> Rshift;
        if (i >= nrec) continue;
        fg_u32 *o = (fg_u32 *)offp[c];
        const uint32_t a = o[r0 + i], b = o[r0 + i + 1];
        uint32_t len = b - a;
        if (b < a) {
            bad = 1;
            len = 0;
        }
        if (EXACT)
            len = ((fg_u32 *)P.elen)[(uint64_t)c * P.rows + r0 + i];
        else if (fld[c].fieldlen && len > fld[c].fieldlen)
            cut = 1;
        cpos[u] = len;
        srcs[u] = a;
    }
    // (a lane without a k-th unit reads unit 0's offsets instead: loads without a branch around
    // them are all requested before the first wait)
#pragma unroll
    for (int k = 0; k < UPL; ++k) {
        const uint32_t u = lane + k * FMT_BLOCK;
        const uint32_t us = u < cells && (u & (R - 1)) < nrec ? u : 0;
        const uint32_t i = us & (R - 1), c = us >> Rshift;
        fg_u32 *o = (fg_u32 *)offp[c];
        ua[k] = o[r0 + i];
        ub[k] = o[r0 + i + 1];
        ulen[k] = EXACT ? ((fg_u32 *)P.elen)[(uint64_t)c * P.rows + r0 + i] : 0;
    }
    // Every lane waits for its offsets here, branch or not: registers with a load that some path
    // never waited for would otherwise make the compiler wait for the value words too at the
    // next reuse of those registers, long before phase 3.
#pragma unroll
    for (int k = 0; k < UPL; ++k) asm volatile("" : "+v"(ua[k]), "+v"(ub[k]), "+v"(ulen[k]));
    // lengths first, then the requests for the value words
    fg_u32 *uptr[UPL];
    uint32_t unw[UPL];
#pragma unroll
    for (int k = 0; k < UPL; ++k) {
        const uint32_t u = lane + k * FMT_BLOCK, i = u & (R - 1), c = u >> Rshift;
        unw[k] = 0;
        uptr[k] = nullptr;
        if (u < cells && i < nrec) {
            uint32_t len = ub[k] - ua[k];
            if (ub[k] < ua[k]) {
                bad = 1;
                len = 0;
            }
            if (EXACT)
                len = ulen[k];
            else if (fld[c].fieldlen && len > fld[c].fieldlen)
                cut = 1;
            ulen[k] = len;
            cpos[u] = len;
            srcs[u] = ua[k];
            const uint64_t ga = (uint64_t)(bytp[c] + ua[k]);
            const uint32_t sh = (uint32_t)ga & 3;
            uptr[k] = (fg_u32 *)(ga - sh);
            unw[k] = len ? (sh + len + 3) >> 2 : 0;
        }
    }
#pragma unroll
    for (int k = 0; k < UPL; ++k) {
#pragma unroll
        for (int j = 0; j < PW; ++j) uw[k][j] = (uint32_t)j < unw[k] && !FSKIP(1) ? uptr[k][j] : 0;
    }
    if (bad) misc[2] = 1;
    if (cut) atomicOr(P.status, FS_NEED_EXACT);
    __syncthreads();

    // ---- phase 2
    if (lane < nrec) {
        const uint64_t r = r0 + lane;
        uint64_t sum = 0;
        uint32_t run = 0;
        for (uint32_t c0 = 0; c0 < ncols; c0 += 4) { // reads of four columns, then their writes
            uint32_t len[4], nl[4], sr[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool in = c0 + q < ncols;
                const uint32_t c = in ? c0 + q : c0;
                len[q] = cpos[(c << Rshift) + lane];
                sr[q] = in ? srcs[(c << Rshift) + lane] : 0;
                nl[q] = fld[c].name_len;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (c0 + q < ncols) {
                    cpos[((c0 + q) << Rshift) + lane] = run;
                    run += 2 * nl[q] + 5 + len[q];
                    sum += sr[q];
                }
        }
        cpos[cells + lane] = run;
        const uint64_t body = EXACT ? P.recpos[r] : r * P.fixed + sum - P.sumbase;
        const FilePos fp = file_pos(r, P.maxcount, P.rows);
        const uint64_t start = body + fp.framing;
        const bool op = fp.opens, cl = fp.closes;
        rstart[lane] = start + (op ? 7 : 0);
        rflag[lane] = (op ? 1 : 0) | (cl ? 2 : 0);
        if (lane == 0) misc[0] = start;
        if (lane == nrec - 1) misc[1] = start + (op ? 7 : 0) + run + 8 + (cl ? 8 : 0);
        if (op && !misc[2]) P.file_off[fp.file] = start;
    }
    __syncthreads();
    if (misc[2]) {
        if (lane == 0) atomicOr(P.status, FS_BAD_OFFSETS);
        return;
    }
    const uint64_t blk_lo = misc[0];
    const uint64_t blk_hi = misc[1] < P.total ? misc[1] : P.total;
    if (blk_lo >= blk_hi) return;

    uint32_t sawcut = 0;
    auto nul_in = [&](uint32_t v, uint32_t m) { // a NUL among the low m bytes of v
        uint32_t z = (v - 0x01010101u) & ~v & 0x80808080u;
        if (m < 4) z &= (1u << (8 * m)) - 1;
        sawcut |= z;
    };
    // the window [wlo, wlo + STAGE) to HBM: whole 16-byte groups, bytes at the block's two ends
    auto copy_out = [&](uint64_t wlo) {
        const uint64_t whi = wlo + STAGE;
        const uint64_t lo = wlo > blk_lo ? wlo : blk_lo;
        const uint64_t hi = whi < blk_hi ? whi : blk_hi;
        for (uint32_t j = lane; j * 16 < STAGE; j += FMT_BLOCK) {
            const uint64_t g = wlo + 16ull * j;
            if (g >= hi) break;
            if (FSKIP(2)) continue;
            if (g >= lo && g + 16 <= hi) {
                *(fu32x4 *)(P.out + g) = *(const fu32x4 *)(stage + 16 * j);
            } else {
                for (uint32_t b = 0; b < 16; ++b)
                    if (g + b >= lo && g + b < hi) P.out[g + b] = stage[16 * j + b];
            }
        }
    };
    // the pieces of unit u: position and length of the leading text (first field only), the value
    // and the text after it
    struct Piece {
        uint64_t vpos;
        uint32_t len, c, pre_off, pre_n, join_off, join_n;
    };
    auto piece_of = [&](uint32_t u) {
        Piece q;
        const uint32_t i = u & (R - 1);
        q.c = u >> Rshift;
        const uint32_t nl = fld[q.c].name_len, fl = rflag[i];
        q.len = cpos[u + R] - cpos[u] - (2 * nl + 5);
        q.vpos = rstart[i] + cpos[u] + nl + 2;
        q.pre_n = q.c ? 0 : nl + 2 + ((fl & 1) ? 7 : 0);
        q.pre_off = (fl & 1) ? P.pre_open_off : P.pre_plain_off;
        q.join_off = fld[q.c].text_off;
        q.join_n = fld[q.c].join_len + ((fl & 2) && q.c + 1 == ncols ? 8 : 0);
        return q;
    };

    const uint64_t w0 = blk_lo & ~15ull;
    if (blk_hi - w0 <= STAGE) {
        // ---- phases 3 and 4, the whole block in one window.  A piece is streamed into aligned
        // words of the window: bytes wait in `acc` until four are there; only the piece's first
        // and last word, shared with the neighbouring pieces, are written byte by byte.
        struct Stream {
            uint64_t acc; // pending bytes, low ones first
            uint32_t f;   // how many (0..3 between pushes)
            uint32_t dp;  // window offset of the next word
            bool head;    // the next word is the piece's first and starts before the piece
            uint32_t al;  // bytes of the first word that belong to the piece before
        };
        auto emit = [&](Stream &st) {
            const uint32_t x = (uint32_t)st.acc;
            if (st.head) {
                if (st.al <= 1) stage[st.dp + 1] = (uint8_t)(x >> 8);
                if (st.al <= 2) stage[st.dp + 2] = (uint8_t)(x >> 16);
                stage[st.dp + 3] = (uint8_t)(x >> 24);
                st.head = false;
            } else {
                *(uint32_t *)(stage + st.dp) = x;
            }
            st.acc >>= 32;
            st.dp += 4;
        };
        auto push4 = [&](Stream &st, uint32_t w) {
            st.acc |= (uint64_t)w << (8 * st.f);
            emit(st);
        };
        auto push = [&](Stream &st, uint32_t w, uint32_t n) { // the low n bytes of w, the rest zero
            st.acc |= (uint64_t)w << (8 * st.f);
            st.f += n;
            if (st.f >= 4) {
                emit(st);
                st.f -= 4;
            }
        };
        // text of the blob: 16-aligned, zero padded to the next multiple of 16
        auto push_text = [&](Stream &st, uint32_t off, uint32_t n) {
            for (uint32_t k = 0; k < n; k += 16) {
                const fu32x4 t = *(const fu32x4 *)(lds + off + k);
                const uint32_t m = n - k;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (m >= 4u * j + 4)
                        push4(st, t[j]);
                    else if (m > 4u * j)
                        push(st, t[j], m - 4 * j);
                }
            }
        };
        auto unit = [&](uint32_t u, const uint32_t *pre, uint32_t klen, uint32_t ka) {
            const uint32_t i = u & (R - 1), c = u >> Rshift;
            const fu32x4 F = *(const fu32x4 *)(lds + 16 * c); // text_off, name_len, fieldlen, join_len
            const uint32_t nl = F[1], fl = rflag[i];
            const uint32_t len = pre ? klen : cpos[u + R] - cpos[u] - (2 * nl + 5);
            const uint32_t a = pre ? ka : srcs[u];
            const uint32_t s = (uint32_t)(rstart[i] - w0) + cpos[u]; // where "<name>" starts
            const uint32_t ps = c ? s + nl + 2 : s - ((fl & 1) ? 7 : 0);
            Stream st;
            st.acc = 0;
            st.al = st.f = ps & 3;
            st.dp = ps & ~3u;
            st.head = st.al != 0;
            if (c == 0) push_text(st, (fl & 1) ? P.pre_open_off : P.pre_plain_off, nl + 2 + ((fl & 1) ? 7 : 0));
            const uint64_t gbase = (uint64_t)(bytp[c] + a);
            const uint32_t sh = (uint32_t)gbase & 3;
            if (pre && sh + len <= 4 * PW) {
#pragma unroll
                for (int j = 0; j < PW; ++j)
                    if ((uint32_t)j * 4 < len) {
                        uint32_t v = __builtin_amdgcn_alignbyte(j + 1 < PW ? pre[j + 1] : 0, pre[j], sh);
                        const uint32_t m = len - 4 * j;
                        if (m >= 4) {
                            if (!EXACT) nul_in(v, 4);
                            push4(st, v);
                        } else {
                            v &= (1u << (8 * m)) - 1;
                            if (!EXACT) nul_in(v | (0x01010101u << (8 * m)), 4);
                            push(st, v, m);
                        }
                    }
            } else if (len) {
                fg_u32 *w = (fg_u32 *)(gbase - sh);
                const uint32_t nwords = (sh + len + 3) >> 2;
                uint32_t lo = w[0];
                for (uint32_t j = 0; j * 4 < len; ++j) {
                    const uint32_t hi = j + 1 < nwords ? w[j + 1] : 0;
                    uint32_t v = __builtin_amdgcn_alignbyte(hi, lo, sh);
                    const uint32_t m = len - 4 * j;
                    if (m >= 4) {
                        if (!EXACT) nul_in(v, 4);
                        push4(st, v);
                    } else {
                        v &= (1u << (8 * m)) - 1;
                        if (!EXACT) nul_in(v | (0x01010101u << (8 * m)), 4);
                        push(st, v, m);
                    }
                    lo = hi;
                }
            }
            push_text(st, F[0], F[3] + ((fl & 2) && c + 1 == ncols ? 8 : 0));
            // what is left belongs to the piece's last word
            const uint32_t x = (uint32_t)st.acc;
            if (st.f > 0) stage[st.dp] = (uint8_t)x;
            if (st.f > 1) stage[st.dp + 1] = (uint8_t)(x >> 8);
            if (st.f > 2) stage[st.dp + 2] = (uint8_t)(x >> 16);
        };
#pragma unroll
        for (int k = 0; k < UPL; ++k) {
            const uint32_t u = lane + k * FMT_BLOCK;
            if (u < cells && (u & (R - 1)) < nrec && !FSKIP(4)) unit(u, uw[k], ulen[k], ua[k]);
        }
        for (uint32_t u = lane + UPL * FMT_BLOCK; u < cells; u += FMT_BLOCK)
            if ((u & (R - 1)) < nrec) unit(u, nullptr, 0, 0);
        __syncthreads();
        copy_out(w0);
    } else {
        // ---- phases 3 and 4 window by window: every piece clipped to the window, byte by byte
        for (uint64_t wlo = w0; wlo < blk_hi; wlo += STAGE) {
            const uint64_t whi = wlo + STAGE;
            // bytes [k0, k1) of n bytes at text position p that fall into this window
            auto clip = [&](uint64_t p, uint32_t n, uint32_t &k0, uint32_t &k1) {
                k0 = p >= wlo ? 0 : (wlo - p < n ? (uint32_t)(wlo - p) : n);
                k1 = p + n <= whi ? n : (p >= whi ? 0 : (uint32_t)(whi - p));
            };
            auto put_lds = [&](uint64_t p, const uint8_t *src, uint32_t n) {
                uint32_t k0, k1;
                clip(p, n, k0, k1);
                const uint32_t s = (uint32_t)(p - wlo); // exact modulo 2^32 for the bytes in range
                for (uint32_t k = k0; k < k1; ++k) stage[s + k] = src[k];
            };
            for (uint32_t u = lane; u < cells; u += FMT_BLOCK) {
                if ((u & (R - 1)) >= nrec) continue;
                const Piece q = piece_of(u);
                if (q.vpos + q.len + q.join_n <= wlo || q.vpos - q.pre_n >= whi) continue;
                if (q.pre_n) put_lds(q.vpos - q.pre_n, lds + q.pre_off, q.pre_n);
                put_lds(q.vpos + q.len, lds + q.join_off, q.join_n);
                uint32_t k0, k1;
                clip(q.vpos, q.len, k0, k1);
                if (k0 >= k1) continue;
                const uint64_t ga = (uint64_t)(bytp[q.c] + srcs[u]) + k0;
                const uint32_t sh = (uint32_t)ga & 3;
                fg_u32 *w = (fg_u32 *)(ga - sh);
                const uint32_t n = k1 - k0;
                const uint32_t nwords = (sh + n + 3) >> 2;
                uint32_t s = (uint32_t)(q.vpos - wlo) + k0;
                uint32_t lo = w[0];
                for (uint32_t j = 0; j * 4 < n; ++j) {
                    const uint32_t hi = j + 1 < nwords ? w[j + 1] : 0;
                    const uint32_t v = __builtin_amdgcn_alignbyte(hi, lo, sh);
                    const uint32_t m = n - j * 4 < 4 ? n - j * 4 : 4;
                    if (!EXACT) nul_in(v, m);
                    stage[s] = (uint8_t)v;
                    if (m > 1) stage[s + 1] = (uint8_t)(v >> 8);
                    if (m > 2) stage[s + 2] = (uint8_t)(v >> 16);
                    if (m > 3) stage[s + 3] = (uint8_t)(v >> 24);
                    s += 4;
                    lo = hi;
                }
            }
            __syncthreads();
            copy_out(wlo);
            __syncthreads();
        }
    }
    if (!EXACT && sawcut) atomicOr(P.status, FS_NEED_EXACT);
}

// info[0] = sum of the columns' last offsets, info[1] = sum of their first offsets,
// info[2] = 1 when a column's last offset passes the end of its arena (nbytes may be null)
__global__ __launch_bounds__(256) void format_prep_kernel(const uint32_t *const *off, const uint64_t *nbytes,
                                                          uint32_t ncols, uint64_t rows, uint64_t *info) {
    __shared__ uint64_t first[256], last[256];
    __shared__ uint32_t bad;
    const uint32_t c = threadIdx.x;
    if (c == 0) bad = 0;
    __syncthreads();
    first[c] = last[c] = 0;
    if (c < ncols) {
        first[c] = off[c][0];
        last[c] = off[c][rows];
        if (last[c] < first[c] || (nbytes && last[c] > nbytes[c])) bad = 1;
    }
    __syncthreads();
    if (c == 0) {
        uint64_t a = 0, b = 0;
        for (uint32_t k = 0; k < ncols; ++k) {
            a += last[k];
            b += first[k];
        }
        info[0] = a;
        info[1] = b;
        info[2] = bad;
    }
}

// exact mode, first step: the length "%s" prints of every value, and every record's length
__global__ __launch_bounds__(256) void format_measure_kernel(const uint32_t *const *off, const uint8_t *const *bytes,
                                                             const uint8_t *blob, uint32_t ncols, uint64_t rows,
                                                             uint32_t fixed, uint32_t *elen, uint64_t *reclen,
                                                             uint32_t *status) {
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const FormatField *fld = (const FormatField *)blob;
    uint64_t run = fixed;
    for (uint32_t c = 0; c < ncols; ++c) {
        const uint32_t a = off[c][r], b = off[c][r + 1];
        uint32_t n = b - a;
        if (b < a) {
            atomicOr(status, FS_BAD_OFFSETS);
            n = 0;
        }
        const uint32_t fl = fld[c].fieldlen;
        if (fl && n > fl) n = fl;
        const uint8_t *p = bytes[c] + a;
        for (uint32_t k = 0; k < n; ++k)
            if (p[k] == 0) {
                n = k;
                break;
            }
        elen[(uint64_t)c * rows + r] = n;
        run += n;
    }
    reclen[r] = run;
}

// ---- exclusive scan of u64 (record lengths -> record starts), three small kernels -------------
constexpr uint32_t SCAN_ITEMS = 8; // per lane; 2048 per block

__device__ __forceinline__ uint64_t block_exclusive(uint64_t v, uint64_t *wsum, uint64_t &total) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t x = v;
    for (int d = 1; d < 64; d <<= 1) {
        uint64_t y = __shfl_up(x, d, 64);
        if ((int)lane >= d) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    uint64_t base = 0, tot = 0;
    for (uint32_t w = 0; w < blockDim.x / 64; ++w) {
        if (w < wave) base += wsum[w];
        tot += wsum[w];
    }
    __syncthreads();
    total = tot;
    return base + x - v;
}

__global__ __launch_bounds__(256) void scan64_sums(const uint64_t *in, uint64_t n, uint64_t *part) {
    __shared__ uint64_t wsum[4];
    const uint64_t b0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * SCAN_ITEMS;
    uint64_t s = 0;
    for (uint32_t k = 0; k < SCAN_ITEMS; ++k)
        if (b0 + k < n) s += in[b0 + k];
    uint64_t total;
    block_exclusive(s, wsum, total);
    if (threadIdx.x == 0) part[blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void scan64_parts(uint64_t *part, uint32_t nparts) {
    __shared__ uint64_t wsum[4];
    uint64_t carry = 0;
    for (uint32_t base = 0; base < nparts; base += 256) {
        const uint32_t i = base + threadIdx.x;
        const uint64_t v = i < nparts ? part[i] : 0;
        uint64_t total;
        const uint64_t e = block_exclusive(v, wsum, total);
        if (i < nparts) part[i] = carry + e;
        carry += total;
    }
}

__global__ __launch_bounds__(256) void scan64_apply(const uint64_t *in, uint64_t n, const uint64_t *part,
                                                    uint64_t *out) {
    __shared__ uint64_t wsum[4];
    const uint64_t b0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * SCAN_ITEMS;
    uint64_t v[SCAN_ITEMS], s = 0;
    for (uint32_t k = 0; k < SCAN_ITEMS; ++k) {
        v[k] = b0 + k < n ? in[b0 + k] : 0;
        s += v[k];
    }
    uint64_t total;
    uint64_t e = part[blockIdx.x] + block_exclusive(s, wsum, total);
    for (uint32_t k = 0; k < SCAN_ITEMS; ++k) {
        if (b0 + k < n) out[b0 + k] = e;
        e += v[k];
        if (b0 + k + 1 == n) out[n] = e;
    }
}

// ---- launchers ----------------------------------------------------------------------------------

hipError_t launch_format_prep(const uint32_t *const *off, const uint64_t *nbytes, uint32_t ncols, uint64_t rows,
                              uint64_t *info, hipStream_t stream) {
    hipLaunchKernelGGL(format_prep_kernel, dim3(1), dim3(256), 0, stream, off, nbytes, ncols, rows, info);
    return hipGetLastError();
}

hipError_t launch_format(bool exact, const FormatParams &P, hipStream_t stream) {
    const uint64_t nblocks = (P.rows + P.R - 1) >> P.Rshift;
    if (nblocks == 0 || nblocks > 0x7fffffffull) return hipErrorInvalidValue;
    if (exact)
        hipLaunchKernelGGL(format_kernel<true>, dim3((uint32_t)nblocks), dim3(FMT_BLOCK), P.lds_total, stream, P);
    else
        hipLaunchKernelGGL(format_kernel<false>, dim3((uint32_t)nblocks), dim3(FMT_BLOCK), P.lds_total, stream, P);
    return hipGetLastError();
}

uint32_t format_scan_parts(uint64_t rows) { return (uint32_t)((rows + 256 * SCAN_ITEMS - 1) / (256 * SCAN_ITEMS)); }

// reclen[rows] -> recpos[rows+1]; part needs format_scan_parts(rows) entries
hipError_t launch_format_measure(const FormatParams &P, uint32_t *elen, uint64_t *reclen, uint64_t *part,
                                 uint64_t *recpos, hipStream_t stream) {
    const uint64_t nb = (P.rows + 255) / 256;
    if (nb == 0 || nb > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(format_measure_kernel, dim3((uint32_t)nb), dim3(256), 0, stream, P.off, P.bytes, P.blob,
                       P.ncols, P.rows, P.fixed, elen, reclen, P.status);
    const uint32_t np = format_scan_parts(P.rows);
    hipLaunchKernelGGL(scan64_sums, dim3(np), dim3(256), 0, stream, reclen, P.rows, part);
    hipLaunchKernelGGL(scan64_parts, dim3(1), dim3(256), 0, stream, part, np);
    hipLaunchKernelGGL(scan64_apply, dim3(np), dim3(256), 0, stream, reclen, P.rows, part, recpos);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// timetostr in bulk (reference public/_public.cpp:631-722): time_t values -> the text of one of
// the reference's thirteen formats, one cell per row in the layout the formatter and the
// parser's batches use (u32 end offsets + bytes; the cells all have the format's width).
// localtime_r is taken for a zone without daylight saving (tz_offset seconds east of UTC).
// Every format is a selection of bytes of "yyyy-mm-dd hh:mi:ss": a block writes that text for
// its rows into LDS, then its piece of the column (TT_BLOCK x width bytes, a multiple of four)
// goes out in dword stores.
constexpr int64_t TT_MIN = -62167219200ll;   // 0000-01-01 00:00:00
constexpr int64_t TT_MAX = 253402300799ll;   // 9999-12-31 23:59:59
__device__ __forceinline__ uint32_t two_digits(uint32_t v) {  // "ab" of v = 10a + b as two bytes, first byte low
    const uint32_t a = (v * 205u) >> 11;  // v / 10 for v < 100
    return (0x30u + a) | ((0x30u + v - 10u * a) << 8);
}
__global__ __launch_bounds__(TT_BLOCK) void timetostr_kernel(const int64_t *tv, uint64_t rows, const TimeFmt F,
                                                             int32_t tz_offset, uint32_t *off, uint8_t *bytes,
                                                             uint32_t aligned, uint32_t *status) {
    __shared__ __align__(4) uint8_t full[TT_BLOCK * 20];
    __shared__ uint8_t sidx[20];
    const uint32_t tid = threadIdx.x;
    const uint64_t row0 = (uint64_t)blockIdx.x * TT_BLOCK;
    const uint64_t row = row0 + tid;
    const uint32_t width = F.width;
    if (tid < 20) sidx[tid] = F.idx[tid];
    if (row < rows) {
        int64_t local = tv[row] + (int64_t)tz_offset;
        if (local < TT_MIN || local > TT_MAX) {  // (the year would not print as four digits)
            atomicOr(status, 1u);
            local = 0;
        }
        int64_t days = local / 86400;
        int32_t sod = (int32_t)(local - days * 86400);
        if (sod < 0) {
            sod += 86400;
            --days;
        }
        // civil date of a day count (days since 1970-01-01), proleptic Gregorian calendar
        // days since 0000-03-01 of one 400-year cycle earlier (January and February of the year
        // 0000 lie in front of that day): >= 0 in the accepted range
        const int64_t z = days + 719468 + 146097;
        const uint32_t era = (uint32_t)(z / 146097);
        const uint32_t doe = (uint32_t)(z - (int64_t)era * 146097);
        const uint32_t yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365;
        const uint32_t doy = doe - (365 * yoe + yoe / 4 - yoe / 100);
        const uint32_t mp = (5 * doy + 2) / 153;
        const uint32_t mday = doy - (153 * mp + 2) / 5 + 1;
        const uint32_t mon = mp < 10 ? mp + 3 : mp - 9;
        const uint32_t year = yoe + era * 400 + (mon <= 2 ? 1u : 0u) - 400u;
        const uint32_t hour = (uint32_t)sod / 3600, rem = (uint32_t)sod - hour * 3600;
        const uint32_t mi = rem / 60, sec = rem - mi * 60;
        const uint32_t yh = year / 100, yl = year - yh * 100;
        // "yyyy-mm-dd hh:mi:ss" + one unused byte, as five words
        uint32_t *w = (uint32_t *)(full + tid * 20);
        const uint32_t mm = two_digits(mon), dd = two_digits(mday), hh = two_digits(hour), mn = two_digits(mi),
                       ss = two_digits(sec);
        w[0] = two_digits(yh) | (two_digits(yl) << 16);
        w[1] = 0x2Du | (mm << 8) | (0x2Du << 24);
        w[2] = dd | (0x20u << 16) | ((hh & 0xFFu) << 24);
        w[3] = (hh >> 8) | (0x3Au << 8) | (mn << 16);
        w[4] = 0x3Au | (ss << 8);
        off[row + 1] = (uint32_t)((row + 1) * width);
    }
    if (row == 0) off[0] = 0;
    __syncthreads();
    const uint32_t nrows = rows - row0 < (uint64_t)TT_BLOCK ? (uint32_t)(rows - row0) : (uint32_t)TT_BLOCK;
    const uint32_t nbytes = nrows * width;
    uint8_t *dst = bytes + row0 * width;
    auto out_byte = [&](uint32_t o) -> uint32_t {
        const uint32_t r = o / width;
        return full[r * 20 + sidx[o - r * width]];
    };
    const uint32_t nwords = aligned ? nbytes / 4 : 0;
    for (uint32_t i = tid; i < nwords; i += TT_BLOCK) {
        const uint32_t o = i * 4;
        ((uint32_t *)dst)[i] = out_byte(o) | (out_byte(o + 1) << 8) | (out_byte(o + 2) << 16) | (out_byte(o + 3) << 24);
    }
    for (uint32_t o = nwords * 4 + tid; o < nbytes; o += TT_BLOCK) dst[o] = (uint8_t)out_byte(o);
}

hipError_t launch_timetostr(const int64_t *tv, uint64_t rows, const TimeFmt &F, int32_t tz_offset, uint32_t *off,
                            uint8_t *bytes, uint32_t *status, hipStream_t stream) {
    const uint64_t nb = (rows + TT_BLOCK - 1) / TT_BLOCK;
    if (nb == 0 || nb > 0x7fffffffull || F.width == 0 || F.width > 19) return hipErrorInvalidValue;
    const uint32_t aligned = ((uintptr_t)bytes & 3u) == 0;
    hipLaunchKernelGGL(timetostr_kernel, dim3((uint32_t)nb), dim3(TT_BLOCK), 0, stream, tv, rows, F, tz_offset, off, bytes,
                       aligned, status);
    return hipGetLastError();
}

} // namespace mdc
