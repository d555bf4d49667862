// mdc_kernels.hip -- gfx950 kernels of the bulk record parser.
//
// One workgroup (256 threads = 4 wave64) per TILE input bytes.  A tile owns the records
// whose terminating '\n' lies inside it.  Per tile:
//
//   0. stage [tile_start-LB, tile_end) into LDS with 16-byte loads; LB grows backwards until
//      the terminator in front of the tile is found (look-behind, normally one 512-byte step)
//   1. every lane classifies its 16-byte chunks with SWAR compares: record terminators
//      ('\n' preceded by endbz, cifile::readline, reference public/_public.cpp:866-892);
//      a block-wide prefix sum of the per-chunk counts ranks every byte by record
//   2. every '<' is looked up in a hash table of the schema's tag texts held in LDS; a hit
//      does an LDS atomicMin on first_pos[pattern][record]: the independent first-open /
//      first-close search of getxmlbuffer (public/_public.cpp:466,469).  Delimited mode marks
//      separator candidates instead (ccmdstr::splittocmd, public/_public.cpp:283-292)
//   3. one lane per record resolves the columns in schema order (clip, negative length,
//      carry-forward: tools/xmltodb.cpp:492-522) into (start,len,filter level) descriptors
//   4. one lane per cell filters (picknumber, public/_public.cpp:178-204), measures and
//      converts (mdc_numparse.h) its cell
//   5. per-column prefix sums give every cell its offset; cells are compacted into an LDS
//      staging area that mirrors the destination's 16-byte phase and leave with 16-byte stores
//
// The kernel runs twice per parse: COUNT mode stops after step 4 and reports per-tile byte
// and row counts; a scan turns them into tile base offsets; WRITE mode repeats the (fully
// deterministic) analysis and stores the columns.
#include <hip/hip_runtime.h>

#include "mdc_device.h"
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
// gathers the four 0x80 flags of m into bits 0..3
__device__ __forceinline__ uint32_t pack4(uint32_t m) { return (((m >> 7) * 0x00204081u) >> 21) & 0xFu; }

__device__ __forceinline__ uint32_t mask16_eq(const uint4 &v, uint32_t pat4) {
    return pack4(bytes_eq(v.x, pat4)) | (pack4(bytes_eq(v.y, pat4)) << 4) | (pack4(bytes_eq(v.z, pat4)) << 8) |
           (pack4(bytes_eq(v.w, pat4)) << 12);
}

__device__ __forceinline__ uint32_t rep4(uint32_t b) { return b * 0x01010101u; }

// inclusive prefix sum across the 64 lanes of a wave
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

struct Ctx {
    uint8_t *sdata;
    const SchemaHdr *H;
    const DevCol *cols;
    const HtEntry *ht;
    const uint8_t *gdata;
    int64_t base_pos;  // global position of LDS index 0 (may be negative)
};

// byte at LDS index li, reading global memory below the staged region
__device__ __forceinline__ uint32_t byte_at(const Ctx &cx, int li, int loaded_lo) {
    return li >= loaded_lo ? cx.sdata[li] : cx.gdata[cx.base_pos + li];
}

// is the '\n' at LDS index p_li a record terminator, i.e. preceded by endbz?
__device__ __forceinline__ bool is_terminator(const Ctx &cx, int p_li, int loaded_lo) {
    const int el = (int)cx.H->endbz_len;
    if (el == 0) return true;
    if (cx.base_pos + p_li - el < 0) return false;
    for (int k = 0; k < el; ++k)
        if (byte_at(cx, p_li - el + k, loaded_lo) != cx.H->endbz[k]) return false;
    return true;
}

// Looks up the tag text that follows the '<' at LDS index q and records a first-position
// event for every (column, open/close) pattern with that text.
__device__ __forceinline__ void match_tag(const Ctx &cx, uint32_t q, uint32_t *evt, uint32_t rl, uint32_t RB) {
    const uint32_t *s32 = (const uint32_t *)cx.sdata;
    const uint32_t a = q + 1;
    const uint32_t wi = a >> 2, sh = a & 3;
    const uint32_t w0 = s32[wi], w1 = s32[wi + 1], w2 = s32[wi + 2];
    uint32_t lo = __builtin_amdgcn_alignbyte(w1, w0, sh);
    uint32_t hi = __builtin_amdgcn_alignbyte(w2, w1, sh);
    // position of the first '>' among the eight bytes
    const uint32_t glo = bytes_eq(lo, 0x3E3E3E3Eu), ghi = bytes_eq(hi, 0x3E3E3E3Eu);
    if (glo) {
        const uint32_t g = (uint32_t)__builtin_ctz(glo) >> 3;  // 0..3
        lo &= 0xFFFFFFFFu >> (8 * (3 - g));
        hi = 0;
    } else if (ghi) {
        const uint32_t g = (uint32_t)__builtin_ctz(ghi) >> 3;
        hi &= 0xFFFFFFFFu >> (8 * (3 - g));
    }
    uint32_t slot = ht_hash(lo, hi) & cx.H->ht_mask;
    for (;;) {
        const HtEntry e = cx.ht[slot];
        if (!(e.meta >> 31)) break;
        if (e.klo == lo && e.khi == hi) {
            const uint32_t tlen = e.meta & 0xFFu, kind = (e.meta >> 8) & 1u, col = (e.meta >> 16) & 0x7FFFu;
            bool ok = true;
            if (tlen >= 8) {  // texts of 8+ bytes share their key with longer ones: compare the tail
                const uint8_t *nm = cx.cols[col].name;
                for (uint32_t k = 8; k < tlen; ++k)
                    if (cx.sdata[a + k] != nm[k - kind]) {
                        ok = false;
                        break;
                    }
                if (ok && cx.sdata[a + tlen] != '>') ok = false;
            }
            if (ok) atomicMin(&evt[(2 * col + kind) * RB + rl], q);
        }
        slot = (slot + 1) & cx.H->ht_mask;
    }
}

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

// ---------------------------------------------------------------------------------------
// indices of the block-wide scalars kept in LDS
enum { MS_LAST = 0, MS_NREC = 1, MS_HASNUL = 2, MS_STAGED = 3, MS_WTOT = 4 /* 4 words */, MS_COUNT = 16 };

template <bool WRITE>
__global__ __launch_bounds__(BLOCK) void tile_kernel(const TileParams P) {
    extern __shared__ __align__(16) uint8_t smem[];
    const LdsLayout &L = P.L;
    uint8_t *sdata = smem + L.data;
    uint32_t *evt = (uint32_t *)(smem + L.evt);
    uint32_t *olen = evt;  // evt is dead once the descriptors exist
    uint32_t *dstart = (uint32_t *)(smem + L.dstart);
    uint32_t *dlen = (uint32_t *)(smem + L.dlen);
    uint16_t *cterm = (uint16_t *)(smem + L.cterm);
    uint16_t *cpre = (uint16_t *)(smem + L.cpre);
    uint32_t *recend = (uint32_t *)(smem + L.recend);
    uint32_t *vmask = (uint32_t *)(smem + L.vmask);
    uint32_t *fmask = (uint32_t *)(smem + L.fmask);
    uint8_t *stage = smem + L.stage;
    volatile uint32_t *ms = (volatile uint32_t *)(smem + L.misc);
    uint32_t *msa = (uint32_t *)(smem + L.misc);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;

    // ---- schema blob -> LDS
    {
        const uint4 *g = (const uint4 *)P.schema;
        uint4 *s = (uint4 *)(smem + L.schema);
        for (uint32_t i = tid; i < P.schema_bytes / 16; i += BLOCK) s[i] = g[i];
        if (tid < MS_COUNT) msa[tid] = 0;
        if (tid == 0) msa[MS_LAST] = NONE;
    }
    __syncthreads();
    Ctx cx;
    cx.sdata = sdata;
    cx.H = (const SchemaHdr *)(smem + L.schema);
    cx.cols = (const DevCol *)(cx.H + 1);
    const uint32_t ncols = cx.H->ncols, npat = cx.H->npat, W = cx.H->words;
    cx.ht = (const HtEntry *)((const uint8_t *)cx.cols + ((ncols * sizeof(DevCol) + 15) & ~15u));
    cx.gdata = P.data;
    const bool csv = cx.H->mode == MODE_CSV;
    const bool carry = cx.H->carry != 0;
    uint32_t *coltot = (uint32_t *)(smem + L.coltot);  // totals of the current batch
    uint32_t *tiletot = coltot + (ncols + 1);          // totals of the batches done so far
    uint32_t *stagebase = tiletot + (ncols + 1);

    const int64_t n = (int64_t)P.n;
    const int64_t tile_start = (int64_t)blockIdx.x * TILE;
    const int64_t tile_end = (tile_start + TILE < n) ? tile_start + TILE : n;
    cx.base_pos = tile_start - LBMAX;
    const int ts_li = LBMAX;
    const int te_li = LBMAX + (int)(tile_end - tile_start);
    const int li_min = cx.base_pos < 0 ? (int)(-cx.base_pos) : 0;

    auto load_chunks = [&](int jlo, int jhi) {
        for (int j = jlo + tid; j < jhi; j += BLOCK) {
            const int64_t pos = cx.base_pos + (int64_t)j * CHUNK;
            uint4 v;
            if (pos + CHUNK <= n) {
                v = *(const uint4 *)(P.data + pos);
            } else {
                uint32_t w[4] = {0, 0, 0, 0};
                for (int k = 0; k < CHUNK; ++k)
                    if (pos + k < n) w[k >> 2] |= (uint32_t)P.data[pos + k] << (8 * (k & 3));
                v = make_uint4(w[0], w[1], w[2], w[3]);
            }
            ((uint4 *)sdata)[j] = v;
        }
    };

    if (tid < (int)ncols + 1) tiletot[tid] = 0;
    for (uint32_t c = tid; c < ncols + 1; c += BLOCK) tiletot[c] = 0;

    // ---- 0. stage the tile, then look behind for the start of the first owned record
    const int j1 = (te_li + CHUNK - 1) / CHUNK;
    load_chunks(ts_li / CHUNK, j1);
    int loaded_lo = ts_li;
    int s0_li = ts_li;
    if (tile_start > 0) {
        int win = LB_FIRST;
        for (;;) {
            int new_lo = loaded_lo - win;
            if (new_lo < li_min) new_lo = li_min;
            load_chunks(new_lo / CHUNK, loaded_lo / CHUNK);
            __syncthreads();
            for (int j = new_lo / CHUNK + tid; j < loaded_lo / CHUNK; j += BLOCK) {
                const uint4 v = ((const uint4 *)sdata)[j];
                uint32_t nl = mask16_eq(v, 0x0A0A0A0Au);
                while (nl) {
                    const int b = __builtin_ctz(nl);
                    nl &= nl - 1;
                    const int p = j * CHUNK + b;
                    if (is_terminator(cx, p, new_lo)) atomicMax((int *)&msa[MS_LAST], p);
                }
            }
            __syncthreads();
            const int last = (int)ms[MS_LAST];
            loaded_lo = new_lo;
            if (last >= 0) {
                s0_li = last + 1;
                break;
            }
            if (cx.base_pos + new_lo == 0) {  // reached the start of the input
                s0_li = new_lo;
                break;
            }
            if (new_lo == 0) {  // record longer than the look-behind window
                if (tid == 0) atomicOr(P.status, ST_RECORD_TOO_LONG);
                s0_li = 0;
                break;
            }
            win = LB_STEP;
        }
    } else {
        __syncthreads();
    }
    const int j0 = s0_li / CHUNK;

    // ---- 1. terminators per chunk, NUL presence, rank prefix
    {
        bool nul = false;
        for (int j = tid; j < NCH; j += BLOCK) {
            uint32_t tm = 0;
            if (j >= j0 && j < j1) {
                const uint4 v = ((const uint4 *)sdata)[j];
                nul |= mask16_eq(v, 0) != 0;
                uint32_t nl = mask16_eq(v, 0x0A0A0A0Au);
                while (nl) {
                    const int b = __builtin_ctz(nl);
                    nl &= nl - 1;
                    const int p = j * CHUNK + b;
                    if (p >= ts_li && p < te_li && is_terminator(cx, p, loaded_lo)) tm |= 1u << b;
                }
            }
            cterm[j] = (uint16_t)tm;
        }
        if (nul) msa[MS_HASNUL] = 1;
    }
    __syncthreads();
    {
        uint32_t cnt[CPT], sum = 0;
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            cnt[k] = __builtin_popcount(cterm[tid * CPT + k]);
            sum += cnt[k];
        }
        const uint32_t incl = wave_incl_scan(sum);
        if (lane == 63) msa[MS_WTOT + wave] = incl;
        __syncthreads();
        uint32_t base = incl - sum;
        for (int w = 0; w < wave; ++w) base += ms[MS_WTOT + w];
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            cpre[tid * CPT + k] = (uint16_t)base;
            base += cnt[k];
        }
        if (tid == BLOCK - 1) msa[MS_NREC] = base;
    }
    __syncthreads();
    const uint32_t nrec = ms[MS_NREC];
    const bool has_nul = ms[MS_HASNUL] != 0;
    const uint32_t RB = L.RB;
    const uint64_t *tb = WRITE ? P.tile_base + (uint64_t)blockIdx.x * (ncols + 1) : nullptr;
    const uint64_t row0 = WRITE ? tb[ncols] : 0;
    if (WRITE && blockIdx.x == 0) {
        if (tid == 0 && P.rec_off) P.rec_off[0] = 0;
        for (uint32_t c = tid; c < ncols; c += BLOCK) P.off[c][0] = 0;
    }
    if (tid == 0) recend[0] = (uint32_t)(s0_li - 1);
    const uint32_t seplen = cx.H->sep_len;

    // ---- batches of RB records
    for (uint32_t rb0 = 0; rb0 < nrec; rb0 += RB) {
        const uint32_t RBn = (nrec - rb0 < RB) ? nrec - rb0 : RB;
        for (uint32_t i = tid; i < RB * npat; i += BLOCK) evt[i] = NONE;
        for (uint32_t i = tid; i < RB * W; i += BLOCK) {
            vmask[i] = 0;
            fmask[i] = 0;
        }
        __syncthreads();

        // ---- 2. events
        for (int j = j0 + tid; j < j1; j += BLOCK) {
            const uint32_t rank0 = cpre[j];
            const uint32_t tm = cterm[j];
            const uint32_t cnt = __builtin_popcount(tm);
            if (rank0 + cnt < rb0 || rank0 >= rb0 + RBn) continue;
            uint32_t t = tm;
            while (t) {
                const uint32_t b = __builtin_ctz(t);
                t &= t - 1;
                const uint32_t r = rank0 + __builtin_popcount(tm & ((1u << b) - 1));
                if (r >= rb0 && r < rb0 + RBn) recend[r - rb0 + 1] = (uint32_t)(j * CHUNK) + b;
            }
            if (!csv) {
                const uint4 v = ((const uint4 *)sdata)[j];
                uint32_t m = mask16_eq(v, 0x3C3C3C3Cu);
                while (m) {
                    const uint32_t b = __builtin_ctz(m);
                    m &= m - 1;
                    const uint32_t q = (uint32_t)(j * CHUNK) + b;
                    if ((int)q < s0_li) continue;
                    const uint32_t r = rank0 + __builtin_popcount(tm & ((1u << b) - 1));
                    if (r < rb0 || r >= rb0 + RBn) continue;
                    match_tag(cx, q, evt, r - rb0, RB);
                }
            }
        }
        __syncthreads();

        // ---- 3. one lane per record: columns in schema order
        if ((uint32_t)tid < RBn) {
            const uint32_t r = tid;
            const uint32_t rs = recend[r] + 1, re = recend[r + 1];
            uint32_t fmw = 0;
            if (!csv) {
                uint32_t cur_s = rs, cur_l = 0, lvl = LVL_CHAR;
                for (uint32_t c = 0; c < ncols; ++c) {
                    const uint32_t o = evt[(2 * c) * RB + r], e = evt[(2 * c + 1) * RB + r];
                    const DevCol &col = cx.cols[c];
                    if (o != NONE && e != NONE) {
                        const uint32_t vs = o + col.namelen + 2;
                        int len = (int)e - (int)vs;
                        if (len < 0)
                            len = (int)(re - vs);  // close tag in front of the open tag: rest of record
                        else if (col.collen > 0 && col.collen < len)
                            len = col.collen;
                        cur_s = vs;
                        cur_l = (uint32_t)len;
                        lvl = LVL_CHAR;
                        fmw |= 1u << (c & 31);
                    } else if (!carry) {
                        cur_l = 0;
                        lvl = LVL_CHAR;
                    }
                    if (col.level > lvl) lvl = col.level;
                    dstart[c * RB + r] = cur_s;
                    dlen[c * RB + r] = cur_l | (lvl << 30);
                    if ((c & 31) == 31 || c + 1 == ncols) {
                        fmask[r * W + (c >> 5)] = fmw;
                        fmw = 0;
                    }
                }
            } else {
                // greedy left-to-right separator search (ccmdstr::splittocmd)
                uint32_t pos = rs;
                bool done = false;
                const uint32_t s0b = cx.H->sep[0];
                for (uint32_t c = 0; c < ncols; ++c) {
                    uint32_t fs = pos, fl = 0;
                    bool have = !done;
                    if (have) {
                        uint32_t p = pos;
                        bool hit = false;
                        while (p + seplen <= re) {
                            if (sdata[p] == s0b) {
                                bool eq = true;
                                for (uint32_t k = 1; k < seplen; ++k)
                                    if (sdata[p + k] != cx.H->sep[k]) {
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
                        if (cx.H->bdelspace) {
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
                        const DevCol &col = cx.cols[c];
                        if (col.collen > 0 && (uint32_t)col.collen < fl) fl = (uint32_t)col.collen;
                        fmw |= 1u << (c & 31);
                    }
                    dstart[c * RB + r] = fs;
                    dlen[c * RB + r] = fl;
                    if ((c & 31) == 31 || c + 1 == ncols) {
                        fmask[r * W + (c >> 5)] = fmw;
                        fmw = 0;
                    }
                }
            }
        }
        __syncthreads();

        // ---- 4. one lane per cell: measure and convert
        const uint32_t items = RBn * ncols;
        for (uint32_t i = tid; i < items; i += BLOCK) {
            const uint32_t c = i / RBn, r = i - c * RBn;
            const uint32_t idx = c * RB + r;
            const uint32_t s = dstart[idx], dl = dlen[idx];
            const uint32_t len = dl & 0x3FFFFFFFu, lvl = dl >> 30;
            const uint32_t typed = cx.cols[c].typed;
            const bool nulcut = carry && !csv;
            uint32_t flen;
            const bool found = (fmask[r * W + (c >> 5)] >> (c & 31)) & 1u;
            const bool convert = typed != T_NONE && (found || (carry && !csv));
            if (!convert && lvl == LVL_CHAR && !(nulcut && has_nul)) {
                flen = len;
            } else {
                uint64_t value = 0;
                bool valid = false, range = false;
                flen = cell_walk<false>(sdata + s, len, lvl, nulcut, convert ? typed : (uint32_t)T_NONE, nullptr,
                                        value, valid, range);
                if (convert) {
                    if (valid) atomicOr(&vmask[r * W + (c >> 5)], 1u << (c & 31));
                    if (range) atomicOr(P.status, ST_NUMERIC_RANGE);
                    if (WRITE && P.val[c]) P.val[c][row0 + rb0 + r] = value;
                }
            }
            if (!convert && WRITE && typed != T_NONE && P.val[c]) P.val[c][row0 + rb0 + r] = 0;
            olen[idx] = flen;
        }
        __syncthreads();

        // ---- 5. per-column exclusive prefix over the batch (one wave per column)
        for (uint32_t c = wave; c < ncols; c += BLOCK / 64) {
            uint32_t v[4], sum = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t r = lane * 4 + k;
                v[k] = r < RBn ? olen[c * RB + r] : 0;
                sum += v[k];
            }
            const uint32_t incl = wave_incl_scan(sum);
            uint32_t run = incl - sum;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t r = lane * 4 + k;
                if (r < RBn) olen[c * RB + r] = run;
                run += v[k];
            }
            if (lane == 63) coltot[c] = incl;
        }
        __syncthreads();

        if (WRITE) {
            // where each column's bytes of this batch go; staging mirrors the 16-byte phase
            if (tid == 0) {
                uint32_t sb = 0;
                bool fits = true;
                for (uint32_t c = 0; c < ncols; ++c) {
                    const uint64_t gb = tb[c] + tiletot[c];
                    const uint32_t a = (uint32_t)(gb & 15);
                    stagebase[c] = sb + a;
                    const uint64_t nxt = (uint64_t)sb + a + coltot[c];
                    if (nxt > L.stage_cap) {
                        fits = false;
                        break;
                    }
                    sb = ((uint32_t)nxt + 15) & ~15u;
                }
                msa[MS_STAGED] = fits ? 1 : 0;
            }
            __syncthreads();
            const bool staged = ms[MS_STAGED] != 0;
            for (uint32_t i = tid; i < items; i += BLOCK) {
                const uint32_t c = i / RBn, r = i - c * RBn;
                const uint32_t idx = c * RB + r;
                const uint32_t ex = olen[idx];
                const uint32_t end = (r + 1 < RBn) ? olen[idx + 1] : coltot[c];
                const uint64_t gb = tb[c] + tiletot[c];
                P.off[c][row0 + rb0 + r + 1] = (uint32_t)(gb + end);
                if (end > ex) {
                    const uint32_t dl = dlen[idx];
                    uint8_t *dst = staged ? stage + stagebase[c] + ex : P.bytes[c] + gb + ex;
                    uint64_t dv;
                    bool b1, b2;
                    cell_walk<true>(sdata + dstart[idx], dl & 0x3FFFFFFFu, dl >> 30, carry && !csv, T_NONE, dst, dv,
                                    b1, b2);
                }
            }
            __syncthreads();
            if (staged) {
                for (uint32_t c = wave; c < ncols; c += BLOCK / 64) {
                    const uint32_t tot = coltot[c];
                    if (!tot) continue;
                    const uint64_t gb = tb[c] + tiletot[c];
                    uint8_t *dst = P.bytes[c] + gb;
                    const uint8_t *src = stage + stagebase[c];
                    uint32_t head = (uint32_t)((16 - (gb & 15)) & 15);
                    if (head > tot) head = tot;
                    if ((uint32_t)lane < head) dst[lane] = src[lane];
                    const uint32_t body = (tot - head) / 16;
                    for (uint32_t u = lane; u < body; u += 64)
                        *(uint4 *)(dst + head + u * 16) = *(const uint4 *)(src + head + u * 16);
                    const uint32_t done = head + body * 16;
                    if ((uint32_t)lane < tot - done) dst[done + lane] = src[done + lane];
                }
            }
            if ((uint32_t)tid < RBn) {
                const uint64_t row = row0 + rb0 + tid;
                if (P.rec_off) P.rec_off[row + 1] = (uint64_t)(cx.base_pos + (int64_t)recend[tid + 1] + 1);
                for (uint32_t w = 0; w < W; ++w) {
                    P.valid[row * W + w] = vmask[tid * W + w];
                    P.found[row * W + w] = fmask[tid * W + w];
                }
            }
        }
        __syncthreads();
        for (uint32_t c = tid; c < ncols; c += BLOCK) tiletot[c] += coltot[c];
        if (tid == 0) recend[0] = recend[RBn];
        __syncthreads();
    }

    if (!WRITE) {
        uint32_t *tc = P.tile_counts + (uint64_t)blockIdx.x * (ncols + 1);
        for (uint32_t c = tid; c < ncols; c += BLOCK) tc[c] = tiletot[c];
        if (tid == 0) tc[ncols] = nrec;
    }
}

// Exclusive prefix over tiles of every counter (one block per counter).
__global__ __launch_bounds__(1024) void scan_tiles_kernel(const uint32_t *counts, uint64_t *base, uint32_t ntiles,
                                                          uint32_t ncnt) {
    __shared__ uint64_t wtot[16];
    __shared__ uint64_t carry_s;
    const uint32_t j = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t t0 = 0; t0 < ntiles; t0 += 1024) {
        const uint32_t t = t0 + tid;
        uint64_t v = t < ntiles ? counts[(uint64_t)t * ncnt + j] : 0;
        uint64_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint64_t u = __shfl_up(incl, d, 64);
            if (lane >= d) incl += u;
        }
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        uint64_t b = carry_s;
        for (int w = 0; w < wave; ++w) b += wtot[w];
        if (t < ntiles) base[(uint64_t)t * ncnt + j] = b + incl - v;
        __syncthreads();
        if (tid == 1023) carry_s = b + incl;
        __syncthreads();
    }
    if (tid == 0) base[(uint64_t)ntiles * ncnt + j] = carry_s;
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

// Second stage of the typed conversion: cells whose text is beyond the 128-bit exact range
// (flagged ST_NUMERIC_RANGE by the tile kernel, which left them invalid) are converted with
// multi-precision arithmetic, reading the cell text back from the column arena.
__global__ __launch_bounds__(256) void long_float_kernel(const uint8_t *schema, uint32_t *const *off,
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
        if (typed != T_F64 && typed != T_F32) continue;
        if (!always && !((found[r * W + (c >> 5)] >> (c & 31)) & 1u)) continue;
        const uint32_t o = off[c][r], len = off[c][r + 1] - o;
        if (len <= 38) continue;
        uint64_t v = 0;
        bool ok = false, range = false;
        cell_walk<false>(bytes[c] + o, len, LVL_CHAR, false, typed, nullptr, v, ok, range);
        if (!range) continue;
        if (big_convert(bytes[c] + o, len, typed == T_F64, v)) {
            val[c][r] = v;
            atomicOr(&valid[r * W + (c >> 5)], 1u << (c & 31));
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
    hipLaunchKernelGGL(long_float_kernel, dim3(blocks), dim3(256), 0, stream, schema, off, bytes, val, valid, found,
                       rows);
    return hipGetLastError();
}

hipError_t launch_tile_kernel(bool write, const TileParams &P, hipStream_t stream) {
    if (P.ntiles == 0) return hipSuccess;
    // per device and cheap: set on every launch so that each GPU of a process gets it
    const void *fn = write ? (const void *)tile_kernel<true> : (const void *)tile_kernel<false>;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    if (write)
        hipLaunchKernelGGL(tile_kernel<true>, dim3(P.ntiles), dim3(BLOCK), P.L.total, stream, P);
    else
        hipLaunchKernelGGL(tile_kernel<false>, dim3(P.ntiles), dim3(BLOCK), P.L.total, stream, P);
    return hipGetLastError();
}

hipError_t launch_scan_tiles(const uint32_t *counts, uint64_t *base, uint32_t ntiles, uint32_t ncnt,
                             hipStream_t stream) {
    hipLaunchKernelGGL(scan_tiles_kernel, dim3(ncnt), dim3(1024), 0, stream, counts, base, ntiles, ncnt);
    return hipGetLastError();
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
