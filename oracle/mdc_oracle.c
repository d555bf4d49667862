/*
 * mdc_oracle.c -- plain-C CPU restatement of the reference's bulk record-parsing path.
 *
 * TEST INFRASTRUCTURE ONLY: this file is the parity checker and the timed CPU "port"
 * baseline.  The product library never includes, links or calls it.
 *
 * Pinning: the reference ships no tests or golden vectors for this path.  This
 * restatement is pinned (a) against the reference's own public/_public.cpp compiled in
 * place by oracle/Makefile (oracle/_ref/libmdc_ref.so, same C interface) on randomized and
 * adversarial inputs, and (b) against the fixtures under tests/golden/ that were generated
 * from that reference build by tests/golden/make_golden.py.
 *
 * Every function cites the reference lines it follows (paths relative to the reference
 * root).  Number conversion is the C library's strtol/strtoul/strtod/strtof, which is what
 * the reference's std::sto* calls reduce to (libstdc++ __stoa).
 */
#define _GNU_SOURCE
#include "mdc_oracle.h"
#include "orc_builder.h"

#include <errno.h>
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>

/* ------------------------------------------------------------------ small helpers */

typedef struct {
    uint8_t *p;
    size_t n, cap;
} bytebuf;

static void bb_reserve(bytebuf *b, size_t extra) {
    if (b->n + extra <= b->cap) return;
    size_t nc = b->cap ? b->cap * 2 : 256;
    while (nc < b->n + extra) nc *= 2;
    b->p = (uint8_t *)realloc(b->p, nc);
    b->cap = nc;
}
static void bb_assign(bytebuf *b, const uint8_t *s, size_t n) {
    b->n = 0;
    bb_reserve(b, n + 1);
    if (n) memmove(b->p, s, n);
    b->n = n;
}

/* std::string::find(needle, from) on raw bytes; returns (size_t)-1 when absent */
static size_t bytes_find(const uint8_t *h, size_t hn, const uint8_t *nd, size_t nn, size_t from) {
    if (from > hn) return (size_t)-1;
    if (nn == 0) return from;
    if (hn - from < nn) return (size_t)-1;
    const uint8_t *r = (const uint8_t *)memmem(h + from, hn - from, nd, nn);
    return r ? (size_t)(r - h) : (size_t)-1;
}

/* ------------------------------------------------------------------ A1 readline */

/* cifile::readline, public/_public.cpp:866-892, applied repeatedly until it returns false.
 * getline() strips the '\n'; a line that hits EOF before a '\n' sets eofbit and is dropped
 * (:876); lines are re-joined with '\n' (:888); the record is complete when the
 * accumulated buffer ends with endbz (:885), or after the first line when endbz is empty
 * (:880-881).  buf.find(endbz, buf.length()-endbz.length()) wraps to npos when the buffer
 * is shorter than endbz. */
int64_t ORC_FN(split_records)(const uint8_t *data, size_t n, const char *endbz, uint64_t *start,
                              uint64_t *len, int64_t cap) {
    size_t el = endbz ? strlen(endbz) : 0;
    size_t pos = 0;      /* stream position */
    size_t recstart = 0; /* where the accumulated buffer begins */
    int64_t count = 0;
    for (;;) {
        const uint8_t *nl = pos < n ? (const uint8_t *)memchr(data + pos, '\n', n - pos) : NULL;
        if (!nl) break; /* eof while reading this line */
        size_t lineend = (size_t)(nl - data);
        size_t buflen = lineend - recstart;
        int complete;
        if (el == 0)
            complete = 1;
        else
            complete = buflen >= el && memcmp(data + lineend - el, endbz, el) == 0;
        pos = lineend + 1;
        if (complete) {
            if (count < cap) {
                if (start) start[count] = recstart;
                if (len) len[count] = buflen;
            }
            ++count;
            recstart = pos;
        }
    }
    return count;
}

/* ------------------------------------------------------------------ A2 getxmlbuffer */

/* getxmlbuffer(const string&,const string&,string&,int), public/_public.cpp:461-481.
 * Both tags are searched from offset 0 independently (:466, :469).  itmplen is computed
 * in size_t and narrowed to int (:476): a close tag in front of the open tag gives a
 * negative int, which skips the clip (:477) and becomes npos in substr (:478), i.e. the
 * rest of the record. */
static int getxml_span(const uint8_t *rec, size_t reclen, const char *name, int ilen, size_t *pos,
                       size_t *len) {
    size_t nl = strlen(name);
    uint8_t st[40 + 256], en[40 + 256];
    if (nl > 256) return 0;
    st[0] = '<';
    memcpy(st + 1, name, nl);
    st[1 + nl] = '>';
    en[0] = '<';
    en[1] = '/';
    memcpy(en + 2, name, nl);
    en[2 + nl] = '>';
    size_t sl = nl + 2, enl = nl + 3;

    size_t startp = bytes_find(rec, reclen, st, sl, 0);
    if (startp == (size_t)-1) return 0;
    size_t endp = bytes_find(rec, reclen, en, enl, 0);
    if (endp == (size_t)-1) return 0;

    int itmplen = (int)(endp - startp - sl);
    if (ilen > 0 && ilen < itmplen) itmplen = ilen;
    size_t from = startp + sl;
    size_t cnt = (size_t)itmplen; /* negative -> huge */
    size_t avail = reclen - from;
    if (cnt > avail) cnt = avail;
    *pos = from;
    *len = cnt;
    return 1;
}

int ORC_FN(getxmlbuffer)(const uint8_t *rec, size_t reclen, const char *name, int ilen,
                         uint8_t *out, size_t outcap, size_t *outlen) {
    size_t p = 0, l = 0;
    if (!getxml_span(rec, reclen, name, ilen, &p, &l)) return 0;
    if (out && outcap) memcpy(out, rec + p, l < outcap ? l : outcap);
    if (outlen) *outlen = l;
    return 1;
}

/* ------------------------------------------------------------------ A3 picknumber */

/* picknumber(const string&,string&,bool,bool), public/_public.cpp:178-204.
 * isdigit() in the C locale (the reference never calls setlocale) is '0'..'9' only. */
size_t ORC_FN(picknumber)(const uint8_t *src, size_t n, uint8_t *dst, int bsigned, int bdot) {
    size_t o = 0;
    for (size_t i = 0; i < n; ++i) {
        uint8_t cc = src[i];
        if (bsigned && (cc == '+' || cc == '-')) {
            dst[o++] = cc;
            continue;
        }
        if (bdot && cc == '.') {
            dst[o++] = cc;
            continue;
        }
        if (cc >= '0' && cc <= '9') dst[o++] = cc;
    }
    return o;
}

/* ------------------------------------------------------------------ A9/A10 trims, case */

/* string overloads, public/_public.cpp:24-31 (deletelchr), :52-59 (deleterchr),
 * :69-75 (deletelrchr).  deleterchr's "if (pos!=0)" test keeps trailing cc characters
 * when the last other character sits at index 0; npos+1 wraps to 0 for an all-cc string. */
size_t ORC_FN(trim)(uint8_t *s, size_t n, int cc, int which) {
    uint8_t c = (uint8_t)cc;
    if (which == 0 || which == 2) {
        size_t pos = 0;
        while (pos < n && s[pos] == c) ++pos;
        if (pos == n) pos = (size_t)-1; /* find_first_not_of -> npos */
        if (pos != 0) {
            size_t cut = pos == (size_t)-1 ? n : pos; /* replace(0,npos,"") erases all */
            memmove(s, s + cut, n - cut);
            n -= cut;
        }
    }
    if (which == 1 || which == 2) {
        size_t pos = (size_t)-1; /* find_last_not_of */
        for (size_t i = n; i-- > 0;)
            if (s[i] != c) {
                pos = i;
                break;
            }
        if (pos != 0) {
            size_t keep = pos + 1; /* npos+1 == 0 */
            if (keep < n) n = keep;
        }
    }
    return n;
}

/* toupper(string&), public/_public.cpp:91-99: ASCII a-z only */
static void ascii_upper(uint8_t *s, size_t n) {
    for (size_t i = 0; i < n; ++i)
        if (s[i] >= 'a' && s[i] <= 'z') s[i] = (uint8_t)(s[i] - 32);
}

/* ------------------------------------------------------------------ A7 typed conversion */

/* The typed getxmlbuffer / ccmdstr::getvalue overloads, public/_public.cpp:506-625 and
 * :339-446: filter with picknumber, convert with std::sto*, map any exception to false.
 * libstdc++'s __stoa: no conversion -> invalid_argument; errno==ERANGE (or an int range
 * check for stoi) -> out_of_range. */
int ORC_FN(convert)(const uint8_t *s, size_t n, int type, uint64_t *out8) {
    if (type == ORC_T_NONE) return 0;
    if (type == ORC_T_BOOL) {
        /* :506-517 / :435-446 */
        int istrue = 0;
        if (n == 4) {
            uint8_t t[4];
            memcpy(t, s, 4);
            ascii_upper(t, 4);
            istrue = memcmp(t, "TRUE", 4) == 0;
        }
        *out8 = (uint64_t)istrue;
        return 1;
    }
    if (type == ORC_T_TIME) {
        /* strtotime, public/_public.cpp:1009-1044: the digits (picknumber without sign and
         * point) must be fourteen; yyyy mm dd hh mi ss go to mktime (libc, the process's zone)
         * with tm_isdst = 0.  Its "return -1" for any other count is the failure here. */
        char d[14];
        size_t nd = 0;
        for (size_t i = 0; i < n; ++i)
            if (s[i] >= '0' && s[i] <= '9') {
                if (nd < 14) d[nd] = (char)s[i];
                ++nd;
            }
        if (nd != 14) return 0;
        struct tm tmv;
        memset(&tmv, 0, sizeof tmv);
        tmv.tm_year = (d[0] - '0') * 1000 + (d[1] - '0') * 100 + (d[2] - '0') * 10 + (d[3] - '0') - 1900;
        tmv.tm_mon = (d[4] - '0') * 10 + (d[5] - '0') - 1;
        tmv.tm_mday = (d[6] - '0') * 10 + (d[7] - '0');
        tmv.tm_hour = (d[8] - '0') * 10 + (d[9] - '0');
        tmv.tm_min = (d[10] - '0') * 10 + (d[11] - '0');
        tmv.tm_sec = (d[12] - '0') * 10 + (d[13] - '0');
        tmv.tm_isdst = 0;
        *out8 = (uint64_t)(int64_t)mktime(&tmv);
        return 1;
    }
    int bsigned = (type == ORC_T_I32 || type == ORC_T_I64 || type == ORC_T_F64 || type == ORC_T_F32);
    int bdot = (type == ORC_T_F64 || type == ORC_T_F32);
    char *f = (char *)malloc(n + 1);
    size_t fl = ORC_FN(picknumber)(s, n, (uint8_t *)f, bsigned, bdot);
    f[fl] = 0;
    char *end = NULL;
    int ok = 0;
    errno = 0;
    switch (type) {
    case ORC_T_I32:
    case ORC_T_U32: { /* both go through stoi (:527, :545) */
        long v = strtol(f, &end, 10);
        if (end != f && errno != ERANGE && v >= INT_MIN && v <= INT_MAX) {
            if (type == ORC_T_I32)
                *out8 = (uint64_t)(int64_t)(int)v;
            else
                *out8 = (uint64_t)(unsigned int)(int)v;
            ok = 1;
        }
        break;
    }
    case ORC_T_I64: { /* stol :563 */
        long v = strtol(f, &end, 10);
        if (end != f && errno != ERANGE) {
            *out8 = (uint64_t)v;
            ok = 1;
        }
        break;
    }
    case ORC_T_U64: { /* stoul :581 */
        unsigned long v = strtoul(f, &end, 10);
        if (end != f && errno != ERANGE) {
            *out8 = (uint64_t)v;
            ok = 1;
        }
        break;
    }
    case ORC_T_F64: { /* stod :599 */
        double v = strtod(f, &end);
        if (end != f && errno != ERANGE) {
            memcpy(out8, &v, 8);
            ok = 1;
        }
        break;
    }
    case ORC_T_F32: { /* stof :617 */
        float v = strtof(f, &end);
        if (end != f && errno != ERANGE) {
            uint32_t b;
            memcpy(&b, &v, 4);
            *out8 = b;
            ok = 1;
        }
        break;
    }
    default:
        break;
    }
    free(f);
    return ok;
}

/* ------------------------------------------------------------------ A4 xmltodb path */

/* _xmltodb record loop (tools/xmltodb.cpp:263-268) + splitbuffer (:492-522).
 * temp lives for one record (:494); getxmlbuffer's return value is ignored so temp keeps
 * the previous column's filtered value when a tag is missing (:498-499); date columns are
 * filtered with picknumber(false,false) (:503-506), number columns with (true,true)
 * (:507-511); the bound value is temp.c_str(), i.e. cut at the first NUL (:518). */
orc_batch *ORC_FN(parse_xml)(const uint8_t *data, size_t n, const char *endbz, const orc_col *cols,
                             int ncols, const int *typed, int carry) {
    int *ty = (int *)calloc((size_t)ncols + 1, sizeof(int));
    default_typed(cols, ncols, typed, ty);
    builder *B = builder_new(ncols, ty);
    orc_batch *b = &B->b;

    int64_t nrec = ORC_FN(split_records)(data, n, endbz, NULL, NULL, 0);
    uint64_t *rs = (uint64_t *)malloc(((size_t)nrec + 1) * 8);
    uint64_t *rl = (uint64_t *)malloc(((size_t)nrec + 1) * 8);
    ORC_FN(split_records)(data, n, endbz, rs, rl, nrec);

    bytebuf temp = {0, 0, 0};
    for (int64_t r = 0; r < nrec; ++r) {
        while (r >= B->cap_rows) builder_grow_rows(B);
        const uint8_t *rec = data + rs[r];
        size_t reclen = (size_t)rl[r];
        b->rec_off[r] = rs[r];
        b->rec_off[r + 1] = rs[r] + rl[r] + 1;
        temp.n = 0; /* string temp; */
        for (int c = 0; c < ncols; ++c) {
            size_t p = 0, l = 0;
            int found = getxml_span(rec, reclen, cols[c].colname, cols[c].collen, &p, &l);
            if (found) {
                bb_assign(&temp, rec + p, l);
                b->found[r * b->words_per_row + (c >> 5)] |= 1u << (c & 31);
            } else if (!carry) {
                temp.n = 0;
            }
            if (strcmp(cols[c].datatype, "date") == 0)
                temp.n = ORC_FN(picknumber)(temp.p, temp.n, temp.p, 0, 0);
            else if (strcmp(cols[c].datatype, "number") == 0)
                temp.n = ORC_FN(picknumber)(temp.p, temp.n, temp.p, 1, 1);
            size_t cl = temp.n;
            if (carry && cl) { /* vcolvalue[i] = temp.c_str(); */
                const uint8_t *z = (const uint8_t *)memchr(temp.p, 0, cl);
                if (z) cl = (size_t)(z - temp.p);
            }
            builder_put_cell(B, r, c, temp.p, cl);
            if (ty[c] != ORC_T_NONE && (carry || found)) {
                uint64_t v = 0;
                if (ORC_FN(convert)(temp.p, cl, ty[c], &v)) {
                    b->val[c][r] = v;
                    b->valid[r * b->words_per_row + (c >> 5)] |= 1u << (c & 31);
                }
            }
        }
    }
    b->rows = nrec;
    free(temp.p);
    free(rs);
    free(rl);
    free(ty);
    return b;
}

orc_batch *ORC_FN(parse_xml_file)(const char *path, const char *endbz, const orc_col *cols,
                                  int ncols, const int *typed, int carry) {
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *buf = (uint8_t *)malloc((size_t)sz + 1);
    size_t got = fread(buf, 1, (size_t)sz, f);
    fclose(f);
    orc_batch *b = ORC_FN(parse_xml)(buf, got, endbz, cols, ncols, typed, carry);
    free(buf);
    return b;
}

/* ------------------------------------------------------------------ A8 delimited path */

/* ccmdstr::splittocmd (public/_public.cpp:274-302) on every readline("") line, then
 * getvalue(ii,string&,ilen) (:304-317) for the cell and the typed getvalue overloads
 * (:339-446) for the value.  The search restarts after each separator (:291) so separators
 * never overlap; the text after the last separator is always a field (:295-299). */
orc_batch *ORC_FN(parse_csv)(const uint8_t *data, size_t n, const char *sep, int bdelspace,
                             int nfields, const int *ilen, const int *typed) {
    int *ty = (int *)calloc((size_t)nfields + 1, sizeof(int));
    for (int c = 0; c < nfields; ++c) ty[c] = typed ? typed[c] : ORC_T_NONE;
    builder *B = builder_new(nfields, ty);
    orc_batch *b = &B->b;
    size_t sl = strlen(sep);

    int64_t nrec = ORC_FN(split_records)(data, n, "", NULL, NULL, 0);
    uint64_t *rs = (uint64_t *)malloc(((size_t)nrec + 1) * 8);
    uint64_t *rl = (uint64_t *)malloc(((size_t)nrec + 1) * 8);
    ORC_FN(split_records)(data, n, "", rs, rl, nrec);

    bytebuf fld = {0, 0, 0};
    for (int64_t r = 0; r < nrec; ++r) {
        while (r >= B->cap_rows) builder_grow_rows(B);
        const uint8_t *rec = data + rs[r];
        size_t reclen = (size_t)rl[r];
        b->rec_off[r] = rs[r];
        b->rec_off[r + 1] = rs[r] + rl[r] + 1;
        size_t pos = 0;
        int done = 0; /* set once the trailing field has been consumed */
        for (int c = 0; c < nfields; ++c) {
            if (done) { /* ii >= m_cmdstr.size(): getvalue returns false */
                builder_put_cell(B, r, c, NULL, 0);
                continue;
            }
            size_t pos1 = bytes_find(rec, reclen, (const uint8_t *)sep, sl, pos);
            size_t flen;
            if (pos1 == (size_t)-1) {
                flen = reclen - pos;
                done = 1;
            } else {
                flen = pos1 - pos;
            }
            bb_assign(&fld, rec + pos, flen);
            if (pos1 != (size_t)-1) pos = pos1 + sl;
            if (bdelspace) fld.n = ORC_FN(trim)(fld.p, fld.n, ' ', 2);
            b->found[r * b->words_per_row + (c >> 5)] |= 1u << (c & 31);
            size_t cl = fld.n;
            if (ilen && ilen[c] > 0 && (size_t)ilen[c] < cl) cl = (size_t)ilen[c];
            builder_put_cell(B, r, c, fld.p, cl);
            if (ty[c] != ORC_T_NONE) {
                uint64_t v = 0;
                if (ORC_FN(convert)(fld.p, cl, ty[c], &v)) {
                    b->val[c][r] = v;
                    b->valid[r * b->words_per_row + (c >> 5)] |= 1u << (c & 31);
                }
            }
        }
    }
    b->rows = nrec;
    free(fld.p);
    free(rs);
    free(rl);
    free(ty);
    return b;
}

void ORC_FN(batch_free)(orc_batch *b) { builder_free(b); }

/* ---- N1: writer side -------------------------------------------------------------------- */

/* bytes "%s" prints of a fetched value: the fetch buffer holds at most fieldlen bytes and a
 * terminating NUL (public/db/oracle/_ooci.cpp:622-627), c_str() ends at the first NUL
 * (tools/dminingoracle.cpp:217-218) */
static size_t value_len(const uint8_t *p, size_t n, int fieldlen) {
    if (fieldlen > 0 && n > (size_t)fieldlen) n = (size_t)fieldlen;
    const void *z = memchr(p, 0, n);
    return z ? (size_t)((const uint8_t *)z - p) : n;
}

static void put(uint8_t *out, uint64_t *pos, const void *src, size_t n) {
    if (out) memcpy(out + *pos, src, n);
    *pos += n;
}

int64_t ORC_FN(format_xml)(const char *const *fieldname, const int *fieldlen, int nfields,
                           const uint32_t *const *off, const uint8_t *const *bytes, int64_t rows,
                           int64_t maxcount, uint8_t *out, uint64_t *file_off, int64_t filecap,
                           int64_t *nfiles) {
    uint64_t pos = 0;
    int64_t files = 0, rpc = 0;
    int isopen = 0;
    for (int64_t r = 0; r < rows; ++r) { /* while (stmtsel.next()==0), :200 */
        ++rpc;
        if (!isopen) {                   /* :202-213 */
            if (file_off && files < filecap) file_off[files] = pos;
            ++files;
            isopen = 1;
            put(out, &pos, "<data>\n", 7);
        }
        for (int i = 0; i < nfields; ++i) { /* :216-218 */
            size_t nl = strlen(fieldname[i]);
            const uint8_t *v = bytes[i] + off[i][r];
            size_t vl = value_len(v, off[i][r + 1] - off[i][r], fieldlen ? fieldlen[i] : 0);
            put(out, &pos, "<", 1);
            put(out, &pos, fieldname[i], nl);
            put(out, &pos, ">", 1);
            put(out, &pos, v, vl);
            put(out, &pos, "</", 2);
            put(out, &pos, fieldname[i], nl);
            put(out, &pos, ">", 1);
        }
        put(out, &pos, "<endl/>\n", 8); /* :219 */
        if (maxcount > 0 && rpc % maxcount == 0) { /* :222-233 */
            put(out, &pos, "</data>\n", 8);
            isopen = 0;
        }
    }
    if (maxcount == 0 || rpc % maxcount > 0) { /* :241-256 */
        if (!isopen) return -1;                /* writeline and closeandrename fail */
        put(out, &pos, "</data>\n", 8);
        isopen = 0;
    }
    if (file_off && files < filecap) file_off[files] = pos;
    if (nfiles) *nfiles = files;
    return (int64_t)pos;
}

/* create the directories above a file name (newdir(name,true), public/_public.cpp:952-980) */
static void make_parents(const char *path) {
    char tmp[4096];
    size_t n = strlen(path);
    if (n >= sizeof tmp) return;
    memcpy(tmp, path, n + 1);
    for (size_t i = 1; i < n; ++i)
        if (tmp[i] == '/') {
            tmp[i] = 0;
            mkdir(tmp, 0755);
            tmp[i] = '/';
        }
}

int64_t ORC_FN(format_xml_files)(const char *const *fieldname, const int *fieldlen, int nfields,
                                 const uint32_t *const *off, const uint8_t *const *bytes,
                                 int64_t rows, int64_t maxcount, const char *outpath,
                                 const char *bfilename, const char *timestamp,
                                 const char *efilename, int first_seq) {
    int64_t nfiles = 0;
    int64_t total = ORC_FN(format_xml)(fieldname, fieldlen, nfields, off, bytes, rows, maxcount, NULL,
                                       NULL, 0, &nfiles);
    if (total < 0) return -1;
    uint8_t *buf = (uint8_t *)malloc((size_t)total + 1);
    uint64_t *fo = (uint64_t *)malloc((size_t)(nfiles + 1) * sizeof(uint64_t));
    if (!buf || !fo) {
        free(buf);
        free(fo);
        return -1;
    }
    ORC_FN(format_xml)(fieldname, fieldlen, nfields, off, bytes, rows, maxcount, buf, fo, nfiles + 1,
                       &nfiles);
    int64_t done = 0;
    for (int64_t k = 0; k < nfiles; ++k) {
        char name[4096], tmp[4200];
        snprintf(name, sizeof name, "%s/%s_%s_%s_%d.xml", outpath, bfilename, timestamp, efilename,
                 first_seq + (int)k); /* :204-205 */
        snprintf(tmp, sizeof tmp, "%s.tmp", name);
        make_parents(name);
        FILE *f = fopen(tmp, "wb");
        if (!f) break;
        size_t len = (size_t)(fo[k + 1] - fo[k]);
        int ok = fwrite(buf + fo[k], 1, len, f) == len;
        ok = (fclose(f) == 0) && ok;
        if (!ok || rename(tmp, name) != 0) break;
        ++done;
    }
    free(buf);
    free(fo);
    return done == nfiles ? nfiles : -1;
}

/* ---- N4: timetostr ----------------------------------------------------------------------- */
/* public/_public.cpp:631-722: localtime_r in the process's zone, year + 1900 and month + 1
 * printed with %04u / %02u in the format's arrangement; the chain of string compares :639-719
 * as a table of (name, printf format, number of fields, first field). */
int ORC_FN(timetostr)(int64_t t, const char *fmt, char *out, size_t cap) {
    time_t tt = (time_t)t;
    struct tm sttm;
    memset(&sttm, 0, sizeof sttm);
    localtime_r(&tt, &sttm);
    unsigned v[6] = {(unsigned)(sttm.tm_year + 1900), (unsigned)(sttm.tm_mon + 1), (unsigned)sttm.tm_mday,
                     (unsigned)sttm.tm_hour, (unsigned)sttm.tm_min, (unsigned)sttm.tm_sec};
    static const struct {
        const char *name, *pf;
        int first;
    } table[] = {
        {"", "%04u-%02u-%02u %02u:%02u:%02u", 0},
        {"yyyy-mm-dd hh24:mi:ss", "%04u-%02u-%02u %02u:%02u:%02u", 0},
        {"yyyy-mm-dd hh24:mi", "%04u-%02u-%02u %02u:%02u", 0},
        {"yyyy-mm-dd hh24", "%04u-%02u-%02u %02u", 0},
        {"yyyy-mm-dd", "%04u-%02u-%02u", 0},
        {"yyyy-mm", "%04u-%02u", 0},
        {"yyyymmddhh24miss", "%04u%02u%02u%02u%02u%02u", 0},
        {"yyyymmddhh24mi", "%04u%02u%02u%02u%02u", 0},
        {"yyyymmddhh24", "%04u%02u%02u%02u", 0},
        {"yyyymmdd", "%04u%02u%02u", 0},
        {"hh24miss", "%02u%02u%02u", 3},
        {"hh24mi", "%02u%02u", 3},
        {"hh24", "%02u", 3},
        {"mi", "%02u", 4},
    };
    for (size_t i = 0; i < sizeof table / sizeof table[0]; ++i)
        if (strcmp(fmt, table[i].name) == 0) {
            unsigned a[6] = {0, 0, 0, 0, 0, 0};
            for (int k = 0; table[i].first + k < 6; ++k) a[k] = v[table[i].first + k];
            /* (arguments beyond those the format names are ignored by printf) */
            int n = snprintf(out, cap, table[i].pf, a[0], a[1], a[2], a[3], a[4], a[5]);
            return n < 0 ? -1 : n;
        }
    return -1;
}

/* ---- N2: routing, matchstr, SQL text ---------------------------------------------------- */

/* one piece of a splittocmd() result: [p, p+n) */
typedef struct {
    const uint8_t *p;
    size_t n;
} piece;

/* ccmdstr::splittocmd(buffer,sep) with bdelspace=false (public/_public.cpp:274-302): greedy
 * left-to-right search; the text after the last separator is the last field.  Returns the
 * number of fields (at most cap are written). */
static size_t split_pieces(const uint8_t *b, size_t n, const char *sep, piece *out, size_t cap) {
    size_t sl = strlen(sep), pos = 0, k = 0;
    for (;;) {
        size_t p = bytes_find(b, n, (const uint8_t *)sep, sl, pos);
        if (p == (size_t)-1) break;
        if (k < cap) {
            out[k].p = b + pos;
            out[k].n = p - pos;
        }
        ++k;
        pos = p + sl;
    }
    if (k < cap) {
        out[k].p = b + pos;
        out[k].n = n - pos;
    }
    return k + 1;
}

/* matchstr, public/_public.cpp:213-263.  Both strings are upper-cased (ASCII); every rule
 * between commas is cut at '*' and the pieces must occur in order, the first at the start of
 * the name (:245-246), the last at its end (:249-250: find from length-piece length; that
 * position wraps when the piece is longer than the name, and nothing is found). */
int ORC_FN(matchstr)(const char *str, const char *rules) {
    size_t fl = strlen(str), rl = strlen(rules);
    if (rl == 0) return 0;
    if (rl == 1 && rules[0] == '*') return 1;
    uint8_t *f = (uint8_t *)malloc(fl + 1), *m = (uint8_t *)malloc(rl + 1);
    memcpy(f, str, fl);
    memcpy(m, rules, rl);
    ascii_upper(f, fl);
    ascii_upper(m, rl);
    size_t nr = split_pieces(m, rl, ",", NULL, 0);
    piece *rule = (piece *)malloc(nr * sizeof(piece));
    split_pieces(m, rl, ",", rule, nr);
    int hit = 0;
    for (size_t ii = 0; ii < nr && !hit; ++ii) {
        if (rule[ii].n == 0) continue; /* :237 */
        size_t np = split_pieces(rule[ii].p, rule[ii].n, "*", NULL, 0);
        piece *sub = (piece *)malloc(np * sizeof(piece));
        split_pieces(rule[ii].p, rule[ii].n, "*", sub, np);
        size_t pos1 = 0, jj;
        for (jj = 0; jj < np; ++jj) {
            const uint8_t *s = sub[jj].p;
            size_t sn = sub[jj].n;
            if (jj == 0) { /* filename.substr(0,len) != piece */
                size_t head = sn < fl ? sn : fl;
                if (head != sn || memcmp(f, s, sn) != 0) break;
            }
            if (jj == np - 1) {
                if (sn > fl) break; /* wrapped start position: npos */
                if (bytes_find(f, fl, s, sn, fl - sn) == (size_t)-1) break;
            }
            size_t pos2 = bytes_find(f, fl, s, sn, pos1); /* :252 */
            if (pos2 == (size_t)-1) break;
            pos1 = pos2 + sn;
        }
        if (jj == np) hit = 1;
        free(sub);
    }
    free(rule);
    free(f);
    free(m);
    return hit;
}

/* value of the char* getxmlbuffer (public/_public.cpp:483-504, len=0: no limit) as the C string
 * its caller reads: cut at the first NUL, and at cap-1 bytes */
static void cfield(const uint8_t *rec, size_t reclen, const char *name, char *dst, size_t cap) {
    size_t p = 0, l = 0;
    if (!getxml_span(rec, reclen, name, 0, &p, &l)) l = 0; /* str stays empty: "" is copied */
    const void *z = l ? memchr(rec + p, 0, l) : NULL;
    if (z) l = (size_t)((const uint8_t *)z - (rec + p));
    if (l > cap - 1) l = cap - 1;
    if (l) memcpy(dst, rec + p, l);
    dst[l] = 0;
}

/* loadxmltotable, tools/xmltodb.cpp:102-129 */
int64_t ORC_FN(load_xmltotable)(const uint8_t *data, size_t n, orc_xmltotable *out, int64_t cap) {
    int64_t nrec = ORC_FN(split_records)(data, n, "<endl/>", NULL, NULL, 0);
    if (nrec <= 0) return 0;
    uint64_t *st = (uint64_t *)malloc((size_t)nrec * sizeof(uint64_t));
    uint64_t *ln = (uint64_t *)malloc((size_t)nrec * sizeof(uint64_t));
    ORC_FN(split_records)(data, n, "<endl/>", st, ln, nrec);
    for (int64_t r = 0; r < nrec && r < cap; ++r) {
        const uint8_t *rec = data + st[r];
        size_t rl = (size_t)ln[r];
        orc_xmltotable *e = &out[r];
        memset(e, 0, sizeof *e); /* :117 */
        cfield(rec, rl, "filename", e->filename, sizeof e->filename);
        cfield(rec, rl, "tname", e->tname, sizeof e->tname);
        size_t p = 0, l = 0; /* :121, the int overload: untouched (0) when it fails */
        if (getxml_span(rec, rl, "uptbz", 0, &p, &l)) {
            uint64_t v;
            if (ORC_FN(convert)(rec + p, l, ORC_T_I32, &v)) e->uptbz = (int)(int64_t)v;
        }
        cfield(rec, rl, "execsql", e->execsql, sizeof e->execsql);
    }
    free(st);
    free(ln);
    return nrec;
}

/* a growing C string */
typedef struct {
    char *p;
    size_t n, cap;
} cstr;
static void cs_add(cstr *s, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
#include <stdarg.h>
static void cs_add(cstr *s, const char *fmt, ...) {
    va_list ap, aq;
    va_start(ap, fmt);
    va_copy(aq, ap);
    int need = vsnprintf(NULL, 0, fmt, ap);
    va_end(ap);
    if (s->n + (size_t)need + 1 > s->cap) {
        s->cap = (s->n + (size_t)need + 1) * 2;
        s->p = (char *)realloc(s->p, s->cap);
    }
    vsnprintf(s->p + s->n, (size_t)need + 1, fmt, aq);
    va_end(aq);
    s->n += (size_t)need;
}
/* deleterchr(str,','), public/_public.cpp:52-59 (same rule as the trim above) */
static void cs_drop_commas(cstr *s) {
    if (!s->p) return;
    s->n = ORC_FN(trim)((uint8_t *)s->p, s->n, ',', 1);
    s->p[s->n] = 0;
}
static void cs_out(const cstr *s, char *dst, size_t cap) {
    if (!dst || !cap) return;
    size_t k = s->n < cap - 1 ? s->n : cap - 1;
    if (k) memcpy(dst, s->p, k);
    dst[k] = 0;
}

/* crtsql, tools/xmltodb.cpp:325-419 */
int ORC_FN(crtsql)(const orc_xmltotable *e, const orc_col *cols, int ncols, char *ins, size_t inscap,
                   char *upd, size_t updcap) {
    cstr names = {0, 0, 0}, binds = {0, 0, 0}, sql = {0, 0, 0};
    cs_add(&names, "%s", "");
    cs_add(&binds, "%s", "");
    int colseq = 1;
    for (int i = 0; i < ncols; ++i) {
        if (strcmp(cols[i].colname, "upttime") == 0) continue; /* :337 */
        cs_add(&names, "%s,", cols[i].colname);
        if (strcmp(cols[i].colname, "keyid") == 0)
            cs_add(&binds, "SEQ_%s.nextval,", e->tname + 2); /* :345; the array is zeroed */
        else if (strcmp(cols[i].datatype, "date") == 0)
            cs_add(&binds, "to_date(:%d,'yyyymmddhh24miss'),", colseq++);
        else
            cs_add(&binds, ":%d,", colseq++);
    }
    cs_drop_commas(&names);
    cs_drop_commas(&binds);
    cs_add(&sql, "insert into %s(%s) values(%s)", e->tname, names.p, binds.p);
    cs_out(&sql, ins, inscap);
    free(names.p);
    free(binds.p);
    free(sql.p);
    if (e->uptbz != 1) return 0; /* :365 */

    cstr set = {0, 0, 0}, where = {0, 0, 0}, usql = {0, 0, 0};
    cs_add(&set, " set ");
    cs_add(&where, " where 1=1");
    colseq = 1;
    for (int i = 0; i < ncols; ++i) {
        if (cols[i].pkseq != 0) continue;
        if (strcmp(cols[i].colname, "keyid") == 0) continue;
        if (strcmp(cols[i].colname, "upttime") == 0)
            cs_add(&set, "upttime=sysdate,");
        else if (strcmp(cols[i].datatype, "date") == 0)
            cs_add(&set, "%s=to_date(:%d,'yyyymmddhh24miss),", cols[i].colname, colseq++); /* :393 as written */
        else
            cs_add(&set, "%s=:%d,", cols[i].colname, colseq++);
    }
    cs_drop_commas(&set);
    /* m_vpkcols: the key columns by their position in the key (ties: schema order) */
    int *pk = (int *)malloc((size_t)(ncols > 0 ? ncols : 1) * sizeof(int)), npk = 0;
    for (int i = 0; i < ncols; ++i)
        if (cols[i].pkseq != 0) {
            int at = npk++;
            while (at > 0 && cols[pk[at - 1]].pkseq > cols[i].pkseq) {
                pk[at] = pk[at - 1];
                --at;
            }
            pk[at] = i;
        }
    for (int k = 0; k < npk; ++k) { /* :403-414 */
        const orc_col *c = &cols[pk[k]];
        if (strcmp(c->datatype, "date") == 0)
            cs_add(&where, " and %s=to_date(:%d,'yyyymmddhh24miss')", c->colname, colseq++);
        else
            cs_add(&where, " and %s=:%d", c->colname, colseq++);
    }
    free(pk);
    cs_add(&usql, "update %s%s%s", e->tname, set.p, where.p);
    cs_out(&usql, upd, updcap);
    free(set.p);
    free(where.p);
    free(usql.p);
    return 1;
}

/* preparesql, tools/xmltodb.cpp:421-477 */
int ORC_FN(bind_plan)(const orc_xmltotable *e, const orc_col *cols, int ncols, int *ins, int *upd,
                      int *nupd) {
    int k = 0, u = 0;
    for (int i = 0; i < ncols; ++i) { /* :431-442 */
        if (strcmp(cols[i].colname, "upttime") == 0) continue;
        if (strcmp(cols[i].colname, "keyid") == 0) continue;
        ins[k++] = i;
    }
    if (e->uptbz == 1) {
        for (int i = 0; i < ncols; ++i) { /* :453-466 */
            if (cols[i].pkseq != 0) continue;
            if (strcmp(cols[i].colname, "keyid") == 0) continue;
            if (strcmp(cols[i].colname, "upttime") == 0) continue;
            upd[u++] = i;
        }
        for (int i = 0; i < ncols; ++i) /* :469-474 */
            if (cols[i].pkseq != 0) upd[u++] = i;
    }
    if (nupd) *nupd = u;
    return k;
}

const char *ORC_FN(kind)(void) { return "port"; }
