/*
 * mdc_oracle.h -- C interface shared by the two CPU checkers:
 *
 *   oracle/mdc_oracle.c     plain-C restatement of the reference algorithm ("port")
 *   oracle/ref_harness.cpp  the reference's own public/_public.cpp compiled in place
 *                           and wrapped behind the same entry points ("reference")
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product library may include, link or
 * call this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * use it, and only as the checker / the timed CPU baseline.
 *
 * Both libraries export every function below with the prefix given by
 * ORC_PREFIX (orc_ for the port, ref_ for the reference build) so that a test can
 * load both and compare them on the same inputs.
 */
#ifndef MDC_ORACLE_H
#define MDC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* mirrors ctcols::st_col, tools/_tools.h:21-27 */
typedef struct orc_col {
    char colname[32];
    char datatype[32]; /* "char" | "date" | "number" after tools/_tools.cpp:47-71 */
    int collen;
    int pkseq;
} orc_col;

/* typed conversions of public/_public.cpp:339-446 and :506-625 */
enum {
    ORC_T_NONE = 0,
    ORC_T_I32 = 1,  /* stoi(picknumber(s,true))        */
    ORC_T_U32 = 2,  /* stoi(picknumber(s))             */
    ORC_T_I64 = 3,  /* stol(picknumber(s,true))        */
    ORC_T_U64 = 4,  /* stoul(picknumber(s))            */
    ORC_T_F64 = 5,  /* stod(picknumber(s,true,true))   */
    ORC_T_F32 = 6,  /* stof(picknumber(s,true,true))   */
    ORC_T_BOOL = 7, /* toupper(s)=="TRUE"              */
    ORC_T_TIME = 8  /* strtotime(s), public/_public.cpp:1009-1044, in the process's time zone */
};

/* Columnar result of parsing a whole buffer; same layout as the GPU batch.
 * Column c, row r: bytes[c][off[c][r] .. off[c][r+1]).  val[c] holds rows 8-byte
 * slots (the typed value in the low bytes, zero when invalid); valid/found are
 * row-major bit masks, words_per_row u32 words per row, bit (c&31) of word c>>5. */
typedef struct orc_batch {
    int64_t rows;
    int ncols;
    int words_per_row;
    uint64_t *rec_off;   /* rows+1 record start offsets; record r = [rec_off[r], rec_off[r+1]-1) */
    uint32_t **off;      /* ncols arrays of rows+1 */
    uint8_t **bytes;     /* ncols byte arenas */
    uint64_t *nbytes;    /* ncols arena sizes */
    uint64_t **val;      /* ncols arrays of rows (NULL where typed[c]==NONE) */
    uint32_t *valid;     /* rows*words_per_row */
    uint32_t *found;     /* rows*words_per_row */
} orc_batch;

#ifndef ORC_PREFIX
#define ORC_PREFIX orc_
#endif
#define ORC_CAT2(a, b) a##b
#define ORC_CAT(a, b) ORC_CAT2(a, b)
#define ORC_FN(name) ORC_CAT(ORC_PREFIX, name)

/* A1  cifile::readline over a whole buffer (public/_public.cpp:866-892).
 * Writes up to cap record (start,len) pairs; returns the number of records found. */
int64_t ORC_FN(split_records)(const uint8_t *data, size_t n, const char *endbz,
                              uint64_t *start, uint64_t *len, int64_t cap);

/* A2  getxmlbuffer(string&) (public/_public.cpp:461-481). Returns 1 when found and copies
 * the value (at most outcap bytes) to out, its full length to *outlen; returns 0 and
 * leaves both untouched otherwise. */
int ORC_FN(getxmlbuffer)(const uint8_t *rec, size_t reclen, const char *name, int ilen,
                         uint8_t *out, size_t outcap, size_t *outlen);

/* A3  picknumber (public/_public.cpp:178-204); dst may alias src. Returns length. */
size_t ORC_FN(picknumber)(const uint8_t *src, size_t n, uint8_t *dst, int bsigned, int bdot);

/* A7  typed conversion of one field's text; returns 1 and stores into *out8 (value in the
 * low bytes) when the reference overload would return true, else 0 (out untouched). */
int ORC_FN(convert)(const uint8_t *s, size_t n, int type, uint64_t *out8);

/* A9  deletelchr / deleterchr / deletelrchr on a byte string (public/_public.cpp:24-75);
 * in place, returns the new length. which: 0=l, 1=r, 2=lr */
size_t ORC_FN(trim)(uint8_t *s, size_t n, int cc, int which);

/* A4  xmltodb path: readline("<endl/>"-style) + splitbuffer (tools/xmltodb.cpp:263-268,
 * 492-522) over a whole buffer.  carry=1 is splitbuffer itself (temp carried over missing
 * tags); carry=0 treats every column as an independent getxmlbuffer call (cell empty and
 * found bit clear when the tag is missing).  typed[c] selects the conversion applied to
 * the cell (may be NULL: number->F64, date->I64, char->NONE). */
orc_batch *ORC_FN(parse_xml)(const uint8_t *data, size_t n, const char *endbz,
                             const orc_col *cols, int ncols, const int *typed, int carry);

/* same, but reading the file through the reference's own file reader */
orc_batch *ORC_FN(parse_xml_file)(const char *path, const char *endbz, const orc_col *cols,
                                  int ncols, const int *typed, int carry);

/* A8  delimited path: readline("") + ccmdstr::splittocmd + getvalue (public/_public.cpp:
 * 274-446).  nfields columns; field f of each line clipped to ilen[f] bytes when >0,
 * converted by typed[f].  Lines with fewer fields leave the cell empty / found bit clear. */
orc_batch *ORC_FN(parse_csv)(const uint8_t *data, size_t n, const char *sep, int bdelspace,
                             int nfields, const int *ilen, const int *typed);

/* N1  writer side: the record formatter of tools/dminingoracle.cpp:196-256.  For every row the
 * fields are written as "<name>value</name>" (value = the cell as a C string: cut at its first
 * NUL, and at fieldlen[c] bytes when fieldlen[c]>0, which is what the fetch buffer of
 * sqlstatement::bindout(pos,string&,len) can hold, public/db/oracle/_ooci.cpp:622-627), then
 * "<endl/>\n"; a file starts with "<data>\n", ends with "</data>\n" and is closed after every
 * maxcount rows (maxcount 0: one file).  All files are laid back to back into out (NULL to
 * size only); file k is [file_off[k], file_off[k+1]) (file_off may be NULL, else holds
 * *nfiles+1 entries, at most filecap).  Returns the total length, or -1 where the reference
 * fails: no rows and maxcount 0 (closeandrename on a file never opened, :241-248). */
int64_t ORC_FN(format_xml)(const char *const *fieldname, const int *fieldlen, int nfields,
                           const uint32_t *const *off, const uint8_t *const *bytes, int64_t rows,
                           int64_t maxcount, uint8_t *out, uint64_t *file_off, int64_t filecap,
                           int64_t *nfiles);

/* same, written to disk the way cofile does (name + ".tmp", renamed on close; missing
 * directories created; public/_public.cpp:894-940) under the names of
 * tools/dminingoracle.cpp:204-205, "%s/%s_%s_%s_%d.xml" with the time stamp passed in and the
 * sequence number starting at first_seq.  Returns the number of files, -1 on failure. */
int64_t ORC_FN(format_xml_files)(const char *const *fieldname, const int *fieldlen, int nfields,
                                 const uint32_t *const *off, const uint8_t *const *bytes,
                                 int64_t rows, int64_t maxcount, const char *outpath,
                                 const char *bfilename, const char *timestamp,
                                 const char *efilename, int first_seq);

/* N2  routing of a file to its table and the SQL text, tools/xmltodb.cpp:102-129, :311-419.
 * orc_xmltotable mirrors st_xmltotable (tools/xmltodb.cpp:35-41). */
typedef struct orc_xmltotable {
    char filename[128];
    char tname[32];
    int uptbz;
    char execsql[256];
} orc_xmltotable;

/* N4  timetostr(ttime,strtime,fmt), public/_public.cpp:631-722, in the process's time zone:
 * the text into out (cap bytes, NUL-terminated), its length returned; -1 for a text that is
 * none of the formats (the reference leaves the string as it is) */
int ORC_FN(timetostr)(int64_t t, const char *fmt, char *out, size_t cap);

/* matchstr(str,rules), public/_public.cpp:213-263: 1 = match */
int ORC_FN(matchstr)(const char *str, const char *rules);

/* loadxmltotable over a buffer (tools/xmltodb.cpp:102-129): writes at most cap entries and
 * returns the number of records.  A value that does not fit its array (undefined behaviour in
 * the reference, which copies it whole) is cut to the array size minus one. */
int64_t ORC_FN(load_xmltotable)(const uint8_t *data, size_t n, orc_xmltotable *out, int64_t cap);

/* crtsql (tools/xmltodb.cpp:325-419): NUL-terminated statement texts into ins / upd (cut to the
 * capacities); upd is written only when e->uptbz == 1.  The key columns of the where part are
 * taken by ascending pkseq (the order in which the catalogue query fills m_vpkcols).  Returns
 * 1 when an update statement was written, else 0. */
int ORC_FN(crtsql)(const orc_xmltotable *e, const orc_col *cols, int ncols, char *ins, size_t inscap,
                   char *upd, size_t updcap);

/* bind order of preparesql (tools/xmltodb.cpp:421-477): schema column of every placeholder of
 * the insert and (uptbz == 1 only) the update statement; returns nins, *nupd the other count */
int ORC_FN(bind_plan)(const orc_xmltotable *e, const orc_col *cols, int ncols, int *ins, int *upd,
                      int *nupd);

void ORC_FN(batch_free)(orc_batch *b);

/* name of the implementation behind this library */
const char *ORC_FN(kind)(void);

#ifdef __cplusplus
}
#endif
#endif
