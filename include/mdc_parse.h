/*
 * mdc_parse.h -- C ABI of the MI355X bulk record parser (libmdc_parse.so).
 *
 * Drop-in boundary for the one data-parallel path of the MDC toolkit: the xmltodb ingest
 * loop (reference tools/xmltodb.cpp:263-304) and the delimited twin built on ccmdstr.  The
 * reference has no FFI; the per-record C++ calls this library replaces, in bulk, are cited
 * on every entry point (paths relative to the reference root).  Plain pointers and sizes
 * only; no C++ or torch types cross this boundary.
 *
 * Model: a *parser* binds a schema to one GPU and owns reusable device workspace; parsing a
 * buffer (already in HBM, or on the host, or a file) yields a *batch*: for every column a
 * compact byte arena + u32 offsets (cell r = bytes[off[r]..off[r+1])), optional typed
 * values, and per-row valid/found bit masks.  Batches live in device memory; host mirrors
 * are made on request.  Every function returns MDC_OK (0) or a negative MDC_ERR_* code,
 * except where a reference file-level code is mirrored (MDC_FILE_OPEN_FAILED == 5,
 * tools/xmltodb.cpp:256-261).  Nothing throws and no output is touched on failure.
 *
 * Thread safety: a parser (and the batches it produced) may be used by one thread at a
 * time; different parsers are independent.  The HIP runtime is initialised lazily by the
 * first call that needs the device, never at load time (xmltodb closes fds 0-63 in
 * main() before any work: tools/xmltodb.cpp:80).
 */
#ifndef MDC_PARSE_H
#define MDC_PARSE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- schema ------------------------------------------------------------------------- */

/* One table column; same layout and meaning as ctcols::st_col (tools/_tools.h:21-27) after
 * the type normalisation of ctcols::allcols (tools/_tools.cpp:47-71): datatype is "char",
 * "date" (collen 14) or "number" (collen 22); collen<=0 means "no clip". */
typedef struct mdc_col {
    char colname[32];
    char datatype[32];
    int collen;
    int pkseq;
} mdc_col;

/* Typed conversion applied to a cell's text; each is the reference overload named
 * (public/_public.cpp:506-625 for getxmlbuffer, :339-446 for ccmdstr::getvalue). */
enum {
    MDC_T_NONE = 0,
    MDC_T_I32 = 1,  /* stoi(picknumber(s,true))       -> int64 slot (sign-extended)      */
    MDC_T_U32 = 2,  /* stoi(picknumber(s))            -> uint64 slot                     */
    MDC_T_I64 = 3,  /* stol(picknumber(s,true))                                          */
    MDC_T_U64 = 4,  /* stoul(picknumber(s))                                              */
    MDC_T_F64 = 5,  /* stod(picknumber(s,true,true))  -> IEEE-754 binary64 bits          */
    MDC_T_F32 = 6,  /* stof(picknumber(s,true,true))  -> binary32 bits in the low word   */
    MDC_T_BOOL = 7, /* toupper(s)=="TRUE"             -> 0/1                             */
    MDC_T_TIME = 8, /* strtotime(s) (public/_public.cpp:1009-1044): the digits of the text   */
                    /* must be exactly 14, yyyymmddhh24miss -> time_t in the int64 slot, as  */
                    /* mktime gives it (fields out of range carried over) for the zone's     */
                    /* standard offset, mdc_opts.tz_offset seconds east of UTC (strtotime    */
                    /* sets tm_isdst to 0, so daylight saving never moves the value); not    */
                    /* valid where strtotime returns -1 for the format                       */
    MDC_T_DEFAULT = -1 /* number -> F64, date -> I64, char -> NONE                       */
};

/* ---- status codes ------------------------------------------------------------------- */
enum {
    MDC_OK = 0,
    MDC_STREAM_END = 1,         /* mdc_stream_next: every file has been handed out             */
    MDC_FILE_OPEN_FAILED = 5,   /* same value as _xmltodb()'s "open failed" (xmltodb.cpp:260) */
    MDC_ERR_ARG = -1,           /* bad argument / unsupported schema (see mdc_last_error)     */
    MDC_ERR_NO_DEVICE = -2,     /* no usable HIP device: there is no CPU fallback             */
    MDC_ERR_HIP = -3,           /* a HIP runtime call failed                                  */
    MDC_ERR_NOMEM = -4,
    MDC_ERR_RECORD_TOO_LONG = -5, /* a record of 2 GiB or more (the reference's positions are ints) */
    MDC_ERR_TOO_LARGE = -6,     /* a column's cells exceed 4 GiB in one batch: split input    */
    MDC_ERR_NUMERIC = -7,       /* numeric text beyond the exact-conversion range            */
    MDC_ERR_IO = -8,
    MDC_ERR_INTERNAL = -9,
    MDC_ERR_BUSY = -10          /* mdc_parse_device_noalloc: another call is using the GPU's scratch array */
};

/* ---- options ------------------------------------------------------------------------ */
typedef struct mdc_opts {
    int device;          /* HIP device ordinal; -1 = device 0                               */
    const char *endbz;   /* record end mark before '\n' (cifile::readline's endbz,          */
                         /* public/_public.cpp:866); NULL = "<endl/>", "" = plain lines    */
    int carry;           /* 1 = splitbuffer semantics: a missing tag keeps the previous      */
                         /* column's value (xmltodb.cpp:494-518); 0 = independent            */
                         /* getxmlbuffer per column (missing tag: empty cell, found bit 0)  */
    const int *typed;    /* per column MDC_T_*; NULL = MDC_T_DEFAULT for every column       */
    unsigned flags;      /* MDC_F_*                                                          */
    int tz_offset;       /* MDC_T_TIME columns: the zone strtotime's mktime would run in,    */
                         /* seconds east of UTC (28800 for the reference's UTC+8 sites; the  */
                         /* standard offset of a zone with daylight saving)                  */
} mdc_opts;

enum {
    MDC_F_NO_RECOFF = 1u, /* do not materialise the record-offset array                     */
    MDC_F_RECORDS_ONLY = 4u, /* the record split alone (cifile::readline, public/_public.cpp:866-892, over a whole */
                          /* file): rows and record offsets are right, every cell is empty and no tag is looked  */
                          /* for; about twice as fast as a parse with a schema                                      */
    MDC_F_ONE_PASS = 2u   /* tagged input, from the parser's second parse on: size and write in ONE kernel.  The */
                          /* output arrays are sized from the previous parse with a margin and every tile      */
                          /* learns its place from the tiles in front of it inside the launch; an input that    */
                          /* does not fit is parsed again in two passes.  Same results; measured slower than    */
                          /* the two passes (2.2 ms against 1.4 ms for 1 GiB), so it is off unless asked for    */
};

typedef struct mdc_parser mdc_parser;
typedef struct mdc_batch mdc_batch;

/* Per-parse device timings in milliseconds (HIP events on the parser's stream). */
typedef struct mdc_timing {
    float count_ms;  /* sizing pass (tile kernel, count mode)        */
    float scan_ms;   /* tile-offset scan                              */
    float write_ms;  /* write pass (emit kernel, plus the full write kernel if it ran) */
    float total_ms;  /* first launch to last completion               */
    uint64_t bytes_in;
    uint64_t bytes_out; /* algorithmic output bytes actually written  */
    int64_t rows;
    int tiles;
    int full_write;  /* 1 = some tiles had no cell map and went through the full write kernel */
    int lookup_tiles; /* tagged input: tiles whose records did not all repeat the sampled record's tags */
                      /* and took the tag table instead (0 for delimited input)                         */
    int long_records; /* records longer than the look-behind window, parsed by the long-record kernel */
    int one_pass;     /* 1 = sizing and writing were one kernel (MDC_F_ONE_PASS); count_ms and scan_ms are 0 */
    int one_pass_redone; /* 1 = a one-pass parse was given up (an output array too small, a long record) and   */
                         /* the input parsed again in two passes: total_ms covers the second parse only         */
} mdc_timing;

/* ---- library ------------------------------------------------------------------------ */

const char *mdc_version(void);
/* number of usable HIP devices, or a negative MDC_ERR_* */
int mdc_device_count(void);
const char *mdc_strerror(int code);
/* text of the last error on the calling thread ("" if none) */
const char *mdc_last_error(void);

/* ---- parsers ------------------------------------------------------------------------ */

/* Replaces, per record: cifile::readline(buf,endbz) (public/_public.cpp:866-892) +
 * splitbuffer (tools/xmltodb.cpp:492-522), i.e. getxmlbuffer(xml,colname,temp,collen)
 * (public/_public.cpp:461-481) + picknumber(temp,temp,..) (public/_public.cpp:178-204) +
 * vcolvalue[i]=temp.c_str() for every column of the schema.  Limits (MDC_ERR_ARG beyond them; the
 * reference has none): column names without '>' and '\n', at most 256 columns, an end mark
 * (opts->endbz) of at most 255 bytes.  An end mark that holds a newline is taken unless two of its
 * occurrences overlap in the input (a newline inside one follows another: cifile::readline only matches
 * the mark within the record gathered so far, so which of the two ends a record is a left-to-right
 * decision over the whole file); such an input is refused with MDC_ERR_ARG by the parse, which looks for
 * it in a pass of its own.  Inputs with such a mark are parsed without the tag template. */
int mdc_parser_create_xml(const mdc_col *cols, int ncols, const mdc_opts *opts, mdc_parser **out);

/* Replaces, per line: cifile::readline(buf) + ccmdstr::splittocmd(buf,sep,bdelspace)
 * (public/_public.cpp:274-302) + getvalue(f,string&,ilen[f]) (:304-317) and the typed
 * getvalue overloads (:339-446) for fields 0..nfields-1.  sep has 1..255 bytes (up to four bytes
 * below 0x80 without '\n' are found with byte masks, other separators by a walk over each record).
 * opts->typed/ilen may be NULL. */
int mdc_parser_create_csv(const char *sep, int bdelspace, int nfields, const int *ilen,
                          const mdc_opts *opts, mdc_parser **out);

void mdc_parser_destroy(mdc_parser *p);

/* Parse n bytes that are already resident in device memory of the parser's GPU.  d_data
 * must be 16-byte aligned and stay valid until the call returns (the call synchronises the
 * parser's stream before returning). */
int mdc_parse_device(mdc_parser *p, const void *d_data, size_t n, mdc_batch **out);
/* Parse n bytes of host memory.  A buffer of 32 MiB or more that is not pinned is copied in 8 MiB
 * pieces by up to four host threads through pinned buffers (the ones file streams keep, see
 * mdc_stream_release_pinned) and uploaded piece by piece; a pinned or small one goes up in one copy. */
int mdc_parse_host(mdc_parser *p, const void *h_data, size_t n, mdc_batch **out);
/* Parse a whole file; MDC_FILE_OPEN_FAILED (5) when it cannot be opened. */
int mdc_parse_file(mdc_parser *p, const char *path, mdc_batch **out);
/* The same parse as mdc_parse_device for callers that cannot allocate device memory (code under
 * HIP graph capture set-up, tight serving loops, harnesses that forbid hipMalloc).  It has no
 * counterpart in the reference.  The call itself allocates and frees nothing on the device: the
 * outputs are arrays the caller made beforehand (device memory of the parser's GPU), and the
 * library's own workspace (per-tile counters, their prefixes, tile headers, the table of output
 * pointers) lies in a static device array of MDC_NOALLOC_SCRATCH_BYTES that every GPU the
 * library runs on holds until the process ends.
 *   limit    per 16 KiB of input, 16 x columns + 52 bytes must fit the array: about 140 MiB of
 *            input with 11 columns, 15 MiB with 128, 8 MiB with 256; beyond that MDC_ERR_TOO_LARGE
 *            (use mdc_parse_device, or parse the input in parts)
 *   outputs  per column: rows_cap + 1 u32 offsets, bytes_cap bytes of cells, and for a typed
 *            column rows_cap u64 values; rows_cap x mdc_batch_words_per_row words each of valid
 *            and found bits; optionally rows_cap + 1 u64 record offsets (d_rec_off NULL: none).
 *            More records than rows_cap, or a column with more bytes than its bytes_cap, gives
 *            MDC_ERR_TOO_LARGE before anything is written to the outputs
 *   results  *rows = records parsed; col_bytes (may be NULL) receives the ncols byte counts.
 *            The arrays hold what the batch of mdc_parse_device would hold, bit for bit
 *   outputs' alignment: d_off 4, d_val 8, d_bytes 16 bytes (MDC_ERR_ARG otherwise); every array may be
 *            sized exactly: nothing is stored behind rows + 1 offsets, rows values, col_bytes[c] bytes
 *   sharing  every call on one GPU uses that GPU's one scratch array: a call that arrives while
 *            another one is under way on the same GPU (two threads, two parsers) is refused with
 *            MDC_ERR_BUSY; calls on different GPUs do not meet.  Nothing in the array needs
 *            clearing between calls (a call writes every word it later reads)
 * It is the slower of the two entry points (no cell map is kept between the two passes). */
typedef struct mdc_out_col {
    uint32_t *d_off;
    uint8_t *d_bytes;
    uint64_t bytes_cap;
    uint64_t *d_val; /* typed columns only */
} mdc_out_col;
typedef struct mdc_outputs {
    int64_t rows_cap;
    const mdc_out_col *cols; /* one per column of the parser's schema */
    uint32_t *d_valid;
    uint32_t *d_found;
    uint64_t *d_rec_off; /* may be NULL */
} mdc_outputs;
#define MDC_NOALLOC_SCRATCH_BYTES (2u << 20)
int mdc_parse_device_noalloc(mdc_parser *p, const void *d_data, size_t n, const mdc_outputs *o, int64_t *rows,
                             uint64_t *col_bytes);
/* timings of the most recent successful parse on this parser */
int mdc_parser_last_timing(const mdc_parser *p, mdc_timing *t);

/* One-shot forms (create a parser, parse, destroy it). */
int mdc_parse_xml_buffer(const void *h_data, size_t n, const mdc_col *cols, int ncols,
                         const mdc_opts *opts, mdc_batch **out);
int mdc_parse_xml_file(const char *path, const mdc_col *cols, int ncols, const mdc_opts *opts,
                       mdc_batch **out);
int mdc_parse_csv_buffer(const void *h_data, size_t n, const char *sep, int bdelspace,
                         int nfields, const int *ilen, const mdc_opts *opts, mdc_batch **out);

/* ---- a list of files through one parser --------------------------------------------- */

/* Replaces the file loop of the reference's ingest (tools/xmltodb.cpp:147-218: for every file
 * of the sorted list: open, read, parse) for plain files and for files compressed by
 * tools/gzipfiles.cpp:59 (gzip, recognised by the magic bytes 1f 8b).  `threads` host threads
 * read and inflate the files ahead into pinned buffers; the upload of file k+1 runs on a copy
 * stream while file k is parsed.  The parser must outlive the stream and must not be used for
 * anything else while the stream is open.  mdc_parse_file reads gzip files the same way. */
typedef struct mdc_stream mdc_stream;
int mdc_stream_open(mdc_parser *p, const char *const *paths, int nfiles, int threads, mdc_stream **out);
/* The next file's batch, in list order; *index (may be NULL) receives its position in paths.
 * MDC_OK, MDC_STREAM_END after the last file, or the file's error (MDC_FILE_OPEN_FAILED,
 * MDC_ERR_IO, a parse error): the stream then goes on with the following file. */
int mdc_stream_next(mdc_stream *s, mdc_batch **out, int *index);
/* uncompressed bytes handed to the parser so far */
uint64_t mdc_stream_bytes(const mdc_stream *s);
void mdc_stream_close(mdc_stream *s);
/* A closed stream's pinned host buffers are kept for the next stream of the process (at most 2 GiB:
 * pinning memory costs about as much as reading a file into it, and the ingest opens a stream per
 * directory scan, tools/xmltodb.cpp:147-218).  This gives them back to the system. */
void mdc_stream_release_pinned(void);
/* A host buffer of at least `want` bytes in pinned memory: one of the buffers above when one fits, else
 * newly pinned; *cap receives its size.  Input read into such a buffer goes to the GPU in one copy
 * (mdc_parse_host).  mdc_host_buffer_put hands it back for the next user.  Host-side helpers without a
 * counterpart in the reference (its cifile reads through an ifstream, public/_public.cpp:828-838). */
int mdc_host_buffer_get(size_t want, void **h_buf, size_t *cap);
void mdc_host_buffer_put(void *h_buf, size_t cap);

/* ---- batches ------------------------------------------------------------------------ */

int64_t mdc_batch_rows(const mdc_batch *b);          /* totalcount of xmltodb.cpp:266 */
int mdc_batch_ncols(const mdc_batch *b);
int mdc_batch_words_per_row(const mdc_batch *b);     /* u32 mask words per row */

/* Device views (valid until mdc_batch_free). */
int mdc_batch_device_column(const mdc_batch *b, int col, const uint32_t **d_off,
                            const uint8_t **d_bytes, uint64_t *nbytes);
int mdc_batch_device_values(const mdc_batch *b, int col, const uint64_t **d_val);
int mdc_batch_device_masks(const mdc_batch *b, const uint32_t **d_valid, const uint32_t **d_found);
int mdc_batch_device_recoff(const mdc_batch *b, const uint64_t **d_rec_off);

/* Host mirrors (copied from the device on first use; owned by the batch). */
int mdc_batch_host_column(mdc_batch *b, int col, const uint32_t **off, const uint8_t **bytes,
                          uint64_t *nbytes);
int mdc_batch_host_values(mdc_batch *b, int col, const uint64_t **val);
int mdc_batch_host_masks(mdc_batch *b, const uint32_t **valid, const uint32_t **found);
int mdc_batch_host_recoff(mdc_batch *b, const uint64_t **rec_off);
/* cell (row,col) on the host: what vcolvalue[col] holds after splitbuffer for record row */
int mdc_batch_cell(mdc_batch *b, int64_t row, int col, const char **p, uint32_t *len);
/* Column as fixed-stride NUL-terminated slots, stride = max(cell length in the batch)+1 or
 * collen+1 when the schema has collen>0: the layout sqlstatement::bindin(pos,string&,len)
 * binds (public/db/oracle/_ooci.cpp:568-573), ready for an array bind.  Host memory owned
 * by the batch.  A cell longer than collen is cut to collen bytes in its slot: that happens only
 * to a value carried over from an earlier, wider column (tools/xmltodb.cpp:494-518), where the
 * reference assigns past the buffer it has bound (undefined there); the compact column
 * (mdc_batch_host_column) keeps such a cell whole. */
int mdc_batch_column_fixed(mdc_batch *b, int col, const char **base, size_t *stride);

void mdc_batch_free(mdc_batch *b);

/* ---- device memory helpers (so callers without a HIP binding can stage inputs) ------ */
int mdc_device_alloc(int device, size_t n, void **d_ptr);
int mdc_device_free(int device, void *d_ptr);
int mdc_device_upload(int device, void *d_dst, const void *h_src, size_t n);
int mdc_device_download(int device, void *h_dst, const void *d_src, size_t n);
int mdc_device_synchronize(int device);

#ifdef __cplusplus
}
#endif
#endif
