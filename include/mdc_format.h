/*
 * mdc_format.h -- C ABI of the writer side of libmdc_parse.so: columns in HBM -> the text of
 * the "<name>value</name>...<endl/>\n" files that the parser of mdc_parse.h reads.
 *
 * Replaces, in bulk, the fetch loop of the reference's extractor
 * (tools/dminingoracle.cpp:196-256): per row one cofile::writeline("<%s>%s</%s>", ...) per
 * field and writeline("<endl/>\n"), files framed by "<data>\n" / "</data>\n" and closed after
 * every maxcount rows, written as name+".tmp" and renamed (cofile, public/_public.cpp:894-940).
 * Paths are relative to the reference root.  Plain pointers and sizes only.
 *
 * Same conventions as mdc_parse.h: MDC_OK (0) or a negative MDC_ERR_*; nothing throws; no
 * output is touched on failure; no CPU implementation of the formatting exists in the
 * library (MDC_ERR_NO_DEVICE without a usable GPU).
 */
#ifndef MDC_FORMAT_H
#define MDC_FORMAT_H

#include "mdc_parse.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mdc_formatter mdc_formatter;
typedef struct mdc_text mdc_text;

/* The field list of the extractor: fieldname[i] / fieldlen[i] are the i-th entries of its
 * <fieldstr> and <fieldlen> arguments (tools/dminingoracle.cpp:432-436).  A value is written
 * as the C string the reference prints with "%s": cut at its first NUL byte, and at
 * fieldlen[i] bytes when fieldlen[i] > 0 (the fetch buffer of
 * sqlstatement::bindout(pos,string&,len) holds len bytes, public/db/oracle/_ooci.cpp:622-627).
 * fieldlen may be NULL (no cut).  1..256 fields, names of 1..255 bytes, at most 12 KiB of tag
 * text (2 x name + 5 bytes per field) in all. */
int mdc_formatter_create(const char *const *fieldname, const int *fieldlen, int nfields, int device,
                         mdc_formatter **out);
void mdc_formatter_destroy(mdc_formatter *f);

/* Format `rows` rows.  Column c is given as the parser's batches hold it: d_off[c] points to
 * rows+1 ascending u32 offsets into the byte arena d_bytes[c], both in device memory of the
 * formatter's GPU (the two pointer arrays themselves are host arrays of nfields entries).
 * maxcount > 0 closes a file after every maxcount rows, 0 writes one file
 * (tools/dminingoracle.cpp:222, :241).  No rows give a text of no files, as the reference
 * opens a file only when the first row arrives (:202). */
int mdc_format_device(mdc_formatter *f, const uint32_t *const *d_off, const uint8_t *const *d_bytes,
                      int64_t rows, int64_t maxcount, mdc_text **out);
/* the same for columns in host memory (copied to the device first) */
int mdc_format_host(mdc_formatter *f, const uint32_t *const *off, const uint8_t *const *bytes,
                    int64_t rows, int64_t maxcount, mdc_text **out);
/* the same for the columns of a parsed batch, which stay in HBM: the batch must have as many
 * columns as the formatter has fields and live on the same GPU */
int mdc_format_batch(mdc_formatter *f, const mdc_batch *b, int64_t maxcount, mdc_text **out);

/* Device timings of the most recent successful format call, in milliseconds. */
typedef struct mdc_format_timing {
    float format_ms;    /* the formatting kernel                                             */
    float total_ms;     /* first launch to last completion, measuring passes included         */
    uint64_t bytes_in;  /* cell bytes + offsets read                                          */
    uint64_t bytes_out; /* text written                                                       */
    int exact_pass;     /* 1 = some value had a NUL byte or exceeded its fieldlen: lengths    */
                        /* were measured in a separate pass before formatting                 */
} mdc_format_timing;
int mdc_formatter_last_timing(const mdc_formatter *f, mdc_format_timing *t);

/* ---- the text: every file back to back ---------------------------------------------- */
int64_t mdc_text_files(const mdc_text *t);
uint64_t mdc_text_bytes(const mdc_text *t);
/* file k is [*begin, *end) of the text */
int mdc_text_file_range(const mdc_text *t, int64_t k, uint64_t *begin, uint64_t *end);
/* device view, valid until mdc_text_free */
int mdc_text_device(const mdc_text *t, const uint8_t **d_data, uint64_t *nbytes);
/* host mirror, copied from the device on first use and owned by the text */
int mdc_text_host(mdc_text *t, const uint8_t **data, uint64_t *nbytes);

/* Write the files as "<outpath>/<bfilename>_<timestamp>_<efilename>_<seq>.xml" with seq counting
 * up from first_seq (tools/dminingoracle.cpp:204-205; the reference starts at 1), each as
 * name+".tmp" renamed when complete, creating missing directories (cofile::open,
 * public/_public.cpp:894-918).  timestamp NULL = the local time now as yyyymmddhh24miss; the
 * reference takes the time when it opens each file, here one stamp names every file of the
 * call.  *nwritten (may be NULL) receives the number of files.  A text of no files writes
 * nothing and returns MDC_OK when it was formatted with maxcount > 0, MDC_ERR_IO when with
 * maxcount 0: there the reference closes a file it never opened and fails (:241-248). */
int mdc_text_write_files(mdc_text *t, const char *outpath, const char *bfilename, const char *timestamp,
                         const char *efilename, int first_seq, int *nwritten);
void mdc_text_free(mdc_text *t);

/* ---- times as text ---------------------------------------------------------------------
 * timetostr(ttime, strtime, fmt) of the reference (public/_public.cpp:631-722) over a column of
 * time_t values in device memory: row i becomes the text the reference prints for d_time[i] in
 * format fmt, written as a column in the layout of the parser's batches and of
 * mdc_format_device: d_off receives rows+1 u32 offsets (i x width), d_bytes rows x width bytes,
 * both caller-allocated device memory on `device` (mdc_device_alloc).  fmt is one of the
 * reference's formats: "" or "yyyy-mm-dd hh24:mi:ss", "yyyy-mm-dd hh24:mi", "yyyy-mm-dd hh24",
 * "yyyy-mm-dd", "yyyy-mm", "yyyymmddhh24miss", "yyyymmddhh24mi", "yyyymmddhh24", "yyyymmdd",
 * "hh24miss", "hh24mi", "hh24", "mi"; for any other text the reference leaves the string as it
 * is, here it is MDC_ERR_ARG.  The reference's localtime_r runs in the process's time zone;
 * here the zone is one without daylight saving, tz_offset seconds east of UTC (as for
 * MDC_T_TIME columns, whose values this call turns back into text).  A time whose local year
 * lies outside 0000..9999 (the reference prints more than four digits, or the year as an
 * unsigned number) gives MDC_ERR_NUMERIC and unspecified output. */
int mdc_timetostr_device(int device, const int64_t *d_time, int64_t rows, const char *fmt, int tz_offset,
                         uint32_t *d_off, uint8_t *d_bytes);
/* bytes of the text a format gives (19 for the default), or MDC_ERR_ARG for a text that is none
 * of the formats */
int mdc_timetostr_width(const char *fmt);

#ifdef __cplusplus
}
#endif
#endif
