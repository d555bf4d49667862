/*
 * mdc_route.h -- host side around the parse: which table a file goes to, the SQL text of its
 * insert and update statements, and the order in which a parsed batch's columns are bound.
 *
 * These are the callers of the record parser in the reference's ingest (tools/xmltodb.cpp):
 * loadxmltotable (:102-129), findxmltotable (:311-323) with matchstr
 * (public/_public.cpp:213-263), crtsql (:325-419) and the bind order of preparesql (:421-477).
 * Pure host code in libmdc_parse.so: no GPU is needed and none is touched.  The schema comes
 * from the caller (mdc_col, i.e. ctcols::st_col) instead of the USER_TAB_COLUMNS queries of
 * tools/_tools.cpp:11-110, so an ingest loop can run without the database catalogue.
 *
 * Every function returns MDC_OK (0) or a negative MDC_ERR_* code unless stated otherwise;
 * text of the last error: mdc_last_error().
 */
#ifndef MDC_ROUTE_H
#define MDC_ROUTE_H

#include <stddef.h>
#include <stdint.h>

#include "mdc_parse.h"

#ifdef __cplusplus
extern "C" {
#endif

/* One routing entry; same fields and sizes as st_xmltotable (tools/xmltodb.cpp:35-41). */
typedef struct mdc_xmltotable {
    char filename[128]; /* match rules for the file name, separated by commas      */
    char tname[32];     /* target table                                            */
    int uptbz;          /* 1 = update the row when the insert meets a duplicate    */
    char execsql[256];  /* statement executed before the file is loaded ("" = none)*/
} mdc_xmltotable;

typedef struct mdc_route mdc_route;

/* loadxmltotable (tools/xmltodb.cpp:102-129): every record of the parameter file
 * (cifile::readline(buffer,"<endl/>")) gives one entry: filename, tname and execsql by the
 * char* getxmlbuffer (public/_public.cpp:483-504, no length limit passed), uptbz by the int
 * one (:519-535); a missing tag or a failed conversion leaves the zeroed field.  The reference
 * copies a value of any length into the fixed arrays (undefined behaviour beyond their size);
 * here such a value is cut to the array size minus one and counted in
 * mdc_route_truncated().  MDC_FILE_OPEN_FAILED when the file cannot be opened (the reference
 * returns false, and the program ends: :107-111, :141). */
int mdc_route_load(const char *inifilename, mdc_route **out);
/* the same from memory */
int mdc_route_load_buffer(const void *data, size_t n, mdc_route **out);
void mdc_route_free(mdc_route *r);
int mdc_route_count(const mdc_route *r);
int mdc_route_truncated(const mdc_route *r);
/* entry i (owned by the route), NULL when i is out of range */
const mdc_xmltotable *mdc_route_entry(const mdc_route *r, int i);

/* findxmltotable (tools/xmltodb.cpp:311-323): index of the first entry whose rules match the
 * file name, -1 when none does (the reference's return code 1, "no routing entry"). */
int mdc_route_find(const mdc_route *r, const char *xmlfile);

/* matchstr(str,rules) (public/_public.cpp:213-263): 1 = match, 0 = no match.  Rules are
 * separated by ',', '*' stands for any run of bytes, comparison ignores ASCII case, an empty
 * rule never matches, "*" alone matches everything. */
int mdc_matchstr(const char *str, const char *rules);

/* crtsql (tools/xmltodb.cpp:325-419): the text of the insert statement and, when
 * entry->uptbz == 1, of the update statement (else updatesql is left as it was, as the
 * reference leaves its global).  Kept as the reference writes them, including the trailing
 * comma that deleterchr leaves when the list is a single one-byte name
 * (public/_public.cpp:52-59) and the quote missing after the date format in the set part
 * (:393).  *inslen / *updlen receive the full lengths (without the NUL); text beyond the
 * capacities is cut (NUL-terminated) and MDC_ERR_TOO_LARGE returned. */
int mdc_crtsql(const mdc_xmltotable *entry, const mdc_col *cols, int ncols, char *insertsql,
               size_t inscap, size_t *inslen, char *updatesql, size_t updcap, size_t *updlen);

/* Bind order of preparesql (tools/xmltodb.cpp:421-477): ins[k] = index of the schema column
 * bound to placeholder :k+1 of the insert statement (every column but upttime and keyid);
 * upd[k] = the same for the update statement (first the non-key columns but keyid and
 * upttime, then the key columns in schema order), filled only when entry->uptbz == 1 (else
 * *nupd = 0).  Both arrays need room for ncols entries.  Each bound column is
 * mdc_batch_column_fixed(batch, col, ...): slots of collen+1 bytes, NUL-terminated, which is
 * what sqlstatement::bindin(pos,string&,len) hands to OCI (public/db/oracle/_ooci.cpp:568-573)
 * for one row; with the slot stride as the skip the same buffer serves an array bind. */
int mdc_bind_plan(const mdc_xmltotable *entry, const mdc_col *cols, int ncols, int *ins, int *nins,
                  int *upd, int *nupd);

/* ---- the execute loop of a file in bulk (tools/xmltodb.cpp:263-306) ----------------------
 *
 * The reference runs, per record: insert; on ORA-00001 (sqlstatement::rc() == 1) and uptbz == 1
 * the update; any other failure is logged and the loop goes on, except ORA-03113, 03114, 03135
 * and 16014 (the connection is gone), which end the file with code 2 before the commit; after the
 * last record: commit, code 0.  With array binds the same decisions are taken once per file:
 * one array insert of every row, the rows it rejected as duplicates gathered (in row order) into
 * the update statement's bind order, one array update of those, then the commit by the caller.
 * The database ends in the same state: a row's update still follows its own insert attempt and
 * precedes the update of any later row, and a lost connection leaves nothing committed. */

/* One bound column as fixed-stride NUL-terminated slots (see mdc_batch_column_fixed). */
typedef struct mdc_bind_slab {
    const char *base;
    size_t stride;
    int col; /* index of the schema column */
    int pad;
} mdc_bind_slab;

/* The slabs of a parsed batch in the bind order of the two statements (mdc_bind_plan): ins[k] is
 * bound to :k+1 of the insert, upd[k] to :k+1 of the update (filled when entry->uptbz == 1, else
 * *nupd = 0).  Both arrays need room for ncols entries; the memory belongs to the batch. */
int mdc_batch_bind_slabs(mdc_batch *b, const mdc_xmltotable *entry, const mdc_col *cols, int ncols,
                         mdc_bind_slab *ins, int *nins, mdc_bind_slab *upd, int *nupd);

/* The caller's array execute (OCIStmtExecute with iters = nrows in batch-error mode, or a loop of
 * single executes): statement 0 = insert, 1 = update; slabs[k] row r is the text bound to :k+1 for
 * that row; rowcode[r] receives 0 for a row that went through, else the database's error number
 * (1 = ORA-00001).  A nonzero return means the statement could not be executed at all. */
typedef int (*mdc_array_exec)(void *ctx, int statement, const mdc_bind_slab *slabs, int nslabs, int64_t nrows,
                              int *rowcode);

typedef struct mdc_ingest_counts {
    int64_t totalcount; /* records of the file (tools/xmltodb.cpp:266)                        */
    int64_t inscount;   /* inserted (:304)                                                     */
    int64_t uptcount;   /* updated after a duplicate (:287)                                    */
    int64_t insfailed;  /* insert failed for another reason (the reference logs them, :291-295) */
    int64_t uptfailed;  /* update failed (logged, :281-286)                                    */
    int64_t duplicates; /* inserts rejected with ORA-00001                                     */
    int result;         /* _xmltodb()'s code: 0 = commit the file, 2 = database error, do not  */
    int fatalcode;      /* the error number that ended the file when result == 2               */
} mdc_ingest_counts;

/* Runs the loop over `rows` rows given as slabs in both bind orders (upd may be NULL when
 * uptbz != 1).  Returns MDC_OK whenever the decisions could be taken (look at counts->result);
 * the commit or rollback stays with the caller, who owns the connection.
 *
 * Order of the statements: the reference executes row by row, a rejected row's update before the
 * next row's insert (tools/xmltodb.cpp:271-303); here ALL rows are inserted with one array execute and
 * the rows rejected with ORA-00001 are then updated with a second one.  The table ends in the same
 * state, and the counts are the same, when the primary key is the table's only unique constraint
 * and no trigger or foreign key makes a row's outcome depend on the rows before it (the reference's
 * own tables are of that kind: note.txt).  Where that does not hold (a second unique index: ORA-00001
 * from it is answered by an update through the primary key, which may match no row; a trigger that
 * reads earlier rows) keep the per-row order by calling this function with runs of rows, down to one
 * row a call.  After a lost connection (result == 2) rows behind the fatal one have been sent to the
 * database as part of the same array execute, and uptcount is 0: the caller rolls the file back, as
 * the reference does (tools/xmltodb.cpp:297-301). */
int mdc_ingest_rows(const mdc_bind_slab *ins, int nins, const mdc_bind_slab *upd, int nupd, int64_t rows, int uptbz,
                    mdc_array_exec exec, void *ctx, mdc_ingest_counts *counts);

#ifdef __cplusplus
}
#endif
#endif
