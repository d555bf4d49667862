/*
 * ref_harness.cpp -- wraps the reference's own public/_public.cpp (compiled from where it
 * lies under the reference tree, never copied) behind the C interface of mdc_oracle.h with
 * the prefix ref_.
 *
 * TEST INFRASTRUCTURE ONLY: validates oracle/mdc_oracle.c, generates tests/golden/ and is
 * the "reference" CPU baseline.  Built by oracle/Makefile into oracle/_ref/ (git-ignored).
 *
 * Everything that computes a result here is a call into namespace idc of the reference:
 * cifile::readline, getxmlbuffer (all overloads), picknumber, ccmdstr, deletel/r/lrchr.
 * The only restated code are loops of the tools that need Oracle OCI headers and libclntsh,
 * which this image does not have: the 12-line column loop of splitbuffer
 * (tools/xmltodb.cpp:492-522), the fetch loop of tools/dminingoracle.cpp:196-256, and
 * loadxmltotable / crtsql / preparesql (tools/xmltodb.cpp:102-129, :325-477), each around the
 * reference's own functions (readline, getxmlbuffer, cofile, sformat, deleterchr, matchstr).
 */
#include "_public.h" /* -I<reference>/public */

#define ORC_PREFIX ref_
#include "mdc_oracle.h"
#include "orc_builder.h"

#include <unistd.h>

#include <algorithm>
#include <vector>

using namespace idc;

namespace {

std::string write_tmp(const uint8_t *data, size_t n) {
    const char *dir = getenv("TMPDIR");
    std::string path = std::string(dir && *dir ? dir : "/tmp") + "/mdc_ref_XXXXXX";
    int fd = mkstemp(&path[0]);
    if (fd < 0) return "";
    size_t off = 0;
    while (off < n) {
        ssize_t w = write(fd, data + off, n - off);
        if (w <= 0) break;
        off += (size_t)w;
    }
    close(fd);
    return path;
}

/* a tag name that occurs in s neither as <name> nor as </name> */
std::string free_tag(const std::string &s) {
    for (int i = 0;; ++i) {
        std::string nm = "q" + std::to_string(i);
        if (s.find("<" + nm + ">") == std::string::npos && s.find("</" + nm + ">") == std::string::npos)
            return nm;
    }
}

void default_typed_cols(const orc_col *cols, int ncols, const int *typed, int *out) {
    for (int c = 0; c < ncols; ++c) {
        if (typed)
            out[c] = typed[c];
        else if (strcmp(cols[c].datatype, "number") == 0)
            out[c] = ORC_T_F64;
        else if (strcmp(cols[c].datatype, "date") == 0)
            out[c] = ORC_T_I64;
        else
            out[c] = ORC_T_NONE;
    }
}

/* the typed overload of the reference applied to a text: wrap it in a tag the text does not
 * contain and call getxmlbuffer(xml, tag, T&) (public/_public.cpp:506-625) */
bool convert_text(const std::string &text, int type, uint64_t *out8) {
    std::string tag = free_tag(text);
    std::string xml = "<" + tag + ">" + text + "</" + tag + ">";
    switch (type) {
    case ORC_T_I32: {
        int v = 0;
        if (!getxmlbuffer(xml, tag, v)) return false;
        *out8 = (uint64_t)(int64_t)v;
        return true;
    }
    case ORC_T_U32: {
        unsigned int v = 0;
        if (!getxmlbuffer(xml, tag, v)) return false;
        *out8 = (uint64_t)v;
        return true;
    }
    case ORC_T_I64: {
        long v = 0;
        if (!getxmlbuffer(xml, tag, v)) return false;
        *out8 = (uint64_t)v;
        return true;
    }
    case ORC_T_U64: {
        unsigned long v = 0;
        if (!getxmlbuffer(xml, tag, v)) return false;
        *out8 = (uint64_t)v;
        return true;
    }
    case ORC_T_F64: {
        double v = 0;
        if (!getxmlbuffer(xml, tag, v)) return false;
        memcpy(out8, &v, 8);
        return true;
    }
    case ORC_T_F32: {
        float v = 0;
        if (!getxmlbuffer(xml, tag, v)) return false;
        uint32_t b;
        memcpy(&b, &v, 4);
        *out8 = b;
        return true;
    }
    case ORC_T_BOOL: {
        bool v = false;
        if (!getxmlbuffer(xml, tag, v)) return false;
        *out8 = v ? 1 : 0;
        return true;
    }
    case ORC_T_TIME: {
        /* strtotime (public/_public.cpp:1009-1044) in the process's time zone; its -1 for a text
         * without fourteen digits is told from a time of -1 by the digit count */
        time_t t = strtotime(text);
        if (t == (time_t)-1 && picknumber(text, false, false).length() != 14) return false;
        *out8 = (uint64_t)(int64_t)t;
        return true;
    }
    default:
        return false;
    }
}

} // namespace

extern "C" {

int64_t ref_split_records(const uint8_t *data, size_t n, const char *endbz, uint64_t *start,
                          uint64_t *len, int64_t cap) {
    std::string path = write_tmp(data, n);
    if (path.empty()) return -1;
    cifile ifile;
    int64_t count = 0;
    if (ifile.open(path)) {
        std::string buf;
        uint64_t pos = 0;
        while (ifile.readline(buf, endbz ? endbz : "")) {
            if (count < cap) {
                if (start) start[count] = pos;
                if (len) len[count] = buf.length();
            }
            pos += buf.length() + 1;
            ++count;
        }
        ifile.close();
    }
    unlink(path.c_str());
    return count;
}

int ref_getxmlbuffer(const uint8_t *rec, size_t reclen, const char *name, int ilen, uint8_t *out,
                     size_t outcap, size_t *outlen) {
    std::string xml((const char *)rec, reclen), value;
    if (!getxmlbuffer(xml, name, value, ilen)) return 0;
    if (out && outcap) memcpy(out, value.data(), value.length() < outcap ? value.length() : outcap);
    if (outlen) *outlen = value.length();
    return 1;
}

size_t ref_picknumber(const uint8_t *src, size_t n, uint8_t *dst, int bsigned, int bdot) {
    std::string s((const char *)src, n), d;
    picknumber(s, d, bsigned != 0, bdot != 0);
    memcpy(dst, d.data(), d.length());
    return d.length();
}

int ref_convert(const uint8_t *s, size_t n, int type, uint64_t *out8) {
    return convert_text(std::string((const char *)s, n), type, out8) ? 1 : 0;
}

size_t ref_trim(uint8_t *s, size_t n, int cc, int which) {
    std::string str((const char *)s, n);
    if (which == 0)
        deletelchr(str, cc);
    else if (which == 1)
        deleterchr(str, cc);
    else
        deletelrchr(str, cc);
    memcpy(s, str.data(), str.length());
    return str.length();
}

orc_batch *ref_parse_xml_file(const char *path, const char *endbz, const orc_col *cols, int ncols,
                              const int *typed, int carry) {
    std::vector<int> ty(ncols + 1);
    default_typed_cols(cols, ncols, typed, ty.data());
    cifile ifile;
    if (!ifile.open(path)) return nullptr;
    builder *B = builder_new(ncols, ty.data());
    orc_batch *b = &B->b;

    std::vector<std::string> vcolvalue(ncols); /* as preparesql(), tools/xmltodb.cpp:424 */
    std::string xmlbuffer;
    uint64_t pos = 0;
    int64_t r = 0;
    while (ifile.readline(xmlbuffer, endbz ? endbz : "")) { /* tools/xmltodb.cpp:264 */
        while (r >= B->cap_rows) builder_grow_rows(B);
        b->rec_off[r] = pos;
        pos += xmlbuffer.length() + 1;
        b->rec_off[r + 1] = pos;

        std::string temp; /* splitbuffer, tools/xmltodb.cpp:494 */
        for (int i = 0; i < ncols; ++i) {
            bool found;
            if (carry) {
                found = getxmlbuffer(xmlbuffer, cols[i].colname, temp, cols[i].collen);
            } else {
                temp.clear();
                found = getxmlbuffer(xmlbuffer, cols[i].colname, temp, cols[i].collen);
            }
            if (strcmp(cols[i].datatype, "date") == 0)
                picknumber(temp, temp, false, false);
            else if (strcmp(cols[i].datatype, "number") == 0)
                picknumber(temp, temp, true, true);
            if (carry)
                vcolvalue[i] = temp.c_str();
            else
                vcolvalue[i] = temp;

            if (found) b->found[r * b->words_per_row + (i >> 5)] |= 1u << (i & 31);
            builder_put_cell(B, r, i, (const uint8_t *)vcolvalue[i].data(), vcolvalue[i].length());
            if (ty[i] != ORC_T_NONE && (carry || found)) {
                uint64_t v = 0;
                if (convert_text(vcolvalue[i], ty[i], &v)) {
                    b->val[i][r] = v;
                    b->valid[r * b->words_per_row + (i >> 5)] |= 1u << (i & 31);
                }
            }
        }
        ++r;
    }
    b->rows = r;
    return b;
}

orc_batch *ref_parse_xml(const uint8_t *data, size_t n, const char *endbz, const orc_col *cols,
                         int ncols, const int *typed, int carry) {
    std::string path = write_tmp(data, n);
    if (path.empty()) return nullptr;
    orc_batch *b = ref_parse_xml_file(path.c_str(), endbz, cols, ncols, typed, carry);
    unlink(path.c_str());
    return b;
}

orc_batch *ref_parse_csv(const uint8_t *data, size_t n, const char *sep, int bdelspace, int nfields,
                         const int *ilen, const int *typed) {
    std::string path = write_tmp(data, n);
    if (path.empty()) return nullptr;
    std::vector<int> ty(nfields + 1);
    for (int c = 0; c < nfields; ++c) ty[c] = typed ? typed[c] : ORC_T_NONE;
    cifile ifile;
    if (!ifile.open(path)) {
        unlink(path.c_str());
        return nullptr;
    }
    builder *B = builder_new(nfields, ty.data());
    orc_batch *b = &B->b;
    std::string line, cell;
    ccmdstr cmd;
    uint64_t pos = 0;
    int64_t r = 0;
    while (ifile.readline(line)) {
        while (r >= B->cap_rows) builder_grow_rows(B);
        b->rec_off[r] = pos;
        pos += line.length() + 1;
        b->rec_off[r + 1] = pos;
        cmd.splittocmd(line, sep, bdelspace != 0);
        for (int c = 0; c < nfields; ++c) {
            cell.clear();
            bool found = cmd.getvalue(c, cell, ilen ? ilen[c] : 0);
            if (!found) cell.clear();
            if (found) b->found[r * b->words_per_row + (c >> 5)] |= 1u << (c & 31);
            builder_put_cell(B, r, c, (const uint8_t *)cell.data(), cell.length());
            if (found && ty[c] != ORC_T_NONE) {
                uint64_t v = 0;
                bool ok = convert_text(cell, ty[c], &v);
                if (!ilen || ilen[c] <= 0) {
                    /* unclipped: must agree with the ccmdstr::getvalue overload itself */
                    uint64_t w = 0;
                    bool ok2 = false;
                    switch (ty[c]) {
                    case ORC_T_I32: { int t = 0; ok2 = cmd.getvalue(c, t); w = (uint64_t)(int64_t)t; break; }
                    case ORC_T_U32: { unsigned int t = 0; ok2 = cmd.getvalue(c, t); w = t; break; }
                    case ORC_T_I64: { long t = 0; ok2 = cmd.getvalue(c, t); w = (uint64_t)t; break; }
                    case ORC_T_U64: { unsigned long t = 0; ok2 = cmd.getvalue(c, t); w = t; break; }
                    case ORC_T_F64: { double t = 0; ok2 = cmd.getvalue(c, t); memcpy(&w, &t, 8); break; }
                    case ORC_T_F32: { float t = 0; ok2 = cmd.getvalue(c, t); uint32_t u; memcpy(&u, &t, 4); w = u; break; }
                    case ORC_T_BOOL: { bool t = false; ok2 = cmd.getvalue(c, t); w = t ? 1 : 0; break; }
                    default: break;
                    }
                    if (ok != ok2 || (ok && v != w)) {
                        fprintf(stderr, "ref_parse_csv: getxmlbuffer/getvalue overloads disagree\n");
                        abort();
                    }
                }
                if (ok) {
                    b->val[c][r] = v;
                    b->valid[r * b->words_per_row + (c >> 5)] |= 1u << (c & 31);
                }
            }
        }
        ++r;
    }
    b->rows = r;
    ifile.close();
    unlink(path.c_str());
    return b;
}

/* The fetch loop of _dminingoracle (tools/dminingoracle.cpp:196-256) with the row source
 * replaced by columns in memory: dminingoracle.cpp itself needs the OCI headers.  The file
 * writing is the reference's own cofile (open / writeline / closeandrename) and sformat; the
 * restated part is the loop around them and the fetch of one row into fieldvalue[], which
 * stands in for OCIStmtFetch into the buffers of bindout(pos,string&,len)
 * (public/db/oracle/_ooci.cpp:622-627: resize(len), at most len bytes and a NUL). */
int64_t ref_format_xml_files(const char *const *fieldname, const int *fieldlen, int nfields,
                             const uint32_t *const *off, const uint8_t *const *bytes, int64_t rows,
                             int64_t maxcount, const char *outpath, const char *bfilename,
                             const char *timestamp, const char *efilename, int first_seq) {
    std::vector<std::string> fieldvalue(nfields);
    cofile ofile;
    std::string xmlfile;
    int iseq = first_seq;
    int64_t rpc = 0, nfiles = 0;

    for (int64_t r = 0; r < rows; ++r) {
        ++rpc; /* stmtsel.next()==0 fetched one more row */
        for (int i = 0; i < nfields; ++i) {
            size_t n = off[i][r + 1] - off[i][r];
            size_t cap = fieldlen && fieldlen[i] > 0 ? (size_t)fieldlen[i] : n;
            fieldvalue[i].assign(cap + 1, '\0');
            memcpy(&fieldvalue[i][0], bytes[i] + off[i][r], n < cap ? n : cap);
        }
        if (ofile.isopen() == false) {
            sformat(xmlfile, "%s/%s_%s_%s_%d.xml", outpath, bfilename, timestamp, efilename, iseq++);
            if (ofile.open(xmlfile) == false) return -1;
            ++nfiles;
            ofile.writeline("<data>\n");
        }
        for (int i = 0; i < nfields; ++i)
            ofile.writeline("<%s>%s</%s>", fieldname[i], fieldvalue[i].c_str(), fieldname[i]);
        ofile.writeline("<endl/>\n");
        if ((maxcount > 0) && (rpc % maxcount == 0)) {
            ofile.writeline("</data>\n");
            if (ofile.closeandrename() == false) return -1;
        }
    }
    if ((maxcount == 0) || (rpc % maxcount > 0)) {
        ofile.writeline("</data>\n");
        if (ofile.closeandrename() == false) return -1;
    }
    return nfiles;
}

/* in memory: written through cofile into a scratch directory and read back */
int64_t ref_format_xml(const char *const *fieldname, const int *fieldlen, int nfields,
                       const uint32_t *const *off, const uint8_t *const *bytes, int64_t rows,
                       int64_t maxcount, uint8_t *out, uint64_t *file_off, int64_t filecap,
                       int64_t *nfiles) {
    const char *base = getenv("TMPDIR");
    std::string dir = std::string(base && *base ? base : "/tmp") + "/mdc_fmt_XXXXXX";
    if (!mkdtemp(&dir[0])) return -1;
    int64_t nf = ref_format_xml_files(fieldname, fieldlen, nfields, off, bytes, rows, maxcount,
                                      dir.c_str(), "b", "t", "e", 1);
    uint64_t pos = 0;
    for (int64_t k = 0; k < (nf < 0 ? 0 : nf); ++k) {
        std::string name = dir + "/b_t_e_" + std::to_string(k + 1) + ".xml";
        if (file_off && k < filecap) file_off[k] = pos;
        FILE *f = fopen(name.c_str(), "rb");
        if (f) {
            char buf[65536];
            size_t got;
            while ((got = fread(buf, 1, sizeof buf, f)) > 0) {
                if (out) memcpy(out + pos, buf, got);
                pos += got;
            }
            fclose(f);
        }
        unlink(name.c_str());
    }
    /* a failed run leaves its .tmp behind, as the reference does */
    unlink((dir + "/b_t_e_1.xml.tmp").c_str());
    rmdir(dir.c_str());
    if (nf < 0) return -1;
    if (file_off && nf < filecap) file_off[nf] = pos;
    if (nfiles) *nfiles = nf;
    return (int64_t)pos;
}

/* N4.  The reference's timetostr as it is; a format it does not know leaves the string
 * untouched, which a marker in the string shows. */
int ref_timetostr(int64_t t, const char *fmt, char *out, size_t cap) {
    std::string s = "\x01";
    timetostr((time_t)t, s, fmt);
    if (s == "\x01") return -1;
    if (cap == 0) return -1;
    size_t n = s.size() < cap - 1 ? s.size() : cap - 1;
    memcpy(out, s.data(), n);
    out[n] = 0;
    return (int)s.size();
}

/* N2.  matchstr is the reference's own.  loadxmltotable, crtsql and preparesql live in
 * tools/xmltodb.cpp (needs OCI): their loops are restated here around the reference's own
 * cifile::readline, getxmlbuffer (char* and int forms), sformat and deleterchr. */
int ref_matchstr(const char *str, const char *rules) { return matchstr(str, rules) ? 1 : 0; }

int64_t ref_load_xmltotable(const uint8_t *data, size_t n, orc_xmltotable *out, int64_t cap) {
    std::string path = write_tmp(data, n);
    cifile ifile;
    if (path.empty() || ifile.open(path) == false) return -1;
    std::string buffer;
    int64_t r = 0;
    /* the char* getxmlbuffer copies a value of any length: room for the whole record, then
     * the cut to the array that the checker interface defines */
    std::vector<char> big;
    while (ifile.readline(buffer, "<endl/>")) {
        if (r < cap) {
            orc_xmltotable *e = &out[r];
            memset(e, 0, sizeof *e);
            big.assign(buffer.size() + 2, 0);
            getxmlbuffer(buffer, "filename", big.data());
            snprintf(e->filename, sizeof e->filename, "%s", big.data());
            big.assign(buffer.size() + 2, 0);
            getxmlbuffer(buffer, "tname", big.data());
            snprintf(e->tname, sizeof e->tname, "%s", big.data());
            getxmlbuffer(buffer, "uptbz", e->uptbz);
            big.assign(buffer.size() + 2, 0);
            getxmlbuffer(buffer, "execsql", big.data());
            snprintf(e->execsql, sizeof e->execsql, "%s", big.data());
        }
        ++r;
    }
    ifile.close();
    unlink(path.c_str());
    return r;
}

int ref_crtsql(const orc_xmltotable *e, const orc_col *cols, int ncols, char *ins, size_t inscap, char *upd,
               size_t updcap) {
    orc_xmltotable st = *e; /* tname + 2 below wants a non-const array, as the global is */
    std::vector<orc_col> vallcols(cols, cols + ncols), vpkcols;
    for (auto &c : vallcols)
        if (c.pkseq != 0) vpkcols.push_back(c);
    std::stable_sort(vpkcols.begin(), vpkcols.end(), [](const orc_col &a, const orc_col &b) { return a.pkseq < b.pkseq; });

    std::string cols_, binds;
    int colseq = 1;
    for (auto &c : vallcols) {
        if (strcmp(c.colname, "upttime") == 0) continue;
        cols_ += sformat("%s,", c.colname);
        if (strcmp(c.colname, "keyid") == 0)
            binds += sformat("SEQ_%s.nextval,", st.tname + 2);
        else if (strcmp(c.datatype, "date") == 0)
            binds += sformat("to_date(:%d,'yyyymmddhh24miss'),", colseq++);
        else
            binds += sformat(":%d,", colseq++);
    }
    deleterchr(cols_, ',');
    deleterchr(binds, ',');
    std::string insertsql = sformat("insert into %s(%s) values(%s)", st.tname, cols_.c_str(), binds.c_str());
    if (ins && inscap) snprintf(ins, inscap, "%s", insertsql.c_str());
    if (st.uptbz != 1) return 0;

    std::string strset = " set ", strwhere = " where 1=1";
    colseq = 1;
    for (auto &c : vallcols) {
        if (c.pkseq != 0) continue;
        if (strcmp(c.colname, "keyid") == 0) continue;
        if (strcmp(c.colname, "upttime") == 0)
            strset += sformat("upttime=sysdate,");
        else if (strcmp(c.datatype, "date") == 0)
            strset += sformat("%s=to_date(:%d,'yyyymmddhh24miss),", c.colname, colseq++);
        else
            strset += sformat("%s=:%d,", c.colname, colseq++);
    }
    deleterchr(strset, ',');
    for (auto &c : vpkcols) {
        if (strcmp(c.datatype, "date") == 0)
            strwhere += sformat(" and %s=to_date(:%d,'yyyymmddhh24miss')", c.colname, colseq++);
        else
            strwhere += sformat(" and %s=:%d", c.colname, colseq++);
    }
    std::string updatesql = sformat("update %s%s%s", st.tname, strset.c_str(), strwhere.c_str());
    if (upd && updcap) snprintf(upd, updcap, "%s", updatesql.c_str());
    return 1;
}

int ref_bind_plan(const orc_xmltotable *e, const orc_col *cols, int ncols, int *ins, int *upd, int *nupd) {
    int colseq = 1, u = 0;
    for (int i = 0; i < ncols; ++i) {
        if (strcmp(cols[i].colname, "upttime") == 0) continue;
        if (strcmp(cols[i].colname, "keyid") == 0) continue;
        ins[colseq++ - 1] = i;
    }
    if (e->uptbz == 1) {
        for (int i = 0; i < ncols; ++i) {
            if (cols[i].pkseq != 0) continue;
            if (strcmp(cols[i].colname, "keyid") == 0) continue;
            if (strcmp(cols[i].colname, "upttime") == 0) continue;
            upd[u++] = i;
        }
        for (int i = 0; i < ncols; ++i) {
            if (cols[i].pkseq == 0) continue;
            upd[u++] = i;
        }
    }
    if (nupd) *nupd = u;
    return colseq - 1;
}

void ref_batch_free(orc_batch *b) { builder_free(b); }

const char *ref_kind(void) { return "reference"; }

} /* extern "C" */
