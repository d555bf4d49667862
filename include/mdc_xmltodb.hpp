/*
 * mdc_xmltodb.hpp -- C++ host layer over the C ABI of mdc_parse.h, shaped like the calls of
 * the reference's ingest loop so that tools/xmltodb.cpp:263-304 can switch to it line by line.
 * Header only; link with -lmdc_parse.
 *
 *   reference (per record)                                   here (per file, then per row)
 *   cifile::open(name)           public/_public.cpp:828      cxmlfile::open(name)      parses the file on the GPU
 *   cifile::readline(buf,endbz)  public/_public.cpp:866      cxmlfile::readline(buf)   record text, same bytes
 *   splitbuffer(buf)             tools/xmltodb.cpp:492       cxmlfile::splitbuffer(v)  cells of the current record
 *   getxmlbuffer(xml,name,v)     public/_public.cpp:461-625  cxmlfile::getxmlbuffer(name,v) for the current record
 *   ccmdstr::splittocmd+getvalue public/_public.cpp:274-446  ccmdfile::getvalue(ii,v)  for the current line
 *   loadxmltotable/findxmltotable tools/xmltodb.cpp:102,311  cxmltotable::loadxmltotable / findxmltotable
 *   crtsql / preparesql          tools/xmltodb.cpp:325,421   cxmltotable::crtsql / bindplan
 *
 * Return conventions are the reference's: bool, true = found/converted, outputs untouched on
 * false, nothing throws.  errmsg() holds the library's text after a failed open().
 */
#ifndef MDC_XMLTODB_HPP
#define MDC_XMLTODB_HPP

#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "mdc_parse.h"
#include "mdc_route.h"

namespace mdc {

/* One column of the target table: ctcols::st_col (tools/_tools.h:21-27) plus the typed
 * conversion wanted for it (MDC_T_*; MDC_T_DEFAULT: number -> double, date -> long). */
struct st_col : mdc_col {
    int typed = MDC_T_DEFAULT;
    st_col() { std::memset(static_cast<mdc_col *>(this), 0, sizeof(mdc_col)); }
    st_col(const std::string &name, const std::string &type, int len, int typed_ = MDC_T_DEFAULT, int pk = 0) : st_col() {
        std::snprintf(colname, sizeof(colname), "%s", name.c_str());
        std::snprintf(datatype, sizeof(datatype), "%s", type.c_str());
        collen = len;
        pkseq = pk;
        typed = typed_;
    }
};

namespace detail {
/* state shared by the two file classes: the input bytes (kept for readline), the batch, the
 * cursor */
class bulkfile {
   public:
    bulkfile() = default;
    bulkfile(const bulkfile &) = delete;
    bulkfile &operator=(const bulkfile &) = delete;
    ~bulkfile() { close(); }

    bool isopen() const { return m_batch != nullptr; }
    /* number of records in the file: totalcount of tools/xmltodb.cpp:266 after the loop */
    int64_t totalcount() const { return m_rows; }
    /* index of the current record (-1 before the first readline) */
    int64_t row() const { return m_row; }
    const std::string &errmsg() const { return m_err; }
    int errcode() const { return m_rc; }
    mdc_batch *batch() const { return m_batch; }

    /* Next record; buf receives what cifile::readline(buf,endbz) returns for it (the text
     * without the final '\n').  false at the end of the file. */
    bool readline(std::string &buf) {
        if (!next()) return false;
        buf.assign(m_data.data() + m_recoff[m_row], (size_t)(m_recoff[m_row + 1] - m_recoff[m_row] - 1));
        return true;
    }
    /* the same without copying the text */
    bool next() {
        if (!m_batch || m_row + 1 >= m_rows) return false;
        ++m_row;
        return true;
    }
    void rewind() { m_row = -1; }

    /* cifile::close / closeandremove (public/_public.cpp:843-864) */
    void close() {
        if (m_batch) mdc_batch_free(m_batch);
        m_batch = nullptr;
        m_rows = 0;
        m_row = -1;
        m_data.clear();
        m_data.shrink_to_fit();
    }
    bool closeandremove() {
        if (!m_batch) return false;
        close();
        return std::remove(m_filename.c_str()) == 0;
    }

   protected:
    bool load(const std::string &filename) {
        close();
        m_filename = filename;
        std::ifstream fin(filename, std::ios::in | std::ios::binary);
        if (!fin.is_open()) return fail(MDC_FILE_OPEN_FAILED, "cannot open " + filename);
        fin.seekg(0, std::ios::end);
        const std::streamoff n = fin.tellg();
        fin.seekg(0, std::ios::beg);
        m_data.resize((size_t)(n > 0 ? n : 0));
        if (n > 0 && !fin.read(&m_data[0], n)) return fail(MDC_ERR_IO, "cannot read " + filename);
        return true;
    }
    bool adopt(int rc, mdc_batch *b) {
        if (rc != MDC_OK) return fail(rc, mdc_last_error());
        m_batch = b;
        m_rows = mdc_batch_rows(b);
        m_row = -1;
        rc = mdc_batch_host_recoff(b, &m_recoff);
        if (rc != MDC_OK) {
            close();
            return fail(rc, mdc_last_error());
        }
        m_rc = MDC_OK;
        m_err.clear();
        return true;
    }
    bool fail(int rc, const std::string &msg) {
        m_rc = rc;
        m_err = msg;
        return false;
    }
    bool cellat(int col, const char **p, uint32_t *len) const {
        return m_batch && m_row >= 0 && mdc_batch_cell(m_batch, m_row, col, p, len) == MDC_OK;
    }
    bool bit(const uint32_t *mask, int words, int col) const {
        return (mask[(size_t)m_row * words + (col >> 5)] >> (col & 31)) & 1u;
    }

    std::string m_filename, m_data, m_err;
    mdc_batch *m_batch = nullptr;
    const uint64_t *m_recoff = nullptr;
    int64_t m_rows = 0, m_row = -1;
    int m_rc = MDC_OK;
};
}  // namespace detail

/* A file of tagged records ("<f>v</f>...<endl/>\n"), parsed in one go. */
class cxmlfile : public detail::bulkfile {
   public:
    /* carry = true: cells are what splitbuffer leaves in vcolvalue (a missing tag keeps the
     * previous column's text); carry = false: every column is an independent getxmlbuffer. */
    explicit cxmlfile(const std::vector<st_col> &cols, bool carry = true, int device = 0)
        : m_cols(cols), m_carry(carry), m_device(device) {}
    ~cxmlfile() {
        close();
        if (m_parser) mdc_parser_destroy(m_parser);
    }

    /* cifile::open + the whole read loop: false when the file cannot be opened (errcode() ==
     * MDC_FILE_OPEN_FAILED, the 5 of tools/xmltodb.cpp:260) or cannot be parsed. */
    bool open(const std::string &filename, const std::string &endbz = "<endl/>") {
        if (!load(filename)) return false;
        if (!ensure_parser(endbz)) return false;
        mdc_batch *b = nullptr;
        const int rc = mdc_parse_host(m_parser, m_data.data(), m_data.size(), &b);
        if (!adopt(rc, b)) return false;
        m_words = mdc_batch_words_per_row(b);
        m_val.assign(m_cols.size(), nullptr);
        return mdc_batch_host_masks(b, &m_valid, &m_found) == MDC_OK || fail(MDC_ERR_INTERNAL, mdc_last_error());
    }

    int colcount() const { return (int)m_cols.size(); }
    int colindex(const std::string &fieldname) const {
        for (size_t i = 0; i < m_cols.size(); ++i)
            if (fieldname == m_cols[i].colname) return (int)i;
        return -1;
    }

    /* tools/xmltodb.cpp:492-522 for the current record: vcolvalue[i] = the text bound to
     * column i.  The vector is resized to the column count. */
    void splitbuffer(std::vector<std::string> &vcolvalue) const {
        vcolvalue.resize(m_cols.size());
        for (size_t i = 0; i < m_cols.size(); ++i) {
            const char *p = nullptr;
            uint32_t len = 0;
            if (cellat((int)i, &p, &len))
                vcolvalue[i].assign(p, len);
            else
                vcolvalue[i].clear();
        }
    }
    /* cell of the current record without a copy */
    bool cell(int col, const char **p, uint32_t *len) const { return cellat(col, p, len); }

    /* getxmlbuffer(xmlbuffer, fieldname, value, ilen) for the current record
     * (public/_public.cpp:461-481); ilen is the column's collen.  false and value untouched
     * when the record lacks <fieldname> or </fieldname>.  Meant for carry = false files. */
    bool getxmlbuffer(const std::string &fieldname, std::string &value) const {
        const int c = colindex(fieldname);
        const char *p;
        uint32_t len;
        if (c < 0 || m_row < 0 || !bit(m_found, m_words, c) || !cellat(c, &p, &len)) return false;
        value.assign(p, len);
        return true;
    }
    /* typed forms (public/_public.cpp:506-625): the column must have been declared with the
     * matching MDC_T_* code; false when the tag is missing or the conversion fails. */
    bool getxmlbuffer(const std::string &fieldname, int &value) const {
        uint64_t v;
        if (!typedvalue(fieldname, MDC_T_I32, v)) return false;
        value = (int)(int64_t)v;
        return true;
    }
    bool getxmlbuffer(const std::string &fieldname, unsigned int &value) const {
        uint64_t v;
        if (!typedvalue(fieldname, MDC_T_U32, v)) return false;
        value = (unsigned int)v;
        return true;
    }
    bool getxmlbuffer(const std::string &fieldname, long &value) const {
        uint64_t v;
        if (!typedvalue(fieldname, MDC_T_I64, v)) return false;
        value = (long)v;
        return true;
    }
    bool getxmlbuffer(const std::string &fieldname, unsigned long &value) const {
        uint64_t v;
        if (!typedvalue(fieldname, MDC_T_U64, v)) return false;
        value = (unsigned long)v;
        return true;
    }
    bool getxmlbuffer(const std::string &fieldname, double &value) const {
        uint64_t v;
        if (!typedvalue(fieldname, MDC_T_F64, v)) return false;
        std::memcpy(&value, &v, sizeof(double));
        return true;
    }
    bool getxmlbuffer(const std::string &fieldname, float &value) const {
        uint64_t v;
        if (!typedvalue(fieldname, MDC_T_F32, v)) return false;
        const uint32_t w = (uint32_t)v;
        std::memcpy(&value, &w, sizeof(float));
        return true;
    }
    bool getxmlbuffer(const std::string &fieldname, bool &value) const {
        uint64_t v;
        if (!typedvalue(fieldname, MDC_T_BOOL, v)) return false;
        value = v != 0;
        return true;
    }

   private:
    bool ensure_parser(const std::string &endbz) {
        if (m_parser && endbz == m_endbz) return true;
        if (m_parser) mdc_parser_destroy(m_parser);
        m_parser = nullptr;
        m_endbz = endbz;
        std::vector<mdc_col> c(m_cols.begin(), m_cols.end());
        std::vector<int> t;
        for (const st_col &x : m_cols) t.push_back(x.typed);
        mdc_opts o;
        std::memset(&o, 0, sizeof(o));
        o.device = m_device;
        o.endbz = m_endbz.c_str();
        o.carry = m_carry ? 1 : 0;
        o.typed = t.data();
        const int rc = mdc_parser_create_xml(c.data(), (int)c.size(), &o, &m_parser);
        return rc == MDC_OK || fail(rc, mdc_last_error());
    }
    bool typedvalue(const std::string &fieldname, int want, uint64_t &v) const {
        const int c = colindex(fieldname);
        if (c < 0 || m_row < 0 || m_cols[c].typed != want) return false;
        if (!bit(m_found, m_words, c) || !bit(m_valid, m_words, c)) return false;
        if (!m_val[c] && mdc_batch_host_values(m_batch, c, &m_val[c]) != MDC_OK) return false;
        v = m_val[c][m_row];
        return true;
    }

    std::vector<st_col> m_cols;
    bool m_carry;
    int m_device;
    std::string m_endbz;
    mdc_parser *m_parser = nullptr;
    const uint32_t *m_valid = nullptr, *m_found = nullptr;
    int m_words = 0;
    mutable std::vector<const uint64_t *> m_val;
};

/* A file of delimited lines, split as ccmdstr::splittocmd(line, sepstr, bdelspace) does
 * (public/_public.cpp:274-302). */
class ccmdfile : public detail::bulkfile {
   public:
    /* nfields: fields materialised per line (0 .. nfields-1); ilen[i] > 0 clips field i as
     * getvalue(i, value, ilen) does; typed[i] is the MDC_T_* wanted for field i. */
    ccmdfile(const std::string &sepstr, bool bdelspace, int nfields, const std::vector<int> &typed = {},
             const std::vector<int> &ilen = {}, int device = 0)
        : m_sep(sepstr), m_bdelspace(bdelspace), m_nfields(nfields), m_typed(typed), m_ilen(ilen), m_device(device) {
        m_typed.resize((size_t)nfields, MDC_T_NONE);
        m_ilen.resize((size_t)nfields, 0);
    }
    ~ccmdfile() {
        close();
        if (m_parser) mdc_parser_destroy(m_parser);
    }

    bool open(const std::string &filename) {
        if (!load(filename)) return false;
        if (!m_parser) {
            mdc_opts o;
            std::memset(&o, 0, sizeof(o));
            o.device = m_device;
            o.endbz = "";
            o.typed = m_typed.data();
            const int rc = mdc_parser_create_csv(m_sep.c_str(), m_bdelspace ? 1 : 0, m_nfields, m_ilen.data(), &o, &m_parser);
            if (rc != MDC_OK) return fail(rc, mdc_last_error());
        }
        mdc_batch *b = nullptr;
        const int rc = mdc_parse_host(m_parser, m_data.data(), m_data.size(), &b);
        if (!adopt(rc, b)) return false;
        m_words = mdc_batch_words_per_row(b);
        m_val.assign((size_t)m_nfields, nullptr);
        return mdc_batch_host_masks(b, &m_valid, &m_found) == MDC_OK || fail(MDC_ERR_INTERNAL, mdc_last_error());
    }

    /* ccmdstr::getvalue(ii, string&) (public/_public.cpp:304-317): false when the current
     * line has fewer than ii+1 fields. */
    bool getvalue(int ii, std::string &value) const {
        const char *p;
        uint32_t len;
        if (ii < 0 || ii >= m_nfields || m_row < 0 || !bit(m_found, m_words, ii) || !cellat(ii, &p, &len)) return false;
        value.assign(p, len);
        return true;
    }
    /* typed forms (public/_public.cpp:339-446) */
    bool getvalue(int ii, int &value) const {
        uint64_t v;
        if (!typedvalue(ii, MDC_T_I32, v)) return false;
        value = (int)(int64_t)v;
        return true;
    }
    bool getvalue(int ii, unsigned int &value) const {
        uint64_t v;
        if (!typedvalue(ii, MDC_T_U32, v)) return false;
        value = (unsigned int)v;
        return true;
    }
    bool getvalue(int ii, long &value) const {
        uint64_t v;
        if (!typedvalue(ii, MDC_T_I64, v)) return false;
        value = (long)v;
        return true;
    }
    bool getvalue(int ii, unsigned long &value) const {
        uint64_t v;
        if (!typedvalue(ii, MDC_T_U64, v)) return false;
        value = (unsigned long)v;
        return true;
    }
    bool getvalue(int ii, double &value) const {
        uint64_t v;
        if (!typedvalue(ii, MDC_T_F64, v)) return false;
        std::memcpy(&value, &v, sizeof(double));
        return true;
    }
    bool getvalue(int ii, float &value) const {
        uint64_t v;
        if (!typedvalue(ii, MDC_T_F32, v)) return false;
        const uint32_t w = (uint32_t)v;
        std::memcpy(&value, &w, sizeof(float));
        return true;
    }
    bool getvalue(int ii, bool &value) const {
        uint64_t v;
        if (!typedvalue(ii, MDC_T_BOOL, v)) return false;
        value = v != 0;
        return true;
    }

   private:
    bool typedvalue(int ii, int want, uint64_t &v) const {
        if (ii < 0 || ii >= m_nfields || m_row < 0 || m_typed[(size_t)ii] != want) return false;
        if (!bit(m_found, m_words, ii) || !bit(m_valid, m_words, ii)) return false;
        if (!m_val[(size_t)ii] && mdc_batch_host_values(m_batch, ii, &m_val[(size_t)ii]) != MDC_OK) return false;
        v = m_val[(size_t)ii][m_row];
        return true;
    }

    std::string m_sep;
    bool m_bdelspace;
    int m_nfields;
    std::vector<int> m_typed, m_ilen;
    int m_device;
    mdc_parser *m_parser = nullptr;
    const uint32_t *m_valid = nullptr, *m_found = nullptr;
    int m_words = 0;
    mutable std::vector<const uint64_t *> m_val;
};

/* The routing table of the ingest and the statements of the matched entry: the globals
 * vxmltotable, stxmltotable, insertsql and updatesql of tools/xmltodb.cpp:35-56 as members. */
class cxmltotable {
   public:
    cxmltotable() { std::memset(&stxmltotable, 0, sizeof(stxmltotable)); }
    cxmltotable(const cxmltotable &) = delete;
    cxmltotable &operator=(const cxmltotable &) = delete;
    ~cxmltotable() {
        if (m_route) mdc_route_free(m_route);
    }

    /* tools/xmltodb.cpp:102-129: false when the parameter file cannot be opened */
    bool loadxmltotable(const std::string &inifilename) {
        if (m_route) mdc_route_free(m_route);
        m_route = nullptr;
        return mdc_route_load(inifilename.c_str(), &m_route) == MDC_OK;
    }
    int size() const { return m_route ? mdc_route_count(m_route) : 0; }
    const mdc_xmltotable *operator[](int i) const { return m_route ? mdc_route_entry(m_route, i) : nullptr; }

    /* tools/xmltodb.cpp:311-323: the first entry whose rules match the file name is copied
     * to stxmltotable; false (return code 1 of _xmltodb) when none matches */
    bool findxmltotable(const std::string &xmlfile) {
        const int i = m_route ? mdc_route_find(m_route, xmlfile.c_str()) : -1;
        if (i < 0) return false;
        stxmltotable = *mdc_route_entry(m_route, i);
        return true;
    }

    /* tools/xmltodb.cpp:325-419 for stxmltotable and the given columns (tcols.m_vallcols);
     * updatesql keeps its previous text when stxmltotable.uptbz != 1, as in the reference */
    bool crtsql(const std::vector<st_col> &vallcols) {
        std::vector<mdc_col> c(vallcols.begin(), vallcols.end());
        size_t il = 0, ul = 0;
        mdc_crtsql(&stxmltotable, c.data(), (int)c.size(), nullptr, 0, &il, nullptr, 0, &ul);  // sizes first
        std::string ins(il + 1, '\0'), upd(ul + 1, '\0');
        if (mdc_crtsql(&stxmltotable, c.data(), (int)c.size(), &ins[0], ins.size(), &il, &upd[0], upd.size(), &ul) != MDC_OK)
            return false;
        insertsql.assign(ins.data(), il);
        if (stxmltotable.uptbz == 1) updatesql.assign(upd.data(), ul);
        return true;
    }
    /* tools/xmltodb.cpp:421-477: columns bound to :1, :2, ... of the two statements */
    bool bindplan(const std::vector<st_col> &vallcols, std::vector<int> &ins, std::vector<int> &upd) const {
        std::vector<mdc_col> c(vallcols.begin(), vallcols.end());
        ins.assign(c.size() + 1, 0);
        upd.assign(c.size() + 1, 0);
        int ni = 0, nu = 0;
        if (mdc_bind_plan(&stxmltotable, c.data(), (int)c.size(), ins.data(), &ni, upd.data(), &nu) != MDC_OK) return false;
        ins.resize((size_t)ni);
        upd.resize((size_t)nu);
        return true;
    }

    mdc_xmltotable stxmltotable;
    std::string insertsql, updatesql;

   private:
    mdc_route *m_route = nullptr;
};

}  // namespace mdc
#endif
