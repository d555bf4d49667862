/*
 * mdc_public_compat.h -- the declarations of the reference's public/_public.h that the ingest
 * loop of tools/xmltodb.cpp uses (:256-268 and splitbuffer :492-522), and the delimited twin
 * ccmdstr, under their own names and argument lists in namespace idc, over libmdc_parse.so.
 * A translation unit that makes those calls compiles unchanged against this header.
 *
 *   reference declaration (public/_public.h)                          here
 *   cifile::open / readline(buf,endbz) / read / close...   :401-430   the file is read whole on open();
 *                                                                    the first readline() has the GPU
 *                                                                    split it into records in one go
 *                                                                    (mdc_parse_device: every record
 *                                                                    terminator of the file at once),
 *                                                                    readline() then hands them out
 *   getxmlbuffer(xmlbuffer,fieldname,value[,ilen]) x 9     :140-148   host code, same results
 *   picknumber x 3                                         :65-67    host code, same results
 *   ccmdstr (splittocmd, size, cmdcount, [], 9 getvalue)   :81-122   host code, same results
 *   deletelchr/deleterchr/deletelrchr, toupper/tolower     :16-44    host code, same results
 *
 * The per-record functions take one record at a time, as their signatures demand, and are plain
 * host code here (written from the behaviour of public/_public.cpp:11-123, :167-211, :265-625,
 * quirks included: see each function).  The bulk forms, in which the GPU also extracts, filters
 * and converts every field of every record, are cifile::splitall() below and the classes of
 * mdc_xmltodb.hpp.
 *
 * Conventions are the reference's: bool results, outputs untouched on false, nothing throws.
 * One difference in failure handling: there is no CPU path for the record split, so open() fails
 * (as for a file that cannot be opened: _xmltodb() then returns 5 and the file goes to the error
 * directory, tools/xmltodb.cpp:256-261, :187-204) when no usable GPU is present, and readline()
 * returns false with errcode() set when endbz holds more than 255 bytes, or holds a newline and two of
 * its occurrences in the file overlap, which the GPU record splitter does not take.  Header only; link
 * with -lmdc_parse.
 */
#ifndef MDC_PUBLIC_COMPAT_H
#define MDC_PUBLIC_COMPAT_H

#include <cstdio>
#include <cstring>
#include <fstream>
#include <ios>
#include <ostream>
#include <string>
#include <vector>

#include "mdc_parse.h"

namespace idc {

/* ---- trimming and case (public/_public.cpp:11-123) -------------------------------------- */

/* C strings: every leading / trailing cc goes. */
inline char *deletelchr(char *str, const int cc = ' ') {
    if (!str) return nullptr;
    size_t lead = 0;
    while (str[lead] == cc && str[lead] != 0) ++lead;  /* (the reference would run past a NUL equal to cc == 0 as well) */
    if (lead) std::memmove(str, str + lead, std::strlen(str + lead) + 1);
    return str;
}
/* std::string: the same.  find_first_not_of gives npos for a string of cc only, and replacing
 * [0, npos) empties it. */
inline std::string &deletelchr(std::string &str, const int cc = ' ') {
    const std::string::size_type keep = str.find_first_not_of((char)cc);
    if (keep != 0) str.erase(0, keep);
    return str;
}
inline char *deleterchr(char *str, const int cc = ' ') {
    if (!str) return nullptr;
    size_t end = std::strlen(str);
    while (end > 0 && str[end - 1] == cc) --end;
    str[end] = 0;
    return str;
}
/* std::string, with the reference's quirk (public/_public.cpp:52-59): the cut is skipped when
 * the last character that is not cc stands at index 0, so "a," stays "a," and " b " trimmed on
 * both sides becomes "b ".  A string of cc only becomes empty (npos + 1 wraps to 0). */
inline std::string &deleterchr(std::string &str, const int cc = ' ') {
    const std::string::size_type last = str.find_last_not_of((char)cc);
    if (last != 0) str.erase(last + 1);
    return str;
}
inline char *deletelrchr(char *str, const int cc = ' ') {
    deletelchr(str, cc);
    return deleterchr(str, cc);
}
inline std::string &deletelrchr(std::string &str, const int cc = ' ') {
    deletelchr(str, cc);
    return deleterchr(str, cc);
}

/* ASCII letters only (public/_public.cpp:77-123) */
inline char *toupper(char *str) {
    if (str)
        for (char *p = str; *p; ++p)
            if (*p >= 'a' && *p <= 'z') *p = (char)(*p - 32);
    return str;
}
inline std::string &toupper(std::string &str) {
    for (char &c : str)
        if (c >= 'a' && c <= 'z') c = (char)(c - 32);
    return str;
}
inline char *tolower(char *str) {
    if (str)
        for (char *p = str; *p; ++p)
            if (*p >= 'A' && *p <= 'Z') *p = (char)(*p + 32);
    return str;
}
inline std::string &tolower(std::string &str) {
    for (char &c : str)
        if (c >= 'A' && c <= 'Z') c = (char)(c + 32);
    return str;
}

/* ---- picknumber (public/_public.cpp:167-211) -------------------------------------------- */

/* Keeps the digits of src, with bsigned also '+' and '-', with bdot also '.'; everything else
 * goes.  src and dest may be the same string.  Digits are the ten ASCII digits (the reference
 * calls isdigit in the "C" locale; bytes above 0x7F are never digits here). */
inline std::string &picknumber(const std::string &src, std::string &dest, const bool bsigned = false, const bool bdot = false) {
    std::string kept;
    kept.reserve(src.size());
    for (const char c : src) {
        const bool digit = c >= '0' && c <= '9';
        const bool sign = bsigned && (c == '+' || c == '-');
        const bool dot = bdot && c == '.';
        if (digit || sign || dot) kept.push_back(c);
    }
    dest.swap(kept);
    return dest;
}
inline std::string picknumber(const std::string &src, const bool bsigned = false, const bool bdot = false) {
    std::string out;
    picknumber(src, out, bsigned, bdot);
    return out;
}
/* dest must have room for the result and a NUL */
inline char *picknumber(const std::string &src, char *dest, const bool bsigned = false, const bool bdot = false) {
    if (!dest) return nullptr;
    const std::string out = picknumber(src, bsigned, bdot);
    std::memcpy(dest, out.data(), out.size());
    dest[out.size()] = 0;
    return dest;
}

namespace compat_detail {
/* copies text into a caller-sized C buffer the way the reference's char* overloads do
 * (public/_public.cpp:319-337, :483-504): with len > 0 the first len + 1 bytes are zeroed and at
 * most len bytes copied; with len == 0 everything is copied; always NUL-terminated */
inline void copy_clipped(const std::string &text, char *value, const int len) {
    if (len > 0) std::memset(value, 0, (size_t)len + 1);
    const size_t n = (len == 0 || text.size() <= (size_t)(unsigned int)len) ? text.size() : (size_t)len;
    std::memcpy(value, text.data(), n);
    value[n] = 0;
}
/* the typed conversions share one shape: filter, then the std::sto* call, whose exceptions
 * (no digits, out of range) mean "false, value untouched" */
template <class T, class F>
inline bool convert(const std::string &filtered, T &value, F sto) {
    try {
        value = (T)sto(filtered);
    } catch (const std::exception &) {
        return false;
    }
    return true;
}
inline bool to_int(const std::string &s, int &v) {  /* sign kept, dot dropped: "12.7" -> 127 */
    return convert(picknumber(s, true), v, [](const std::string &t) { return std::stoi(t); });
}
inline bool to_uint(const std::string &s, unsigned int &v) {  /* stoi's range: "4000000000" fails */
    return convert(picknumber(s), v, [](const std::string &t) { return std::stoi(t); });
}
inline bool to_long(const std::string &s, long &v) {
    return convert(picknumber(s, true), v, [](const std::string &t) { return std::stol(t); });
}
inline bool to_ulong(const std::string &s, unsigned long &v) {
    return convert(picknumber(s), v, [](const std::string &t) { return std::stoul(t); });
}
inline bool to_double(const std::string &s, double &v) {  /* "1e5" -> 15: the 'e' is filtered */
    return convert(picknumber(s, true, true), v, [](const std::string &t) { return std::stod(t); });
}
inline bool to_float(const std::string &s, float &v) {
    return convert(picknumber(s, true, true), v, [](const std::string &t) { return std::stof(t); });
}
inline bool to_bool(std::string s) { return toupper(s) == "TRUE"; }  /* the whole text, no trimming */
}  // namespace compat_detail

/* ---- getxmlbuffer (public/_public.cpp:461-625) ------------------------------------------ */

/* The text between the first "<fieldname>" and the first "</fieldname>" of xmlbuffer, both looked
 * for from the start of the buffer, independently.  false and value untouched when either is
 * missing.  When the close tag stands in front of the open tag the result is everything behind
 * the open tag, and ilen does not clip it; otherwise ilen > 0 clips to ilen bytes. */
inline bool getxmlbuffer(const std::string &xmlbuffer, const std::string &fieldname, std::string &value, const int ilen = 0) {
    const std::string open = "<" + fieldname + ">", close = "</" + fieldname + ">";
    const std::string::size_type at_open = xmlbuffer.find(open);
    if (at_open == std::string::npos) return false;
    const std::string::size_type at_close = xmlbuffer.find(close);
    if (at_close == std::string::npos) return false;
    const std::string::size_type from = at_open + open.size();
    int take = (int)(at_close - from);  /* negative: substr then takes the rest of the buffer */
    if (ilen > 0 && ilen < take) take = ilen;
    value = xmlbuffer.substr(from, (std::string::size_type)take);
    return true;
}
/* true whether or not the tag is there (an absent tag gives the empty string); value must have
 * room for len + 1 bytes, or for the whole text and a NUL when len == 0 */
inline bool getxmlbuffer(const std::string &xmlbuffer, const std::string &fieldname, char *value, const int ilen = 0) {
    if (!value) return false;
    std::string text;
    getxmlbuffer(xmlbuffer, fieldname, text);
    compat_detail::copy_clipped(text, value, ilen);
    return true;
}
inline bool getxmlbuffer(const std::string &xmlbuffer, const std::string &fieldname, bool &value) {
    std::string text;
    if (!getxmlbuffer(xmlbuffer, fieldname, text)) return false;
    value = compat_detail::to_bool(text);
    return true;
}
inline bool getxmlbuffer(const std::string &xmlbuffer, const std::string &fieldname, int &value) {
    std::string text;
    return getxmlbuffer(xmlbuffer, fieldname, text) && compat_detail::to_int(text, value);
}
inline bool getxmlbuffer(const std::string &xmlbuffer, const std::string &fieldname, unsigned int &value) {
    std::string text;
    return getxmlbuffer(xmlbuffer, fieldname, text) && compat_detail::to_uint(text, value);
}
inline bool getxmlbuffer(const std::string &xmlbuffer, const std::string &fieldname, long &value) {
    std::string text;
    return getxmlbuffer(xmlbuffer, fieldname, text) && compat_detail::to_long(text, value);
}
inline bool getxmlbuffer(const std::string &xmlbuffer, const std::string &fieldname, unsigned long &value) {
    std::string text;
    return getxmlbuffer(xmlbuffer, fieldname, text) && compat_detail::to_ulong(text, value);
}
inline bool getxmlbuffer(const std::string &xmlbuffer, const std::string &fieldname, double &value) {
    std::string text;
    return getxmlbuffer(xmlbuffer, fieldname, text) && compat_detail::to_double(text, value);
}
inline bool getxmlbuffer(const std::string &xmlbuffer, const std::string &fieldname, float &value) {
    std::string text;
    return getxmlbuffer(xmlbuffer, fieldname, text) && compat_detail::to_float(text, value);
}

/* ---- ccmdstr (public/_public.cpp:265-459) ----------------------------------------------- */

class ccmdstr {
   private:
    std::vector<std::string> m_cmdstr;
    ccmdstr(const ccmdstr &) = delete;
    ccmdstr &operator=(const ccmdstr &) = delete;

   public:
    ccmdstr() {}
    ccmdstr(const std::string &buffer, const std::string &sepstr, const bool bdelspace = false) { splittocmd(buffer, sepstr, bdelspace); }
    ~ccmdstr() {}

    const std::string &operator[](int ii) const { return m_cmdstr[ii]; }

    /* Left to right, separators do not overlap ("x~!~y~!~!~z" by "~!~": x, y, !~z); n separators
     * give n + 1 fields, an empty buffer one empty field.  bdelspace trims blanks on both sides
     * with deletelrchr, quirk included.  An empty sepstr matches at the position searched from
     * without advancing: the reference never returns from that, neither is it supported here
     * (the buffer comes back as one field). */
    void splittocmd(const std::string &buffer, const std::string &sepstr, const bool bdelspace = false) {
        m_cmdstr.clear();
        std::string::size_type from = 0;
        if (!sepstr.empty())
            for (std::string::size_type hit; (hit = buffer.find(sepstr, from)) != std::string::npos; from = hit + sepstr.size())
                add(buffer.substr(from, hit - from), bdelspace);
        add(buffer.substr(from), bdelspace);
    }

    int size() const { return (int)m_cmdstr.size(); }
    int cmdcount() const { return (int)m_cmdstr.size(); }

    /* false when the line has no field ii (a negative ii included); ilen > 0 clips */
    bool getvalue(const int ii, std::string &value, const int ilen = 0) const {
        if (!has(ii)) return false;
        const std::string &f = m_cmdstr[(size_t)ii];
        value = (ilen > 0 && (size_t)ilen < f.size()) ? f.substr(0, (size_t)ilen) : f;
        return true;
    }
    bool getvalue(const int ii, char *value, const int ilen = 0) const {
        if (!has(ii) || !value) return false;
        compat_detail::copy_clipped(m_cmdstr[(size_t)ii], value, ilen);
        return true;
    }
    bool getvalue(const int ii, int &value) const { return has(ii) && compat_detail::to_int(m_cmdstr[(size_t)ii], value); }
    bool getvalue(const int ii, unsigned int &value) const { return has(ii) && compat_detail::to_uint(m_cmdstr[(size_t)ii], value); }
    bool getvalue(const int ii, long &value) const { return has(ii) && compat_detail::to_long(m_cmdstr[(size_t)ii], value); }
    bool getvalue(const int ii, unsigned long &value) const { return has(ii) && compat_detail::to_ulong(m_cmdstr[(size_t)ii], value); }
    bool getvalue(const int ii, double &value) const { return has(ii) && compat_detail::to_double(m_cmdstr[(size_t)ii], value); }
    bool getvalue(const int ii, float &value) const { return has(ii) && compat_detail::to_float(m_cmdstr[(size_t)ii], value); }
    bool getvalue(const int ii, bool &value) const {
        if (!has(ii)) return false;
        value = compat_detail::to_bool(m_cmdstr[(size_t)ii]);
        return true;
    }

   private:
    bool has(const int ii) const { return ii >= 0 && (size_t)ii < m_cmdstr.size(); }
    void add(std::string field, const bool bdelspace) {
        if (bdelspace) deletelrchr(field);
        m_cmdstr.push_back(std::move(field));
    }
};

inline std::ostream &operator<<(std::ostream &out, const ccmdstr &cc) {
    for (int ii = 0; ii < cc.size(); ++ii) out << "[" << ii << "]=" << cc[ii] << std::endl;
    return out;
}

/* ---- cifile (public/_public.cpp:828-892) ------------------------------------------------ */

class cifile {
   public:
    cifile() {}
    cifile(const cifile &) = delete;
    cifile &operator=(const cifile &) = delete;
    ~cifile() { close(); }

    bool isopen() const { return m_open; }

    /* Reads the whole file into pinned host memory, from where it goes to the GPU in one copy (the mode
     * is accepted for the signature's sake; bytes are taken as they are, which is what the reference's
     * text mode does on Linux).  false when the file cannot be opened or read, or when there is no
     * usable GPU to split it with (errcode() tells which). */
    bool open(const std::string &filename, const std::ios::openmode mode = std::ios::in) {
        (void)mode;
        close();
        m_filename = filename;
        std::FILE *f = std::fopen(filename.c_str(), "rb");
        if (!f) return fail(MDC_FILE_OPEN_FAILED, "cannot open " + filename);
        const int ndev = mdc_device_count();
        if (ndev < 1) {
            std::fclose(f);
            return fail(MDC_ERR_NO_DEVICE, ndev < 0 ? std::string("no usable GPU: ") + mdc_last_error() : std::string("no usable GPU"));
        }
        /* (a file that grows while it is read: the buffer grows with it) */
        size_t cap = 0;
        for (;;) {
            if (m_size == cap) {
                void *q = nullptr;
                size_t qcap = 0;
                std::fseek(f, 0, SEEK_END);
                const long end = std::ftell(f);
                std::fseek(f, (long)m_size, SEEK_SET);
                const size_t want = (end > 0 && (size_t)end > m_size ? (size_t)end : m_size) + 4096;
                if (mdc_host_buffer_get(want, &q, &qcap) != MDC_OK) {
                    std::fclose(f);
                    return fail(MDC_ERR_NOMEM, std::string("no pinned memory for ") + filename);
                }
                if (m_size) std::memcpy(q, m_buf, m_size);
                if (m_buf) mdc_host_buffer_put(m_buf, m_cap);
                m_buf = (char *)q;
                m_cap = cap = qcap;
            }
            const size_t got = std::fread(m_buf + m_size, 1, cap - m_size, f);
            m_size += got;
            if (got == 0) break;
        }
        const bool bad = std::ferror(f) != 0;
        std::fclose(f);
        if (bad) return fail(MDC_ERR_IO, "cannot read " + filename);
        m_open = true;
        m_pos = 0;
        m_rc = MDC_OK;
        m_err.clear();
        return true;
    }

    /* The next record: the bytes from the current position up to and including endbz, where endbz
     * stands right in front of a '\n' (that '\n' is consumed, not returned; earlier '\n' inside the
     * record are kept, '\r' is an ordinary byte); with endbz == "" the next line.  false at the end
     * of the file: what follows the last terminator is never a record, a last line without '\n'
     * included, and buf then holds those bytes up to the last '\n', as the reference leaves them. */
    bool readline(std::string &buf, const std::string &endbz = "") {
        buf.clear();
        if (!m_open) return false;
        if (!m_split || endbz != m_endbz) {
            if (!split(endbz)) return false;
        }
        const size_t el = m_endbz.size();
        /* first terminator at or behind the position whose record is long enough to end with endbz
         * (it always is, unless read() left the position inside a mark) */
        size_t lo = 0, hi = (size_t)m_nrec;
        while (lo < hi) {
            const size_t mid = (lo + hi) / 2;
            if (m_recoff[mid + 1] <= m_pos) lo = mid + 1; else hi = mid;
        }
        while (lo < (size_t)m_nrec && m_recoff[lo + 1] - 1 - m_pos < el) ++lo;
        if (lo >= (size_t)m_nrec) {
            /* no further record: the reference has by now appended every remaining complete line */
            size_t nl = m_size;
            while (nl > m_pos && m_buf[nl - 1] != '\n') --nl;  /* one behind the last '\n' at or behind the position */
            if (nl > m_pos) buf.assign(m_buf + m_pos, nl - m_pos);
            m_pos = m_size;
            m_eof = true;
            return false;
        }
        const size_t end = (size_t)m_recoff[lo + 1];  /* one behind the terminating '\n' */
        buf.assign(m_buf + m_pos, end - 1 - m_pos);
        m_pos = end;
        return true;
    }

    /* bytes from the current position, as many as there are (at most bufsize); their count */
    int read(void *buf, const int bufsize) {
        if (!m_open || bufsize <= 0 || m_eof) return 0;
        const size_t left = m_size - m_pos, n = left < (size_t)bufsize ? left : (size_t)bufsize;
        std::memcpy(buf, m_buf + m_pos, n);
        m_pos += n;
        if (n < (size_t)bufsize) m_eof = true;
        return (int)n;
    }

    bool closeandremove() {
        if (!m_open) return false;
        close();
        return std::remove(m_filename.c_str()) == 0;
    }

    void close() {
        if (m_batch) mdc_batch_free(m_batch);
        m_batch = nullptr;
        if (m_parser) mdc_parser_destroy(m_parser);
        m_parser = nullptr;
        m_split = false;
        m_open = false;
        m_eof = false;
        m_nrec = 0;
        m_pos = 0;
        m_recoff = nullptr;
        if (m_buf) mdc_host_buffer_put(m_buf, m_cap);
        m_buf = nullptr;
        m_size = m_cap = 0;
    }

    /* ---- beyond the reference's interface ------------------------------------------------- */
    /* why the last open() or readline() failed: an MDC_* code and the library's text */
    int errcode() const { return m_rc; }
    const std::string &errmsg() const { return m_err; }
    /* records the file holds for the end mark of the last readline (totalcount after the loop of
     * tools/xmltodb.cpp:263-304), or -1 before the first readline */
    int64_t records() const { return m_split ? m_nrec : -1; }
    /* The bulk form of the loop body: every field of every record of the file on the GPU, i.e.
     * what splitbuffer (tools/xmltodb.cpp:492-522) leaves in vcolvalue for each record in turn.
     * The batch belongs to the caller (mdc_batch_free); cell (row, col) through mdc_batch_cell. */
    mdc_batch *splitall(const mdc_col *cols, int ncols, const std::string &endbz = "<endl/>", const int *typed = nullptr) {
        if (!m_open) return nullptr;
        mdc_opts o;
        std::memset(&o, 0, sizeof o);
        o.endbz = endbz.c_str();
        o.carry = 1;
        o.typed = typed;
        mdc_batch *b = nullptr;
        const int rc = mdc_parse_xml_buffer(m_buf, m_size, cols, ncols, &o, &b);
        if (rc != MDC_OK) {
            fail(rc, mdc_last_error());
            return nullptr;
        }
        return b;
    }

   private:
    bool fail(int rc, const std::string &msg) {
        m_rc = rc;
        m_err = msg;
        return false;
    }
    /* the record terminators of the whole file for this end mark, in one parse on the GPU */
    bool split(const std::string &endbz) {
        if (m_batch) mdc_batch_free(m_batch);
        m_batch = nullptr;
        if (m_parser) mdc_parser_destroy(m_parser);
        m_parser = nullptr;
        m_split = false;
        if (endbz.size() > 255 || endbz.find('\0') != std::string::npos)
            return fail(MDC_ERR_ARG, "readline: an end mark with a NUL or more than 255 bytes is not supported");
        /* only the record offsets are wanted: the record split alone (the one column a schema must have
         * gets no cells) */
        mdc_col col;
        std::memset(&col, 0, sizeof col);
        std::strcpy(col.colname, "\x01");
        std::strcpy(col.datatype, "char");
        const int untyped = MDC_T_NONE;
        mdc_opts o;
        std::memset(&o, 0, sizeof o);
        o.endbz = endbz.c_str();
        o.typed = &untyped;
        o.flags = MDC_F_RECORDS_ONLY;
        int rc = mdc_parser_create_xml(&col, 1, &o, &m_parser);
        if (rc == MDC_OK) rc = mdc_parse_host(m_parser, m_buf, m_size, &m_batch);
        if (rc == MDC_OK) rc = mdc_batch_host_recoff(m_batch, &m_recoff);
        if (rc != MDC_OK) return fail(rc, mdc_last_error());
        m_nrec = mdc_batch_rows(m_batch);
        m_endbz = endbz;
        m_split = true;
        return true;
    }

    std::string m_filename, m_endbz, m_err;
    char *m_buf = nullptr;  /* the file's bytes, in pinned host memory of the library's pool */
    size_t m_size = 0, m_cap = 0;
    bool m_open = false, m_split = false, m_eof = false;
    size_t m_pos = 0;
    mdc_parser *m_parser = nullptr;
    mdc_batch *m_batch = nullptr;
    const uint64_t *m_recoff = nullptr;  /* m_nrec + 1 entries: record r is [recoff[r], recoff[r+1] - 1) */
    int64_t m_nrec = 0;
    int m_rc = MDC_OK;
};

}  // namespace idc
#endif
