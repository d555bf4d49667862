// Host side around the parse: routing entries of the parameter file, file-name matching, SQL
// text and bind order.  Pure host code (no HIP call); declared in include/mdc_route.h.
//
// Reference behaviour followed (paths relative to the reference root):
//   loadxmltotable   tools/xmltodb.cpp:102-129
//   findxmltotable   tools/xmltodb.cpp:311-323
//   crtsql           tools/xmltodb.cpp:325-419
//   preparesql       tools/xmltodb.cpp:421-477 (bind order only)
//   matchstr         public/_public.cpp:213-263
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "mdc_route.h"

extern "C" int mdc_internal_fail(int code, const char *msg);

namespace {

using std::string;
const size_t NPOS = string::npos;

// ---- pieces of the record parser, for one small buffer on the host -------------------------

// Records of cifile::readline(buf,"<endl/>") (public/_public.cpp:866-892): a record ends at a
// '\n' that follows the mark; what follows the last such '\n' is dropped.
std::vector<string> records_of(const char *d, size_t n, const string &endbz) {
    std::vector<string> out;
    size_t start = 0;
    for (size_t i = 0; i < n; ++i) {
        if (d[i] != '\n') continue;
        const size_t len = i - start;  // the line(s) so far, without this '\n'
        if (len >= endbz.size() && std::memcmp(d + i - endbz.size(), endbz.data(), endbz.size()) == 0) {
            out.emplace_back(d + start, len);
            start = i + 1;
        }
    }
    return out;
}

// getxmlbuffer(xml,name,value) without a length limit (public/_public.cpp:461-481): both tags
// are searched from the start of the record, independently; a close tag in front of the open
// tag gives the rest of the record.
bool field_of(const string &xml, const string &name, string &value) {
    const string s = "<" + name + ">", e = "</" + name + ">";
    const size_t sp = xml.find(s);
    if (sp == NPOS) return false;
    const size_t ep = xml.find(e);
    if (ep == NPOS) return false;
    const int len = (int)ep - (int)sp - (int)s.size();
    value = len < 0 ? xml.substr(sp + s.size()) : xml.substr(sp + s.size(), (size_t)len);
    return true;
}

// getxmlbuffer(xml,name,int&) (public/_public.cpp:519-535): stoi of the digits and signs of
// the value; false when there is no number in front or it does not fit an int.
bool int_field_of(const string &xml, const string &name, int &value) {
    string v;
    if (!field_of(xml, name, v)) return false;
    string f;
    for (char c : v)
        if ((c >= '0' && c <= '9') || c == '+' || c == '-') f += c;
    size_t i = 0;
    bool neg = false;
    if (i < f.size() && (f[i] == '+' || f[i] == '-')) neg = f[i++] == '-';
    if (i >= f.size() || f[i] < '0' || f[i] > '9') return false;
    long long acc = 0;
    for (; i < f.size() && f[i] >= '0' && f[i] <= '9'; ++i) {
        acc = acc * 10 + (f[i] - '0');
        if (acc > 2147483648LL) return false;
    }
    if (neg) acc = -acc;
    if (acc > 2147483647LL) return false;
    value = (int)acc;
    return true;
}

// The char* getxmlbuffer (public/_public.cpp:483-504) copies the whole value and the caller
// reads it as a C string: cut at the first NUL; cut to the array as well (see the header).
bool put(char *dst, size_t cap, const string &v) {
    size_t n = v.find('\0');
    if (n == NPOS) n = v.size();
    const bool cut = n > cap - 1;
    if (cut) n = cap - 1;
    std::memcpy(dst, v.data(), n);
    dst[n] = 0;
    return cut;
}

// ---- matchstr -------------------------------------------------------------------------------

void upper(string &s) {
    for (char &c : s)
        if (c >= 'a' && c <= 'z') c = (char)(c - 32);
}

// ccmdstr::splittocmd(buffer,sep) without trimming (public/_public.cpp:274-302)
std::vector<string> split(const string &b, const string &sep) {
    std::vector<string> out;
    size_t pos = 0;
    for (;;) {
        const size_t p = b.find(sep, pos);
        if (p == NPOS) break;
        out.push_back(b.substr(pos, p - pos));
        pos = p + sep.size();
    }
    out.push_back(b.substr(pos));
    return out;
}

bool match(const string &str, const string &rules) {
    if (rules.empty()) return false;
    if (rules == "*") return true;
    string filename = str, m = rules;
    upper(filename);
    upper(m);
    for (const string &rule : split(m, ",")) {
        if (rule.empty()) continue;
        const std::vector<string> part = split(rule, "*");
        size_t pos1 = 0, jj = 0;
        for (; jj < part.size(); ++jj) {
            const string &p = part[jj];
            if (jj == 0 && filename.compare(0, p.size(), p) != 0) break;  // must start with the first piece
            if (jj == part.size() - 1) {                                  // and end with the last one
                // (the reference searches from length - piece length, which wraps when the
                // piece is longer than the name and then finds nothing)
                if (p.size() > filename.size() || filename.find(p, filename.size() - p.size()) == NPOS) break;
            }
            const size_t pos2 = filename.find(p, pos1);
            if (pos2 == NPOS) break;
            pos1 = pos2 + p.size();
        }
        if (jj == part.size()) return true;
    }
    return false;
}

// ---- crtsql ---------------------------------------------------------------------------------

// deleterchr(str,',') (public/_public.cpp:52-59): nothing is removed when the last byte that
// is not a comma stands at index 0.
void drop_trailing_commas(string &s) {
    const size_t pos = s.find_last_not_of(',');
    if (pos != 0) s.erase(pos + 1);  // (npos + 1 == 0: a string of commas only becomes empty)
}

string cname(const char *a, size_t cap) { return string(a, strnlen(a, cap)); }

int emit(const string &s, char *dst, size_t cap, size_t *len) {
    if (len) *len = s.size();
    if (!dst || cap == 0) return s.empty() ? MDC_OK : MDC_ERR_TOO_LARGE;
    const size_t n = s.size() < cap - 1 ? s.size() : cap - 1;
    std::memcpy(dst, s.data(), n);
    dst[n] = 0;
    return n == s.size() ? MDC_OK : MDC_ERR_TOO_LARGE;
}

}  // namespace

struct mdc_route {
    std::vector<mdc_xmltotable> entries;
    int truncated = 0;
};

extern "C" int mdc_route_load_buffer(const void *data, size_t n, mdc_route **out) {
    if (!out || (!data && n)) return mdc_internal_fail(MDC_ERR_ARG, "mdc_route_load_buffer: null argument");
    mdc_route *r = new (std::nothrow) mdc_route;
    if (!r) return mdc_internal_fail(MDC_ERR_NOMEM, "mdc_route_load_buffer: out of memory");
    for (const string &rec : records_of((const char *)data, n, "<endl/>")) {
        mdc_xmltotable e;
        std::memset(&e, 0, sizeof(e));
        string v;
        // (the char* form copies an empty string when the tag is missing: the field stays "")
        v.clear();
        field_of(rec, "filename", v);
        r->truncated += put(e.filename, sizeof(e.filename), v);
        v.clear();
        field_of(rec, "tname", v);
        r->truncated += put(e.tname, sizeof(e.tname), v);
        int_field_of(rec, "uptbz", e.uptbz);
        v.clear();
        field_of(rec, "execsql", v);
        r->truncated += put(e.execsql, sizeof(e.execsql), v);
        r->entries.push_back(e);
    }
    *out = r;
    return MDC_OK;
}

extern "C" int mdc_route_load(const char *inifilename, mdc_route **out) {
    if (!inifilename || !out) return mdc_internal_fail(MDC_ERR_ARG, "mdc_route_load: null argument");
    FILE *f = std::fopen(inifilename, "rb");
    if (!f) {
        mdc_internal_fail(MDC_FILE_OPEN_FAILED, (string("cannot open ") + inifilename).c_str());
        return MDC_FILE_OPEN_FAILED;
    }
    string buf;
    char tmp[65536];
    size_t k;
    while ((k = std::fread(tmp, 1, sizeof(tmp), f)) > 0) buf.append(tmp, k);
    const bool bad = std::ferror(f) != 0;
    std::fclose(f);
    if (bad) return mdc_internal_fail(MDC_ERR_IO, (string("cannot read ") + inifilename).c_str());
    return mdc_route_load_buffer(buf.data(), buf.size(), out);
}

extern "C" void mdc_route_free(mdc_route *r) { delete r; }
extern "C" int mdc_route_count(const mdc_route *r) { return r ? (int)r->entries.size() : -1; }
extern "C" int mdc_route_truncated(const mdc_route *r) { return r ? r->truncated : -1; }
extern "C" const mdc_xmltotable *mdc_route_entry(const mdc_route *r, int i) {
    return r && i >= 0 && (size_t)i < r->entries.size() ? &r->entries[(size_t)i] : nullptr;
}

extern "C" int mdc_matchstr(const char *str, const char *rules) { return str && rules && match(str, rules) ? 1 : 0; }

extern "C" int mdc_route_find(const mdc_route *r, const char *xmlfile) {
    if (!r || !xmlfile) return -1;
    for (size_t i = 0; i < r->entries.size(); ++i)
        if (match(xmlfile, cname(r->entries[i].filename, sizeof(r->entries[i].filename)))) return (int)i;
    return -1;
}

extern "C" int mdc_crtsql(const mdc_xmltotable *entry, const mdc_col *cols, int ncols, char *insertsql, size_t inscap,
                          size_t *inslen, char *updatesql, size_t updcap, size_t *updlen) {
    if (!entry || (!cols && ncols) || ncols < 0) return mdc_internal_fail(MDC_ERR_ARG, "mdc_crtsql: bad argument");
    const string tname = cname(entry->tname, sizeof(entry->tname));
    // "SEQ_" + the table name without its first two bytes ("T_"); the array is zeroed, so a
    // shorter name gives an empty tail
    const string seq = tname.size() > 2 ? tname.substr(2) : string();
    string names, binds;
    int colseq = 1;
    for (int i = 0; i < ncols; ++i) {
        const string col = cname(cols[i].colname, sizeof(cols[i].colname));
        const string type = cname(cols[i].datatype, sizeof(cols[i].datatype));
        if (col == "upttime") continue;  // filled by its default (sysdate)
        names += col + ",";
        if (col == "keyid")
            binds += "SEQ_" + seq + ".nextval,";
        else if (type == "date")
            binds += "to_date(:" + std::to_string(colseq++) + ",'yyyymmddhh24miss'),";
        else
            binds += ":" + std::to_string(colseq++) + ",";
    }
    drop_trailing_commas(names);
    drop_trailing_commas(binds);
    const string ins = "insert into " + tname + "(" + names + ") values(" + binds + ")";
    int rc = emit(ins, insertsql, inscap, inslen);
    if (entry->uptbz != 1) {
        if (updlen) *updlen = 0;
        return rc == MDC_OK ? MDC_OK : mdc_internal_fail(rc, "mdc_crtsql: insertsql does not fit");
    }

    string set = " set ", where = " where 1=1";
    colseq = 1;
    for (int i = 0; i < ncols; ++i) {
        const string col = cname(cols[i].colname, sizeof(cols[i].colname));
        const string type = cname(cols[i].datatype, sizeof(cols[i].datatype));
        if (cols[i].pkseq != 0 || col == "keyid") continue;
        if (col == "upttime")
            set += "upttime=sysdate,";
        else if (type == "date")
            set += col + "=to_date(:" + std::to_string(colseq++) + ",'yyyymmddhh24miss),";  // (sic, :393)
        else
            set += col + "=:" + std::to_string(colseq++) + ",";
    }
    drop_trailing_commas(set);
    // key columns in the order of their position in the key (m_vpkcols, tools/_tools.cpp:92-142)
    std::vector<int> pk;
    for (int i = 0; i < ncols; ++i)
        if (cols[i].pkseq != 0) pk.push_back(i);
    for (size_t a = 1; a < pk.size(); ++a)  // stable insertion sort by pkseq
        for (size_t b = a; b > 0 && cols[pk[b - 1]].pkseq > cols[pk[b]].pkseq; --b) std::swap(pk[b - 1], pk[b]);
    for (int i : pk) {
        const string col = cname(cols[i].colname, sizeof(cols[i].colname));
        const string type = cname(cols[i].datatype, sizeof(cols[i].datatype));
        if (type == "date")
            where += " and " + col + "=to_date(:" + std::to_string(colseq++) + ",'yyyymmddhh24miss')";
        else
            where += " and " + col + "=:" + std::to_string(colseq++);
    }
    const string upd = "update " + tname + set + where;
    const int rc2 = emit(upd, updatesql, updcap, updlen);
    if (rc != MDC_OK || rc2 != MDC_OK) return mdc_internal_fail(MDC_ERR_TOO_LARGE, "mdc_crtsql: statement text does not fit");
    return MDC_OK;
}

extern "C" int mdc_bind_plan(const mdc_xmltotable *entry, const mdc_col *cols, int ncols, int *ins, int *nins, int *upd,
                             int *nupd) {
    if (!entry || (!cols && ncols) || ncols < 0 || !ins || !nins)
        return mdc_internal_fail(MDC_ERR_ARG, "mdc_bind_plan: bad argument");
    auto is = [&](int i, const char *name) { return cname(cols[i].colname, sizeof(cols[i].colname)) == name; };
    int k = 0;
    for (int i = 0; i < ncols; ++i)
        if (!is(i, "upttime") && !is(i, "keyid")) ins[k++] = i;
    *nins = k;
    if (nupd) *nupd = 0;
    if (entry->uptbz != 1) return MDC_OK;
    if (!upd || !nupd) return mdc_internal_fail(MDC_ERR_ARG, "mdc_bind_plan: update plan wanted but no room given");
    k = 0;
    for (int i = 0; i < ncols; ++i)
        if (cols[i].pkseq == 0 && !is(i, "keyid") && !is(i, "upttime")) upd[k++] = i;
    for (int i = 0; i < ncols; ++i)
        if (cols[i].pkseq != 0) upd[k++] = i;
    *nupd = k;
    return MDC_OK;
}

// ------------------------------------------------------------------------------------------
// the execute loop in bulk (tools/xmltodb.cpp:263-306)

extern "C" int mdc_batch_bind_slabs(mdc_batch *b, const mdc_xmltotable *entry, const mdc_col *cols, int ncols,
                                    mdc_bind_slab *ins, int *nins, mdc_bind_slab *upd, int *nupd) {
    if (!b || !entry || !cols || ncols < 1 || !ins || !nins)
        return mdc_internal_fail(MDC_ERR_ARG, "mdc_batch_bind_slabs: bad argument");
    if (mdc_batch_ncols(b) != ncols) return mdc_internal_fail(MDC_ERR_ARG, "mdc_batch_bind_slabs: the batch has another number of columns");
    std::vector<int> pi((size_t)ncols), pu((size_t)ncols);
    int ni = 0, nu = 0;
    int rc = mdc_bind_plan(entry, cols, ncols, pi.data(), &ni, pu.data(), &nu);
    if (rc) return rc;
    if (nu && (!upd || !nupd)) return mdc_internal_fail(MDC_ERR_ARG, "mdc_batch_bind_slabs: update slabs wanted but no room given");
    auto fill = [&](const int *plan, int n, mdc_bind_slab *out) {
        for (int k = 0; k < n; ++k) {
            out[k].col = plan[k];
            out[k].pad = 0;
            const int r = mdc_batch_column_fixed(b, plan[k], &out[k].base, &out[k].stride);
            if (r) return r;
        }
        return (int)MDC_OK;
    };
    if ((rc = fill(pi.data(), ni, ins))) return rc;
    *nins = ni;
    if (nupd) *nupd = 0;
    if (nu) {
        if ((rc = fill(pu.data(), nu, upd))) return rc;
        *nupd = nu;
    }
    return MDC_OK;
}

// the error numbers that mean the connection is gone (tools/xmltodb.cpp:297-301)
static bool connection_lost(int code) { return code == 3113 || code == 3114 || code == 3135 || code == 16014; }

extern "C" int mdc_ingest_rows(const mdc_bind_slab *ins, int nins, const mdc_bind_slab *upd, int nupd, int64_t rows,
                               int uptbz, mdc_array_exec exec, void *ctx, mdc_ingest_counts *counts) {
    if (!counts || !exec || rows < 0 || nins < 0 || nupd < 0 || (nins && !ins) || (nupd && !upd))
        return mdc_internal_fail(MDC_ERR_ARG, "mdc_ingest_rows: bad argument");
    if (uptbz == 1 && !upd && rows) return mdc_internal_fail(MDC_ERR_ARG, "mdc_ingest_rows: uptbz is 1 but there are no update slabs");
    memset(counts, 0, sizeof *counts);
    counts->totalcount = rows;
    if (rows == 0) return MDC_OK;
    std::vector<int> code((size_t)rows, 0);
    if (exec(ctx, 0, ins, nins, rows, code.data()) != 0) {
        counts->result = 2;
        return MDC_OK;
    }
    // the decisions of :271-303, row by row in file order
    std::vector<int64_t> dup;
    for (int64_t r = 0; r < rows; ++r) {
        const int c = code[(size_t)r];
        if (c == 0) {
            ++counts->inscount;
        } else if (c == 1) {
            ++counts->duplicates;
            if (uptbz == 1) dup.push_back(r);
        } else {
            ++counts->insfailed;
            if (connection_lost(c)) {  // (the reference returns here: no row behind this one is looked at, nothing is committed)
                counts->result = 2;
                counts->fatalcode = c;
                return MDC_OK;
            }
        }
    }
    if (dup.empty()) return MDC_OK;
    // the rejected rows, in row order, in the update statement's bind order
    std::vector<std::vector<char>> store((size_t)nupd);
    std::vector<mdc_bind_slab> us((size_t)nupd);
    for (int k = 0; k < nupd; ++k) {
        const size_t st = upd[k].stride;
        store[(size_t)k].resize(dup.size() * st);
        for (size_t i = 0; i < dup.size(); ++i) memcpy(&store[(size_t)k][i * st], upd[k].base + (size_t)dup[i] * st, st);
        us[(size_t)k] = upd[k];
        us[(size_t)k].base = store[(size_t)k].data();
    }
    std::vector<int> ucode(dup.size(), 0);
    if (exec(ctx, 1, us.data(), nupd, (int64_t)dup.size(), ucode.data()) != 0) {
        counts->result = 2;
        return MDC_OK;
    }
    for (const int c : ucode) {
        if (c == 0)
            ++counts->uptcount;
        else
            ++counts->uptfailed;  // (data the table does not take: logged by the reference, the file goes on, :281-286)
    }
    return MDC_OK;
}
