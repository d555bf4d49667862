// The ingest loop of tools/xmltodb.cpp:263-304 on top of mdc_xmltodb.hpp: open a file of tagged
// records, walk the records, split each into the column texts that would be bound to the insert
// statement, and "execute" (here: print).  Also shows the typed getxmlbuffer forms and the
// delimited twin.
//
//   xmltodb_loop xml  <file> <name:type:collen[:typed],...> [nocarry]
//   xmltodb_loop csv  <file> <sep> <nfields> [typed,typed,...]
//
// Output, one line per record: the row number, then per column "found valid hex(cell) hex(value)".
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <sstream>

#include "mdc_xmltodb.hpp"

static std::vector<std::string> split(const std::string &s, char sep) {
    std::vector<std::string> out;
    std::stringstream ss(s);
    std::string item;
    while (std::getline(ss, item, sep)) out.push_back(item);
    return out;
}

static void print_hex(const char *p, uint32_t n) {
    if (n == 0) std::printf("-");
    for (uint32_t i = 0; i < n; ++i) std::printf("%02x", (unsigned)(unsigned char)p[i]);
}

static void print_columns(mdc_batch *b, int64_t row, int ncols, const std::vector<int> &typed) {
    const uint32_t *valid, *found;
    mdc_batch_host_masks(b, &valid, &found);
    const int words = mdc_batch_words_per_row(b);
    for (int c = 0; c < ncols; ++c) {
        const char *p = nullptr;
        uint32_t len = 0;
        mdc_batch_cell(b, row, c, &p, &len);
        const unsigned f = (found[row * words + (c >> 5)] >> (c & 31)) & 1u;
        const unsigned v = (valid[row * words + (c >> 5)] >> (c & 31)) & 1u;
        std::printf(" %u %u ", f, v);
        print_hex(p, len);
        const uint64_t *val = nullptr;
        if (typed[c] != MDC_T_NONE && mdc_batch_host_values(b, c, &val) == MDC_OK && val)
            std::printf(" %016" PRIx64, val[row]);
        else
            std::printf(" -");
    }
    std::printf("\n");
}

int main(int argc, char **argv) {
    if (argc < 4) {
        std::fprintf(stderr, "usage: %s xml <file> <name:type:collen[:typed],...> [nocarry]\n"
                             "       %s csv <file> <sep> <nfields> [typed,...]\n"
                             "       %s route <inifile> <xmlfile name> <name:type:collen:typed:pkseq,...>\n",
                     argv[0], argv[0], argv[0]);
        return 2;
    }
    const std::string mode = argv[1], filename = argv[2];
    if (mode == "route") {
        // tools/xmltodb.cpp:141, :236-246: load the routing entries, find the file's entry, build the
        // statements and the bind order for the table's columns
        if (argc < 5) return 2;
        std::vector<mdc::st_col> cols;
        for (const std::string &spec : split(argv[4], ',')) {
            const std::vector<std::string> f = split(spec, ':');
            if (f.size() < 3) return 2;
            cols.emplace_back(f[0], f[1], std::atoi(f[2].c_str()), f.size() > 3 ? std::atoi(f[3].c_str()) : MDC_T_NONE,
                              f.size() > 4 ? std::atoi(f[4].c_str()) : 0);
        }
        mdc::cxmltotable tab;
        if (!tab.loadxmltotable(filename)) {
            std::fprintf(stderr, "loadxmltotable(%s) failed: %s\n", filename.c_str(), mdc_last_error());
            return 5;
        }
        if (!tab.findxmltotable(argv[3])) return 1;  // "no routing entry"
        std::vector<int> bi, bu;
        if (!tab.crtsql(cols) || !tab.bindplan(cols, bi, bu)) return 2;
        std::printf("tname %s\nuptbz %d\nexecsql %s\ninsert %s\n", tab.stxmltotable.tname, tab.stxmltotable.uptbz,
                    tab.stxmltotable.execsql, tab.insertsql.c_str());
        if (tab.stxmltotable.uptbz == 1) std::printf("update %s\n", tab.updatesql.c_str());
        std::printf("bind_insert");
        for (int c : bi) std::printf(" %d", c);
        std::printf("\nbind_update");
        for (int c : bu) std::printf(" %d", c);
        std::printf("\n");
        return 0;
    }
    if (mode == "xml") {
        std::vector<mdc::st_col> cols;
        std::vector<int> typed;
        for (const std::string &spec : split(argv[3], ',')) {
            const std::vector<std::string> f = split(spec, ':');
            if (f.size() < 3) return 2;
            cols.emplace_back(f[0], f[1], std::atoi(f[2].c_str()), f.size() > 3 ? std::atoi(f[3].c_str()) : MDC_T_NONE);
            typed.push_back(cols.back().typed);
        }
        const bool carry = !(argc > 4 && std::string(argv[4]) == "nocarry");
        mdc::cxmlfile ifile(cols, carry);
        if (!ifile.open(filename)) {  // tools/xmltodb.cpp:256-261
            std::fprintf(stderr, "open(%s) failed: %s\n", filename.c_str(), ifile.errmsg().c_str());
            return ifile.errcode() == MDC_FILE_OPEN_FAILED ? 5 : 1;
        }
        std::string strbuffer;
        std::vector<std::string> vcolvalue;
        int64_t totalcount = 0;
        while (ifile.readline(strbuffer)) {  // tools/xmltodb.cpp:264
            ++totalcount;
            ifile.splitbuffer(vcolvalue);    // tools/xmltodb.cpp:268
            std::printf("%" PRId64, ifile.row());
            print_columns(ifile.batch(), ifile.row(), ifile.colcount(), typed);
            // the texts handed to the statement are the cells themselves
            for (int c = 0; c < ifile.colcount(); ++c) {
                const char *p;
                uint32_t len;
                if (!ifile.cell(c, &p, &len) || vcolvalue[c] != std::string(p, len)) return 3;
            }
        }
        std::printf("total %" PRId64 " of %" PRId64 "\n", totalcount, ifile.totalcount());
        return 0;
    }
    if (mode == "csv" && argc >= 5) {
        const int nfields = std::atoi(argv[4]);
        std::vector<int> typed((size_t)nfields, MDC_T_NONE);
        if (argc > 5) {
            const std::vector<std::string> t = split(argv[5], ',');
            for (size_t i = 0; i < t.size() && i < typed.size(); ++i) typed[i] = std::atoi(t[i].c_str());
        }
        mdc::ccmdfile ifile(argv[3], false, nfields, typed);
        if (!ifile.open(filename)) {
            std::fprintf(stderr, "open(%s) failed: %s\n", filename.c_str(), ifile.errmsg().c_str());
            return ifile.errcode() == MDC_FILE_OPEN_FAILED ? 5 : 1;
        }
        while (ifile.next()) {
            std::printf("%" PRId64, ifile.row());
            print_columns(ifile.batch(), ifile.row(), nfields, typed);
        }
        std::printf("total %" PRId64 " of %" PRId64 "\n", ifile.row() + 1, ifile.totalcount());
        return 0;
    }
    return 2;
}
