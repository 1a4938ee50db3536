// qname.hip -- the QNAME passes, native on the HOST.  No device code in this file.
// Replaces the per-line Python of uq.py:348-352, 394-444 (prefix / suffix / separator inference),
// 555-678 (column typing) and 717-736 (column encoding).  The logic is order-dependent string
// heuristics, so it is restated sequentially, statement for statement; what it cannot reproduce
// exactly (separators that are regex metacharacters, integers beyond int64, a QNAME whose separators
// do not appear in the inferred order) is reported as status 1 and the Python implementation
// (uq_amd/qname.py) takes over.  The column arrays it produces go to the GPU for sort/unique/gather.
#include <algorithm>
#include <string>
#include <unordered_set>
#include <vector>
#include "common.h"

namespace {
struct Column {
    int format = 0;                 // 0 mapping, 1 integers
    std::unordered_set<std::string> set;
    std::vector<std::string> map;   // sorted, final
    long long mn = 0, mx = 0;
    int itemsize = 1;
    bool offset = false;
    std::string last; bool has_last = false;
};
}  // namespace

struct uq_qname {
    std::string prefix, suffix, separators, json;
    std::vector<Column> cols;
    std::vector<std::vector<uint8_t>> data;
    uint64_t n = 0;
};

namespace {
inline bool is_space(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13); }

// Python 2 int(str): optional whitespace, optional sign, decimal digits.  Returns false on ValueError;
// sets *big when the value does not fit comfortably in int64 (caller falls back to Python).
bool parse_int(const char* s, size_t n, long long* out, bool* big) {
    size_t i = 0;
    while (i < n && is_space((unsigned char)s[i])) ++i;
    while (n > i && is_space((unsigned char)s[n - 1])) --n;
    if (i >= n) return false;
    bool neg = false;
    if (s[i] == '+' || s[i] == '-') { neg = s[i] == '-'; ++i; }
    if (i >= n) return false;
    if (n - i > 18) {
        for (size_t k = i; k < n; ++k)
            if (s[k] < '0' || s[k] > '9') return false;
        *big = true;
        return true;
    }
    long long v = 0;
    for (; i < n; ++i) {
        if (s[i] < '0' || s[i] > '9') return false;
        v = v * 10 + (s[i] - '0');
    }
    *out = neg ? -v : v;
    return true;
}

size_t count_char(const char* s, size_t n, char c) {
    size_t k = 0;
    for (size_t i = 0; i < n; ++i) k += s[i] == c;
    return k;
}

bool regex_special(char c) {
    switch (c) {
        case '.': case '^': case '$': case '*': case '+': case '?': case '{': case '}': case '[': case ']':
        case '\\': case '|': case '(': case ')': case '-': return true;
        default: return false;
    }
}

void json_escape(const std::string& s, std::string& o) {
    o += '"';
    char buf[8];
    for (unsigned char c : s) {
        if (c == '"' || c == '\\') { o += '\\'; o += (char)c; }
        else if (c < 0x20 || c >= 0x7F) { snprintf(buf, sizeof(buf), "\\u%04x", c); o += buf; }
        else o += (char)c;
    }
    o += '"';
}

struct Ladder { unsigned long long lim; int itemsize; };
const Ladder kLadder[4] = {{255ull, 1}, {65535ull, 2}, {4294967295ull, 4}, {18446744073709551615ull, 8}};
}  // namespace

// status: 0 = done; 1 = use the Python implementation; 2 = the reference itself refuses this input
// (message in uq_last_error()).
extern "C" int uq_qname_analyse(const uint8_t* h_buf, const uint64_t* ls, uint64_t n, uq_qname** out, int* h_status) {
    UQ_REQUIRE(h_buf && ls && out && h_status, "uq_qname_analyse: null argument");
    *out = nullptr; *h_status = 2;
    UQ_REQUIRE(n >= 1, "uq_qname_analyse: no reads");
    auto name = [&](uint64_t r, size_t* len) { *len = (size_t)(ls[4 * r + 1] - ls[4 * r] - 1); return (const char*)h_buf + ls[4 * r]; };
    size_t l1; const char* line1 = name(0, &l1);
    if (l1 == 0 || line1[0] != '@') { uq_set_error("ERROR: This does not look like a FASTA/FASTQ file! (first line does not start with @)"); return 0; }
    uq_qname* Q = new uq_qname();
    Q->n = n;
    // ---- pass 1: uq.py:348-352, 394-413
    std::string prefix(line1, l1), suffix(line1, l1);
    std::vector<std::pair<char, long long>> seps;      // insertion-ordered dict
    bool not_sep[256] = {false};
    auto find_sep = [&](char c) { for (size_t i = 0; i < seps.size(); ++i) if (seps[i].first == c) return (int)i; return -1; };
    for (uint64_t r = 1; r < n; ++r) {
        size_t ql; const char* q = name(r, &ql);
        if (!(ql >= prefix.size() && memcmp(q, prefix.data(), prefix.size()) == 0)) {
            for (size_t idx = 0; idx < prefix.size(); ++idx) {
                if (idx >= ql) { uq_set_error("IndexError: a QNAME is a strict prefix of the common prefix (the reference crashes here, uq.py:397)"); delete Q; return 0; }
                if (prefix[idx] != q[idx]) {
                    for (size_t k = idx; k < prefix.size(); ++k) {
                        const char c = prefix[k];
                        if (!not_sep[(unsigned char)c]) {
                            int f = find_sep(c);
                            if (f < 0) seps.push_back({c, 1}); else seps[f].second += 1;
                        }
                    }
                    prefix.resize(idx);
                    break;
                }
            }
        }
        if (!(ql >= suffix.size() && memcmp(q + ql - suffix.size(), suffix.data(), suffix.size()) == 0)) {
            const size_t sl = suffix.size();
            for (size_t idx = 0; idx < sl; ++idx) {
                if (idx >= ql) { uq_set_error("IndexError: a QNAME is shorter than the common suffix (the reference crashes here, uq.py:405)"); delete Q; return 0; }
                if (suffix[sl - 1 - idx] != q[ql - 1 - idx]) {
                    suffix = idx == 0 ? std::string() : suffix.substr(sl - idx);
                    break;
                }
            }
        }
        if (!seps.empty()) {
            const size_t pl = prefix.size() < ql ? prefix.size() : ql;
            for (size_t i = 0; i < seps.size();) {
                if ((long long)count_char(q + pl, ql - pl, seps[i].first) != seps[i].second) {
                    not_sep[(unsigned char)seps[i].first] = true;
                    seps.erase(seps.begin() + i);
                } else ++i;
            }
        }
    }
    for (size_t i = 0; i < seps.size();) {                                    // uq.py:428-431
        const long long c = (long long)count_char(suffix.data(), suffix.size(), seps[i].first);
        if (c) { seps[i].second -= c; if (seps[i].second == 0) { seps.erase(seps.begin() + i); continue; } }
        ++i;
    }
    if (seps.empty()) { uq_set_error("ERROR: the QNAMEs share no constant-count separator; the reference cannot encode such files either (SURVEY.md Q13)"); delete Q; return 0; }
    bool in_set[256] = {false};
    for (auto& s : seps) { in_set[(unsigned char)s.first] = true; if (regex_special(s.first)) { delete Q; *h_status = 1; return 0; } }
    auto order_seps = [&](const char* q, size_t ql) {                         // uq.py:433-436
        std::string o;
        const long long end = (long long)ql - 1 - (long long)suffix.size();
        for (long long i = (long long)prefix.size(); i < end; ++i)
            if (in_set[(unsigned char)q[i]]) o += q[i];
        return o;
    };
    size_t ll; const char* last = name(n - 1, &ll);
    const std::string separators = order_seps(last, ll);
    if (separators != order_seps(line1, l1)) {
        uq_set_error("ERROR: Sorry, the separators used in this file's QNAME/headers are so unusual/improbable that the reference gives up; so does this implementation");
        delete Q; return 0;
    }
    if (separators.empty()) { delete Q; *h_status = 1; return 0; }            // re.split('') territory: let Python decide
    // ---- pass 2: uq.py:571-638.  Fields = pieces of name[len(prefix) : len - len(suffix)] between the separators
    const size_t ncols = separators.size() + 1;
    Q->cols.resize(ncols);
    std::vector<uint32_t> fo((size_t)n * (ncols + 1));                        // field offsets inside each QNAME line
    uint64_t target = 10000;
    auto check_format = [&](uint64_t entries_read, int* status) {             // uq.py:586-602
        for (auto& c : Q->cols) {
            if (c.format == 0 && c.set.size() > entries_read / 10) {
                bool first = true;
                for (auto& s : c.set) {
                    long long v = 0; bool big = false;
                    if (!parse_int(s.data(), s.size(), &v, &big)) { *status = 2; uq_set_error("Encoding QNAMEs as strings has not been implimented yet."); return; }
                    if (big) { *status = 1; return; }
                    if (first) { c.mn = c.mx = v; first = false; } else { if (v < c.mn) c.mn = v; if (v > c.mx) c.mx = v; }
                }
                c.format = 1;
                c.set.clear(); c.has_last = false;
            }
        }
    };
    int status = 0;
    for (uint64_t r = 0; r < n && status == 0; ++r) {
        size_t ql; const char* q = name(r, &ql);
        const size_t start = prefix.size(), end = ql >= suffix.size() ? ql - suffix.size() : 0;
        uint32_t* f = &fo[(size_t)r * (ncols + 1)];
        size_t k = 0;
        f[0] = (uint32_t)start;
        for (size_t i = start; i < end; ++i) {
            if (in_set[(unsigned char)q[i]]) {
                if (k >= separators.size() || q[i] != separators[k]) { status = 1; break; }   // not the inferred order: regex semantics needed
                f[++k] = (uint32_t)(i + 1);
            }
        }
        if (status) break;
        if (k != separators.size() || end < start) { status = 1; break; }
        f[ncols] = (uint32_t)(end + 1);
        for (size_t c = 0; c < ncols; ++c) {
            Column& col = Q->cols[c];
            const char* s = q + f[c];
            const size_t sl = f[c + 1] - 1 - f[c];
            if (col.format == 0) {
                if (!(col.has_last && col.last.size() == sl && memcmp(col.last.data(), s, sl) == 0)) {
                    col.last.assign(s, sl); col.has_last = true;
                    col.set.insert(col.last);
                }
            } else {
                long long v = 0; bool big = false;
                if (!parse_int(s, sl, &v, &big)) { status = 2; uq_set_error("Encoding QNAMEs as strings has not been implimented yet."); break; }
                if (big) { status = 1; break; }
                if (v < col.mn) col.mn = v; else if (v > col.mx) col.mx = v;
            }
        }
        if (status == 0 && r == target) { check_format(r, &status); target *= 2; }
    }
    if (status == 0) check_format(n - 1, &status);                            // uq.py:638
    if (status) { delete Q; *h_status = status; return 0; }
    // ---- final typing: uq.py:641-670
    for (auto& c : Q->cols) {
        if (c.format == 0) {
            unsigned long long map_len = c.set.size();
            for (auto& l : kLadder) if (map_len <= l.lim) { map_len = l.lim; c.itemsize = l.itemsize; break; }
            bool all_int = true, first = true, big = false;
            long long mn = 0, mx = 0;
            for (auto& s : c.set) {
                long long v = 0;
                if (!parse_int(s.data(), s.size(), &v, &big)) { all_int = false; break; }
                if (big) break;
                if (first) { mn = mx = v; first = false; } else { if (v < mn) mn = v; if (v > mx) mx = v; }
            }
            if (big) { delete Q; *h_status = 1; return 0; }
            if (all_int && !c.set.empty() && (unsigned long long)(mx - mn) <= map_len) {
                c.format = 1; c.mn = mn; c.mx = mx;
                c.offset = mn < 0 || (unsigned long long)mx > map_len;
                c.set.clear();
            } else {
                c.map.assign(c.set.begin(), c.set.end());
                std::sort(c.map.begin(), c.map.end());
                c.set.clear();
            }
        } else {
            unsigned long long int_len = (unsigned long long)(c.mx - c.mn);
            for (auto& l : kLadder) if (int_len <= l.lim) { int_len = l.lim; c.itemsize = l.itemsize; break; }
            c.offset = c.mn < 0 || (unsigned long long)c.mx > int_len;
        }
    }
    // ---- pass 4: uq.py:717-736
    Q->data.resize(ncols);
    for (size_t c = 0; c < ncols; ++c) {
        Column& col = Q->cols[c];
        Q->data[c].resize((size_t)n * col.itemsize);
        uint8_t* d = Q->data[c].data();
        std::string key; size_t last_idx = 0; bool has = false;
        for (uint64_t r = 0; r < n; ++r) {
            size_t ql; const char* q = name(r, &ql);
            const uint32_t* f = &fo[(size_t)r * (ncols + 1)];
            const char* s = q + f[c];
            const size_t sl = f[c + 1] - 1 - f[c];
            unsigned long long v;
            if (col.format == 0) {
                if (!(has && key.size() == sl && memcmp(key.data(), s, sl) == 0)) {
                    key.assign(s, sl); has = true;
                    last_idx = (size_t)(std::lower_bound(col.map.begin(), col.map.end(), key) - col.map.begin());   // bisect_left
                }
                v = last_idx;
            } else {
                long long iv = 0; bool big = false;
                parse_int(s, sl, &iv, &big);
                v = (unsigned long long)(col.offset ? iv - col.mn : iv);
            }
            memcpy(d + (size_t)r * col.itemsize, &v, col.itemsize);       // little-endian host
        }
    }
    // ---- metadata as JSON (the QNAME_* part of config.json, uq.py:692-695)
    std::string& j = Q->json;
    j = "{\"prefix\":"; json_escape(prefix, j);
    j += ",\"suffix\":"; json_escape(suffix, j);
    j += ",\"separators\":"; json_escape(separators, j);
    j += ",\"columns\":[";
    static const char* dt[9] = {"", "uint8", "uint16", "", "uint32", "", "", "", "uint64"};
    for (size_t c = 0; c < ncols; ++c) {
        Column& col = Q->cols[c];
        if (c) j += ',';
        j += "{\"name\":\"QNAME_" + std::to_string(c + 1) + "\",\"dtype\":\"" + dt[col.itemsize] + "\",";
        if (col.format == 1) {
            j += "\"format\":\"integers\",\"min\":" + std::to_string(col.mn) + ",\"max\":" + std::to_string(col.mx) + ",\"offset\":" + (col.offset ? "true" : "false") + "}";
        } else {
            j += "\"format\":\"mapping\",\"map\":[";
            for (size_t k = 0; k < col.map.size(); ++k) { if (k) j += ','; json_escape(col.map[k], j); }
            j += "]}";
        }
    }
    j += "]}";
    Q->prefix = prefix; Q->suffix = suffix; Q->separators = separators;
    *out = Q; *h_status = 0;
    return 0;
}

extern "C" int uq_qname_json(const uq_qname* q, const char** h_json) {
    UQ_REQUIRE(q && h_json, "uq_qname_json: null argument");
    *h_json = q->json.c_str();
    return 0;
}

extern "C" int uq_qname_column(const uq_qname* q, int col, void* h_out, uint64_t capacity_bytes) {
    UQ_REQUIRE(q && h_out && col >= 0 && (size_t)col < q->data.size(), "uq_qname_column: bad argument");
    UQ_REQUIRE(capacity_bytes >= q->data[col].size(), "uq_qname_column: buffer too small");
    memcpy(h_out, q->data[col].data(), q->data[col].size());
    return 0;
}

extern "C" int uq_qname_free(uq_qname* q) { delete q; return 0; }
