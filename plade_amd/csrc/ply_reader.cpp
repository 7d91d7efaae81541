// plade_amd/csrc/ply_reader.cpp -- see ply_reader.h.
//
// What is accepted, what is refused and which bits come out follow the reference's ingest, pinned by
// tests/test_ply_reader.py against that ingest compiled from its own sources (oracle/_ref: rply.c + ply_reader.cpp):
//   header   rply's grammar (code/3rd_party/rply/rply.c:330-353 magic, :395-421 header loop, :1193-1283 format / comment /
//            obj_info / element / property): words separated by blanks " \n\r\t" whatever the line structure, `format <mode>
//            1.*`, comments and obj_infos run to the end of their line, anything else between elements is an error;
//   body     elements in file order, one value per word in ascii files -- whole-word strtol / strtod with the range of the
//            DECLARED type (rply.c:1420-1482: an out-of-range or non-finite number fails the file) --, fixed-size chunks in the
//            two binary modes (rply.c:1484-1538); every element is parsed to its end (a malformed face list fails the cloud);
//   cloud    PlyReader::collect_elements (code/PLADE/ply_reader.cpp:277-386) + load_ply_cloud (code/PLADE/util.cpp:1505-1546):
//            the `vertex` element's x y z (else X Y Z) and nx ny nz, each triple complete and of a FLOATING type (float /
//            double; integer-typed coordinates are not points), every value through double to float; no points, points
//            without normals or an empty cloud fail.
// One difference in mechanism, none in result: a binary file in the host's byte order whose vertex element is exactly
// `float x y z nx ny nz` -- PLADE's own sample data, and what the CLI's batch mode reads 128 times per 64-pair list -- is one
// fread into the caller's array instead of six callbacks per point.
#include "ply_reader.h"
#include "plade_hip.h"

#include <algorithm>
#include <cctype>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace plade {

namespace {

// rply.c:76-81: both spellings; the index & 7 is the storage type
const char *const kTypeNames[16] = {"int8", "uint8", "int16", "uint16", "int32", "uint32", "float32", "float64",
                                    "char", "uchar", "short", "ushort", "int", "uint", "float", "double"};
const int kTypeSize[8] = {1, 1, 2, 2, 4, 4, 4, 8};
constexpr size_t kWordMax = 256, kLineMax = 1024;     // rply.c:63-64

int type_index(const std::string &t) {
    for (int i = 0; i < 16; ++i)
        if (t == kTypeNames[i]) return i & 7;
    return -1;
}
inline bool floating(int type) { return type == 6 || type == 7; }   // ply_reader.cpp:330, :346

struct Prop {
    std::string name;
    int type = -1;                         // scalar: storage type
    bool is_list = false;
    int count_type = -1, item_type = -1;
};
struct Elem {
    std::string name;
    long long count = 0;
    std::vector<Prop> props;
};

inline bool blank(unsigned char c) { return c == ' ' || c == '\n' || c == '\r' || c == '\t'; }   // rply.c:962

// a cursor over bytes that are in memory
struct Cursor {
    const unsigned char *p, *end;
    // the next word (rply.c:955-1002); false at the end of the data or for a word rply refuses
    bool word(std::string &w) {
        while (p < end && blank(*p)) ++p;
        if (p >= end) return false;
        const unsigned char *s = p;
        while (p < end && !blank(*p) && *p != 0) ++p;
        w.assign(reinterpret_cast<const char *>(s), (size_t)(p - s));
        if (p < end) ++p;                  // the character behind the word is consumed with it (ply_finish_word)
        return !w.empty() && w.size() < kWordMax;
    }
    // the rest of the line (rply.c:1012-1045)
    bool line() {
        const unsigned char *s = p;
        while (p < end && *p != '\n') ++p;
        if (p >= end) return false;
        const size_t len = (size_t)(p - s);
        ++p;
        return len < kLineMax;
    }
    bool chunk(void *dst, size_t n, bool swap) {
        if ((size_t)(end - p) < n) return false;
        unsigned char *d = static_cast<unsigned char *>(dst);
        if (swap) for (size_t i = 0; i < n; ++i) d[i] = p[n - 1 - i];
        else memcpy(d, p, n);
        p += n;
        return true;
    }
};

struct Header {
    std::vector<Elem> elems;
    int mode = -1;                          // 0 ascii, 1 binary little endian, 2 binary big endian
    size_t data_offset = 0;
};

// false + msg: not a header rply accepts; `incomplete`: the bytes ran out before end_header (the caller may have more)
bool parse_header(const unsigned char *buf, size_t n, Header &h, std::string &msg, bool &incomplete) {
    incomplete = false;
    auto eof = [&](const char *m) { incomplete = true; msg = m; return false; };
    if (n < 5) return eof("Unable to read magic number from file");
    if (buf[0] != 'p' || buf[1] != 'l' || buf[2] != 'y' || !isspace(buf[3])) { msg = "Wrong magic number. Expected 'ply'"; return false; }
    const bool rn = buf[3] == '\r' && buf[4] == '\n';
    Cursor c{buf + 3, buf + n};
    std::string w;
    if (!c.word(w)) return eof("Unexpected end of file");
    if (w != "format") { msg = "Invalid file format"; return false; }
    if (!c.word(w)) return eof("Unexpected end of file");
    h.mode = w == "ascii" ? 0 : w == "binary_little_endian" ? 1 : w == "binary_big_endian" ? 2 : -1;
    if (h.mode < 0) { msg = "Invalid file format"; return false; }
    if (!c.word(w)) return eof("Unexpected end of file");
    if (strncmp(w.c_str(), "1.0", 2) != 0) { msg = "Invalid file format"; return false; }      // rply.c:1208 (the reference's patch)
    if (!c.word(w)) return eof("Unexpected end of file");
    // rply.c:404-411 with :1213-1283 inlined: `w` is always the word to be classified next
    bool in_element = false;
    for (;;) {
        if (w == "end_header") break;
        if (w == "comment" || w == "obj_info") {
            if (!c.line()) { if (c.p >= c.end) return eof("Unexpected end of file"); msg = "Line too long"; return false; }
        } else if (w == "element") {
            Elem e;
            if (!c.word(e.name)) return eof("Unexpected end of file");
            if (!c.word(w)) return eof("Unexpected end of file");
            char *endp = nullptr;
            const long long cnt = strtoll(w.c_str(), &endp, 10);      // sscanf("%ld"): a leading number is enough
            if (endp == w.c_str()) { msg = "Expected number got '" + w + "'"; return false; }
            e.count = cnt;
            h.elems.push_back(e);
            in_element = true;
        } else if (w == "property" && in_element) {
            Prop p;
            std::string t;
            if (!c.word(t)) return eof("Unexpected end of file");
            if (t == "list") {
                std::string ct, it;
                if (!c.word(ct) || !c.word(it)) return eof("Unexpected end of file");
                p.is_list = true; p.count_type = type_index(ct); p.item_type = type_index(it);
                if (p.count_type < 0 || p.item_type < 0) { msg = "Unexpected token '" + (p.count_type < 0 ? ct : it) + "'"; return false; }
            } else {
                p.type = type_index(t);
                if (p.type < 0) { msg = "Unexpected token '" + t + "'"; return false; }
            }
            if (!c.word(p.name)) return eof("Unexpected end of file");
            h.elems.back().props.push_back(p);
        } else { msg = "Unexpected token '" + w + "'"; return false; }
        if (!c.word(w)) return eof("Unexpected end of file");
    }
    if (rn) {                                 // rply.c:412-419: "\r\n" files have one more character in front of the data
        if (c.p >= c.end) return eof("Unexpected end of file");
        ++c.p;
    }
    h.data_offset = (size_t)(c.p - buf);
    return true;
}

bool host_is_little_endian() {
    const uint16_t x = 1;
    return *reinterpret_cast<const unsigned char *>(&x) == 1;
}

// one value of the declared storage type (rply.c:1420-1538)
bool read_value(Cursor &c, int type, int mode, bool swap, std::string &w, double &v) {
    if (mode == 0) {
        if (!c.word(w)) return false;
        char *endp = nullptr;
        if (type < 6) {
            const long x = strtol(w.c_str(), &endp, 10);
            if (*endp) return false;
            static const double lo[6] = {-128.0, 0.0, -32768.0, 0.0, -2147483648.0, 0.0};
            static const double hi[6] = {127.0, 255.0, 32767.0, 65535.0, 2147483647.0, 4294967295.0};
            v = (double)x;
            return !(v > hi[type] || v < lo[type]);
        }
        v = strtod(w.c_str(), &endp);
        if (*endp) return false;
        const double lim = type == 6 ? (double)FLT_MAX : DBL_MAX;
        return !(v < -lim || v > lim);
    }
    switch (type) {
        case 0: { int8_t x; if (!c.chunk(&x, 1, false)) return false; v = x; return true; }
        case 1: { uint8_t x; if (!c.chunk(&x, 1, false)) return false; v = x; return true; }
        case 2: { int16_t x; if (!c.chunk(&x, 2, swap)) return false; v = x; return true; }
        case 3: { uint16_t x; if (!c.chunk(&x, 2, swap)) return false; v = x; return true; }
        case 4: { int32_t x; if (!c.chunk(&x, 4, swap)) return false; v = x; return true; }
        case 5: { uint32_t x; if (!c.chunk(&x, 4, swap)) return false; v = x; return true; }
        case 6: { float x; if (!c.chunk(&x, 4, swap)) return false; v = x; return true; }
        default: return c.chunk(&v, 8, swap);
    }
}

struct FileCloser { FILE *f; ~FileCloser() { if (f) fclose(f); } };

bool read_body(const std::string &path, std::vector<float> &out, std::string &err, std::vector<std::string> *warnings,
               const std::function<void()> *before_grow) {
    auto size_to = [&](size_t floats) {
        if (out.size() == floats) return;
        if (floats > out.capacity() && before_grow) (*before_grow)();     // the allocation is about to move
        out.resize(floats);
    };
    // `out` may be a reused staging array: it is only resized when the point count differs (no re-zeroing)
    auto fail = [&](const std::string &m) { out.clear(); err = m; return false; };
    FileCloser fc{fopen(path.c_str(), "rb")};
    FILE *f = fc.f;
    if (!f) return fail("failed to open ply file: " + path);
    long long file_size = -1;
    if (fseek(f, 0, SEEK_END) == 0) file_size = ftell(f);
    rewind(f);
    if (file_size < 0) return fail("failed to read ply header");
    // the header from the first 64 KB (all of the file if that is not enough)
    std::vector<unsigned char> data((size_t)std::min<long long>(file_size, 65536));
    if (!data.empty() && fread(data.data(), 1, data.size(), f) != data.size()) return fail("failed to read ply header");
    Header h;
    std::string msg;
    bool incomplete = false, whole = (long long)data.size() == file_size;
    if (!parse_header(data.data(), data.size(), h, msg, incomplete)) {
        if (!incomplete || whole) return fail("failed to read ply header (" + msg + "): " + path);
        const size_t have = data.size();
        data.resize((size_t)file_size);
        if (fread(data.data() + have, 1, data.size() - have, f) != data.size() - have) return fail("failed to read ply header");
        whole = true;
        h = Header();
        if (!parse_header(data.data(), data.size(), h, msg, incomplete)) return fail("failed to read ply header (" + msg + "): " + path);
    }
    const bool swap = (h.mode == 1 && !host_is_little_endian()) || (h.mode == 2 && host_is_little_endian());
    // which element, which columns (ply_reader.cpp:277-386, util.cpp:1519-1543): the first element called "vertex" that has
    // instances; elements without instances have no properties for the reference (ply_reader.cpp:101-102)
    int vertex = -1;
    for (size_t i = 0; i < h.elems.size(); ++i)
        if (h.elems[i].name == "vertex" && h.elems[i].count > 0 && !h.elems[i].props.empty()) { vertex = (int)i; break; }
    // col[q]: the column whose values become coordinate q; kZeros: the reference delivers zeros there (below); -1: no such property
    constexpr int kZeros = -2;
    int col[6] = {-1, -1, -1, -1, -1, -1};
    if (vertex >= 0) {
        const Elem &e = h.elems[vertex];
        // A name that occurs once is that column if its type is floating.  A name that occurs several times: the reference
        // creates one array per property but rply attaches every callback to the FIRST property of the name (rply.c:424-438,
        // ply_find_property), the last registration winning -- the first column's values land in the LAST array of the name and
        // the others stay zero-filled --, and collect_elements then takes the first floating-typed array of the name
        // (ply_reader.cpp:217-230): data only if that is the last one.
        auto find = [&](const char *name) {
            int first = -1, last = -1, chosen = -1;
            for (size_t k = 0; k < e.props.size(); ++k) {
                if (e.props[k].is_list || e.props[k].name != name) continue;
                if (first < 0) first = (int)k;
                last = (int)k;
                if (chosen < 0 && floating(e.props[k].type)) chosen = (int)k;
            }
            if (chosen < 0) return -1;
            return chosen == last ? first : kZeros;
        };
        int a = find("x"), b = find("y"), c = find("z");
        if (a == -1 || b == -1 || c == -1) { a = find("X"); b = find("Y"); c = find("Z"); }
        if (a != -1 && b != -1 && c != -1) { col[0] = a; col[1] = b; col[2] = c; }
        a = find("nx"); b = find("ny"); c = find("nz");
        if (a != -1 && b != -1 && c != -1) { col[3] = a; col[4] = b; col[5] = c; }
    }
    const bool has_pos = col[0] != -1, has_nrm = col[3] != -1;
    // an element cannot hold more instances than the rest of the file has bytes for (the count is untrusted input); counts
    // beyond the C ABI's 32-bit point count are refused
    {
        unsigned long long need = 0;
        for (const Elem &e : h.elems) {
            if (e.count <= 0) continue;
            if (e.count > 0xffffffffll) return fail("invalid element count in PLY header: " + path);
            size_t min_row = 0;
            for (const Prop &p : e.props) min_row += h.mode == 0 ? 2 : (size_t)kTypeSize[p.is_list ? p.count_type : p.type];
            need += (unsigned long long)e.count * min_row;
        }
        if (need > (unsigned long long)(file_size - (long long)h.data_offset) + 1) return fail("error occurred while parsing ply file: " + path);
    }
    // fast path: the vertex element comes first, binary in the host's byte order, exactly float x y z nx ny nz, and nothing
    // behind it that could still fail the file
    bool plain6 = vertex == 0 && h.mode != 0 && !swap && h.elems[0].props.size() == 6;
    for (int w = 0; w < 6 && plain6; ++w) plain6 = col[w] == w && h.elems[0].props[w].type == 6;
    for (size_t i = 1; i < h.elems.size() && plain6; ++i) plain6 = h.elems[i].count <= 0 || h.elems[i].props.empty();
    if (plain6) {
        const size_t n = (size_t)h.elems[0].count;
        size_to(6 * n);
        if (whole) {
            if (data.size() - h.data_offset < 24 * n) return fail("error occurred while parsing ply file: " + path);
            memcpy(out.data(), data.data() + h.data_offset, 24 * n);
        } else {
            if (fseek(f, (long)h.data_offset, SEEK_SET) != 0 || fread(out.data(), 24, n, f) != n) return fail("error occurred while parsing ply file: " + path);
        }
    } else {
        if (!whole) {
            const size_t have = data.size();
            data.resize((size_t)file_size);
            if (fread(data.data() + have, 1, data.size() - have, f) != data.size() - have) return fail("error occurred while parsing ply file: " + path);
        }
        if (vertex >= 0 && has_pos && has_nrm) size_to(6 * (size_t)h.elems[vertex].count);
        Cursor c{data.data() + h.data_offset, data.data() + data.size()};
        std::string w;
        for (size_t ei = 0; ei < h.elems.size(); ++ei) {
            const Elem &e = h.elems[ei];
            const bool keep = (int)ei == vertex && has_pos && has_nrm;
            const size_t np = e.props.size();
            std::vector<int> slots(np, -1);
            // (two coordinates fed by one column cannot happen: the names differ)
            if (keep) {
                for (int q = 0; q < 6; ++q) if (col[q] >= 0) slots[col[q]] = q;
                for (int q = 0; q < 6; ++q) if (col[q] == kZeros) for (long long i = 0; i < e.count; ++i) out[6 * (size_t)i + q] = 0.f;
            }
            // fixed-size binary rows: bounds once per element, no per-value checks
            bool fixed = h.mode != 0;
            size_t row = 0;
            for (const Prop &p : e.props) { if (p.is_list) fixed = false; else row += kTypeSize[p.type]; }
            if (fixed && e.count > 0) {
                if ((unsigned long long)(c.end - c.p) < (unsigned long long)e.count * row) return fail("error occurred while parsing ply file: " + path);
                if (!keep) { c.p += (size_t)e.count * row; continue; }
            }
            for (long long i = 0; i < e.count; ++i)
                for (size_t k = 0; k < np; ++k) {
                    const Prop &p = e.props[k];
                    double v;
                    if (p.is_list) {
                        if (!read_value(c, p.count_type, h.mode, swap, w, v)) return fail("error occurred while parsing ply file: " + path);
                        const long len = (long)v;
                        for (long q = 0; q < len; ++q)
                            if (!read_value(c, p.item_type, h.mode, swap, w, v)) return fail("error occurred while parsing ply file: " + path);
                        continue;
                    }
                    if (!read_value(c, p.type, h.mode, swap, w, v)) return fail("error occurred while parsing ply file: " + path);
                    if (slots[k] >= 0) out[6 * (size_t)i + slots[k]] = (float)v;      // double -> float (ply_reader.cpp:197)
                }
        }
    }
    // util.cpp:1519-1545
    if (vertex < 0 || (!has_pos && !has_nrm)) return fail(vertex < 0 && h.elems.empty() ? "failed to read ply file (no elements): " + path
                                                                                          : "no points in the vertex element of " + path);
    if (has_pos != has_nrm) return fail("the number of points does not equal to the number of normals in the file");
    if (warnings) {
        const Elem &e = h.elems[vertex];
        auto has = [&](const char *a, const char *b, const char *c3, bool fl) {
            int found = 0;
            for (const char *nm : {a, b, c3})
                for (const Prop &p : e.props) if (!p.is_list && floating(p.type) == fl && p.name == nm) { ++found; break; }
            return found == 3;
        };
        if (has("r", "g", "b", true) || has("red", "green", "blue", false) || has("diffuse_red", "diffuse_green", "diffuse_blue", false))
            warnings->push_back("Warning: ignored property 'color'");                     // util.cpp:1530
        for (size_t i = 0; i < h.elems.size(); ++i)
            if (h.elems[i].name != "vertex" && h.elems[i].count > 0 && !h.elems[i].props.empty())
                warnings->push_back("Warning: unknown element '" + h.elems[i].name);       // util.cpp:1542 (the quote is not closed there either)
        if (!out.empty()) {                                                                // ply_reader.cpp:351-355
            const float len = std::sqrt(out[3] * out[3] + out[4] * out[4] + out[5] * out[5]);
            if (std::abs(1.0 - len) > 1e-4)
                warnings->push_back("normals (defined on element 'vertex') not normalized (length of the first normal vector is " + std::to_string(len) + ")");
        }
    }
    return !out.empty();
}

}  // namespace

bool read_ply_pos_nrm(const std::string &path, std::vector<float> &out, std::string &err, std::vector<std::string> *warnings,
                      const std::function<void()> *before_grow) {
    // never throws: a header that asks for more memory than there is ends in `false` like any other malformed file
    try {
        if (read_body(path, out, err, warnings, before_grow)) return true;
        if (err.empty()) err = "empty point cloud in " + path;
        out.clear();
        return false;
    } catch (const std::exception &e) {
        out.clear();
        err = std::string("cannot read PLY file ") + path + ": " + e.what();
        return false;
    }
}

bool write_ply_pos_nrm(const std::string &path, const float *pos_nrm, size_t n) {
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return false;
    fprintf(f, "ply\nformat %s 1.0\nelement vertex %zu\n", host_is_little_endian() ? "binary_little_endian" : "binary_big_endian", n);
    fprintf(f, "property float x\nproperty float y\nproperty float z\nproperty float nx\nproperty float ny\nproperty float nz\nend_header\n");
    const bool ok = fwrite(pos_nrm, 24, n, f) == n;
    fclose(f);
    return ok;
}

}  // namespace plade

// ---- C ABI (include/plade_hip.h) ---------------------------------------------------------------------------------------------
extern "C" int plade_ply_read(const char *path, float **pos_nrm, uint64_t *n, char *err, size_t err_cap) {
    if (err && err_cap) err[0] = 0;
    if (!path || !pos_nrm || !n) return PLADE_EINVAL;
    *pos_nrm = nullptr; *n = 0;
    std::vector<float> buf;
    std::string msg;
    if (!plade::read_ply_pos_nrm(path, buf, msg, nullptr, nullptr)) {
        if (err && err_cap) { strncpy(err, msg.c_str(), err_cap - 1); err[err_cap - 1] = 0; }
        return PLADE_EINVAL;
    }
    float *o = static_cast<float *>(malloc(buf.size() * sizeof(float)));
    if (!o) { if (err && err_cap) { strncpy(err, "out of memory", err_cap - 1); err[err_cap - 1] = 0; } return PLADE_EINVAL; }
    memcpy(o, buf.data(), buf.size() * sizeof(float));
    *pos_nrm = o; *n = buf.size() / 6;
    return PLADE_OK;
}
extern "C" void plade_ply_free(float *pos_nrm) { free(pos_nrm); }
