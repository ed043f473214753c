#include "matio.h"

#include <zlib.h>

#include <cstdio>
#include <cstring>
#include <ctime>

namespace matio {
namespace {

enum : uint32_t {
    miINT8 = 1, miUINT8 = 2, miINT16 = 3, miUINT16 = 4, miINT32 = 5, miUINT32 = 6, miSINGLE = 7,
    miDOUBLE = 9, miINT64 = 12, miUINT64 = 13, miMATRIX = 14, miCOMPRESSED = 15, miUTF8 = 16
};
enum : uint32_t { mxDOUBLE = 6, mxSINGLE = 7, mxINT8 = 8, mxUINT64 = 15 };

struct Cursor {
    const uint8_t *p;
    size_t n, off = 0;
    bool swap;
    bool need(size_t k) const { return off + k <= n; }
    uint32_t u32(size_t at) const
    {
        uint32_t v;
        std::memcpy(&v, p + at, 4);
        return swap ? __builtin_bswap32(v) : v;
    }
};

struct Tag {
    uint32_t type = 0, nbytes = 0;
    size_t data = 0, next = 0;     // offsets into the cursor's buffer
};

// Parses the tag at c.off (normal or small-element form); does not advance.
bool read_tag(const Cursor &c, Tag &t)
{
    if (!c.need(8)) return false;
    const uint32_t w0 = c.u32(c.off);
    if (w0 >> 16) {                // small element: 2-byte size, 2-byte type, 4 data bytes
        t.type = w0 & 0xffff;
        t.nbytes = w0 >> 16;
        if (t.nbytes > 4) return false;
        t.data = c.off + 4;
        t.next = c.off + 8;
    } else {
        t.type = w0;
        t.nbytes = c.u32(c.off + 4);
        t.data = c.off + 8;
        // compressed elements are not padded; everything else is padded to 8 bytes
        const size_t pad = t.type == miCOMPRESSED ? 0 : (8 - t.nbytes % 8) % 8;
        t.next = t.data + t.nbytes + pad;
        if (t.data + t.nbytes > c.n) return false;
        if (t.next > c.n) t.next = c.n;
    }
    return true;
}

template <typename S>
S load_as(const uint8_t *p, bool swap)
{
    S v;
    std::memcpy(&v, p, sizeof(S));
    if (swap) {
        uint8_t *b = reinterpret_cast<uint8_t *>(&v);
        for (size_t i = 0; i < sizeof(S) / 2; i++) std::swap(b[i], b[sizeof(S) - 1 - i]);
    }
    return v;
}

template <typename D>
bool convert(const Cursor &c, const Tag &t, size_t count, std::vector<D> &dst)
{
    auto run = [&](auto sample) {
        typedef decltype(sample) S;
        if (t.nbytes < count * sizeof(S)) return false;
        dst.resize(count);
        const uint8_t *p = c.p + t.data;
        if (!c.swap && std::is_same<S, D>::value) {
            std::memcpy(dst.data(), p, count * sizeof(S));
        } else {
            for (size_t i = 0; i < count; i++) dst[i] = (D)load_as<S>(p + i * sizeof(S), c.swap);
        }
        return true;
    };
    switch (t.type) {
        case miDOUBLE: return run(double());
        case miSINGLE: return run(float());
        case miINT8: return run(int8_t());
        case miUINT8: return run(uint8_t());
        case miINT16: return run(int16_t());
        case miUINT16: return run(uint16_t());
        case miINT32: return run(int32_t());
        case miUINT32: return run(uint32_t());
        case miINT64: return run(int64_t());
        case miUINT64: return run(uint64_t());
        default: return false;
    }
}

// Parses one miMATRIX payload [begin, end) of c's buffer.
bool parse_matrix(Cursor c, size_t begin, size_t end, const std::vector<std::string> &names, std::vector<Array> &out,
                  std::string &err)
{
    c.off = begin;
    c.n = end;
    Tag flags, dims, name, real;
    if (!read_tag(c, flags) || flags.type != miUINT32 || flags.nbytes < 8) return true;   // not a plain array: skip
    const uint32_t f0 = c.u32(flags.data);
    const uint32_t cls = f0 & 0xff;
    const bool complex_ = (f0 >> 11) & 1;
    c.off = flags.next;
    if (!read_tag(c, dims) || dims.type != miINT32) { err = "malformed dimensions element"; return false; }
    Array a;
    for (uint32_t i = 0; i < dims.nbytes / 4; i++) a.dims.push_back((size_t)(int32_t)c.u32(dims.data + 4 * i));
    c.off = dims.next;
    if (!read_tag(c, name) || (name.type != miINT8 && name.type != miUINT8 && name.type != miUTF8)) {
        err = "malformed array name";
        return false;
    }
    a.name.assign(reinterpret_cast<const char *>(c.p + name.data), name.nbytes);
    c.off = name.next;
    if (!names.empty()) {
        bool wanted = false;
        for (auto &w : names) wanted |= (w == a.name);
        if (!wanted) return true;
    }
    if (cls < mxDOUBLE || cls > mxUINT64) return true;   // cell/struct/char/sparse: not ours
    if (complex_) { err = "variable '" + a.name + "' is complex"; return false; }
    if (!read_tag(c, real)) { err = "variable '" + a.name + "' has no data element"; return false; }
    a.is_single = cls == mxSINGLE;
    const size_t count = a.numel();
    const bool ok = a.is_single ? convert(c, real, count, a.f32) : convert(c, real, count, a.f64);
    if (!ok) { err = "variable '" + a.name + "': data element shorter than its dimensions"; return false; }
    out.push_back(std::move(a));
    return true;
}

bool inflate_all(const uint8_t *src, size_t n, std::vector<uint8_t> &dst, std::string &err)
{
    z_stream zs;
    std::memset(&zs, 0, sizeof(zs));
    if (inflateInit(&zs) != Z_OK) { err = "zlib init failed"; return false; }
    dst.resize(n * 4 + 1024);
    zs.next_in = const_cast<Bytef *>(src);
    zs.avail_in = (uInt)n;
    size_t have = 0;
    int rc;
    do {
        if (have == dst.size()) dst.resize(dst.size() * 2);
        zs.next_out = dst.data() + have;
        const size_t room = dst.size() - have;
        zs.avail_out = (uInt)(room > 0x40000000u ? 0x40000000u : room);
        const size_t before = zs.avail_out;
        rc = inflate(&zs, Z_NO_FLUSH);
        have += before - zs.avail_out;
    } while (rc == Z_OK);
    inflateEnd(&zs);
    if (rc != Z_STREAM_END) { err = "corrupt compressed element"; return false; }
    dst.resize(have);
    return true;
}

}  // namespace

bool read(const std::string &path, const std::vector<std::string> &names, std::vector<Array> &out, std::string &err)
{
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) { err = "cannot open '" + path + "'"; return false; }
    std::fseek(f, 0, SEEK_END);
    const long sz = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> buf((size_t)(sz > 0 ? sz : 0));
    const size_t got = buf.empty() ? 0 : std::fread(buf.data(), 1, buf.size(), f);
    std::fclose(f);
    if (got != buf.size() || buf.size() < 128) { err = "'" + path + "' is not a MAT-file (too short)"; return false; }
    if (std::memcmp(buf.data(), "MATLAB 7.3 MAT-file", 19) == 0) {
        err = "'" + path + "' is a v7.3 (HDF5) MAT-file; save it with -v7 or -v6";
        return false;
    }
    if (std::memcmp(buf.data(), "MATLAB 5.0 MAT-file", 19) != 0) { err = "'" + path + "' has no MAT-file header"; return false; }
    Cursor c{buf.data(), buf.size()};
    if (buf[126] == 'I' && buf[127] == 'M') c.swap = false;
    else if (buf[126] == 'M' && buf[127] == 'I') c.swap = true;
    else { err = "bad endian indicator"; return false; }
    c.off = 128;
    while (c.off + 8 <= c.n) {
        Tag t;
        if (!read_tag(c, t)) { err = "truncated data element"; return false; }
        if (t.type == miCOMPRESSED) {
            std::vector<uint8_t> raw;
            if (!inflate_all(c.p + t.data, t.nbytes, raw, err)) return false;
            Cursor ci{raw.data(), raw.size()};
            ci.swap = c.swap;
            Tag ti;
            if (!read_tag(ci, ti)) { err = "truncated compressed element"; return false; }
            if (ti.type == miMATRIX && !parse_matrix(ci, ti.data, ti.data + ti.nbytes, names, out, err)) return false;
        } else if (t.type == miMATRIX) {
            if (!parse_matrix(c, t.data, t.data + t.nbytes, names, out, err)) return false;
        }
        c.off = t.next;
    }
    return true;
}

bool write(const std::string &path, const std::vector<DoubleMatrix> &vars, std::string &err)
{
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) { err = "cannot create '" + path + "'"; return false; }
    char hdr[128];
    std::memset(hdr, ' ', sizeof(hdr));
    const std::time_t now = std::time(nullptr);
    char stamp[64];
    std::strftime(stamp, sizeof(stamp), "%a %b %d %H:%M:%S %Y", std::localtime(&now));
    const int k = std::snprintf(hdr, 116, "MATLAB 5.0 MAT-file, Platform: GLNXA64, Created on: %s", stamp);
    hdr[k] = ' ';
    std::memset(hdr + 116, 0, 8);
    hdr[124] = 0x00; hdr[125] = 0x01;        // version 0x0100, little endian
    hdr[126] = 'I'; hdr[127] = 'M';
    bool ok = std::fwrite(hdr, 1, 128, f) == 128;
    auto put32 = [&](uint32_t v) { ok &= std::fwrite(&v, 4, 1, f) == 1; };
    auto pad8 = [&](size_t n) {
        static const char z[8] = {0};
        if (n % 8) ok &= std::fwrite(z, 1, 8 - n % 8, f) == 8 - n % 8;
    };
    for (auto &v : vars) {
        const size_t nl = v.name.size();
        const size_t data_bytes = v.rows * v.cols * 8;
        const size_t name_el = nl <= 4 ? 8 : 8 + (nl + 7) / 8 * 8;
        const size_t total = 16 + 16 + name_el + 8 + (data_bytes + 7) / 8 * 8;
        put32(miMATRIX); put32((uint32_t)total);
        put32(miUINT32); put32(8); put32(mxDOUBLE); put32(0);              // array flags
        put32(miINT32); put32(8); put32((uint32_t)v.rows); put32((uint32_t)v.cols);
        if (nl <= 4) {                                                       // small element
            put32((uint32_t)(nl << 16) | miINT8);
            char nm[4] = {0, 0, 0, 0};
            std::memcpy(nm, v.name.data(), nl);
            ok &= std::fwrite(nm, 1, 4, f) == 4;
        } else {
            put32(miINT8); put32((uint32_t)nl);
            ok &= std::fwrite(v.name.data(), 1, nl, f) == nl;
            pad8(nl);
        }
        put32(miDOUBLE); put32((uint32_t)data_bytes);
        ok &= data_bytes == 0 || std::fwrite(v.data, 1, data_bytes, f) == data_bytes;
        pad8(data_bytes);
    }
    ok &= std::fclose(f) == 0;
    if (!ok) err = "short write to '" + path + "'";
    return ok;
}

}  // namespace matio
