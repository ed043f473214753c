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
        if (count > t.nbytes / sizeof(S)) return false;   // (no product that could wrap)
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
    size_t total = 1;
    for (uint32_t i = 0; i < dims.nbytes / 4; i++) {
        const int32_t dv = (int32_t)c.u32(dims.data + 4 * i);
        // sizes come from the file: a negative one, or a product that wraps, must not size a buffer
        if (dv < 0 || (dv > 0 && total > (size_t)0x7fffffffffffull / (size_t)dv)) {
            err = "array dimensions out of range";
            return false;
        }
        total *= (size_t)dv;
        a.dims.push_back((size_t)dv);
    }
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

// ---------------------------------------------------------------------------
// streaming reader

struct Stream::Impl {
    FILE *f = nullptr;
    bool swap = false;
    // the element being walked: either `raw_left` bytes straight from the file, or a zlib stream
    // fed from `comp_left` bytes of the file
    bool compressed = false;
    size_t raw_left = 0, comp_left = 0;
    z_stream zs;
    bool zs_live = false;
    std::vector<uint8_t> in;      // compressed input window
    std::vector<uint8_t> stage;   // values in their storage type, before conversion
    uint32_t data_type = 0;       // mi* type of the data element
    size_t data_left = 0;         // bytes of it not handed out yet

    bool get(void *dst, size_t n)   // next n bytes of the current element
    {
        uint8_t *out = static_cast<uint8_t *>(dst);
        if (!compressed) {
            if (n > raw_left || std::fread(out, 1, n, f) != n) return false;
            raw_left -= n;
            return true;
        }
        zs.next_out = out;
        while (n) {
            if (zs.avail_in == 0) {
                if (comp_left == 0) return false;
                const size_t want = comp_left < in.size() ? comp_left : in.size();
                if (std::fread(in.data(), 1, want, f) != want) return false;
                comp_left -= want;
                zs.next_in = in.data();
                zs.avail_in = (uInt)want;
            }
            const uInt chunk = (uInt)(n > 0x40000000u ? 0x40000000u : n);
            zs.avail_out = chunk;
            const int rc = inflate(&zs, Z_NO_FLUSH);
            n -= chunk - zs.avail_out;
            if (rc == Z_STREAM_END && n) return false;
            if (rc != Z_OK && rc != Z_STREAM_END) return false;
        }
        return true;
    }
    bool skip(size_t n)
    {
        uint8_t tmp[4096];
        while (n) {
            const size_t k = n < sizeof(tmp) ? n : sizeof(tmp);
            if (!get(tmp, k)) return false;
            n -= k;
        }
        return true;
    }
    uint32_t u32(const uint8_t *p) const
    {
        uint32_t v;
        std::memcpy(&v, p, 4);
        return swap ? __builtin_bswap32(v) : v;
    }
    // one sub-element tag: type, byte count, bytes of padding after the data; small elements
    // carry their (at most 4) data bytes in `small`
    bool tag(uint32_t &type, uint32_t &nbytes, size_t &pad, uint8_t small[4], bool &is_small)
    {
        uint8_t t[8];
        if (!get(t, 8)) return false;
        const uint32_t w0 = u32(t);
        if (w0 >> 16) {
            type = w0 & 0xffff;
            nbytes = w0 >> 16;
            if (nbytes > 4) return false;
            std::memcpy(small, t + 4, 4);
            is_small = true;
            pad = 0;
        } else {
            type = w0;
            nbytes = u32(t + 4);
            is_small = false;
            pad = (8 - nbytes % 8) % 8;
        }
        return true;
    }
    void end_element()
    {
        if (zs_live) {
            inflateEnd(&zs);
            zs_live = false;
        }
    }
};

Stream::Stream() : im_(new Impl) {}
Stream::~Stream()
{
    close();
    delete im_;
}

void Stream::close()
{
    im_->end_element();
    if (im_->f) std::fclose(im_->f);
    im_->f = nullptr;
}

bool Stream::open(const std::string &path, const std::string &name, std::string &err)
{
    close();
    Impl &m = *im_;
    m.f = std::fopen(path.c_str(), "rb");
    if (!m.f) { err = "cannot open '" + path + "'"; return false; }
    uint8_t hdr[128];
    if (std::fread(hdr, 1, 128, m.f) != 128) { err = "'" + path + "' is not a MAT-file (too short)"; return false; }
    if (std::memcmp(hdr, "MATLAB 7.3 MAT-file", 19) == 0) {
        err = "'" + path + "' is a v7.3 (HDF5) MAT-file; save it with -v7 or -v6";
        return false;
    }
    if (std::memcmp(hdr, "MATLAB 5.0 MAT-file", 19) != 0) { err = "'" + path + "' has no MAT-file header"; return false; }
    if (hdr[126] == 'I' && hdr[127] == 'M') m.swap = false;
    else if (hdr[126] == 'M' && hdr[127] == 'I') m.swap = true;
    else { err = "bad endian indicator"; return false; }
    m.in.resize(1 << 20);
    for (;;) {
        uint8_t t[8];
        const long at = std::ftell(m.f);
        if (std::fread(t, 1, 8, m.f) != 8) break;   // end of file: not found
        const uint32_t type = m.u32(t), nbytes = m.u32(t + 4);
        const long next = at + 8 + (long)nbytes + (type == miCOMPRESSED ? 0 : (long)((8 - nbytes % 8) % 8));
        bool is_matrix = false;
        if (type == miCOMPRESSED) {
            m.compressed = true;
            m.comp_left = nbytes;
            std::memset(&m.zs, 0, sizeof(m.zs));
            if (inflateInit(&m.zs) != Z_OK) { err = "zlib init failed"; return false; }
            m.zs_live = true;
            uint8_t inner[8];
            if (m.get(inner, 8) && m.u32(inner) == miMATRIX) is_matrix = true;
        } else if (type == miMATRIX) {
            m.compressed = false;
            m.raw_left = nbytes;
            is_matrix = true;
        }
        if (is_matrix) {
            uint32_t ty, nb;
            size_t pad;
            uint8_t small[4];
            bool sm;
            // array flags, dimensions, name
            uint8_t flags[8];
            bool ok = m.tag(ty, nb, pad, small, sm) && ty == miUINT32 && nb == 8 && !sm && m.get(flags, 8);
            uint32_t cls = 0;
            bool complex_ = false;
            if (ok) {
                const uint32_t f0 = m.u32(flags);
                cls = f0 & 0xff;
                complex_ = (f0 >> 11) & 1;
            }
            std::vector<size_t> dims;
            size_t total = 1;
            if (ok) ok = m.tag(ty, nb, pad, small, sm) && ty == miINT32 && nb % 4 == 0 && nb <= 64;
            if (ok) {
                uint8_t db[64 + 8];
                if (sm) std::memcpy(db, small, 4);
                else ok = m.get(db, nb + pad);
                for (uint32_t i = 0; ok && i < nb / 4; i++) {
                    const int32_t dv = (int32_t)m.u32(db + 4 * i);
                    if (dv < 0 || (dv > 0 && total > (size_t)0x7fffffffffffull / (size_t)dv)) {
                        err = "array dimensions out of range";
                        return false;
                    }
                    total *= (size_t)dv;
                    dims.push_back((size_t)dv);
                }
            }
            std::string nm;
            if (ok) ok = m.tag(ty, nb, pad, small, sm) && (ty == miINT8 || ty == miUINT8 || ty == miUTF8) && nb <= 4096;
            if (ok) {
                if (sm) {
                    nm.assign(reinterpret_cast<const char *>(small), nb);
                } else {
                    std::vector<uint8_t> nbuf(nb + pad);
                    ok = m.get(nbuf.data(), nb + pad);
                    if (ok) nm.assign(reinterpret_cast<const char *>(nbuf.data()), nb);
                }
            }
            if (ok && nm == name) {
                if (cls < mxDOUBLE || cls > mxUINT64) { err = "variable '" + name + "' is not a numeric array"; return false; }
                if (complex_) { err = "variable '" + name + "' is complex"; return false; }
                if (!m.tag(ty, nb, pad, small, sm)) { err = "variable '" + name + "' has no data element"; return false; }
                size_t esz = 0;
                switch (ty) {
                    case miDOUBLE: case miINT64: case miUINT64: esz = 8; break;
                    case miSINGLE: case miINT32: case miUINT32: esz = 4; break;
                    case miINT16: case miUINT16: esz = 2; break;
                    case miINT8: case miUINT8: esz = 1; break;
                    default: err = "variable '" + name + "': unsupported storage type"; return false;
                }
                dims_ = dims;
                numel_ = dims.empty() ? 0 : total;
                done_ = 0;
                is_single_ = cls == mxSINGLE;
                if (numel_ > nb / esz) { err = "variable '" + name + "': data element shorter than its dimensions"; return false; }
                if (sm) {   // at most 4 bytes inline: serve them from the staging buffer
                    err = "variable '" + name + "' is stored as a small element; use matio::read";
                    return false;
                }
                m.data_type = ty;
                m.data_left = (size_t)nb;
                return true;
            }
        }
        m.end_element();
        if (std::fseek(m.f, next, SEEK_SET) != 0) break;
    }
    err = "'" + path + "' has no variable '" + name + "'";
    return false;
}

template <typename D>
bool Stream::next_impl(D *dst, size_t count, std::string &err)
{
    Impl &m = *im_;
    if (count > remaining()) { err = "read past the end of the variable"; return false; }
    auto run = [&](auto sample) {
        typedef decltype(sample) S;
        constexpr size_t CH = (4u << 20) / sizeof(S);
        size_t left = count;
        while (left) {
            const size_t k = left < CH ? left : CH;
            if (!m.swap && std::is_same<S, D>::value) {
                if (!m.get(dst, k * sizeof(S))) return false;
            } else {
                m.stage.resize(k * sizeof(S));
                if (!m.get(m.stage.data(), k * sizeof(S))) return false;
                for (size_t i = 0; i < k; i++) dst[i] = (D)load_as<S>(m.stage.data() + i * sizeof(S), m.swap);
            }
            dst += k;
            left -= k;
        }
        return true;
    };
    bool ok = false;
    switch (m.data_type) {
        case miDOUBLE: ok = run(double()); break;
        case miSINGLE: ok = run(float()); break;
        case miINT8: ok = run(int8_t()); break;
        case miUINT8: ok = run(uint8_t()); break;
        case miINT16: ok = run(int16_t()); break;
        case miUINT16: ok = run(uint16_t()); break;
        case miINT32: ok = run(int32_t()); break;
        case miUINT32: ok = run(uint32_t()); break;
        case miINT64: ok = run(int64_t()); break;
        case miUINT64: ok = run(uint64_t()); break;
        default: break;
    }
    if (!ok) { err = "truncated or corrupt data element"; return false; }
    done_ += count;
    return true;
}

bool Stream::next(float *dst, size_t count, std::string &err) { return next_impl(dst, count, err); }
bool Stream::next(double *dst, size_t count, std::string &err) { return next_impl(dst, count, err); }

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
