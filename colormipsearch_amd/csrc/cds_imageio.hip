// cds_imageio.hip -- host side only: MIP file decoding and threaded library ingest.
//
// Replaces, for the files the reference's pipeline reads (color-depth MIPs: 8-bit RGB TIFF or PNG; gradient images: 8/16-bit
// gray PNG or TIFF), ImageArrayUtils.readImageArray (API/imageprocessing/ImageArrayUtils.java:98-121, which hands TIFFs to
// ImageJ's Opener / LocalTiffDecoder.java and everything else to ImageIO) and its PackBits strip reader
// (packBitsUncompress :229-258).  Supported: baseline TIFF (II / MM, strips, chunky, 8- or 16-bit samples, 1 / 3 / 4 samples per
// pixel, compression none / PackBits 32773 / LZW 5 / Deflate 8, 32946, predictor 2) and non-interlaced PNG (gray 8/16, RGB 8,
// RGBA 8, palette 8).  Everything else is refused with a message (no silent fallback).
// cds_sparse_library_append_files decodes on `threads` host threads, two batches in flight: one being decoded while the
// previous one is uploaded and turned into signal-pixel records on the GPU.
#include <dlfcn.h>
#include <fcntl.h>
#include <strings.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

#include "cds_internal.hpp"
#include "cds_imageio.hpp"

namespace cds {

constexpr uint32_t kMaxImageSide = 16384;  // (a header is untrusted input: bounds the buffers a file can make us allocate)

// Pinned host buffers of the file loaders, kept between calls: pinning memory costs the driver ~0.15 ms per MB (a call of
// cds_library_load_files over 64-slot batches pins ~700 MB: 100 ms, as long as decoding its files), so a released buffer waits here for
// the next call that fits it (up to 2 GB idle; process lifetime).
struct PinnedPool {
    std::mutex mu;
    std::vector<std::pair<uint8_t *, size_t>> idle;
    size_t idle_bytes = 0;
    uint8_t *get(size_t bytes) {
        {
            std::lock_guard<std::mutex> lk(mu);
            size_t best = SIZE_MAX, at = 0;
            for (size_t i = 0; i < idle.size(); i++)
                if (idle[i].second >= bytes && idle[i].second < best) best = idle[i].second, at = i;
            if (best != SIZE_MAX && best <= 2 * bytes + ((size_t)1 << 20)) {
                uint8_t *p = idle[at].first;
                sizes_[p] = best;
                idle.erase(idle.begin() + (std::ptrdiff_t)at);
                idle_bytes -= best;
                return p;
            }
        }
        uint8_t *p = nullptr;
        if (hipHostMalloc((void **)&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
        std::lock_guard<std::mutex> lk(mu);
        sizes_[p] = bytes;
        return p;
    }
    void put(uint8_t *p) {
        if (!p) return;
        {
            std::lock_guard<std::mutex> lk(mu);
            auto it = sizes_.find(p);
            const size_t b = it == sizes_.end() ? 0 : it->second;
            if (it != sizes_.end()) sizes_.erase(it);
            if (b && idle_bytes + b <= ((size_t)2 << 30)) {
                idle.push_back({p, b});
                idle_bytes += b;
                return;
            }
        }
        hipHostFree(p);
    }
    std::unordered_map<uint8_t *, size_t> sizes_;
};
static PinnedPool g_pinned;
uint8_t *pinned_get(size_t bytes) { return g_pinned.get(bytes); }
void pinned_put(uint8_t *p) { g_pinned.put(p); }

// One-shot zlib-stream decompression into a buffer of known size (a PNG's filtered scanlines, a Deflate TIFF strip).  libdeflate -- present
// in the image as libdeflate.so.0, no headers: bound by dlopen, three functions of its stable C API -- inflates 2-3x as fast as zlib, and a
// 16-bit gradient PNG is 1.4 MB behind zlib: the whole cost of loading kernel 2's inputs from files.  zlib's uncompress stays
// the fallback (no libdeflate, CDS_IMAGEIO_ZLIB=1, or libdeflate refusing a stream that zlib may still read).
struct Deflater {
    void *(*alloc)() = nullptr;
    int (*zlib_decompress)(void *, const void *, size_t, void *, size_t, size_t *) = nullptr;
    int (*deflate_decompress)(void *, const void *, size_t, void *, size_t, size_t *) = nullptr;   // a raw deflate stream (zip entries)
    void (*release)(void *) = nullptr;
    Deflater() {
        if (getenv("CDS_IMAGEIO_ZLIB")) return;
        void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("libdeflate.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        alloc = (decltype(alloc))dlsym(h, "libdeflate_alloc_decompressor");
        zlib_decompress = (decltype(zlib_decompress))dlsym(h, "libdeflate_zlib_decompress");
        deflate_decompress = (decltype(deflate_decompress))dlsym(h, "libdeflate_deflate_decompress");
        release = (decltype(release))dlsym(h, "libdeflate_free_decompressor");
        if (!alloc || !zlib_decompress || !release) alloc = nullptr;
    }
};
static bool zlib_inflate_exact(const uint8_t *src, size_t len, uint8_t *dst, size_t want, bool raw = false) {
    static const Deflater lib;
    if (lib.alloc && (!raw || lib.deflate_decompress)) {
        struct Holder {   // one decompressor per thread, freed with the thread
            void *d = nullptr;
            void (*release)(void *) = nullptr;
            ~Holder() { if (d && release) release(d); }
        };
        thread_local Holder hd;
        if (!hd.d) hd.d = lib.alloc(), hd.release = lib.release;
        if (hd.d) {
            size_t got = 0;
            if ((raw ? lib.deflate_decompress : lib.zlib_decompress)(hd.d, src, len, dst, want, &got) == 0 && got == want) return true;   // (LIBDEFLATE_SUCCESS == 0)
        }
    }
    if (raw) {   // zlib, windowBits < 0: no zlib header, no checksum
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, -MAX_WBITS) != Z_OK) return false;
        zs.next_in = const_cast<Bytef *>(src), zs.avail_in = (uInt)len;
        zs.next_out = dst, zs.avail_out = (uInt)want;
        const int rc = inflate(&zs, Z_FINISH);
        const bool ok = rc == Z_STREAM_END && zs.total_out == want;
        inflateEnd(&zs);
        return ok;
    }
    uLongf got = (uLongf)want;
    return uncompress(dst, &got, src, (uLong)len) == Z_OK && got == want;
}

struct ImgInfo {
    int W = 0, H = 0, channels = 0, bits = 0;  // channels of the DECODED image (alpha dropped, palette expanded)
};

struct Reader {
    const uint8_t *p;
    size_t n;
    bool le;
    bool ok(size_t off, size_t len) const { return off <= n && len <= n - off; }
    uint16_t u16(size_t off) const { return le ? (uint16_t)(p[off] | p[off + 1] << 8) : (uint16_t)(p[off] << 8 | p[off + 1]); }
    uint32_t u32(size_t off) const {
        return le ? (uint32_t)p[off] | (uint32_t)p[off + 1] << 8 | (uint32_t)p[off + 2] << 16 | (uint32_t)p[off + 3] << 24
                  : (uint32_t)p[off] << 24 | (uint32_t)p[off + 1] << 16 | (uint32_t)p[off + 2] << 8 | (uint32_t)p[off + 3];
    }
};

// ------------------------------------------------------------------------------------------------ TIFF
struct TiffDir {
    uint32_t W = 0, H = 0, bits = 8, compression = 1, photometric = 2, spp = 1, rows_per_strip = 0xffffffffu, planar = 1, predictor = 1;
    std::vector<uint32_t> offsets, counts;
    bool tiled = false;
};

static bool tiff_values(const Reader &r, size_t entry, std::vector<uint32_t> &out, std::string &err) {
    const uint16_t type = r.u16(entry + 2);
    const uint32_t count = r.u32(entry + 4);
    const size_t size = type == 1 || type == 2 || type == 6 || type == 7 ? 1 : type == 3 || type == 8 ? 2 : type == 4 || type == 9 ? 4 : 0;
    if (!size) {
        err = "unsupported TIFF field type";
        return false;
    }
    if (count > (1u << 24)) {
        err = "implausible TIFF field count";
        return false;
    }
    size_t off = entry + 8;
    if ((size_t)count * size > 4) {
        off = r.u32(entry + 8);
        if (!r.ok(off, (size_t)count * size)) {
            err = "TIFF field data outside the file";
            return false;
        }
    }
    out.resize(count);
    for (uint32_t i = 0; i < count; i++) out[i] = size == 1 ? r.p[off + i] : size == 2 ? r.u16(off + 2 * (size_t)i) : r.u32(off + 4 * (size_t)i);
    return true;
}

static bool tiff_parse(const uint8_t *data, size_t n, Reader &r, TiffDir &d, std::string &err) {
    if (n < 8) {
        err = "file too short";
        return false;
    }
    r.p = data; r.n = n;
    if (data[0] == 'I' && data[1] == 'I') r.le = true;
    else if (data[0] == 'M' && data[1] == 'M') r.le = false;
    else {
        err = "not a TIFF";
        return false;
    }
    if (r.u16(2) != 42) {
        err = r.u16(2) == 43 ? "BigTIFF is not supported" : "not a TIFF";
        return false;
    }
    const size_t ifd = r.u32(4);
    if (!r.ok(ifd, 2)) {
        err = "TIFF directory outside the file";
        return false;
    }
    const int ne = r.u16(ifd);
    if (!r.ok(ifd + 2, (size_t)ne * 12)) {
        err = "TIFF directory outside the file";
        return false;
    }
    std::vector<uint32_t> v;
    for (int i = 0; i < ne; i++) {
        const size_t e = ifd + 2 + (size_t)i * 12;
        const int tag = r.u16(e);
        switch (tag) {
            case 256: case 257: case 258: case 259: case 262: case 273: case 277: case 278: case 279: case 284: case 317:
                if (!tiff_values(r, e, v, err)) return false;
                if (v.empty()) continue;
                break;
            case 322: case 323: case 324: case 325: d.tiled = true; continue;
            default: continue;
        }
        switch (tag) {
            case 256: d.W = v[0]; break;
            case 257: d.H = v[0]; break;
            case 258:
                d.bits = v[0];
                for (uint32_t b : v)
                    if (b != v[0]) {
                        err = "TIFF samples of different widths";
                        return false;
                    }
                break;
            case 259: d.compression = v[0]; break;
            case 262: d.photometric = v[0]; break;
            case 273: d.offsets = v; break;
            case 277: d.spp = v[0]; break;
            case 278: d.rows_per_strip = v[0]; break;
            case 279: d.counts = v; break;
            case 284: d.planar = v[0]; break;
            case 317: d.predictor = v[0]; break;
        }
    }
    if (d.tiled) { err = "tiled TIFFs are not supported"; return false; }
    if (d.W == 0 || d.H == 0 || d.W > kMaxImageSide || d.H > kMaxImageSide) { err = "TIFF without a valid size (1..16384 per side)"; return false; }
    if (d.bits != 8 && d.bits != 16) { err = "TIFF bits per sample must be 8 or 16"; return false; }
    if (d.spp != 1 && d.spp != 3 && d.spp != 4) { err = "TIFF samples per pixel must be 1, 3 or 4"; return false; }
    if (d.spp >= 3 && d.bits != 8) { err = "16-bit colour TIFFs are not supported"; return false; }
    if (d.planar != 1 && d.spp > 1) { err = "planar TIFFs are not supported"; return false; }
    if (d.photometric == 3) { err = "palette TIFFs are not supported"; return false; }
    if (d.photometric > 2) { err = "unsupported TIFF photometric interpretation"; return false; }
    if (d.compression != 1 && d.compression != 5 && d.compression != 8 && d.compression != 32946 && d.compression != 32773) {
        err = "unsupported TIFF compression " + std::to_string(d.compression);
        return false;
    }
    if (d.predictor != 1 && d.predictor != 2) { err = "unsupported TIFF predictor"; return false; }
    if (d.offsets.empty() || d.offsets.size() != d.counts.size()) {
        // ImageJ writes uncompressed single-strip files without StripByteCounts
        if (d.offsets.size() == 1 && d.counts.empty() && d.compression == 1) d.counts.assign(1, (uint32_t)std::min<size_t>(n - d.offsets[0], 0xffffffffu));
        else { err = "TIFF strip tables missing or inconsistent"; return false; }
    }
    if (d.rows_per_strip == 0 || d.rows_per_strip > d.H) d.rows_per_strip = d.H;
    if ((size_t)(d.H + d.rows_per_strip - 1) / d.rows_per_strip != d.offsets.size()) { err = "TIFF strip count does not match the image height"; return false; }
    return true;
}

static bool packbits_decode(const uint8_t *in, size_t n, uint8_t *out, size_t want) {
    size_t i = 0, o = 0;
    while (o < want && i < n) {
        const int8_t c = (int8_t)in[i++];
        if (c >= 0) {
            size_t len = (size_t)c + 1;
            if (i + len > n) len = n - i;
            if (o + len > want) len = want - o;
            memcpy(out + o, in + i, len);
            i += (size_t)c + 1;
            o += len;
        } else if (c != -128) {
            size_t len = (size_t)(1 - c);
            if (i >= n) break;
            if (o + len > want) len = want - o;
            memset(out + o, in[i++], len);
            o += len;
        }
    }
    return o == want;
}

// TIFF LZW (MSB-first codes, 9..12 bits, ClearCode 256, EoiCode 257, "early change" like every current writer: the code
// width grows as soon as the entry just added is number 2^bits - 2)
static bool lzw_decode(const uint8_t *in, size_t n, uint8_t *out, size_t want) {
    // Every table string is a piece of the output already written: the entry added after emitting `code` is the previous
    // code's string plus the first byte of this one, i.e. the bytes starting where the previous string was emitted, one
    // longer.  So an entry is (position, length) and emitting a code is a forward copy inside the output (overlapping for
    // the KwKwK case, hence byte order matters) -- no prefix chains to walk backwards.
    uint32_t pos[4096];
    uint32_t len[4096];
    int next = 258, bits = 9, old = -1;
    uint32_t acc = 0;
    int nacc = 0;
    size_t i = 0, o = 0, o_old = 0;
    size_t old_len = 0;   // full (unclipped) length of the previous code's string
    while (o < want) {
        while (nacc < bits && i < n) { acc = (acc << 8) | in[i++]; nacc += 8; }
        if (nacc < bits) break;
        const int code = (int)((acc >> (nacc - bits)) & ((1u << bits) - 1));
        nacc -= bits;
        if (code == 257) break;
        if (code == 256) { next = 258; bits = 9; old = -1; continue; }
        size_t L;
        const size_t at = o;
        if (code < 256) {
            if (old < 0 || code < next) {
                out[o++] = (uint8_t)code;
                L = 1;
            } else {
                return false;
            }
        } else {
            if (old < 0) return false;
            size_t src;
            if (code < next) {
                src = pos[code], L = len[code];
            } else if (code == next && next < 4096) {
                src = o_old, L = old_len + 1;   // KwKwK: the entry being defined right now
            } else {
                return false;
            }
            const size_t m = std::min(L, want - o);
            if (src + m <= o) memcpy(out + o, out + src, m);
            else for (size_t k = 0; k < m; k++) out[o + k] = out[src + k];
            o += m;
        }
        if (old >= 0 && next < 4096) {
            pos[next] = (uint32_t)o_old;
            len[next] = (uint32_t)(old_len + 1);
            next++;
            if (next >= (1 << bits) - 1 && bits < 12) bits++;
        }
        old = code;
        o_old = at;
        old_len = L;
    }
    return o == want;
}

static bool tiff_decode(const uint8_t *data, size_t n, const ImgInfo &info, void *outv, std::string &err) {
    Reader r{};
    TiffDir d;
    if (!tiff_parse(data, n, r, d, err)) return false;
    const size_t bps = d.bits / 8, src_px = (size_t)d.spp * bps, row_bytes = (size_t)d.W * src_px;
    std::vector<uint8_t> strip;
    uint8_t *out = static_cast<uint8_t *>(outv);
    const size_t dst_px = (size_t)info.channels * bps;
    for (size_t s = 0; s < d.offsets.size(); s++) {
        const size_t y0 = s * d.rows_per_strip, rows = std::min<size_t>(d.rows_per_strip, d.H - y0), want = rows * row_bytes;
        if (!r.ok(d.offsets[s], d.counts[s])) { err = "TIFF strip outside the file"; return false; }
        const uint8_t *src = data + d.offsets[s];
        const size_t len = d.counts[s];
        strip.resize(want);
        bool ok = true;
        switch (d.compression) {
            case 1: ok = len >= want; if (ok) memcpy(strip.data(), src, want); break;
            case 32773: ok = packbits_decode(src, len, strip.data(), want); break;
            case 5: ok = lzw_decode(src, len, strip.data(), want); break;
            default: ok = zlib_inflate_exact(src, len, strip.data(), want);
        }
        if (!ok) { err = "TIFF strip " + std::to_string(s) + " is truncated or corrupt"; return false; }
        for (size_t yy = 0; yy < rows; yy++) {
            uint8_t *row = strip.data() + yy * row_bytes;
            if (d.predictor == 2) {
                if (bps == 1) {
                    for (size_t k = d.spp; k < row_bytes; k++) row[k] = (uint8_t)(row[k] + row[k - d.spp]);
                } else {
                    // 16-bit gray: samples in file byte order
                    uint16_t prev = r.le ? (uint16_t)(row[0] | row[1] << 8) : (uint16_t)(row[0] << 8 | row[1]);
                    for (size_t x = 1; x < d.W; x++) {
                        uint8_t *q = row + 2 * x;
                        uint16_t v = r.le ? (uint16_t)(q[0] | q[1] << 8) : (uint16_t)(q[0] << 8 | q[1]);
                        v = (uint16_t)(v + prev);
                        prev = v;
                        if (r.le) { q[0] = (uint8_t)v; q[1] = (uint8_t)(v >> 8); } else { q[0] = (uint8_t)(v >> 8); q[1] = (uint8_t)v; }
                    }
                }
            }
            uint8_t *dst = out + (y0 + yy) * (size_t)d.W * dst_px;
            if (bps == 2) {  // gray 16: to host order, WhiteIsZero inverted
                uint16_t *o16 = reinterpret_cast<uint16_t *>(dst);
                for (size_t x = 0; x < d.W; x++) {
                    const uint8_t *q = row + 2 * x;
                    uint16_t v = r.le ? (uint16_t)(q[0] | q[1] << 8) : (uint16_t)(q[0] << 8 | q[1]);
                    o16[x] = d.photometric == 0 ? (uint16_t)~v : v;
                }
            } else if (d.spp == 1) {
                for (size_t x = 0; x < d.W; x++) dst[x] = d.photometric == 0 ? (uint8_t)~row[x] : row[x];
            } else if (d.spp == 3) {
                memcpy(dst, row, row_bytes);
            } else {  // RGBA -> RGB
                for (size_t x = 0; x < d.W; x++) { dst[3 * x] = row[4 * x]; dst[3 * x + 1] = row[4 * x + 1]; dst[3 * x + 2] = row[4 * x + 2]; }
            }
        }
    }
    return true;
}

// ------------------------------------------------------------------------------------------------ PNG
static const uint8_t kPngSig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};

struct PngHead {
    uint32_t W = 0, H = 0;
    int depth = 0, ctype = 0, interlace = 0;
};

static bool png_head(const uint8_t *data, size_t n, PngHead &h, std::string &err) {
    if (n < 33 || memcmp(data, kPngSig, 8) != 0) { err = "not a PNG"; return false; }
    Reader r{data, n, false};
    if (r.u32(8) != 13 || memcmp(data + 12, "IHDR", 4) != 0) { err = "PNG without IHDR"; return false; }
    h.W = r.u32(16); h.H = r.u32(20); h.depth = data[24]; h.ctype = data[25]; h.interlace = data[28];
    if (h.W == 0 || h.H == 0 || h.W > kMaxImageSide || h.H > kMaxImageSide) { err = "PNG without a valid size (1..16384 per side)"; return false; }
    if (h.interlace) { err = "interlaced PNGs are not supported"; return false; }
    const bool ok = (h.ctype == 0 && (h.depth == 8 || h.depth == 16)) || ((h.ctype == 2 || h.ctype == 6 || h.ctype == 3) && h.depth == 8);
    if (!ok) { err = "unsupported PNG colour type / bit depth"; return false; }
    return true;
}

// One scanline of a PNG, unfiltered.  One loop per filter type, and the left / up-left neighbours of the BPP byte lanes of a pixel are kept
// in REGISTERS: read back from the row just written, every byte of a Sub / Average / Paeth row waits for a store-to-load forward (round 5:
// 4.7 ms per 16-bit gradient image, more than twice its inflate).  The first pixel of a row has no left neighbour (a = c = 0).
template <int BPP>
static bool png_unfilter_row(int ft, const uint8_t *s, const uint8_t *pv, uint8_t *cu, size_t stride) {
    uint8_t a[BPP], c[BPP];
    for (int k = 0; k < BPP; k++) a[k] = c[k] = 0;
    const size_t whole = stride / BPP * BPP;
    switch (ft) {
        case 0: memcpy(cu, s, stride); return true;
        case 2:
            for (size_t i = 0; i < stride; i++) cu[i] = (uint8_t)(s[i] + pv[i]);
            return true;
        case 1:
            for (size_t i = 0; i < whole; i += BPP)
                for (int k = 0; k < BPP; k++) cu[i + k] = a[k] = (uint8_t)(s[i + k] + a[k]);
            return true;
        case 3:
            for (size_t i = 0; i < whole; i += BPP)
                for (int k = 0; k < BPP; k++) cu[i + k] = a[k] = (uint8_t)(s[i + k] + ((a[k] + pv[i + k]) >> 1));
            return true;
        case 4:
            // Paeth: the predictor is whichever of a (left), b (up), c (up-left) is nearest to a + b - c, ties in that order; written with the
            // differences (pa = |b - c|, pb = |a - c|, pc = |a + b - 2c|) so that it compiles to compares and conditional moves
            for (size_t i = 0; i < whole; i += BPP)
                for (int k = 0; k < BPP; k++) {
                    const int av = a[k], b = pv[i + k], cv = c[k];
                    const int db = b - cv, da = av - cv, dc = db + da;
                    const int pa = (db ^ (db >> 31)) - (db >> 31), pb = (da ^ (da >> 31)) - (da >> 31), pc = (dc ^ (dc >> 31)) - (dc >> 31);
                    // selects as masks (hipcc's host compiler made BRANCHES of the conditional form: on a noisy ramp they mispredict every
                    // other byte, 5.8 ns per byte): pred = pb < pa ? b : a, then pc < min(pa, pb) ? c : pred
                    const int m1 = -(int)(pb < pa);
                    const int best = (pa & ~m1) | (pb & m1);
                    int pred = (av & ~m1) | (b & m1);
                    const int m2 = -(int)(pc < best);
                    pred = (pred & ~m2) | (cv & m2);
                    cu[i + k] = a[k] = (uint8_t)(s[i + k] + pred);
                    c[k] = (uint8_t)b;
                }
            return true;
        default: return false;
    }
}

static bool png_decode(const uint8_t *data, size_t n, const ImgInfo &info, void *outv, std::string &err) {
    PngHead h;
    if (!png_head(data, n, h, err)) return false;
    Reader r{data, n, false};
    std::vector<uint8_t> idat, plte;
    for (size_t off = 8; off + 12 <= n;) {
        const uint32_t len = r.u32(off);
        if (!r.ok(off + 8, (size_t)len + 4)) { err = "PNG chunk outside the file"; return false; }
        if (memcmp(data + off + 4, "IDAT", 4) == 0) idat.insert(idat.end(), data + off + 8, data + off + 8 + len);
        else if (memcmp(data + off + 4, "PLTE", 4) == 0) plte.assign(data + off + 8, data + off + 8 + len);
        else if (memcmp(data + off + 4, "IEND", 4) == 0) break;
        off += (size_t)len + 12;
    }
    const int src_ch = h.ctype == 0 || h.ctype == 3 ? 1 : h.ctype == 2 ? 3 : 4;
    const size_t bpp = (size_t)src_ch * h.depth / 8, stride = (size_t)h.W * bpp;
    std::vector<uint8_t> raw((stride + 1) * h.H);
    if (!zlib_inflate_exact(idat.data(), idat.size(), raw.data(), raw.size())) { err = "PNG data is truncated or corrupt"; return false; }
    if (h.ctype == 3 && plte.size() < 3) { err = "palette PNG without PLTE"; return false; }
    std::vector<uint8_t> prev(stride, 0), cur(stride);
    uint8_t *out = static_cast<uint8_t *>(outv);
    for (size_t y = 0; y < h.H; y++) {
        const uint8_t *line = raw.data() + y * (stride + 1);
        const int ft = line[0];
        const uint8_t *s = line + 1;
        if (!(bpp == 1 ? png_unfilter_row<1>(ft, s, prev.data(), cur.data(), stride) : bpp == 2 ? png_unfilter_row<2>(ft, s, prev.data(), cur.data(), stride)
              : bpp == 3 ? png_unfilter_row<3>(ft, s, prev.data(), cur.data(), stride) : png_unfilter_row<4>(ft, s, prev.data(), cur.data(), stride))) {
            err = "PNG filter type out of range";
            return false;
        }
        if (h.ctype == 0 && h.depth == 16) {
            uint16_t *o16 = reinterpret_cast<uint16_t *>(out) + y * h.W;
            for (size_t x = 0; x < h.W; x++) o16[x] = (uint16_t)(cur[2 * x] << 8 | cur[2 * x + 1]);
        } else if (h.ctype == 0) {
            memcpy(out + y * h.W, cur.data(), h.W);
        } else if (h.ctype == 2) {
            memcpy(out + y * stride, cur.data(), stride);
        } else if (h.ctype == 6) {
            uint8_t *d = out + y * (size_t)h.W * 3;
            for (size_t x = 0; x < h.W; x++) { d[3 * x] = cur[4 * x]; d[3 * x + 1] = cur[4 * x + 1]; d[3 * x + 2] = cur[4 * x + 2]; }
        } else {
            uint8_t *d = out + y * (size_t)h.W * 3;
            for (size_t x = 0; x < h.W; x++) {
                const size_t k = (size_t)cur[x] * 3;
                if (k + 3 > plte.size()) { err = "PNG palette index out of range"; return false; }
                d[3 * x] = plte[k]; d[3 * x + 1] = plte[k + 1]; d[3 * x + 2] = plte[k + 2];
            }
        }
        prev.swap(cur);
    }
    (void)info;
    return true;
}

static bool image_probe(const uint8_t *data, size_t n, ImgInfo &info, std::string &err) {
    if (n >= 8 && memcmp(data, kPngSig, 8) == 0) {
        PngHead h;
        if (!png_head(data, n, h, err)) return false;
        info.W = (int)h.W; info.H = (int)h.H; info.bits = h.depth; info.channels = h.ctype == 0 ? 1 : 3;
        return true;
    }
    Reader r{};
    TiffDir d;
    if (!tiff_parse(data, n, r, d, err)) return false;
    info.W = (int)d.W; info.H = (int)d.H; info.bits = (int)d.bits; info.channels = d.spp == 1 ? 1 : 3;
    return true;
}

static bool image_decode(const uint8_t *data, size_t n, const ImgInfo &info, void *out, std::string &err) {
    if (n >= 8 && memcmp(data, kPngSig, 8) == 0) return png_decode(data, n, info, out, err);
    return tiff_decode(data, n, info, out, err);
}

// ---- zip archives: the reference's FileData of type zipEntry (API/model/FileData.java:22-26; NeuronMIPUtils.openInputStream, API/mips/
// NeuronMIPUtils.java:177-199: the MIPs of a library packed into one archive) --------------------------------------------------------
// A path of the form `<archive>.zip!<entry>` (or `...zip!/<entry>`) names an entry of a zip archive.  Stored and deflated entries, zip64
// (a library has more than 65,535 MIPs and more than 4 GB), CRC checked.  An entry that is not found by its full name is looked up by
// its file name alone, as the reference does ("slightly longer test", NeuronMIPUtils.java:163-170).  The central directory of an
// archive is parsed once and kept (the 8 most recent archives; a changed size or modification time re-reads it).
struct ZipEntry {
    uint64_t local_off = 0, csize = 0, usize = 0;
    uint32_t crc = 0;
    uint16_t method = 0, flags = 0;
};
struct ZipArchive {
    int fd = -1;
    uint64_t size = 0;
    int64_t mtime_ns = 0;
    std::unordered_map<std::string, ZipEntry> by_name, by_base;
    ~ZipArchive() { if (fd >= 0) close(fd); }
};
static uint16_t le16(const uint8_t *p) { return (uint16_t)(p[0] | p[1] << 8); }
static uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
static uint64_t le64(const uint8_t *p) { return (uint64_t)le32(p) | (uint64_t)le32(p + 4) << 32; }
static bool pread_all(int fd, void *dst, size_t n, uint64_t at) {
    uint8_t *q = static_cast<uint8_t *>(dst);
    while (n) {
        const ssize_t r = pread(fd, q, n, (off_t)at);
        if (r <= 0) return false;
        q += r, n -= (size_t)r, at += (uint64_t)r;
    }
    return true;
}
static std::shared_ptr<ZipArchive> zip_parse(const std::string &path, std::string &err) {
    auto za = std::make_shared<ZipArchive>();
    za->fd = open(path.c_str(), O_RDONLY | O_CLOEXEC);
    struct stat sb;
    if (za->fd < 0 || fstat(za->fd, &sb) != 0) { err = "cannot open " + path; return nullptr; }
    za->size = (uint64_t)sb.st_size;
    za->mtime_ns = (int64_t)sb.st_mtim.tv_sec * 1000000000ll + sb.st_mtim.tv_nsec;
    // end of central directory: the last 22 .. 22 + 65535 bytes
    const size_t tail = (size_t)std::min<uint64_t>(za->size, 22 + 65535);
    std::vector<uint8_t> t(tail);
    if (tail < 22 || !pread_all(za->fd, t.data(), tail, za->size - tail)) { err = path + " is not a zip archive"; return nullptr; }
    ptrdiff_t e = -1;
    for (ptrdiff_t i = (ptrdiff_t)tail - 22; i >= 0; i--)
        if (le32(&t[(size_t)i]) == 0x06054b50u && (size_t)i + 22 + le16(&t[(size_t)i + 20]) <= tail) { e = i; break; }
    if (e < 0) { err = path + " is not a zip archive (no end-of-central-directory record)"; return nullptr; }
    uint64_t n = le16(&t[(size_t)e + 10]), cd_size = le32(&t[(size_t)e + 12]), cd_off = le32(&t[(size_t)e + 16]);
    if (n == 0xffffu || cd_size == 0xffffffffu || cd_off == 0xffffffffu) {   // zip64: the locator sits in front of this record
        const uint64_t eocd_at = za->size - tail + (uint64_t)e;
        uint8_t loc[20], rec[56];
        if (eocd_at < 20 || !pread_all(za->fd, loc, 20, eocd_at - 20) || le32(loc) != 0x07064b50u) { err = path + ": zip64 locator missing"; return nullptr; }
        const uint64_t rec_at = le64(loc + 8);
        if (rec_at + 56 > za->size || !pread_all(za->fd, rec, 56, rec_at) || le32(rec) != 0x06064b50u) { err = path + ": zip64 end record missing"; return nullptr; }
        n = le64(rec + 32), cd_size = le64(rec + 40), cd_off = le64(rec + 48);
    }
    if (cd_off > za->size || cd_size > za->size - cd_off || n > cd_size / 46) { err = path + ": central directory outside the file"; return nullptr; }
    std::vector<uint8_t> cd((size_t)cd_size);
    if (cd_size && !pread_all(za->fd, cd.data(), (size_t)cd_size, cd_off)) { err = path + ": cannot read the central directory"; return nullptr; }
    size_t at = 0;
    for (uint64_t i = 0; i < n; i++) {
        if (at + 46 > cd.size() || le32(&cd[at]) != 0x02014b50u) { err = path + ": damaged central directory"; return nullptr; }
        ZipEntry ze;
        ze.flags = le16(&cd[at + 8]), ze.method = le16(&cd[at + 10]), ze.crc = le32(&cd[at + 16]);
        ze.csize = le32(&cd[at + 20]), ze.usize = le32(&cd[at + 24]), ze.local_off = le32(&cd[at + 42]);
        const size_t nl = le16(&cd[at + 28]), xl = le16(&cd[at + 30]), cl = le16(&cd[at + 32]);
        if (at + 46 + nl + xl + cl > cd.size()) { err = path + ": damaged central directory"; return nullptr; }
        std::string name(reinterpret_cast<const char *>(&cd[at + 46]), nl);
        for (size_t x = at + 46 + nl, xe = x + xl; x + 4 <= xe;) {   // the zip64 extra field replaces the 32-bit fields that read 0xffffffff
            const uint16_t id = le16(&cd[x]), len = le16(&cd[x + 2]);
            if (x + 4 + len > xe) break;
            if (id == 0x0001) {
                size_t q = x + 4;
                const size_t qe = q + len;
                if (ze.usize == 0xffffffffu && q + 8 <= qe) ze.usize = le64(&cd[q]), q += 8;
                if (ze.csize == 0xffffffffu && q + 8 <= qe) ze.csize = le64(&cd[q]), q += 8;
                if (ze.local_off == 0xffffffffu && q + 8 <= qe) ze.local_off = le64(&cd[q]), q += 8;
            }
            x += 4 + (size_t)len;
        }
        if (!name.empty() && name.back() != '/') {
            const size_t sl = name.rfind('/');
            za->by_base.emplace(sl == std::string::npos ? name : name.substr(sl + 1), ze);   // (the first entry of that file name wins)
            za->by_name[std::move(name)] = ze;
        }
        at += 46 + nl + xl + cl;
    }
    return za;
}
static std::shared_ptr<ZipArchive> zip_archive(const std::string &path, std::string &err) {
    static std::mutex mu;
    static std::vector<std::pair<std::string, std::shared_ptr<ZipArchive>>> cache;   // most recent last
    struct stat sb;
    if (stat(path.c_str(), &sb) != 0) { err = "cannot open " + path; return nullptr; }
    const int64_t mt = (int64_t)sb.st_mtim.tv_sec * 1000000000ll + sb.st_mtim.tv_nsec;
    {
        std::lock_guard<std::mutex> lk(mu);
        for (size_t i = 0; i < cache.size(); i++)
            if (cache[i].first == path) {
                if (cache[i].second->size == (uint64_t)sb.st_size && cache[i].second->mtime_ns == mt) {
                    auto hit = cache[i];
                    cache.erase(cache.begin() + (std::ptrdiff_t)i);
                    cache.push_back(hit);
                    return hit.second;
                }
                cache.erase(cache.begin() + (std::ptrdiff_t)i);
                break;
            }
    }
    auto za = zip_parse(path, err);   // (outside the lock: two threads may parse the same archive once each)
    if (!za) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    cache.emplace_back(path, za);
    if (cache.size() > 8) cache.erase(cache.begin());
    return za;
}
static bool zip_read(const std::string &archive, const std::string &entry, std::vector<uint8_t> &buf, std::string &err) {
    auto za = zip_archive(archive, err);
    if (!za) return false;
    auto it = za->by_name.find(entry);
    if (it == za->by_name.end()) {
        const size_t sl = entry.rfind('/');
        it = za->by_base.find(sl == std::string::npos ? entry : entry.substr(sl + 1));
        if (it == za->by_base.end()) { err = "no entry " + entry + " in " + archive; return false; }
    }
    const ZipEntry &ze = it->second;
    const std::string what = archive + "!" + entry;
    if (ze.flags & 1) { err = what + " is encrypted"; return false; }
    if (ze.method != 0 && ze.method != 8) { err = what + ": compression method " + std::to_string(ze.method) + " is not supported (stored and deflated entries are)"; return false; }
    constexpr uint64_t kMaxEntry = (uint64_t)1 << 30;   // (a header is untrusted input)
    if (ze.usize > kMaxEntry || ze.csize > kMaxEntry || (ze.method == 0 && ze.csize != ze.usize)) { err = what + ": implausible entry size"; return false; }
    uint8_t lh[30];
    if (ze.local_off + 30 > za->size || !pread_all(za->fd, lh, 30, ze.local_off) || le32(lh) != 0x04034b50u) { err = what + ": damaged local header"; return false; }
    const uint64_t data = ze.local_off + 30 + le16(lh + 26) + le16(lh + 28);
    if (data > za->size || ze.csize > za->size - data) { err = what + ": entry data outside the archive"; return false; }
    buf.resize((size_t)ze.usize);
    if (ze.method == 0) {
        if (ze.usize && !pread_all(za->fd, buf.data(), (size_t)ze.usize, data)) { err = "cannot read " + what; return false; }
    } else {
        thread_local std::vector<uint8_t> comp;
        comp.resize((size_t)ze.csize);
        if (ze.csize && !pread_all(za->fd, comp.data(), (size_t)ze.csize, data)) { err = "cannot read " + what; return false; }
        if (!zlib_inflate_exact(comp.data(), comp.size(), buf.data(), buf.size(), true)) { err = what + ": damaged deflate stream"; return false; }
    }
    if ((uint32_t)crc32(0L, buf.data(), (uInt)buf.size()) != ze.crc) { err = what + ": CRC mismatch"; return false; }
    return true;
}

static bool read_file(const char *path, std::vector<uint8_t> &buf, std::string &err) {
    {   // `<archive>.zip!<entry>`: an entry of a zip archive
        const char *bang = nullptr;
        for (const char *q = path; (q = strchr(q, '!')) != nullptr; q++)
            if (q - path >= 4 && strncasecmp(q - 4, ".zip", 4) == 0) { bang = q; break; }
        if (bang) return zip_read(std::string(path, (size_t)(bang - path)), std::string(bang[1] == '/' ? bang + 2 : bang + 1), buf, err);
    }
    FILE *f = fopen(path, "rb");
    if (!f) { err = std::string("cannot open ") + path; return false; }
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (sz < 0) { fclose(f); err = std::string("cannot read ") + path; return false; }
    buf.resize((size_t)sz);
    const size_t got = sz ? fread(buf.data(), 1, (size_t)sz, f) : 0;
    fclose(f);
    if (got != (size_t)sz) { err = std::string("cannot read ") + path; return false; }
    return true;
}

bool decode_file_into(const char *path, int W, int H, int kind, void *dst, std::string &err) {
    thread_local std::vector<uint8_t> file, tmp8;
    const size_t px = (size_t)W * H;
    ImgInfo info;
    try {
        if (!read_file(path, file, err) || !image_probe(file.data(), file.size(), info, err)) return false;
        if (info.W != W || info.H != H) { err = "image size differs from the library's"; return false; }
        if (kind == 0) {
            if (info.channels != 3 || info.bits != 8) { err = "is not an 8-bit RGB image"; return false; }
            return image_decode(file.data(), file.size(), info, dst, err);
        }
        uint16_t *d16 = static_cast<uint16_t *>(dst);
        if (info.channels != 1) { err = "gradient image must be grayscale"; return false; }
        if (info.bits == 16) return image_decode(file.data(), file.size(), info, d16, err);
        if (info.bits != 8) { err = "gradient image must be 8- or 16-bit"; return false; }
        tmp8.resize(px);   // GRAY8: widened (ImageArray.get semantics, ByteImageArray.java:3-16)
        if (!image_decode(file.data(), file.size(), info, tmp8.data(), err)) return false;
        for (size_t q = 0; q < px; q++) d16[q] = tmp8[q];
        return true;
    } catch (const std::exception &e) {   // (an exception must not leave a worker thread)
        err = e.what();
        return false;
    }
}

}  // namespace cds

using namespace cds;

extern "C" {

// (no C++ exception may cross the C ABI: allocation failures inside the decoders become error codes here)
#define CDS_NOEXCEPT_BEGIN try {
#define CDS_NOEXCEPT_END                                                                         \
    }                                                                                            \
    catch (const std::bad_alloc &) { return fail(CDS_ERR_OUT_OF_MEMORY, "out of host memory"); } \
    catch (const std::exception &e) { return fail(CDS_ERR_INVALID_ARGUMENT, e.what()); }

int cds_image_probe(const uint8_t *data, size_t size, cds_image_info *info) {
    if (!data || !info) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    CDS_NOEXCEPT_BEGIN
    ImgInfo i;
    std::string err;
    if (!image_probe(data, size, i, err)) return fail(CDS_ERR_INVALID_ARGUMENT, err);
    info->width = i.W; info->height = i.H; info->channels = i.channels; info->bits = i.bits;
    return CDS_OK;
    CDS_NOEXCEPT_END
}

int cds_image_decode(const uint8_t *data, size_t size, void *out, size_t out_bytes) {
    if (!data || !out) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    CDS_NOEXCEPT_BEGIN
    ImgInfo i;
    std::string err;
    if (!image_probe(data, size, i, err)) return fail(CDS_ERR_INVALID_ARGUMENT, err);
    const size_t need = (size_t)i.W * i.H * i.channels * (i.bits / 8);
    if (out_bytes < need) return fail(CDS_ERR_INVALID_ARGUMENT, "output buffer too small");
    if (!image_decode(data, size, i, out, err)) return fail(CDS_ERR_INVALID_ARGUMENT, err);
    return CDS_OK;
    CDS_NOEXCEPT_END
}

static int append_files_impl(cds_sparse_library *sl, const char *const *paths, int64_t count, int32_t threads);

int cds_sparse_library_append_files(cds_sparse_library *sl, const char *const *paths, int64_t count, int32_t threads) {
    CDS_NOEXCEPT_BEGIN
    const int64_t n_images0 = sl ? sl->n_images : 0, n_px0 = sl ? sl->n_px : 0;
    (void)cds::take_density_refusal();
    const int rc = append_files_impl(sl, paths, count, threads);
    if (rc != CDS_OK && cds::take_density_refusal()) {   // a DENSITY refusal leaves nothing of the call behind (the message survives: same
                                                          // thread); any other failure -- an unsupported file among them -- keeps the batches
                                                          // appended before it, as the header says
        const std::string msg = cds_last_error();
        cds::sparse_library_rollback(sl, n_images0, n_px0);
        return fail(rc, msg);
    }
    return rc;
    CDS_NOEXCEPT_END
}

// ---- the records of a decoded image, built on the HOST (round 5) -------------------------------------------------------------------
// cds_sparse_library_append_files used to send every decoded frame (2 MB) to the device and convert it there: both compressed formats
// then waited for the upload.  The decode threads now build the image's records themselves -- the format of
// cds_internal.hpp / sparse_rows_kernel: strip by strip, row by row, x ascending; (x - strip start) << 22 | band << 16 | second << 8 |
// max, bit 31 on the first record of its (strip, row) cell, one filler per empty cell -- and ~80 KB per image cross PCIe through the
// import path, whose device-side validation (sparse_validate_kernel) checks every cell like any other imported shard.  The band is
// made with the DEVICE's reciprocal table (cds_context::rcp: v_rcp_f32 of 0..255 as this GPU computes it) and the same single-precision
// multiplies and truncation as px_record in cds_sparse.hip, so the records are bit-identical to the device-built ones (tests).
struct HostRecords {
    std::vector<uint32_t> rec;       // the image's records
    std::vector<uint32_t> cells;     // [nstrips][H + 1] offsets of the (strip, row) cells
};
static void host_records_of_image(const uint8_t *rgb, int W, int H, int nstrips, int strip_w, uint32_t min_thr, const float *rcp, HostRecords &out) {
    out.rec.clear();
    out.cells.assign((size_t)nstrips * (H + 1), 0u);
    for (int k = 0; k < nstrips; k++) {
        const int x0 = k * strip_w, x1 = std::min(W, (k + 1) * strip_w);
        for (int y = 0; y < H; y++) {
            out.cells[(size_t)k * (H + 1) + y] = (uint32_t)out.rec.size();
            const uint8_t *row = rgb + ((size_t)y * W + x0) * 3;
            const size_t at = out.rec.size();
            for (int x = x0; x < x1;) {
                if (x + 8 <= x1) {   // eight black pixels at a time (a MIP is a few percent signal)
                    uint64_t a, b, c;
                    memcpy(&a, row, 8), memcpy(&b, row + 8, 8), memcpy(&c, row + 16, 8);
                    if ((a | b | c) == 0) { x += 8, row += 24; continue; }
                }
                const uint32_t r = row[0], g = row[1], bl = row[2];
                row += 3;
                const uint32_t mx = std::max(r, std::max(g, bl)), mn = std::min(r, std::min(g, bl)), md = r + g + bl - mx - mn;
                if (mx > md && mx > min_thr) {
                    const uint32_t idx = (r > g ? 4u : 0u) | (g > bl ? 8u : 0u) | (bl > r ? 16u : 0u);
                    const uint32_t cls = (0x03125460u >> idx) & 7u;
                    volatile float hf = (float)md * rcp[mx];       // one IEEE single multiply, as the device's v_mul_f32
                    volatile float h5 = hf * 5.0f;
                    const uint32_t band = cls * 5u + (uint32_t)h5;
                    out.rec.push_back((band << kRecBandShift) | (md << 8) | mx | ((uint32_t)(x - x0) << kRecXShift));
                }
                x++;
            }
            if (out.rec.size() == at) out.rec.push_back(kRecRowFlag);   // the filler of an empty cell
            else out.rec[at] |= kRecRowFlag;
        }
        out.cells[(size_t)k * (H + 1) + H] = (uint32_t)out.rec.size();
    }
}

static int append_files_impl(cds_sparse_library *sl, const char *const *paths, int64_t count, int32_t threads) {
    if (!sl || (count > 0 && !paths)) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (count < 0) return fail(CDS_ERR_INVALID_ARGUMENT, "count is negative");
    if (count == 0) return CDS_OK;
    if (cds_sparse_library_size(sl) + count > sl->cap_images && !sl->growable) return fail(CDS_ERR_OUT_OF_MEMORY, "sparse library is full (images)");
    const int nthreads = std::max(1, std::min(threads > 0 ? threads : (int)std::thread::hardware_concurrency(), 64));
    const int64_t batch = 64;  // = the dense staging slots of cds_sparse_library_append_host
    const size_t img_bytes = (size_t)3 * sl->W * sl->H;
    static const bool dense_upload = getenv("CDS_INGEST_DENSE_UPLOAD") != nullptr;   // (A/B: round 2's path -- decoded frames to the device, records made there)
    if (!dense_upload) {
        // ---- records built by the decode threads: two batches in flight (decode + build || upload + validate) ----
        const int nstrips = sl->nstrips, strip_w = sl->strip_w, W = sl->W, H = sl->H;
        const uint32_t min_thr = (uint32_t)sl->min_threshold;
        const float *rcp = sl->ctx->rcp;
        struct Batch {
            std::vector<HostRecords> img;
            std::vector<uint64_t> off;
            std::vector<uint32_t> cells, rec;
        } bt[2];
        std::string first_err;
        std::mutex err_mu;
        auto decode_batch = [&](int64_t b0, int64_t nb, Batch &B) {
            B.img.resize((size_t)nb);
            std::atomic<int64_t> next{0};
            auto worker = [&]() {
                std::vector<uint8_t> file, frame;
                for (int64_t i = next.fetch_add(1); i < nb; i = next.fetch_add(1)) {
                    std::string err;
                    ImgInfo info;
                    bool ok = false;
                    try {
                        frame.resize(img_bytes);
                        ok = read_file(paths[b0 + i], file, err) && image_probe(file.data(), file.size(), info, err);
                        if (ok && (info.W != W || info.H != H || info.channels != 3 || info.bits != 8)) {
                            err = "is not an 8-bit RGB image of the library's size";
                            ok = false;
                        }
                        if (ok) ok = image_decode(file.data(), file.size(), info, frame.data(), err);
                        if (ok) host_records_of_image(frame.data(), W, H, nstrips, strip_w, min_thr, rcp, B.img[(size_t)i]);
                    } catch (const std::exception &e) {   // (an exception must not leave a worker thread)
                        err = e.what();
                        ok = false;
                    }
                    if (!ok) {
                        std::lock_guard<std::mutex> lk(err_mu);
                        if (first_err.empty()) first_err = std::string(paths[b0 + i]) + ": " + err;
                    }
                }
            };
            std::vector<std::thread> pool;
            try {
                for (int t = 1; t < std::min<int64_t>(nthreads, nb); t++) pool.emplace_back(worker);
            } catch (const std::exception &) {   // (fewer threads than asked for: the ones that started do the work)
            }
            worker();
            for (auto &th : pool) th.join();
            if (!first_err.empty()) return;
            // the batch as ONE import: image offsets, cell offsets, records back to back
            const size_t cells = (size_t)nstrips * (H + 1);
            B.off.assign((size_t)nb + 1, 0);
            for (int64_t i = 0; i < nb; i++) B.off[(size_t)i + 1] = B.off[(size_t)i] + B.img[(size_t)i].rec.size();
            B.cells.resize((size_t)nb * cells);
            B.rec.resize((size_t)B.off[(size_t)nb]);
            for (int64_t i = 0; i < nb; i++) {
                memcpy(B.cells.data() + (size_t)i * cells, B.img[(size_t)i].cells.data(), cells * sizeof(uint32_t));
                memcpy(B.rec.data() + B.off[(size_t)i], B.img[(size_t)i].rec.data(), B.img[(size_t)i].rec.size() * sizeof(uint32_t));
            }
        };
        int rc_gpu = CDS_OK;
        std::string gpu_err;
        std::thread gpu;
        struct Joiner {
            std::thread &t;
            ~Joiner() { if (t.joinable()) t.join(); }
        } joiner{gpu};
        static const bool decode_only = getenv("CDS_INGEST_DECODE_ONLY") != nullptr;   // (measurement: host decode + record building alone)
        for (int64_t b0 = 0, k = 0; b0 < count; b0 += batch, k++) {
            const int64_t nb = std::min(batch, count - b0);
            // (batch k decodes into the buffers of batch k - 2, whose import was joined before batch k - 1's was started)
            Batch &B = bt[k & 1];
            decode_batch(b0, nb, B);
            if (gpu.joinable()) gpu.join();
            if (!first_err.empty()) return fail(CDS_ERR_INVALID_ARGUMENT, first_err);
            if (rc_gpu) return fail(rc_gpu, gpu_err);
            if (decode_only) continue;
            gpu = std::thread([sl, nb, &B, nstrips, strip_w, &rc_gpu, &gpu_err]() {
                rc_gpu = cds_sparse_library_import(sl, nb, B.off.data(), B.cells.data(), B.rec.data(), nstrips, strip_w, sl->min_threshold);
                if (rc_gpu) gpu_err = cds_last_error();   // (the message is thread-local: carry it over)
            });
        }
        if (gpu.joinable()) gpu.join();
        if (rc_gpu) return fail(rc_gpu, gpu_err);
        return CDS_OK;
    }
    // the decoded batches live in pinned host memory: the upload is then one DMA at PCIe rate (pageable memory: 17 GB/s)
    struct Pinned {
        uint8_t *p = nullptr;
        ~Pinned() { if (p) hipHostFree(p); }
        uint8_t *data() const { return p; }
    } buf[2];
    CDS_HIP(hipSetDevice(sl->ctx->device));
    CDS_HIP(hipHostMalloc((void **)&buf[0].p, (size_t)std::min(batch, count) * img_bytes, hipHostMallocDefault));
    if (count > batch) CDS_HIP(hipHostMalloc((void **)&buf[1].p, (size_t)std::min(batch, count - batch) * img_bytes, hipHostMallocDefault));
    std::string first_err;
    std::mutex err_mu;
    auto decode_batch = [&](int64_t b0, int64_t nb, uint8_t *dst) {
        std::atomic<int64_t> next{0};
        auto worker = [&]() {
            std::vector<uint8_t> file;
            for (int64_t i = next.fetch_add(1); i < nb; i = next.fetch_add(1)) {
                std::string err;
                ImgInfo info;
                bool ok = false;
                try {
                    ok = read_file(paths[b0 + i], file, err) && image_probe(file.data(), file.size(), info, err);
                    if (ok && (info.W != sl->W || info.H != sl->H || info.channels != 3 || info.bits != 8)) {
                        err = "is not an 8-bit RGB image of the library's size";
                        ok = false;
                    }
                    if (ok) ok = image_decode(file.data(), file.size(), info, dst + (size_t)i * img_bytes, err);
                } catch (const std::exception &e) {   // (an exception must not leave a worker thread)
                    err = e.what();
                    ok = false;
                }
                if (!ok) {
                    std::lock_guard<std::mutex> lk(err_mu);
                    if (first_err.empty()) first_err = std::string(paths[b0 + i]) + ": " + err;
                }
            }
        };
        std::vector<std::thread> pool;
        try {
            for (int t = 1; t < std::min<int64_t>(nthreads, nb); t++) pool.emplace_back(worker);
        } catch (const std::exception &) {   // (fewer threads than asked for: the ones that started do the work)
        }
        worker();
        for (auto &th : pool) th.join();
    };
    // batch k is decoded by the pool while batch k-1 is uploaded and converted on the GPU by one more thread
    int rc_gpu = CDS_OK;
    std::string gpu_err;
    std::thread gpu;
    struct Joiner {   // (an exception leaving this function must not destroy a joinable thread)
        std::thread &t;
        ~Joiner() { if (t.joinable()) t.join(); }
    } joiner{gpu};
    for (int64_t b0 = 0, k = 0; b0 < count; b0 += batch, k++) {
        const int64_t nb = std::min(batch, count - b0);
        decode_batch(b0, nb, buf[k & 1].data());
        if (gpu.joinable()) gpu.join();
        if (!first_err.empty()) return fail(CDS_ERR_INVALID_ARGUMENT, first_err);
        if (rc_gpu) return fail(rc_gpu, gpu_err);
        const uint8_t *src = buf[k & 1].data();
        static const bool decode_only = getenv("CDS_INGEST_DECODE_ONLY") != nullptr;   // (measurement: host decode rate alone)
        if (decode_only) continue;
        gpu = std::thread([sl, nb, src, &rc_gpu, &gpu_err]() {
            rc_gpu = cds_sparse_library_append_host(sl, nb, src);
            if (rc_gpu) gpu_err = cds_last_error();   // (the message is thread-local: carry it over)
        });
    }
    if (gpu.joinable()) gpu.join();
    if (rc_gpu) return fail(rc_gpu, gpu_err);
    return CDS_OK;
}

int cds_library_load_files(cds_library *lib, int64_t first, int64_t count, const char *const *target_paths,
                           const char *const *gradient_paths, const char *const *zgap_paths, int32_t threads) {
    CDS_NOEXCEPT_BEGIN
    if (!lib || (count > 0 && !target_paths)) return fail(CDS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (first < 0 || count < 0 || first + count > lib->capacity) return fail(CDS_ERR_INVALID_ARGUMENT, "slot range outside the library");
    if (count == 0) return CDS_OK;
    const int W = lib->W, H = lib->H;
    const size_t px = (size_t)W * H, rgb_bytes = 3 * px, grad_bytes = 2 * px;
    const int nthreads = std::max(1, std::min(threads > 0 ? threads : (int)std::thread::hardware_concurrency(), 64));
    const int64_t batch = 64;
    const bool any_grad = gradient_paths != nullptr, any_zgap = zgap_paths != nullptr;
    struct Pinned {   // (from the pool above: pinning ~700 MB per call would cost as much as decoding the call's files)
        uint8_t *p = nullptr;
        ~Pinned() { g_pinned.put(p); }
    } tb[2], gb[2], zb[2];
    CDS_HIP(hipSetDevice(lib->ctx->device));
    for (int k = 0; k < (count > batch ? 2 : 1); k++) {
        const size_t n = (size_t)std::min(batch, count);
        bool ok = (tb[k].p = g_pinned.get(n * rgb_bytes)) != nullptr;
        if (ok && any_grad) ok = (gb[k].p = g_pinned.get(n * grad_bytes)) != nullptr;
        if (ok && any_zgap) ok = (zb[k].p = g_pinned.get(n * rgb_bytes)) != nullptr;
        if (!ok) return fail(CDS_ERR_OUT_OF_MEMORY, "no pinned host memory for the decoded images");
    }
    std::string first_err;
    std::mutex err_mu;
    // one work item = one FILE (three per slot): the pool is busy even when a batch holds fewer slots than threads
    auto decode_batch = [&](int64_t b0, int64_t nb, int k) {
        std::atomic<int64_t> next{0};
        auto worker = [&]() {
            for (int64_t w = next.fetch_add(1); w < 3 * nb; w = next.fetch_add(1)) {
                const int64_t i = w / 3;
                const int kind = (int)(w % 3);   // 0 target, 1 gradient, 2 zgap
                const char *path = kind == 0 ? target_paths[b0 + i] : kind == 1 ? (any_grad ? gradient_paths[b0 + i] : nullptr) : (any_zgap ? zgap_paths[b0 + i] : nullptr);
                if (!path) {
                    if (kind == 0) {
                        std::lock_guard<std::mutex> lk(err_mu);
                        if (first_err.empty()) first_err = "a target path is NULL";
                    } else if (kind == 1 && any_grad) memset(gb[k].p + (size_t)i * grad_bytes, 0, grad_bytes);
                    else if (kind == 2 && any_zgap) memset(zb[k].p + (size_t)i * rgb_bytes, 0, rgb_bytes);
                    continue;
                }
                std::string err;
                void *dst = kind == 0 ? (void *)(tb[k].p + (size_t)i * rgb_bytes) : kind == 1 ? (void *)(gb[k].p + (size_t)i * grad_bytes) : (void *)(zb[k].p + (size_t)i * rgb_bytes);
                const bool ok = decode_file_into(path, W, H, kind == 1 ? 1 : 0, dst, err);
                if (!ok) {
                    std::lock_guard<std::mutex> lk(err_mu);
                    if (first_err.empty()) first_err = std::string(path) + ": " + err;
                }
            }
        };
        std::vector<std::thread> pool;
        try {
            for (int t = 1; t < std::min<int64_t>(nthreads, 3 * nb); t++) pool.emplace_back(worker);
        } catch (const std::exception &) {
        }
        worker();
        for (auto &th : pool) th.join();
    };
    int rc_gpu = CDS_OK;
    std::string gpu_err;
    std::thread gpu;
    struct Joiner {
        std::thread &t;
        ~Joiner() { if (t.joinable()) t.join(); }
    } joiner{gpu};
    static const bool timed = getenv("CDS_TIME_LOAD") != nullptr;   // (diagnostic: decode / wait-for-upload seconds of the call on stderr)
    double t_dec = 0, t_wait = 0;
    const auto t_call = std::chrono::steady_clock::now();
    for (int64_t b0 = 0, k = 0; b0 < count; b0 += batch, k++) {
        const int64_t nb = std::min(batch, count - b0);
        const int kk = (int)(k & 1);
        const auto t0 = std::chrono::steady_clock::now();
        decode_batch(b0, nb, kk);   // (into the buffers of batch k - 2, whose upload was joined before batch k - 1's was started)
        const auto t1 = std::chrono::steady_clock::now();
        if (gpu.joinable()) gpu.join();
        t_dec += std::chrono::duration<double>(t1 - t0).count();
        t_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
        if (!first_err.empty()) return fail(CDS_ERR_INVALID_ARGUMENT, first_err);
        if (rc_gpu) return fail(rc_gpu, gpu_err);
        gpu = std::thread([&, b0, nb, kk]() {
            int rc = cds_library_upload(lib, first + b0, nb, tb[kk].p);
            if (rc == CDS_OK && (any_grad || any_zgap))
                rc = cds_library_upload_variants(lib, first + b0, nb, any_grad ? reinterpret_cast<const uint16_t *>(gb[kk].p) : nullptr, any_zgap ? zb[kk].p : nullptr);
            if (rc == CDS_OK) {   // a slot whose variant file was not given has none (the upload marked the whole range)
                std::lock_guard<std::mutex> lk(lib->mu);
                for (int64_t i = 0; i < nb; i++) {
                    if (any_grad && !gradient_paths[b0 + i]) lib->has_grad[(size_t)(first + b0 + i)] = 0;
                    if (any_zgap && !zgap_paths[b0 + i]) lib->has_zgap[(size_t)(first + b0 + i)] = 0;
                }
            }
            rc_gpu = rc;
            if (rc) gpu_err = cds_last_error();
        });
    }
    if (gpu.joinable()) gpu.join();
    if (timed)
        fprintf(stderr, "[cds load_files] %lld slots: %.3f s (decode %.3f s, waiting for uploads %.3f s, %d threads)\n", (long long)count,
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t_call).count(), t_dec, t_wait, nthreads);
    if (rc_gpu) return fail(rc_gpu, gpu_err);
    return CDS_OK;
    CDS_NOEXCEPT_END
}

}  // extern "C"
