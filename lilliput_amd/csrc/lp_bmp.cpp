// lp_bmp.cpp -- see lp_bmp.h. Follows cv::BmpDecoder's observable behaviour, quirks included (pinned against the reference's own
// grfmt_bmp.cpp.o on generated and damaged files, tests/test_bmp.py): a 12-byte OS/2 header always yields a grey image; 32-bit files
// are three channels unless they use bit fields; skipped pixels of an RLE stream take palette entry 0; a run that would leave its row
// ends the decode with an error; 5-5-5 / 5-6-5 samples are shifted up, not replicated.
#include "lp_bmp.h"

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

namespace {

struct Short {}; // the byte stream ran out: cv::RLByteStream throws, the decoder's caller answers false

struct Reader {
    const uint8_t* p;
    size_t n, i;
    uint8_t byte() { if (i >= n) throw Short(); return p[i++]; }
    uint32_t word() { const uint32_t a = byte(); return a | ((uint32_t)byte() << 8); }
    uint32_t dword() { const uint32_t a = word(); return a | (word() << 16); }
    void bytes(uint8_t* dst, size_t k)
    {
        const size_t have = i < n ? n - i : 0;
        memcpy(dst, p + i, std::min(k, have)); // what is there is delivered before the stream reports its end
        if (k > have) { i = n; throw Short(); }
        i += k;
    }
    void skip(long k) { i = (size_t)((long)i + k); if ((long)i < 0) i = 0; } // the position may run past the end: the next read finds out
    void seek(long k) { i = k < 0 ? 0 : (size_t)k; }
};

inline uint8_t grey_of(unsigned b, unsigned g, unsigned r) { return (uint8_t)((b * 1868u + g * 9617u + r * 4899u + (1u << 13)) >> 14); } // icvCvt_BGR2Gray_8u_C3C1R

} // namespace

bool lp_bmp_read_info(const uint8_t* data, size_t len, LpBmpInfo& info)
{
    Reader s{data, len, 0};
    bool ok = false, iscolor = false;
    info = LpBmpInfo();
    memset(info.palette, 0, sizeof(info.palette));
    for (int c = 0; c < 4; c++) { info.mask[c] = 0; info.shift[c] = -1; }
    long height = 0;
    try {
        s.skip(10);
        info.offset = (int)s.dword();
        const int size = (int)s.dword();
        if (size <= 0) return false; // CV_Assert(size > 0)
        if (size >= 36) {
            info.width = (int)s.dword();
            height = (int)s.dword();
            info.bpp = (int)(s.dword() >> 16);
            const int comp = (int)s.dword();
            if (comp < 0 || comp > 3) return false;
            info.compression = comp;
            s.skip(12);
            const int clrused = (int)s.dword();
            if (info.bpp == 32 && comp == 3 && size >= 56) {
                s.skip(4); // important colours
                for (int c = 0; c < 4; c++) {
                    uint32_t m = s.dword();
                    info.mask[c] = m;
                    if (m) {
                        int sh = 0;
                        while (!(m & 1u)) { m >>= 1; sh++; }
                        info.shift[c] = sh;
                    }
                }
                s.skip(size - 56);
            } else
                s.skip(size - 36);
            const int b = info.bpp;
            if (info.width > 0 && height != 0 &&
                (((b == 1 || b == 4 || b == 8 || b == 24 || b == 32) && comp == 0) || ((b == 16 || b == 32) && (comp == 0 || comp == 3)) || (b == 4 && comp == 2) ||
                 (b == 8 && comp == 1))) {
                iscolor = true;
                ok = true;
                if (b <= 8) {
                    if (clrused < 0 || clrused > 256) return false;
                    s.bytes(&info.palette[0][0], (size_t)(clrused == 0 ? 1 << b : clrused) * 4);
                    iscolor = false; // IsColorPalette over the 1 << bpp entries
                    for (int k = 0; k < (1 << b); k++)
                        if (info.palette[k][0] != info.palette[k][1] || info.palette[k][0] != info.palette[k][2]) { iscolor = true; break; }
                } else if (b == 16 && comp == 3) {
                    const uint32_t red = s.dword(), green = s.dword(), blue = s.dword();
                    if (blue == 0x1f && green == 0x3e0 && red == 0x7c00) info.bpp = 15;
                    else if (blue == 0x1f && green == 0x7e0 && red == 0xf800) {}
                    else ok = false;
                } else if (b == 16 && comp == 0)
                    info.bpp = 15;
            }
        } else if (size == 12) {
            info.width = (int)s.word();
            height = (int)s.word();
            info.bpp = (int)(s.dword() >> 16);
            info.compression = 0;
            const int b = info.bpp;
            if (info.width > 0 && height != 0 && (b == 1 || b == 4 || b == 8 || b == 24 || b == 32)) {
                if (b <= 8) {
                    uint8_t buf[256 * 3];
                    s.bytes(buf, (size_t)(1 << b) * 3);
                    for (int k = 0; k < (1 << b); k++) { info.palette[k][0] = buf[3 * k]; info.palette[k][1] = buf[3 * k + 1]; info.palette[k][2] = buf[3 * k + 2]; }
                }
                ok = true; // (and iscolor stays false: the OS/2 form always comes out grey)
            }
        }
    } catch (const Short&) { return false; }
    info.channels = iscolor ? ((info.bpp == 32 && info.compression != 0) ? 4 : 3) : 1;
    info.bottom_up = height > 0;
    info.height = (int)(height < 0 ? -height : height);
    return ok;
}

bool lp_bmp_read_data(const uint8_t* data, size_t len, const LpBmpInfo& bi, uint8_t* out, size_t step_in)
{
    const int W = bi.width, H = bi.height, nch = bi.channels > 1 ? 3 : 1, cn = bi.channels;
    const bool color = cn > 1;
    if ((uint64_t)H * (uint64_t)W * (uint64_t)nch >= (1ull << 30)) return false; // "doesn't support large images >= 1Gb"
    if (bi.offset < 0) return false;
    // cv::BmpDecoder computes the row pitch in 32-bit int; a header whose width makes that overflow (W = 2^26 at 32 bits per pixel passes
    // the size test above with H = 1) ends in a failed allocation there and a refused decode here -- never in a wrapped pitch
    const int64_t pitch64 = (((int64_t)W * (bi.bpp != 15 ? bi.bpp : 16) + 7) / 8 + 3) & ~(int64_t)3;
    if (W <= 0 || H <= 0 || pitch64 <= 0 || pitch64 > 0x7fffffff - 64 || (int64_t)W * nch > 0x7fffffff) return false;
    const int src_pitch = (int)pitch64;
    const int width3 = W * nch;
    long step = (long)step_in;
    uint8_t* d = out;
    if (bi.bottom_up) { d += (size_t)(H - 1) * step_in; step = -step; }
    std::vector<uint8_t> srcv;
    try { srcv.resize((size_t)src_pitch + 32); } catch (const std::exception&) { return false; } // nothing unwinds through the C ABI
    uint8_t* src = srcv.data();
    uint8_t grey_pal[256];
    memset(grey_pal, 0, sizeof(grey_pal));
    if (!color && bi.bpp <= 8)
        for (int k = 0; k < (1 << bi.bpp); k++) grey_pal[k] = grey_of(bi.palette[k][0], bi.palette[k][1], bi.palette[k][2]);
    auto put = [&](uint8_t* p, int idx) { // one pixel of a paletted image
        if (color) { p[0] = bi.palette[idx][0]; p[1] = bi.palette[idx][1]; p[2] = bi.palette[idx][2]; }
        else p[0] = grey_pal[idx];
    };
    // FillUniColor / FillUniGray: `count` bytes' worth of pixels of one palette entry, across row ends; y counts finished rows
    auto fill = [&](uint8_t* p, uint8_t*& line_end, int& y, int count, int idx) -> uint8_t* {
        do {
            uint8_t* end = p + count;
            if (end > line_end) end = line_end;
            count -= (int)(end - p);
            for (; p < end; p += nch) put(p, idx);
            if (p >= line_end) {
                line_end += step;
                p = line_end - width3;
                if (++y >= H) break;
            }
        } while (count > 0);
        return p;
    };
    Reader s{data, len, 0};
    bool result = false;
    try {
        s.seek(bi.offset);
        switch (bi.bpp) {
        case 1:
            for (int y = 0; y < H; y++, d += step) {
                s.bytes(src, (size_t)src_pitch);
                for (int x = 0; x < W; x++) put(d + (size_t)x * nch, (src[x >> 3] >> (7 - (x & 7))) & 1);
            }
            result = true;
            break;
        case 4:
            if (bi.compression == 0) {
                for (int y = 0; y < H; y++, d += step) {
                    s.bytes(src, (size_t)src_pitch);
                    for (int x = 0; x < W; x++) put(d + (size_t)x * nch, (x & 1) ? src[x >> 1] & 15 : src[x >> 1] >> 4);
                }
                result = true;
            } else if (bi.compression == 2) {
                uint8_t* line_end = d + width3;
                int y = 0;
                for (;;) {
                    int code = (int)s.word();
                    const int n = code & 255;
                    code >>= 8;
                    if (n != 0) { // encoded mode: two alternating indices
                        const int idx[2] = {code >> 4, code & 15};
                        uint8_t* end = d + (size_t)n * nch;
                        if (end > line_end) goto rle4_bad;
                        int t = 0;
                        do { put(d, idx[t]); t ^= 1; } while ((d += nch) < end);
                    } else if (code > 2) { // absolute mode
                        if (d + (size_t)code * nch > line_end) goto rle4_bad;
                        const int sz = (((code + 1) >> 1) + 1) & ~1;
                        if ((size_t)sz >= srcv.size()) return false;
                        s.bytes(src, (size_t)sz);
                        for (int x = 0; x < code; x++, d += nch) put(d, (x & 1) ? src[x >> 1] & 15 : src[x >> 1] >> 4);
                    } else { // end of line (0), end of bitmap (1), delta (2): what is skipped takes palette entry 0
                        // (unlike in the RLE8 branch, end-of-bitmap only finishes the row at hand here: before the last row the decoder
                        // reads on, and a stream that really ended there fails for lack of data)
                        // A delta's vertical part is read and NOT applied: the horizontal part alone is filled (and wraps into the following
                        // rows). Both pinned by probing the reference's decoder (tests/test_bmp.py keeps the probes).
                        int x_shift3 = (int)(line_end - d);
                        if (code == 2) {
                            x_shift3 = (int)s.byte() * nch;
                            (void)s.byte();
                        }
                        d = fill(d, line_end, y, x_shift3, 0);
                        if (y >= H) break;
                    }
                }
                result = true;
            rle4_bad:;
            }
            break;
        case 8:
            if (bi.compression == 0) {
                for (int y = 0; y < H; y++, d += step) {
                    s.bytes(src, (size_t)src_pitch);
                    for (int x = 0; x < W; x++) put(d + (size_t)x * nch, src[x]);
                }
                result = true;
            } else if (bi.compression == 1) {
                uint8_t* line_end = d + width3;
                int line_end_flag = 0, y = 0;
                for (;;) {
                    int code = (int)s.word();
                    int n = code & 255;
                    code >>= 8;
                    if (n != 0) { // encoded mode
                        const int prev_y = y;
                        n *= nch;
                        if (d + n > line_end) goto rle8_bad;
                        d = fill(d, line_end, y, n, code);
                        line_end_flag = y - prev_y;
                        if (y >= H) break;
                    } else if (code > 2) { // absolute mode
                        const int prev_y = y, code3 = code * nch;
                        if (d + code3 > line_end) goto rle8_bad;
                        const int sz = (code + 1) & ~1;
                        if ((size_t)sz >= srcv.size()) return false;
                        s.bytes(src, (size_t)sz);
                        for (int x = 0; x < code; x++, d += nch) put(d, src[x]);
                        line_end_flag = y - prev_y;
                    } else {
                        int x_shift3 = (int)(line_end - d);
                        int y_shift = H - y;
                        if (code || !line_end_flag || x_shift3 < width3) {
                            if (code == 2) {
                                x_shift3 = (int)s.byte() * nch;
                                y_shift = (int)s.byte();
                            }
                            x_shift3 += (y_shift * width3) & ((code == 0) - 1);
                            if (y >= H) break;
                            d = fill(d, line_end, y, x_shift3, 0);
                            if (y >= H) break;
                        }
                        line_end_flag = 0;
                        if (y >= H) break;
                    }
                }
                result = true;
            rle8_bad:;
            }
            break;
        case 15:
        case 16:
            for (int y = 0; y < H; y++, d += step) {
                s.bytes(src, (size_t)src_pitch);
                for (int x = 0; x < W; x++) {
                    const unsigned t = src[2 * x] | ((unsigned)src[2 * x + 1] << 8);
                    const unsigned b = (t << 3) & 255u, g = bi.bpp == 15 ? (t >> 2) & ~7u & 255u : (t >> 3) & ~3u & 255u, r = bi.bpp == 15 ? (t >> 7) & ~7u & 255u : (t >> 8) & ~7u & 255u;
                    if (color) { d[3 * x] = (uint8_t)b; d[3 * x + 1] = (uint8_t)g; d[3 * x + 2] = (uint8_t)r; }
                    else d[x] = grey_of(b, g, r);
                }
            }
            result = true;
            break;
        case 24:
            for (int y = 0; y < H; y++, d += step) {
                s.bytes(src, (size_t)src_pitch);
                if (color) memcpy(d, src, (size_t)W * 3);
                else for (int x = 0; x < W; x++) d[x] = grey_of(src[3 * x], src[3 * x + 1], src[3 * x + 2]);
            }
            result = true;
            break;
        case 32:
            for (int y = 0; y < H; y++, d += step) {
                s.bytes(src, (size_t)src_pitch);
                if (!color) for (int x = 0; x < W; x++) d[x] = grey_of(src[4 * x], src[4 * x + 1], src[4 * x + 2]);
                else if (cn == 3) for (int x = 0; x < W; x++) { d[3 * x] = src[4 * x]; d[3 * x + 1] = src[4 * x + 1]; d[3 * x + 2] = src[4 * x + 2]; }
                else if (bi.shift[0] >= 0 && bi.shift[1] >= 0 && bi.shift[2] >= 0) { // bit fields with masks read from the header
                    for (int x = 0; x < W; x++) {
                        uint32_t v;
                        memcpy(&v, src + 4 * x, 4);
                        // a field is scaled to eight bits by its mask's own maximum, in single precision (found by probing the reference's
                        // decoder with arbitrary masks: (uchar)(x * (255.f / max)) reproduces it on every sample, the integer and the
                        // double-precision forms do not)
                        auto field = [&](int c) {
                            const float scale = 255.f / (float)(bi.mask[c] >> bi.shift[c]);
                            return (uint8_t)(int)((float)((bi.mask[c] & v) >> bi.shift[c]) * scale);
                        };
                        d[4 * x] = field(2);
                        d[4 * x + 1] = field(1);
                        d[4 * x + 2] = field(0);
                        d[4 * x + 3] = bi.shift[3] >= 0 ? field(3) : 255;
                    }
                } else
                    memcpy(d, src, (size_t)W * 4);
            }
            result = true;
            break;
        default: break;
        }
    } catch (const Short&) { return false; }
    catch (const std::exception&) { return false; } // an allocation inside: the decode is refused, nothing unwinds through the C ABI
    return result;
}
