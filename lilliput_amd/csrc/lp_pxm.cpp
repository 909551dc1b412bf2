// lp_pxm.cpp -- see lp_pxm.h. The reader restates grfmt_pxm.cpp's ReadNumber (what separates two numbers, where a comment may stand,
// the ONE byte that is consumed behind a number -- so a binary file whose samples follow the maximum value without a separator loses its
// first sample, like in the reference) over a byte stream whose read past the end is the decoder's failure (RBaseStream throws).
#include "lp_pxm.h"

#include <limits.h>
#include <string.h>

#include <vector>

namespace {
struct Eos {};
struct Bad {};
struct Stream {
    const uint8_t* d;
    size_t n, p;
    int byte()
    {
        if (p >= n) throw Eos();
        return d[p++];
    }
};
inline bool is_space(int c) { return c == ' ' || (c >= '\t' && c <= '\r'); }
inline bool is_digit(int c) { return c >= '0' && c <= '9'; }

// grfmt_pxm.cpp ReadNumber: skips white space and '#' comments (to the end of the line), anything else that is no digit is an error;
// digits up to maxdigits (0 = any number), values above INT_MAX are an error; unless maxdigits stopped it, the byte behind the number is consumed
int read_number(Stream& s, int maxdigits = 0)
{
    int code = s.byte();
    while (!is_digit(code)) {
        if (code == '#') {
            do code = s.byte();
            while (code != '\n' && code != '\r');
            code = s.byte();
        } else if (is_space(code)) {
            while (is_space(code)) code = s.byte();
        } else
            throw Bad();
    }
    long long val = 0;
    int digits = 0;
    do {
        val = val * 10 + (code - '0');
        if (val > INT_MAX) throw Bad();
        digits++;
        if (maxdigits != 0 && digits >= maxdigits) break;
        code = s.byte();
    } while (is_digit(code));
    return (int)val;
}
} // namespace

bool lp_pxm_signature(const uint8_t* d, size_t n) // PxMDecoder::checkSignature
{
    return n >= 3 && d[0] == 'P' && d[1] >= '1' && d[1] <= '6' && is_space(d[2]);
}

bool lp_pxm_read_info(const uint8_t* d, size_t n, LpPxmInfo& info)
{
    if (!lp_pxm_signature(d, n)) return false;
    Stream s{d, n, 2};
    try {
        const int code = d[1];
        info.bpp = code == '1' || code == '4' ? 1 : code == '2' || code == '5' ? 8 : 24;
        info.binary = code >= '4';
        info.channels = info.bpp > 8 ? 3 : 1;
        info.width = read_number(s);
        info.height = read_number(s);
        info.maxval = info.bpp == 1 ? 1 : read_number(s);
        if (info.maxval > 65535) return false;
        if (!(info.width > 0 && info.height > 0 && info.maxval > 0)) return false;
        info.offset = (int)s.p;
        return true;
    } catch (...) {
        return false;
    }
}

bool lp_pxm_read_data(const uint8_t* d, size_t n, const LpPxmInfo& info, uint8_t* out, size_t step)
{
    const bool wide = info.maxval > 255;                          // the decoder's 16-bit types: the 8-bit Mat gets the upper byte
    if ((long long)info.width * info.channels > INT_MAX / 2) return false; // (no caller has a Mat with such rows; keeps the int arithmetic below honest)
    const int w = info.width, nch = info.channels, width3 = w * nch;
    Stream s{d, n, (size_t)info.offset};
    uint8_t lut[256];
    if (!wide) // gray_palette: ASCII samples are scaled to 0..255 (binary ones are copied as they are); a bitmap's 1 is black
        for (int i = 0; i <= info.maxval; i++) lut[i] = (uint8_t)((i * 255 / info.maxval) ^ (info.bpp == 1 ? 255 : 0));
    try {
        std::vector<uint8_t> row((size_t)width3); // a row reaches the Mat whole: the reference converts a complete source row, so a row
                                                  // that fails half way leaves the Mat's row untouched
        if (info.bpp == 1) {
            const size_t pitch = ((size_t)w + 7) / 8;
            for (int y = 0; y < info.height; y++, out += step) {
                if (!info.binary) {
                    for (int x = 0; x < w; x++) row[(size_t)x] = lut[read_number(s, 1) != 0];
                    memcpy(out, row.data(), (size_t)w);
                } else {
                    if (s.n - s.p < pitch) { s.p = s.n; throw Eos(); }
                    const uint8_t* src = d + s.p;
                    s.p += pitch;
                    for (int x = 0; x < w; x++) out[x] = lut[(src[x >> 3] >> (7 - (x & 7))) & 1];
                }
            }
            return true;
        }
        const size_t pitch = (size_t)width3 * (wide ? 2 : 1);
        for (int y = 0; y < info.height; y++, out += step) {
            if (!info.binary) {
                for (int x = 0; x < width3; x++) {
                    int code = read_number(s);
                    if ((unsigned)code > (unsigned)info.maxval) code = info.maxval;
                    const uint8_t v = wide ? (uint8_t)(code >> 8) : lut[code];
                    if (nch == 3) row[(size_t)(x - x % 3 + (2 - x % 3))] = v; // R G B in the file, B G R in the Mat
                    else row[(size_t)x] = v;
                }
                memcpy(out, row.data(), (size_t)width3);
            } else {
                if (s.n - s.p < pitch) { s.p = s.n; throw Eos(); } // (getBytes: the row is read whole or the decoder fails)
                const uint8_t* src = d + s.p;
                s.p += pitch;
                const int sz = wide ? 2 : 1; // big-endian samples: the upper byte comes first
                if (nch == 3)
                    for (int x = 0; x < w; x++) {
                        out[3 * x + 0] = src[(3 * x + 2) * sz];
                        out[3 * x + 1] = src[(3 * x + 1) * sz];
                        out[3 * x + 2] = src[(3 * x + 0) * sz];
                    }
                else if (wide)
                    for (int x = 0; x < w; x++) out[x] = src[2 * x];
                else
                    memcpy(out, src, (size_t)w);
            }
        }
        return true;
    } catch (...) {
        return false;
    }
}
