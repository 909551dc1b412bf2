// lp_abi_gifenc.cpp -- the giflib_encoder_* half of the reference's giflib.hpp C ABI (/root/reference/giflib.hpp:45-50,
// implemented there by giflib.cpp:762-1306 on top of giflib 5.2.2's EGif* writer). A GIF can only be written from a GIF
// (the palettes, frame delays and extension blocks come from the decoder, giflib.go:239-256): per frame the encoder maps
// the resized BGRA frame back to palette indices and LZW-codes them.
//   * the BGRA -> index mapping runs on the device (k_gifenc_first / k_gifenc_fill / k_gifenc_map, see LpGifEncOp) on the
//     frame where the resize left it; only the one-byte-per-pixel index raster comes back to the host;
//   * the container and the LZW coder are host code restating egif_lib.c (EGifPutScreenDesc, EGifPutExtension*,
//     EGifPutImageDesc, EGifPutLine / EGifCompressLine / EGifCompressOutput / EGifBufferedOutput, EGifCloseFile): a serial
//     bit stream, byte-identical to what the reference's libgif writes (tests/test_gif.py).
#include <limits.h>
#include <stdio.h>
#include <string.h>

#include <unordered_map>

#include "lp_abi.h"
#include "lp_abi_gif.h"
#include "lp_launch.h"
#include "lp_abi_guard.h"

namespace {
const int kLzMaxCode = 4095, kFlush = 4096, kFirstCode = 4097;
enum { DISPOSAL_UNSPECIFIED = 0, DISPOSE_DO_NOT = 1 };
const int GRAPHICS_EXT = 0xF9;

int gif_bit_size(int n) // GifBitSize
{
    int i;
    for (i = 1; i <= 8; i++)
        if ((1 << i) >= n) break;
    return i;
}
}

struct giflib_encoder_struct {
    uint8_t* dst = nullptr;
    size_t dst_len = 0, off = 0;
    bool write_failed = false;
    // screen
    int swidth = 0, sheight = 0, sbackground = 0;
    LpGifColorMap global_map;
    // current frame
    std::vector<LpGifExtBlock> ext;
    LpGifColorMap frame_map;
    bool interlace = false;
    std::vector<uint8_t> pixels;
    // what the palette cache and the transparency trick remember from the previous frame
    LpGifColorMap prev_map;
    int prev_disposal = 0;
    bool have_written_first_frame = false;
    std::shared_ptr<LpDevBlock> lookup, first, fresh, prev_bgra, out_idx, palette;
    // EGif compressor state (GifFilePrivateType)
    int bits_per_pixel = 0, clear_code = 0, eof_code = 0, running_code = 0, running_bits = 0, max_code1 = 0, crnt_code = 0, shift_state = 0;
    unsigned long shift_dword = 0, pixel_count = 0;
    uint8_t buf[256];
    std::unordered_map<uint32_t, int> dict; // giflib's hash table is an exact dictionary: (prefix code << 8 | pixel) -> code

    size_t write(const uint8_t* p, size_t n) // encode_func, giflib.cpp:762-771: a write that does not fit is dropped as a whole
    {
        if (off + n > dst_len) { write_failed = true; return 0; }
        memcpy(dst + off, p, n);
        off += n;
        return n;
    }
    void put_word(int w) { const uint8_t b[2] = {(uint8_t)(w & 0xff), (uint8_t)((w >> 8) & 0xff)}; write(b, 2); }

    void buffered_output(int c) // EGifBufferedOutput
    {
        if (c == kFlush) {
            if (buf[0] != 0) write(buf, (size_t)buf[0] + 1);
            buf[0] = 0;
            write(buf, 1); // the empty block that ends the image data
        } else {
            if (buf[0] == 255) { write(buf, 256); buf[0] = 0; }
            buf[++buf[0]] = (uint8_t)c;
        }
    }
    void compress_output(int code) // EGifCompressOutput
    {
        if (code == kFlush) {
            while (shift_state > 0) {
                buffered_output((int)(shift_dword & 0xff));
                shift_dword >>= 8;
                shift_state -= 8;
            }
            shift_state = 0;
            buffered_output(kFlush);
        } else {
            shift_dword |= (unsigned long)code << shift_state;
            shift_state += running_bits;
            while (shift_state >= 8) {
                buffered_output((int)(shift_dword & 0xff));
                shift_dword >>= 8;
                shift_state -= 8;
            }
        }
        // codes above 4095 are signalling values and never widen the code size
        if (running_code >= max_code1 && code <= 4095) max_code1 = 1 << ++running_bits;
    }
    void setup_compress(int bpp) // EGifSetupCompress
    {
        if (bpp < 2) bpp = 2;
        const uint8_t b = (uint8_t)bpp;
        write(&b, 1);
        buf[0] = 0;
        bits_per_pixel = bpp;
        clear_code = 1 << bpp;
        eof_code = clear_code + 1;
        running_code = eof_code + 1;
        running_bits = bpp + 1;
        max_code1 = 1 << running_bits;
        crnt_code = kFirstCode;
        shift_state = 0;
        shift_dword = 0;
        dict.clear();
        compress_output(clear_code);
    }
    bool put_line(uint8_t* line, int len) // EGifPutLine + EGifCompressLine
    {
        if (pixel_count < (unsigned long)len) return false; // E_GIF_ERR_DATA_TOO_BIG
        pixel_count -= (unsigned long)len;
        const uint8_t mask = (uint8_t)((1 << bits_per_pixel) - 1);
        for (int i = 0; i < len; i++) line[i] &= mask; // indices that do not fit the code size are folded, as giflib does
        int i = 0, code;
        if (crnt_code == kFirstCode) code = line[i++];
        else code = crnt_code;
        while (i < len) {
            const int pixel = line[i++];
            const uint32_t key = ((uint32_t)code << 8) + (uint32_t)pixel;
            auto it = dict.find(key);
            if (it != dict.end()) code = it->second;
            else {
                compress_output(code);
                code = pixel;
                if (running_code >= kLzMaxCode) { // the table is full: tell the decoder to start over
                    compress_output(clear_code);
                    running_code = eof_code + 1;
                    running_bits = bits_per_pixel + 1;
                    max_code1 = 1 << running_bits;
                    dict.clear();
                } else
                    dict[key] = running_code++;
            }
        }
        crnt_code = code;
        if (pixel_count == 0) {
            compress_output(code);
            compress_output(eof_code);
            compress_output(kFlush);
        }
        return !write_failed;
    }
    bool write_extensions() // giflib.cpp:1107-1133
    {
        for (size_t i = 0; i < ext.size(); i++) {
            const LpGifExtBlock& ep = ext[i];
            if (ep.function != 0) { const uint8_t lead[2] = {0x21, (uint8_t)ep.function}; write(lead, 2); } // EGifPutExtensionLeader
            const uint8_t n = (uint8_t)ep.bytes.size();
            write(&n, 1);                                                                                  // EGifPutExtensionBlock
            write(ep.bytes.data(), ep.bytes.size());
            if (i + 1 == ext.size() || ext[i + 1].function != 0) { const uint8_t z = 0; write(&z, 1); }   // EGifPutExtensionTrailer
        }
        return !write_failed;
    }
};

static void frame_gcb_of(const std::vector<LpGifExtBlock>& ext, LpGifGcb* gcb, bool* ok)
{
    *gcb = LpGifGcb();
    *ok = true;
    for (const LpGifExtBlock& b : ext)
        if (b.function == GRAPHICS_EXT) *ok = LpGifReader::extension_to_gcb(b.bytes.size(), b.bytes.data(), gcb) == LP_GIF_OK;
}

extern "C" {

giflib_encoder giflib_encoder_create(void* buf, size_t buf_len) // giflib.cpp:773-797
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    if (!buf) return nullptr;
    auto e = new giflib_encoder_struct();
    e->dst = (uint8_t*)buf;
    e->dst_len = buf_len;
    memset(e->buf, 0, sizeof(e->buf));
    return e;
}
LP_ABI_CATCH("giflib_encoder_create", return nullptr)

bool giflib_encoder_init(giflib_encoder e, const giflib_decoder d, int width, int height) // giflib.cpp:800-851 + EGifPutScreenDesc
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    if (!e || !d) return false;
    const LpGifReader& g = lp_gif_reader(d);
    e->swidth = width;
    e->sheight = height;
    e->global_map = g.global_map;
    e->sbackground = (g.global_map.count && g.sbackground >= 0 && g.sbackground < g.global_map.count) ? g.sbackground : 0;
    e->write((const uint8_t*)"GIF89a", 6); // every output is written as GIF89a (EGifSetGifVersion(true))
    e->put_word(width);
    e->put_word(height);
    uint8_t b[3];
    b[0] = (uint8_t)((g.global_map.count ? 0x80 : 0x00) | ((g.scolor_resolution - 1) << 4) | (g.global_map.count ? gif_bit_size(g.global_map.count) - 1 : 0x07));
    if (g.global_map.count && g.global_sort_flag) b[0] |= 0x08;
    b[1] = (uint8_t)e->sbackground;
    b[2] = (uint8_t)g.aspect_byte;
    e->write(b, 3);
    for (int i = 0; i < g.global_map.count; i++) e->write(g.global_map.rgb[i], 3);
    return !e->write_failed;
}
LP_ABI_CATCH("giflib_encoder_init", return false)

bool giflib_encoder_encode_frame(giflib_encoder e, const giflib_decoder d, const opencv_mat frame) // giflib.cpp:1135-1214
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    if (!e || !d || !frame) return false;
    const LpGifReader& g = lp_gif_reader(d);
    auto m = static_cast<LpMat*>(const_cast<void*>((const void*)frame));
    // ---- giflib_encoder_setup_frame (giflib.cpp:853-917): this frame's interlace flag, local palette and extension blocks
    e->interlace = g.interlace;
    e->frame_map = g.local_map;
    e->ext = g.ext_blocks;
    LpGifGcb gcb;
    bool gcb_ok;
    frame_gcb_of(e->ext, &gcb, &gcb_ok);
    if (gcb_ok && gcb.transparent != -1) {
        // a transparent index that equals an opaque background colour of the GLOBAL palette is not needed: drop it
        const bool have_map = e->frame_map.count || e->global_map.count;
        if (have_map && !e->frame_map.count && gcb.transparent == e->sbackground && lp_gif_bg_alpha(d) == 255) {
            gcb.transparent = -1;
            for (LpGifExtBlock& b : e->ext)
                if (b.function == GRAPHICS_EXT && b.bytes.size() >= 4) {
                    b.bytes[0] = (uint8_t)(((gcb.disposal & 7) << 2) | (gcb.user_input ? 2 : 0));
                    b.bytes[1] = (uint8_t)(gcb.delay & 0xff);
                    b.bytes[2] = (uint8_t)((gcb.delay >> 8) & 0xff);
                    b.bytes[3] = 0xff; // EGifGCBToExtension stores NO_TRANSPARENT_COLOR (-1) as it is
                }
        }
    }
    // ---- giflib_encoder_render_frame (giflib.cpp:931-1105)
    const int fw = m->cols, fh = m->rows;
    if (fw > e->swidth || fh > e->sheight) { fprintf(stderr, "encountered error, gif frame larger than gif global size\n"); return false; }
    if (m->type != CV_8UC4 || fw != e->swidth || fh != e->sheight) {
        // the reference copies a screen-sized block out of the frame for the next frame's comparison: anything else is out of contract
        fprintf(stderr, "lilliput_hip: giflib_encoder_encode_frame needs a %dx%d CV_8UC4 matrix\n", e->swidth, e->sheight);
        return false;
    }
    const LpGifColorMap& map = e->frame_map.count ? e->frame_map : e->global_map;
    if (!map.count) { fprintf(stderr, "encountered error, gif frame has no color map\n"); return false; }
    LpEngineLease lease;
    LpEngine* eng = lease.get();
    if (!eng || !lp_mat_to_device(m, eng)) return false;
    const size_t npx = (size_t)fw * fh;
    if (!e->lookup) {
        e->lookup = lp_dev_alloc(65536); e->first = lp_dev_alloc(131072); e->fresh = lp_dev_alloc(131072);
        e->palette = lp_dev_alloc(1024); e->prev_bgra = lp_dev_alloc(npx * 4); e->out_idx = lp_dev_alloc(npx);
        if (!e->lookup || !e->first || !e->fresh || !e->palette || !e->prev_bgra || !e->out_idx) return false;
    }
    bool clear_lookup = true; // the cache survives from frame to frame only while the palette stays byte-identical
    if (e->have_written_first_frame && e->prev_map.count == map.count) clear_lookup = memcmp(e->prev_map.rgb, map.rgb, (size_t)map.count * 3) != 0;
    frame_gcb_of(e->ext, &gcb, &gcb_ok);
    hipStream_t st = eng->stream();
    if (clear_lookup && hipMemsetAsync(e->lookup->p, 0xff, 65536, st) != hipSuccess) return false;
    if (hipMemsetAsync(e->first->p, 0xff, 131072, st) != hipSuccess) return false;
    uint8_t pal[1024];
    memset(pal, 0, sizeof(pal));
    for (int i = 0; i < map.count; i++) { pal[4 * i] = map.rgb[i][0]; pal[4 * i + 1] = map.rgb[i][1]; pal[4 * i + 2] = map.rgb[i][2]; }
    if (hipMemcpyAsync(e->palette->p, pal, 1024, hipMemcpyHostToDevice, st) != hipSuccess) return false;
    LpGifEncOp op;
    memset(&op, 0, sizeof(op));
    op.frame = lp_mat_frame(m);
    op.prev_off = (uint64_t)(uintptr_t)e->prev_bgra->p;
    op.lookup_off = (uint64_t)(uintptr_t)e->lookup->p;
    op.first_off = (uint64_t)(uintptr_t)e->first->p;
    op.fresh_off = (uint64_t)(uintptr_t)e->fresh->p;
    op.palette_off = (uint64_t)(uintptr_t)e->palette->p;
    op.out_off = (uint64_t)(uintptr_t)e->out_idx->p;
    op.color_count = map.count;
    op.transparent = gcb.transparent;
    op.use_prev = e->have_written_first_frame && (e->prev_disposal == DISPOSAL_UNSPECIFIED || e->prev_disposal == DISPOSE_DO_NOT);
    lp_launch_gifenc(st, op);
    e->pixels.resize(npx);
    if (hipMemcpyAsync(e->pixels.data(), e->out_idx->p, npx, hipMemcpyDeviceToHost, st) != hipSuccess) return false;
    if (hipMemcpy2DAsync(e->prev_bgra->p, (size_t)fw * 4, (uint8_t*)m->dev->p + m->dev_off, m->dev_step, (size_t)fw * 4, (size_t)fh, hipMemcpyDeviceToDevice, st) != hipSuccess) return false;
    if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) return false;
    e->prev_map = map;
    e->prev_disposal = gcb.disposal;
    // ---- extensions, image descriptor, image data
    if (!e->write_extensions()) return false;
    {   // EGifPutImageDesc
        const uint8_t sep = 0x2c;
        e->write(&sep, 1);
        e->put_word(0); e->put_word(0); e->put_word(fw); e->put_word(fh);
        const uint8_t flags = (uint8_t)((e->frame_map.count ? 0x80 : 0) | (e->interlace ? 0x40 : 0) | (e->frame_map.count ? gif_bit_size(e->frame_map.count) - 1 : 0));
        e->write(&flags, 1);
        for (int i = 0; i < e->frame_map.count; i++) e->write(e->frame_map.rgb[i], 3);
        e->pixel_count = (unsigned long)npx;
        e->setup_compress(gif_bit_size(map.count));
    }
    if (e->interlace) {
        static const int offset[4] = {0, 4, 2, 1}, jump[4] = {8, 8, 4, 2};
        for (int i = 0; i < 4; i++)
            for (int j = offset[i]; j < fh; j += jump[i])
                if (!e->put_line(e->pixels.data() + (size_t)j * fw, fw)) { fprintf(stderr, "encountered error, could not serialize gif line\n"); return false; }
    } else
        for (int i = 0; i < fh; i++)
            if (!e->put_line(e->pixels.data() + (size_t)i * fw, fw)) return false;
    e->have_written_first_frame = true;
    return !e->write_failed;
}
LP_ABI_CATCH("giflib_encoder_encode_frame", return false)

bool giflib_encoder_flush(giflib_encoder e, const giflib_decoder d) // giflib.cpp:1216-1250 + EGifCloseFile
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    if (!e || !d) return false;
    e->ext = lp_gif_reader(d).ext_blocks; // whatever followed the last frame
    if (!e->write_extensions()) return false;
    const uint8_t term = 0x3b;
    e->write(&term, 1);
    return !e->write_failed;
}
LP_ABI_CATCH("giflib_encoder_flush", return false)

void giflib_encoder_release(giflib_encoder e) { delete e; }
int giflib_encoder_get_output_length(giflib_encoder e) { return (int)e->off; }

} // extern "C"
