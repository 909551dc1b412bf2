// lp_gif.cpp -- see lp_gif.h. Function by function this follows giflib 5.2.2 dgif_lib.c as the reference uses it
// (/root/reference/giflib.cpp:73-347, 621-724, 1308-1431); comments name the giflib routine and the error it would raise.
#include "lp_gif.h"

#include <string.h>

namespace {
const int kLzMaxCode = 4095; // LZ_MAX_CODE
const int kLzBits = 12;      // LZ_BITS
const int kNoSuchCode = 4098;
}

size_t LpGifReader::read(uint8_t* dst, size_t n) // the reference's decode_func (giflib.cpp:73-81): short reads at the end
{
    const size_t left = len_ - pos_, k = n < left ? n : left;
    memcpy(dst, data_ + pos_, k);
    pos_ += k;
    return k;
}

bool LpGifReader::open(const uint8_t* data, size_t len) // DGifOpen + DGifGetScreenDesc
{
    data_ = data;
    len_ = len;
    pos_ = 0;
    memset(buf_, 0, sizeof(buf_));
    memset(stack_, 0, sizeof(stack_));
    memset(suffix_, 0, sizeof(suffix_));
    memset(prefix_, 0, sizeof(prefix_));
    uint8_t b[6];
    if (read(b, 6) != 6) return false;              // D_GIF_ERR_READ_FAILED
    if (memcmp(b, "GIF", 3) != 0) return false;     // D_GIF_ERR_NOT_GIF_FILE (the version digits are not checked)
    uint8_t w[4];
    if (read(w, 2) != 2 || read(w + 2, 2) != 2) return false;
    swidth = w[0] | (w[1] << 8);
    sheight = w[2] | (w[3] << 8);
    uint8_t s[3];
    if (read(s, 3) != 3) return false;
    sbackground = s[1];
    scolor_resolution = (((s[0] & 0x70) + 1) >> 4) + 1;
    aspect_byte = s[2];
    global_sort_flag = (s[0] & 0x08) != 0;
    global_map.count = 0;
    if (s[0] & 0x80) {
        const int n = 1 << ((s[0] & 7) + 1);
        for (int i = 0; i < n; i++)
            if (read(global_map.rgb[i], 3) != 3) return false;
        global_map.count = n;
    }
    return true;
}

int LpGifReader::get_record_type(int* type) // DGifGetRecordType
{
    uint8_t c;
    if (read(&c, 1) != 1) return LP_GIF_ERROR;
    switch (c) {
    case ',': *type = LP_GIF_REC_IMAGE; return LP_GIF_OK;
    case '!': *type = LP_GIF_REC_EXTENSION; return LP_GIF_OK;
    case ';': *type = LP_GIF_REC_TERMINATE; return LP_GIF_OK;
    default: *type = LP_GIF_REC_UNDEFINED; return LP_GIF_ERROR; // D_GIF_ERR_WRONG_RECORD
    }
}

int LpGifReader::get_extension(int* function, const uint8_t** block) // DGifGetExtension
{
    uint8_t c;
    if (read(&c, 1) != 1) return LP_GIF_ERROR;
    *function = c;
    return get_extension_next(block);
}

int LpGifReader::get_extension_next(const uint8_t** block) // DGifGetExtensionNext
{
    uint8_t c;
    if (read(&c, 1) != 1) return LP_GIF_ERROR;
    if (c > 0) {
        buf_[0] = c;
        if (read(buf_ + 1, c) != c) return LP_GIF_ERROR;
        *block = buf_;
    } else
        *block = nullptr;
    return LP_GIF_OK;
}

int LpGifReader::extension_to_gcb(size_t len, const uint8_t* bytes, LpGifGcb* gcb) // DGifExtensionToGCB
{
    if (len != 4) return LP_GIF_ERROR;
    gcb->disposal = (bytes[0] >> 2) & 7;
    gcb->user_input = (bytes[0] & 2) != 0;
    gcb->delay = bytes[1] | (bytes[2] << 8);
    gcb->transparent = (bytes[0] & 1) ? bytes[3] : -1;
    return LP_GIF_OK;
}

int LpGifReader::get_image_header() // DGifGetImageHeader
{
    uint8_t w[8];
    for (int i = 0; i < 4; i++)
        if (read(w + 2 * i, 2) != 2) return LP_GIF_ERROR;
    left = w[0] | (w[1] << 8);
    top = w[2] | (w[3] << 8);
    width = w[4] | (w[5] << 8);
    height = w[6] | (w[7] << 8);
    uint8_t c;
    if (read(&c, 1) != 1) { local_map.count = 0; return LP_GIF_ERROR; }
    interlace = (c & 0x40) != 0;
    local_map.count = 0;
    if (c & 0x80) {
        const int n = 1 << ((c & 7) + 1);
        for (int i = 0; i < n; i++)
            if (read(local_map.rgb[i], 3) != 3) return LP_GIF_ERROR;
        local_map.count = n;
    }
    pixel_count_ = (unsigned long)((long)width * (long)height);
    return setup_decompress();
}

int LpGifReader::setup_decompress() // DGifSetupDecompress
{
    uint8_t code_size;
    if (read(&code_size, 1) < 1) return LP_GIF_ERROR;
    if (code_size > 8) return LP_GIF_ERROR; // "can only happen on a severely malformed GIF"
    buf_[0] = 0;
    bits_per_pixel_ = code_size;
    clear_code_ = 1 << code_size;
    eof_code_ = clear_code_ + 1;
    running_code_ = eof_code_ + 1;
    running_bits_ = code_size + 1;
    max_code1_ = 1 << running_bits_;
    stack_ptr_ = 0;
    last_code_ = kNoSuchCode;
    shift_state_ = 0;
    shift_dword_ = 0;
    for (int i = 0; i <= kLzMaxCode; i++) prefix_[i] = kNoSuchCode;
    return LP_GIF_OK;
}

int LpGifReader::get_code_next(const uint8_t** block) // DGifGetCodeNext
{
    uint8_t c;
    if (read(&c, 1) != 1) return LP_GIF_ERROR;
    if (c > 0) {
        buf_[0] = c;
        if (read(buf_ + 1, c) != c) return LP_GIF_ERROR;
        *block = buf_;
    } else {
        *block = nullptr;
        buf_[0] = 0;
        pixel_count_ = 0;
    }
    return LP_GIF_OK;
}

int LpGifReader::get_line(uint8_t* line, int len) // DGifGetLine
{
    if (!len) len = width;
    pixel_count_ -= (unsigned long)(long)len;
    if (pixel_count_ > 0xffff0000UL) return LP_GIF_ERROR; // D_GIF_ERR_DATA_TOO_BIG
    if (decompress_line(line, len) != LP_GIF_OK) return LP_GIF_ERROR;
    if (pixel_count_ == 0) {
        // the image is complete: swallow the rest of its data sub-blocks up to the terminator
        const uint8_t* dummy;
        do {
            if (get_code_next(&dummy) == LP_GIF_ERROR) return LP_GIF_ERROR;
        } while (dummy != nullptr);
    }
    return LP_GIF_OK;
}

int LpGifReader::buffered_input(uint8_t* next) // DGifBufferedInput
{
    if (buf_[0] == 0) {
        if (read(buf_, 1) != 1) return LP_GIF_ERROR;
        if (buf_[0] == 0) return LP_GIF_ERROR; // D_GIF_ERR_IMAGE_DEFECT: the terminator arrived before the LZW end code
        if (read(buf_ + 1, buf_[0]) != buf_[0]) return LP_GIF_ERROR;
        *next = buf_[1];
        buf_[1] = 2; // position of the next unread byte
        buf_[0]--;
    } else {
        *next = buf_[buf_[1]++];
        buf_[0]--;
    }
    return LP_GIF_OK;
}

int LpGifReader::decompress_input(int* code) // DGifDecompressInput
{
    static const unsigned short masks[] = {0x0000, 0x0001, 0x0003, 0x0007, 0x000f, 0x001f, 0x003f, 0x007f, 0x00ff, 0x01ff, 0x03ff, 0x07ff, 0x0fff};
    if (running_bits_ > kLzBits) return LP_GIF_ERROR; // D_GIF_ERR_IMAGE_DEFECT
    while (shift_state_ < running_bits_) {
        uint8_t b;
        if (buffered_input(&b) == LP_GIF_ERROR) return LP_GIF_ERROR;
        shift_dword_ |= (unsigned long)b << shift_state_;
        shift_state_ += 8;
    }
    *code = (int)(shift_dword_ & masks[running_bits_]);
    shift_dword_ >>= running_bits_;
    shift_state_ -= running_bits_;
    // codes above 4095 are signalling values: at 12 bits the table is simply kept as it is
    if (running_code_ < kLzMaxCode + 2 && ++running_code_ > max_code1_ && running_bits_ < kLzBits) {
        max_code1_ <<= 1;
        running_bits_++;
    }
    return LP_GIF_OK;
}

int LpGifReader::prefix_char(int code, int clear) const // DGifGetPrefixChar
{
    int i = 0;
    while (code > clear && i++ <= kLzMaxCode) {
        if (code > kLzMaxCode) return kNoSuchCode;
        code = prefix_[code];
    }
    return code;
}

int LpGifReader::decompress_line(uint8_t* line, int len) // DGifDecompressLine
{
    int i = 0, code, prefix;
    int sp = stack_ptr_, last = last_code_;
    const int eof = eof_code_, clear = clear_code_;
    if (sp > kLzMaxCode) return LP_GIF_ERROR;
    while (sp != 0 && i < len) line[i++] = stack_[--sp]; // pixels left over from the previous call
    while (i < len) {
        if (decompress_input(&code) == LP_GIF_ERROR) return LP_GIF_ERROR;
        if (code == eof) return LP_GIF_ERROR; // D_GIF_ERR_EOF_TOO_SOON
        if (code == clear) {
            for (int j = 0; j <= kLzMaxCode; j++) prefix_[j] = kNoSuchCode;
            running_code_ = eof_code_ + 1;
            running_bits_ = bits_per_pixel_ + 1;
            max_code1_ = 1 << running_bits_;
            last = last_code_ = kNoSuchCode;
            continue;
        }
        if (code < clear) {
            line[i++] = (uint8_t)code;
        } else {
            if (prefix_[code] == kNoSuchCode) {
                prefix = last;
                // only legal when the code is the one about to be defined (KwKwK); either way giflib pushes a first character
                const int c = prefix_char(code == running_code_ - 2 ? last : code, clear);
                suffix_[running_code_ - 2] = stack_[sp++] = (uint8_t)c;
            } else
                prefix = code;
            // a defective image could loop forever here: the stack depth bounds the walk
            while (sp < kLzMaxCode && prefix > clear && prefix <= kLzMaxCode) {
                stack_[sp++] = suffix_[prefix];
                prefix = prefix_[prefix];
            }
            if (sp >= kLzMaxCode || prefix > kLzMaxCode) return LP_GIF_ERROR; // D_GIF_ERR_IMAGE_DEFECT
            stack_[sp++] = (uint8_t)prefix;
            while (sp != 0 && i < len) line[i++] = stack_[--sp];
        }
        if (last != kNoSuchCode && running_code_ - 2 < kLzMaxCode + 1 && prefix_[running_code_ - 2] == kNoSuchCode) {
            prefix_[running_code_ - 2] = last;
            suffix_[running_code_ - 2] = (uint8_t)prefix_char(code == running_code_ - 2 ? last : code, clear);
        }
        last = code;
    }
    last_code_ = last;
    stack_ptr_ = sp;
    return LP_GIF_OK;
}
