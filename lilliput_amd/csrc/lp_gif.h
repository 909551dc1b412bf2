// lp_gif.h -- host-side GIF container + LZW reader behind the giflib_decoder_* ABI.
// The reference drives giflib 5.2.2 (DGifOpen / DGifGetRecordType / DGifGetExtension[Next] / DGifGetImageHeader /
// DGifGetLine / DGifGetCodeNext, /root/reference/giflib.cpp:73-347, 621-724); giflib's source is not part of the reference
// tree (prebuilt deps/linux/amd64/lib/libgif.a), so this restates the published behaviour of dgif_lib.c call by call,
// including what it reports as an error and what it silently tolerates. tests/test_gif.py compares it with the real library.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <vector>

enum { LP_GIF_OK = 1, LP_GIF_ERROR = 0 };
enum { LP_GIF_REC_UNDEFINED = 0, LP_GIF_REC_IMAGE = 2, LP_GIF_REC_EXTENSION = 3, LP_GIF_REC_TERMINATE = 4 };

struct LpGifColorMap {
    int count = 0; // 0 = absent
    uint8_t rgb[256][3];
};

struct LpGifExtBlock {
    int function; // 0 = continuation
    std::vector<uint8_t> bytes;
};

struct LpGifGcb {
    int disposal = 0;
    bool user_input = false;
    int delay = 0;
    int transparent = -1;
};

class LpGifReader {
  public:
    // DGifOpen: signature, logical screen descriptor, global colour map
    bool open(const uint8_t* data, size_t len);
    int swidth = 0, sheight = 0, sbackground = 0;
    int scolor_resolution = 0, aspect_byte = 0; // SColorResolution, AspectByte: carried over by the encoder
    bool global_sort_flag = false;              // SColorMap->SortFlag
    LpGifColorMap global_map;
    // current image descriptor (DGifGetImageHeader)
    int left = 0, top = 0, width = 0, height = 0;
    bool interlace = false;
    LpGifColorMap local_map;
    std::vector<LpGifExtBlock> ext_blocks; // gif->ExtensionBlocks

    int get_record_type(int* type);                 // DGifGetRecordType
    int get_extension(int* function, const uint8_t** block); // DGifGetExtension: *block = NULL or [len, bytes...]
    int get_extension_next(const uint8_t** block);  // DGifGetExtensionNext
    int get_image_header();                         // DGifGetImageHeader
    int get_line(uint8_t* line, int len);           // DGifGetLine
    int get_code_next(const uint8_t** block);       // DGifGetCodeNext
    static int extension_to_gcb(size_t len, const uint8_t* bytes, LpGifGcb* gcb); // DGifExtensionToGCB

  private:
    size_t read(uint8_t* dst, size_t n);
    int setup_decompress();
    int decompress_line(uint8_t* line, int len);
    int decompress_input(int* code);
    int buffered_input(uint8_t* next);
    int prefix_char(int code, int clear) const;

    const uint8_t* data_ = nullptr;
    size_t len_ = 0, pos_ = 0;
    // GifFilePrivateType
    int bits_per_pixel_ = 0, clear_code_ = 0, eof_code_ = 0, running_code_ = 0, running_bits_ = 0, max_code1_ = 0, last_code_ = 0, stack_ptr_ = 0,
        shift_state_ = 0;
    unsigned long shift_dword_ = 0;
    unsigned long pixel_count_ = 0;
    uint8_t buf_[256];
    uint8_t stack_[4096];
    uint8_t suffix_[4096];
    int prefix_[4096];
};
