// lp_png.h -- host-side PNG container walk shared by the colour-metadata readers (lp_abi_meta.cpp) and the PNG decoder
// (lp_abi_opencv.cpp): what libpng 1.6.47's png_read_info keeps, and what it rejects, up to the first IDAT.
// The reference reaches libpng through OpenCV's PngDecoder (/root/reference/opencv.cpp:99-171) and directly
// (/root/reference/opencv.cpp:314-395); libpng's source is not in the reference tree, the behaviour restated here is probed
// against the reference's prebuilt libpng16.a in tests/test_meta.py and tests/test_png.py.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <memory>
#include <new>
#include <type_traits>
#include <utility>
#include <vector>

struct LpPngInfo {
    // IHDR
    uint32_t width = 0, height = 0;
    int depth = 0, color_type = 0, interlace = 0;
    // PLTE / tRNS as libpng keeps them
    int num_palette = 0;
    uint8_t palette[256][3];
    int num_trans = 0;             // palette images: alpha entries; grey / RGB: 1 when a colour key is present
    uint8_t trans_alpha[256];
    uint16_t trans_key[3] = {0, 0, 0}; // grey in [0]; R, G, B
    // colour signalling
    bool have_cicp = false;
    uint8_t cicp[4] = {0, 0, 0, 0};
    std::vector<uint8_t> icc;
    size_t idat_off = 0;           // offset of the first IDAT chunk's length field
};

// png_read_info: false = libpng would have raised an error before reaching the image data.
bool lp_png_read_info(const uint8_t* s, size_t n, LpPngInfo& out);

// A byte buffer whose resize() does not clear what it adds (std::vector<uint8_t>::resize value-initialises: a memset of the whole image
// in front of an inflate that writes every byte of it).
template <class T>
struct LpDefaultInit : std::allocator<T> {
    template <class U> struct rebind { typedef LpDefaultInit<U> other; };
    using std::allocator<T>::allocator;
    template <class U> void construct(U* p) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new (static_cast<void*>(p)) U; }
    template <class U, class... A> void construct(U* p, A&&... a) { ::new (static_cast<void*>(p)) U(std::forward<A>(a)...); }
};
typedef std::vector<uint8_t, LpDefaultInit<uint8_t>> LpBytes;

// Image data: concatenates the IDAT run that starts at info.idat_off, inflates exactly the bytes the image needs into `filtered`
// (per Adam7 pass / per row: one filter-type byte + the packed row), then walks the chunks up to IEND like png_read_end.
// false = libpng would have failed (CRC error in an IDAT / critical chunk, broken or short zlib stream, truncated file, missing IEND).
bool lp_png_read_idat(const uint8_t* s, size_t n, const LpPngInfo& info, LpBytes& filtered);

// Bytes of filtered data the image needs, and the geometry of Adam7 pass p (0..6): pass_w/pass_h may be 0.
size_t lp_png_filtered_size(const LpPngInfo& info);
void lp_png_pass_geometry(const LpPngInfo& info, int pass, uint32_t* pw, uint32_t* ph, uint32_t* x0, uint32_t* y0, uint32_t* dx, uint32_t* dy);
inline int lp_png_channels_in_file(int color_type) { return color_type == 0 || color_type == 3 ? 1 : color_type == 4 ? 2 : color_type == 2 ? 3 : 4; }
// which inflater lp_png_read_idat tries first on an ordinary stream: 1 = lp_inflate.cpp (default), 0 = zlib only; returns the previous setting
int lp_png_set_inflater(int own);
