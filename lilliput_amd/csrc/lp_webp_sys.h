// lp_webp_sys.h -- the part of libwebp's public C API (webp/decode.h, webp/encode.h) this library calls, declared here because the
// build image carries the shared library (libwebp.so.7 = libwebp 1.2.2) without its development headers.
//
// Role: the VP8 / VP8L bitstreams of a WebP file are serial entropy-coded data (boolean arithmetic coder, Huffman + LZ77); like
// DEFLATE for PNG (zlib) and LZW for GIF they are coded on the host -- SURVEY.md section 7 -- and everything around them is this
// library's own: the RIFF container (lp_webp.cpp), frame compositing, crop, resize (device). The reference reaches the same
// functions through /root/reference/webp.cpp:302-362 (WebPDecodeBGR(A)Into) and :707-751 (WebPEncodeBGR(A) / Lossless).
//
// Layouts follow the documented public structs; WebP*Internal take the ABI version the caller was written against (major part checked
// by the library: 0x02xx).
#pragma once
#include <stddef.h>
#include <stdint.h>

extern "C" {

#define LP_WEBP_DECODER_ABI 0x0209
#define LP_WEBP_ENCODER_ABI 0x020f

struct WebPBitstreamFeatures {
    int width, height;
    int has_alpha;
    int has_animation;
    int format;             // 0 undefined / mixed, 1 lossy, 2 lossless
    uint32_t pad[5];
};
int WebPGetFeaturesInternal(const uint8_t* data, size_t size, WebPBitstreamFeatures* features, int abi_version); // 0 = VP8_STATUS_OK
uint8_t* WebPDecodeBGRInto(const uint8_t* data, size_t size, uint8_t* out, size_t out_size, int out_stride);
uint8_t* WebPDecodeBGRAInto(const uint8_t* data, size_t size, uint8_t* out, size_t out_size, int out_stride);
int WebPGetDecoderVersion(void);

size_t WebPEncodeBGR(const uint8_t* bgr, int width, int height, int stride, float quality, uint8_t** output);
size_t WebPEncodeBGRA(const uint8_t* bgra, int width, int height, int stride, float quality, uint8_t** output);
size_t WebPEncodeLosslessBGR(const uint8_t* bgr, int width, int height, int stride, uint8_t** output);
size_t WebPEncodeLosslessBGRA(const uint8_t* bgra, int width, int height, int stride, uint8_t** output);
void WebPFree(void* ptr);

struct WebPConfig {
    int lossless;
    float quality;
    int method;
    int image_hint;
    int target_size;
    float target_PSNR;
    int segments, sns_strength, filter_strength, filter_sharpness, filter_type, autofilter;
    int alpha_compression, alpha_filtering, alpha_quality;
    int pass;
    int show_compressed, preprocessing, partitions, partition_limit, emulate_jpeg_size, thread_level, low_memory;
    int near_lossless, exact;
    int use_delta_palette, use_sharp_yuv;
    int qmin, qmax;
};
int WebPConfigInitInternal(WebPConfig* config, int preset /* 0 = WEBP_PRESET_DEFAULT */, float quality, int abi_version);
int WebPValidateConfig(const WebPConfig* config);

struct WebPPicture;
typedef int (*WebPWriterFunction)(const uint8_t* data, size_t data_size, const WebPPicture* picture);
struct WebPPicture {
    int use_argb;
    int colorspace;
    int width, height;
    uint8_t *y, *u, *v;
    int y_stride, uv_stride;
    uint8_t* a;
    int a_stride;
    uint32_t pad1[2];
    uint32_t* argb;
    int argb_stride;
    uint32_t pad2[3];
    WebPWriterFunction writer;
    void* custom_ptr;
    int extra_info_type;
    uint8_t* extra_info;
    void* stats;
    int error_code;
    int (*progress_hook)(int percent, const WebPPicture* picture);
    void* user_data;
    uint32_t pad3[3];
    uint8_t *pad4, *pad5;
    uint32_t pad6[8];
    void* memory_;
    void* memory_argb_;
    void* pad7[2];
};
int WebPPictureInitInternal(WebPPicture* picture, int abi_version);
int WebPPictureImportBGR(WebPPicture* picture, const uint8_t* bgr, int stride);
int WebPPictureImportBGRA(WebPPicture* picture, const uint8_t* bgra, int stride);
void WebPPictureFree(WebPPicture* picture);
int WebPEncode(const WebPConfig* config, WebPPicture* picture);

struct WebPMemoryWriter {
    uint8_t* mem;
    size_t size, max_size;
    uint32_t pad[1];
};
void WebPMemoryWriterInit(WebPMemoryWriter* writer);
int WebPMemoryWrite(const uint8_t* data, size_t data_size, const WebPPicture* picture);
void WebPMemoryWriterClear(WebPMemoryWriter* writer);

} // extern "C"
