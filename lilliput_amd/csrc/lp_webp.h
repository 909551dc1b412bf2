// lp_webp.h -- the WebP container (RIFF / VP8X / ANIM / ANMF / ALPH / ICCP) as the reference uses it through libwebpmux
// (/root/reference/webp.cpp:61-134 webp_decoder_create: WebPMuxCreate, WebPMuxGetFeatures, WebPMuxGetFrame, WebPMuxGetCanvasSize,
// WebPMuxGetAnimationParams; :501-577 the ICCP mux on output). libwebpmux is not part of the build image and a container walk is no
// codec work, so it is written here: reader (frame list with offsets / duration / dispose / blend, the standalone bitstream
// WebPMuxGetFrame hands out per frame) and writer (still image with ICCP, animation). The VP8 / VP8L payloads themselves go to
// libwebp (lp_webp_sys.h).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <vector>

enum { LP_WEBP_FLAG_ANIM = 0x02, LP_WEBP_FLAG_XMP = 0x04, LP_WEBP_FLAG_EXIF = 0x08, LP_WEBP_FLAG_ALPHA = 0x10, LP_WEBP_FLAG_ICCP = 0x20 };

struct LpWebpFrame {
    int x_offset = 0, y_offset = 0;     // ANMF: 2 x the stored 24-bit values
    int width = 0, height = 0;          // of the frame's own bitstream
    int duration = 1;                   // ms; libwebpmux reports 1 for the image of a non-animated file
    int dispose = 0;                    // WebPMuxAnimDispose: 0 none, 1 background
    int blend = 0;                      // WebPMuxAnimBlend: 0 blend, 1 no blend
    const uint8_t* alph = nullptr;      // ALPH payload (lossy image with an alpha plane)
    size_t alph_size = 0;
    const uint8_t* img = nullptr;       // VP8 / VP8L payload
    size_t img_size = 0;
    bool lossless = false;              // VP8L
    bool has_alpha = false;             // an ALPH chunk, or the alpha bit of a VP8L header
};

struct LpWebpFile {
    bool has_vp8x = false;
    uint32_t flags = 0;                 // VP8X flags, or what WebPMuxGetFeatures derives for a file without VP8X
    int canvas_w = 0, canvas_h = 0;
    bool has_anim_chunk = false;
    uint32_t bgcolor = 0, loop_count = 0;
    const uint8_t* icc = nullptr;
    size_t icc_size = 0;
    std::vector<LpWebpFrame> frames;
};

// WebPMuxCreate + MuxValidate: false = the reference's webp_decoder_create would have returned NULL for this buffer.
bool lp_webp_parse(const uint8_t* data, size_t len, LpWebpFile* out);
// The self-contained WebP file of one frame (SynthesizeBitstream): RIFF [+ VP8X + ALPH] + VP8 / VP8L.
void lp_webp_frame_bitstream(const LpWebpFrame& f, std::vector<uint8_t>& out);

// ---- writer
struct LpWebpEncodedImage {            // the chunks of one encoded image, cut out of the encoder's own RIFF output
    std::vector<uint8_t> alph, img;
    bool lossless = false, has_alpha = false;
    int width = 0, height = 0;
};
bool lp_webp_split_encoded(const uint8_t* riff, size_t len, LpWebpEncodedImage* out);
// Still image: the bare RIFF when nothing needs a VP8X chunk, else VP8X [+ ICCP] [+ ALPH] + image (WebPMuxSetImage + WebPMuxSetChunk("ICCP") + WebPMuxAssemble).
void lp_webp_write_still(const LpWebpEncodedImage& im, const uint8_t* icc, size_t icc_len, std::vector<uint8_t>& out);
struct LpWebpAnimFrame { LpWebpEncodedImage im; int x_offset, y_offset, duration, dispose, blend; };
void lp_webp_write_animation(int canvas_w, int canvas_h, uint32_t bgcolor, uint32_t loop_count, const std::vector<LpWebpAnimFrame>& frames, const uint8_t* icc,
                             size_t icc_len, std::vector<uint8_t>& out);
