// lp_bmp.h -- BMP sources of the opencv_decoder ABI: the files cv::findDecoder hands to cv::BmpDecoder in the reference
// (/root/reference/opencv.cpp:99-171 -> OpenCV 4.11 modules/imgcodecs/src/grfmt_bmp.cpp). Header walk and pixel unpacking run on the
// host -- the format is uncompressed rows or a byte-serial RLE, memcpy-class work -- and the frame enters the device with the next
// opencv_* call like any frame a host codec produced; what follows (orientation, crop, resize, encode) is the device path.
#pragma once
#include <stddef.h>
#include <stdint.h>

struct LpBmpInfo {
    int width = 0, height = 0;   // height already positive
    int bpp = 0;                 // 1, 4, 8, 15 (5-5-5), 16 (5-6-5), 24, 32
    int compression = 0;         // 0 BI_RGB, 1 BI_RLE8, 2 BI_RLE4, 3 BI_BITFIELDS
    bool bottom_up = true;
    int offset = 0;              // of the pixel data in the file
    int channels = 3;            // of the Mat cv::BmpDecoder announces: 1 (grey palette -- or ANY file with the 12-byte OS/2 header), 3, 4 (32-bit bit fields)
    uint8_t palette[256][4];     // b, g, r, reserved
    uint32_t mask[4];            // r, g, b, a of a 32-bit bit-field file with a header of 56 bytes or more
    int shift[4];                // position of each mask's lowest set bit, -1 without a mask
};
// cv::BmpDecoder::readHeader: false = the decoder refuses the file (opencv_decoder_read_header answers false)
bool lp_bmp_read_info(const uint8_t* data, size_t len, LpBmpInfo& info);
// cv::BmpDecoder::readData into rows `step` bytes apart of info.channels bytes per pixel; false = the decoder fails (short data, a run
// that leaves its row); the pixels written so far stay, as they do in the reference
bool lp_bmp_read_data(const uint8_t* data, size_t len, const LpBmpInfo& info, uint8_t* out, size_t step);
