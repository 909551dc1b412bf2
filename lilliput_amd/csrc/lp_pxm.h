// lp_pxm.h -- PBM / PGM / PPM sources ("P1".."P6") of the opencv_decoder ABI: the files cv::findDecoder hands to cv::PxMDecoder in the
// reference (/root/reference/opencv.cpp:99-171 -> OpenCV 4.11 modules/imgcodecs/src/grfmt_pxm.cpp). Like BMP (lp_bmp.h): header walk
// and sample unpacking on the host -- raw or ASCII samples, memcpy-class work -- and the frame enters the device with the next opencv_*
// call; orientation, crop, resize and encode are the device path. Pinned against the reference's own decoder object code
// (oracle/ref_pxm_driver.cpp, tests/test_pxm.py).
#pragma once
#include <stddef.h>
#include <stdint.h>

struct LpPxmInfo {
    int width = 0, height = 0;
    int bpp = 0;          // 1 (P1 / P4), 8 (P2 / P5), 24 (P3 / P6)
    bool binary = false;  // P4 / P5 / P6
    int maxval = 1;       // 1 for bitmaps; samples above 255 make the decoder's type 16-bit (the Go layer demotes it: opencv.go:250-267)
    int channels = 1;     // 1 or 3
    int offset = 0;       // of the first sample in the file
};
// cv::findDecoder's signature test for this decoder + cv::PxMDecoder::readHeader: false = not such a file / the decoder refuses it
bool lp_pxm_signature(const uint8_t* data, size_t len);
bool lp_pxm_read_info(const uint8_t* data, size_t len, LpPxmInfo& info);
// cv::PxMDecoder::readData into an 8-bit Mat of info.channels channels, rows `step` bytes apart (BGR for P3 / P6; 16-bit samples give
// their upper byte); false = the decoder fails (the file ends early, a character that is no digit): the rows written so far stay
bool lp_pxm_read_data(const uint8_t* data, size_t len, const LpPxmInfo& info, uint8_t* out, size_t step);
