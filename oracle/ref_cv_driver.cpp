// oracle/ref_cv_driver.cpp -- TEST INFRASTRUCTURE ONLY. Calls OpenCV 4.11's own CMYK -> BGR row conversion
// (modules/imgcodecs/src/utils.cpp icvCvt_CMYK2BGR_8u_C4C3R, what cv::JpegDecoder::readData applies to four-component JPEGs) out of
// the reference's prebuilt libopencv_imgcodecs.a: the object file is extracted where the archive lies and linked as it is.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <string>

namespace cv {
template <class T> struct Size_ { T width, height; };
void icvCvt_CMYK2BGR_8u_C4C3R(const unsigned char* cmyk, int cmyk_step, unsigned char* bgr, int bgr_step, Size_<int> size);
void error(int, const std::string&, const char*, const char*, int) { abort(); } // the only core symbol utils.cpp.o refers to
} // namespace cv

extern "C" void ref_cv_cmyk2bgr(const uint8_t* cmyk, uint8_t* bgr, int n)
{
    cv::Size_<int> sz;
    sz.width = n;
    sz.height = 1;
    cv::icvCvt_CMYK2BGR_8u_C4C3R(cmyk, 0, bgr, 0, sz);
}
