// oracle/ref_pxm_driver.cpp -- TEST INFRASTRUCTURE ONLY. Runs OpenCV 4.11's own PBM / PGM / PPM decoder (modules/imgcodecs/src/grfmt_pxm.cpp:
// the class cv::findDecoder hands a "P1".."P6" buffer to in the reference, opencv.cpp:99-171) out of the reference's prebuilt
// libopencv_imgcodecs.a -- grfmt_pxm.cpp.o, grfmt_base.cpp.o, bitstrm.cpp.o, utils.cpp.o extracted where the archive lies and linked as
// they are; the core symbols they need come from ref_cvstubs.h, the class's members are called through their mangled names (as for
// cv::BmpDecoder in ref_bmp_driver.cpp).
#include "ref_cvstubs.h"

#include <stdarg.h>

namespace cv {
enum ExifTagName { EXIF_NONE };
struct ExifEntry_t { char pad[256]; };
class ExifReader {
public:
    ExifReader();
    ~ExifReader();
    ExifEntry_t getTag(const ExifTagName tag) const;
};
ExifReader::ExifReader() {}
ExifReader::~ExifReader() {}
ExifEntry_t ExifReader::getTag(const ExifTagName) const { abort(); }
String format(const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    return String(buf);
}
} // namespace cv

extern "C" {
void _ZN2cv10PxMDecoderC1Ev(void* self);
void _ZN2cv10PxMDecoderD1Ev(void* self);
bool _ZN2cv10PxMDecoder10readHeaderEv(void* self);
bool _ZN2cv10PxMDecoder8readDataERNS_3MatE(void* self, cv::Mat* img);
bool _ZN2cv12ImageDecoder4Impl9setSourceERKNS_3MatE(void* self, const cv::Mat* buf);
bool _ZNK2cv10PxMDecoder14checkSignatureERKNSt7__cxx1112basic_stringIcSt11char_traitsIcESaIcEEE(const void* self, const std::string* signature);
static int field(const void* self, int i) { int v; memcpy(&v, (const char*)self + 8 + 4 * i, 4); return v; }

// 0: decoded; *type = the decoder's own type (CV_8UC1 / CV_8UC3 / CV_16UC1 / CV_16UC3), out = h x w x channels BYTES: the pixels as
// opencv_decoder_read_data returns them into the Mat the Go layer hands it -- the decoder's channels at 8-bit depth (opencv.go:250-267
// demotes 16-bit types). 1: readHeader refused the file; 2: readData failed; -1: cap too small
int ref_pxm_decode(const uint8_t* data, size_t len, int* w, int* h, int* type, uint8_t* out, size_t cap)
{
    alignas(64) static thread_local unsigned char obj[16384];
    memset(obj, 0, sizeof(obj));
    _ZN2cv10PxMDecoderC1Ev(obj);
    int rc = 0;
    try {
        cv::Mat buf(1, (int)len, CV_8U, const_cast<uint8_t*>(data));
        try {
            // cv::findDecoder(buf) (loadsave.cpp): the decoder sees a file only after its checkSignature accepted the first bytes
            const std::string signature((const char*)data, len < 64 ? len : 64);
            if (!_ZNK2cv10PxMDecoder14checkSignatureERKNSt7__cxx1112basic_stringIcSt11char_traitsIcESaIcEEE(obj, &signature) ||
                !_ZN2cv12ImageDecoder4Impl9setSourceERKNS_3MatE(obj, &buf) || !_ZN2cv10PxMDecoder10readHeaderEv(obj)) rc = 1;
        } catch (...) { rc = 1; }
        if (!rc) {
            *w = field(obj, 0); *h = field(obj, 1); *type = field(obj, 2);
            const int cn = CV_MAT_CN(*type);
            const size_t need = (size_t)*w * *h * cn;
            if (*w <= 0 || *h <= 0 || need > cap) rc = -1;
            else {
                cv::Mat img(*h, *w, CV_MAKETYPE(CV_8U, cn), out);
                if (!_ZN2cv10PxMDecoder8readDataERNS_3MatE(obj, &img)) rc = 2;
            }
        }
    } catch (...) { rc = rc ? rc : 2; }
    try { _ZN2cv10PxMDecoderD1Ev(obj); } catch (...) {}
    return rc;
}

std::string _ZNK2cv12ImageDecoder4Impl14getDescriptionB5cxx11Ev(const void* self);
// what opencv_decoder_get_description hands the Go layer for such a file (opencv.cpp:110-118)
int ref_pxm_description(char* out, size_t cap)
{
    alignas(64) static thread_local unsigned char obj[16384];
    memset(obj, 0, sizeof(obj));
    _ZN2cv10PxMDecoderC1Ev(obj);
    const std::string d = _ZNK2cv12ImageDecoder4Impl14getDescriptionB5cxx11Ev(obj);
    snprintf(out, cap, "%s", d.c_str());
    _ZN2cv10PxMDecoderD1Ev(obj);
    return (int)d.size();
}
}
