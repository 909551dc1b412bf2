// oracle/ref_bmp_driver.cpp -- TEST INFRASTRUCTURE ONLY. Runs OpenCV 4.11's own BMP decoder (modules/imgcodecs/src/grfmt_bmp.cpp, the
// class cv::findDecoder hands a "BM" buffer to in the reference: opencv.cpp:99-171) out of the reference's prebuilt
// libopencv_imgcodecs.a: grfmt_bmp.cpp.o, grfmt_base.cpp.o, bitstrm.cpp.o and utils.cpp.o are extracted where the archive lies and
// linked as they are. libopencv_core.a is not part of the reference's shipped deps, so the handful of core symbols those objects
// refer to are defined here in the smallest form that serves a 1 x N byte buffer and a caller-allocated image; the decoder's own
// class definition is private to the OpenCV sources, so its member functions are called through their mangled names.
#include "ref_cvstubs.h"

namespace cv {
// ---- what grfmt_base.cpp.o drags in and a BMP never reaches
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
} // namespace cv

extern "C" {
void _ZN2cv10BmpDecoderC1Ev(void* self);
void _ZN2cv10BmpDecoderD1Ev(void* self);
bool _ZN2cv10BmpDecoder10readHeaderEv(void* self);
bool _ZN2cv10BmpDecoder8readDataERNS_3MatE(void* self, cv::Mat* img);
bool _ZN2cv12ImageDecoder4Impl9setSourceERKNS_3MatE(void* self, const cv::Mat* buf);

// cv::ImageDecoder::Impl::getDescription() const, returning std::string by value
std::string _ZNK2cv12ImageDecoder4Impl14getDescriptionB5cxx11Ev(const void* self);
// header fields sit behind the vtable pointer: int m_width, m_height, m_type (grfmt_base.hpp)
static int field(const void* self, int i) { int v; memcpy(&v, (const char*)self + 8 + 4 * i, 4); return v; }

// 0: decoded, out = h x w x channels of *type (CV_8UC1 / C3 / C4), the way opencv_decoder_read_data fills a Mat of the decoder's own
// type; 1: readHeader refused the file; 2: readData failed; -1: cap too small
int ref_bmp_decode(const uint8_t* data, size_t len, int* w, int* h, int* type, uint8_t* out, size_t cap)
{
    alignas(64) static thread_local unsigned char obj[16384];
    memset(obj, 0, sizeof(obj));
    _ZN2cv10BmpDecoderC1Ev(obj);
    int rc = 0;
    try {
        cv::Mat buf(1, (int)len, CV_8U, const_cast<uint8_t*>(data));
        try { // opencv_decoder_read_header catches what readHeader throws and answers false (opencv.cpp:127-140)
            if (!_ZN2cv12ImageDecoder4Impl9setSourceERKNS_3MatE(obj, &buf) || !_ZN2cv10BmpDecoder10readHeaderEv(obj)) rc = 1;
        } catch (...) { rc = 1; }
        if (!rc) {
            *w = field(obj, 0); *h = field(obj, 1); *type = field(obj, 2);
            const size_t need = (size_t)*w * *h * CV_MAT_CN(*type);
            if (*w <= 0 || *h <= 0 || need > cap) rc = -1;
            else {
                cv::Mat img(*h, *w, *type, out);
                if (!_ZN2cv10BmpDecoder8readDataERNS_3MatE(obj, &img)) rc = 2;
            }
        }
    } catch (...) { rc = rc ? rc : 2; }
    try { _ZN2cv10BmpDecoderD1Ev(obj); } catch (...) {}
    return rc;
}

// what opencv_decoder_get_description hands the Go layer for a BMP (opencv.cpp:110-118)
int ref_bmp_description(char* out, size_t cap)
{
    alignas(64) static thread_local unsigned char obj[16384];
    memset(obj, 0, sizeof(obj));
    _ZN2cv10BmpDecoderC1Ev(obj);
    const std::string d = _ZNK2cv12ImageDecoder4Impl14getDescriptionB5cxx11Ev(obj);
    snprintf(out, cap, "%s", d.c_str());
    _ZN2cv10BmpDecoderD1Ev(obj);
    return (int)d.size();
}
}
