// oracle/ref_bmp_driver.cpp -- TEST INFRASTRUCTURE ONLY. Runs OpenCV 4.11's own BMP decoder (modules/imgcodecs/src/grfmt_bmp.cpp, the
// class cv::findDecoder hands a "BM" buffer to in the reference: opencv.cpp:99-171) out of the reference's prebuilt
// libopencv_imgcodecs.a: grfmt_bmp.cpp.o, grfmt_base.cpp.o, bitstrm.cpp.o and utils.cpp.o are extracted where the archive lies and
// linked as they are. libopencv_core.a is not part of the reference's shipped deps, so the handful of core symbols those objects
// refer to are defined here in the smallest form that serves a 1 x N byte buffer and a caller-allocated image; the decoder's own
// class definition is private to the OpenCV sources, so its member functions are called through their mangled names.
#include <opencv2/core.hpp>
#include <opencv2/imgcodecs.hpp>

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <string>

namespace cv {
// ---- cv::Mat, as far as the decoder needs it (headers over memory somebody else owns; no reference counting, no allocation)
Mat::Mat() CV_NOEXCEPT : flags(MAGIC_VAL), dims(0), rows(0), cols(0), data(0), datastart(0), dataend(0), datalimit(0), allocator(0), u(0), size(&rows), step(0) {}
Mat::Mat(const Mat& m)
    : flags(m.flags), dims(m.dims), rows(m.rows), cols(m.cols), data(m.data), datastart(m.datastart), dataend(m.dataend), datalimit(m.datalimit), allocator(0), u(0),
      size(&rows), step(0)
{
    step[0] = m.step[0];
    step[1] = m.step[1];
}
Mat::Mat(int r, int c, int type, void* p, size_t st)
    : flags(MAGIC_VAL + (type & TYPE_MASK) + CONTINUOUS_FLAG), dims(2), rows(r), cols(c), data((uchar*)p), datastart((uchar*)p), dataend(0), datalimit(0), allocator(0), u(0),
      size(&rows), step(0)
{
    const size_t esz = CV_ELEM_SIZE(type), row = st == AUTO_STEP ? (size_t)c * esz : st;
    step[0] = row;
    step[1] = esz;
    datalimit = datastart + row * r;
    dataend = datalimit - row + (size_t)c * esz;
}
Mat::~Mat() {}
void Mat::release() { data = 0; datastart = dataend = datalimit = 0; rows = cols = 0; }
bool Mat::empty() const { return data == 0 || rows * cols == 0; }
size_t Mat::total() const { return (size_t)rows * cols; }
void Mat::reserve(size_t) { abort(); } // the encoder's
void Mat::resize(size_t) { abort(); }
void Mat::updateContinuityFlag() { flags |= CONTINUOUS_FLAG; }
Mat& Mat::operator=(const Mat& m)
{
    flags = m.flags; dims = m.dims; rows = m.rows; cols = m.cols; data = m.data; datastart = m.datastart; dataend = m.dataend; datalimit = m.datalimit;
    step[0] = m.step[0]; step[1] = m.step[1];
    return *this;
}
void cvtColor(InputArray, OutputArray, int, int, AlgorithmHint) { abort(); }
// ---- errors: CV_Assert / CV_Error end here; the decoder's callers catch what it throws
Exception::Exception() : code(0), line(0) {}
Exception::Exception(int c, const String& e, const String& f, const String& fi, int l) : code(c), err(e), func(f), file(fi), line(l) { msg = e; }
Exception::~Exception() throw() {}
const char* Exception::what() const throw() { return msg.c_str(); }
void Exception::formatMessage() {}
void error(int code, const String& err, const char* func, const char* file, int line) { throw Exception(code, err, func ? func : "", file ? file : "", line); }
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
Animation::Animation(int, Scalar) {} // the members were zeroed with the object (empty vectors)
namespace utils { namespace logging {
enum LogLevel { LOG_LEVEL_SILENT = 0 };
struct LogTag;
namespace internal {
LogTag* getGlobalLogTag() { return 0; }
void writeLogMessageEx(LogLevel, const char*, const char*, int, const char*, const char* message) { fprintf(stderr, "opencv: %s\n", message ? message : ""); }
}}}
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
