// oracle/ref_cvstubs.h -- TEST INFRASTRUCTURE ONLY. The handful of OpenCV core symbols that the decoder objects extracted from the
// reference's libopencv_imgcodecs.a refer to (libopencv_core.a is not part of the reference's shipped deps), in the smallest form that
// serves a 1 x N byte buffer and a caller-allocated image. Shared by ref_bmp_driver.cpp (cv::BmpDecoder) and ref_jpegcv_driver.cpp
// (cv::JpegDecoder). Include once per shared object.
#pragma once
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
void Mat::reserve(size_t) { abort(); }
// cv::Mat::resize(nelems) as the patched encoder's destination manager uses it on its 1-column CV_8U destination (a Mat over the caller's
// buffer whose datalimit tells the capacity, /root/reference/opencv.cpp:38-49): rows grow in place while the bytes fit; beyond the
// capacity a new block is taken and the data pointer CHANGES -- which is how opencv.go:890-895 notices ErrBufTooSmall. (The block is
// leaked: test infrastructure, once per overflowing call.)
void Mat::resize(size_t nelems)
{
    const size_t row = step[0] ? (size_t)step[0] : (size_t)CV_ELEM_SIZE(flags & TYPE_MASK) * cols;
    if ((size_t)(datalimit - datastart) < nelems * row) {
        const size_t ncap = nelems * row * 2 + 64;
        uchar* nb = (uchar*)malloc(ncap);
        memcpy(nb, data, (size_t)rows * row);
        data = nb;
        datastart = nb;
        datalimit = nb + ncap;
    }
    rows = (int)nelems;
    dataend = data + nelems * row;
}
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
Animation::Animation(int, Scalar) {} // the members were zeroed with the object (empty vectors)
namespace utils { namespace logging {
enum LogLevel { LOG_LEVEL_SILENT = 0 };
struct LogTag;
namespace internal {
LogTag* getGlobalLogTag() { return 0; }
void writeLogMessageEx(LogLevel, const char*, const char*, int, const char*, const char* message) { if (getenv("REF_CV_LOG")) fprintf(stderr, "opencv: %s\n", message ? message : ""); }
}}}
} // namespace cv
