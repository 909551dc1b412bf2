// oracle/ref_jpegcv_driver.cpp -- TEST INFRASTRUCTURE ONLY. Runs the reference's own JPEG decoder class: cv::JpegDecoder
// (modules/imgcodecs/src/grfmt_jpeg.cpp of the patched OpenCV 4.11 the reference links; the class cv::findDecoder hands a JPEG buffer
// to behind opencv_decoder_create / read_header / read_data, /root/reference/opencv.cpp:99-171) out of the reference's prebuilt
// libopencv_imgcodecs.a -- grfmt_jpeg.cpp.o, grfmt_base.cpp.o, utils.cpp.o, exif.cpp.o extracted where the archive lies and linked
// as they are -- over the reference's own prebuilt libjpeg.a (libjpeg-turbo 3.1.0). The core symbols those objects need come from
// ref_cvstubs.h. What this adds over ref_driver.c (the same libjpeg.a behind jpeg_mem_src): OpenCV's own source manager
// (fill_input_buffer answers FALSE: a decoder that runs out of bytes is SUSPENDED, jpeg_read_scanlines returns 0 rows and readData
// answers false -- jpeg_mem_src would have faked an EOI and painted the rest grey), its readHeader / readData call sequence, its error
// handling (result = true is set before jpeg_finish_decompress: what follows the last scanline cannot fail the image).
// The decoder's class definition is private to OpenCV's sources: its members are called through their mangled names, the header
// fields are read at their offsets behind the vtable pointer (grfmt_base.hpp: int m_width, m_height, m_type).
#include "ref_cvstubs.h"

#include <stdarg.h>

#include <jpeglib.h>

namespace cv {
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

// libjpeg's warnings ("Corrupt JPEG data: premature end of data segment") go to stderr through jpeg_std_error's output_message; the
// link wraps jpeg_std_error so that they can be silenced (REF_CV_LOG=1 lets them through). Nothing else about the error manager changes.
extern "C" struct jpeg_error_mgr* __real_jpeg_std_error(struct jpeg_error_mgr* err);
static void quiet_message(j_common_ptr) {}
extern "C" struct jpeg_error_mgr* __wrap_jpeg_std_error(struct jpeg_error_mgr* err)
{
    struct jpeg_error_mgr* r = __real_jpeg_std_error(err);
    if (!getenv("REF_CV_LOG")) r->output_message = quiet_message;
    return r;
}

extern "C" {
void _ZN2cv11JpegDecoderC1Ev(void* self);
void _ZN2cv11JpegDecoderD1Ev(void* self);
bool _ZN2cv11JpegDecoder10readHeaderEv(void* self);
bool _ZN2cv11JpegDecoder8readDataERNS_3MatE(void* self, cv::Mat* img);
bool _ZN2cv12ImageDecoder4Impl9setSourceERKNS_3MatE(void* self, const cv::Mat* buf);
static int field(const void* self, int i) { int v; memcpy(&v, (const char*)self + 8 + 4 * i, 4); return v; }
// the encoder class of the same object file (what cv::ImageEncoder(".jpeg", dst) holds behind opencv_encoder_create / opencv_encoder_write,
// /root/reference/opencv.cpp:173-194): constructor, the patched base class's setDestination(Mat&), write(img, params)
void _ZN2cv11JpegEncoderC1Ev(void* self);
void _ZN2cv11JpegEncoderD1Ev(void* self);
bool _ZN2cv11JpegEncoder5writeERKNS_3MatERKSt6vectorIiSaIiEE(void* self, const cv::Mat* img, const std::vector<int>* params);
bool _ZN2cv12ImageEncoder4Impl14setDestinationERNS_3MatE(void* self, cv::Mat* dst);

// 0: decoded, out = h x w x channels (CV_8UC1 / CV_8UC3) the way opencv_decoder_read_data fills a Mat of the decoder's own type;
// 1: readHeader refused the file; 2: readData failed (what the Go layer reports as ErrDecodingFailed, opencv.go:828-831); -1: cap too small
int ref_cvjpeg_decode(const uint8_t* data, size_t len, int* w, int* h, int* type, uint8_t* out, size_t cap)
{
    alignas(64) static thread_local unsigned char obj[16384];
    memset(obj, 0, sizeof(obj));
    _ZN2cv11JpegDecoderC1Ev(obj);
    int rc = 0;
    try {
        cv::Mat buf(1, (int)len, CV_8U, const_cast<uint8_t*>(data));
        try { // opencv_decoder_read_header catches what readHeader throws and answers false (opencv.cpp:127-140)
            if (!_ZN2cv12ImageDecoder4Impl9setSourceERKNS_3MatE(obj, &buf) || !_ZN2cv11JpegDecoder10readHeaderEv(obj)) rc = 1;
        } catch (...) { rc = 1; }
        if (!rc) {
            *w = field(obj, 0); *h = field(obj, 1); *type = field(obj, 2);
            const size_t need = (size_t)*w * *h * CV_MAT_CN(*type);
            if (*w <= 0 || *h <= 0 || need > cap) rc = -1;
            else {
                cv::Mat img(*h, *w, *type, out);
                if (!_ZN2cv11JpegDecoder8readDataERNS_3MatE(obj, &img)) rc = 2;
            }
        }
    } catch (...) { rc = rc ? rc : 2; }
    try { _ZN2cv11JpegDecoderD1Ev(obj); } catch (...) {}
    return rc;
}

// cv::JpegEncoder::write the way opencv_encoder_write drives it: img = h x w of `type` (CV_8UC1 / CV_8UC3 / CV_8UC4 ...) with row step
// `step`, params = the int pairs of opencv.go:877-884 as they are (any key, any value). The destination is a 0 x 1 CV_8U Mat over `out`
// with datalimit = out + cap, like opencv_mat_create_empty_from_data. Returns the encoded length; -1: write answered false or threw;
// -2: the result did not fit and the Mat moved to a block of its own (*moved_len = its length: what the Go layer calls ErrBufTooSmall).
long ref_cvjpeg_encode(const uint8_t* px, int w, int h, int type, size_t step, const int* params, int nparams, uint8_t* out, size_t cap, long* moved_len)
{
    alignas(64) static thread_local unsigned char obj[16384];
    memset(obj, 0, sizeof(obj));
    _ZN2cv11JpegEncoderC1Ev(obj);
    long rc = -1;
    try {
        cv::Mat dst(0, 1, CV_8U, out);
        dst.dataend = dst.data;
        dst.datalimit = dst.data + cap;
        cv::Mat img(h, w, type, const_cast<uint8_t*>(px), step);
        const std::vector<int> p(params, params + nparams);
        if (_ZN2cv12ImageEncoder4Impl14setDestinationERNS_3MatE(obj, &dst) && _ZN2cv11JpegEncoder5writeERKNS_3MatERKSt6vectorIiSaIiEE(obj, &img, &p)) {
            if (dst.data == out) rc = dst.rows;
            else { rc = -2; if (moved_len) *moved_len = dst.rows; }
        }
    } catch (...) { rc = -1; }
    try { _ZN2cv11JpegEncoderD1Ev(obj); } catch (...) {}
    return rc;
}
}
