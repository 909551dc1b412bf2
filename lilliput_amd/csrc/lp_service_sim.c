/* lp_service_sim.c -- a stand-in for the Go service that links this library: N OS threads ("goroutines on their Ms"), each with ONE
 * ImageOps for its lifetime (/root/reference/README.md:82-85, ops.go:83-91), each doing per request what lilliput's callers do:
 *     NewDecoder(buf) -> Header() -> ops.Transform(decoder, options, dst) -> decoder.Close()      (lilliput.go:129-164, ops.go:352-444)
 * through Part C of include/lilliput_hip.h -- plain C against the public header, nothing internal -- or (part = 1) through PART A: the
 * opencv_* calls that UNCHANGED ops.go / opencv.go issue for that request, in their order (one_request_part_a below cites each). bench.py --workload abi drives it
 * (the image has no Go toolchain; Python threads would put the GIL between the calls). Built as ../liblilliput_service_sim.so, which
 * links liblilliput_hip.so like a cgo build would; it is measurement scaffolding, not part of the product library.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../../include/lilliput_hip.h"

typedef struct {
    const void* const* srcs;
    const size_t* lens;
    int nsrc;
    long jobs;
    int width, height, quality, max_size, resize_method;
    const char* file_type;      /* ".jpeg", ".webp", ".png" ... (ImageOptions.FileType) */
    const int* enc_opts;        /* EncodeOptions as key, value, key, value ...; NULL: {JpegQuality: quality} */
    size_t enc_opts_len, dst_cap;
    atomic_long next, ok, failed;
    int first_error;
    pthread_barrier_t gate;
    uint8_t* keep; size_t keep_cap; long* keep_len;
    float* lat_ms;              /* per job, optional */
    int part;                   /* 0: Part C (lilliput_image_ops_transform), 1: Part A (the opencv_* sequence of ops.go) */
} sim_t;

/* ---- Part A: what the Go package does per request, call for call (static JPEG / PNG source -> JPEG, Fit or Resize) ---- */
typedef struct { uint8_t* buf; size_t len; opencv_mat mat; int width, height, pixel_type; } framebuffer_t; /* lilliput.Framebuffer, opencv.go:72-84 */
typedef struct { framebuffer_t fb[2]; int active; uint8_t* icc; } go_ops_t;                                  /* lilliput.ImageOps, ops.go:67-91 */

static int fb_resize_mat(framebuffer_t* f, int w, int h, int ptype) /* opencv.go:250-267 */
{
    if (f->mat) { opencv_mat_release(f->mat); f->mat = NULL; }
    if (opencv_type_depth(ptype) > 8) ptype = opencv_type_convert_depth(ptype, CV_8U);
    opencv_mat m = opencv_mat_create_from_data(w, h, ptype, f->buf, f->len);
    if (!m) return LILLIPUT_ERR_BUF_TOO_SMALL;
    f->mat = m; f->width = w; f->height = h; f->pixel_type = ptype;
    return 0;
}

static int one_request_part_a(sim_t* s, go_ops_t* o, const void* src, size_t len, uint8_t* dst, size_t cap, size_t* n)
{
    int rc = 0;
    /* NewDecoder -> newOpenCVDecoder (opencv.go:442-463) */
    opencv_mat buf = opencv_mat_create_from_data((int)len, 1, CV_8U, (void*)src, len);
    if (!buf) return LILLIPUT_ERR_INVALID_IMAGE;
    opencv_decoder d = opencv_decoder_create(buf);
    if (!d) { opencv_mat_release(buf); return LILLIPUT_ERR_INVALID_IMAGE; }
    opencv_encoder enc = NULL;
    opencv_mat edst = NULL;
    /* Header() (opencv.go:636-660) */
    if (!opencv_decoder_read_header(d)) { rc = LILLIPUT_ERR_INVALID_IMAGE; goto out; }
    {
        const int w = opencv_decoder_get_width(d), h = opencv_decoder_get_height(d), pt = opencv_decoder_get_pixel_type(d), ori = opencv_decoder_get_orientation(d);
        /* initializeTransform -> newOpenCVEncoder (ops.go:483-546, opencv.go:847-870): the destination Mat, the encoder, the source's ICC profile */
        edst = opencv_mat_create_empty_from_data((int)cap, dst);
        enc = edst ? opencv_encoder_create(s->file_type ? s->file_type : ".jpeg", edst) : NULL;
        if (!enc) { rc = LILLIPUT_ERR_INVALID_IMAGE; goto out; }
        (void)opencv_decoder_get_jpeg_icc((void*)src, len, o->icc, 32768);
        /* o.decode -> openCVDecoder.DecodeTo (opencv.go:816-839) */
        framebuffer_t* a = &o->fb[o->active];
        if ((rc = fb_resize_mat(a, w, h, pt))) goto out;
        if (!opencv_decoder_read_data(d, a->mat)) { rc = LILLIPUT_ERR_DECODING_FAILED; goto out; }
        /* o.normalizeOrientation -> Framebuffer.OrientationTransform (ops.go:392, opencv.go:271-279) */
        opencv_mat_orientation_transform((CVImageOrientation)ori, a->mat);
        a->width = opencv_mat_get_width(a->mat);
        a->height = opencv_mat_get_height(a->mat);
        /* transformCurrentFrame (ops.go:449-479) */
        framebuffer_t* b = &o->fb[1 - o->active];
        if (s->resize_method == LILLIPUT_OPS_FIT) { /* o.fit -> Framebuffer.Fit (ops.go:170-204, opencv.go:326-374) */
            int nw, nh, left, top, wpc, hpc;
            lilliput_calculate_expected_size(w, h, s->width, s->height, &nw, &nh);
            lilliput_fit_crop_rect(a->width, a->height, nw, nh, &left, &top, &wpc, &hpc);
            opencv_mat view = opencv_mat_crop(a->mat, left, top, wpc, hpc);
            if (!view) { rc = LILLIPUT_ERR_INVALID_IMAGE; goto out; }
            rc = fb_resize_mat(b, nw, nh, a->pixel_type);
            if (!rc) opencv_mat_resize(view, b->mat, nw, nh, CV_INTER_AREA);
            opencv_mat_release(view);
            if (rc) goto out;
            o->active = 1 - o->active;
        } else if (s->resize_method == LILLIPUT_OPS_RESIZE) { /* o.resize -> Framebuffer.ResizeTo (ops.go:209-238, opencv.go:294-309) */
            const int rw = s->width < 1 ? 1 : s->width, rh = s->height < 1 ? 1 : s->height;
            if ((rc = fb_resize_mat(b, rw, rh, a->pixel_type))) goto out;
            opencv_mat_resize(a->mat, b->mat, rw, rh, CV_INTER_AREA);
            o->active = 1 - o->active;
        }
        /* o.encode -> openCVEncoder.Encode (opencv.go:872-900) */
        {
            const int def[2] = {CV_IMWRITE_JPEG_QUALITY, s->quality};
            const int* eo = s->enc_opts ? s->enc_opts : def;
            const size_t eon = s->enc_opts ? s->enc_opts_len : 2;
            if (!opencv_encoder_write(enc, o->fb[o->active].mat, eo, eon)) { rc = LILLIPUT_ERR_INVALID_IMAGE; goto out; }
            if (opencv_mat_get_data(edst) != (void*)dst) { rc = LILLIPUT_ERR_BUF_TOO_SMALL; goto out; } /* "mat pointer got reallocated" */
            *n = (size_t)opencv_mat_get_height(edst);
        }
    }
out:
    /* enc.Close (deferred in Transform), then the caller's decoder.Close (opencv.go:663-667, 902-905) */
    if (enc) opencv_encoder_release(enc);
    if (edst) opencv_mat_release(edst);
    opencv_decoder_release(d);
    opencv_mat_release(buf);
    return rc;
}

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

static int one_request(sim_t* s, lilliput_image_ops ops, const void* src, size_t len, uint8_t* dst, size_t cap, size_t* n)
{
    lilliput_decoder d = NULL;
    int rc = lilliput_new_decoder(src, len, &d);
    if (rc) return rc;
    int w = 0, h = 0;
    rc = lilliput_decoder_header(d, &w, &h, NULL, NULL, NULL, NULL);
    if (!rc) {
        const int enc[2] = {CV_IMWRITE_JPEG_QUALITY, s->quality};
        lilliput_image_options o;
        memset(&o, 0, sizeof(o));
        o.file_type = s->file_type ? s->file_type : ".jpeg";
        o.width = s->width; o.height = s->height;
        o.resize_method = s->resize_method;
        o.encode_options = s->enc_opts ? s->enc_opts : enc;
        o.encode_options_len = s->enc_opts ? s->enc_opts_len : 2;
        o.encode_timeout_ns = 60ll * 1000000000ll;
        rc = lilliput_image_ops_transform(ops, d, &o, dst, cap, n);
    }
    lilliput_decoder_close(d);
    return rc;
}

static void* sim_worker_part_a(void* arg)
{
    sim_t* s = (sim_t*)arg;
    go_ops_t o; /* NewImageOps(maxSize): two framebuffers of maxSize x maxSize x 4 bytes (ops.go:83-91); pages nobody touches stay unmapped */
    memset(&o, 0, sizeof(o));
    for (int k = 0; k < 2; k++) { o.fb[k].len = (size_t)s->max_size * (size_t)s->max_size * 4; o.fb[k].buf = (uint8_t*)malloc(o.fb[k].len); }
    o.icc = (uint8_t*)malloc(32768);
    const size_t cap = s->dst_cap ? s->dst_cap : (size_t)4u << 20;
    uint8_t* dst = (uint8_t*)malloc(cap);
    const int usable = o.fb[0].buf && o.fb[1].buf && o.icc && dst;
    size_t n = 0;
    if (usable) (void)one_request_part_a(s, &o, s->srcs[0], s->lens[0], dst, cap, &n);
    pthread_barrier_wait(&s->gate);
    for (;;) {
        const long j = atomic_fetch_add(&s->next, 1);
        if (j >= s->jobs) break;
        const int k = (int)(j % s->nsrc);
        const double t0 = now_s();
        const int rc = usable ? one_request_part_a(s, &o, s->srcs[k], s->lens[k], dst, cap, &n) : LILLIPUT_ERR_DEVICE;
        if (s->lat_ms) s->lat_ms[j] = (float)((now_s() - t0) * 1e3);
        if (rc == 0) atomic_fetch_add(&s->ok, 1);
        else { atomic_fetch_add(&s->failed, 1); s->first_error = rc; }
        if (j < s->nsrc && s->keep) {
            s->keep_len[k] = rc == 0 ? (long)n : -(long)rc;
            if (rc == 0 && n <= s->keep_cap) memcpy(s->keep + (size_t)k * s->keep_cap, dst, n);
        }
    }
    pthread_barrier_wait(&s->gate);
    for (int k = 0; k < 2; k++) { if (o.fb[k].mat) opencv_mat_release(o.fb[k].mat); free(o.fb[k].buf); }
    free(o.icc);
    free(dst);
    return NULL;
}

static void* sim_worker(void* arg)
{
    sim_t* s = (sim_t*)arg;
    if (s->part == 1) return sim_worker_part_a(arg);
    lilliput_image_ops ops = lilliput_new_image_ops(s->max_size);
    const size_t cap = s->dst_cap ? s->dst_cap : (size_t)4u << 20;
    uint8_t* dst = (uint8_t*)malloc(cap);
    size_t n = 0;
    if (ops) (void)one_request(s, ops, s->srcs[0], s->lens[0], dst, cap, &n); /* untimed: the thread's first call builds what a running service has */
    pthread_barrier_wait(&s->gate);
    for (;;) {
        const long j = atomic_fetch_add(&s->next, 1);
        if (j >= s->jobs) break;
        const int k = (int)(j % s->nsrc);
        const double t0 = now_s();
        const int rc = ops ? one_request(s, ops, s->srcs[k], s->lens[k], dst, cap, &n) : LILLIPUT_ERR_DEVICE;
        if (s->lat_ms) s->lat_ms[j] = (float)((now_s() - t0) * 1e3);
        if (rc == 0) atomic_fetch_add(&s->ok, 1);
        else { atomic_fetch_add(&s->failed, 1); s->first_error = rc; }
        if (j < s->nsrc && s->keep) {
            s->keep_len[k] = rc == 0 ? (long)n : -(long)rc;
            if (rc == 0 && n <= s->keep_cap) memcpy(s->keep + (size_t)k * s->keep_cap, dst, n);
        }
    }
    pthread_barrier_wait(&s->gate);
    if (ops) lilliput_image_ops_close(ops);
    free(dst);
    return NULL;
}

/* `jobs` requests (request j carries source j % nsrc) served by `threads` workers; *seconds = wall time from the moment every worker
 * stands ready to the moment the queue is empty. Returns the number of successful requests; *first_error = a LILLIPUT_* code if any
 * failed. keep / keep_cap / keep_len: optional copies of the first response per distinct source (nsrc slots). lat_ms: optional,
 * one float per job. resize_method: LILLIPUT_OPS_FIT / _RESIZE / _NO_RESIZE. */
long lilliput_service_sim_run2(const void* const* srcs, const size_t* lens, int nsrc, int threads, long jobs, int width, int height, const char* file_type, const int* enc_opts,
                               size_t enc_opts_len, int resize_method, int max_size, size_t dst_cap, double* seconds, int* first_error, uint8_t* keep, size_t keep_cap,
                               long* keep_len, float* lat_ms);

long lilliput_service_sim_run(const void* const* srcs, const size_t* lens, int nsrc, int threads, long jobs, int width, int height, int quality, int resize_method,
                              int max_size, double* seconds, int* first_error, uint8_t* keep, size_t keep_cap, long* keep_len, float* lat_ms)
{
    const int enc[2] = {CV_IMWRITE_JPEG_QUALITY, quality};
    return lilliput_service_sim_run2(srcs, lens, nsrc, threads, jobs, width, height, ".jpeg", enc, 2, resize_method, max_size, 0, seconds, first_error, keep, keep_cap, keep_len, lat_ms);
}

/* The general form: any output type ImageOps.Transform serves (file_type, EncodeOptions as a flat key / value list), dst_cap bytes of
 * output buffer per worker (0: 4 MiB). */
long lilliput_service_sim_run3(int part, const void* const* srcs, const size_t* lens, int nsrc, int threads, long jobs, int width, int height, const char* file_type,
                               const int* enc_opts, size_t enc_opts_len, int resize_method, int max_size, size_t dst_cap, double* seconds, int* first_error, uint8_t* keep,
                               size_t keep_cap, long* keep_len, float* lat_ms);
long lilliput_service_sim_run2(const void* const* srcs, const size_t* lens, int nsrc, int threads, long jobs, int width, int height, const char* file_type, const int* enc_opts,
                               size_t enc_opts_len, int resize_method, int max_size, size_t dst_cap, double* seconds, int* first_error, uint8_t* keep, size_t keep_cap,
                               long* keep_len, float* lat_ms)
{
    return lilliput_service_sim_run3(0, srcs, lens, nsrc, threads, jobs, width, height, file_type, enc_opts, enc_opts_len, resize_method, max_size, dst_cap, seconds, first_error, keep,
                                     keep_cap, keep_len, lat_ms);
}

/* part: 0 = every request through Part C (lilliput_image_ops_transform), 1 = through Part A, the opencv_* calls of unchanged ops.go */
long lilliput_service_sim_run3(int part, const void* const* srcs, const size_t* lens, int nsrc, int threads, long jobs, int width, int height, const char* file_type,
                               const int* enc_opts, size_t enc_opts_len, int resize_method, int max_size, size_t dst_cap, double* seconds, int* first_error, uint8_t* keep,
                               size_t keep_cap, long* keep_len, float* lat_ms)
{
    const int quality = 0;
    if (threads < 1) threads = 1;
    sim_t s;
    memset(&s, 0, sizeof(s));
    s.srcs = srcs; s.lens = lens; s.nsrc = nsrc; s.jobs = jobs;
    s.width = width; s.height = height; s.quality = quality; s.max_size = max_size; s.resize_method = resize_method;
    s.file_type = file_type; s.enc_opts = enc_opts; s.enc_opts_len = enc_opts_len; s.dst_cap = dst_cap;
    s.keep = keep; s.keep_cap = keep_cap; s.keep_len = keep_len; s.lat_ms = lat_ms;
    s.part = part;
    atomic_init(&s.next, 0); atomic_init(&s.ok, 0); atomic_init(&s.failed, 0);
    pthread_barrier_init(&s.gate, NULL, (unsigned)threads + 1);
    pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
    pthread_attr_t at;
    pthread_attr_init(&at);
    pthread_attr_setstacksize(&at, 1u << 20);
    for (int i = 0; i < threads; i++) pthread_create(&th[i], &at, sim_worker, &s);
    pthread_barrier_wait(&s.gate);
    const double t0 = now_s();
    pthread_barrier_wait(&s.gate);
    *seconds = now_s() - t0;
    for (int i = 0; i < threads; i++) pthread_join(th[i], NULL);
    free(th);
    pthread_attr_destroy(&at);
    pthread_barrier_destroy(&s.gate);
    if (first_error) *first_error = s.first_error;
    return atomic_load(&s.ok);
}
