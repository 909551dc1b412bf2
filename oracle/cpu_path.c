/*
 * oracle/cpu_path.c -- TEST / MEASUREMENT INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * The reference CPU path of ImageOps.Transform as a C worker loop, for bench.py's `cpu_baseline` leg: N pthreads, each with the buffers
 * an ImageOps holds for its lifetime (/root/reference/ops.go:83-91: two framebuffers allocated once per ImageOps; README.md:82-85: one
 * ImageOps per goroutine), pulling jobs from one atomic counter and running
 *     decode (opencv.cpp:166-171) -> orientation (opencv.cpp:217-221) -> Fit crop + INTER_AREA (opencv.go:331-363, opencv.cpp:196-215)
 *     -> JPEG encode (opencv.cpp:185-194)
 * with no Python between the stages. Round 3 timed the same stages from a Python thread pool (fresh 50 MB numpy frames per call, glue
 * under the GIL) and understated the CPU path about five-fold.
 *
 * The codecs come in as function pointers so that one loop serves both baselines bench.py can report:
 *   kind "reference": the reference's own libjpeg-turbo 3.1.0 / libpng 1.6.47 / libwebp 1.5.0 through oracle/_ref/libref*.so,
 *   kind "port"     : this directory's restatement (lo_jpeg_decode_pixels / lo_jpeg_encode).
 * The resize is imgproc_oracle.c's restatement of cv::resize(INTER_AREA) either way (libopencv_imgproc.a is absent from the mount);
 * it is single-threaded per image like everything else here -- images saturate the cores, which is the throughput-optimal way to
 * run the CPU path.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* jpeg_oracle.c / imgproc_oracle.c (same library) */
typedef struct lo_jpeg_info_s lo_jpeg_info;
int lo_jpeg_header_brief(const uint8_t* d, size_t n, int* w, int* h, int* ncomp, int* orientation);
int lo_resize_area(const uint8_t* src, int sw, int sh, size_t sstep, int cn, uint8_t* dst, int dw, int dh, size_t dstep);
void lo_orientation(const uint8_t* src, int w, int h, size_t sstep, int cn, int orientation, uint8_t* dst, int* dw, int* dh);
long lo_jpeg_encode(const uint8_t* px, int W, int H, int ch, size_t stride, int quality, uint8_t* out, size_t cap, void* dbg);

typedef int (*dec_jpeg_fn)(const uint8_t*, size_t, uint8_t*, size_t, int*, int*, int*);
typedef long (*enc_jpeg_fn)(const uint8_t*, int, int, int, size_t, int, uint8_t*, size_t);
typedef int (*dec_png_fn)(const uint8_t*, size_t, uint8_t*, size_t, int info[6]);
typedef long (*dec_webp_fn)(const uint8_t*, size_t, int, uint8_t*, size_t, int meta[8]);
typedef int (*info_webp_fn)(const uint8_t*, size_t, uint32_t out[8]);
typedef size_t (*enc_webp_fn)(const uint8_t*, int, int, int, float, const uint8_t*, size_t, uint8_t*, size_t);
typedef int (*dec_avif_fn)(const uint8_t*, size_t, uint8_t*, size_t, int info[4]);
/* animated sources (BASELINE configs[3]): the reference's giflib + its restated compositing (ref_gif_driver.c rg_*), libwebp's animation
 * decoder and the reference's animation writer (ref_webp_driver.c ref_webp_play / ref_webp_encode_anim) */
typedef void* (*gif_open_fn)(const uint8_t*, size_t, int dims[2]);
typedef int (*gif_next_fn)(void*, uint8_t* canvas, int meta[11], uint8_t* indices, size_t cap);
typedef void (*gif_close_fn)(void*);
typedef int (*webp_play_fn)(const uint8_t*, size_t, uint8_t* out, size_t cap, int* w, int* h, int* timestamps, int max_frames, uint32_t info[2]);
typedef size_t (*enc_anim_fn)(const uint8_t* frames, int n, int w, int h, int cn, const int* delays, float quality, uint32_t loop_count, uint32_t bgcolor, uint8_t* out, size_t cap);

typedef struct {
    dec_jpeg_fn dec_jpeg;       /* JPEG bytes -> BGR / grey rows */
    enc_jpeg_fn enc_jpeg;       /* NULL: lo_jpeg_encode */
    dec_png_fn dec_png;         /* may be NULL: PNG items fail */
    dec_webp_fn dec_webp;       /* may be NULL */
    info_webp_fn info_webp;
    int width, height, quality;
    int resize_method;          /* 0 none, 1 Fit, 2 Resize (ops.go:18-22) */
    enc_webp_fn enc_webp;       /* non-NULL: WebP output at webp_quality (webp.cpp:707-751) instead of JPEG */
    float webp_quality;
    dec_avif_fn dec_avif;       /* may be NULL: AVIF items fail (ref_avif_driver.c: the reference's libavif + dav1d, avif.cpp:164-321) */
    int animated;               /* non-zero: every frame of a GIF / WebP source -> Fit -> the animation writer at webp_quality (lo_path_transform_anim) */
    gif_open_fn gif_open; gif_next_fn gif_next; gif_close_fn gif_close;
    webp_play_fn webp_play;
    enc_anim_fn enc_anim;
} lo_path_cfg;

typedef struct { uint8_t *frame, *oriented, *thumb; size_t frame_cap, oriented_cap, thumb_cap; int* delays; size_t delays_cap; } lo_path_scratch;

static int need(uint8_t** p, size_t* cap, size_t bytes)
{
    if (bytes <= *cap) return 0;
    free(*p);
    *p = (uint8_t*)malloc(bytes + 64);
    *cap = *p ? bytes : 0;
    return *p ? 0 : -1;
}

/* ops.go:243-255 calculateExpectedSize */
static void expected_size(int ow, int oh, int rw, int rh, int* nw, int* nh)
{
    const int m = ow < oh ? ow : oh;
    if (rw == rh && rw > m) { *nw = m; *nh = m; return; }
    if (rw > ow && rh > oh && rw != rh) { *nw = ow; *nh = oh; return; }
    *nw = rw; *nh = rh;
}

/* opencv.go:331-363 Framebuffer.Fit: the centre crop that has the output's aspect ratio */
static void fit_crop(int fw, int fh, int width, int height, int* left, int* top, int* wpc, int* hpc)
{
    const double aspect_in = (double)fw / (double)fh, aspect_out = (double)width / (double)height;
    if (aspect_in > aspect_out) { *wpc = (int)(aspect_out * (double)fh + 0.5); *hpc = fh; }
    else { *hpc = (int)((double)fw / aspect_out + 0.5); *wpc = fw; }
    if (*wpc < 1) *wpc = 1;
    if (*hpc < 1) *hpc = 1;
    *left = (int)((double)(fw - *wpc) * 0.5);
    *top = (int)((double)(fh - *hpc) * 0.5);
    if (*left < 0) *left = 0;
    if (*top < 0) *top = 0;
}

static uint32_t le32(const uint8_t* p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }

/* One ImageOps.Transform of a still source -> JPEG. Returns the output length, or a negative code. */
long lo_path_transform(const lo_path_cfg* cfg, lo_path_scratch* s, const uint8_t* d, size_t n, uint8_t* out, size_t cap)
{
    int w = 0, h = 0, cn = 0, orientation = 1;
    const uint8_t* px = NULL;
    size_t stride = 0;
    if (n >= 32 && !memcmp(d, "LPPIXELS", 8)) { /* a frame some host codec decoded (include/lilliput_hip.h lilliput_hip_pixels_header) */
        w = (int)le32(d + 8); h = (int)le32(d + 12); cn = (int)le32(d + 16); stride = le32(d + 20); orientation = (int)le32(d + 24);
        if (!stride) stride = (size_t)w * cn;
        if (w <= 0 || h <= 0 || (cn != 1 && cn != 3 && cn != 4) || 32 + stride * (size_t)(h - 1) + (size_t)w * cn > n) return -2;
        px = d + 32;
    } else if (n >= 8 && !memcmp(d, "\x89PNG\r\n\x1a\n", 8)) {
        int info[6];
        if (!cfg->dec_png || cfg->dec_png(d, n, NULL, 0, info)) return -2;
        w = info[0]; h = info[1]; cn = info[2];
        if (need(&s->frame, &s->frame_cap, (size_t)w * h * cn)) return -4;
        if (cfg->dec_png(d, n, s->frame, s->frame_cap, info)) return -2;
        px = s->frame; stride = (size_t)w * cn;
    } else if (n >= 12 && !memcmp(d, "RIFF", 4) && !memcmp(d + 8, "WEBP", 4)) {
        uint32_t inf[8];
        int meta[8];
        if (!cfg->dec_webp || !cfg->info_webp || cfg->info_webp(d, n, inf) != 1) return -2; /* ref_webp_info: 1 = webp_decoder_create would succeed */
        if (need(&s->frame, &s->frame_cap, (size_t)inf[0] * inf[1] * 4 + 16)) return -4;
        if (cfg->dec_webp(d, n, 1, s->frame, s->frame_cap, meta) < 0) return -2;
        w = meta[0]; h = meta[1]; cn = meta[2];
        px = s->frame; stride = (size_t)w * cn;
    } else if (n >= 12 && !memcmp(d + 4, "ftyp", 4) && (!memcmp(d + 8, "avif", 4) || !memcmp(d + 8, "avis", 4))) { /* lilliput.go:136-164: the AVIF signature */
        int info[4];
        if (!cfg->dec_avif || cfg->dec_avif(d, n, NULL, 0, info) != -3) return -2;
        if (need(&s->frame, &s->frame_cap, (size_t)info[0] * info[1] * info[2])) return -4;
        if (cfg->dec_avif(d, n, s->frame, s->frame_cap, info)) return -2;
        w = info[0]; h = info[1]; cn = info[2]; orientation = info[3];
        px = s->frame; stride = (size_t)w * cn;
    } else {
        int nc = 0;
        if (lo_jpeg_header_brief(d, n, &w, &h, &nc, &orientation)) return -2;
        if (need(&s->frame, &s->frame_cap, (size_t)w * h * (nc == 1 ? 1 : 3))) return -4;
        if (cfg->dec_jpeg(d, n, s->frame, s->frame_cap, &w, &h, &cn)) return -2;
        px = s->frame; stride = (size_t)w * cn;
    }
    int fw = w, fh = h;
    if (orientation > 1 && orientation <= 8) { /* ops.go:392: applied unconditionally */
        if (need(&s->oriented, &s->oriented_cap, (size_t)w * h * cn)) return -4;
        lo_orientation(px, w, h, stride, cn, orientation, s->oriented, &fw, &fh);
        px = s->oriented; stride = (size_t)fw * cn;
    }
    int ow = fw, oh = fh;
    if (cfg->resize_method != 0) {
        /* inputCanvasSize (ops.go:474-479): header dimensions, swapped only when the caller asked for normalisation (not modelled: never set here) */
        int hw = fw, hh = fh;
        if (orientation >= 5 && orientation <= 8) { hw = fh; hh = fw; }
        int left = 0, top = 0, wpc = fw, hpc = fh;
        if (cfg->resize_method == 1) {
            expected_size(hw, hh, cfg->width, cfg->height, &ow, &oh);
            fit_crop(fw, fh, ow, oh, &left, &top, &wpc, &hpc);
        } else { ow = cfg->width < 1 ? 1 : cfg->width; oh = cfg->height < 1 ? 1 : cfg->height; }
        if (need(&s->thumb, &s->thumb_cap, (size_t)ow * oh * cn)) return -4;
        lo_resize_area(px + (size_t)top * stride + (size_t)left * cn, wpc, hpc, stride, cn, s->thumb, ow, oh, (size_t)ow * cn);
        px = s->thumb; stride = (size_t)ow * cn;
    }
    if (cfg->enc_webp) { /* the writers take tightly packed BGR / BGRA rows: every buffer above is */
        if (cn == 1 || stride != (size_t)ow * cn) return -5;
        const size_t n2 = cfg->enc_webp(px, ow, oh, cn, cfg->webp_quality, NULL, 0, out, cap);
        return n2 ? (long)n2 : -6;
    }
    return cfg->enc_jpeg ? cfg->enc_jpeg(px, ow, oh, cn, stride, cfg->quality, out, cap) : lo_jpeg_encode(px, ow, oh, cn, stride, cfg->quality, out, cap, NULL);
}

/* One ImageOps.Transform of an animated source -> animated WebP (ops.go:352-444 looping over the frames: every composited canvas through
 * Fit, into the writer with the frame's duration). *frames = frames written. Returns the output length, or a negative code. */
long lo_path_transform_anim(const lo_path_cfg* cfg, lo_path_scratch* s, const uint8_t* d, size_t n, uint8_t* out, size_t cap, int* frames)
{
    *frames = 0;
    if (!cfg->enc_anim) return -2;
    int w = 0, h = 0, nf = 0;
    uint32_t loop_bg[2] = {0, 0xFFFFFFFFu};
    int ow = 0, oh = 0, left = 0, top = 0, wpc = 0, hpc = 0;
    if (n >= 6 && !memcmp(d, "GIF", 3)) {
        if (!cfg->gif_open || !cfg->gif_next || !cfg->gif_close) return -2;
        int dims[2];
        void* g = cfg->gif_open(d, n, dims);
        if (!g) return -2;
        w = dims[0]; h = dims[1];
        if (need(&s->frame, &s->frame_cap, (size_t)w * h * 4)) { cfg->gif_close(g); return -4; }
        memset(s->frame, 0, (size_t)w * h * 4); /* the Go Framebuffer starts cleared */
        expected_size(w, h, cfg->width, cfg->height, &ow, &oh);
        fit_crop(w, h, ow, oh, &left, &top, &wpc, &hpc);
        for (;;) {
            int meta[11];
            const int st = cfg->gif_next(g, s->frame, meta, NULL, 0);
            if (st == 1) break;                                   /* end of file */
            if (st) { cfg->gif_close(g); return -2; }
            const size_t fb = (size_t)ow * oh * 4;
            if ((size_t)(nf + 1) * fb > s->thumb_cap) { /* grow, keeping the frames so far */
                const size_t want = ((size_t)(nf + 1) * fb) * 2;
                uint8_t* p = (uint8_t*)realloc(s->thumb, want + 64);
                if (!p) { cfg->gif_close(g); return -4; }
                s->thumb = p; s->thumb_cap = want;
            }
            if ((size_t)(nf + 1) * sizeof(int) > s->delays_cap) {
                const size_t want = (size_t)(nf + 64) * 2 * sizeof(int);
                int* p = (int*)realloc(s->delays, want);
                if (!p) { cfg->gif_close(g); return -4; }
                s->delays = p; s->delays_cap = want;
            }
            lo_resize_area(s->frame + (size_t)top * w * 4 + (size_t)left * 4, wpc, hpc, (size_t)w * 4, 4, s->thumb + (size_t)nf * fb, ow, oh, (size_t)ow * 4);
            s->delays[nf] = meta[6] > 0 ? meta[6] * 10 : 0;      /* GIF delays are hundredths of a second */
            nf++;
        }
        cfg->gif_close(g);
    } else if (n >= 12 && !memcmp(d, "RIFF", 4) && !memcmp(d + 8, "WEBP", 4)) {
        uint32_t inf[8];
        if (!cfg->webp_play || !cfg->info_webp || cfg->info_webp(d, n, inf) != 1) return -2;
        const int maxf = (int)inf[3];
        if (need(&s->frame, &s->frame_cap, (size_t)inf[0] * inf[1] * 4 * (size_t)maxf + 16)) return -4;
        if ((size_t)(maxf + 1) * sizeof(int) > s->delays_cap) {
            int* p = (int*)realloc(s->delays, (size_t)(maxf + 1) * 2 * sizeof(int));
            if (!p) return -4;
            s->delays = p; s->delays_cap = (size_t)(maxf + 1) * 2 * sizeof(int);
        }
        nf = cfg->webp_play(d, n, s->frame, s->frame_cap, &w, &h, s->delays, maxf, loop_bg);
        if (nf <= 0) return -2;
        for (int i = nf - 1; i > 0; i--) s->delays[i] -= s->delays[i - 1]; /* end timestamps -> durations */
        expected_size(w, h, cfg->width, cfg->height, &ow, &oh);
        fit_crop(w, h, ow, oh, &left, &top, &wpc, &hpc);
        const size_t fb = (size_t)ow * oh * 4;
        if (need(&s->thumb, &s->thumb_cap, (size_t)nf * fb)) return -4;
        for (int i = 0; i < nf; i++)
            lo_resize_area(s->frame + (size_t)i * w * h * 4 + (size_t)top * w * 4 + (size_t)left * 4, wpc, hpc, (size_t)w * 4, 4, s->thumb + (size_t)i * fb, ow, oh, (size_t)ow * 4);
    } else
        return -2;
    if (!nf) return -2;
    const size_t n2 = cfg->enc_anim(s->thumb, nf, ow, oh, 4, s->delays, cfg->webp_quality, loop_bg[0], loop_bg[1], out, cap);
    *frames = nf;
    return n2 ? (long)n2 : -6;
}

void lo_path_scratch_free(lo_path_scratch* s) { free(s->frame); free(s->oriented); free(s->thumb); free(s->delays); memset(s, 0, sizeof(*s)); }

typedef struct {
    const lo_path_cfg* cfg;
    const uint8_t* const* srcs;
    const size_t* lens;
    int nsrc;
    long jobs;
    atomic_long next, ok, failed, frames;
    pthread_barrier_t start;
    /* the output of the first job on every distinct source, for the caller to compare with the Python-level oracle */
    uint8_t* keep; size_t keep_cap; long* keep_len;
} run_t;

static void* worker(void* arg)
{
    run_t* r = (run_t*)arg;
    lo_path_scratch s;
    memset(&s, 0, sizeof(s));
    const size_t cap = 8u << 20;
    uint8_t* out = (uint8_t*)malloc(cap);
    /* one untimed transform: the worker's buffers exist and their pages are touched, as in a service that has been up for a second */
    int nf = 0;
    if (r->cfg->animated) (void)lo_path_transform_anim(r->cfg, &s, r->srcs[0], r->lens[0], out, cap, &nf);
    else (void)lo_path_transform(r->cfg, &s, r->srcs[0], r->lens[0], out, cap);
    pthread_barrier_wait(&r->start);
    for (;;) {
        const long j = atomic_fetch_add(&r->next, 1);
        if (j >= r->jobs) break;
        const int k = (int)(j % r->nsrc);
        nf = 1;
        const long n = r->cfg->animated ? lo_path_transform_anim(r->cfg, &s, r->srcs[k], r->lens[k], out, cap, &nf) : lo_path_transform(r->cfg, &s, r->srcs[k], r->lens[k], out, cap);
        if (n > 0) { atomic_fetch_add(&r->ok, 1); atomic_fetch_add(&r->frames, nf); } else atomic_fetch_add(&r->failed, 1);
        if (j < r->nsrc && r->keep) {
            r->keep_len[k] = n;
            if (n > 0 && (size_t)n <= r->keep_cap) memcpy(r->keep + (size_t)k * r->keep_cap, out, (size_t)n);
        }
    }
    pthread_barrier_wait(&r->start);
    lo_path_scratch_free(&s);
    free(out);
    return NULL;
}

static long g_last_frames = 0;
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

/* `jobs` transforms (job j works on source j % nsrc) on `threads` workers. *seconds = wall time between the barrier that releases the
 * workers and the one they reach when the queue is empty. keep (optional): nsrc slots of keep_cap bytes + keep_len[nsrc].
 * Returns the number of successful transforms. */
long lo_path_run(const lo_path_cfg* cfg, const uint8_t* const* srcs, const size_t* lens, int nsrc, long jobs, int threads, double* seconds,
                 uint8_t* keep, size_t keep_cap, long* keep_len)
{
    if (threads < 1) threads = 1;
    run_t r;
    memset(&r, 0, sizeof(r));
    r.cfg = cfg; r.srcs = srcs; r.lens = lens; r.nsrc = nsrc; r.jobs = jobs;
    r.keep = keep; r.keep_cap = keep_cap; r.keep_len = keep_len;
    atomic_init(&r.next, 0); atomic_init(&r.ok, 0); atomic_init(&r.failed, 0); atomic_init(&r.frames, 0);
    pthread_barrier_init(&r.start, NULL, (unsigned)threads + 1);
    pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
    for (int i = 0; i < threads; i++) pthread_create(&th[i], NULL, worker, &r);
    pthread_barrier_wait(&r.start);
    const double t0 = now_s();
    pthread_barrier_wait(&r.start);
    *seconds = now_s() - t0;
    for (int i = 0; i < threads; i++) pthread_join(th[i], NULL);
    free(th);
    pthread_barrier_destroy(&r.start);
    g_last_frames = atomic_load(&r.frames);
    return atomic_load(&r.ok);
}

long lo_path_last_frames(void) { return g_last_frames; } /* frames the successful transforms of the last lo_path_run wrote (animated mode) */
