/*
 * oracle/ref_avif_driver.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * The reference's own AVIF decode: libavif 1.x + dav1d + libyuv from /root/reference/deps/linux/amd64/lib, driven the way
 * /root/reference/avif.cpp:164-237 (avif_decoder_create) and :277-321 (avif_decoder_decode: avifImageYUVToRGB into 8-bit BGR / BGRA)
 * drive them for a still image. libavif.a also holds the aom codec glue (codec_aom.c.o), whose libaom.a is among the blobs missing
 * from this mount: oracle/Makefile leaves that one object out and this file supplies the two entry points avif.c's codec table names
 * for it (no aom codec: decoding goes to dav1d, which is what libavif picks for decoding anyway). Decode only -- which is all the
 * firehose needs (AVIF sources -> JPEG): the checker's answer for an AVIF item (oracle.transform_any_frame) and the reference CPU
 * path's decode in bench.py's cpu_baseline leg (oracle/cpu_path.c). Never part of the measured GPU path: there the AV1 decode is the
 * bench's host feeder (Pillow's bundled libavif), and the frames enter the library through the hand-over item.
 */
#include <avif/avif.h>
#include <stdint.h>
#include <string.h>

const char* avifCodecVersionAOM(void) { return "absent"; }
struct avifCodec* avifCodecCreateAOM(void) { return 0; }

/* EXIF-style orientation of the image's irot / imir properties: avif.cpp:277-321 avif_decoder_get_orientation */
static int orientation_of(const avifImage* im)
{
    const int angle = (im->transformFlags & AVIF_TRANSFORM_IROT) ? (im->irot.angle & 3) : 0;
    const int mirrored = (im->transformFlags & AVIF_TRANSFORM_IMIR) != 0;
    const int axis = mirrored ? (im->imir.axis & 1) : 0;
    if (!mirrored) { static const int o[4] = {1, 8, 3, 6}; return o[angle]; }          /* TL, LB, BR, RT */
    switch (angle) {
    case 0: return axis == 0 ? 4 : 2;   /* BL : TR */
    case 1: return axis == 0 ? 5 : 7;   /* LT : RB */
    case 2: return axis == 0 ? 2 : 4;
    default: return axis == 0 ? 7 : 5;
    }
}

/* info[0..3] = width, height, channels (3 BGR / 4 BGRA), orientation. 0: decoded into out (tightly packed rows); -3: cap too small
 * (info is set); 1: not an AVIF libavif takes (strict mode) / decode failed */
int ref_avif_decode(const uint8_t* data, size_t len, uint8_t* out, size_t cap, int info[4])
{
    avifDecoder* d = avifDecoderCreate();
    if (!d) return 1;
    d->strictFlags = AVIF_STRICT_ENABLED;
    if (avifDecoderSetIOMemory(d, data, len) != AVIF_RESULT_OK || avifDecoderParse(d) != AVIF_RESULT_OK || avifDecoderNextImage(d) != AVIF_RESULT_OK) {
        avifDecoderDestroy(d);
        return 1;
    }
    avifRGBImage rgb;
    avifRGBImageSetDefaults(&rgb, d->image);
    rgb.depth = 8;
    rgb.format = d->image->alphaPlane ? AVIF_RGB_FORMAT_BGRA : AVIF_RGB_FORMAT_BGR;
    info[0] = (int)d->image->width; info[1] = (int)d->image->height; info[2] = d->image->alphaPlane ? 4 : 3; info[3] = orientation_of(d->image);
    int rc = 0;
    if ((size_t)info[0] * info[1] * info[2] > cap) rc = -3;
    else {
        rgb.pixels = out;
        rgb.rowBytes = (uint32_t)(info[0] * info[2]);
        if (avifImageYUVToRGB(d->image, &rgb) != AVIF_RESULT_OK) rc = 1;
    }
    avifDecoderDestroy(d);
    return rc;
}
