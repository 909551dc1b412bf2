// lp_ops_logic.h -- the pure control logic of lilliput's Go layer, restated in C++ (no Go toolchain in the
// build image). Integer / float64 arithmetic only; must match the reference bit for bit.
#pragma once
#include <stddef.h>
#include <stdint.h>

// ops.go:243-255 calculateExpectedSize
inline void lp_calculate_expected_size(int ow, int oh, int rw, int rh, int* w, int* h)
{
    int mn = ow < oh ? ow : oh;
    if (rw == rh && rw > mn) { *w = mn; *h = mn; }
    else if (rw > ow && rh > oh && rw != rh) { *w = ow; *h = oh; }
    else { *w = rw; *h = rh; }
}

// opencv.go:331-363 Framebuffer.Fit crop rectangle (float64, int(x+0.5), offsets truncated)
inline void lp_fit_crop_rect(int fw, int fh, int width, int height, int* left, int* top, int* wpc, int* hpc)
{
    double aspect_in = (double)fw / (double)fh;
    double aspect_out = (double)width / (double)height;
    int w, h;
    if (aspect_in > aspect_out) { w = (int)((aspect_out * (double)fh) + 0.5); h = fh; }
    else { h = (int)(((double)fw / aspect_out) + 0.5); w = fw; }
    if (w < 1) w = 1;
    if (h < 1) h = 1;
    int l = (int)((double)(fw - w) * 0.5);
    if (l < 0) l = 0;
    int t = (int)((double)(fh - h) * 0.5);
    if (t < 0) t = 0;
    *left = l; *top = t; *wpc = w; *hpc = h;
}

inline bool lp_swaps_axes(int orientation) { return orientation >= 5 && orientation <= 8; } // opencv.go:174-180

struct LpOpsPlan {
    bool resize;                        // false: the (oriented) frame is encoded as is
    int crop_x, crop_y, crop_w, crop_h; // source view (opencv_mat_crop)
    int out_w, out_h;                   // opencv_mat_resize target
};

// transformCurrentFrame for a single-frame source (ops.go:449-479): hdr_* are the header dims,
// frame_* the dims of the frame after the unconditional orientation transform (ops.go:392).
inline LpOpsPlan lp_plan_static_transform(int hdr_w, int hdr_h, int orientation, int req_w, int req_h, int method, bool normalize_orientation,
                                          int frame_w, int frame_h)
{
    LpOpsPlan p = {false, 0, 0, frame_w, frame_h, frame_w, frame_h};
    if (method == 0) return p; // ImageOpsNoResize && !animated
    int in_w = hdr_w, in_h = hdr_h; // inputCanvasSize
    if (normalize_orientation && lp_swaps_axes(orientation)) { in_w = hdr_h; in_h = hdr_w; }
    p.resize = true;
    if (method == 1) { // ImageOpsFit -> o.fit -> Framebuffer.Fit
        int nw, nh;
        lp_calculate_expected_size(in_w, in_h, req_w, req_h, &nw, &nh);
        lp_fit_crop_rect(frame_w, frame_h, nw, nh, &p.crop_x, &p.crop_y, &p.crop_w, &p.crop_h);
        p.out_w = nw; p.out_h = nh;
    } else { // ImageOpsResize -> Framebuffer.ResizeTo
        p.out_w = req_w < 1 ? 1 : req_w;
        p.out_h = req_h < 1 ? 1 : req_h;
    }
    return p;
}

// opencv.go:468-614 content-length sniffers, opencv.go:617-637 APNG detection
int lp_detect_content_length(const uint8_t* buf, size_t len);
bool lp_detect_apng(const uint8_t* buf, size_t len);
