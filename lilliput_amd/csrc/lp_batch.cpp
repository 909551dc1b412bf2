// lp_batch.cpp -- Part B of include/lilliput_hip.h: the batched JPEG -> (orientation, Fit/Resize) -> JPEG
// entry point. Semantics per item are those of ImageOps.Transform for a static JPEG source
// (/root/reference/ops.go:352-479, opencv.go:326-374, 816-900); the images of a batch are independent,
// so a multi-GPU caller simply gives each device's batch object its own shard of the items.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <memory>
#include <string>
#include <thread>

#include <algorithm>
#include <vector>

#include "lp_abi.h"
#include "lp_ops_logic.h"

// One engine (= one HIP stream + its arenas) per worker; a batch is split into contiguous parts, one per worker, and the
// workers run concurrently on host threads so that one part's HBM-bound stages (IDCT, resample, unstuff) overlap another
// part's VALU-bound Huffman stages on the same GPU.
struct LpBatchPart {
    std::unique_ptr<LpEngine> eng;
    std::vector<LpJpegHeader> hdrs;     // parsed headers of this part's items (upload order)
    std::vector<int> items;             // their indices in the item array
    float acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t rounds = 0;
    double tw[6] = {0, 0, 0, 0, 0, 0};
    std::string err;
    int rc = 0;
};

struct LpBatch {
    int device = 0;
    std::vector<LpBatchPart> parts;
    std::vector<int> parse_status;      // LILLIPUT_* per uploaded item
    // results of the last run, indexed like the item array
    std::vector<int> status, out_w, out_h;
    std::vector<uint32_t> out_len;
    std::vector<std::vector<uint8_t>> out_bytes;   // host copies fetched during run (download hands them to the caller)
    size_t n_items = 0;
    uint32_t S = 0, C = 0;
    LpTimings tm = {};
    // sources other than JPEG that the one-image path can serve (GIF: first frame through the animated composite path; PNG): copied at
    // upload, transformed one by one on the calling thread while the JPEG parts run
    std::vector<std::pair<int, std::vector<uint8_t>>> other;
    std::vector<lilliput_image_ops> other_ops; // one per worker
    ~LpBatch() { for (auto o : other_ops) if (o) lilliput_image_ops_close(o); }
    LpEngine& eng0() { return *parts[0].eng; }
    bool ensure_parts(size_t n)
    {
        while (parts.size() < n) {
            LpBatchPart p;
            p.eng.reset(new LpEngine(device));
            if (!p.eng->ok()) { lp_set_error(p.eng->last_error()); return false; }
            p.eng->set_subsequence(S, C);
            parts.push_back(std::move(p));
        }
        return true;
    }
};

static int map_parse(int rc)
{
    switch (rc) {
    case LP_PARSE_OK: return LILLIPUT_OK;
    case LP_PARSE_UNSUPPORTED: return LILLIPUT_ERR_UNSUPPORTED;
    default: return LILLIPUT_ERR_INVALID_IMAGE;
    }
}

static int map_status(int st)
{
    switch (st) {
    case LP_OK: return LILLIPUT_OK;
    case LP_ERR_INVALID_IMAGE: return LILLIPUT_ERR_INVALID_IMAGE;
    case LP_ERR_DECODE_FAILED: return LILLIPUT_ERR_DECODING_FAILED;
    case LP_ERR_BUF_TOO_SMALL: return LILLIPUT_ERR_BUF_TOO_SMALL;
    case LP_ERR_UNSUPPORTED: return LILLIPUT_ERR_UNSUPPORTED;
    default: return LILLIPUT_ERR_DEVICE;
    }
}

extern "C" {

lilliput_hip_batch lilliput_hip_batch_create(int device)
{
    auto b = new LpBatch();
    b->device = device;
    if (!b->ensure_parts(1)) { delete b; return nullptr; }
    return b;
}

void lilliput_hip_batch_destroy(lilliput_hip_batch b) { delete static_cast<LpBatch*>(b); }

void lilliput_hip_batch_set_subsequence(lilliput_hip_batch bb, unsigned S, unsigned C)
{
    auto b = static_cast<LpBatch*>(bb);
    b->S = S; b->C = C;
    for (auto& p : b->parts) p.eng->set_subsequence(S, C);
}

void lilliput_hip_batch_timings(lilliput_hip_batch bb, float out_ms[10], int* verify_rounds)
{
    const LpTimings& t = static_cast<LpBatch*>(bb)->tm;
    out_ms[6] = t.huff_spec_ms; out_ms[7] = t.huff_verify_ms; out_ms[8] = t.huff_scan_ms; out_ms[9] = t.huff_write_ms;
    out_ms[0] = t.unstuff_ms; out_ms[1] = t.huff_ms; out_ms[2] = t.idct_ms; out_ms[3] = t.color_ms; out_ms[4] = t.resize_ms; out_ms[5] = t.encode_ms;
    if (verify_rounds) *verify_rounds = (int)t.verify_rounds;
}

static int batch_streams(int requested)
{
    int n = requested;
    if (n <= 0) { const char* e = getenv("LILLIPUT_HIP_STREAMS"); n = e ? atoi(e) : 4; }
    return std::max(1, std::min(8, n));
}

int lilliput_hip_batch_upload2(lilliput_hip_batch bb, const lilliput_batch_item* items, size_t n, int streams)
{
    auto b = static_cast<LpBatch*>(bb);
    if (!b) return LILLIPUT_ERR_DEVICE;
    b->n_items = n;
    b->parse_status.assign(n, LILLIPUT_OK);
    b->other.clear();
    std::vector<int> valid;
    std::vector<LpJpegHeader> hv;
    for (size_t i = 0; i < n; i++) {
        LpJpegHeader h;
        const uint8_t* sp = (const uint8_t*)items[i].src;
        static const uint8_t png_sig[8] = {0x89, 0x50, 0x4E, 0x47, 0x0D, 0x0A, 0x1A, 0x0A};
        if (sp && ((items[i].src_len >= 6 && (memcmp(sp, "GIF87a", 6) == 0 || memcmp(sp, "GIF89a", 6) == 0)) || // lilliput.go:100-102 isGIF
                   (items[i].src_len >= 8 && memcmp(sp, png_sig, 8) == 0))) {
            b->other.emplace_back((int)i, std::vector<uint8_t>(sp, sp + items[i].src_len));
            continue;
        }
        int rc = (items[i].src && items[i].src_len) ? lp_jpeg_parse((const uint8_t*)items[i].src, items[i].src_len, &h) : LP_PARSE_NOT_JPEG;
        b->parse_status[i] = map_parse(rc);
        // The reference sizes its frame buffers with NewImageOps(maxSize) and answers ErrBufTooSmall for anything larger
        // (opencv.go:250-267 resizeMat); the batch has the same bound -- 8192 x 8192 pixels unless LILLIPUT_HIP_BATCH_MAX_PIXELS says
        // otherwise -- so that one file with an absurd frame header cannot take the other images' arenas down with it.
        static const uint64_t max_px = getenv("LILLIPUT_HIP_BATCH_MAX_PIXELS") ? strtoull(getenv("LILLIPUT_HIP_BATCH_MAX_PIXELS"), nullptr, 10) : 8192ull * 8192ull;
        if (rc == LP_PARSE_OK && (uint64_t)h.j.width * h.j.height > max_px) { b->parse_status[i] = LILLIPUT_ERR_BUF_TOO_SMALL; continue; }
        if (rc == LP_PARSE_OK) { valid.push_back((int)i); hv.push_back(h); }
    }
    // contiguous parts, one per worker; small batches stay on one stream
    size_t np = (size_t)batch_streams(streams);
    if (valid.size() < 2 * np) np = 1;
    if (!b->ensure_parts(np)) return LILLIPUT_ERR_DEVICE;
    for (auto& p : b->parts) { p.hdrs.clear(); p.items.clear(); }
    for (size_t q = 0; q < valid.size(); q++) {
        LpBatchPart& p = b->parts[q * np / valid.size()];
        p.items.push_back(valid[q]);
        p.hdrs.push_back(hv[q]);
    }
    for (size_t k = 0; k < np; k++) {
        LpBatchPart& p = b->parts[k];
        if (p.items.empty()) continue;
        std::vector<LpJpegSrc> srcs;
        for (int it : p.items) srcs.push_back(LpJpegSrc{(const uint8_t*)items[it].src, items[it].src_len});
        int rc = p.eng->upload_jpegs(srcs.data(), (int)srcs.size(), p.hdrs.data());
        if (rc) { lp_set_error(p.eng->last_error()); return map_status(rc); }
    }
    return LILLIPUT_OK;
}

int lilliput_hip_batch_upload(lilliput_hip_batch bb, const lilliput_batch_item* items, size_t n) { return lilliput_hip_batch_upload2(bb, items, n, 0); }

// Every stage of ImageOps.Transform for one part of the batch (runs on the part's own host thread).
static int run_part(LpBatch* b, LpBatchPart& part, const lilliput_batch_options* opt, bool trace)
{
    LpEngine& eng = *part.eng;
    const size_t nv = part.items.size();
    float* acc = part.acc;
    for (int i = 0; i < 10; i++) acc[i] = 0;
    for (int i = 0; i < 6; i++) part.tw[i] = 0;
    part.rounds = 0;
    if (!nv) return LILLIPUT_OK;
    auto fail = [&](const std::string& m) { part.err = m; return LILLIPUT_ERR_DEVICE; };
    const int quality = opt->jpeg_quality > 0 ? opt->jpeg_quality : 95;
    // chunk size: bound the working set (coefficients + planes + BGR frame ~ 7.5 B/pixel + oriented copy)
    size_t max_px = 1;
    for (auto& h : part.hdrs) max_px = std::max(max_px, (size_t)h.j.mcus_x * h.j.hmax * 8 * h.j.mcus_y * h.j.vmax * 8);
    size_t chunk = opt->chunk > 0 ? (size_t)opt->chunk : std::max<size_t>(1, std::min<size_t>(128, (size_t)(24ull << 30) / (max_px * 12)));
    uint32_t& rounds = part.rounds;
    double* tw = part.tw;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_prev = now();
    auto lap = [&](int k) { const double t = now(); tw[k] += t - t_prev; t_prev = t; };
    (void)trace;
    eng.enable_timing(true);
    for (size_t first = 0; first < nv; first += chunk) {
        const int cnt = (int)std::min(chunk, nv - first);
        // frame heap: thumbnails for the fused images; decoded frame (+ oriented copy) + resized frame for the others
        size_t need = 0;
        for (int k = 0; k < cnt; k++) {
            const LpJpeg& j = part.hdrs[first + k].j;
            const bool swap = j.orientation >= 5;
            const int OW = swap ? (int)j.height : (int)j.width, OH = swap ? (int)j.width : (int)j.height;
            LpOpsPlan plan = lp_plan_static_transform((int)j.width, (int)j.height, (int)j.orientation, opt->width, opt->height, opt->resize_method,
                                                      opt->normalize_orientation != 0, OW, OH);
            int ix, iy;
            const bool fused = plan.resize && j.ncomp != 4 && !j.generic_sampling && lp_resize_mode(plan.crop_w, plan.crop_h, plan.out_w, plan.out_h, &ix, &iy) == 1;
            const size_t cn = j.ncomp == 1 ? 1 : 3, fb = (size_t)j.width * j.height * cn;
            if (!fused) need += fb + 512 + (j.orientation != 1 ? fb + 512 : 0);
            if (plan.resize) need += (size_t)plan.out_w * plan.out_h * cn + 512;
        }
        if (!eng.heap_reserve(need + 4096)) return fail("frame heap allocation failed");
        eng.heap_reset();
        // Plan every image first (ops.go:449-479 + opencv.go:294-374). Integer-scale area resizes take the fused
        // path (planes -> thumbnail, orientation and crop folded into the addressing); everything else goes through
        // a materialised BGR frame exactly like the one-image ABI does.
        std::vector<LpFrame> frames((size_t)cnt);
        std::vector<int> st((size_t)cnt, 0);
        std::vector<uint8_t> want((size_t)cnt, 1);
        std::vector<LpOpsPlan> plans((size_t)cnt);
        std::vector<LpFusedOp> fops;
        std::vector<int> fidx;
        memset(frames.data(), 0, sizeof(LpFrame) * (size_t)cnt);
        for (int k = 0; k < cnt; k++) {
            const LpJpeg& j = part.hdrs[first + k].j;
            const bool swap = j.orientation >= 5;
            const int OW = swap ? (int)j.height : (int)j.width, OH = swap ? (int)j.width : (int)j.height;
            LpOpsPlan plan = lp_plan_static_transform((int)j.width, (int)j.height, (int)j.orientation, opt->width, opt->height, opt->resize_method,
                                                      opt->normalize_orientation != 0, OW, OH);
            plans[(size_t)k] = plan;
            int ix = 1, iy = 1;
            if (j.ncomp == 4 || j.generic_sampling) continue; // CMYK / YCCK, unusual sampling factors: converted to a BGR frame first (the fused kernels read grey / YCbCr / RGB planes)
            if (!plan.resize || lp_resize_mode(plan.crop_w, plan.crop_h, plan.out_w, plan.out_h, &ix, &iy) != 1) continue;
            // oriented-frame rectangle of destination (dx, dy) -> rectangle of the un-oriented decoded image (cv::ExifTransform inverse)
            auto map = [&](int dx, int dy, int* fx, int* fy) {
                const int ox0 = plan.crop_x + dx * ix, oy0 = plan.crop_y + dy * iy;
                switch (j.orientation) {
                case 2: *fx = OW - ox0 - ix; *fy = oy0; break;
                case 3: *fx = OW - ox0 - ix; *fy = OH - oy0 - iy; break;
                case 4: *fx = ox0; *fy = OH - oy0 - iy; break;
                case 5: *fx = oy0; *fy = ox0; break;
                case 6: *fx = oy0; *fy = OW - ox0 - ix; break;
                case 7: *fx = OH - oy0 - iy; *fy = OW - ox0 - ix; break;
                case 8: *fx = OH - oy0 - iy; *fy = ox0; break;
                default: *fx = ox0; *fy = oy0; break;
                }
            };
            LpFusedOp op;
            memset(&op, 0, sizeof(op));
            op.img = (uint32_t)k;
            op.rw = (uint32_t)(swap ? iy : ix);
            op.rh = (uint32_t)(swap ? ix : iy);
            int ax, ay, bx, by;
            map(0, 0, &op.x0, &op.y0);
            map(1, 0, &ax, &ay);
            map(0, 1, &bx, &by);
            op.dxx = ax - op.x0; op.dxy = ay - op.y0; op.dyx = bx - op.x0; op.dyy = by - op.y0;
            op.inv_area = 1.f / (float)(ix * iy);
            op.round_2x2 = (ix == 2 && iy == 2) ? 1 : 0;
            op.dst.w = (uint32_t)plan.out_w; op.dst.h = (uint32_t)plan.out_h; op.dst.cn = j.ncomp == 1 ? 1 : 3;
            op.dst.stride = op.dst.w * op.dst.cn;
            uint8_t* p = eng.heap_alloc((size_t)op.dst.stride * op.dst.h);
            if (!p) return fail("frame heap exhausted");
            op.dst.off = (uint64_t)(uintptr_t)p;
            fops.push_back(op);
            fidx.push_back(k);
            want[(size_t)k] = 0;
        }
        lap(0);
        int rc = eng.decode_uploaded((int)first, cnt, frames.data(), st.data(), want.data());
        lap(1);
        if (rc == LP_ERR_DEVICE) return fail(eng.last_error());
        { const LpTimings& t = eng.timings(); acc[0] += t.unstuff_ms; acc[1] += t.huff_ms; acc[2] += t.idct_ms; acc[3] += t.color_ms; acc[6] += t.huff_spec_ms; acc[7] += t.huff_verify_ms; acc[8] += t.huff_scan_ms; acc[9] += t.huff_write_ms; rounds = std::max(rounds, t.verify_rounds); }
        std::vector<LpFrame> final_frames = frames;
        if (!fops.empty()) {
            if (eng.fused_resample(fops.data(), (int)fops.size())) return fail(eng.last_error());
            acc[4] += eng.timings().resize_ms;
            for (size_t q = 0; q < fops.size(); q++) final_frames[(size_t)fidx[q]] = fops[q].dst;
        }
        lap(2);
        // orientation (ops.go:392: unconditional) for the images that kept a frame
        std::vector<LpOrientOp> oops;
        std::vector<int> oidx;
        for (int k = 0; k < cnt; k++) {
            const LpJpeg& j = part.hdrs[first + k].j;
            if (st[(size_t)k] || j.orientation == 1 || !want[(size_t)k]) continue;
            LpOrientOp op;
            memset(&op, 0, sizeof(op));
            op.src = frames[(size_t)k];
            op.orientation = j.orientation;
            const bool swap = j.orientation >= 5;
            op.dst = op.src;
            op.dst.w = swap ? op.src.h : op.src.w;
            op.dst.h = swap ? op.src.w : op.src.h;
            op.dst.stride = op.dst.w * op.src.cn;
            uint8_t* p = eng.heap_alloc((size_t)op.dst.stride * op.dst.h);
            if (!p) return fail("frame heap exhausted");
            op.dst.off = (uint64_t)(uintptr_t)p;
            oops.push_back(op);
            oidx.push_back(k);
        }
        if (!oops.empty()) {
            if (eng.orient(oops.data(), (int)oops.size())) return fail(eng.last_error());
            for (size_t q = 0; q < oops.size(); q++) { frames[(size_t)oidx[q]] = oops[q].dst; final_frames[(size_t)oidx[q]] = oops[q].dst; }
        }
        // fit / resize through frames
        std::vector<LpResizeReq> rqs;
        std::vector<LpFrame> rdst;
        std::vector<int> ridx;
        for (int k = 0; k < cnt; k++) {
            if (st[(size_t)k] || !want[(size_t)k]) continue;
            const LpFrame& f = frames[(size_t)k];
            const LpOpsPlan& plan = plans[(size_t)k];
            if (!plan.resize) continue;
            LpResizeReq rq;
            rq.src = f;
            rq.crop_x = (uint32_t)plan.crop_x; rq.crop_y = (uint32_t)plan.crop_y; rq.crop_w = (uint32_t)plan.crop_w; rq.crop_h = (uint32_t)plan.crop_h;
            rq.dst_w = (uint32_t)plan.out_w; rq.dst_h = (uint32_t)plan.out_h;
            LpFrame d;
            memset(&d, 0, sizeof(d));
            uint8_t* p = eng.heap_alloc((size_t)plan.out_w * plan.out_h * f.cn);
            if (!p) return fail("frame heap exhausted");
            d.off = (uint64_t)(uintptr_t)p;
            rqs.push_back(rq);
            rdst.push_back(d);
            ridx.push_back(k);
        }
        if (!rqs.empty()) {
            std::vector<int> rst(rqs.size(), 0);
            if (eng.resize(rqs.data(), (int)rqs.size(), rdst.data(), rst.data()) == LP_ERR_DEVICE) return fail(eng.last_error());
            acc[4] += eng.timings().resize_ms;
            for (size_t q = 0; q < rqs.size(); q++) {
                if (rst[q]) st[(size_t)ridx[q]] = rst[q];
                else final_frames[(size_t)ridx[q]] = rdst[q];
            }
        }
        // encode (opencv.go:872-900)
        std::vector<LpEncodeReq> erq;
        std::vector<int> eidx;
        for (int k = 0; k < cnt; k++) {
            if (st[(size_t)k]) continue;
            LpEncodeReq e;
            e.src = final_frames[(size_t)k];
            e.quality = quality;
            e.out_cap = (size_t)e.src.w * e.src.h * 4 + 4096;
            erq.push_back(e);
            eidx.push_back(k);
        }
        if (!erq.empty()) {
            std::vector<int> est(erq.size(), 0);
            std::vector<uint32_t> elen(erq.size(), 0);
            std::vector<std::vector<uint8_t>> prog;
            if (opt->jpeg_progressive) { // EncodeOptions[JpegProgressive]: FDCT on the device, the scans on host threads
                if (eng.encode_jpegs_progressive(erq.data(), (int)erq.size(), est.data(), prog) == LP_ERR_DEVICE) return fail(eng.last_error());
                lap(3);
                lap(4);
            } else {
                if (eng.encode_jpegs(erq.data(), (int)erq.size(), est.data(), elen.data()) == LP_ERR_DEVICE) return fail(eng.last_error());
                acc[5] += eng.timings().encode_ms;
                lap(3);
                if (eng.encoded_fetch_all()) return fail(eng.last_error());
                lap(4);
            }
            for (size_t q = 0; q < erq.size(); q++) {
                const int k = eidx[q];
                const size_t item = (size_t)part.items[first + (size_t)k];
                if (est[q]) { st[(size_t)k] = est[q]; continue; }
                b->out_w[item] = (int)erq[q].src.w;
                b->out_h[item] = (int)erq[q].src.h;
                if (opt->jpeg_progressive) {
                    b->out_len[item] = (uint32_t)prog[q].size();
                    b->out_bytes[item].swap(prog[q]);
                } else {
                    b->out_len[item] = elen[q];
                    b->out_bytes[item].assign(eng.encoded_host((int)q), eng.encoded_host((int)q) + elen[q]);
                }
            }
        }
        for (int k = 0; k < cnt; k++) {
            const size_t item = (size_t)part.items[first + (size_t)k];
            if (st[(size_t)k]) b->status[item] = map_status(st[(size_t)k]);
        }
    }
    lap(5);
    eng.enable_timing(false);
    return LILLIPUT_OK;
}

// GIF and PNG items: Decoder + ImageOps.Transform of the Go-API mirror (for GIF the JPEG writer returns after the first composited frame)
static void run_other(LpBatch* b, const lilliput_batch_options* opt)
{
    if (b->other.empty()) return;
    const int enc_opts[4] = {CV_IMWRITE_JPEG_PROGRESSIVE, opt->jpeg_progressive ? 1 : 0, CV_IMWRITE_JPEG_QUALITY, opt->jpeg_quality};
    lilliput_image_options io;
    memset(&io, 0, sizeof(io));
    io.file_type = ".jpeg";
    io.width = opt->width; io.height = opt->height;
    io.resize_method = opt->resize_method;
    io.normalize_orientation = opt->normalize_orientation;
    io.encode_options = enc_opts;
    io.encode_options_len = opt->jpeg_quality ? 4 : 2;
    io.encode_timeout_ns = 30ll * 1000000000ll;
    // The host side of these sources is serial per image (inflate, LZW), so the items are spread over a few workers, each with its
    // own ImageOps (ops.go: "one ImageOps per goroutine") and its own per-thread engine on the batch's device.
    const size_t nw = std::min<size_t>(b->other.size(), (size_t)std::max(1, std::min(8, (int)std::thread::hardware_concurrency() / 8)));
    while (b->other_ops.size() < nw) b->other_ops.push_back(lilliput_new_image_ops(8192));
    std::atomic<size_t> next{0};
    auto worker = [&](size_t wi) {
        const int prev_dev = lp_thread_device(b->device);
        lilliput_image_ops ops = b->other_ops[wi];
        for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= b->other.size()) break;
            auto& it = b->other[k];
            const size_t i = (size_t)it.first;
            if (!ops) { b->status[i] = LILLIPUT_ERR_DEVICE; continue; }
            lilliput_decoder d = nullptr;
            int rc = lilliput_new_decoder(it.second.data(), it.second.size(), &d);
            if (!rc) {
                int w = 0, h = 0;
                (void)lilliput_decoder_header(d, &w, &h, nullptr, nullptr, nullptr, nullptr);
                std::vector<uint8_t>& out = b->out_bytes[i];
                out.resize((size_t)std::max(opt->width, 1) * std::max(opt->height, 1) * 3 + ((size_t)w * h * 3 + 65536));
                size_t n = 0;
                rc = lilliput_image_ops_transform(ops, d, &io, out.data(), out.size(), &n);
                if (!rc) {
                    out.resize(n);
                    b->out_len[i] = (uint32_t)n;
                    if (opt->resize_method == LILLIPUT_OPS_NO_RESIZE) { b->out_w[i] = w; b->out_h[i] = h; }
                    else if (opt->resize_method == LILLIPUT_OPS_FIT) lilliput_calculate_expected_size(w, h, opt->width, opt->height, &b->out_w[i], &b->out_h[i]);
                    else { b->out_w[i] = std::max(opt->width, 1); b->out_h[i] = std::max(opt->height, 1); }
                }
                lilliput_decoder_close(d);
            }
            b->status[i] = rc;
        }
        (void)lp_thread_device(prev_dev);
    };
    std::vector<std::thread> th;
    for (size_t wi = 1; wi < nw; wi++) th.emplace_back(worker, wi);
    worker(0);
    for (auto& t : th) t.join();
}

int lilliput_hip_batch_run(lilliput_hip_batch bb, const lilliput_batch_options* opt)
{
    auto b = static_cast<LpBatch*>(bb);
    if (!b || !opt) return LILLIPUT_ERR_DEVICE;
    const size_t n = b->n_items;
    b->status = b->parse_status;
    b->out_w.assign(n, 0);
    b->out_h.assign(n, 0);
    b->out_len.assign(n, 0);
    b->out_bytes.assign(n, std::vector<uint8_t>());
    // LILLIPUT_HIP_TRACE=1: host wall-clock per phase of a run (plan / decode / resample / encode / fetch / copy-out)
    const bool trace = getenv("LILLIPUT_HIP_TRACE") != nullptr;
    const auto t_run0 = std::chrono::steady_clock::now();
    size_t active = 0;
    for (auto& p : b->parts) active += p.items.empty() ? 0 : 1;
    if (active <= 1 && b->other.empty()) {
        for (auto& p : b->parts) p.rc = p.items.empty() ? LILLIPUT_OK : run_part(b, p, opt, trace);
    } else {
        std::vector<std::thread> th;
        for (auto& p : b->parts)
            if (!p.items.empty()) th.emplace_back([b, &p, opt, trace] { p.rc = run_part(b, p, opt, trace); });
            else p.rc = LILLIPUT_OK;
        run_other(b, opt);
        for (auto& t : th) t.join();
    }
    float acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t rounds = 0;
    int rc = LILLIPUT_OK;
    for (auto& p : b->parts) {
        if (p.rc) { rc = p.rc; lp_set_error(p.err); }
        if (p.items.empty()) continue;
        for (int i = 0; i < 10; i++) acc[i] += p.acc[i];
        rounds = std::max(rounds, p.rounds);
        if (trace)
            fprintf(stderr, "[lilliput_hip] part of %zu: plan %.2f ms, decode %.2f, resample %.2f, orient+resize+encode %.2f, fetch %.2f, copy-out %.2f | kernels: unstuff %.2f huff %.2f idct %.2f colour %.2f resize %.2f encode %.2f\n",
                    p.items.size(), p.tw[0], p.tw[1], p.tw[2], p.tw[3], p.tw[4], p.tw[5], p.acc[0], p.acc[1], p.acc[2], p.acc[3], p.acc[4], p.acc[5]);
    }
    if (trace)
        fprintf(stderr, "[lilliput_hip] run of %zu items: %.2f ms wall\n", n, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_run0).count());
    b->tm = LpTimings{acc[0], acc[1], acc[2], acc[3], acc[4], acc[5], rounds, acc[6], acc[7], acc[8], acc[9]}; // read by lilliput_hip_batch_timings
    return rc;
}

int lilliput_hip_batch_download(lilliput_hip_batch bb, lilliput_batch_item* items, size_t n)
{
    auto b = static_cast<LpBatch*>(bb);
    if (!b || n != b->n_items) return -1;
    int failed = 0;
    for (size_t i = 0; i < n; i++) {
        items[i].status = b->status[i];
        items[i].dst_len = 0;
        items[i].out_width = b->out_w[i];
        items[i].out_height = b->out_h[i];
        if (b->status[i] == LILLIPUT_OK) {
            if (b->out_len[i] > items[i].dst_cap || !items[i].dst) items[i].status = LILLIPUT_ERR_BUF_TOO_SMALL;
            else {
                memcpy(items[i].dst, b->out_bytes[i].data(), b->out_len[i]);
                items[i].dst_len = b->out_len[i];
            }
        }
        if (items[i].status) failed++;
    }
    return failed;
}

int lilliput_hip_batch_transform(lilliput_hip_batch b, lilliput_batch_item* items, size_t n, const lilliput_batch_options* opt)
{
    int rc = lilliput_hip_batch_upload(b, items, n);
    if (rc) { for (size_t i = 0; i < n; i++) { items[i].status = rc; items[i].dst_len = 0; } return (int)n; }
    rc = lilliput_hip_batch_run(b, opt);
    if (rc) { for (size_t i = 0; i < n; i++) { items[i].status = rc; items[i].dst_len = 0; } return (int)n; }
    return lilliput_hip_batch_download(b, items, n);
}

// ---- stage-level access for parity tests
static int decode_one(LpBatch* b, const void* src, size_t len, LpJpegHeader* h, LpFrame* f)
{
    int rc = lp_jpeg_parse((const uint8_t*)src, len, h);
    if (rc) return map_parse(rc);
    const LpJpeg& j = h->j;
    size_t fb = (size_t)j.width * j.height * (j.ncomp == 1 ? 1 : 3);
    if (!b->eng0().heap_reserve(fb + 4096)) return LILLIPUT_ERR_DEVICE;
    b->eng0().heap_reset();
    LpJpegSrc s{(const uint8_t*)src, len};
    memset(f, 0, sizeof(*f));
    int st = 0;
    rc = b->eng0().decode_jpegs(&s, 1, h, f, &st);
    if (rc || st) { lp_set_error(b->eng0().last_error()); return map_status(rc ? rc : st); }
    return LILLIPUT_OK;
}

int lilliput_hip_decode_jpeg(lilliput_hip_batch bb, const void* src, size_t len, void* dst, size_t cap, int* w, int* h, int* channels, int* orientation)
{
    auto b = static_cast<LpBatch*>(bb);
    LpJpegHeader hd;
    LpFrame f;
    int rc = decode_one(b, src, len, &hd, &f);
    if (rc) return rc;
    *w = (int)f.w; *h = (int)f.h; *channels = (int)f.cn; *orientation = hd.j.orientation;
    size_t nb = (size_t)f.stride * f.h;
    if (nb > cap) return LILLIPUT_ERR_BUF_TOO_SMALL;
    if (hipMemcpyAsync(dst, (const void*)(uintptr_t)f.off, nb, hipMemcpyDeviceToHost, b->eng0().stream()) != hipSuccess) return LILLIPUT_ERR_DEVICE;
    return b->eng0().sync() ? LILLIPUT_ERR_DEVICE : LILLIPUT_OK;
}

int lilliput_hip_decode_jpeg_coefs(lilliput_hip_batch bb, const void* src, size_t len, int comp, int16_t* dst, size_t cap_elems, int* bw, int* bh)
{
    auto b = static_cast<LpBatch*>(bb);
    LpJpegHeader hd;
    LpFrame f;
    int rc = decode_one(b, src, len, &hd, &f);
    if (rc) return rc;
    if (comp < 0 || comp >= hd.j.ncomp) return LILLIPUT_ERR_INVALID_IMAGE;
    *bw = (int)hd.j.bw[comp]; *bh = (int)hd.j.bh[comp];
    return map_status(b->eng0().copy_coefs(0, comp, dst, cap_elems));
}

int lilliput_hip_decode_jpeg_plane(lilliput_hip_batch bb, const void* src, size_t len, int comp, uint8_t* dst, size_t cap, int* pw, int* ph)
{
    auto b = static_cast<LpBatch*>(bb);
    LpJpegHeader hd;
    LpFrame f;
    int rc = decode_one(b, src, len, &hd, &f);
    if (rc) return rc;
    if (comp < 0 || comp >= hd.j.ncomp) return LILLIPUT_ERR_INVALID_IMAGE;
    *pw = (int)hd.j.bw[comp] * 8; *ph = (int)hd.j.bh[comp] * 8;
    return map_status(b->eng0().copy_plane(0, comp, dst, cap));
}

} // extern "C"
