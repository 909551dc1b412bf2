// lp_launch.h -- host-callable launchers for the gfx950 kernels (defined in the .hip files).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "lp_types.h"

// decode
void lp_launch_unstuff(hipStream_t s, const LpJpeg* d_imgs, uint32_t nimg, uint32_t max_chunks, const uint8_t* d_raw, uint2* d_chunk_cnt,
                       LpJpegState* d_states, uint32_t* d_clean, uint32_t* d_rst);
// Everything the Huffman kernels share (device pointers; arrays indexed by LpJpeg::sub_off + subsequence).
struct LpHuffArgs {
    const LpJpeg* imgs;
    LpJpegState* states;
    const LpHuffSet* huffs;
    uint32_t nimg, max_sub, tot_sub;
    const uint32_t* clean;
    const uint32_t* rst;
    LpCkptPk* ckpts;            // [K][tot_sub]
    LpSubState* spec_exit;      // results of the speculative pass (immutable afterwards)
    LpSubSum* spec_total;
    LpSubState* cur_exit;       // current (verified) exit state / sums of every subsequence
    LpSubSum* cur_total;
    LpSubState* entry_used;     // entry state a subsequence was last verified against
    LpSubSum* prefix;            // exclusive scan of cur_total
    uint32_t* changed;          // [LP_VERIFY_ROUNDS + 1] counters, see lp_launch_huff_verify
    int8_t* coef8;              // 64 x int8 per block, decode order; -128 = escape (see DevSink in lp_kernels_decode.hip)
    int16_t* wide;              // wide slots: 64 x int16, only escaped positions are valid
    uint32_t* wide_id;          // block -> wide slot (valid for blocks holding an escape)
    int16_t* dc16;              // DC coefficient of every block
    LpCkSched sched;
};
void lp_launch_huff_spec(hipStream_t s, const LpHuffArgs& a);
uint32_t lp_huff_write_slots(); // WRITE workgroups resident on the current device at once
void lp_launch_huff_verify(hipStream_t s, const LpHuffArgs& a, uint32_t round); // counts into a.changed[round]; idle when changed[round - 1] == 0
#define LP_VERIFY_MAX 12        // ... at most (small launches with short subsequences may queue more: LpEngine::vr_; the counter array holds 16)
#define LP_VERIFY_ROUNDS 4      // rounds enqueued without looking (photographic streams settle in two); more only after a host check
void lp_launch_sub_scan(hipStream_t s, const LpHuffArgs& a);
void lp_launch_reset_tail_state(hipStream_t s, LpJpegState* d_states, uint32_t n);
void lp_launch_copy_small(hipStream_t s, void* dst, const void* src_pinned, size_t bytes); // bytes rounded up to 16: both buffers padded accordingly
// up to LP_SMALL_SEGS such copies and small zero fills as one launch (LpEngine::SmallBatch)
#define LP_SMALL_SEGS 6
struct LpSmallSeg { void* dst; const void* src; uint32_t n16; uint32_t tail; }; // src == nullptr: zero n16 * 16 + tail bytes; else copy n16 groups (tail unused)
struct LpSmallOps { LpSmallSeg s[LP_SMALL_SEGS]; };
void lp_launch_small_ops(hipStream_t s, const LpSmallOps& ops, uint32_t n);
void lp_launch_huff_write(hipStream_t s, const LpHuffArgs& a);
// zero-copy ingest: pieces in pinned, device-mapped host memory -> the raw arena (see k_gather_raw). The two arrays live in mapped pinned memory.
struct LpGatherPiece { const uint8_t* src; uint64_t dst_off; uint32_t len; uint32_t pad; };
void lp_launch_gather_raw(hipStream_t s, const LpGatherPiece* d_pcs, const uint32_t* d_tile_first, uint32_t npieces, uint8_t* d_arena, uint32_t workgroups);
uint32_t lp_dc_scan_max_ranges();
void lp_launch_dc_scan(hipStream_t s, const LpJpeg* d_imgs, uint32_t nimg, uint32_t max_mcus, int16_t* d_dc, void* d_partials /* nimg * lp_dc_scan_max_ranges() * 16 bytes */);
void lp_launch_idct(hipStream_t s, const LpJpeg* d_imgs, const LpJpegState* d_states, uint32_t nimg, uint32_t max_bw, uint32_t max_rows, const int8_t* d_coef8,
                    const int16_t* d_wide, const uint32_t* d_wide_id, const int16_t* d_dc, uint8_t* d_planes, uint32_t which /* 1 baseline, 2 progressive */,
                    const int16_t* d_pcoef);
// progressive scans [first, first + n) of d_scans (one dependency level): lane = scan, `lpw` lanes per 64-thread workgroup
// only_sequential: skip the progressive scans (lp_launch_prog_wave takes them)
void lp_launch_prog_scans(hipStream_t s, const LpProgScan* d_scans, uint32_t first, uint32_t n, uint32_t lpw, bool only_sequential, const LpJpeg* d_streams,
                          LpJpegState* d_stream_states, const LpProgHuff* d_huffs, const uint32_t* d_clean, const uint32_t* d_rst, int16_t* d_pcoef);
// the same level, one WAVE per scan (lp_kernels_prog.hip): the progressive scans; sequential ones (LpProgScan::sequential) are left to the lanes above
// d_progress != 0: PIPELINED -- scans [first, first + n) span every dependency level of the range, sorted by level; d_deps[i] names the scans
// scan first + i stays behind (indices relative to first), d_progress = n progress words + the ticket counter, all zero
void lp_launch_prog_wave(hipStream_t s, const LpProgScan* d_scans, uint32_t first, uint32_t n, const LpJpeg* d_streams, LpJpegState* d_stream_states,
                         const LpProgHuff* d_huffs, const uint32_t* d_clean, const uint32_t* d_rst, int16_t* d_pcoef, const LpProgDep* d_deps, uint32_t* d_progress);
// pixels
void lp_launch_ycc_to_frame(hipStream_t s, const LpJpeg* d_imgs, uint32_t nimg, uint32_t max_w, uint32_t max_h, bool any_generic, bool any_420,
                            const uint8_t* d_planes, const LpFrame* d_dsts, uint8_t* d_frames);
void lp_launch_resample_fused(hipStream_t s, const LpJpeg* d_imgs, const LpFusedOp* d_ops, uint32_t nops, uint32_t max_px, bool general,
                              uint32_t fast_mask, uint32_t fast_grid, const uint8_t* d_planes);
struct LpArea420Op;
void lp_launch_area_420(hipStream_t s, const LpJpeg* d_imgs, const LpArea420Op* d_ops, uint32_t nops, const uint32_t mask[3], uint32_t max_dw, uint32_t max_dh,
                        const LpTap* d_taps, const uint32_t* d_ranges, const uint8_t* d_planes);
void lp_launch_orient(hipStream_t s, const LpOrientOp* d_ops, uint32_t nimg, uint32_t max_w, uint32_t max_h, const uint8_t* d_src, uint8_t* d_dst);
void lp_launch_resize(hipStream_t s, const LpResizeOp* d_ops, uint32_t nimg, uint32_t modes_present, uint32_t area3_mask, uint32_t max_dw, uint32_t max_dh,
                      const LpTap* d_taps, const uint32_t* d_ranges, const uint8_t* d_src, uint8_t* d_dst);
void lp_launch_composite(hipStream_t s, const LpCompositeOp& op, const uint8_t* d_src, uint8_t* d_dst);
void lp_launch_gif_frame(hipStream_t s, const LpGifFrameOp& op);
void lp_launch_png(hipStream_t s, const LpPngOp& op);
void lp_launch_png_filter(hipStream_t s, const LpPngEncOp& op);
void lp_launch_gifenc(hipStream_t s, const LpGifEncOp& op);
void lp_launch_tonemap(hipStream_t s, const LpToneOp& op, int pass, uint32_t* n_wg); // pass 0..3, see LpToneOp
void lp_launch_webp_yuv420(hipStream_t s, const LpFrame& f, const LpWebpYuvTab* d_tab, uint8_t* d_y, uint8_t* d_u, uint8_t* d_v, uint32_t* d_flag);
void lp_launch_gather_samples(hipStream_t s, const LpFrame& f, const uint32_t* d_idx, uint32_t w, uint32_t h, uint8_t* d_out);
// encode
void lp_launch_encode(hipStream_t s, const LpEncJob* d_jobs, LpEncState* d_states, uint32_t nimg, uint32_t max_blocks, const uint8_t* d_frames,
                      int16_t* d_coef, uint32_t* d_blk_bits, uint32_t* d_bits, const uint8_t* d_hdrs, uint8_t* d_out);
// colour conversion + downsampling + FDCT + quantisation only (the coefficients go to the host for progressive output)
void lp_launch_enc_fdct(hipStream_t s, const LpEncJob* d_jobs, uint32_t nimg, uint32_t max_blocks, int16_t* d_coef);
void lp_launch_enc_pack(hipStream_t s, const LpEncJob* d_jobs, const LpEncState* d_states, uint32_t nimg, const uint32_t* d_pk_off, const uint8_t* d_out,
                        uint8_t* d_packed);
