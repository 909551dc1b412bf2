// lp_launch.h -- host-callable launchers for the gfx950 kernels (defined in the .hip files).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "lp_types.h"

// decode
void lp_launch_unstuff(hipStream_t s, const LpJpeg* d_imgs, uint32_t nimg, uint32_t max_chunks, const uint8_t* d_raw, uint2* d_chunk_cnt,
                       LpJpegState* d_states, uint32_t* d_clean, uint32_t* d_rst, uint32_t S);
void lp_launch_huff_count(hipStream_t s, bool verify, const LpJpeg* d_imgs, const LpJpegState* d_states, const LpHuffSet* d_huffs,
                          uint32_t nimg, uint32_t max_sub, const uint32_t* d_clean, const uint32_t* d_rst, LpCkpt* d_ckpt,
                          LpSubState* d_exit, LpSubState* d_entry, LpSubSum* d_tot, uint32_t* d_changed, uint32_t S, uint32_t C, uint32_t K);
void lp_launch_sub_scan(hipStream_t s, const LpJpeg* d_imgs, LpJpegState* d_states, uint32_t nimg, const LpSubSum* d_tot, LpSubSum* d_prefix);
void lp_launch_huff_write(hipStream_t s, const LpJpeg* d_imgs, const LpJpegState* d_states, const LpHuffSet* d_huffs, uint32_t nimg,
                          uint32_t max_sub, const uint32_t* d_clean, const uint32_t* d_rst, const LpSubState* d_exit, const LpSubSum* d_prefix,
                          int16_t* d_coef, uint32_t S);
void lp_launch_idct(hipStream_t s, const LpJpeg* d_imgs, const LpJpegState* d_states, uint32_t nimg, uint32_t max_tiles, const int16_t* d_coef,
                    uint8_t* d_planes);
// pixels
void lp_launch_ycc_to_frame(hipStream_t s, const LpJpeg* d_imgs, uint32_t nimg, uint32_t max_w, uint32_t max_h, const uint8_t* d_planes,
                            const LpFrame* d_dsts, uint8_t* d_frames);
void lp_launch_resample_fused(hipStream_t s, const LpJpeg* d_imgs, const LpFusedOp* d_ops, uint32_t nops, uint32_t max_px, const uint8_t* d_planes);
void lp_launch_orient(hipStream_t s, const LpOrientOp* d_ops, uint32_t nimg, uint32_t max_w, uint32_t max_h, const uint8_t* d_src, uint8_t* d_dst);
void lp_launch_resize(hipStream_t s, const LpResizeOp* d_ops, uint32_t nimg, uint32_t modes_present, uint32_t max_dw, uint32_t max_dh,
                      const LpTap* d_taps, const uint32_t* d_ranges, const uint8_t* d_src, uint8_t* d_dst);
void lp_launch_composite(hipStream_t s, const LpCompositeOp& op, const uint8_t* d_src, uint8_t* d_dst);
// encode
void lp_launch_encode(hipStream_t s, const LpEncJob* d_jobs, LpEncState* d_states, uint32_t nimg, uint32_t max_blocks, const uint8_t* d_frames,
                      int16_t* d_coef, uint32_t* d_blk_bits, uint32_t* d_bits, const uint8_t* d_hdrs, uint8_t* d_out);
