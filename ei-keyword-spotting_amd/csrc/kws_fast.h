// kws_fast.h -- execution plan of KWS_MODE_FAST (internal to libkws_mi355x.so): the tolerance-mode form of the hot path.
//
// The exact kernels (kws_mfcc.hip, kws_nn_f32.hip) replay the reference's floating-point operation order and are bound by the
// instruction stream that order forces.  BASELINE.json's north_star grants 1e-4 on fp32 scores; this mode
// spends that tolerance where the reference's order is expensive and keeps it where a different order would be visible:
//   * the 256-point FFT still performs KissFFT's radix-4,4,4,2 butterflies with the reference's operation order (only their lane
//     changes; multiplications by a unit twiddle are skipped and power-of-two scales are folded, both exact): FFT round-off is
//     relative to the loudest bin, so any other factorisation moves the weak bins' log-mel energies by ~1e-4
//     (tools/fast_mode_study.py) -- not affordable;
//   * power spectrum re*re + im*im in fp32 (no fp64 sqrt-then-square), frame energy by a lane reduction, mel gather in any order;
//   * DCT-II as a [frames x NF] x [NF x NF/2+1] product on v_mfma_f32_16x16x4_f32 (true fp32);
//   * cmvnw with O(1) running sums per (row, column) on pivot-shifted data instead of two 101-term walks;
//   * float32 graphs: CONV_2D blocks -- the pointwise 1x1 halves of a depthwise-separable CNN included -- as
//     [time x taps*C] x [taps*C x out_c] on v_mfma_f32_16x16x4_f32 in the SAME launch, DEPTHWISE_CONV_2D taps on the vector ALU: the
//     feature matrix never leaves LDS (one wave = one clip from PCM to scores).
// Clips whose cmvnw is ill-conditioned (a near-constant column: the reference's answer there is decided by its own rounding
// sequence, SURVEY section 4 "silence canary") are detected, listed, and re-run by the exact kernels in the same call.
#pragma once
#include <stdint.h>

#include "kws_plan.h"

#define KWS_FAST_MAX_BLOCKS 8        // conv / depthwise / pointwise blocks of a fused float32 graph (= KWS_MAX_BLOCKS)
#define KWS_FAST_NZ_MAX 12        // longest mel filter (filters 0..31) kept in registers
#define KWS_FAST_NZ2 8            // longest of filters 32..39
#define KWS_FAST_DCT_GROUPS 5     // NF / 8 <= 5 k-groups of the DCT's operand fragments
#define KWS_FAST_MEL_CHUNK 8      // frames per pass of the spectral loop: eight lanes per frame; their power rows feed one mel pass
#define KWS_FAST_CMVN_EXT 8       // rows a cmvnw first window may count more often than the others (sparse form, see ext_off)
#define KWS_FAST_XS 144           // floats per frame of the FFT exchange buffer: 64 positions + 2 floats of padding per 8; = 16 mod 64
#define KWS_FAST_WAVE 64
#ifndef KWS_FAST_WPS
#define KWS_FAST_WPS 2            // waves per SIMD THIS compilation of kws_fast.hip is built for (registers via __launch_bounds__).  The library holds two:
                                  // 2 (every form) and 3 (kws_fast_w3.o: the float32-network forms at <= 168 registers, symbols *_w3); a plan says which
                                  // one runs it (KwsFastPlan::wps, kws_fast_plan.cpp)
#endif
#define KWS_FAST_ZF 320           // floats per in-place FFT buffer of the exact kernel (kws_device.h KWS_ZF)

struct KwsFastBlock {
    int in_w, in_c, in_cp;        // time steps, channels, channels padded to a multiple of 8 (the contraction's k-groups)
    int out_c, taps, pad_left, out_w;
    int pool, pool_stride, pool_w;
    int in_stride;                // floats per activation row of the block's input image in LDS (in_w rows: SAME padding is
                                  // applied to the operand registers, not stored)
    int m_tiles, n_tiles;         // 16-row / 16-channel output tiles (<= 4 x 2: every accumulator stays in registers)
    int vrows;                    // output rows beyond the tiles (out_w = 16 m_tiles + vrows, vrows <= 2) computed on the vector ALU: the
                                  // 49-frame window would otherwise pay a fourth row tile for one row
    int stage_stride;             // row stride of the un-pooled staging image (odd)
    int fpool;                    // 1: MAX_POOL_2D with non-overlapping windows of >= 4 rows is taken on the raw accumulators (register
                                  // maxima + LDS float-max atomics into [pool_w][32]); the epilogue then runs on the pooled values only
    int w_off, bias_off, addc_off;   // float offsets into the workgroup's shared LDS block
    int st_off;                   // k-step table [taps * in_cp / 8 + 3] x int4 { image offset (floats), tap, weight offset (floats), - }
    int has_add;
    int dw, mult;                 // DEPTHWISE_CONV_2D (reference/depthwiseconv_float.h:25): output channel n reads input channel n / mult; its taps run on
                                  // the vector ALU from the LDS image (fast_dwconv), weights [tap][out_c] at w_off
    float conv_min, conv_max, add_min, add_max, pool_min, pool_max;
    // reciprocals for the kernel's item -> (row, channel) splits, worked out once by the plan: a 32-bit division costs a wave ~35 vector
    // instructions even when every lane divides the same two numbers.  o_cp = the next block's in_cp (the last block's: its own out_c).
    unsigned inv_pool16;          // 2^16 / pool + 1:            r / pool for r < 64
    unsigned inv_ppr20;           // 2^20 / (in_cp / 2) + 1:     i / (in_cp / 2) for i < 1024
    unsigned inv_ocp20;           // 2^20 / o_cp + 1:            i / o_cp for i < 4096
    unsigned inv_npad20;          // 2^20 / max(o_cp - out_c, 1) + 1
    int dw_nseg, dw_seg_rows;     // un-pooled depthwise blocks: max(1, 64 / o_cp) segments of the time axis, rows per segment
    // ---- the contraction on v_mfma_f32_16x16x32_f16 (round 5; DESIGN.md 4.4: "split operands").  An fp32 value is carried as two
    //      halves, x s = hi + lo (s a power of two, hi = half(x s), lo = half(x s - hi): 22 significant bits), and a product as
    //      hi hi + hi lo + lo hi in an fp32 accumulator -- what is dropped is 2^-22 of a product, the size of fp32's own rounding of it --
    //      at sixteen times the rate of the fp32 matrix instruction.  K is walked in groups of eight channels of one tap
    //      (g = tap * in_cp / 8 + cg), four groups (one per lane / 16) to a k-step.
    int hconv;                    // 1: this block's contraction runs on split halves (CONV_2D blocks whose image fits the in-place split)
    int h_ks;                     // k-steps = ceil(taps * (in_cp / 8) / 4)
    int h_tab_off;                // shared LDS (floats): [h_ks + 2][4] x int2 { byte offset of the group's 16 bytes inside the image relative to
                                  // the lane's row 0 (tap * row bytes + 16 cg), tap } ; groups past the last one: tap = 1 << 20 (reads the zero block)
    int h_b_off;                  // shared LDS (floats) of the weight fragments [h_ks][2 (hi, lo)][n_tiles][64 lanes] x 16 bytes, or -1: they are
                                  // read from h_b_global (the LDS block is full)
    const void *h_b_global;       // the same fragments in device memory (always present)
    float h_inv_wscale;           // 1 / (the power of two the weights were multiplied by before the split)
};

struct KwsFastPlan {
    // ---- mel stage: power spectra of KWS_FAST_MEL_CHUNK frames, bins [bmin, bmin + nbins) only, then lane = filter
    int bmin, nbins, pstride;
    int nz;                       // longest of filters 0..31 (template parameter of the launch: 4, 8 or 12)
    int nf2p;                     // filters >= 32 are served nf2p (a power of two) per frame slot; 0: at most 32 filters
    // a filter's taps are consecutive bins: first bin as an offset into a frame's power row + weights (zero beyond the filter's end)
    const int *tap_start1;        // [64]                       filter lane & 31
    const float *tap_w1;          // [64][KWS_FAST_NZ_MAX]
    const int *tap_start2;        // [64]                       filter 32 + lane % nf2p
    const float *tap_w2;          // [64][KWS_FAST_NZ2]
    // ---- DCT on the matrix cores
    int dct_groups, dct_nt;       // NF / 8 k-groups, ceil((NF/2+1) / 16) output tiles
    int dct_off;                  // shared LDS: [dct_groups][2][dct_nt][64] B fragments: 2 cos(pi n (2k+1) / 2NF) * ortho scale
    float stale_scale;            // sqrtf(1/(2NF)): coefficients above NF/2 keep the log-mel input x 2 x this (fast-dct-fft.cpp:71)
    // ---- cmvnw
    int cr, cg;                   // rows per lane, columns per pass (13 x 16 or 17 x 20)
    int cnt_off, upd_off;         // shared LDS: cnt[64/cg][n_frames rounded up to 8] window multiplicities of each row group's first
                                  // window; upd[n_frames] = offset of the padded row leaving | entering << 16 (floats, image relative)
    int ext_off;                  // shared LDS, or -1: per row group [1 + 2 * KWS_FAST_CMVN_EXT] = base multiplicity m0 of its first window,
                                  // then (row offset as int bits, extra multiplicity) for the rows counted more than m0 times
    float inv_win;
    // The guard (kws_fast_plan.cpp: build_guard).  What the fp32 re-ordering moved in a cepstral coefficient, or in a
    // window's mean, comes out of cmvnw divided by the window's deviation; weighted with the graph's logit gain per column (kws_gain.cpp)
    // and summed in quadrature over the clip's windows it estimates the variance V of a logit difference's error:
    //     V = v_net + sum over windows (r, c) of ((g.abs + g.lev x level + g.rel x |mean|) / (deviation + eps))^2,
    // level = the clip's log-mel level: mean over the frames of |mean over the filters of the log-mel energies| (fp32 rounds relative to it).  The clip stays in the fast kernel iff
    //     V x max(g_c1 x P^2, g_c2) <= 1,     P = the largest p (1 - p) of its scores (1/4 where the network runs in another kernel):
    // g_c1 = (k sigma x margin / score tolerance)^2, g_c2 = (k sigma / largest logit error the linearisation is trusted for)^2.
    int guard_off;                // shared LDS: [n_cepstral rounded up to cg] x float4 { abs, lev, rel, alternative rel: column 0 with its means replayed; the other columns for a clip with digitally silent frames }
    int guard_cep_off;            // the same for windows that arrive as the exact kernels' cepstra (no spectral terms): continuous mode, second tier
    float v_net_feat;             // V of a window that arrives as the exact kernels' FEATURES (kws_fast_kernel's feat_in): the network's own arithmetic only
    float g_c1, g_c2, v_net, lvl_inv;   // lvl_inv = 1 / (n_frames x sqrt(filters)): level = lvl_inv x sum over the frames of |the DCT's coefficient 0|
    float c0_factor, c0_abs, c0_rel, c0_inv_rows;   // column 0: the exact window means are only computed when c0_factor x (plain deviation of
                                  // the column) < c0_abs + c0_rel x (its largest magnitude); c0_factor = sqrt(c0_mult n_frames / win_size),
                                  // c0_mult = how often every window holds every row at least (0: always compute them)
    int pad_off;                  // shared LDS: [n_frames + 2 pad] ints, numpy::pad_1d_symmetric's row order (column 0's exact window means)
    // Digitally silent frames (frame energy exactly 0: zero handling turns every mel energy into FLT_EPSILON) have ONE cepstral row in the
    // reference -- the DCT of a constant vector, a handful of rounding residues -- which kws_create records from the exact kernels on an
    // all-zero window (kws_fast_plan.cpp: record_silent_row).  The PCM forms write that row's DCT outputs over their own for such frames, so
    // those rows carry no spectral error: the guard's absolute / per-level terms are scaled by sqrt(live rows / rows) and its level is the
    // live rows' (round 6).  sil_off: shared LDS, 32 floats (columns 0 .. 31; only 1 .. NF/2 are read), or -1: not recorded
    int sil_off;
    float sys_t2;                 // (deviation / |mean|)^2 below which a column's window sums are taken to round systematically (its alternative rel coefficient)
    // ---- per-wave LDS: F = image [n_frames][fs] (log-mel -> cepstra -> features = block 0's input) + log energies [n_frames];
    //      R1 = the FFT's exchange buffer (reused for the eight power rows), later the other activation image
    int fs, f_floats, r1_floats, wave_floats, shared_floats, n_waves;
    int zero_off;                 // shared LDS (floats): 16 bytes of zeros, 16-byte aligned -- what an operand fetch outside an image reads (split-operand blocks)
    int sink_off;                 // F + sink_off + lane: where a lane's stores that fall outside an image go (no branch per value)
    int stash_off;                // F + stash_off: 48 floats that survive from one clip of a wave to its next (the paired tail pass)
    const float *shared_init;     // global image of the workgroup's shared LDS block (weights, biases, cmvnw tables)
    // ---- int8 two-block graph fused behind the features (kws_nn_int8_dev.h: nn_mfma_clip): qnet = bytes per activation row of its first
    //      block (16 / 64), 0 = none; q_floats = floats of the workgroup's shared LDS block its tables take, behind shared_floats
    int qnet, q_floats;
    int mfe;                      // 1: the model's DSP block is MFE -- the kernel stops after the mel filterbank (kws_fast_kernel<..., MFE>)
    // ---- float32 network fused behind the features (fuse = 0: features / int8 tensor go to HBM instead)
    int fuse, n_blocks;
    KwsFastBlock blk[KWS_FAST_MAX_BLOCKS];
    int fc_in, fc_out, fc_w_off, fc_b_off;
    float fc_min, fc_max, beta;
    int n_labels;
    // ---- (round 6; at the end: the offsets of everything above stay what the two-waves-per-SIMD forms were tuned with)
    int wps;                      // waves per SIMD this plan is laid out for: 2, or 3 = the build of kws_fast.hip with <= 168 registers (12 / 11 waves per workgroup,
                                  // 13-row cmvnw groups whatever the column count, twiddle table and one sink in the shared block, fragments that do not fit read from L2)
    // (wps = 3 only) shared LDS: the spectral pass loop's last-level and split twiddles per lane & 7 --
    // [8][12] x float2 { tw[k], tw[2 k], tw[3 k] } for k = fl + 8 a, then [8][8] x float2 super twiddles of the lane's eight bin pairs -- read per
    // pass instead of living in 40 registers (a register reloaded from scratch inside the pass loop waits for the pass's sample prefetch: one counter)
    int twl_off;
    // (wps = 3 only) two words in device memory, zero between launches: the ticket counter the kernel deals its clips out by, and the count of waves that
    // have left (kws_fast.hip: the clip loop and the kernel's end).  (A launch that does not run to its end -- a device fault -- leaves them dirty: like the rest of
    // the handle's device state, they are only good for launches that complete.)
    int *tickets;
};
