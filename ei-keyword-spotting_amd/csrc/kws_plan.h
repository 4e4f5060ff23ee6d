// kws_plan.h -- host <-> device execution plans (internal to libkws_mi355x.so).
//
// All tables are built ONCE on the host, in the same precision and order the reference builds them per clip
// ("batch-first, table-once, fused"), uploaded to HBM, and referenced from these structs.
#pragma once
#include <stdlib.h>

// Development switches (environment variables KWS_DEV_*: occupancy experiments, A/B runs, forcing a tier in a test) exist only in a library
// built with -DKWS_DEV_SWITCHES (make dev -> libkws_mi355x_dev.so); the product library does not read the environment and does not
// contain their names.
#ifdef KWS_DEV_SWITCHES
#define KWS_DEV_ENV(name) getenv(name)
#else
#define KWS_DEV_ENV(name) ((const char *)nullptr)
#endif
#include <stdint.h>

struct KwsDspPlan {
    int n_samples;      // EI_CLASSIFIER_RAW_SAMPLE_COUNT (16000)
    int n_frames;       // speechpy numframes (49)                      processing.hpp:260-284
    int frame_stride;   // samples (320)
    int frame_len;      // samples fetched per frame (320); only min(frame_len, fft_len) are used
    int fft_len;        // 256
    int n_bins;         // fft_len/2+1
    int n_filters;      // 32
    int n_cepstral;     // 13
    int win_size;       // 101
    int pad;            // (win_size-1)/2
    int pre_shift;      // 1
    float pre_cof;      // 0.98
    float inv_fft;      // 1/fft_len (power of two => exact)
    float dct_s0, dct_s1;   // sqrtf(1/(4N)), sqrtf(1/(2N))                numpy.hpp:392-397
    int max_nz;         // longest mel filter (non-zero taps)
    int filt_nnz;       // non-zero weights of the whole filterbank (length of filt_bin / filt_w)
    // device tables
    const float2 *tw;        // [fft_len/2]  kiss_fft twiddles              kiss_fft.cpp:351-357
    const float2 *stw;       // [fft_len/4]  kiss_fftr super twiddles       kiss_fftr.cpp:52-58
    const int *filt_start;   // [n_filters+1] CSR over the transposed filterbank, ascending bin
    const int *filt_bin;
    const float *filt_w;
    const float2 *dct_tw;    // [n_filters/2]
    const float2 *dct_stw;   // [n_filters/4]
    const float *dct_cos;    // [n_filters/2+1] cosf((float)(i*pi/(2N)))    fast-dct-fft.cpp:71-74
    const float *dct_sin;
    const int *pad_map;      // [n_frames+2*pad] numpy::pad_1d_symmetric row map   numpy.hpp:479-541
    // configurations outside the tuned kernel's instantiations run on the general kernels (kws_generic.hip): kf_factor's factor
    // lists (kiss_fft.cpp:303-324: p, m pairs) of the frame transform (fft_len / 2 points) and of the DCT's (n_filters / 2 points)
    int generic;
    // generic = 1, but the SPECTRAL stage alone fits the tuned kernel's instantiations (fft 256, 32 / 40 filters, aligned frames, short mel filters) -- what
    // makes the plan general is its frame count or its cmvnw window: int16 windows then take kws_mfcc8_kernel over chunks of at most kws_mfcc_max_frames
    // frames (frames are independent but for pre-emphasis' predecessor sample, below) and only cmvnw runs on the general kernel (round 6; kws_api.cpp)
    int spectral_tuned;
    // kws_mfcc8_kernel, the window's first sample: its predecessor is x[wrap_index] of the window -- n_samples - 1, the reference's wrap
    // (processing.hpp:68, 104-106) -- unless the launch names another (a chunk that starts inside the window: -1, the sample before it)
    int wrap_index;
    int fft_levels, dct_levels;
    int fft_fac[24], dct_fac[24];
    // per launch, continuous mode: the rolling feature buffer of a stream is a ring over its first ring_rows rows (the rows behind
    // them are the reference's never-written tail).  Producers (WITH_CMVN = false, MFE rows): output row r goes to physical row
    // (ring_row0 + r) % ring_rows; consumers (kws_cmvn_nn_kernel, kws_unring_kernel): logical row i < ring_rows is physical row
    // (i + ring_head) % ring_rows.  ring_rows = 0: plain rows.
    int ring_rows, ring_row0, ring_head;
    // per launch, WITH_CMVN = false only: when set the kernel stops after speechpy::feature::mfe (feature.hpp:193-318)
    // and writes the mel energies [window][frame][filter] and frame energies [window][frame] (both after zero handling)
    float *mfe_mel, *mfe_energy;
};

// One "conv block" of the Edge Impulse 1-D CNN family:
//   RESHAPE -> CONV_2D(1xK) | DEPTHWISE_CONV_2D(1xK) -> RESHAPE -> [ADD(bias, ReLU)] -> RESHAPE -> [MAX_POOL_2D(P)] -> RESHAPE
#define KWS_MAX_BLOCKS 8
struct KwsConvBlock {
    int in_w, in_c, in_cpad;   // time steps, channels, channels padded to a multiple of 16
    int out_c;
    int taps, pad_left;        // filter width, SAME padding on the left
    int out_w;                 // conv output width (stride 1)
    int pool, pool_stride, pool_w;   // pool window, stride, pooled width
    int in_zp;                 // input zero point (= padding value so that (x + offset) == 0)
    int out_zp, act_min, act_max;    // conv output zero point and clamp
    int depthwise, depth_mult; // DEPTHWISE_CONV_2D: output channel oc reads input channel oc / depth_mult
    int has_lut;               // an ADD follows the convolution (add_lut is not the identity)
    int mfma;                  // generic kernel: this block's convolution runs on the matrix cores (CONV_2D, no pooling,
                               // in_cpad 16 / 32 / 64, out_c <= 32, out_w <= 64)
    int w_bytes;               // bytes of w
    const int8_t *w;           // conv: [out_c][taps][in_cpad], zero padded; depthwise: [out_c][taps padded to 4]
    const int32_t *bias_eff;   // [out_c] bias + input_offset * sum(w)
    const int32_t *mult;       // [out_c] per-channel quantized multiplier
    const int32_t *shift;      // [out_c]
    const int8_t *add_lut;     // [out_c][256] ADD(bias tensor)+ReLU folded: out = lut[c][x+128]
};

// FULLY_CONNECTED limits: input vector length, and weights per model (their LDS copy: bytes for int8, floats * 4 for float32)
#define KWS_FC_IN_MAX 1024
#define KWS_FC_W_MAX 32768

struct KwsNnPlan {
    int n_blocks;
    KwsConvBlock blk[KWS_MAX_BLOCKS];
    int n_features;            // 637
    int fc_in, fc_out;
    int fc_in_off, fc_w_off, fc_out_zp, fc_mult, fc_shift, fc_act_min, fc_act_max;
    const int8_t *fc_w;        // [fc_out][fc_in]
    const int32_t *fc_bias;    // [fc_out]
    // softmax (int8 -> int8): exp LUT indexed by (max - x) in [0,255]; 0 where diff < diff_min
    const int32_t *sm_exp;     // [256] exp_on_negative_values(rescaled diff), Q0.31
    const uint8_t *sm_valid;   // [256] diff >= diff_min
    float in_scale;            // input tensor quantisation (ei_run_classifier.h:436-444)
    int in_zp;
    float out_scale;           // output dequantisation
    int out_zp;
    int n_labels;
};

// ---- float32 models (the reference's float TFLite-Micro kernels: reference/conv.h:28-99, add.h:179-215,
//      pooling.h:189-237, fully_connected.h:26-60, softmax.h:31-63) -------------------------------------------
struct KwsConvBlockF32 {
    int in_w, in_c, out_c, taps, pad_left, out_w;
    int pool, pool_stride, pool_w;
    int has_add;
    int depthwise, depth_mult;     // DEPTHWISE_CONV_2D (filter [1][1][taps][out_c]); else CONV_2D (filter [out_c][1][taps][in_c])
    int tb, ob;                    // register blocking of the conv: time steps x output channels per lane
    int fused_pool;                // tb == pool == pool_stride: the lane max-pools its own window, no staging of the conv output
    unsigned inv_item20, inv_outc20;   // 2^20 / d + 1 for the kernel's item splits (d = channel blocks per time block, or out_c), 0 where items x d >= 2^20
    int ntb, rows;                 // set with the blocking (kws_nn_f32_pick_blocking): time blocks per channel block, rows of the zero-padded input image
                                   // -- derived values the kernel would otherwise divide for on every clip (a 32-bit division is ~35 vector instructions)
    float conv_min, conv_max;      // fused activation range of the convolution
    float add_min, add_max;        // fused activation range of the ADD (ReLU: [0, max])
    float pool_min, pool_max;
    const float *w;                // conv: [out_c][taps][in_c]; depthwise: [taps][out_c]
    const float *bias;             // [out_c]
    const float *addc;             // [out_c] constant operand of the ADD
};

struct KwsNnPlanF32 {
    int n_blocks;
    KwsConvBlockF32 blk[KWS_MAX_BLOCKS];
    int n_features, fc_in, fc_out, n_labels;
    float fc_min, fc_max, beta;
    const float *fc_w;             // [fc_out][fc_in]
    const float *fc_bias;          // [fc_out]
};
