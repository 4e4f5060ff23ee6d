/*
 * oracle/kws_oracle.h -- TEST INFRASTRUCTURE, not product code.
 *
 * Plain-C11 CPU restatement of the reference's run_classifier() arithmetic
 * (Edge Impulse SDK MFCC DSP block + TFLite-Micro int8 reference kernels).
 * It is the checker the HIP path is compared against on the GPU box, where the
 * reference sources do not exist.  PARITY PINNED: tests/test_oracle_vs_reference.py
 * compares every function below, bit for bit, with the unmodified reference
 * compiled by oracle/Makefile (target `ref`), and tests/golden/ holds vectors
 * generated from that reference.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * Path shorthands in the citations:
 *   SDK/   = embedded-demos/stm32cubeide/nucleo-l476-keyword-spotting/ei-keyword-spotting/edge-impulse-sdk/
 *   MODEL/ = .../nucleo-l476-keyword-spotting/ei-keyword-spotting/
 *   TFL/   = SDK/tensorflow/lite/
 */
#ifndef KWS_ORACLE_H
#define KWS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* mirrors ei_dsp_config_mfcc_t (MODEL/model-parameters/model_metadata.h:89-101) + EI_CLASSIFIER_FREQUENCY */
typedef struct {
    int   num_cepstral;
    float frame_length;
    float frame_stride;
    int   num_filters;
    int   fft_length;
    int   win_size;
    int   low_frequency;
    int   high_frequency;   /* 0 => sampling_frequency/2 (feature.hpp:203-205) */
    float pre_cof;
    int   pre_shift;
    int   sampling_frequency;
    int   quantize_filterbank;  /* EIDSP_QUANTIZE_FILTERBANK (SDK/dsp/config.hpp:75-77; the SDK's default is 1, the demos build with 0): the
                                   triangle weights are snapped to numpy.hpp:52's table of fractions */
} kwso_mfcc_config;

/* ---- DSP leaves ------------------------------------------------------- */
float kwso_log(float a);                               /* SDK/dsp/numpy.hpp:1350-1371 */
float kwso_frequency_to_mel(float f);                  /* SDK/dsp/speechpy/functions.hpp:42-44 */
float kwso_mel_to_frequency(float mel);                /* functions.hpp:52-54 */
int   kwso_num_frames(size_t n, const kwso_mfcc_config *c);      /* processing.hpp:260-284 */
int   kwso_frame_length_samples(const kwso_mfcc_config *c);      /* processing.hpp:208 */
/* fb_t[coeff][num_filters] (transposed), coeff = fft_length/2+1        feature.hpp:54-171 */
int   kwso_filterbanks(const kwso_mfcc_config *c, float *fb_t);
float kwso_quantize_zero_one(float value);             /* numpy.hpp:423-468: dequantize_zero_one(quantize_zero_one(value)) */
/* y = pre-emphasised samples [offset, offset+length)                   processing.hpp:52-138 */
int   kwso_preemphasis(const int16_t *pcm, size_t n, float cof, int shift,
                       size_t offset, size_t length, float *out);
/* kiss_fftr restatement: complex spectrum, out_ri[(nfft/2+1)*2]        kissfft/kiss_fftr.cpp:66-120 */
int   kwso_rfft_complex(const float *in, int nfft, float *out_ri);
/* power spectrum of one frame (truncate / zero-pad to fft_length)      processing.hpp:295-312 */
int   kwso_power_spectrum(const float *frame, size_t frame_size, float *out, int fft_length);
/* mel energies [frames][num_filters] (after zero_handling) + energies  feature.hpp:193-318 */
int   kwso_mfe(const int16_t *pcm, size_t n, const kwso_mfcc_config *c, float *features, float *energies);
int   kwso_dct2_ortho(float *inout, int n);            /* numpy.hpp:378-401 + dct/fast-dct-fft.cpp:37-80 */
/* MFCC before CMVN [frames][num_cepstral]                              feature.hpp:370-439 */
int   kwso_mfcc_nocmvn(const int16_t *pcm, size_t n, const kwso_mfcc_config *c, float *out);
/* in-place windowed CMVN                                               processing.hpp:326-389 */
int   kwso_cmvnw(float *m, int rows, int cols, int win_size, int variance_normalization);
/* MFE block of the L432 SDK copy: cmvnw(..., scale) + numpy::normalize, extract_mfe_features */
int   kwso_normalize(float *m, size_t n);
int   kwso_cmvnw_scale(float *m, int rows, int cols, int win_size, int variance_normalization, int scale);
int   kwso_extract_mfe(const int16_t *pcm, size_t n, const kwso_mfcc_config *c, float *features);
/* extract_mfcc_features: features[frames*num_cepstral]                 classifier/ei_run_dsp.h:256-308 */
int   kwso_extract_mfcc(const int16_t *pcm, size_t n, const kwso_mfcc_config *c, float *features);

/* ---- fixed-point helpers (exported for unit tests) -------------------- */
int32_t kwso_srdhm(int32_t a, int32_t b);              /* gemmlowp fixedpoint.h:329-339 */
int32_t kwso_rdivpot(int32_t x, int exponent);         /* fixedpoint.h:357-368 */
int32_t kwso_mbqm(int32_t x, int32_t mult, int shift); /* TFL/kernels/internal/common.h:153-162 */
void    kwso_quantize_multiplier(double m, int32_t *q, int *shift); /* quantization_util.cc:53-91 */
int32_t kwso_exp_on_negative_values_q5_26(int32_t a);  /* fixedpoint.h:746-790 (tIntegerBits=5) */
int32_t kwso_one_over_one_plus_x(int32_t a);           /* fixedpoint.h:842-862 */

/* ---- model blob (tools/eon_import.py) and the NN ---------------------- */
typedef struct kwso_model kwso_model;
kwso_model *kwso_model_load(const void *blob, size_t nbytes);
void        kwso_model_free(kwso_model *m);
int         kwso_model_label_count(const kwso_model *m);
int         kwso_model_dsp_block(const kwso_model *m);   /* 0: MFCC block, 1: MFE block (L432 SDK copy) */
const char *kwso_model_label(const kwso_model *m, int i);
int         kwso_model_feature_count(const kwso_model *m);
int         kwso_model_raw_sample_count(const kwso_model *m);
void        kwso_model_mfcc_config(const kwso_model *m, kwso_mfcc_config *c);
int         kwso_model_tensor_count(const kwso_model *m);
int         kwso_model_tensor_bytes(const kwso_model *m, int id);

/* float features -> int8 input tensor        classifier/ei_run_classifier.h:436-444 */
void kwso_quantize_input(const kwso_model *m, const float *features, int8_t *q);
/* Run the graph from an int8 input.  If taps != NULL it must hold the concatenation of ALL
 * tensors (tensor id order, kwso_model_tensor_bytes each); every op output is copied there.
 * out_q[label_count] = int8 output tensor.    trained_model_compiled.cpp:457-465 + TFL kernels */
int  kwso_nn_invoke(const kwso_model *m, const int8_t *input_q, int8_t *out_q, int8_t *taps);
/* float32 models (TFLM float reference kernels): conv.h:28-99, add.h:179-215, pooling.h:189-237,
 * fully_connected.h:26-60, softmax.h:31-63 */
int  kwso_model_is_float(const kwso_model *m);
int  kwso_nn_invoke_f32(const kwso_model *m, const float *input, float *out, float *taps);
/* dequantise                                  ei_run_classifier.h:466-482 */
void kwso_dequantize_output(const kwso_model *m, const int8_t *out_q, float *scores);
/* run_inference = quantise + invoke + dequantise         ei_run_classifier.h:293-493 */
int  kwso_run_inference(const kwso_model *m, const float *features, float *scores);
/* run_classifier on one clip                              ei_run_classifier.h:650-714 */
int  kwso_run_classifier(const kwso_model *m, const int16_t *pcm, size_t n, float *scores);
/* batch form: pcm[B][n], scores[B][label_count]; features_out/q_out optional */
int  kwso_run_classifier_batch(const kwso_model *m, const int16_t *pcm, size_t n, size_t B,
                               float *scores, float *features_out, int8_t *q_out);
/* timing loop for bench.py cpu_baseline (kind "port"); returns seconds */
double kwso_time_run_classifier(const kwso_model *m, const int16_t *pcm, size_t n_clips, size_t n,
                                int iters, float *checksum);

/* continuous (sliced) mode: run_classifier_init / run_classifier_continuous, ei_run_classifier.h:164-282 */
typedef struct kwso_continuous kwso_continuous;
kwso_continuous *kwso_continuous_create(const kwso_model *m);
void kwso_continuous_free(kwso_continuous *s);
void kwso_continuous_init(kwso_continuous *s);
int  kwso_continuous_step(kwso_continuous *s, const int16_t *slice, size_t n, const float *end_of_signal, float *scores,
                          int *produced);

/* mix_audio (/root/reference/dataset-curation.py:93-137) for one clip, without the resampling of librosa.load: word (may be NULL)
 * of word_len float32 samples, noise window (may be NULL) of n samples -> PCM16.  PARITY UNPINNED: librosa / soundfile cannot be
 * imported where this was written; restated from the script's text. */
void kwso_mix_audio(const float *word, int word_len, const float *noise_window, float word_vol, float bg_vol, int n, int16_t *out);

/* synthetic test clips: include/kws/kws_synth.h */
void kwso_synth_fill(uint32_t seed, uint32_t first_clip, uint32_t n_clips, uint32_t clip_len, int16_t *out);

#ifdef __cplusplus
}
#endif
#endif
