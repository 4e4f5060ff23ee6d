/*
 * oracle/ref_driver.cpp -- TEST INFRASTRUCTURE, not product code.
 *
 * A thin driver that is compiled TOGETHER WITH the unmodified Edge Impulse SDK
 * sources where they lie under /root/reference (never copied into this repo) to
 * produce oracle/_ref/libei_ref_<model>.so.  It exposes, through a plain C ABI,
 *   - the reference's public entry point run_classifier() on an int16 clip,
 *   - taps on the public leaves it is made of (extract_mfcc_features,
 *     speechpy::feature::mfe / mfcc, numpy::log / dct2, processing::cmvnw,
 *     feature::filterbanks, trained_model_* and the per-op int8 tensors),
 *   - a timing loop used as bench.py's cpu_baseline (kind "reference").
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * the resulting library.  Build recipe: oracle/Makefile (target `ref`).
 *
 * Reference interfaces used (L476 copy):
 *   SDK/classifier/ei_run_classifier.h:650  run_classifier
 *   SDK/classifier/ei_run_classifier.h:293  run_inference
 *   SDK/classifier/ei_run_dsp.h:256         extract_mfcc_features
 *   SDK/dsp/speechpy/feature.hpp:54,193,370 filterbanks / mfe / mfcc
 *   SDK/dsp/speechpy/processing.hpp:326     cmvnw
 *   MODEL/tflite-model/trained_model_compiled.cpp:380-475 trained_model_*
 */
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>
#include <time.h>
#include <vector>

/* The generated model is included (not linked) so that the per-op tensors in
 * its anonymous namespace (tflTensors[]) can be tapped. */
#include "tflite-model/trained_model_compiled.cpp"
#include "edge-impulse-sdk/classifier/ei_run_classifier.h"

#include "edge-impulse-sdk/tensorflow/lite/kernels/internal/reference/conv.h"
#include "edge-impulse-sdk/tensorflow/lite/kernels/internal/reference/add.h"
#include "edge-impulse-sdk/tensorflow/lite/kernels/internal/reference/pooling.h"
#include "edge-impulse-sdk/tensorflow/lite/kernels/internal/reference/fully_connected.h"
#include "edge-impulse-sdk/tensorflow/lite/kernels/internal/reference/softmax.h"

using namespace ei;

/* ---- porting hooks (SDK/porting/ei_classifier_porting.h:45-76) ---------- */
static int g_print = 0;
/* boundary fixtures (tools/make_golden.py --only-debug-cancel): the text the reference prints with debug = true is captured instead of
 * going to stdout, and the cancellation hook the reference polls (ei_run_classifier.h:221, 489, 689) answers EI_IMPULSE_CANCELED on its
 * g_cancel_at-th call (1-based; 0 = never) */
static char *g_cap = NULL;
static size_t g_cap_size = 0, g_cap_len = 0;
static int g_cancel_at = 0, g_cancel_polls = 0;
EI_IMPULSE_ERROR ei_run_impulse_check_canceled() {
    g_cancel_polls++;
    return (g_cancel_at > 0 && g_cancel_polls == g_cancel_at) ? EI_IMPULSE_CANCELED : EI_IMPULSE_OK;
}
EI_IMPULSE_ERROR ei_sleep(int32_t) { return EI_IMPULSE_OK; }
uint64_t ei_read_timer_us() {
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return (uint64_t)ts.tv_sec * 1000000ull + ts.tv_nsec / 1000;
}
uint64_t ei_read_timer_ms() { return ei_read_timer_us() / 1000; }
void ei_printf(const char *format, ...) {
    va_list a;
    if (g_cap) {
        va_start(a, format);
        if (g_cap_len + 1 < g_cap_size) {
            const int n = vsnprintf(g_cap + g_cap_len, g_cap_size - g_cap_len, format, a);
            if (n > 0) g_cap_len = g_cap_len + (size_t)n < g_cap_size ? g_cap_len + (size_t)n : g_cap_size - 1;
        }
        va_end(a);
        return;
    }
    if (!g_print) return;
    va_start(a, format); vprintf(format, a); va_end(a);
}
void ei_printf_float(float f) { ei_printf("%f", f); }
void DebugLog(const char *s) { ei_printf("%s", s); }

/* ---- signal_t over an int16 clip (L476/Core/Src/main.cpp:526-531) ------- */
static const int16_t *g_pcm = NULL;
static size_t g_pcm_len = 0;
static size_t g_get_data_calls = 0;
/* (offset, length, return value) of every call, for the boundary tests: what a caller's callback sees, in order */
static long long *g_trace = NULL;
static int g_trace_cap = 0, g_trace_n = 0;
static int pcm_get_data(size_t offset, size_t length, float *out_ptr) {
    g_get_data_calls++;
    const int r = offset + length > g_pcm_len ? -1 : numpy::int16_to_float(g_pcm + offset, out_ptr, length);
    if (g_trace && g_trace_n < g_trace_cap) {
        g_trace[3 * g_trace_n] = (long long)offset; g_trace[3 * g_trace_n + 1] = (long long)length; g_trace[3 * g_trace_n + 2] = r;
    }
    if (g_trace) g_trace_n++;
    return r;
}

extern "C" {

int eiref_label_count(void) { return EI_CLASSIFIER_LABEL_COUNT; }
const char *eiref_label(int i) { return ei_classifier_inferencing_categories[i]; }
int eiref_feature_count(void) { return EI_CLASSIFIER_NN_INPUT_FRAME_SIZE; }
void eiref_set_print(int on) { g_print = on; }

/* copy of the DSP config the model was exported with (model_metadata.h:120) */
void eiref_get_mfcc_config(int *ints /*[8]*/, float *floats /*[3]*/) {
    ei_dsp_config_mfcc_t *c = (ei_dsp_config_mfcc_t *)ei_dsp_blocks[0].config;
    ints[0] = c->axes; ints[1] = c->num_cepstral; ints[2] = c->num_filters;
    ints[3] = c->fft_length; ints[4] = c->win_size; ints[5] = c->low_frequency;
    ints[6] = c->high_frequency; ints[7] = c->pre_shift;
    floats[0] = c->frame_length; floats[1] = c->frame_stride; floats[2] = c->pre_cof;
}

/* Full reference path. scores[LABEL_COUNT]; returns EI_IMPULSE_ERROR.
 * total_length_after / get_data call count are reported for the boundary tests. */
int eiref_run_classifier(const int16_t *pcm, size_t n, float *scores,
                         size_t *total_length_after, size_t *get_data_calls) {
    g_pcm = pcm; g_pcm_len = n; g_get_data_calls = 0;
    signal_t signal;
    signal.total_length = n;
    signal.get_data = &pcm_get_data;
    ei_impulse_result_t result;
    memset(&result, 0, sizeof(result));
    EI_IMPULSE_ERROR r = run_classifier(&signal, &result, g_print != 0);
    for (int i = 0; i < EI_CLASSIFIER_LABEL_COUNT; i++) scores[i] = result.classification[i].value;
    if (total_length_after) *total_length_after = signal.total_length;
    if (get_data_calls) *get_data_calls = g_get_data_calls;
    return (int)r;
}

/* Continuous (sliced) mode, SDK/classifier/ei_run_classifier.h:184-282.  One call = one slice of
 * EI_CLASSIFIER_SLICE_SIZE samples.  produced = 1 when the call ran inference (feature buffer full).
 * NOTE the reference's function-static `first_run` (ei_run_dsp.h:313) is never reset: only the very first call of
 * the PROCESS sees it false. */
void eiref_continuous_init(void) { run_classifier_init(); }
int eiref_continuous(const int16_t *slice, size_t n, float *scores, int *produced, size_t *total_length_after) {
    g_pcm = slice; g_pcm_len = n; g_get_data_calls = 0;
    signal_t signal;
    signal.total_length = n;
    signal.get_data = &pcm_get_data;
    ei_impulse_result_t result;
    memset(&result, 0, sizeof(result));
    EI_IMPULSE_ERROR r = run_classifier_continuous(&signal, &result, false);
    *produced = result.classification[0].label != NULL;
    for (int i = 0; i < EI_CLASSIFIER_LABEL_COUNT; i++) scores[i] = result.classification[i].value;
    if (total_length_after) *total_length_after = signal.total_length;
    return (int)r;
}
int eiref_slice_size(void) { return EI_CLASSIFIER_SLICE_SIZE; }

/* ---- boundary behaviour: debug text and the cancellation hook -------------------------------------------------------------
 * eiref_capture(buf, cap): ei_printf appends to buf (NULL: back to stdout / silence); eiref_capture_len() = bytes written so far.
 * eiref_cancel_at(n): the hook's n-th call from now on answers EI_IMPULSE_CANCELED (0: never); eiref_cancel_polls() = calls seen since.
 * The *_full entry points hand the caller's ei_impulse_result_t back byte for byte, pre-filled with 0xA5 so that what the reference
 * leaves untouched can be seen; labels[i] = 1 when classification[i].label points at the model's i-th category string. */
void eiref_capture(char *buf, size_t cap) { g_cap = buf; g_cap_size = cap; g_cap_len = 0; if (buf && cap) buf[0] = 0; }
size_t eiref_capture_len(void) { return g_cap_len; }
void eiref_cancel_at(int n) { g_cancel_at = n; g_cancel_polls = 0; }
int eiref_cancel_polls(void) { return g_cancel_polls; }
int eiref_result_size(void) { return (int)sizeof(ei_impulse_result_t); }
static void result_out(const ei_impulse_result_t &r, unsigned char *bytes, int *labels) {
    memcpy(bytes, &r, sizeof r);
    for (int i = 0; i < EI_CLASSIFIER_LABEL_COUNT; i++) labels[i] = r.classification[i].label == ei_classifier_inferencing_categories[i];
}
int eiref_run_classifier_full(const int16_t *pcm, size_t n, int debug, unsigned char *result_bytes, int *labels) {
    g_pcm = pcm; g_pcm_len = n; g_get_data_calls = 0;
    signal_t signal;
    signal.total_length = n;
    signal.get_data = &pcm_get_data;
    ei_impulse_result_t result;
    memset(&result, 0xA5, sizeof(result));
    EI_IMPULSE_ERROR r = run_classifier(&signal, &result, debug != 0);
    result_out(result, result_bytes, labels);
    return (int)r;
}
int eiref_run_inference_full(const float *features, int debug, unsigned char *result_bytes, int *labels) {
    matrix_t fm(1, EI_CLASSIFIER_NN_INPUT_FRAME_SIZE, (float *)features);
    ei_impulse_result_t result;
    memset(&result, 0xA5, sizeof(result));
    EI_IMPULSE_ERROR r = run_inference(&fm, &result, debug != 0);
    result_out(result, result_bytes, labels);
    return (int)r;
}
int eiref_continuous_full(const int16_t *slice, size_t n, int debug, unsigned char *result_bytes, int *labels) {
    g_pcm = slice; g_pcm_len = n; g_get_data_calls = 0;
    signal_t signal;
    signal.total_length = n;
    signal.get_data = &pcm_get_data;
    ei_impulse_result_t result;
    memset(&result, 0xA5, sizeof(result));
    EI_IMPULSE_ERROR r = run_classifier_continuous(&signal, &result, debug != 0);
    result_out(result, result_bytes, labels);
    return (int)r;
}
/* arm (buf != NULL) / disarm the get_data trace; eiref_trace_count = calls seen since it was armed (may exceed cap) */
void eiref_trace_get_data(long long *buf /*[3 * cap]*/, int cap) { g_trace = buf; g_trace_cap = cap; g_trace_n = 0; }
int eiref_trace_count(void) { return g_trace_n; }

/* extract_mfcc_features with an arbitrary ei_dsp_config_mfcc_t.
 * features must hold rows*num_cepstral floats; returns EIDSP code. */
int eiref_extract_mfcc(const int16_t *pcm, size_t n, int num_cepstral, float frame_length,
                       float frame_stride, int num_filters, int fft_length, int win_size,
                       int low_frequency, int high_frequency, float pre_cof, int pre_shift,
                       float *features, size_t features_cap) {
    g_pcm = pcm; g_pcm_len = n;
    signal_t signal;
    signal.total_length = n;
    signal.get_data = &pcm_get_data;
    ei_dsp_config_mfcc_t cfg = { 1, num_cepstral, frame_length, frame_stride, num_filters,
                                 fft_length, win_size, low_frequency, high_frequency,
                                 pre_cof, pre_shift };
    matrix_t fm(1, features_cap, features);
    return extract_mfcc_features(&signal, &fm, &cfg);
}

/* speechpy::feature::mfcc without CMVN (pre-emphasis as extract_mfcc_features does it) */
int eiref_mfcc_nocmvn(const int16_t *pcm, size_t n, int num_cepstral, float frame_length,
                      float frame_stride, int num_filters, int fft_length,
                      int low_frequency, int high_frequency, float pre_cof, int pre_shift,
                      float *out /*[rows*num_cepstral]*/) {
    g_pcm = pcm; g_pcm_len = n;
    signal_t signal;
    signal.total_length = n;
    signal.get_data = &pcm_get_data;
    class speechpy::processing::preemphasis pre(&signal, pre_shift, pre_cof);
    preemphasis = &pre;
    signal_t ps;
    ps.total_length = n;
    ps.get_data = &preemphasized_audio_signal_get_data;
    matrix_size_t sz = speechpy::feature::calculate_mfcc_buffer_size(
        n, EI_CLASSIFIER_FREQUENCY, frame_length, frame_stride, num_cepstral);
    matrix_t fm(sz.rows, sz.cols, out);
    return speechpy::feature::mfcc(&fm, &ps, EI_CLASSIFIER_FREQUENCY, frame_length, frame_stride,
                                   num_cepstral, num_filters, fft_length, low_frequency, high_frequency);
}

/* speechpy::feature::mfe: mel energies (after zero_handling, before log) + frame energies */
int eiref_mfe(const int16_t *pcm, size_t n, float frame_length, float frame_stride,
              int num_filters, int fft_length, int low_frequency, int high_frequency,
              float pre_cof, int pre_shift, float *out_features, float *out_energies) {
    g_pcm = pcm; g_pcm_len = n;
    signal_t signal;
    signal.total_length = n;
    signal.get_data = &pcm_get_data;
    class speechpy::processing::preemphasis pre(&signal, pre_shift, pre_cof);
    preemphasis = &pre;
    signal_t ps;
    ps.total_length = n;
    ps.get_data = &preemphasized_audio_signal_get_data;
    matrix_size_t sz = speechpy::feature::calculate_mfe_buffer_size(
        n, EI_CLASSIFIER_FREQUENCY, frame_length, frame_stride, num_filters);
    matrix_t fm(sz.rows, sz.cols, out_features);
    matrix_t em(sz.rows, 1, out_energies);
    return speechpy::feature::mfe(&fm, &em, &ps, EI_CLASSIFIER_FREQUENCY, frame_length, frame_stride,
                                  num_filters, fft_length, low_frequency, high_frequency);
}

/* pre-emphasised samples as the frame loop sees them (processing.hpp:52-138) */
int eiref_preemphasis(const int16_t *pcm, size_t n, float pre_cof, int pre_shift,
                      size_t offset, size_t length, float *out) {
    g_pcm = pcm; g_pcm_len = n;
    signal_t signal;
    signal.total_length = n;
    signal.get_data = &pcm_get_data;
    class speechpy::processing::preemphasis pre(&signal, pre_shift, pre_cof);
    return pre.get_data(offset, length, out);
}

int eiref_power_spectrum(float *frame, size_t frame_size, float *out, int fft_length) {
    return speechpy::processing::power_spectrum(frame, frame_size, out, fft_length / 2 + 1, fft_length);
}

/* filterbank in the transposed [coefficients][num_filters] layout mfe() uses */
int eiref_filterbanks(int num_filters, int fft_length, int low_frequency, int high_frequency, float *out) {
    int coefficients = fft_length / 2 + 1;
#if EIDSP_QUANTIZE_FILTERBANK
    /* the build with the SDK's default option (oracle/Makefile: libei_ref_l476_qfb.so): the matrix holds table indices; hand the
       de-quantised weights out, as numpy::dot_by_row reads them (numpy.hpp:222-250) */
    EI_DSP_QUANTIZED_MATRIX(fb, num_filters, coefficients, &numpy::dequantize_zero_one);
    if (!fb.buffer) return -1;
    int r = speechpy::feature::filterbanks(&fb, num_filters, coefficients, EI_CLASSIFIER_FREQUENCY, low_frequency, high_frequency, true);
    for (int i = 0; i < num_filters * coefficients; i++) out[i] = fb.dequantization_fn(fb.buffer[i]);
    return r;
#else
    matrix_t fb(num_filters, coefficients, out);
    memset(out, 0, sizeof(float) * num_filters * coefficients);
    return speechpy::feature::filterbanks(&fb, num_filters, coefficients, EI_CLASSIFIER_FREQUENCY,
                                          low_frequency, high_frequency, true);
#endif
}
int eiref_quantize_filterbank(void) { return EIDSP_QUANTIZE_FILTERBANK; }
/* numpy::quantize_zero_one / dequantize_zero_one (numpy.hpp:423-468) and the table behind them */
float eiref_quantize_zero_one(float v) { return numpy::dequantize_zero_one(numpy::quantize_zero_one(v)); }
int eiref_quantized_table(float *out, int cap) {
    const int n = (int)(sizeof(quantized_values_one_zero) / sizeof(float));
    for (int i = 0; i < n && i < cap; i++) out[i] = quantized_values_one_zero[i];
    return n;
}

float eiref_log(float x) { return numpy::log(x); }
float eiref_frequency_to_mel(float f) { return speechpy::functions::frequency_to_mel(f); }
float eiref_mel_to_frequency(float m) { return speechpy::functions::mel_to_frequency(m); }
int eiref_dct2_ortho(float *inout, size_t n) { return numpy::dct2(inout, n, DCT_NORMALIZATION_ORTHO); }
int eiref_cmvnw(float *inout, int rows, int cols, int win_size, int variance_normalization) {
    matrix_t m(rows, cols, inout);
    return speechpy::processing::cmvnw(&m, win_size, variance_normalization != 0);
}
int eiref_num_frames(size_t n, float frame_length, float frame_stride) {
    return speechpy::processing::calculate_no_of_stack_frames(n, EI_CLASSIFIER_FREQUENCY, frame_length,
                                                              frame_stride, false);
}
/* kiss_fftr tap: complex spectrum of a real frame (n_fft even) */
int eiref_rfft_complex(const float *in, size_t n_fft, float *out_ri /*[(n_fft/2+1)*2]*/) {
    return numpy::rfft(in, n_fft, (fft_complex_t *)out_ri, n_fft / 2 + 1, n_fft);
}

/* run_inference on a float feature vector (quantise + NN + dequantise) */
int eiref_run_inference(const float *features, float *scores) {
    matrix_t fm(1, EI_CLASSIFIER_NN_INPUT_FRAME_SIZE, (float *)features);
    ei_impulse_result_t result;
    memset(&result, 0, sizeof(result));
    EI_IMPULSE_ERROR r = run_inference(&fm, &result, false);
    for (int i = 0; i < EI_CLASSIFIER_LABEL_COUNT; i++) scores[i] = result.classification[i].value;
    return (int)r;
}

/* NN only, from a given int8 input tensor, with every op output tapped.
 * tensor_ids[k] selects tflTensors[] entries to copy out after invoke; because the
 * arena is shared between tensors, the taps are taken by re-running the graph node by
 * node.  out is a flat byte buffer, sizes[k] receives tensor bytes. */
int eiref_nn_taps(const int8_t *input, int n_taps, const int *tensor_ids, int8_t *out, int *sizes) {
    if (trained_model_init(ei_aligned_malloc) != kTfLiteOk) return -6;
    TfLiteTensor *in = trained_model_input(0);
    memcpy(in->data.int8, input, in->bytes);
    size_t off = 0;
    int rc = 0;
    const size_t n_nodes = sizeof(nodeData) / sizeof(nodeData[0]);
    /* sizes[] = -1 until produced */
    for (int k = 0; k < n_taps; k++) sizes[k] = -1;
    size_t *offs = (size_t *)malloc(sizeof(size_t) * n_taps);
    for (int k = 0; k < n_taps; k++) { offs[k] = off; off += tflTensors[tensor_ids[k]].bytes; }
    for (size_t i = 0; i < n_nodes; ++i) {
        if (registrations[nodeData[i].used_op_index].invoke(&ctx, &tflNodes[i]) != kTfLiteOk) { rc = -3; break; }
        int produced = tflNodes[i].outputs->data[0];
        for (int k = 0; k < n_taps; k++) {
            if (tensor_ids[k] == produced) {
                memcpy(out + offs[k], tflTensors[produced].data.raw, tflTensors[produced].bytes);
                sizes[k] = (int)tflTensors[produced].bytes;
            }
        }
    }
    free(offs);
    trained_model_reset(ei_aligned_free);
    return rc;
}

int eiref_tensor_count(void) { return (int)(sizeof(tensorData) / sizeof(tensorData[0])); }
int eiref_tensor_bytes(int id) { return (int)tensorData[id].bytes; }

/* ---- float32 TFLite-Micro reference kernels, called as leaves (no float model is shipped) ----
 * TFL/kernels/internal/reference/{conv.h:28-99, add.h:179-215, pooling.h:189-237, fully_connected.h:26-60,
 * softmax.h:31-63} */
void eiref_f32_conv(const float *in, int in_h, int in_w, int in_c, const float *flt, int out_c, int f_h, int f_w,
                    const float *bias, int pad_w, int pad_h, float amin, float amax, float *out, int out_h, int out_w) {
    tflite::ConvParams p;
    memset(&p, 0, sizeof(p));
    p.padding_values.width = pad_w; p.padding_values.height = pad_h;
    p.stride_width = 1; p.stride_height = 1; p.dilation_width_factor = 1; p.dilation_height_factor = 1;
    p.float_activation_min = amin; p.float_activation_max = amax;
    tflite::reference_ops::Conv(p, tflite::RuntimeShape({1, in_h, in_w, in_c}), in, tflite::RuntimeShape({out_c, f_h, f_w, in_c}), flt,
                                tflite::RuntimeShape({out_c}), bias, tflite::RuntimeShape({1, out_h, out_w, out_c}), out,
                                tflite::RuntimeShape(), nullptr);
}
void eiref_f32_add_bcast(const float *a, const int *da /*[4]*/, const float *b, const int *db /*[4]*/, const int *dout,
                         float amin, float amax, float *out) {
    tflite::ArithmeticParams p;
    memset(&p, 0, sizeof(p));
    p.float_activation_min = amin; p.float_activation_max = amax;
    tflite::reference_ops::BroadcastAdd4DSlow(p, tflite::RuntimeShape({da[0], da[1], da[2], da[3]}), a,
                                              tflite::RuntimeShape({db[0], db[1], db[2], db[3]}), b,
                                              tflite::RuntimeShape({dout[0], dout[1], dout[2], dout[3]}), out);
}
void eiref_f32_maxpool(const float *in, int in_h, int in_w, int c, int f_h, int f_w, int stride_h, int stride_w,
                       float amin, float amax, float *out, int out_h, int out_w) {
    tflite::PoolParams p;
    memset(&p, 0, sizeof(p));
    p.stride_height = stride_h; p.stride_width = stride_w; p.filter_height = f_h; p.filter_width = f_w;
    p.float_activation_min = amin; p.float_activation_max = amax;
    tflite::reference_ops::MaxPool(p, tflite::RuntimeShape({1, in_h, in_w, c}), in, tflite::RuntimeShape({1, out_h, out_w, c}), out);
}
void eiref_f32_fc(const float *in, int accum, const float *w, int out_d, const float *bias, float amin, float amax, float *out) {
    tflite::FullyConnectedParams p;
    memset(&p, 0, sizeof(p));
    p.float_activation_min = amin; p.float_activation_max = amax;
    tflite::reference_ops::FullyConnected(p, tflite::RuntimeShape({1, accum}), in, tflite::RuntimeShape({out_d, accum}), w,
                                          tflite::RuntimeShape({out_d}), bias, tflite::RuntimeShape({1, out_d}), out);
}
void eiref_f32_softmax(const float *in, int depth, float beta, float *out) {
    tflite::SoftmaxParams p;
    memset(&p, 0, sizeof(p));
    p.beta = beta;
    tflite::reference_ops::Softmax(p, tflite::RuntimeShape({1, depth}), in, tflite::RuntimeShape({1, depth}), out);
}

/* ---- a .kwsm graph (tools/eon_import.py layout) run through the reference's OWN TFLite-Micro op registrations ----------
 * The same harness MODEL/tflite-model/trained_model_compiled.cpp:380-465 is (TfLiteContext with
 * AllocatePersistentBuffer / RequestScratchBufferInArena / GetScratchBuffer, tensors + nodes tables, then
 * init -> prepare -> invoke of TFL/micro/kernels/{reshape,conv,depthwise_conv,add,pooling,fully_connected,softmax}.cc),
 * but driven by data, so that graphs the reference does not ship (float twins, depthwise graphs, other shapes) are still
 * evaluated by the reference's op code, Prepare-time arithmetic included.
 * input: the bytes of the graph's input tensor; taps (optional): every tensor, tensor-id order, concatenated. */
namespace graphrun {
struct Rd {
    const uint8_t *p, *end; bool bad;
    uint32_t u32() { if (p + 4 > end) { bad = true; return 0; } uint32_t v; memcpy(&v, p, 4); p += 4; return v; }
    int32_t i32() { return (int32_t)u32(); }
    float f32() { uint32_t u = u32(); float f; memcpy(&f, &u, 4); return f; }
    const uint8_t *bytes(size_t n) { size_t pn = (n + 3) & ~(size_t)3; if (p + pn > end) { bad = true; return nullptr; } const uint8_t *q = p; p += pn; return q; }
};
static std::vector<void *> allocs;
static void *keep(void *p) { allocs.push_back(p); return p; }
static TfLiteIntArray *int_array(const std::vector<int> &v) {
    TfLiteIntArray *a = (TfLiteIntArray *)keep(malloc(sizeof(int) * (v.size() + 1)));
    a->size = (int)v.size();
    for (size_t i = 0; i < v.size(); i++) a->data[i] = v[i];
    return a;
}
static TfLiteFloatArray *float_array(const std::vector<float> &v) {
    TfLiteFloatArray *a = (TfLiteFloatArray *)keep(malloc(sizeof(float) * (v.size() + 1)));
    a->size = (int)v.size();
    for (size_t i = 0; i < v.size(); i++) a->data[i] = v[i];
    return a;
}
static TfLiteStatus alloc_persistent(TfLiteContext *, size_t bytes, void **ptr) { *ptr = keep(calloc(1, bytes ? bytes : 1)); return *ptr ? kTfLiteOk : kTfLiteError; }
static std::vector<void *> scratch;
static TfLiteStatus request_scratch(TfLiteContext *c, size_t bytes, int *idx) {
    void *p; if (alloc_persistent(c, bytes, &p) != kTfLiteOk) return kTfLiteError;
    scratch.push_back(p); *idx = (int)scratch.size() - 1; return kTfLiteOk;
}
static void *get_scratch(TfLiteContext *, int idx) { return idx >= 0 && idx < (int)scratch.size() ? scratch[idx] : nullptr; }
static void report_error(TfLiteContext *, const char *fmt, ...) { va_list a; va_start(a, fmt); vfprintf(stderr, fmt, a); va_end(a); fputc('\n', stderr); }
}  // namespace graphrun

int eiref_graph_run(const uint8_t *blob, size_t nbytes, const void *input, size_t input_bytes, void *output, size_t output_bytes,
                    uint8_t *taps) {
    using namespace graphrun;
    if (nbytes < 8 || memcmp(blob, "KWSM", 4) != 0) return -1;
    Rd r{ blob + 4, blob + nbytes, false };
    const uint32_t version = r.u32();
    if (version != 1 && version != 2) return -1;
    const uint32_t nt = r.u32(), nn = r.u32(), nl = r.u32(), t_in = r.u32(), t_out = r.u32();
    for (int i = 0; i < 3 + 8 + 3 + (version == 2 ? 1 : 0); i++) r.u32();   /* sizes + dsp block: not needed here */
    for (uint32_t i = 0; i < nl; i++) { uint32_t len = r.u32(); r.bytes(len); }
    if (r.bad || nt > 4096 || nn > 4096 || t_in >= nt || t_out >= nt) return -1;
    allocs.clear(); scratch.clear();
    std::vector<TfLiteTensor> T(nt);
    memset(T.data(), 0, sizeof(TfLiteTensor) * nt);
    for (uint32_t i = 0; i < nt; i++) {
        TfLiteTensor &t = T[i];
        const uint32_t type = r.u32(), nd = r.u32();
        std::vector<int> dims;
        for (uint32_t k = 0; k < nd; k++) dims.push_back(r.i32());
        const bool is_const = r.u32() != 0;
        const uint32_t nq = r.u32();
        std::vector<float> sc; std::vector<int> zp;
        for (uint32_t k = 0; k < nq; k++) sc.push_back(r.f32());
        for (uint32_t k = 0; k < nq; k++) zp.push_back(r.i32());
        const int qdim = r.i32();
        const uint32_t nb = r.u32();
        if (r.bad) return -1;
        t.type = (TfLiteType)type;                                    /* 1 f32, 2 i32, 9 i8: TfLiteType's own values */
        t.dims = int_array(dims);
        t.bytes = nb;
        t.is_variable = 0;
        if (is_const) {
            const uint8_t *d = r.bytes(nb);
            if (!d) return -1;
            void *c = keep(malloc(nb ? nb : 1)); memcpy(c, d, nb);
            t.data.data = c; t.allocation_type = kTfLiteMmapRo;
        } else {
            t.data.data = keep(calloc(1, nb ? nb : 1)); t.allocation_type = kTfLiteArenaRw;
        }
        if (nq) {
            TfLiteAffineQuantization *q = (TfLiteAffineQuantization *)keep(malloc(sizeof(TfLiteAffineQuantization)));
            q->scale = float_array(sc); q->zero_point = int_array(zp); q->quantized_dimension = qdim;
            t.quantization.type = kTfLiteAffineQuantization; t.quantization.params = q;
            t.params.scale = sc[0]; t.params.zero_point = zp[0];
        } else {
            t.quantization.type = kTfLiteNoQuantization; t.quantization.params = nullptr;
        }
    }
    TfLiteContext c;
    memset(&c, 0, sizeof(c));
    c.AllocatePersistentBuffer = &alloc_persistent;
    c.RequestScratchBufferInArena = &request_scratch;
    c.GetScratchBuffer = &get_scratch;
    c.ReportError = &report_error;
    c.tensors = T.data();
    c.tensors_size = nt;
    std::vector<TfLiteNode> N(nn);
    std::vector<TfLiteRegistration> reg(nn);
    memset(N.data(), 0, sizeof(TfLiteNode) * nn);
    int rc = 0;
    for (uint32_t i = 0; i < nn && !rc; i++) {
        const uint32_t op = r.u32(), ni = r.u32();
        std::vector<int> in, out;
        for (uint32_t k = 0; k < ni; k++) in.push_back(r.i32());
        const uint32_t no = r.u32();
        for (uint32_t k = 0; k < no; k++) out.push_back(r.i32());
        int p[8];
        for (int k = 0; k < 8; k++) p[k] = r.i32();
        const float beta = r.f32();
        if (r.bad) { rc = -1; break; }
        N[i].inputs = int_array(in); N[i].outputs = int_array(out);
        switch (op) {
        case 0: { reg[i] = *tflite::ops::micro::Register_RESHAPE();
                  N[i].builtin_data = keep(calloc(1, sizeof(TfLiteReshapeParams))); break; }
        case 1: { reg[i] = *tflite::ops::micro::Register_CONV_2D();
                  TfLiteConvParams *q = (TfLiteConvParams *)keep(calloc(1, sizeof(TfLiteConvParams)));
                  q->padding = (TfLitePadding)p[0]; q->stride_width = p[1]; q->stride_height = p[2]; q->activation = (TfLiteFusedActivation)p[3];
                  q->dilation_width_factor = p[4]; q->dilation_height_factor = p[5]; N[i].builtin_data = q; break; }
        case 6: { reg[i] = *tflite::ops::micro::Register_DEPTHWISE_CONV_2D();
                  TfLiteDepthwiseConvParams *q = (TfLiteDepthwiseConvParams *)keep(calloc(1, sizeof(TfLiteDepthwiseConvParams)));
                  q->padding = (TfLitePadding)p[0]; q->stride_width = p[1]; q->stride_height = p[2]; q->activation = (TfLiteFusedActivation)p[3];
                  q->dilation_width_factor = p[4]; q->dilation_height_factor = p[5]; q->depth_multiplier = p[6]; N[i].builtin_data = q; break; }
        case 2: { reg[i] = *tflite::ops::micro::Register_ADD();
                  TfLiteAddParams *q = (TfLiteAddParams *)keep(calloc(1, sizeof(TfLiteAddParams)));
                  q->activation = (TfLiteFusedActivation)p[0]; N[i].builtin_data = q; break; }
        case 3: { reg[i] = *tflite::ops::micro::Register_MAX_POOL_2D();
                  TfLitePoolParams *q = (TfLitePoolParams *)keep(calloc(1, sizeof(TfLitePoolParams)));
                  q->padding = (TfLitePadding)p[0]; q->stride_width = p[1]; q->stride_height = p[2]; q->filter_width = p[3];
                  q->filter_height = p[4]; q->activation = (TfLiteFusedActivation)p[5]; N[i].builtin_data = q; break; }
        case 4: { reg[i] = *tflite::ops::micro::Register_FULLY_CONNECTED();
                  TfLiteFullyConnectedParams *q = (TfLiteFullyConnectedParams *)keep(calloc(1, sizeof(TfLiteFullyConnectedParams)));
                  q->activation = (TfLiteFusedActivation)p[0]; N[i].builtin_data = q; break; }
        case 5: { reg[i] = *tflite::ops::micro::Register_SOFTMAX();
                  TfLiteSoftmaxParams *q = (TfLiteSoftmaxParams *)keep(calloc(1, sizeof(TfLiteSoftmaxParams)));
                  q->beta = beta; N[i].builtin_data = q; break; }
        default: rc = -2;
        }
    }
    for (uint32_t i = 0; i < nn && !rc; i++)
        if (reg[i].init) N[i].user_data = reg[i].init(&c, (const char *)N[i].builtin_data, 0);
    for (uint32_t i = 0; i < nn && !rc; i++)
        if (reg[i].prepare && reg[i].prepare(&c, &N[i]) != kTfLiteOk) rc = -3;
    if (!rc) {
        if (input_bytes != T[t_in].bytes || output_bytes != T[t_out].bytes) rc = -4;
        else memcpy(T[t_in].data.data, input, input_bytes);
    }
    for (uint32_t i = 0; i < nn && !rc; i++)
        if (reg[i].invoke(&c, &N[i]) != kTfLiteOk) rc = -5;
    if (!rc) {
        memcpy(output, T[t_out].data.data, output_bytes);
        if (taps) { size_t off = 0; for (uint32_t i = 0; i < nt; i++) { memcpy(taps + off, T[i].data.data, T[i].bytes); off += T[i].bytes; } }
    }
    for (void *p : allocs) free(p);
    allocs.clear(); scratch.clear();
    return rc;
}

/* CPU baseline: loop run_classifier over n_clips clips ([n_clips][len] int16), `iters` passes.
 * Returns seconds of wall time; checksum defeats dead-code elimination. */
double eiref_time_run_classifier(const int16_t *pcm, size_t n_clips, size_t len, int iters, float *checksum) {
    float acc = 0.f;
    uint64_t t0 = ei_read_timer_us();
    float scores[EI_CLASSIFIER_LABEL_COUNT];
    for (int it = 0; it < iters; it++) {
        for (size_t c = 0; c < n_clips; c++) {
            eiref_run_classifier(pcm + c * len, len, scores, NULL, NULL);
            acc += scores[0];
        }
    }
    uint64_t t1 = ei_read_timer_us();
    if (checksum) *checksum = acc;
    return (double)(t1 - t0) * 1e-6;
}

/* CPU baseline for a graph the reference does not ship (BASELINE configs 2, 3, 5): per clip, the reference's
 * extract_mfcc_features() with the blob's DSP settings, (float graphs) the feature copy of ei_run_classifier.h:447-452 or
 * (int8 graphs) its quantisation loop :436-444, then the graph through the reference's op registrations -- set up and torn
 * down per clip as run_inference does with trained_model_init / trained_model_reset (:341-345, :486-488). */
double eiref_time_graph_classifier(const uint8_t *blob, size_t nbytes, const int16_t *pcm, size_t n_clips, size_t len, int iters,
                                   float *checksum) {
    if (nbytes < 8 + 4 * (5 + 3 + 11) || memcmp(blob, "KWSM", 4) != 0) return -1.0;
    const uint8_t *q = blob + 8;
    uint32_t hdr[8]; memcpy(hdr, q, sizeof(hdr)); q += sizeof(hdr);              /* nt nn nl t_in t_out raw freq nfeat */
    int32_t d[8]; memcpy(d, q, sizeof(d)); q += sizeof(d);                        /* axes ncep nfilt fft win low high shift */
    float f[3]; memcpy(f, q, sizeof(f));                                          /* frame_length frame_stride pre_cof */
    const uint32_t t_in = hdr[3], t_out = hdr[4], nfeat = hdr[7], nl = hdr[2];
    if (nl == 0 || nl > 64) return -1.0;
    /* input tensor type / quantisation: walk to tensor t_in (labels first) */
    graphrun::Rd r{ blob + 8 + 4 * (8 + 8 + 3), blob + nbytes, false };
    for (uint32_t i = 0; i < nl; i++) { uint32_t n = r.u32(); r.bytes(n); }
    uint32_t in_type = 0; float in_scale = 1.f; int in_zp = 0; float out_scale = 1.f; int out_zp = 0;
    for (uint32_t i = 0; i <= (t_in > t_out ? t_in : t_out); i++) {
        uint32_t type = r.u32(), nd = r.u32();
        for (uint32_t k = 0; k < nd; k++) r.i32();
        uint32_t is_const = r.u32(), nq = r.u32();
        float sc0 = 1.f; int zp0 = 0;
        for (uint32_t k = 0; k < nq; k++) { float v = r.f32(); if (k == 0) sc0 = v; }
        for (uint32_t k = 0; k < nq; k++) { int v = r.i32(); if (k == 0) zp0 = v; }
        r.i32(); uint32_t nb = r.u32();
        if (is_const) r.bytes(nb);
        if (i == t_in) { in_type = type; in_scale = sc0; in_zp = zp0; }
        if (i == t_out) { out_scale = sc0; out_zp = zp0; }
    }
    if (r.bad) return -1.0;
    std::vector<float> feat(nfeat), scores(nl);
    std::vector<int8_t> qin(nfeat), qout(nl);
    float acc = 0.f;
    uint64_t t0 = ei_read_timer_us();
    for (int it = 0; it < iters; it++)
        for (size_t c = 0; c < n_clips; c++) {
            if (eiref_extract_mfcc(pcm + c * len, len, d[1], f[0], f[1], d[2], d[3], d[4], d[5], d[6], f[2], d[7], feat.data(), nfeat) != 0) return -2.0;
            int rc;
            if (in_type == 1) {
                rc = eiref_graph_run(blob, nbytes, feat.data(), nfeat * 4, scores.data(), nl * 4, nullptr);
            } else {
                for (uint32_t ix = 0; ix < nfeat; ix++) qin[ix] = static_cast<int8_t>(round(feat[ix] / in_scale) + in_zp);
                rc = eiref_graph_run(blob, nbytes, qin.data(), nfeat, qout.data(), nl, nullptr);
                for (uint32_t ix = 0; ix < nl; ix++) scores[ix] = (float)(qout[ix] - out_zp) * out_scale;
            }
            if (rc != 0) return -3.0;
            acc += scores[0];
        }
    uint64_t t1 = ei_read_timer_us();
    if (checksum) *checksum = acc;
    return (double)(t1 - t0) * 1e-6;
}

} /* extern "C" */
