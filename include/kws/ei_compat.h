/*
 * ei_compat.h -- the DROP-IN boundary: the Edge Impulse C SDK classifier API, implemented by
 * libkws_mi355x.so on an AMD MI355X instead of by the SDK's CPU code.
 *
 * Every declaration below replaces, with the same name, argument meaning and error behaviour, the
 * reference interface cited next to it (paths relative to
 * embedded-demos/stm32cubeide/nucleo-l476-keyword-spotting/ei-keyword-spotting/edge-impulse-sdk/).
 * An application written against the SDK (e.g. L476/Core/Src/main.cpp:190-199) keeps its source: it
 * includes this header instead of "edge-impulse-sdk/classifier/ei_run_classifier.h" and links
 * -lkws_mi355x.  Plain C, no HIP or torch types in any signature.
 */
#ifndef KWS_EI_COMPAT_H
#define KWS_EI_COMPAT_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The reference sizes ei_impulse_result_t with a compile-time constant of the exported model
 * (model-parameters/model_metadata.h:46).  Define EI_CLASSIFIER_LABEL_COUNT before including this
 * header to match the model you load; it defaults to the shipped no/noise/unknown/yes model. */
#ifndef EI_CLASSIFIER_LABEL_COUNT
#define EI_CLASSIFIER_LABEL_COUNT 4
#endif
#ifndef EI_CLASSIFIER_SLICES_PER_MODEL_WINDOW
#define EI_CLASSIFIER_SLICES_PER_MODEL_WINDOW 4          /* model_metadata.h:66-68 */
#endif

/* porting/ei_classifier_porting.h:34-43 (+ codes < -16 added for the GPU runtime) */
typedef enum {
    EI_IMPULSE_OK = 0,
    EI_IMPULSE_ERROR_SHAPES_DONT_MATCH = -1,
    EI_IMPULSE_CANCELED = -2,
    EI_IMPULSE_TFLITE_ERROR = -3,
    EI_IMPULSE_DSP_ERROR = -5,
    EI_IMPULSE_TFLITE_ARENA_ALLOC_FAILED = -6,
    EI_IMPULSE_CUBEAI_ERROR = -7,
    EI_IMPULSE_ALLOC_FAILED = -8,
    KWS_ERROR_NO_MODEL = -17,          /* no model loaded (reference: model is compiled in) */
    KWS_ERROR_UNSUPPORTED_MODEL = -18, /* graph / DSP config outside what the HIP kernels implement */
    KWS_ERROR_HIP = -19,               /* HIP runtime error, missing device or missing code object */
    KWS_ERROR_BAD_ARGUMENT = -20
} EI_IMPULSE_ERROR;

/* dsp/numpy_types.h:234-253 with EIDSP_SIGNAL_C_FN_POINTER=1: 16 bytes on x86-64.
 * get_data(offset, length, out) must write `length` floats starting at sample `offset`,
 * returns 0 on success; it is never asked for data beyond total_length.  It is asked what the reference's DSP block asks, in that order
 * (dsp/speechpy/processing.hpp:68, 86-94 under feature.hpp:263-281): the window's last sample, then per frame the sample before it and the
 * frame -- 98 calls for the shipped 49-frame window; an MFE block: its frames only (tests/golden/get_data_trace_l476.npz).
 * The SDK's DEFAULT in C++ is the other form (get_data is a std::function, 40 bytes with libstdc++): a C++ application written for
 * that form defines KWS_SIGNAL_STD_FUNCTION before including this header -- signal_t is then that class, the C structure is called
 * kws_c_signal_t, and inline overloads at the end of this header bridge the two (nothing in the library's ABI changes). */
#if defined(__cplusplus) && defined(KWS_SIGNAL_STD_FUNCTION)
typedef struct kws_c_signal {
    int (*get_data)(size_t, size_t, float *);
    size_t total_length;
} kws_c_signal_t;
#define KWS_C_SIGNAL_T kws_c_signal_t
#else
typedef struct ei_signal_t {
    int (*get_data)(size_t, size_t, float *);
    size_t total_length;
} signal_t;
#define KWS_C_SIGNAL_T signal_t
#endif

/* dsp/numpy_types.h:55-127 (ei::matrix_t data members; the C++ class adds ctor/dtor only) */
typedef struct ei_matrix {
    float *buffer;
    uint32_t rows;
    uint32_t cols;
    bool buffer_managed_by_me;
} ei_matrix_t;

/* classifier/ei_classifier_types.h:30-52 */
typedef struct {
    const char *label;
    float value;
} ei_impulse_result_classification_t;

typedef struct {
    int sampling;
    int dsp;
    int classification;
    int anomaly;
} ei_impulse_result_timing_t;

typedef struct {
    ei_impulse_result_classification_t classification[EI_CLASSIFIER_LABEL_COUNT];
    float anomaly;
    ei_impulse_result_timing_t timing;
} ei_impulse_result_t;

/* The result layout this translation unit is compiled for.  The reference compiles model and application together; here the
 * model is loaded at run time, so the library compares this number with the loaded model's label count before it writes into
 * an ei_impulse_result_t and returns EI_IMPULSE_ERROR_SHAPES_DONT_MATCH on a mismatch (a weak definition: one copy per program;
 * the library sees it when the application is linked against it). */
#ifndef KWS_BUILDING_LIBRARY
__attribute__((weak)) extern const int kws_app_label_count;
#ifdef __cplusplus
extern __attribute__((weak)) const int kws_app_label_count = EI_CLASSIFIER_LABEL_COUNT;      /* (a C++ const needs `extern` to be visible to the library) */
#else
__attribute__((weak)) const int kws_app_label_count = EI_CLASSIFIER_LABEL_COUNT;
#endif
#endif

typedef struct {
    uint32_t buf_idx;
    float running_sum;
    float maf_buffer[EI_CLASSIFIER_SLICES_PER_MODEL_WINDOW >> 1];
} ei_impulse_maf;

/* classifier/ei_run_classifier.h:650  -- DSP blocks + run_inference on one window of audio.
 * `debug` prints the features and per-class scores through ei_printf, as the reference does.
 * signal->total_length is normally the model's window; as in the reference (classifier/ei_run_dsp.h:277-286) a window of another length with
 * at least one and at most the model's number of frames is classified from the frames that fit, the rest of the network's input at zero
 * (tests/golden/other_length_l476.npz; an MFE block: mfe_other_length_l432.npz).  More frames, or none: EI_IMPULSE_DSP_ERROR -- the library's contract; the reference's
 * default build asserts there (dsp/config.hpp:65-67) and has no result to match.  A non-zero get_data return: EI_IMPULSE_DSP_ERROR. */
EI_IMPULSE_ERROR run_classifier(KWS_C_SIGNAL_T *signal, ei_impulse_result_t *result, bool debug);

/* classifier/ei_run_classifier.h:293  -- quantise, run the network, dequantise */
EI_IMPULSE_ERROR run_inference(ei_matrix_t *fmatrix, ei_impulse_result_t *result, bool debug);

/* classifier/ei_run_classifier.h:164, 184, 134 -- continuous (sliced) mode */
void run_classifier_init(void);
EI_IMPULSE_ERROR run_classifier_continuous(KWS_C_SIGNAL_T *signal, ei_impulse_result_t *result, bool debug);
float run_moving_average_filter(ei_impulse_maf *maf, float classification);

/* porting/ei_classifier_porting.h:45-76 -- platform hooks.  The reference requires the application
 * to define them; this library ships weak defaults (stdout printf, CLOCK_MONOTONIC, never cancelled)
 * that an application overrides simply by defining the symbol. */
EI_IMPULSE_ERROR ei_run_impulse_check_canceled(void);
EI_IMPULSE_ERROR ei_sleep(int32_t time_ms);
uint64_t ei_read_timer_ms(void);
uint64_t ei_read_timer_us(void);
void ei_printf(const char *format, ...);
void ei_printf_float(float f);

#ifdef __cplusplus
}
#endif

#if defined(__cplusplus) && defined(KWS_SIGNAL_STD_FUNCTION)
#include "ei_compat_cxx.hpp"      /* ei::signal_t with a std::function member + inline bridges onto the C entry points above */
#endif
#endif
