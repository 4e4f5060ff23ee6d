/*
 * examples/run_classifier_demo.c -- a plain-C application written against the Edge Impulse SDK classifier API
 * (the shape of L476/Core/Src/main.cpp:190-199 and :526-531) running on libkws_mi355x.so instead of the SDK.
 *
 *   cc -std=c11 -Iinclude examples/run_classifier_demo.c -Lei-keyword-spotting_amd -lkws_mi355x \
 *      -Wl,-rpath,$PWD/ei-keyword-spotting_amd -o run_classifier_demo
 *   KWS_MODEL=models/l476_no_yes.kwsm ./run_classifier_demo [seed [latency-iterations]]
 *
 * It classifies one synthetic 1 s clip in one-shot mode and then streams 2 s of audio through
 * run_classifier_continuous() in 250 ms slices, printing what the reference demo prints.
 */
#define _POSIX_C_SOURCE 199309L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "kws/ei_compat.h"
#include "kws/kws_synth.h"

#define RAW_SAMPLE_COUNT 16000
#define SLICE_SIZE (RAW_SAMPLE_COUNT / EI_CLASSIFIER_SLICES_PER_MODEL_WINDOW)

static int16_t audio[3 * RAW_SAMPLE_COUNT];
static const int16_t *window;      /* what get_audio_signal_data reads from */
static size_t window_len;

/* main.cpp:526-531: numpy::int16_to_float(&buffer[offset], out_ptr, length) */
static int get_audio_signal_data(size_t offset, size_t length, float *out_ptr)
{
    if (offset + length > window_len) return -1;
    for (size_t i = 0; i < length; i++) out_ptr[i] = (float)window[offset + i] / 32768;
    return 0;
}

static double now_us(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec * 1e6 + (double)ts.tv_nsec * 1e-3;
}

int main(int argc, char **argv)
{
    uint32_t seed = argc > 1 ? (uint32_t)atoi(argv[1]) : 1;
    int latency_iters = argc > 2 ? atoi(argv[2]) : 0;
    kws_synth_fill(seed, 0, 3, RAW_SAMPLE_COUNT, audio);

    signal_t signal;
    ei_impulse_result_t result;
    memset(&result, 0, sizeof(result));
    window = audio; window_len = RAW_SAMPLE_COUNT;
    signal.total_length = RAW_SAMPLE_COUNT;
    signal.get_data = &get_audio_signal_data;
    EI_IMPULSE_ERROR r = run_classifier(&signal, &result, false);
    if (r != EI_IMPULSE_OK) { printf("ERROR: Failed to run classifier (%d)\n", r); return 1; }
    printf("one-shot predictions (DSP: %d ms, NN: %d ms)\n", result.timing.dsp, result.timing.classification);
    for (size_t ix = 0; ix < EI_CLASSIFIER_LABEL_COUNT; ix++)
        printf("    %s: %.5f\n", result.classification[ix].label, result.classification[ix].value);

    if (latency_iters > 0) {          /* wall time of one drop-in call: callback gather + H2D + kernels + D2H */
        double t0 = now_us();
        for (int it = 0; it < latency_iters; it++) {
            r = run_classifier(&signal, &result, false);
            if (r != EI_IMPULSE_OK) { printf("ERROR: Failed to run classifier (%d)\n", r); return 1; }
        }
        printf("latency: run_classifier %.1f us per call (%d calls)\n", (now_us() - t0) / latency_iters, latency_iters);
    }

    run_classifier_init();
    for (int slice = 0; slice < 8; slice++) {
        window = audio + RAW_SAMPLE_COUNT + (size_t)slice * SLICE_SIZE; window_len = SLICE_SIZE;
        signal.total_length = SLICE_SIZE;               /* the reference demo resets it every iteration */
        signal.get_data = &get_audio_signal_data;
        memset(&result, 0, sizeof(result));
        r = run_classifier_continuous(&signal, &result, false);
        if (r != EI_IMPULSE_OK) { printf("ERROR: Failed to run classifier (%d)\n", r); return 1; }
        if (!result.classification[0].label) { printf("slice %d: filling the feature buffer\n", slice); continue; }
        printf("slice %d:", slice);
        for (size_t ix = 0; ix < EI_CLASSIFIER_LABEL_COUNT; ix++)
            printf("  %s %.5f", result.classification[ix].label, result.classification[ix].value);
        printf("\n");
    }
    if (latency_iters > 0) {
        double t0 = now_us();
        for (int it = 0; it < latency_iters; it++) {
            window = audio + RAW_SAMPLE_COUNT + (size_t)(it & 3) * SLICE_SIZE; window_len = SLICE_SIZE;
            signal.total_length = SLICE_SIZE;
            r = run_classifier_continuous(&signal, &result, false);
            if (r != EI_IMPULSE_OK) { printf("ERROR: Failed to run classifier (%d)\n", r); return 1; }
        }
        printf("latency: run_classifier_continuous %.1f us per 250 ms slice (%d calls)\n", (now_us() - t0) / latency_iters, latency_iters);
    }
    return 0;
}
