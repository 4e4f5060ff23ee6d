/*
 * examples/run_classifier_demo.c -- a plain-C application written against the Edge Impulse SDK classifier API
 * (the shape of L476/Core/Src/main.cpp:190-199 and :526-531) running on libkws_mi355x.so instead of the SDK.
 *
 *   cc -std=c11 -Iinclude examples/run_classifier_demo.c -Lei-keyword-spotting_amd -lkws_mi355x \
 *      -Wl,-rpath,$PWD/ei-keyword-spotting_amd -o run_classifier_demo
 *   KWS_MODEL=models/l476_no_yes.kwsm ./run_classifier_demo [seed]
 *
 * It classifies one synthetic 1 s clip in one-shot mode and then streams 2 s of audio through
 * run_classifier_continuous() in 250 ms slices, printing what the reference demo prints.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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

int main(int argc, char **argv)
{
    uint32_t seed = argc > 1 ? (uint32_t)atoi(argv[1]) : 1;
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
    return 0;
}
