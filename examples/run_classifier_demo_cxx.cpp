// run_classifier_demo_cxx.cpp -- a C++ application written for the SDK's DEFAULT signal_t (get_data is a std::function: the audio source
// is a lambda that captures its buffer, as an application without EIDSP_SIGNAL_C_FN_POINTER=1 would write it), served by libkws_mi355x.so.
//   g++ -std=c++14 -DKWS_SIGNAL_STD_FUNCTION -Iinclude examples/run_classifier_demo_cxx.cpp -L... -lkws_mi355x
// Prints the same lines as examples/run_classifier_demo.c's one-shot part.
#define KWS_SIGNAL_STD_FUNCTION
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "kws/ei_compat.h"
#include "kws/kws_synth.h"

int main(int argc, char **argv)
{
    const unsigned seed = argc > 1 ? (unsigned)atoi(argv[1]) : 1;
    std::vector<int16_t> audio(16000);
    kws_synth_fill(seed, 0, 1, (uint32_t)audio.size(), audio.data());
    signal_t signal;                                    // ei::signal_t: std::function member
    signal.total_length = audio.size();
    signal.get_data = [&audio](size_t offset, size_t length, float *out) {      // numpy::int16_to_float, as the demos' get_data does
        for (size_t i = 0; i < length; i++) out[i] = (float)audio[offset + i] / 32768.0f;
        return 0;
    };
    ei_impulse_result_t result = {};
    const EI_IMPULSE_ERROR r = run_classifier(&signal, &result);                // `debug` defaults to false, as in the SDK's C++ signature
    if (r != EI_IMPULSE_OK) {
        printf("ERR: Failed to run classifier (%d)\n", (int)r);
        return 1;
    }
    printf("Predictions (DSP: %d ms., Classification: %d ms.):\n", result.timing.dsp, result.timing.classification);
    for (int ix = 0; ix < EI_CLASSIFIER_LABEL_COUNT; ix++) printf("    %s: %.5f\n", result.classification[ix].label, result.classification[ix].value);
    return 0;
}
