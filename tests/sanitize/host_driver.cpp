// host_driver.cpp -- TEST INFRASTRUCTURE ONLY (tests/test_sanitizers.py): feeds .kwsm blobs (the shipped ones and mutated /
// truncated copies written by the test) to the HOST code of the library, built with -fsanitize=address,undefined against
// hip_stub.cpp.  A blob the parser and the plan builders accept is then walked through the C ABI's host logic (argument checks,
// scratch growth, the two-stream host pipeline, the continuous-mode bookkeeping, the SDK entry points); kernels do not run (see
// hip_stub.cpp), so no output value means anything -- the run is about memory safety and undefined behaviour only.
// One line per blob: "<path> rc <kws_create's return code>".  Exit status 0 unless a sanitizer aborts the process.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/kws/kws.h"

static std::vector<float> g_audio;
static bool g_tracing = false;
static int get_data(size_t off, size_t len, float *out)
{
    const int r = off + len > g_audio.size() ? -1 : 0;
    if (g_tracing) printf(" %zu %zu %d", off, len, r);
    if (r) return r;
    memcpy(out, g_audio.data() + off, len * sizeof(float));
    return 0;
}

static void walk(kws_handle *h)
{
    const size_t n = (size_t)kws_clip_samples(h), F = (size_t)kws_feature_count(h), C = (size_t)kws_label_count(h);
    for (int i = 0; i < (int)C; i++) (void)kws_label(h, i);
    (void)kws_nn_kernel_name(h); (void)kws_mfcc_kernel_name(h); (void)kws_frame_count(h); (void)kws_filter_count(h);
    const size_t B = 5;
    std::vector<int16_t> pcm(B * n + 8, 1);
    std::vector<float> scores(B * C + 1), feats(B * F + 1), thr(4 * 256), rel(256);
    std::vector<int8_t> q(B * F + 1);
    const bool is_float = kws_model_is_float(h) != 0;
    for (int mode = 0; mode < 2; mode++) {
        if (kws_set_mode(h, mode) != EI_IMPULSE_OK) continue;
        (void)kws_fast_guard(h, 1, thr.data());
        (void)kws_fast_guard(h, 2, thr.data());
        (void)kws_fast_gain(h, rel.data());
        kws_fast_tolerance tolinfo;
        (void)kws_fast_tolerance_info(h, &tolinfo);
        // "device" pointers are host heap under the stub: the entry points' host logic runs, launches are no-ops
        (void)kws_run_classifier_batch_device(h, pcm.data(), B, scores.data(), feats.data(), is_float ? nullptr : q.data(), nullptr);
        (void)kws_run_classifier_batch_device(h, pcm.data(), B, scores.data(), nullptr, nullptr, nullptr);
        (void)kws_run_classifier_batch_device(h, pcm.data(), 0, scores.data(), nullptr, nullptr, nullptr);
        (void)kws_extract_mfcc_batch_device(h, pcm.data(), B, feats.data(), is_float ? nullptr : q.data(), nullptr);
        (void)kws_mfcc_batch_device(h, pcm.data(), B, feats.data(), nullptr);
        (void)kws_cmvn_inference_batch_device(h, feats.data(), B, scores.data(), feats.data(), nullptr, nullptr);
        (void)kws_run_inference_batch_device(h, feats.data(), B, scores.data(), nullptr);
        size_t nfb = 0;
        (void)kws_fast_fallback_count(h, &nfb);
        (void)kws_fast_exact_count(h, &nfb);
    }
    (void)kws_set_mode(h, KWS_MODE_EXACT);
    (void)kws_run_classifier_batch(h, pcm.data(), B, scores.data(), feats.data(), is_float ? nullptr : q.data());
    (void)kws_nn_batch(h, q.data(), 2, scores.data(), nullptr, nullptr, nullptr);
    (void)kws_nn_f32_batch_device(h, feats.data(), 2, scores.data(), nullptr, nullptr);
    (void)kws_mfe_batch_device(h, pcm.data(), 2, feats.data(), nullptr, nullptr);
    kws_stream_batch *sb = nullptr;
    if (kws_streams_create(h, 3, &sb) == EI_IMPULSE_OK) {
        std::vector<int16_t> slice(3 * (n / 4) + 8, 2);
        int produced = 0;
        for (int k = 0; k < 6; k++) (void)kws_streams_step_device(sb, slice.data(), n / 4, nullptr, scores.data(), &produced, nullptr);
        (void)kws_streams_init(sb);
        (void)kws_streams_step_device(sb, slice.data(), n / 4, nullptr, scores.data(), &produced, nullptr);
        (void)kws_streams_step_device(sb, slice.data(), n / 4 + 1, nullptr, scores.data(), &produced, nullptr);      // wrong slice length
        kws_streams_destroy(sb);
    }
    // the SDK entry points on this model: ei_compat.h publishes this program's EI_CLASSIFIER_LABEL_COUNT (4): a model with another
    // label count must be refused before anything is written into the 4-label result struct
    (void)kws_set_default_model(h);
    g_audio.assign(n, 0.25f);
    signal_t sig;
    sig.get_data = &get_data;
    sig.total_length = n;
    ei_impulse_result_t res;
    memset(&res, 0, sizeof res);
    (void)run_classifier(&sig, &res, false);
    // windows of another length (kws_plan_for_length: a plan per frame count, cached pad maps; the gather with fewer frames), and the refused ones
    for (size_t len : { n - 1, n + 1, n / 2, n / 2 + 7, (size_t)640, (size_t)641, (size_t)639, (size_t)1, (size_t)0, n + n / 16, 2 * n, n - 1 }) {
        g_audio.assign(len + 4, 0.125f);
        sig.total_length = len;
        (void)run_classifier(&sig, &res, false);              // (debug = true would print to stdout, which the test parses line by line)
    }
    g_audio.assign(n, 0.25f);
    run_classifier_init();
    sig.total_length = n / 4;
    (void)run_classifier_continuous(&sig, &res, false);
    sig.total_length = n / 8;                                    // a shorter slice: fewer frames per call
    (void)run_classifier_continuous(&sig, &res, false);
}

// --trace: the (offset, length, return value) of every call the SDK entry points make to the application's get_data, in order -- host
// logic that runs before any device work, so the stub runtime shows it; tests/test_get_data_sequence.py compares it with what the
// compiled reference asks its callback (tests/golden/get_data_trace_l476.npz).  The callback refuses ranges beyond the real buffer, as the
// reference's test driver does (oracle/ref_driver.cpp pcm_get_data).  The process's first continuous call is the reference's `first_run`.
static void print_trace(kws_handle *h)
{
    const size_t n = (size_t)kws_clip_samples(h);
    (void)kws_set_default_model(h);
    signal_t sig;
    sig.get_data = &get_data;
    ei_impulse_result_t res;
    auto one = [&](const char *tag, size_t real, size_t claimed, bool continuous) {
        g_audio.assign(real, 0.25f);
        sig.total_length = claimed;
        memset(&res, 0, sizeof res);
        printf("TRACE %s", tag);
        g_tracing = true;
        const EI_IMPULSE_ERROR rc = continuous ? run_classifier_continuous(&sig, &res, false) : run_classifier(&sig, &res, false);
        g_tracing = false;
        printf(" | total_length_after %zu error %d\n", sig.total_length, rc == EI_IMPULSE_DSP_ERROR ? 1 : 0);
    };
    one("oneshot", n, n, false);
    one("oneshot_short", n - 1, n - 1, false);
    run_classifier_init();
    one("continuous_first", n / 4, n / 4, true);
    one("continuous_second", n / 4, n / 4, true);
    one("continuous_short", n / 8, n / 8, true);
}

// --gain: print what kws_create calibrated for a float32 graph (kws_gain.cpp is host code: the stub runtime is all it needs) --
// tests/test_gain_calibration.py compares it with Jacobians of the oracle's network
static void print_gain(const char *path, kws_handle *h)
{
    kws_fast_tolerance t;
    std::vector<float> g(256, 0.0f), coef(4 * 256, 0.0f);
    if (kws_fast_tolerance_info(h, &t) != EI_IMPULSE_OK || kws_fast_guard(h, 1, coef.data()) != EI_IMPULSE_OK) { printf("GAIN %s none\n", path); return; }
    printf("GAIN %s calibrated %d columns %d frames %d k %.9g lin %.9g cap %.9g c1 %.9g c2 %.9g sigma_net %.9g total %.9g uniform_tol %.9g entry %d :", path, t.calibrated,
           t.n_columns, t.n_frames, t.k_sigma, t.lin_margin, t.logit_cap, t.g_c1, t.g_c2, t.sigma_net, t.total_gain, t.uniform_feature_tol, t.entry_tier);
    if (t.calibrated && kws_fast_gain(h, g.data()) == EI_IMPULSE_OK)
        for (int c = 0; c < t.n_columns; c++) printf(" %.9g", g[c]);
    printf("\n");
}

int main(int argc, char **argv)
{
    const bool gain_only = argc > 1 && strcmp(argv[1], "--gain") == 0, trace_only = argc > 1 && strcmp(argv[1], "--trace") == 0;
    for (int i = gain_only || trace_only ? 2 : 1; i < argc; i++) {
        FILE *f = fopen(argv[i], "rb");
        if (!f) { printf("%s rc open-failed\n", argv[i]); continue; }
        std::vector<unsigned char> blob;
        unsigned char tmp[4096];
        size_t k;
        while ((k = fread(tmp, 1, sizeof tmp, f)) > 0) blob.insert(blob.end(), tmp, tmp + k);
        fclose(f);
        kws_handle *h = nullptr;
        const EI_IMPULSE_ERROR rc = kws_create(blob.data(), blob.size(), 0, &h);
        printf("%s rc %d\n", argv[i], (int)rc);
        fflush(stdout);
        if (rc == EI_IMPULSE_OK && gain_only) print_gain(argv[i], h);
        else if (rc == EI_IMPULSE_OK && trace_only) print_trace(h);
        else if (rc == EI_IMPULSE_OK) walk(h);
        if (rc == EI_IMPULSE_OK) kws_destroy(h);
    }
    return 0;
}
