/* boundary_driver.c -- TEST PROGRAM (tests/test_boundary_hooks.py): an application written against include/kws/ei_compat.h that supplies its own
 * porting hooks -- ei_printf / ei_printf_float (captured) and ei_run_impulse_check_canceled (answers EI_IMPULSE_CANCELED on its n-th call) --
 * and walks the boundary scenarios of tools/make_golden.py (BOUNDARY_SCENARIOS) through run_classifier / run_inference /
 * run_classifier_continuous in the order the fixture was recorded in (a fresh process: the first continuous call is special in the reference,
 * ei_run_dsp.h:313).  Plain C11: the host side stays C.
 *
 *   boundary_driver <input.bin> <scenarios.txt>
 *   input.bin     int16 clips[4][16000], then float features_a[n_features]
 *   scenarios.txt one per line: name kind(oneshot|inference|continuous|init) what(clip|features|slice|-) index debug cancel_at
 * Output per scenario: "SCEN <name> <rc> <polls> <text bytes>\n", "RES <hex of the caller's ei_impulse_result_t, pre-filled with 0xA5>\n",
 * "LAB <label or - per class>\n", then the captured text and "\nEND\n".                                                            */
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kws/ei_compat.h"

static char g_text[1 << 17];
static size_t g_len;
static int g_cancel_at, g_polls;

void ei_printf(const char *format, ...)
{
    va_list a;
    va_start(a, format);
    if (g_len + 1 < sizeof g_text) {
        const int n = vsnprintf(g_text + g_len, sizeof g_text - g_len, format, a);
        if (n > 0) g_len = g_len + (size_t)n < sizeof g_text ? g_len + (size_t)n : sizeof g_text - 1;
    }
    va_end(a);
}
void ei_printf_float(float f) { ei_printf("%f", f); }
EI_IMPULSE_ERROR ei_run_impulse_check_canceled(void)
{
    g_polls++;
    return (g_cancel_at > 0 && g_polls == g_cancel_at) ? EI_IMPULSE_CANCELED : EI_IMPULSE_OK;
}

static const int16_t *g_pcm;
static size_t g_pcm_len;
static int get_data(size_t offset, size_t length, float *out)
{
    if (offset + length > g_pcm_len) return -1;
    for (size_t i = 0; i < length; i++) out[i] = (float)g_pcm[offset + i] / 32768.0f;      /* numpy::int16_to_float (numpy.hpp:1289-1298) */
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    static int16_t clips[4 * 16000];
    static float feats[4096];
    if (fread(clips, sizeof(int16_t), 4 * 16000, f) != 4 * 16000) return 2;
    const size_t nfeat = fread(feats, sizeof(float), 4096, f);
    fclose(f);
    const int16_t *stream = clips + 2 * 16000;
    FILE *sc = fopen(argv[2], "r");
    if (!sc) return 2;
    char name[64], kind[32], what[32];
    int index, debug, cancel_at;
    while (fscanf(sc, "%63s %31s %31s %d %d %d", name, kind, what, &index, &debug, &cancel_at) == 6) {
        if (strcmp(kind, "init") == 0) { run_classifier_init(); continue; }
        ei_impulse_result_t res;
        memset(&res, 0xA5, sizeof res);
        g_len = 0; g_text[0] = 0; g_polls = 0; g_cancel_at = cancel_at;
        EI_IMPULSE_ERROR rc;
        if (strcmp(kind, "inference") == 0) {
            ei_matrix_t m;
            memset(&m, 0, sizeof m);
            m.buffer = feats; m.rows = 1; m.cols = (uint32_t)nfeat;
            rc = run_inference(&m, &res, debug != 0);
        } else {
            signal_t sig;
            const int cont = strcmp(kind, "continuous") == 0;
            g_pcm = cont ? stream + (size_t)index * 4000 : clips + (size_t)index * 16000;
            g_pcm_len = cont ? 4000 : 16000;
            sig.get_data = &get_data;
            sig.total_length = g_pcm_len;
            rc = cont ? run_classifier_continuous(&sig, &res, debug != 0) : run_classifier(&sig, &res, debug != 0);
        }
        g_cancel_at = 0;
        printf("SCEN %s %d %d %zu\nRES ", name, (int)rc, g_polls, g_len);
        const unsigned char *b = (const unsigned char *)&res;
        for (size_t i = 0; i < sizeof res; i++) printf("%02x", b[i]);
        printf("\nLAB");
        for (int i = 0; i < EI_CLASSIFIER_LABEL_COUNT; i++) {
            uintptr_t p;
            memcpy(&p, &res.classification[i].label, sizeof p);
            uintptr_t untouched;
            memset(&untouched, 0xA5, sizeof untouched);
            printf(" %s", p == untouched || p == 0 ? "-" : res.classification[i].label);
        }
        printf("\n");
        fwrite(g_text, 1, g_len, stdout);
        printf("\nEND\n");
        if (rc != EI_IMPULSE_OK && rc != EI_IMPULSE_CANCELED) { printf("FAILED %d\n", (int)rc); return 1; }
    }
    fclose(sc);
    return 0;
}
