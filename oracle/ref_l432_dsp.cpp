// oracle/ref_l432_dsp.cpp -- TEST INFRASTRUCTURE ONLY.
//
// The second SDK copy of the reference (nucleo-l432-keyword-spotting/keyword-spotting-02-v3) cannot be built as a whole here:
// its dsp/speechpy/feature.hpp:32-33 includes the STM32 "main.h".  The normalisation of its MFE block does not need that
// header: speechpy::processing::cmvnw(matrix, win_size, variance_normalization, scale) (dsp/speechpy/processing.hpp:327-399)
// and numpy::normalize (dsp/numpy.hpp:1391-1429) live in headers that compile on the host.  This driver includes those two
// headers WHERE THEY LIE and exposes the calls extract_mfe_features makes (classifier/ei_run_dsp.h:369-418:
// feature::mfe -> cmvnw(win_size, false, true)).  feature::mfe itself is textually identical in both SDK copies (only the
// include lines differ), so the L476 build (ref_driver.cpp) pins it.
#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>     // the SDK headers below expect their includer to have pulled in the standard library

#include "edge-impulse-sdk/dsp/speechpy/processing.hpp"

using namespace ei;

// porting hook the DSP headers print through (porting/ei_classifier_porting.h)
void ei_printf(const char *format, ...) {
    va_list ap;
    va_start(ap, format);
    vfprintf(stderr, format, ap);
    va_end(ap);
}

extern "C" {

// processing.hpp:327: cmvnw in place on m[rows][cols]
int eiref432_cmvnw(float *m, int rows, int cols, int win_size, int variance_normalization, int scale)
{
    matrix_t mat(rows, cols, m);
    return speechpy::processing::cmvnw(&mat, (uint16_t)win_size, variance_normalization != 0, scale != 0);
}

// numpy.hpp:1391: (matrix - min) * (1 / (max - min)) in place
int eiref432_normalize(float *m, int rows, int cols)
{
    matrix_t mat(rows, cols, m);
    return numpy::normalize(&mat);
}

}  // extern "C"
