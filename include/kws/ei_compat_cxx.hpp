// ei_compat_cxx.hpp -- included by ei_compat.h when a C++ application defines KWS_SIGNAL_STD_FUNCTION (do not include it directly).
#ifndef KWS_EI_COMPAT_CXX_HPP
#define KWS_EI_COMPAT_CXX_HPP
/* The SDK's default signal_t of a C++ build (dsp/numpy_types.h:244-249: `std::function<int(size_t, size_t, float *)> get_data`), for
 * applications that hand the classifier a lambda or a bound member function.  Header-only: the overloads below run in the application's own
 * translation unit (its own std::function layout) and call the C entry points through a trampoline; the signal in flight is kept per thread
 * (the reference's run_classifier is not re-entrant either).  `debug` defaults to false as in ei_run_classifier.h:650,184. */
#include <functional>
namespace ei {
struct signal_t {
    std::function<int(size_t offset, size_t length, float *out_ptr)> get_data;
    size_t total_length;
};
}  // namespace ei
using ei::signal_t;
namespace kws_detail {
inline ei::signal_t *&current_signal() { static thread_local ei::signal_t *s = nullptr; return s; }
inline int signal_trampoline(size_t offset, size_t length, float *out) { return current_signal()->get_data(offset, length, out); }
template <EI_IMPULSE_ERROR (*FN)(kws_c_signal_t *, ei_impulse_result_t *, bool)>
inline EI_IMPULSE_ERROR call_with(ei::signal_t *signal, ei_impulse_result_t *result, bool debug)
{
    kws_c_signal_t c = { &signal_trampoline, signal->total_length };
    ei::signal_t *const outer = current_signal();
    current_signal() = signal;
    const EI_IMPULSE_ERROR r = FN(&c, result, debug);
    current_signal() = outer;
    signal->total_length = c.total_length;      /* continuous mode claims one more frame length (ei_run_dsp.h:322-324): the caller's struct sees it */
    return r;
}
}  // namespace kws_detail
inline EI_IMPULSE_ERROR run_classifier(ei::signal_t *signal, ei_impulse_result_t *result, bool debug = false)
{
    return kws_detail::call_with<&::run_classifier>(signal, result, debug);
}
inline EI_IMPULSE_ERROR run_classifier_continuous(ei::signal_t *signal, ei_impulse_result_t *result, bool debug = false)
{
    return kws_detail::call_with<&::run_classifier_continuous>(signal, result, debug);
}
#endif
