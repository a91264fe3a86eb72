// The other translation units of the library, as far as bsk_api.hip's host code links against them: the
// large-table pipeline declines (its host code is a launch sequence without threads or staging of its own).
#include "../../bspy_amd/csrc/bsk_host.hpp"
template <typename T>
bsk_status gather_or_binned_any(bsk_spline, bool, const Params<T> &, long long, T *, long long, const Wrt &, hipStream_t)
{
    return BSK_ERR_UNSUPPORTED;
}
template bsk_status gather_or_binned_any<float>(bsk_spline, bool, const Params<float> &, long long, float *, long long, const Wrt &, hipStream_t);
template bsk_status gather_or_binned_any<double>(bsk_spline, bool, const Params<double> &, long long, double *, long long, const Wrt &, hipStream_t);
template <typename T>
bsk_status cellsort_jacobian_any(bsk_spline, const Params<T> &, long long, T *, hipStream_t)
{
    return BSK_ERR_UNSUPPORTED;
}
template bsk_status cellsort_jacobian_any<float>(bsk_spline, const Params<float> &, long long, float *, hipStream_t);
template bsk_status cellsort_jacobian_any<double>(bsk_spline, const Params<double> &, long long, double *, hipStream_t);

template <typename T>
bsk_status slab2_any(bsk_spline, bool, const Params<T> &, long long, T *, long long, const Wrt &, hipStream_t)
{
    return BSK_ERR_UNSUPPORTED;
}
template bsk_status slab2_any<float>(bsk_spline, bool, const Params<float> &, long long, float *, long long, const Wrt &, hipStream_t);
template bsk_status slab2_any<double>(bsk_spline, bool, const Params<double> &, long long, double *, long long, const Wrt &, hipStream_t);
