// pgcn_device.h -- small device helpers shared by the SpMM translation units.
#ifndef PGCN_DEVICE_H
#define PGCN_DEVICE_H
#include <hip/hip_runtime.h>

template <int VEC> struct VecT;
template <> struct VecT<1> { using type = float; };
template <> struct VecT<2> { using type = float2; };
template <> struct VecT<4> { using type = float4; };

template <int VEC>
__device__ __forceinline__ void vload(float (&x)[VEC], const float *p) {
    using V = typename VecT<VEC>::type;
    const V v = *reinterpret_cast<const V *>(p);
    if constexpr (VEC == 1) { x[0] = v; }
    if constexpr (VEC == 2) { x[0] = v.x; x[1] = v.y; }
    if constexpr (VEC == 4) { x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w; }
}

template <int VEC>
__device__ __forceinline__ void vstore(float *p, const float (&x)[VEC]) {
    using V = typename VecT<VEC>::type;
    if constexpr (VEC == 1) { *p = x[0]; }
    if constexpr (VEC == 2) { *reinterpret_cast<V *>(p) = make_float2(x[0], x[1]); }
    if constexpr (VEC == 4) { *reinterpret_cast<V *>(p) = make_float4(x[0], x[1], x[2], x[3]); }
}
#endif
