// Deterministic fp32 device math shared by every kernel that needs exp(): a pure fmaf polynomial, so the
// result is bit-identical to the CPU statement of the same formula (the parity tests rely on that).
// Built with -ffp-contract=off: every fused multiply-add below is explicit.
#pragma once
#include <hip/hip_runtime.h>

namespace eagle {

__device__ __forceinline__ float d_expf(float x)
{
    x = x > 88.0f ? 88.0f : x;
    x = x < -87.0f ? -87.0f : x;
    const float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693145751953125f, x);
    r = fmaf(n, -1.42860682030941723e-6f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    const float r2 = r * r;
    p = fmaf(p, r2, r);
    p = p + 1.0f;
    return p * __uint_as_float((unsigned)((int)n + 127) << 23);
}
__device__ __forceinline__ float d_sigmoidf(float x) { return 1.0f / (1.0f + d_expf(-x)); }
__device__ __forceinline__ float d_act(float v, int act)
{
    if (act == 1) return v > 0.0f ? v : 0.0f;
    if (act == 2) return v * d_sigmoidf(v);
    return v;
}

}  // namespace eagle
