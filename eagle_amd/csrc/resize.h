// cv2.resize(u8, INTER_LINEAR) restated for device code: half-pixel centres, 11-bit fixed-point taps, the exact 2x
// decimation handled as cv2 does (2 x 2 area average).  Same arithmetic as elementwise.hip::resize_px (K1), with an explicit row stride so
// that a sub-rectangle of a frame can be the source (reid.hip: player crops).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace eagle {

struct ResizeTap { int s0, s1; int a0, a1; };

// scale = source pixels per destination pixel: ssize / dsize for cv2.resize(dsize), 1 / fx for cv2.resize((0, 0), fx, fy)
__device__ __forceinline__ ResizeTap resize_tap_scaled(int d, double scale, int ssize)
{
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { s = 0; f = 0.f; }
    if (s >= ssize - 1) { s = ssize - 1; f = 0.f; }
    ResizeTap r;
    r.s0 = s; r.s1 = (s + 1 < ssize) ? s + 1 : s;
    r.a0 = (int)(short)lrintf((1.f - f) * 2048.f);
    r.a1 = (int)(short)lrintf(f * 2048.f);
    return r;
}
__device__ __forceinline__ ResizeTap resize_tap(int d, int dsize, int ssize) { return resize_tap_scaled(d, (double)ssize / (double)dsize, ssize); }

// resized RGB u8 pixel (dy, dx) of an (sh, sw) -> (dh, dw) resize; src is BGR with `rs` bytes per row
__device__ __forceinline__ void resize_px_strided(const uint8_t* src, size_t rs, int sh, int sw, int dh, int dw, int dy, int dx, int rgb[3])
{
    if (sh == dh && sw == dw) {
        const uint8_t* p = src + dy * rs + dx * 3;
        rgb[0] = p[2]; rgb[1] = p[1]; rgb[2] = p[0];
    } else if (sh == 2 * dh && sw == 2 * dw) {
        const uint8_t* p = src + (size_t)(2 * dy) * rs + (size_t)(2 * dx) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) rgb[2 - c] = (p[c] + p[3 + c] + p[rs + c] + p[rs + 3 + c] + 2) >> 2;
    } else {
        const ResizeTap ax = resize_tap(dx, dw, sw), ay = resize_tap(dy, dh, sh);
        const uint8_t* r0 = src + ay.s0 * rs; const uint8_t* r1 = src + ay.s1 * rs;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int t0 = r0[ax.s0 * 3 + c] * ax.a0 + r0[ax.s1 * 3 + c] * ax.a1;
            const int t1 = r1[ax.s0 * 3 + c] * ax.a0 + r1[ax.s1 * 3 + c] * ax.a1;
            rgb[2 - c] = (((ay.a0 * (t0 >> 4)) >> 16) + ((ay.a1 * (t1 >> 4)) >> 16) + 2) >> 2;
        }
    }
}

}  // namespace eagle
