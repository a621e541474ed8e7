// Synthetic co-runners for tools/probe_lk_concurrency.py: each kernel stresses ONE hardware feature next to K12 (developer tool).
//   0 lds    : ds_write_b128 / ds_read_b128 over 20 KB of dynamic LDS          1 oob : raw buffer loads / stores with out-of-range lanes
//   2 mfma   : v_mfma_f32_16x16x32_f16 loop, no LDS, no memory                 3 l1  : streaming 16-byte global loads + stores
//   4 churn  : near-empty workgroups that only allocate 20 KB of LDS           5 ldsbyte : ds_write_b8 / ds_read_u8 traffic
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdlib>
#include <cstring>
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
using rsrc_t = __amdgpu_buffer_rsrc_t;

__global__ __launch_bounds__(256, 2) void k_lds(unsigned* sink, int rounds)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4 v = {threadIdx.x, blockIdx.x, 0xDEADBEEFu, 0xFFFFFFFFu};
    for (int r = 0; r < rounds; ++r) {
        for (int o = threadIdx.x * 16; o < 20480; o += 256 * 16) *(u32x4*)(smem + o) = v;
        __syncthreads();
        for (int o = threadIdx.x * 16; o < 20480; o += 256 * 16) { u32x4 t = *(const u32x4*)(smem + ((o + 4096) % 20480)); v.x += t.y; v.z ^= t.w; }
        __syncthreads();
    }
    if (v.x == 0x12345678u) sink[0] = v.z;
}
__global__ __launch_bounds__(256, 2) void k_ldsbyte(unsigned* sink, int rounds)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned acc = threadIdx.x;
    for (int r = 0; r < rounds; ++r) {
        for (int o = threadIdx.x; o < 20480; o += 256) smem[o] = (char)(0xA5 ^ o ^ r);
        __syncthreads();
        for (int o = threadIdx.x; o < 20480; o += 256) acc += (unsigned char)smem[(o + 777) % 20480];
        __syncthreads();
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256, 2) void k_oob(const unsigned* src, unsigned* dst, unsigned* sink, int rounds, int bytes)
{
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7FFFFFFF, 0x00020000);
    const rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, 0x7FFFFFFF, 0x00020000);
    u32x4 acc = {0, 0, 0, 0};
    const unsigned base = ((blockIdx.x * 256u + threadIdx.x) * 16u) % (unsigned)(bytes - 64);
    for (int r = 0; r < rounds; ++r) {
        const unsigned off = ((threadIdx.x + r) & 3) ? ((base + r * 4096u) % (unsigned)(bytes - 64)) & ~15u : 0x80000000u;   // a quarter of the lanes out of range
        const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
        acc.x += t.x; acc.y ^= t.y; acc.z += t.z; acc.w ^= t.w;
        __builtin_amdgcn_raw_buffer_store_b128(acc, rd, ((threadIdx.x + r) & 1) ? off : 0x80000000u, 0, 0);
    }
    if (acc.x == 0x12345678u) sink[0] = acc.y;
}
__global__ __launch_bounds__(256, 2) void k_mfma(unsigned* sink, int rounds)
{
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(1.0f - i * 0.01f); }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int r = 0; r < rounds; ++r) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, a, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, b, c3, 0, 0, 0);
    }
    if (c0[0] + c1[1] + c2[2] + c3[3] == 123.456f) sink[0] = 1;
}
__global__ __launch_bounds__(256, 2) void k_l1(const u32x4* src, u32x4* dst, unsigned* sink, int rounds, int n16)
{
    u32x4 acc = {0, 0, 0, 0};
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) % n16;
    for (int r = 0; r < rounds; ++r) {
        const u32x4 t = src[i];
        acc.x += t.x; acc.y ^= t.y;
        dst[i] = acc;
        i = (i + 256 * 1024 + 64) % n16;
    }
    if (acc.x == 0x12345678u) sink[0] = acc.y;
}
__global__ __launch_bounds__(256, 2) void k_churn(unsigned* sink)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    smem[threadIdx.x] = (char)threadIdx.x;
    __syncthreads();
    if (smem[(threadIdx.x + 1) & 255] == 77 && blockIdx.x == 0x7FFFFFF) sink[0] = 1;
}

static hipStream_t g_s = nullptr;
static unsigned *g_src = nullptr, *g_dst = nullptr, *g_sink = nullptr;
static const int BYTES = 64 << 20;
extern "C" int hammer_launch(int kind, int n)
{
    if (!g_s) {
        if (hipStreamCreateWithFlags(&g_s, hipStreamNonBlocking) != hipSuccess) return -1;
        if (hipMalloc((void**)&g_src, BYTES) != hipSuccess || hipMalloc((void**)&g_dst, BYTES) != hipSuccess || hipMalloc((void**)&g_sink, 256) != hipSuccess) return -2;
        hipMemset(g_src, 0x5A, BYTES); hipMemset(g_dst, 0, BYTES);
    }
    for (int i = 0; i < n; ++i) {
        switch (kind) {
        case 0: hipLaunchKernelGGL(k_lds, dim3(4096), dim3(256), 20480, g_s, g_sink, 8); break;
        case 1: hipLaunchKernelGGL(k_oob, dim3(4096), dim3(256), 0, g_s, g_src, g_dst, g_sink, 64, BYTES); break;
        case 2: hipLaunchKernelGGL(k_mfma, dim3(4096), dim3(256), 0, g_s, g_sink, 512); break;
        case 3: hipLaunchKernelGGL(k_l1, dim3(4096), dim3(256), 0, g_s, (const u32x4*)g_src, (u32x4*)g_dst, g_sink, 64, BYTES / 16); break;
        case 4: hipLaunchKernelGGL(k_churn, dim3(65536), dim3(256), 20480, g_s, g_sink); break;
        case 5: hipLaunchKernelGGL(k_ldsbyte, dim3(4096), dim3(256), 20480, g_s, g_sink, 2); break;
        default: return -3;
        }
    }
    return hipStreamSynchronize(g_s) == hipSuccess ? 0 : -4;
}

// ---- victims: one class of K12's arithmetic each, run next to a hammer and compared with their own idle-GPU output --------------
__device__ __forceinline__ int v_wave_sum32(int v)
{
#define STEP(CTRL_, ROWM_, BANKM_) v += __builtin_amdgcn_update_dpp(0, v, CTRL_, ROWM_, BANKM_, false);
    STEP(0x111, 0xf, 0xf) STEP(0x112, 0xf, 0xf) STEP(0x114, 0xf, 0xe) STEP(0x118, 0xf, 0xc) STEP(0x142, 0xa, 0xf) STEP(0x143, 0xc, 0xf)
#undef STEP
    return v;
}
__device__ __forceinline__ unsigned lcg(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }

template <int KIND>
__global__ __launch_bounds__(256) void k_victim(unsigned long long* out, int rounds)
{
    __shared__ long long red[4][3];
    __shared__ unsigned char bytes[1024];
    __shared__ int words[324];
    const int tid = threadIdx.x;
    unsigned s = tid * 2654435761u + blockIdx.x * 40503u + 12345u;
    unsigned long long h = 0;
    for (int e = tid; e < 1024; e += 256) bytes[e] = (unsigned char)(e * 7 + blockIdx.x);
    for (int e = tid; e < 324; e += 256) words[e] = (e * 13 + blockIdx.x) & 255;
    __syncthreads();
    for (int r = 0; r < rounds; ++r) {
        const int a = (int)(lcg(s) >> 8) - (1 << 23), b = (int)(lcg(s) >> 16) - (1 << 15);
        if constexpr (KIND == 0) {                        // DPP wave reduction
            const int t = v_wave_sum32(a >> 4);
            h = h * 31 + (unsigned)__builtin_amdgcn_readlane(t, 63);
        } else if constexpr (KIND == 1) {                 // 32x32 -> 64 integer products
            long long p = (long long)a * b;
            h = h * 31 + (unsigned long long)p;
        } else if constexpr (KIND == 2) {                 // fp64
            const double x = (double)a * 1e-3, y = (double)b * 1e-2;
            const double z = x * x + y * y;
            h = h * 31 + (unsigned long long)__double_as_longlong(z) + (fabs(x + y) < 0.01 ? 1 : 0);
        } else if constexpr (KIND == 3) {                 // sqrt / reciprocal / rounding
            const float x = (float)a * 1e-3f, y = (float)b * 0.37f;
            const float z = sqrtf(x * x + 4.f * y * y) + 1.f / (y * y + 1.f) + rintf(x * 0.3f) + floorf(y * 0.7f);
            h = h * 31 + (unsigned)__float_as_int(z);
        } else if constexpr (KIND == 4) {                 // int64 -> float, float fma chain as in the 2x2 solve
            const long long q = (long long)a * b;
            const float f = (float)q * (1.f / (1 << 20)), g = (float)(q >> 3) * (1.f / (1 << 20));
            const float d = (float)((f * g - g * f * 0.5f) * 1.25f);
            h = h * 31 + (unsigned)__float_as_int(d) + (unsigned)__float_as_int(f);
        } else if constexpr (KIND == 5) {                 // LDS byte / word reads with the bilinear integer arithmetic
            const int i0 = (lcg(s) >> 10) % 990, j0 = (lcg(s) >> 12) % 300;
            const int v = bytes[i0] * 5000 + bytes[i0 + 1] * 3000 + bytes[i0 + 32] * 6000 + bytes[i0 + 33] * 2384;
            const int w = words[j0] * 5000 + words[j0 + 1] * 3000 + words[j0 + 18] * 6000 + words[j0 + 19] * 2384;
            h = h * 31 + (unsigned)(((v + (1 << 8)) >> 9) - (short)((w + (1 << 8)) >> 9));
        } else if constexpr (KIND == 6) {                 // the full four-wave exact sum (DPP + LDS exchange + barriers)
            const int a32 = v_wave_sum32(a >> 4), b32 = v_wave_sum32(b), c32 = v_wave_sum32(a >> 9);
            __syncthreads();
            if ((tid & 63) == 63) { red[tid >> 6][0] = a32; red[tid >> 6][1] = b32; red[tid >> 6][2] = c32; }
            __syncthreads();
            const long long A = red[0][0] + red[1][0] + red[2][0] + red[3][0], B = red[0][1] + red[1][1] + red[2][1] + red[3][1], C = red[0][2] + red[1][2] + red[2][2] + red[3][2];
            h = h * 31 + (unsigned long long)(A * 3 + B * 5 + C * 7);
        } else if constexpr (KIND == 7) {                 // ds_bpermute (shuffle) reduction
            int t = a >> 4;
            for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o, 64);
            h = h * 31 + (unsigned)__shfl(t, 0, 64);
        } else if constexpr (KIND == 9) {                 // packed fp32 (v_pk_mul_f32 / v_pk_add_f32) against the scalar instructions on the same operands
            using f2 = __attribute__((ext_vector_type(2))) float;
            const float x0 = (float)a * 1e-3f, x1 = (float)b * 0.37f, y0 = (float)(a >> 3) * 0.11f, y1 = (float)(b >> 2) * 1e-2f;
            f2 X = {x0, x1}, Y = {y0, y1}, M, S;
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(M) : "v"(X), "v"(Y));
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(S) : "v"(M), "v"(X));
            float m0, m1, s0, s1;
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m0) : "v"(x0), "v"(y0));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m1) : "v"(x1), "v"(y1));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(s0) : "v"(m0), "v"(x0));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(s1) : "v"(m1), "v"(x1));
            const bool bad = __float_as_int(M.x) != __float_as_int(m0) || __float_as_int(M.y) != __float_as_int(m1) || __float_as_int(S.x) != __float_as_int(s0) || __float_as_int(S.y) != __float_as_int(s1);
            if (bad) atomicAdd((unsigned long long*)&out[57 * 256], 1ull);      // in-kernel disagreement counter behind the per-thread results
            h = h * 31 + (unsigned)__float_as_int(S.x) + ((unsigned long long)(unsigned)__float_as_int(S.y) << 32);
        } else if constexpr (KIND == 8) {                 // plain 32-bit integer / float VALU only
            const float x = (float)a * 0.25f + (float)b;
            h = h * 31 + (unsigned)(a * 3 + (b ^ (a >> 3))) + (unsigned)__float_as_int(x * x + 1.5f);
        }
    }
    out[blockIdx.x * 256 + tid] = h;
}

static unsigned long long *g_vout = nullptr, *g_vref[16] = {};
static hipStream_t g_vs = nullptr;
static const int VBLOCKS = 57;
static void victim_launch(int kind, int rounds)
{
    switch (kind) {
#define VK(K_) case K_: hipLaunchKernelGGL(k_victim<K_>, dim3(VBLOCKS), dim3(256), 0, g_vs, g_vout, rounds); break;
    VK(0) VK(1) VK(2) VK(3) VK(4) VK(5) VK(6) VK(7) VK(8) VK(9)
#undef VK
    }
}
// n launches of victim `kind`; launch 0 of the first call (made on an idle GPU) is the reference.  Returns the number of launches
// whose output differs from it; *bad_threads = differing thread results summed over the launches.
extern "C" int victim_run(int kind, int n, int rounds, long long* bad_threads)
{
    const size_t NB = (size_t)VBLOCKS * 256 * sizeof(unsigned long long);
    if (!g_vs) {
        if (hipStreamCreateWithFlags(&g_vs, hipStreamNonBlocking) != hipSuccess) return -1;
        if (hipMalloc((void**)&g_vout, NB + 64) != hipSuccess) return -2;
        hipMemset(g_vout, 0, NB + 64);
    }
    static unsigned long long* host = (unsigned long long*)malloc(NB);
    int bad = 0; long long bt = 0;
    for (int i = 0; i < n; ++i) {
        victim_launch(kind, rounds);
        if (hipMemcpyAsync(host, g_vout, NB, hipMemcpyDeviceToHost, g_vs) != hipSuccess || hipStreamSynchronize(g_vs) != hipSuccess) return -3;
        if (!g_vref[kind]) { g_vref[kind] = (unsigned long long*)malloc(NB); memcpy(g_vref[kind], host, NB); continue; }
        long long d = 0;
        for (size_t k = 0; k < (size_t)VBLOCKS * 256; ++k) d += host[k] != g_vref[kind][k];
        if (d) { ++bad; bt += d; }
    }
    if (bad_threads) *bad_threads = bt;
    if (kind == 9) {                                   // packed-vs-scalar disagreements counted inside the kernel (cumulative)
        unsigned long long c = 0;
        hipMemcpy(&c, g_vout + (size_t)VBLOCKS * 256, 8, hipMemcpyDeviceToHost);
        if (bad_threads) *bad_threads = (long long)c;
    }
    return bad;
}
