// Numerics probe for the split-precision convolution family (round 3): how close does
//   D = Ahi*Bhi + Ahi*Blo + Alo*Bhi   on v_mfma_f32_16x16x32_f16 (fp32 accumulate)
// come to the exact product of fp32 operands, compared with the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32) and plain fp16?
// Variants: (s1) single accumulator, power-of-two pre-scaling of both operands; (s2) two accumulators (hi*hi | cross terms with the
// lo parts scaled by 2^11); (u) no scaling at all (shows what fp16 subnormal `lo` parts cost / whether the MFMA flushes them).
// Build: hipcc --offload-arch=gfx950 -O2 -o split_mfma_probe split_mfma_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

using half8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;

// A: [16][K] row-major fp32, B: [K][16] stored as Bt [16][K] (column n contiguous over k); out: [mode][16][16]
__global__ void probe(const float* A, const float* Bt, int K, float sa, float sb, float* out)
{
    const int lane = threadIdx.x, q = lane >> 4, lx = lane & 15;
    f32x4 acc32 = {0, 0, 0, 0}, acc16 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0}, s2a = {0, 0, 0, 0}, s2b = {0, 0, 0, 0}, u = {0, 0, 0, 0};
    for (int k0 = 0; k0 < K; k0 += 32) {
        half8 ah, al, bh, bl, ah2, al2, bh2, bl2, ahu, alu, bhu, blu, a16, b16;
        for (int j = 0; j < 8; ++j) {
            const float a = A[lx * K + k0 + q * 8 + j], b = Bt[lx * K + k0 + q * 8 + j];
            a16[j] = (_Float16)a; b16[j] = (_Float16)b;
            // s1: scaled operands, lo unscaled relative to hi
            const float as = a * sa, bs = b * sb;
            ah[j] = (_Float16)as; al[j] = (_Float16)(as - (float)ah[j]);
            bh[j] = (_Float16)bs; bl[j] = (_Float16)(bs - (float)bh[j]);
            // s2: unscaled hi, lo * 2^11
            ah2[j] = (_Float16)a; al2[j] = (_Float16)((a - (float)ah2[j]) * 2048.0f);
            bh2[j] = (_Float16)b; bl2[j] = (_Float16)((b - (float)bh2[j]) * 2048.0f);
            // u: nothing scaled
            ahu[j] = (_Float16)a; alu[j] = (_Float16)(a - (float)ahu[j]);
            bhu[j] = (_Float16)b; blu[j] = (_Float16)(b - (float)bhu[j]);
        }
        acc16 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16, b16, acc16, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, s1, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, s1, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, s1, 0, 0, 0);
        s2a = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah2, bh2, s2a, 0, 0, 0);
        s2b = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah2, bl2, s2b, 0, 0, 0);
        s2b = __builtin_amdgcn_mfma_f32_16x16x32_f16(al2, bh2, s2b, 0, 0, 0);
        u = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahu, bhu, u, 0, 0, 0);
        u = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahu, blu, u, 0, 0, 0);
        u = __builtin_amdgcn_mfma_f32_16x16x32_f16(alu, bhu, u, 0, 0, 0);
    }
    for (int k = 0; k < K; k += 4) {
        const float a = A[lx * K + k + q], b = Bt[lx * K + k + q];
        acc32 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc32, 0, 0, 0);
    }
    // C/D layout: col = lane & 15, row = (lane >> 4) * 4 + r; A row index = lx for the A operand => D[row = A row][col = B col]
    for (int r = 0; r < 4; ++r) {
        const int row = q * 4 + r, col = lx, o = row * 16 + col;
        out[0 * 256 + o] = acc32[r];
        out[1 * 256 + o] = acc16[r];
        out[2 * 256 + o] = s1[r] / (sa * sb);
        out[3 * 256 + o] = s2a[r] + s2b[r] * (1.0f / 2048.0f);
        out[4 * 256 + o] = u[r];
    }
}

int main()
{
    const int Ks[] = {288, 864, 1728, 3456};
    const float wscales[] = {1.0f, 0.05f, 0.002f};
    const float xscales[] = {1.0f, 0.01f};
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    float *dA, *dB, *dO;
    hipMalloc(&dA, 16 * 4096 * 4); hipMalloc(&dB, 16 * 4096 * 4); hipMalloc(&dO, 5 * 256 * 4);
    printf("K wmag xmag | err/rms(out): f32mfma f16 split1 split2 split_unscaled\n");
    for (int K : Ks)
        for (float ws : wscales)
            for (float xs : xscales) {
                double e[5] = {0, 0, 0, 0, 0}, m[5] = {0, 0, 0, 0, 0}, rms = 0;
                const int trials = 20;
                for (int t = 0; t < trials; ++t) {
                    std::vector<float> A(16 * K), B(16 * K), O(5 * 256);
                    float amax = 0;
                    for (auto& v : A) { v = nd(rng) * ws; amax = std::max(amax, std::fabs(v)); }
                    for (auto& v : B) { v = std::max(0.f, nd(rng)) * xs; }          // post-ReLU activations
                    int ea; std::frexp(amax, &ea);                                   // amax in [2^(ea-1), 2^ea)
                    const float sa = std::ldexp(1.0f, 15 - ea), sb = 16.0f;          // weights up to [2^14, 2^15), activations x 2^4
                    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
                    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
                    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, K, sa, sb, dO);
                    hipMemcpy(O.data(), dO, O.size() * 4, hipMemcpyDeviceToHost);
                    for (int i = 0; i < 16; ++i)
                        for (int j = 0; j < 16; ++j) {
                            double ref = 0;
                            for (int k = 0; k < K; ++k) ref += (double)A[i * K + k] * (double)B[j * K + k];
                            rms += ref * ref;
                            for (int md = 0; md < 5; ++md) {
                                const double d = (double)O[md * 256 + i * 16 + j] - ref;
                                e[md] += d * d; m[md] = std::max(m[md], std::fabs(d));
                            }
                        }
                }
                rms = std::sqrt(rms / (trials * 256));
                printf("%4d %.3g %.3g | rms-rel:", K, ws, xs);
                for (int md = 0; md < 5; ++md) printf(" %.3e", std::sqrt(e[md] / (trials * 256)) / rms);
                printf(" | max-rel:");
                for (int md = 0; md < 5; ++md) printf(" %.3e", m[md] / rms);
                printf("\n");
            }
    // subnormal test: A = 2^-20 (fp16 subnormal), B = 1: does the f16 MFMA keep it?
    {
        std::vector<float> A(16 * 32, 0.f), B(16 * 32, 0.f), O(5 * 256);
        A[0] = std::ldexp(1.0f, -20); B[0] = 1.0f;      // D[0][0] = 2^-20 through the plain fp16 path (mode 1)
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, 32, 1.0f, 1.0f, dO);
        hipMemcpy(O.data(), dO, O.size() * 4, hipMemcpyDeviceToHost);
        printf("subnormal fp16 A input 2^-20 * 1 through f16 MFMA: %g (expected %g) => %s\n", O[256], std::ldexp(1.0, -20), O[256] != 0 ? "kept" : "FLUSHED");
    }
    return 0;
}
