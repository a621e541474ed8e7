// Developer micro-benchmark: NHWC tile-shaped copy (the convolution's access pattern without any math) vs a linear copy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
struct A { const uint4* x; const uint4* r; uint4* y; int N, H, W, G, TH, TW, tiles_x, tiles_y, mode, halo, persistent; };
// mode bit0: read x tile, bit1: read r tile, bit2: write y tile.  G = 16-byte groups per pixel.
__global__ __launch_bounds__(256) void k_tile(A a)
{
    const int ntiles = a.tiles_x * a.tiles_y * a.N;
    int t0 = blockIdx.x, t1 = blockIdx.x + 1;
    if (a.persistent) { t0 = (int)((long)ntiles * blockIdx.x / gridDim.x); t1 = (int)((long)ntiles * (blockIdx.x + 1) / gridDim.x); }
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int t = t0; t < t1; ++t) {
        const int tx = t % a.tiles_x, r = t / a.tiles_x, ty = r % a.tiles_y, n = r / a.tiles_y;
        const int oy0 = ty * a.TH, ox0 = tx * a.TW;
        if (a.mode & 1) {
            const int hh = a.TH + 2 * a.halo, hw = a.TW + 2 * a.halo;
            for (int i = threadIdx.x; i < hh * hw * a.G; i += 256) {
                const int pix = i / a.G, g = i - pix * a.G, hy = pix / hw, hx = pix - hy * hw;
                const int iy = oy0 - a.halo + hy, ix = ox0 - a.halo + hx;
                if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) { uint4 v = a.x[((size_t)(n * a.H + iy) * a.W + ix) * a.G + g]; acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w; }
            }
        }
        for (int i = threadIdx.x; i < a.TH * a.TW * a.G; i += 256) {
            const int pix = i / a.G, g = i - pix * a.G, py = pix / a.TW, px = pix - py * a.TW;
            const int oy = oy0 + py, ox = ox0 + px;
            if (oy < a.H && ox < a.W) {
                const size_t o = ((size_t)(n * a.H + oy) * a.W + ox) * a.G + g;
                uint4 v = acc;
                if (a.mode & 2) { uint4 rr = a.r[o]; v.x += rr.x; v.y ^= rr.y; v.z += rr.z; v.w ^= rr.w; }
                if (a.mode & 4) a.y[o] = v; else if (v.x == 0x1234567) a.y[0] = v;
            }
        }
    }
}
int main(int argc, char** argv)
{
    const int N = 50, H = 135, W = 240, C = argc > 1 ? atoi(argv[1]) : 48, G = C / 8;
    const size_t n16 = (size_t)N * H * W * G, bytes = n16 * 16;
    uint4 *x, *y, *r; CK(hipMalloc(&x, bytes)); CK(hipMalloc(&y, bytes)); CK(hipMalloc(&r, bytes)); CK(hipMemset(x, 1, bytes)); CK(hipMemset(r, 2, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int cfg = 0; cfg < 3; ++cfg) {
        const int TH = cfg == 0 ? 16 : (cfg == 1 ? 8 : 4), TW = cfg == 0 ? 16 : (cfg == 1 ? 32 : 64);
        for (int persistent = 0; persistent < 2; ++persistent)
            for (int halo = 0; halo < 2; ++halo)
                for (int mode : {1, 4, 5, 7}) {
                    A a{x, r, y, N, H, W, G, TH, TW, (W + TW - 1) / TW, (H + TH - 1) / TH, mode, halo, persistent};
                    const int ntiles = a.tiles_x * a.tiles_y * N, grid = persistent ? 1024 : ntiles;
                    float best = 1e9;
                    for (int it = 0; it < 4; ++it) {
                        CK(hipEventRecord(e0));
                        for (int k = 0; k < 5; ++k) k_tile<<<grid, 256>>>(a);
                        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5; if (ms < best) best = ms;
                    }
                    const int nb = ((mode & 1) ? 1 : 0) + ((mode & 2) ? 1 : 0) + ((mode & 4) ? 1 : 0);
                    printf("C=%d tile %2dx%2d persistent=%d halo=%d mode=%d  %7.1f us  %5.2f TB/s (algorithmic, %d x %zu MB)\n", C, TH, TW, persistent, halo, mode, best * 1e3,
                           nb * (double)bytes / (best * 1e-3) / 1e12, nb, bytes >> 20);
                }
    }
    return 0;
}
