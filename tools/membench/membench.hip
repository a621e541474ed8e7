// Developer micro-benchmark: streaming read / write / copy bandwidth of one MI355X with 16-byte accesses (not part of the product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k_write(uint4* y, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = make_uint4(i, 1, 2, 3); }
__global__ void k_read(const uint4* x, size_t n, uint4* sink) { uint4 a = make_uint4(0, 0, 0, 0); for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = x[i]; a.x ^= v.x; a.y ^= v.y; a.z ^= v.z; a.w ^= v.w; } if (a.x == 0x12345) *sink = a; }
__global__ void k_copy(const uint4* x, uint4* y, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = x[i]; }
__global__ void k_copy2(const uint4* x, const uint4* r, uint4* y, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 a = x[i], b = r[i]; y[i] = make_uint4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); } }
int main(int argc, char** argv)
{
    const size_t mb = argc > 1 ? atoi(argv[1]) : 156;
    const size_t bytes = mb << 20, n = bytes / 16;
    uint4 *x, *y, *r; hipMalloc(&x, bytes); hipMalloc(&y, bytes); hipMalloc(&r, bytes); hipMemset(x, 1, bytes); hipMemset(r, 2, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {256 * 4, 256 * 8, 256 * 16, 256 * 64}) {
        for (int mode = 0; mode < 4; ++mode) {
            float best = 1e9;
            for (int it = 0; it < 6; ++it) {
                hipEventRecord(e0);
                for (int k = 0; k < 5; ++k) {
                    if (mode == 0) k_write<<<grid, 256>>>(y, n);
                    if (mode == 1) k_read<<<grid, 256>>>(x, n, y);
                    if (mode == 2) k_copy<<<grid, 256>>>(x, y, n);
                    if (mode == 3) k_copy2<<<grid, 256>>>(x, r, y, n);
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5; if (ms < best) best = ms;
            }
            const char* nm[] = {"write", "read", "copy", "add2"};
            const double moved = bytes * (mode == 2 ? 2.0 : (mode == 3 ? 3.0 : 1.0));
            printf("%zu MB grid %5d %-5s %7.1f us  %6.2f TB/s\n", mb, grid, nm[mode], best * 1e3, moved / (best * 1e-3) / 1e12);
        }
    }
    return 0;
}
