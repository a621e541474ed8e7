// Experiment (round 2, third design): "ping-pong" 3x3 stride-1 implicit-GEMM convolution.  Eight waves per workgroup = two groups of four
// (one wave of each group per SIMD).  The groups run the same (load | compute) stream one phase apart: while group X issues the 72 MFMAs
// of a (Cin-chunk, kernel-row) stage from fragments it already holds in registers, group Y reads its fragments of its stage from LDS
// and the refill of the LDS ring is requested; s_barrier; roles swap.  A SIMD's matrix pipe therefore always has one wave with nothing but
// MFMAs to issue.  Operands by LDS-DMA (no staging registers, no ds_write) into a two-deep ring that runs CONTINUOUSLY over the
// workgroup's items (the next item's first stages are requested during the current item's last ones); the epilogue is wave-private
// (own LDS strip, no barrier).  Stand-alone harness: correctness spot check + timing.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -ffp-contract=off [-DTIMING] pp_main.hip -o bench_pp
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

using half8 = __attribute__((ext_vector_type(8))) _Float16;
using half4 = __attribute__((ext_vector_type(4))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned int;
using rsrc_t = __amdgpu_buffer_rsrc_t;
#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))
constexpr unsigned OOB_OFF = 0x80000000u;

struct G8Args {
    const _Float16* x; const _Float16* w; const float* bias; _Float16* y; const _Float16* r1;
    int N, H, W, C, CO;          // C: input channels (= pixel stride), CO: output channels (= pixel stride)
    int tiles_x, tiles_y, nchunks, gy, relu;
    void* dbg;
};


// tile 8 x 32 output pixels, halo 10 x 34; 8 waves = 4 (pixel quarters) x 2 (Cout halves = the two ping-pong groups)
template <int NT2>
__global__ __launch_bounds__(512, 1) void conv_pp_kernel(G8Args a)
{
    constexpr int BN = 2 * NT2 * 16;
    constexpr int WST = 3 * 4 * BN * 16;                 // bytes of one weight stage (3 taps x 4 channel groups x BN x 16 B)
    constexpr int WSLABS = WST / 1024;
    constexpr int HW_ = 34, HPIX = 340, HSLABS = 22, HB = HSLABS * 1024;
    constexpr int PW = 4;
    constexpr int BNW = NT2 * 16, RS = BNW * 2 + 16, GO = BNW / 8, STRIP = 16 * RS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const Wb = smem;                                // 2 x WST
    char* const Hb = smem + 2 * WST;                      // 2 x HB
    char* const strip = smem + 2 * WST + 2 * HB + (threadIdx.x >> 6) * STRIP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave & 3, wn = wave >> 2, q = lane >> 4, lx = lane & 15;
    const int grp = wn;
    const int gy = a.gy, nitems = a.tiles_x * a.tiles_y * a.N * gy, nwg = gridDim.x;
    int item0, item_end;
    {
        const int b = blockIdx.x, xcd = b & 7, k = b >> 3;
        const int wgs_here = (nwg + 7 - xcd) >> 3;
        const int qn = nitems >> 3, rn = nitems & 7;
        const int x0 = xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn;
        const int xc = qn + (xcd < rn ? 1 : 0);
        item0 = x0 + (int)((long)xc * k / wgs_here);
        item_end = x0 + (int)((long)xc * (k + 1) / wgs_here);
    }
    if (item0 >= item_end) return;
    const int nloc = item_end - item0;
    const int S = a.nchunks * 3, G = nloc * S, GC = nloc * a.nchunks;
    const rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, 0x7FFFFFFF, 0x00020000);
    const rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, 0x7FFFFFFF, 0x00020000);
    // halo DMA geometry of this lane's (at most 3) slabs: pixel, source channel group (swizzled: slots 0/2 and 1/3 trade places on
    // every other group of four pixels, which makes the 16-lane ds_read_b128 groups of a fragment read conflict-free)
    int hrel[3], hyx[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int slab = wave + 8 * k, e = slab * 64 + lane;
        const int pix = e >> 2, slot = e & 3;
        const int hy = pix / HW_, hx = pix - hy * HW_;
        const int cg = slot ^ (((pix >> 2) & 1) << 1);
        hrel[k] = ((hy * a.W + hx) * a.C + cg * 8) * 2;
        hyx[k] = (slab < HSLABS && pix < HPIX) ? ((hy << 16) | hx) : (0x4000 << 16);
    }
    int pbase[PW];
#pragma unroll
    for (int p = 0; p < PW; ++p) {
        const int s = wm * PW + p, row = s >> 1, xb = s & 1;
        pbase[p] = row * HW_ + xb * 16 + lx;
    }
    const int wfrag = (q * BN + wn * NT2 * 16 + lx) * 16;

    auto decode = [&](int item, int& n, int& ty, int& tx, int& nb) {
        int t = item / gy; nb = item - t * gy;
        tx = t % a.tiles_x; t /= a.tiles_x;
        ty = t % a.tiles_y; n = t / a.tiles_y;
    };
    // ---- the refill stream: global stage index = local item * S + stage; ring slot = index & 1 ---------------------------------
    int w_g = 0, w_s = 0, w_item = item0, w_nb;            // next weight stage to request
    { int n_, ty_, tx_; decode(w_item, n_, ty_, tx_, w_nb); }
    auto issue_w = [&]() {
        if (w_g >= G) return;
        const int ch = w_s / 3, ky = w_s - ch * 3;
        const unsigned src = (unsigned)((w_nb * a.nchunks + ch) * 36 + ky * 12) * (unsigned)(BN * 16);
        char* dst = Wb + (w_g & 1) * WST;
        for (int slab = wave; slab < WSLABS; slab += 8)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, LDSP(dst + slab * 1024), 16, (unsigned)(slab * 1024 + lane * 16), src, 0, 0);
        ++w_g;
        if (++w_s == S) { w_s = 0; ++w_item; int n_, ty_, tx_; decode(w_item, n_, ty_, tx_, w_nb); }
    };
    int h_g = 0, h_ch = 0, h_item = item0;                 // next halo chunk to request
    unsigned hoff[3];
    auto halo_offsets = [&](int item) {
        int n, ty, tx, nb; decode(item, n, ty, tx, nb);
        const int iy0 = ty * 8 - 1, ix0 = tx * 32 - 1;
        const int gb = (((n * a.H + iy0) * a.W + ix0) * a.C) * 2;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int iy = iy0 + (hyx[k] >> 16), ix = ix0 + (hyx[k] & 0xFFFF);
            hoff[k] = ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) ? (unsigned)(gb + hrel[k]) : OOB_OFF;
        }
    };
    halo_offsets(h_item);
    auto issue_h = [&]() {
        if (h_g >= GC) return;
        char* dst = Hb + (h_g & 1) * HB;
        const unsigned so = (unsigned)(h_ch * 32) * 2u;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (wave + 8 * k < HSLABS)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, LDSP(dst + (wave + 8 * k) * 1024), 16, hoff[k], so, 0, 0);
        ++h_g;
        if (++h_ch == a.nchunks) { h_ch = 0; ++h_item; if (h_g < GC) halo_offsets(h_item); }
    };
    // phase duty at even phase 2j (j >= 1): request weight stage j + 1 and, when j % 3 == 0, halo chunk j / 3 + 1
    auto duty = [&](int j) {
        issue_w();
        if (j % 3 == 0) issue_h();
    };
    issue_h(); issue_h(); issue_w(); issue_w();            // ring slots 0 and 1
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();            // group Y runs one phase behind
#ifdef TIMING
    long long t_load = 0, t_comp = 0, t_bar = 0, t_epi = 0; long long tprev = __builtin_readcyclecounter(); const long long t_k0 = tprev;
#define TS(var_) { const long long now_ = __builtin_readcyclecounter(); var_ += now_ - tprev; tprev = now_; }
#else
#define TS(var_)
#endif
    int g = 0;
    for (int item = item0; item < item_end; ++item) {
        int n, ty, tx, nb; decode(item, n, ty, tx, nb);
        const int oy0 = ty * 8, ox0 = tx * 32;
        f32x4 acc[NT2][PW];
#pragma unroll
        for (int i = 0; i < NT2; ++i)
#pragma unroll
            for (int p = 0; p < PW; ++p) acc[i][p] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < S; ++s, ++g) {
            // ---- load phase: X at phase 2g (even: refill duty j = g), Y at phase 2g + 1 (odd: certifies its requests)
            if (grp == 0 && g >= 1) duty(g);
            const int ky = s % 3;
            const char* wb = Wb + (g & 1) * WST + wfrag;
            const char* hb = Hb + ((g / 3) & 1) * HB;
            half8 wa[3][NT2], xb[3][PW];
            auto read_frags = [&](int kx) {
#pragma unroll
                for (int tt = 0; tt < NT2; ++tt) wa[kx][tt] = *(const half8*)(wb + kx * (4 * BN * 16) + tt * 256);
#pragma unroll
                for (int p = 0; p < PW; ++p) {
                    const int pix = pbase[p] + ky * HW_ + kx;
                    xb[kx][p] = *(const half8*)(hb + pix * 64 + ((q ^ (((pix >> 2) & 1) << 1)) << 4));
                }
            };
            read_frags(0); read_frags(1);
            if (NT2 < 6) read_frags(2);                    // NT2 = 6: the third K-step's fragments take the first one's registers during the compute phase
            if (grp == 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            TS(t_load)
            __builtin_amdgcn_s_barrier();
            TS(t_bar)
            // ---- compute phase: X at phase 2g + 1 (odd: certifies), Y at phase 2g + 2 (even: refill duty j = g + 1)
            if (grp == 1) duty(g + 1);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
                for (int tt = 0; tt < NT2; ++tt)
#pragma unroll
                    for (int p = 0; p < PW; ++p)
                        acc[tt][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[kx][tt], xb[kx][p], acc[tt][p], 0, 0, 0);
                if (NT2 >= 6 && kx == 0) { __builtin_amdgcn_sched_barrier(0); read_frags(2); __builtin_amdgcn_sched_barrier(0); }
            }
            TS(t_comp)
            if (s == S - 1) {
                // epilogue, wave-private: bias, residual, ReLU in the MFMA layout; fp16 through this wave's LDS strip; 16-byte stores
                const int co0 = nb * BN + wn * BNW + q * 4;
#pragma unroll
                for (int p = 0; p < PW; ++p) {
                    const int s2 = wm * PW + p, row = s2 >> 1, xb2 = s2 & 1;
                    const int oy = oy0 + row, ox = ox0 + xb2 * 16 + lx;
                    const bool ok = oy < a.H && ox < a.W;
                    const size_t pidx = (size_t)(n * a.H + oy) * a.W + ox;
#pragma unroll
                    for (int tt = 0; tt < NT2; ++tt) {
                        const float4 bv = *(const float4*)(a.bias + co0 + tt * 16);
                        float v[4] = {acc[tt][p][0] + bv.x, acc[tt][p][1] + bv.y, acc[tt][p][2] + bv.z, acc[tt][p][3] + bv.w};
                        if (a.r1 && ok) {
                            const half4 rv = *(const half4*)(a.r1 + pidx * a.CO + co0 + tt * 16);
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = (float)rv[r] + v[r];
                        }
                        if (a.relu) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
                        }
                        half4 o = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                        *(half4*)(strip + lx * RS + (tt * 16 + q * 4) * 2) = o;
                    }
#pragma unroll
                    for (int e0 = 0; e0 < 16 * GO; e0 += 64) {
                        const int e = e0 + lane;
                        const int px = e / GO, gq = e - px * GO;
                        const int ox2 = ox0 + xb2 * 16 + px;
                        if (e < 16 * GO && oy < a.H && ox2 < a.W) {
                            const u32x4 v = *(const u32x4*)(strip + px * RS + gq * 16);
                            *(u32x4*)(a.y + ((size_t)(n * a.H + oy) * a.W + ox2) * a.CO + nb * BN + wn * BNW + gq * 8) = v;
                        }
                    }
                }
                TS(t_epi)
            }
            if (grp == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            TS(t_bar)
        }
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
#ifdef TIMING
    if (blockIdx.x == 17 && lane == 0) {
        long long* d = (long long*)a.dbg + wave * 8;
        d[0] = t_load; d[1] = t_bar; d[2] = t_comp; d[3] = t_epi; d[4] = __builtin_readcyclecounter() - t_k0; d[5] = G;
    }
#endif
}

// ---- harness ------------------------------------------------------------------------------------------------------------------
struct Shape { int n, h, w, c; const char* name; };
template <int NT2>
static void run(const Shape& sh, int wgs_per_cu)
{
    constexpr int BN = 2 * NT2 * 16;
    const int cin = sh.c, cout = sh.c, nch = cin / 32, gy = cout / BN;
    if (cout % BN) { printf("%s: cout %% %d != 0\n", sh.name, BN); return; }
    const size_t nx = (size_t)sh.n * sh.h * sh.w * cin, ny = (size_t)sh.n * sh.h * sh.w * cout;
    std::vector<_Float16> hx(nx);
    for (size_t i = 0; i < nx; ++i) hx[i] = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
    std::vector<float> hw((size_t)9 * cin * cout), hb(cout, 0.1f);
    for (auto& v : hw) v = (rand() % 2001 - 1000) / 20000.0f;
    // weight image [nb][chunk][tap*4 + cg][BN][8]
    std::vector<_Float16> tw((size_t)gy * nch * 36 * BN * 8);
    size_t o = 0;
    for (int b = 0; b < gy; ++b)
        for (int ch = 0; ch < nch; ++ch)
            for (int g = 0; g < 36; ++g)
                for (int nn = 0; nn < BN; ++nn)
                    for (int j = 0; j < 8; ++j) {
                        const int tap = g / 4, cg = g % 4;
                        tw[o++] = (_Float16)hw[((size_t)tap * cin + ch * 32 + cg * 8 + j) * cout + b * BN + nn];
                    }
    void *dx, *dy, *dw, *db;
    hipMalloc(&dx, nx * 2); hipMalloc(&dy, ny * 2); hipMalloc(&dw, tw.size() * 2); hipMalloc(&db, cout * 4);
    hipMemcpy(dx, hx.data(), nx * 2, hipMemcpyHostToDevice); hipMemcpy(dw, tw.data(), tw.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(db, hb.data(), cout * 4, hipMemcpyHostToDevice);
    hipMemset(dy, 0xFF, ny * 2);
    G8Args a{(const _Float16*)dx, (const _Float16*)dw, (const float*)db, (_Float16*)dy, (const _Float16*)dx, sh.n, sh.h, sh.w, cin, cout,
             (sh.w + 31) / 32, (sh.h + 7) / 8, nch, gy, 1, nullptr};
    hipMalloc(&a.dbg, 4096); hipMemset(a.dbg, 0, 4096);
    constexpr int WST = 3 * 4 * BN * 16;
    const size_t lds = (size_t)2 * WST + 2 * 22 * 1024 + 8 * 16 * (NT2 * 32 + 16);
    hipFuncSetAttribute((const void*)conv_pp_kernel<NT2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int items = a.tiles_x * a.tiles_y * sh.n * gy;
    const int grid = std::min(items, 256 * wgs_per_cu);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(conv_pp_kernel<NT2>, dim3(grid), dim3(512), lds, nullptr, a);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed: %s\n", sh.name, hipGetErrorString(hipGetLastError())); return; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int R = 20;
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < R; ++i) hipLaunchKernelGGL(conv_pp_kernel<NT2>, dim3(grid), dim3(512), lds, nullptr, a);
    hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<_Float16> hy(ny); hipMemcpy(hy.data(), dy, ny * 2, hipMemcpyDeviceToHost);
    double maxerr = 0; int nan = 0;
    for (size_t i = 0; i < ny; ++i) nan += ((float)hy[i] != (float)hy[i]);
    unsigned long long lcg = 12345;
    for (int t = 0; t < 600; ++t) {
        lcg = lcg * 6364136223846793005ULL + 1442695040888963407ULL;
        size_t oo = (size_t)(lcg >> 20) % ny;
        if (t < 60) {                                      // image corners / edges explicitly
            const int yy = (t & 1) ? sh.h - 1 : 0, xx = (t & 2) ? sh.w - 1 : (t & 4 ? 33 % sh.w : 0);
            oo = (((size_t)(t % sh.n) * sh.h + yy) * sh.w + xx) * cout + (t * 7) % cout;
        }
        const int co = (int)(oo % cout); size_t pp = oo / cout;
        const int ox = (int)(pp % sh.w); pp /= sh.w; const int oy = (int)(pp % sh.h); const int nn = (int)(pp / sh.h);
        float acc = 0.f;
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = oy - 1 + ky, ix = ox - 1 + kx;
                if (iy < 0 || iy >= sh.h || ix < 0 || ix >= sh.w) continue;
                for (int ci = 0; ci < cin; ++ci)
                    acc += (float)hx[(((size_t)nn * sh.h + iy) * sh.w + ix) * cin + ci] * (float)(_Float16)hw[((size_t)(ky * 3 + kx) * cin + ci) * cout + co];
            }
        float v = acc + 0.1f + (float)hx[(((size_t)nn * sh.h + oy) * sh.w + ox) * cin + co];
        v = v > 0 ? v : 0;
        const double e = fabs((double)v - (double)(float)hy[oo]) / (1.0 + fabs((double)v));
        if (e > maxerr) maxerr = e;
    }
#ifdef TIMING
    { long long d[64]; hipMemcpy(d, a.dbg, sizeof(d), hipMemcpyDeviceToHost);
      for (int w = 0; w < 8; ++w) printf("   wave %d: %lld ticks over %lld stages: load phases %lld, barriers %lld, compute phases %lld, epilogues %lld\n", w, d[w*8+4], d[w*8+5], d[w*8+0], d[w*8+1], d[w*8+2], d[w*8+3]); }
#endif
    const double fl = 2.0 * sh.n * sh.h * sh.w * (double)cout * cin * 9;
    printf("%-20s pp NT2=%d BN=%d wgs/cu=%d items=%d  %8.1f us  %7.1f TFLOP/s   NaN %d  max rel err %.2e\n", sh.name, NT2, BN, wgs_per_cu, items,
           ms / R * 1e3, fl / (ms / R * 1e-3) / 1e12, nan, maxerr);
    hipFree(dx); hipFree(dy); hipFree(dw); hipFree(db);
}

int main(int argc, char** argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 8;
    const int wgs = argc > 2 ? atoi(argv[2]) : 1;
    Shape s96{B, 68, 120, 96, "96->96@68x120"}, s192{B, 34, 60, 192, "192->192@34x60"}, s384{B, 17, 30, 384, "384->384@17x30"};
    if (!getenv("ONLY192")) run<3>(s96, wgs);
    run<6>(s192, wgs);
    if (!getenv("ONLY192")) { run<3>(s192, wgs); run<6>(s384, wgs); }
    return 0;
}
