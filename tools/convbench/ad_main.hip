// Experiment (round 2, fourth design): "A-direct" 3x3 stride-1 implicit-GEMM convolution.  Eight waves per workgroup = PG pixel groups x CQ
// Cout groups; a wave owns 8 sub-tiles of 16 pixels x 48 output channels (24 MFMAs per K-step of 32).
//   * WEIGHT fragments never touch LDS: the weight image is fragment-major (one 16-byte piece per lane), so a wave loads its three
//     A fragments of a K-step with three coalesced buffer_load_dwordx4 straight into registers, two K-steps ahead (ring of three);
//   * ACTIVATIONS: halo tile of a 32-channel chunk by LDS-DMA into a two-deep ring that runs continuously over the workgroup's items,
//     one s_barrier per chunk (9 K-steps = 216 MFMAs per wave); B fragments by conflict-free ds_read_b128 one K-step ahead;
//   * epilogue wave-private (own LDS strip, no barrier).
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -ffp-contract=off [-DTIMING] ad_main.hip -o bench_ad
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


// CQ = 4: tile 8 x 32 pixels x 192 channels (waves: 2 pixel halves x 4 Cout quarters); CQ = 2: tile 16 x 32 x 96 (4 x 2)
template <int CQ, int PG, bool RES = true>
__global__ __launch_bounds__(64 * CQ * PG, 512 / (64 * CQ * PG)) void conv_ad_kernel(G8Args a)
{
#if __HIP_DEVICE_COMPILE__        // (the host pass of this hipcc drops the stub of this template when it sees the body; tool file only)
    constexpr int NT = 3, PW = 8, NW = CQ * PG, BN = CQ * 48, TH = 4 * PG;
    constexpr int KSTEP = 4 * BN * 16;                   // bytes of one K-step (32 input channels of one tap) of the weight image
    constexpr int HW_ = 34, HPIX = (TH + 2) * HW_, PS = 96;   // pixel stride 96 B = 4 channel groups + 2 pad slots: lanes lx -> consecutive pixels, q -> +16 B is conflict-free for ds_read_b128
    constexpr int HSLABS = ((HPIX * PS + 1023) / 1024 + NW - 1) / NW * NW, HB = HSLABS * 1024, HK = HSLABS / NW;      // every wave requests the same number of slabs (static vmcnt arithmetic)
    constexpr int RS = 48 * 2 + 16, STRIP = 32 * RS;     // one output row of the wave (32 pixels x 48 channels) per pass
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const Hb = smem;                                // 2 x HB
    char* const strip = smem + 2 * HB + (threadIdx.x >> 6) * STRIP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cq = wave % CQ, pg = wave / CQ, q = lane >> 4, lx = lane & 15;
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
    const int nloc = item_end - item0, nch = a.nchunks, GC = nloc * nch;
    const rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, 0x7FFFFFFF, 0x00020000);
    const rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, 0x7FFFFFFF, 0x00020000);
    const rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.r1, 0, 0x7FFFFFFF, 0x00020000);
    const rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, 0x7FFFFFFF, 0x00020000);
    int hpk[HK];                                          // per slab: halo row | column << 8 | channel slot << 16, or -1 (padding)
#pragma unroll
    for (int k = 0; k < HK; ++k) {
        const int e = (wave + NW * k) * 64 + lane;
        const int pix = e / 6, slot = e - pix * 6;
        const int hy = pix / HW_, hx = pix - hy * HW_;
        hpk[k] = (pix < HPIX && slot < 4) ? (hy | (hx << 8) | (slot << 16)) : -1;
    }
    const int bbase = ((pg * 4) * HW_ + lx) * PS + q * 16;      // B fragments: one address register, everything else is an immediate
    const unsigned wlane = (unsigned)((q * BN + cq * 48 + lx) * 16);

    auto decode = [&](int item, int& n, int& ty, int& tx, int& nb) {
        int t = item / gy; nb = item - t * gy;
        tx = t % a.tiles_x; t /= a.tiles_x;
        ty = t % a.tiles_y; n = t / a.tiles_y;
    };
    // ---- halo ring: global chunk index = local item * nch + chunk; slot = index & 1 -------------------------------------------
    int h_g = 0, h_ch = 0, h_item = item0;
    int h_iy0, h_ix0, h_gb;                                // geometry of the item the next request belongs to (scalars)
    auto halo_origin = [&](int item) {
        int n, ty, tx, nb; decode(item, n, ty, tx, nb);
        h_iy0 = ty * TH - 1; h_ix0 = tx * 32 - 1;
        h_gb = (((n * a.H + h_iy0) * a.W + h_ix0) * a.C) * 2;
    };
    halo_origin(h_item);
    auto issue_h = [&]() {
        char* dst = Hb + (h_g & 1) * HB;                  // (past the last chunk: the same requests again, harmlessly, so that the count stays static)
        const unsigned so = (unsigned)(h_ch * 32) * 2u;
#pragma unroll
        for (int k = 0; k < HK; ++k) {
            const int hy = hpk[k] & 0xFF, hx = (hpk[k] >> 8) & 0xFF, slot = (hpk[k] >> 16) & 7;
            const int iy = h_iy0 + hy, ix = h_ix0 + hx;
            const unsigned off = (hpk[k] >= 0 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                                     ? (unsigned)(h_gb + ((hy * a.W + hx) * a.C + slot * 8) * 2) : OOB_OFF;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, LDSP(dst + (wave + NW * k) * 1024), 16, off, so, 0, 0);
        }
        if (h_g + 1 >= GC) return;
        ++h_g;
        if (++h_ch == nch) { h_ch = 0; ++h_item; halo_origin(h_item); }
    };
    issue_h();
#ifdef TIMING
    long long t_wait = 0, t_main = 0, t_epi = 0, t_eld = 0; const long long rt0 = __builtin_amdgcn_s_memrealtime(); long long tprev = __builtin_readcyclecounter(); const long long t_k0 = tprev;
#define TS(var_) { const long long now_ = __builtin_readcyclecounter(); var_ += now_ - tprev; tprev = now_; }
#else
#define TS(var_)
#endif
    int gc = 0;
    u32x4 A[3][NT];
    for (int item = item0; item < item_end; ++item) {
        int n, ty, tx, nb; decode(item, n, ty, tx, nb);
        const int oy0 = ty * TH, ox0 = tx * 32;
        f32x4 acc[NT][PW];
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int p = 0; p < PW; ++p) acc[i][p] = f32x4{0.f, 0.f, 0.f, 0.f};
        // A ring: fragments of K-steps kk, kk + 1, kk + 2 (slot = kk % 3; nine K-steps per chunk keep the slots static).  The ring runs on
        // across items: the last two requests of an item fetch K-steps 0 and 1 of the NEXT item (its Cout block may differ).
        unsigned wsrc = (unsigned)(nb * nch * 36) * (unsigned)(BN * 16);     // K-step 0 of this item
        const unsigned wend = wsrc + (unsigned)(nch * 9) * KSTEP;
        unsigned wnext;
        { int n2, ty2, tx2, nb2; decode(item + 1 < item_end ? item + 1 : item, n2, ty2, tx2, nb2); wnext = (unsigned)(nb2 * nch * 36) * (unsigned)(BN * 16); }
        auto load_a = [&](int slot) {
            const unsigned so = wsrc < wend ? wsrc : wnext + (wsrc - wend);
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) A[slot][tt] = __builtin_amdgcn_raw_buffer_load_b128(wrs, wlane + tt * 256, so, 0);
            wsrc += KSTEP;
        };
#ifndef ACONT
#define ACONT 1
#endif
        if (!ACONT || item == item0) { load_a(0); load_a(1); } else wsrc += 2 * KSTEP;
        for (int ch = 0; ch < nch; ++ch, ++gc) {
            // chunk gc has landed (requested one chunk ago); every wave is done with the other slot
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");        // everything older than the two K-steps of A in flight
            TS(t_wait)
            __builtin_amdgcn_s_barrier();
            issue_h();                                              // chunk gc + 1 into the slot chunk gc - 1 used
            const char* hb = Hb + (gc & 1) * HB + bbase;
            half8 B[2][PW];
            auto read_b = [&](int slot, int kk) {
                const int ky = kk / 3, kx = kk - ky * 3;
#pragma unroll
                for (int p = 0; p < PW; ++p)
                    B[slot][p] = *(const half8*)(hb + (((p >> 1) + ky) * HW_ + (p & 1) * 16 + kx) * PS);
            };
            read_b(0, 0);
#pragma unroll
            for (int kk = 0; kk < 9; ++kk) {
#ifndef RPRE
#define RPRE 1
#endif
                if (RPRE && kk >= 7 && ch == nch - 1 && RES) {
                    // the item's last two K-steps have no weights left to request: their ring slots take the residual pieces of output rows
                    // 0 and 1 instead (same number of requests on both paths, so every vmcnt stays static)
                    const int r2 = kk - 7, oy = oy0 + pg * 4 + r2;
                    int lane_ = lane;
                    asm volatile("" : "+v"(lane_));        // keeps the piece geometry out of the loop-invariant registers
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const int e = i * 64 + lane_, px = e / 6, gq = e - px * 6, ox2 = ox0 + px;
                        const unsigned off = (oy < a.H && ox2 < a.W) ? (unsigned)((((n * a.H + oy) * a.W + ox2) * a.CO + nb * BN + cq * 48 + gq * 8) * 2) : OOB_OFF;
                        A[(kk + 2) % 3][i] = __builtin_amdgcn_raw_buffer_load_b128(rrs, off, 0, 0);
                    }
                    wsrc += KSTEP;
                } else
                load_a((kk + 2) % 3);
                if (kk + 1 < 9) read_b((kk + 1) & 1, kk + 1);
#pragma unroll
                for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                    for (int p = 0; p < PW; ++p)
                        acc[tt][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16((half8)A[kk % 3][tt], B[kk & 1][p], acc[tt][p], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            TS(t_main)
        }
        // epilogue, wave-private, no barrier.  The residual is fetched as coalesced 16-byte pieces for all four rows at once (one memory
        // latency per item), goes through the strip into the MFMA layout, and the result returns through the strip as 16-byte pieces.
        const int co0 = nb * BN + cq * 48;
        __builtin_amdgcn_sched_barrier(0);
        float4 bias[NT];
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) bias[tt] = *(const float4*)(a.bias + co0 + tt * 16 + q * 4);
        // piece e = i * 64 + lane of a row: pixel e / 6, 16-byte group e % 6; out-of-image pieces get the out-of-range offset (loads return
        // zeros, stores are dropped): no divergent branches
        // piece e = i * 64 + lane of a row: pixel e / 6, 16-byte group e % 6; out-of-image pieces get the out-of-range offset (loads return
        // zeros, stores are dropped): no divergent branches.  Offsets are recomputed where they are used (registers are scarce here).
        int pstrip[3], ppx[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int e = i * 64 + lane, px = e / 6, gq = e - px * 6;
            pstrip[i] = px * RS + gq * 16;
            ppx[i] = px | (gq << 8);
        }
        auto piece_off = [&](int r2, int i) -> unsigned {
            const int oy = oy0 + pg * 4 + r2, ox2 = ox0 + (ppx[i] & 0xFF), gq = ppx[i] >> 8;
            return (oy < a.H && ox2 < a.W) ? (unsigned)((((n * a.H + oy) * a.W + ox2) * a.CO + co0 + gq * 8) * 2) : OOB_OFF;
        };
#ifndef EPI
#define EPI 0
#endif
#if EPI == 0
        u32x4 rres[4][3];
        if (RES) {
#pragma unroll
            for (int r2 = 0; r2 < 4; ++r2)
#pragma unroll
                for (int i = 0; i < 3; ++i) rres[r2][i] = (RPRE && r2 < 2) ? A[r2][i] : __builtin_amdgcn_raw_buffer_load_b128(rrs, piece_off(r2, i), 0, 0);
        }
#else
        // residual straight in the MFMA layout: 8 bytes per lane (a pixel's 96 bytes are completed by this wave's 12 requests)
        u32x2 rres[4][2][NT];
        unsigned qoff[4][2];
#pragma unroll
        for (int r2 = 0; r2 < 4; ++r2)
#pragma unroll
            for (int xb2 = 0; xb2 < 2; ++xb2) {
                const int oy = oy0 + pg * 4 + r2, ox = ox0 + xb2 * 16 + lx;
                qoff[r2][xb2] = (oy < a.H && ox < a.W) ? (unsigned)((((n * a.H + oy) * a.W + ox) * a.CO + co0 + q * 4) * 2) : OOB_OFF;
            }
        if (RES) {
#pragma unroll
            for (int r2 = 0; r2 < 4; ++r2)
#pragma unroll
                for (int xb2 = 0; xb2 < 2; ++xb2)
#pragma unroll
                    for (int tt = 0; tt < NT; ++tt) rres[r2][xb2][tt] = __builtin_amdgcn_raw_buffer_load_b64(rrs, qoff[r2][xb2] + tt * 32, 0, 0);
        }
#endif
#if defined(TIMING) && TIMING == 2
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TS(t_eld)
#endif
#pragma unroll
        for (int r2 = 0; r2 < 4; ++r2) {
#if EPI == 0
            if (RES) {
#pragma unroll
                for (int i = 0; i < 3; ++i) *(u32x4*)(strip + pstrip[i]) = rres[r2][i];
            }
#endif
#pragma unroll
            for (int xb2 = 0; xb2 < 2; ++xb2) {
                const int p = r2 * 2 + xb2;
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) {
                    char* sp = strip + (xb2 * 16 + lx) * RS + (tt * 16 + q * 4) * 2;
                    float v[4] = {acc[tt][p][0] + bias[tt].x, acc[tt][p][1] + bias[tt].y, acc[tt][p][2] + bias[tt].z, acc[tt][p][3] + bias[tt].w};
                    if (RES) {
#if EPI == 0
                        const half4 rv = *(const half4*)sp;
#else
                        const half4 rv = (half4)rres[r2][xb2][tt];
#endif
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = (float)rv[r] + v[r];
                    }
                    if (a.relu) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
                    }
                    half4 o = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
#if EPI == 2
                    __builtin_amdgcn_raw_buffer_store_b64((u32x2)o, yrs, qoff[r2][xb2] + tt * 32, 0, 0);
#else
                    *(half4*)sp = o;
#endif
                }
            }
#if EPI != 2
#pragma unroll
            for (int i = 0; i < 3; ++i)
                __builtin_amdgcn_raw_buffer_store_b128(*(const u32x4*)(strip + pstrip[i]), yrs, piece_off(r2, i), 0, 0);
#endif
        }
        TS(t_epi)
    }
#ifdef TIMING
    if (blockIdx.x == 17 && lane == 0) {
        long long* d = (long long*)a.dbg + wave * 8;
        d[0] = t_wait; d[1] = t_main; d[2] = t_epi; d[3] = t_eld; d[4] = __builtin_readcyclecounter() - t_k0; d[5] = GC; d[6] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
#endif
#endif
}

// ---- harness ------------------------------------------------------------------------------------------------------------------
struct Shape { int n, h, w, c; const char* name; };
template <int CQ, int PG>
static void run(const Shape& sh)
{
    constexpr int BN = CQ * 48, NT2 = CQ, TH = 4 * PG, NW = CQ * PG, wgs_per_cu = 8 / NW;
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
             (sh.w + 31) / 32, (sh.h + TH - 1) / TH, nch, gy, 1, nullptr};
    hipMalloc(&a.dbg, 4096); hipMemset(a.dbg, 0, 4096);
    const size_t lds = (size_t)2 * (((((TH + 2) * 34 * 96 + 1023) / 1024 + NW - 1) / NW * NW) * 1024) + NW * 32 * 112;
    hipFuncSetAttribute((const void*)conv_ad_kernel<CQ, PG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int items = a.tiles_x * a.tiles_y * sh.n * gy;
    const int grid = std::min(items, 256 * wgs_per_cu);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((conv_ad_kernel<CQ, PG>), dim3(grid), dim3(64 * NW), lds, nullptr, a);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed: %s\n", sh.name, hipGetErrorString(hipGetLastError())); return; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int R = 20;
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < R; ++i) hipLaunchKernelGGL((conv_ad_kernel<CQ, PG>), dim3(grid), dim3(64 * NW), lds, nullptr, a);
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
      for (int w = 0; w < 1; ++w) printf("   wave %d: %lld ticks (%.0f MHz) over %lld chunks: waits %lld, barrier+main %lld, epilogues %lld + their loads %lld\n", w, d[w*8+4], d[w*8+4] / (d[w*8+6] / 100.0), d[w*8+5], d[w*8+0], d[w*8+1], d[w*8+2], d[w*8+3]); }
#endif
    const double fl = 2.0 * sh.n * sh.h * sh.w * (double)cout * cin * 9;
    printf("%-20s ad CQ=%d PG=%d BN=%d wgs/cu=%d items=%d  %8.1f us  %7.1f TFLOP/s   NaN %d  max rel err %.2e\n", sh.name, CQ, PG, BN, wgs_per_cu, items,
           ms / R * 1e3, fl / (ms / R * 1e-3) / 1e12, nan, maxerr);
    hipFree(dx); hipFree(dy); hipFree(dw); hipFree(db);
}

int main(int argc, char** argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 8;
    const int wgs = argc > 2 ? atoi(argv[2]) : 1;
    Shape s96{B, 68, 120, 96, "96->96@68x120"}, s192{B, 34, 60, 192, "192->192@34x60"}, s384{B, 17, 30, 384, "384->384@17x30"};
    (void)wgs;
    run<2, 4>(s96); run<2, 2>(s96);
    run<4, 2>(s192); run<4, 1>(s192);
    run<4, 2>(s384); run<4, 1>(s384);
    return 0;
}
