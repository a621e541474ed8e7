// Developer experiment: do an HBM-bound convolution (48->48@135x240, weight-stationary kernel) and an MFMA-bound one (192->192@34x60 or
// 96->96@68x120, A-direct kernel) finish sooner when they run CONCURRENTLY on two streams than back to back?  Persistent kernels fill
// the chip, so co-residency needs each of them restricted to half of its resident workgroups (EAGLE_CONV_AD_SLOTS / EAGLE_CONV_WS_PER_CU).
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -ffp-contract=off -DCONV_SRC='"../../eagle_amd/csrc/conv.hip"' -I../../eagle_amd/csrc -I../../include co_main.hip -o bench_co
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include CONV_SRC
#ifndef CONV_SINGLE_FILE      // the library keeps its kernel instances in separate translation units (build time); a tool is one file
#include "../../eagle_amd/csrc/conv_inst_0.hip"
#include "../../eagle_amd/csrc/conv_inst_1.hip"
#include "../../eagle_amd/csrc/conv_inst_2.hip"
#include "../../eagle_amd/csrc/conv_inst_3.hip"
#include "../../eagle_amd/csrc/conv_ad_s1.hip"
#include "../../eagle_amd/csrc/conv_ad_s2.hip"
#endif
namespace eagle {
void ensure_max_dynamic_lds(const void* fn, int bytes) { (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); }
void fail(int code, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n"); exit(1); }
}
using namespace eagle;
struct Layer { ConvLaunch L; double flop; const char* name; };
static Layer make(int n, int h, int w, int c, const char* name)
{
    Layer r; r.name = name;
    ConvLaunch& L = r.L;
    L.cfg = conv_choose(EAGLE_PREC_F16, 3, 1, c, c, w, true);
    const size_t nx = (size_t)n * h * w * c;
    std::vector<_Float16> hx(nx);
    for (size_t i = 0; i < nx; ++i) hx[i] = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
    std::vector<float> hw((size_t)9 * c * c), hb(c, 0.1f);
    for (auto& v : hw) v = (rand() % 2001 - 1000) / 20000.0f;
    std::vector<char> tiled(conv_weight_elems(EAGLE_PREC_F16, L.cfg) * 2);
    conv_tile_weights(EAGLE_PREC_F16, L.cfg, hw.data(), c, c, tiled.data());
    void *dx, *dy, *dw, *db;
    (void)hipMalloc(&dx, nx * 2); (void)hipMalloc(&dy, nx * 2); (void)hipMalloc(&dw, tiled.size()); (void)hipMalloc(&db, c * 4);
    (void)hipMemcpy(dx, hx.data(), nx * 2, hipMemcpyHostToDevice); (void)hipMemcpy(dw, tiled.data(), tiled.size(), hipMemcpyHostToDevice);
    (void)hipMemcpy(db, hb.data(), c * 4, hipMemcpyHostToDevice);
    L.x.p = dx; L.x.n = n; L.x.h = h; L.x.w = w; L.x.c = L.x.cs = c;
    L.y = L.x; L.y.p = dy; L.r1 = L.x;
    L.w = dw; L.bias = (const float*)db; L.post_act = 1;
    r.flop = 2.0 * n * h * w * (double)c * c * 9;
    printf("%s: kc=%d nt=%d variant=%d\n", name, L.cfg.kc, L.cfg.nt, L.cfg.variant);
    return r;
}
int main(int argc, char** argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 50;
    Layer a = make(B, 135, 240, 48, "48->48@135x240"), b = make(B, 34, 60, 192, "192->192@34x60"), c = make(B, 68, 120, 96, "96->96@68x120");
    hipStream_t s0, s1; (void)hipStreamCreate(&s0); (void)hipStreamCreate(&s1);
    hipEvent_t e0, e1, j; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventCreate(&j);
    const int R = 16;
    auto timed = [&](const char* what, auto fn) {
        fn(); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0, s0);
        fn();
        (void)hipEventRecord(e1, s0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-60s %8.1f us per (pair)\n", what, ms / R * 1e3);
    };
    for (Layer* m : {&b, &c}) {
        printf("---- %s with %s\n", a.name, m->name);
        timed("48->48 alone", [&] { for (int i = 0; i < R; ++i) conv_launch(EAGLE_PREC_F16, a.L, s0); });
        timed("partner alone", [&] { for (int i = 0; i < R; ++i) conv_launch(EAGLE_PREC_F16, m->L, s0); });
        timed("back to back on one stream", [&] { for (int i = 0; i < R; ++i) { conv_launch(EAGLE_PREC_F16, a.L, s0); conv_launch(EAGLE_PREC_F16, m->L, s0); } });
        timed("two streams (s1 forks from and joins s0)", [&] {
            (void)hipEventRecord(j, s0); (void)hipStreamWaitEvent(s1, j, 0);
            for (int i = 0; i < R; ++i) { conv_launch(EAGLE_PREC_F16, a.L, s0); conv_launch(EAGLE_PREC_F16, m->L, s1); }
            (void)hipEventRecord(j, s1); (void)hipStreamWaitEvent(s0, j, 0);
        });
    }
    return 0;
}
