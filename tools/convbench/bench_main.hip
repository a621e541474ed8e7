// Developer micro-benchmark: times one convolution shape with the kernel family compiled in via -DCONV_SRC=...
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
struct Shape { int n, h, w, cin, cout, ks, s; const char* name; };
int main(int argc, char** argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 8;
    Shape shapes[] = {{B, 135, 240, 48, 48, 3, 1, "48->48@135x240"}, {B, 68, 120, 96, 96, 3, 1, "96->96@68x120"},
                      {B, 34, 60, 192, 192, 3, 1, "192->192@34x60"}, {B, 17, 30, 384, 384, 3, 1, "384->384@17x30"},
                      {B, 135, 240, 64, 256, 1, 1, "64->256 1x1@135x240"}, {B, 135, 240, 48, 96, 3, 2, "48->96 s2"}};
    for (auto& sh : shapes) {
        const int ho = (sh.h + 2 * (sh.ks / 2) - sh.ks) / sh.s + 1, wo = (sh.w + 2 * (sh.ks / 2) - sh.ks) / sh.s + 1;
        ConvLaunch L;
        L.cfg = conv_choose(EAGLE_PREC_F16, sh.ks, sh.s, sh.cin, sh.cout, wo, true);
        if (getenv("KC")) L.cfg.kc = atoi(getenv("KC"));
        if (getenv("NT")) L.cfg.nt = atoi(getenv("NT"));
        if (getenv("VAR")) L.cfg.variant = atoi(getenv("VAR"));
        if (getenv("WX")) L.cfg.wx = atoi(getenv("WX"));
        if (getenv("KC") && atoi(getenv("KC")) == 0) L.cfg.kc = sh.cin;
        if (getenv("ONLY") && atoi(getenv("ONLY")) != sh.cin) continue;
        if (!conv_supported(EAGLE_PREC_F16, L.cfg)) { printf("%s: unsupported kc=%d nt=%d\n", sh.name, L.cfg.kc, L.cfg.nt); continue; }
        size_t nx = (size_t)sh.n * sh.h * sh.w * sh.cin, ny = (size_t)sh.n * ho * wo * sh.cout;
        std::vector<_Float16> hx(nx);
        for (size_t i = 0; i < nx; ++i) hx[i] = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
        std::vector<float> hw((size_t)sh.ks * sh.ks * sh.cin * sh.cout), hb(sh.cout, 0.1f);
        for (auto& v : hw) v = (rand() % 2001 - 1000) / 20000.0f;
        std::vector<char> tiled(conv_weight_elems(EAGLE_PREC_F16, L.cfg) * 2);
        conv_tile_weights(EAGLE_PREC_F16, L.cfg, hw.data(), sh.cin, sh.cout, tiled.data());
        void *dx, *dy, *dw, *db;
        hipMalloc(&dx, nx * 2); hipMalloc(&dy, ny * 2); hipMalloc(&dw, tiled.size()); hipMalloc(&db, sh.cout * 4);
        hipMemcpy(dx, hx.data(), nx * 2, hipMemcpyHostToDevice); hipMemcpy(dw, tiled.data(), tiled.size(), hipMemcpyHostToDevice);
        hipMemcpy(db, hb.data(), sh.cout * 4, hipMemcpyHostToDevice);
        hipMemset(dy, 0xFF, ny * 2);                         // unwritten outputs stay NaN and poison the checksum
        L.x.p = dx; L.x.n = sh.n; L.x.h = sh.h; L.x.w = sh.w; L.x.c = L.x.cs = sh.cin;
        L.y.p = dy; L.y.n = sh.n; L.y.h = ho; L.y.w = wo; L.y.c = L.y.cs = sh.cout;
        L.r1 = L.x; L.r1.p = (sh.s == 1 && sh.cin == sh.cout) ? dx : nullptr;
        L.w = dw; L.bias = (const float*)db; L.post_act = 1;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 3; ++i) conv_launch(EAGLE_PREC_F16, L, nullptr);
        hipDeviceSynchronize();
        const int R = 20;
        hipEventRecord(e0, nullptr);
        for (int i = 0; i < R; ++i) conv_launch(EAGLE_PREC_F16, L, nullptr);
        hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fl = 2.0 * sh.n * ho * wo * (double)sh.cout * sh.cin * sh.ks * sh.ks;
        std::vector<_Float16> hy(ny); hipMemcpy(hy.data(), dy, ny * 2, hipMemcpyDeviceToHost);
        double cs = 0; for (size_t i = 0; i < ny; i += 97) cs += (float)hy[i];
        // spot check against a direct fp32 convolution of the fp16-rounded operands (bias 0.1, residual = x when shapes allow, ReLU)
        double maxerr = 0; int nan = 0;
        for (size_t i = 0; i < ny; ++i) nan += ((float)hy[i] != (float)hy[i]);
        unsigned long long lcg = 12345;
        for (int t = 0; t < 400; ++t) {
            lcg = lcg * 6364136223846793005ULL + 1442695040888963407ULL;
            const size_t o = (size_t)(lcg >> 20) % ny;
            const int co = (int)(o % sh.cout); size_t pp = o / sh.cout;
            const int ox = (int)(pp % wo); pp /= wo; const int oy = (int)(pp % ho); const int nn = (int)(pp / ho);
            float acc = 0.f;
            for (int ky = 0; ky < sh.ks; ++ky)
                for (int kx = 0; kx < sh.ks; ++kx) {
                    const int iy = oy * sh.s - sh.ks / 2 + ky, ix = ox * sh.s - sh.ks / 2 + kx;
                    if (iy < 0 || iy >= sh.h || ix < 0 || ix >= sh.w) continue;
                    for (int ci = 0; ci < sh.cin; ++ci)
                        acc += (float)hx[(((size_t)nn * sh.h + iy) * sh.w + ix) * sh.cin + ci] * (float)(_Float16)hw[((size_t)(ky * sh.ks + kx) * sh.cin + ci) * sh.cout + co];
                }
            float v = acc + 0.1f;
            if (sh.s == 1 && sh.cin == sh.cout) v += (float)hx[(((size_t)nn * sh.h + oy) * sh.w + ox) * sh.cin + co];
            v = v > 0 ? v : 0;
            const double e = fabs((double)v - (double)(float)hy[o]) / (1.0 + fabs((double)v));
            if (e > maxerr) maxerr = e;
        }
        printf("   [check] NaN outputs %d, max rel err of 400 samples %.2e\n", nan, maxerr);
        printf("%-22s kc=%2d nt=%d var=%d wx=%d  %8.1f us  %7.1f TFLOP/s  checksum %.3f\n", sh.name, L.cfg.kc, L.cfg.nt, L.cfg.variant, L.cfg.wx, ms / R * 1e3, fl / (ms / R * 1e-3) / 1e12, cs);
        hipFree(dx); hipFree(dy); hipFree(dw); hipFree(db);
    }
    return 0;
}
