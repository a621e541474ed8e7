// Timing / autotune driver for the split-precision (EAGLE_PREC_F32S) convolution family.  Links against the built library
// (eagle_amd/libeagle_hip.so exports the internal eagle:: launchers), so it always measures the kernels the product runs.
//   build: hipcc --offload-arch=gfx950 -O2 -std=c++17 -I../../eagle_amd/csrc split_tune.hip -L../../eagle_amd -leagle_hip -Wl,-rpath,'$ORIGIN/../../eagle_amd' -o split_tune.out
//   run:   ./split_tune.out layers.csv [prec]        (layers.csv lines: ks,s,cin,cout,h,w,n as written by EAGLE_DUMP_LAYERS)
// For every layer shape: every (variant, kc, nt, wx) instance that exists and fits in LDS, with the residual / activation pattern the
// shape has in HRNet, random operands (constant fills clock higher and mis-rank configurations).
// Prints "T,ks,s,cin,cout,h,w,n,kc,nt,wx,variant,res,us".
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <tuple>
#include <vector>

#include "common.h"
using namespace eagle;

int main(int argc, char** argv)
{
    const int prec = argc > 2 ? atoi(argv[2]) : EAGLE_PREC_F32S;
    FILE* f = fopen(argv[1], "r");
    if (!f) { fprintf(stderr, "cannot open %s\n", argv[1]); return 1; }
    std::set<std::tuple<int, int, int, int, int, int, int>> shapes;
    int ks, s, cin, cout, h, w, n;
    while (fscanf(f, "%d,%d,%d,%d,%d,%d,%d", &ks, &s, &cin, &cout, &h, &w, &n) == 7) shapes.insert({ks, s, cin, cout, h, w, n});
    const int esz = prec == EAGLE_PREC_F16 ? 2 : 4;
    size_t maxb = 0;
    for (auto& sh : shapes) {
        std::tie(ks, s, cin, cout, h, w, n) = sh;
        maxb = std::max(maxb, (size_t)n * h * w * std::max(cin, cout) * esz);
    }
    void *dx, *dy, *dr, *dw, *db;
    const size_t wbytes = 128u << 20;
    hipMalloc(&dx, maxb); hipMalloc(&dy, maxb); hipMalloc(&dr, maxb); hipMalloc(&dw, wbytes); hipMalloc(&db, 1 << 16);
    {
        std::vector<_Float16> hr(maxb / 2); unsigned st = 12345u;
        for (auto& v : hr) { st = st * 1664525u + 1013904223u; v = (_Float16)(((int)(st >> 16) % 2001 - 1000) / 1000.0f); }
        hipMemcpy(dx, hr.data(), maxb, hipMemcpyHostToDevice);
        hipMemcpy(dr, hr.data(), maxb, hipMemcpyHostToDevice);
        std::vector<_Float16> hw2(wbytes / 2);
        for (auto& v : hw2) { st = st * 1664525u + 1013904223u; v = (_Float16)(((int)(st >> 16) % 2001 - 1000) / 20000.0f); }
        hipMemcpy(dw, hw2.data(), wbytes, hipMemcpyHostToDevice);
    }
    hipMemset(db, 0, 1 << 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    static const int kcs[] = {64, 48, 32, 24, 16, 8}, nts[] = {12, 6, 4, 3, 2, 1};
    static const int variants[] = {0, 2, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 18, 19, 21, 22, 23, 24};
    const char* only = getenv("TUNE_ONLY");          // "variant" or "variant,kc,nt": restrict the sweep
    for (auto& sh : shapes) {
        std::tie(ks, s, cin, cout, h, w, n) = sh;
        const int ho = (h + 2 * (ks / 2) - ks) / s + 1, wo = (w + 2 * (ks / 2) - ks) / s + 1;
        for (int res = 0; res <= ((ks == 3 && s == 1 && cin == cout) ? 1 : 0); ++res)
        for (int variant : variants) for (int kc : kcs) for (int nt : nts) for (int wx = 1; wx <= 3; ++wx) {
            if (cin % kc || cout % (16 * nt)) continue;
            if (variant >= 8 && variant != 18 && wx != 2) continue;
            if ((variant == 18) != (wx == 3)) continue;          // the 8 x 48 tile is the only user of wx = 3
            if (only) { int ov = -1, okc = -1, ont = -1; sscanf(only, "%d,%d,%d", &ov, &okc, &ont); if (ov != variant || (okc > 0 && okc != kc) || (ont > 0 && ont != nt)) continue; }
            ConvLaunch L;
            L.cfg.ks = ks; L.cfg.stride = s; L.cfg.kc = kc; L.cfg.nt = nt; L.cfg.wx = wx; L.cfg.cin = cin; L.cfg.cout_pad = cout; L.cfg.variant = variant;
            if (!conv_supported(prec, L.cfg)) continue;
            if (conv_lds_bytes(prec, L.cfg) > 160 * 1024 - 256) continue;
            if (conv_weight_elems(prec, L.cfg) * 2 > wbytes) continue;
            const int fmt = prec_tensor_fmt(prec);
            L.x.p = dx; L.x.n = n; L.x.h = h; L.x.w = w; L.x.c = L.x.cs = cin; L.x.f32 = fmt;
            L.y.p = dy; L.y.n = n; L.y.h = ho; L.y.w = wo; L.y.c = L.y.cs = cout; L.y.f32 = fmt;
            L.w = dw; L.bias = (const float*)db; L.post_act = 1; L.descale = 1.0f / 65536.0f;
            if (res) { L.r1 = L.y; L.r1.p = dr; }
            try {
                conv_launch(prec, L, nullptr);
                if (hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); continue; }
                const int R = 6;
                hipEventRecord(e0, nullptr);
                for (int i = 0; i < R; ++i) conv_launch(prec, L, nullptr);
                hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                printf("T,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%.2f\n", ks, s, cin, cout, h, w, n, kc, nt, wx, variant, res, ms / R * 1e3);
            } catch (const Err& e) { (void)hipGetLastError(); if (getenv("TUNE_VERBOSE")) fprintf(stderr, "skip: %s\n", e.msg.c_str()); }
        }
        fflush(stdout);
    }
    return 0;
}
