// Autotune driver: for every layer shape in argv[1] (lines "ks,s,cin,cout,h,w,n") time every supported fp16
// (KC, NT, wx) configuration; prints "T,ks,s,cin,cout,h,w,n,kc,nt,wx,variant,us".
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <tuple>
#include <vector>
#include CONV_SRC
namespace eagle {
void fail(int code, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n"); throw 1; }
}
using namespace eagle;
int main(int argc, char** argv)
{
    FILE* f = fopen(argv[1], "r");
    std::set<std::tuple<int, int, int, int, int, int, int>> shapes;
    int ks, s, cin, cout, h, w, n;
    while (fscanf(f, "%d,%d,%d,%d,%d,%d,%d", &ks, &s, &cin, &cout, &h, &w, &n) == 7) shapes.insert({ks, s, cin, cout, h, w, n});
    size_t maxb = 0;
    for (auto& sh : shapes) {
        std::tie(ks, s, cin, cout, h, w, n) = sh;
        maxb = std::max(maxb, (size_t)n * h * w * std::max(cin, cout) * 2);
    }
    void *dx, *dy, *dw, *db;
    hipMalloc(&dx, maxb); hipMalloc(&dy, maxb); hipMalloc(&dw, 64 << 20); hipMalloc(&db, 1 << 16);
    {   // random operands (constant fills clock higher and mis-rank configurations)
        std::vector<_Float16> hr(maxb / 2); unsigned st = 12345u;
        for (auto& v : hr) { st = st * 1664525u + 1013904223u; v = (_Float16)(((int)(st >> 16) % 2001 - 1000) / 1000.0f); }
        hipMemcpy(dx, hr.data(), maxb, hipMemcpyHostToDevice);
        std::vector<_Float16> hw2((64 << 20) / 2);
        for (auto& v : hw2) { st = st * 1664525u + 1013904223u; v = (_Float16)(((int)(st >> 16) % 2001 - 1000) / 20000.0f); }
        hipMemcpy(dw, hw2.data(), 64 << 20, hipMemcpyHostToDevice);
    } hipMemset(db, 0, 1 << 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    static const int kcs[] = {96, 64, 48, 32, 16, 8}, nts[] = {6, 4, 3, 2, 1};
    static const int variants[] = {0, 2, 3, 4, 6, 7};     // 1, 5 (LDS-DMA) only exist under -DEAGLE_CONV_EXPERIMENTAL
    for (auto& sh : shapes) {
        std::tie(ks, s, cin, cout, h, w, n) = sh;
        const int ho = (h + 2 * (ks / 2) - ks) / s + 1, wo = (w + 2 * (ks / 2) - ks) / s + 1;
        for (int variant : variants) for (int kc : kcs) for (int nt : nts) for (int wx = 1; wx <= 2; ++wx) {
            if (cin % kc || cout % (16 * nt)) continue;
            if ((variant == 6 || variant == 7) && kc != cin) continue;
            ConvLaunch L;
            L.cfg.ks = ks; L.cfg.stride = s; L.cfg.kc = kc; L.cfg.nt = nt; L.cfg.wx = wx; L.cfg.cin = cin; L.cfg.cout_pad = cout; L.cfg.variant = variant;
            if (!conv_supported(EAGLE_PREC_F16, L.cfg)) continue;
            if (conv_lds_bytes(EAGLE_PREC_F16, L.cfg) > 160 * 1024 - 256) continue;
            if (conv_weight_elems(EAGLE_PREC_F16, L.cfg) * 2 > (64u << 20)) continue;
            L.x.p = dx; L.x.n = n; L.x.h = h; L.x.w = w; L.x.c = L.x.cs = cin;
            L.y.p = dy; L.y.n = n; L.y.h = ho; L.y.w = wo; L.y.c = L.y.cs = cout;
            L.w = dw; L.bias = (const float*)db; L.post_act = 1;
            if (ks == 3 && s == 1 && cin == cout) { L.r1 = L.x; }                // BasicBlock-style residual (every second 3x3 of HRNet has one)
            if (getenv("TUNE_VERBOSE")) { fprintf(stderr, "try ks=%d s=%d cin=%d cout=%d h=%d w=%d kc=%d nt=%d wx=%d var=%d lds=%zu\n", ks, s, cin, cout, h, w, kc, nt, wx, variant, conv_lds_bytes(EAGLE_PREC_F16, L.cfg)); fflush(stderr); }
            try {
                conv_launch(EAGLE_PREC_F16, L, nullptr);
                if (hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); continue; }
                const int R = 6;
                hipEventRecord(e0, nullptr);
                for (int i = 0; i < R; ++i) conv_launch(EAGLE_PREC_F16, L, nullptr);
                hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                printf("T,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%.2f\n", ks, s, cin, cout, h, w, n, kc, nt, wx, variant, ms / R * 1e3);
            } catch (int) { (void)hipGetLastError(); }
        }
        fflush(stdout);
    }
    return 0;
}
