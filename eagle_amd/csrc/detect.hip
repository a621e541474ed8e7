// YOLOv8 Detect tail on the GPU: DFL decode + class sigmoid + confidence filter (K6), then per-frame
// descending-confidence sort, class-offset IoU suppression, scale_boxes and the reference's detection ->
// object-dict integer rules (K7).  Replaces ultralytics' Detect inference path, ops.non_max_suppression +
// torchvision.ops.nms + scale_boxes and eagle/models/coordinate_model.py:598-627.
#include "common.h"
#include "dmath.h"

namespace eagle {

// ------------------------------------------------------------------------------------------------------------
// K6: a workgroup decodes 64 consecutive anchors of one pyramid level of one frame.  The 64 x (64 box + 16 class) logits are
// read with fully coalesced 16-byte loads (the anchors' rows are contiguous in the NHWC head outputs) into LDS; then one
// thread per (anchor, box side) runs the DFL soft-max expectation over its 16 logits in the oracle's order (sequential fp32
// sum, oracle/eo_prims.c::eo_yolo_decode_level), and one thread per anchor assembles the box and the best class.
// Round 1's one-thread-per-anchor kernel walked 256-byte rows per lane (0.34-0.67 TB/s); rows in LDS are padded by 4 dwords
// so that the (anchor, side) threads' ds_read_b128 are conflict-free.
// ------------------------------------------------------------------------------------------------------------
struct DecodeArgs { DetLevel lv[3]; int n_lv, n, nc, A; float floor_; DetScratch sc; int blk0[4]; };   // blk0[l]: first block of level l within a frame
#define DEC_ANCH 64
#define DEC_BS 68            // LDS row stride of the box logits (dwords)
#define DEC_CS 20            // ... of the class logits

__global__ __launch_bounds__(256) void yolo_decode_kernel(DecodeArgs a)
{
    __shared__ __attribute__((aligned(16))) float sbox[DEC_ANCH * DEC_BS];
    __shared__ __attribute__((aligned(16))) float scls[DEC_ANCH * DEC_CS];
    __shared__ float sd[DEC_ANCH * 4];
    const int tid = threadIdx.x;
    const int f = blockIdx.y;
    int l = 0;
    while (l + 1 < a.n_lv && (int)blockIdx.x >= a.blk0[l + 1]) ++l;
    const DetLevel& L = a.lv[l];
    const int cell0 = ((int)blockIdx.x - a.blk0[l]) * DEC_ANCH, ncell = L.gh * L.gw;
    const int na = min(DEC_ANCH, ncell - cell0);
    {   // coalesced staging: box rows are 64 floats (box.cs == 64: the head conv owns its tensor), class rows cls.cs floats
        const float4* bsrc = (const float4*)((const float*)L.box.p + ((size_t)f * ncell + cell0) * L.box.cs + L.box.off);
        const int bq = L.box.cs / 4;                       // float4 per row
        for (int e = tid; e < na * 16; e += 256) {
            const int r = e >> 4, q = e & 15;
            *(float4*)(sbox + r * DEC_BS + q * 4) = bsrc[r * bq + q];
        }
        const float4* csrc = (const float4*)((const float*)L.cls.p + ((size_t)f * ncell + cell0) * L.cls.cs + L.cls.off);
        const int cq = L.cls.cs / 4, cn = (a.nc + 3) / 4;
        for (int e = tid; e < na * cn; e += 256) {
            const int r = e / cn, q = e - r * cn;
            *(float4*)(scls + r * DEC_CS + q * 4) = csrc[r * cq + q];
        }
    }
    __syncthreads();
    {   // (anchor, side): DFL expectation, sequential order
        const int r = tid >> 2, sde = tid & 3;
        if (r < na) {
            float lg[16];
            const float4* q = (const float4*)(sbox + r * DEC_BS + sde * 16);
#pragma unroll
            for (int i = 0; i < 4; ++i) { const float4 v = q[i]; lg[4 * i] = v.x; lg[4 * i + 1] = v.y; lg[4 * i + 2] = v.z; lg[4 * i + 3] = v.w; }
            float m = lg[0];
#pragma unroll
            for (int i = 1; i < 16; ++i) m = lg[i] > m ? lg[i] : m;
            float den = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) { lg[i] = d_expf(lg[i] - m); den = den + lg[i]; }
            float num = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) num = fmaf((float)i, lg[i] / den, num);
            sd[r * 4 + sde] = num;
        }
    }
    __syncthreads();
    if (tid < na) {
        const int cell = cell0 + tid, an = L.a0 + cell;
        const int gy = cell / L.gw, gx = cell - gy * L.gw;
        const float ax = (float)gx + 0.5f, ay = (float)gy + 0.5f;
        const float4 d = *(const float4*)(sd + tid * 4);
        const float x1 = ax - d.x, y1 = ay - d.y, x2 = ax + d.z, y2 = ay + d.w;
        const float cx = ((x1 + x2) / 2.0f) * L.stride, cy = ((y1 + y2) / 2.0f) * L.stride;
        const float bw = (x2 - x1) * L.stride, bh = (y2 - y1) * L.stride;
        const float hw = bw / 2.0f, hh = bh / 2.0f;
        float best = -1.f; int bj = 0;
        for (int c = 0; c < a.nc; ++c) {
            const float pr = d_sigmoidf(scls[tid * DEC_CS + c]);
            if (pr > best) { best = pr; bj = c; }
        }
        const size_t o = (size_t)f * a.A + an;
        *(float4*)(a.sc.boxes + o * 4) = make_float4(cx - hw, cy - hh, cx + hw, cy + hh);
        a.sc.conf[o] = best;
        a.sc.cls[o] = bj;
        if (best > a.floor_) {
            const int pos = atomicAdd(a.sc.count + f, 1);
            a.sc.keys[(size_t)f * a.A + pos] = ((unsigned long long)__float_as_uint(best) << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)an);
        }
    }
}

void yolo_decode_launch(const DetLevel* lv, int n_lv, int n, int nc, float conf_floor, const DetScratch& sc, hipStream_t s)
{
    DecodeArgs a;
    for (int i = 0; i < n_lv; ++i) a.lv[i] = lv[i];
    a.n_lv = n_lv; a.n = n; a.nc = nc; a.A = sc.A; a.floor_ = conf_floor; a.sc = sc;
    int nblk = 0;
    for (int i = 0; i < n_lv; ++i) {
        if (lv[i].box.cs % 4 || lv[i].cls.cs % 4 || lv[i].box.off % 4 || lv[i].cls.off % 4 || nc > 16 || !lv[i].box.f32 || !lv[i].cls.f32)
            fail(EAGLE_E_INVALID, "yolo_decode: head tensors must be fp32 with 16-byte aligned rows and at most 16 classes");
        a.blk0[i] = nblk;
        nblk += (lv[i].gh * lv[i].gw + DEC_ANCH - 1) / DEC_ANCH;
    }
    a.blk0[3] = nblk;
    HIP_CHECK(hipMemsetAsync(sc.count, 0, sizeof(int) * n, s));
    hipLaunchKernelGGL(yolo_decode_kernel, dim3(nblk, n), dim3(256), 0, s, a);
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------------------
// K7: one workgroup (1024 threads) per frame.
// ------------------------------------------------------------------------------------------------------------
#define NMS_T 1024
struct NmsArgs { DetScratch sc; PostParams pp; EagleFrameResult* out; int cap; };

__global__ __launch_bounds__(NMS_T) void nms_kernel(NmsArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long* keys = (unsigned long long*)smem;            // cap entries
    unsigned char* dead = (unsigned char*)(keys + a.cap);            // cap bytes
    __shared__ int kept[EAGLE_MAX_DET];
    __shared__ int s_nk;
    const int f = blockIdx.x, tid = threadIdx.x;
    const int A = a.sc.A;
    const int cnt = min(a.sc.count[f], A);
    int np2 = 1;
    while (np2 < cnt) np2 <<= 1;
    const unsigned long long* gk = a.sc.keys + (size_t)f * A;
    for (int i = tid; i < np2; i += NMS_T) { keys[i] = i < cnt ? gk[i] : 0ull; dead[i] = 0; }
    __syncthreads();
    // bitonic sort, descending
    for (int k = 2; k <= np2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < np2; i += NMS_T) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long x = keys[i], y = keys[ixj];
                    const bool desc = (i & k) == 0;
                    if (desc ? (x < y) : (x > y)) { keys[i] = y; keys[ixj] = x; }
                }
            }
            __syncthreads();
        }
    const float* boxes = a.sc.boxes + (size_t)f * A * 4;
    const int* cls = a.sc.cls + (size_t)f * A;
    if (tid == 0) s_nk = 0;
    __syncthreads();
    // greedy suppression in sorted order (torchvision.ops.nms on boxes + cls*7680, IoU > thr suppresses), 64 candidates at a time (round 5; until then: two workgroup
    // barriers per KEPT box, 226 us per launch).  (1) ONE wave walks the block sequentially with no barrier: lane l holds candidate b0 + l, the 64-bit `alive` mask is
    // uniform, the box of the next alive candidate is broadcast with a cross-lane read and every later lane tests itself against it (one ballot per kept box).  (2) after
    // ONE barrier all threads test the candidates behind the block against the block's kept boxes (staged in LDS).  The same comparisons with the same fp32 expression as
    // the one-box-at-a-time loop, in an order that cannot change their outcome: a candidate dies iff some EARLIER KEPT box overlaps it, and whether a box is kept is decided
    // before anything behind it is looked at.  The 300-box cap (ultralytics max_det) stops the walk at the first alive candidate that would be number 301, as before.
    __shared__ float s_kb[64][5];                              // kept boxes of the block: x1, y1, x2, y2 (class offset added), area
    __shared__ int s_kc, s_stop;
    if (tid == 0) s_stop = 0;
    __syncthreads();
    for (int b0 = 0; b0 < cnt; b0 += 64) {
        if (tid < 64) {
            const int j = b0 + tid;
            const bool valid = j < cnt && !dead[j];
            float x1 = 0.f, y1 = 0.f, x2 = 0.f, y2 = 0.f, area = 0.f;
            if (valid) {
                const int aj = (int)(0xFFFFFFFFu - (unsigned)(keys[j] & 0xFFFFFFFFull));
                const float offj = (float)cls[aj] * 7680.0f;
                const float4 bj4 = *(const float4*)(boxes + (size_t)aj * 4);
                x1 = bj4.x + offj; y1 = bj4.y + offj; x2 = bj4.z + offj; y2 = bj4.w + offj;
                area = (x2 - x1) * (y2 - y1);
            }
            unsigned long long alive = __ballot(valid);
            int nk = s_nk, kc = 0, stop = 0;
            while (alive) {
                const int i = __builtin_ctzll(alive);
                if (nk >= EAGLE_MAX_DET) { stop = 1; break; }
                alive &= alive - 1;                            // box b0 + i is kept
                const float bx1 = __shfl(x1, i), by1 = __shfl(y1, i), bx2 = __shfl(x2, i), by2 = __shfl(y2, i), areai = __shfl(area, i);
                if (tid == i) { kept[nk] = b0 + i; s_kb[kc][0] = bx1; s_kb[kc][1] = by1; s_kb[kc][2] = bx2; s_kb[kc][3] = by2; s_kb[kc][4] = areai; }
                ++nk; ++kc;
                const float xx1 = fmaxf(bx1, x1), yy1 = fmaxf(by1, y1), xx2 = fminf(bx2, x2), yy2 = fminf(by2, y2);
                const float iw = fmaxf(0.f, xx2 - xx1), ih = fmaxf(0.f, yy2 - yy1);
                const float inter = iw * ih;
                const float ovr = inter / (areai + area - inter);
                alive &= ~__ballot(tid > i && ovr > a.pp.nms_iou);
            }
            if (tid == 0) { s_nk = nk; s_kc = kc; s_stop = stop; }
        }
        __syncthreads();
        if (s_stop) break;
        const int kc = s_kc;
        if (kc > 0)
            for (int j = b0 + 64 + tid; j < cnt; j += NMS_T) {
                if (dead[j]) continue;
                const int aj = (int)(0xFFFFFFFFu - (unsigned)(keys[j] & 0xFFFFFFFFull));
                const float offj = (float)cls[aj] * 7680.0f;
                const float4 bj4 = *(const float4*)(boxes + (size_t)aj * 4);
                const float cx1 = bj4.x + offj, cy1 = bj4.y + offj, cx2 = bj4.z + offj, cy2 = bj4.w + offj;
                const float areaj = (cx2 - cx1) * (cy2 - cy1);
                for (int k = 0; k < kc; ++k) {
                    const float bx1 = s_kb[k][0], by1 = s_kb[k][1], bx2 = s_kb[k][2], by2 = s_kb[k][3], areai = s_kb[k][4];
                    const float xx1 = fmaxf(bx1, cx1), yy1 = fmaxf(by1, cy1), xx2 = fminf(bx2, cx2), yy2 = fminf(by2, cy2);
                    const float iw = fmaxf(0.f, xx2 - xx1), ih = fmaxf(0.f, yy2 - yy1);
                    const float inter = iw * ih;
                    const float ovr = inter / (areai + areaj - inter);
                    if (ovr > a.pp.nms_iou) { dead[j] = 1; break; }
                }
            }
        __syncthreads();                                       // dead[] of the next block is final; s_kb may be rewritten
    }
    __syncthreads();
    const int K = s_nk;
    EagleFrameResult* R = a.out + f;
    if (tid == 0) { R->n_det = K; R->n_candidates = cnt; }
    // scale_boxes + detection -> object rules (cm.py:598-627)
    const double gain_d = fmin((double)a.pp.in_h / a.pp.frame_h, (double)a.pp.in_w / a.pp.frame_w);
    const float gain = (float)gain_d;
    const float padx = (float)nearbyint((a.pp.in_w - a.pp.frame_w * gain_d) / 2 - 0.1);
    const float pady = (float)nearbyint((a.pp.in_h - a.pp.frame_h * gain_d) / 2 - 0.1);
    const float fw = (float)a.pp.frame_w, fh = (float)a.pp.frame_h;
    for (int k = tid; k < K; k += NMS_T) {
        const int i = kept[k];
        const int ai = (int)(0xFFFFFFFFu - (unsigned)(keys[i] & 0xFFFFFFFFull));
        const float4 b = *(const float4*)(boxes + (size_t)ai * 4);
        EagleDet d;
        d.x1 = fminf(fmaxf((b.x - padx) / gain, 0.f), fw);
        d.y1 = fminf(fmaxf((b.y - pady) / gain, 0.f), fh);
        d.x2 = fminf(fmaxf((b.z - padx) / gain, 0.f), fw);
        d.y2 = fminf(fmaxf((b.w - pady) / gain, 0.f), fh);
        d.conf = __uint_as_float((unsigned)(keys[i] >> 32));
        d.cls = cls[ai];
        int ix1 = (int)d.x1, iy1 = (int)d.y1, ix2 = (int)d.x2, iy2 = (int)d.y2;
        const bool conf_ok = !((double)d.conf < a.pp.detector_conf);
        d.id = -1; d.reported = 0;
        if (d.cls == 0 || d.cls == 1) {
            ix1 = min(max(ix1, 0), a.pp.frame_w - 1); ix2 = min(max(ix2, 0), a.pp.frame_w - 1);
            iy1 = min(max(iy1, 0), a.pp.frame_h - 1); iy2 = min(max(iy2, 0), a.pp.frame_h - 1);
            d.id = k; d.reported = conf_ok;
        } else if (d.cls == 2) {
            int e = 0;
            for (int m = 0; m < k; ++m) {
                const int am = (int)(0xFFFFFFFFu - (unsigned)(keys[kept[m]] & 0xFFFFFFFFull));
                e += (cls[am] == 2);
            }
            d.id = e; d.reported = conf_ok;
        }
        d.bx1 = ix1; d.by1 = iy1; d.bx2 = ix2; d.by2 = iy2;
        d.foot_x = (ix1 + ix2) / 2; d.foot_y = iy2;
        d.pitch_xf = d.pitch_yf = 0.f; d.pitch_x = d.pitch_y = 0; d.in_bounds = 0; d.pad[0] = d.pad[1] = 0;
        R->det[k] = d;
    }
}

void nms_launch(const DetScratch& sc, int n, const PostParams& pp, EagleFrameResult* d_out, hipStream_t s)
{
    NmsArgs a; a.sc = sc; a.pp = pp; a.out = d_out;
    int cap = 1;
    while (cap < sc.A) cap <<= 1;
    a.cap = cap;
    const size_t lds = (size_t)cap * 9;
    ensure_max_dynamic_lds((const void*)nms_kernel, 160 * 1024 - 4096);
    if (lds > 160 * 1024 - 4096) fail(EAGLE_E_INVALID, "too many anchors for the NMS workgroup: %d", sc.A);
    hipLaunchKernelGGL(nms_kernel, dim3(n), dim3(NMS_T), lds, s, a);
    HIP_CHECK(hipGetLastError());
}

}  // namespace eagle
