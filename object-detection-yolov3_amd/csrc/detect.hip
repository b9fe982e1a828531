// Detection-side kernels: anchor decode (model.py:122-212), the YOLO loss with its
// analytic backward (model.py:230-354) and class-wise NMS (bbox_utils.py:200-281).
// Compiled with -ffp-contract=off: the NMS arithmetic must round exactly like the
// reference's NumPy float32 elementwise ops so the integer keep indices match.
#include <stdlib.h>

#include "common.h"

#define Y3_MAX_ANCHORS 16
#define Y3_MAX_SCALES 4

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// ---------------------------------------------------------------------------
// decode
// ---------------------------------------------------------------------------
struct DecodeArgs {
    const float* fm[Y3_MAX_SCALES];
    int ld[Y3_MAX_SCALES], gh[Y3_MAX_SCALES], gw[Y3_MAX_SCALES], start[Y3_MAX_SCALES + 1];
    float sx[Y3_MAX_SCALES], sy[Y3_MAX_SCALES];
    float aw[Y3_MAX_ANCHORS], ah[Y3_MAX_ANCHORS];
    int nscales, A, K, N, nb;
    float* out;
};

__global__ void decode_kernel(const DecodeArgs p) {
    const long long total = (long long)p.N * p.nb;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const int D = 5 + p.K;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int n = (int)(i / p.nb);
        const int b = (int)(i - (long long)n * p.nb);
        int s = 0;
        while (s + 1 < p.nscales && b >= p.start[s + 1]) ++s;
        const int r = b - p.start[s];
        const int a = r % p.A;
        const int cell = r / p.A;
        const int gy = cell / p.gw[s], gx = cell - gy * p.gw[s];
        const float* t = p.fm[s] + ((long long)n * p.gh[s] * p.gw[s] + cell) * p.ld[s] + a * D;
        // reorg_layer: box_xy = (sigmoid(t_xy) + offset) * stride ; box_wh = exp(t_wh) * anchor
        const float cx = (sigmoidf_(t[0]) + (float)gx) * p.sx[s];
        const float cy = (sigmoidf_(t[1]) + (float)gy) * p.sy[s];
        const float w = expf(t[2]) * p.aw[a];
        const float h = expf(t[3]) * p.ah[a];
        float* o = p.out + i * D;
        o[0] = cx - w / 2.0f;
        o[1] = cy - h / 2.0f;
        o[2] = cx + w / 2.0f;
        o[3] = cy + h / 2.0f;
        o[4] = sigmoidf_(t[4]);
        for (int k = 0; k < p.K; ++k) o[5 + k] = sigmoidf_(t[5 + k]);
    }
}

extern "C" int y3_decode_fwd(const y3_tensor* fm, int nscales, const float* anchors_host, int num_anchors, int num_classes, int img_h,
                             int img_w, float* out, y3_stream_t stream) {
    Y3_CHECK_ARG(fm && anchors_host && out, "decode_fwd: null pointer");
    Y3_CHECK_ARG(nscales >= 1 && nscales <= Y3_MAX_SCALES, "decode_fwd: nscales %d", nscales);
    Y3_CHECK_ARG(num_anchors >= 1 && num_anchors <= Y3_MAX_ANCHORS && num_classes >= 1, "decode_fwd: anchors/classes");
    DecodeArgs p = {};
    p.nscales = nscales;
    p.A = num_anchors;
    p.K = num_classes;
    p.N = fm[0].n;
    int nb = 0;
    for (int s = 0; s < nscales; ++s) {
        Y3_CHECK_ARG(fm[s].ptr && fm[s].n == p.N && fm[s].c == num_anchors * (5 + num_classes) && fm[s].ld >= fm[s].c, "decode_fwd: feature map %d geometry", s);
        p.fm[s] = fm[s].ptr;
        p.ld[s] = fm[s].ld;
        p.gh[s] = fm[s].h;
        p.gw[s] = fm[s].w;
        p.start[s] = nb;
        nb += fm[s].h * fm[s].w * num_anchors;
        // Q6: stride = img_size[0:2] // grid = (s_y, s_x) multiplies (x, y)
        p.sx[s] = (float)(img_h / fm[s].h);
        p.sy[s] = (float)(img_w / fm[s].w);
    }
    p.start[nscales] = nb;
    p.nb = nb;
    for (int a = 0; a < num_anchors; ++a) {
        p.aw[a] = anchors_host[2 * a];
        p.ah[a] = anchors_host[2 * a + 1];
    }
    p.out = out;
    const long long total = (long long)p.N * nb;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(decode_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    Y3_CHECK_LAUNCH("decode");
    return Y3_OK;
}

// ---------------------------------------------------------------------------
// loss forward + backward, one scale
// ---------------------------------------------------------------------------
struct LossArgs {
    const float* fm;
    const float* gt;
    float* dfm;
    int fm_ld, dfm_ld;
    int N, G_h, G_w, A, K;
    float sx, sy;
    float aw[Y3_MAX_ANCHORS], ah[Y3_MAX_ANCHORS];
    float inv_b, gscale;  // 1/local batch, 1/(local batch * global batch)
    int* present;         // [A] flags: anchor a has at least one GT cell in this batch
    float* partials;      // [blocks][4]
};

__global__ void loss_present_kernel(const float* __restrict__ gt, long long ncell_anchor, int A, int D, int* present) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < ncell_anchor; i += stride)
        if (gt[i * D + 4] != 0.f) atomicOr(&present[i % A], 1);
}

__device__ __forceinline__ float sig_ce(float z, float x) {  // labels z, logits x (App. C6)
    return fmaxf(x, 0.f) - x * z + log1pf(expf(-fabsf(x)));
}

__global__ __launch_bounds__(256) void loss_kernel(const LossArgs p) {
    __shared__ float sm[4][256];
    const int D = 5 + p.K;
    const long long total = (long long)p.N * p.G_h * p.G_w * p.A;
    const long long stride = (long long)gridDim.x * blockDim.x;
    float l_xy = 0.f, l_wh = 0.f, l_obj = 0.f, l_cls = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int a = (int)(i % p.A);
        const long long cell = i / p.A;  // n*G*G + gy*G + gx
        const int gx = (int)(cell % p.G_w);
        const int gy = (int)((cell / p.G_w) % p.G_h);
        const float* t = p.fm + cell * p.fm_ld + a * D;
        const float* g = p.gt + i * D;
        float* d = p.dfm + cell * p.dfm_ld + a * D;
        const float aw = p.aw[a], ah = p.ah[a];
        const float offx = (float)gx, offy = (float)gy;
        const float sgx = sigmoidf_(t[0]), sgy = sigmoidf_(t[1]);
        const float bx = (sgx + offx) * p.sx, by = (sgy + offy) * p.sy;
        const float ew = expf(t[2]), eh = expf(t[3]);
        const float bw = ew * aw, bh = eh * ah;
        const float gm = g[4];

        // ignore mask (Q7): best IoU against origin-centred anchor-sized boxes of the anchors present in the batch
        float best = -INFINITY;
        for (int q = 0; q < p.A; ++q) {
            if (!p.present[q]) continue;
            const float tw = p.aw[q], th = p.ah[q];
            const float ix = fmaxf(fminf(bx + bw / 2.0f, tw / 2.0f) - fmaxf(bx - bw / 2.0f, -tw / 2.0f), 0.f);
            const float iy = fmaxf(fminf(by + bh / 2.0f, th / 2.0f) - fmaxf(by - bh / 2.0f, -th / 2.0f), 0.f);
            const float inter = ix * iy;
            const float iou = inter / (bw * bh + tw * th - inter);
            best = fmaxf(best, iou);
        }
        const float ignore = best < 0.5f ? 1.f : 0.f;
        const float valid = gm + (1.f - gm) * ignore;

        // objectness
        l_obj += valid * sig_ce(gm, t[4]);
        d[4] = valid * (sigmoidf_(t[4]) - gm) * p.gscale;
        // class
        for (int k = 0; k < p.K; ++k) {
            l_cls += gm * sig_ce(g[5 + k], t[5 + k]);
            d[5 + k] = gm * (sigmoidf_(t[5 + k]) - g[5 + k]) * p.gscale;
        }
        // xy: squared error in logit space of the clipped in-cell position
        {
            const float txr = g[0] / p.sx - offx, tyr = g[1] / p.sy - offy;
            const float pxr = bx / p.sx - offx, pyr = by / p.sy - offy;
            const float tx = fminf(fmaxf(txr, 0.01f), 0.99f), ty = fminf(fmaxf(tyr, 0.01f), 0.99f);
            const float px = fminf(fmaxf(pxr, 0.01f), 0.99f), py = fminf(fmaxf(pyr, 0.01f), 0.99f);
            const float ltx = -logf(1.0f / tx - 1.0f), lty = -logf(1.0f / ty - 1.0f);
            const float lpx = -logf(1.0f / px - 1.0f), lpy = -logf(1.0f / py - 1.0f);
            const float ex = ltx - lpx, ey = lty - lpy;
            l_xy += (ex * ex + ey * ey) * gm;
            // d/dt: -2*e * dlogit/dp * [clip passes] * sigmoid'(t)     (the *stride /stride pair is the identity)
            const float gx_ = (pxr >= 0.01f && pxr <= 0.99f) ? (sgx * (1.f - sgx)) / (px * (1.f - px)) : 0.f;
            const float gy_ = (pyr >= 0.01f && pyr <= 0.99f) ? (sgy * (1.f - sgy)) / (py * (1.f - py)) : 0.f;
            d[0] = gm * (-2.f * ex) * gx_ * p.gscale;
            d[1] = gm * (-2.f * ey) * gy_ * p.gscale;
        }
        // wh: squared error of log(size / anchor)
        {
            float tw = g[2] / aw, th = g[3] / ah;
            float pw = bw / aw, ph = bh / ah;
            const bool pw_nz = pw != 0.f, ph_nz = ph != 0.f;
            if (tw == 0.f) tw = 1.f;
            if (th == 0.f) th = 1.f;
            if (!pw_nz) pw = 1.f;
            if (!ph_nz) ph = 1.f;
            const float ltw = logf(fminf(fmaxf(tw, 1e-9f), 1e9f)), lth = logf(fminf(fmaxf(th, 1e-9f), 1e9f));
            const float lpw = logf(fminf(fmaxf(pw, 1e-9f), 1e9f)), lph = logf(fminf(fmaxf(ph, 1e-9f), 1e9f));
            const float ew_ = ltw - lpw, eh_ = lth - lph;
            l_wh += (ew_ * ew_ + eh_ * eh_) * gm;
            // d log(clip(q))/dt = [q in range, q != 0] * (dq/dt)/q = 1
            const float gw_ = (pw_nz && pw >= 1e-9f && pw <= 1e9f) ? 1.f : 0.f;
            const float gh_ = (ph_nz && ph >= 1e-9f && ph <= 1e9f) ? 1.f : 0.f;
            d[2] = gm * (-2.f * ew_) * gw_ * p.gscale;
            d[3] = gm * (-2.f * eh_) * gh_ * p.gscale;
        }
    }
    sm[0][threadIdx.x] = l_xy;
    sm[1][threadIdx.x] = l_wh;
    sm[2][threadIdx.x] = l_obj;
    sm[3][threadIdx.x] = l_cls;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o)
#pragma unroll
            for (int j = 0; j < 4; ++j) sm[j][threadIdx.x] += sm[j][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x < 4) p.partials[blockIdx.x * 4 + threadIdx.x] = sm[threadIdx.x][0] * p.inv_b;
}

__global__ void loss_clear_kernel(int* present) {
    if (threadIdx.x < Y3_MAX_ANCHORS) present[threadIdx.x] = 0;
}
__global__ void loss_finalize_kernel(const float* partials, int nblocks, float* loss4) {
    if (threadIdx.x < 4) {
        float s = 0.f;
        for (int b = 0; b < nblocks; ++b) s += partials[b * 4 + threadIdx.x];
        loss4[threadIdx.x] += s;
    }
}

#define Y3_LOSS_BLOCKS 64
extern "C" size_t y3_loss_workspace_bytes(void) { return (Y3_MAX_ANCHORS + Y3_LOSS_BLOCKS * 4) * sizeof(float); }

extern "C" int y3_loss_fwd_bwd(const y3_tensor* fm, const float* gt, const float* anchors_host, int num_anchors, int num_classes, int img_h,
                               int img_w, float global_batch, float* loss4, const y3_tensor* dfm, void* workspace, y3_stream_t stream) {
    Y3_CHECK_ARG(fm && fm->ptr && dfm && dfm->ptr && gt && anchors_host && loss4 && workspace, "loss_fwd_bwd: null pointer");
    Y3_CHECK_ARG(num_anchors >= 1 && num_anchors <= Y3_MAX_ANCHORS && num_classes >= 1, "loss_fwd_bwd: anchors/classes");
    const int D = num_anchors * (5 + num_classes);
    Y3_CHECK_ARG(fm->c == D && dfm->c == D && fm->ld >= D && dfm->ld >= D && dfm->n == fm->n && dfm->h == fm->h && dfm->w == fm->w, "loss_fwd_bwd: geometry");
    hipStream_t st = (hipStream_t)stream;
    LossArgs p = {};
    p.fm = fm->ptr;
    p.gt = gt;
    p.dfm = dfm->ptr;
    p.fm_ld = fm->ld;
    p.dfm_ld = dfm->ld;
    p.N = fm->n;
    p.G_h = fm->h;
    p.G_w = fm->w;
    p.A = num_anchors;
    p.K = num_classes;
    p.sx = (float)(img_h / fm->h);  // Q6
    p.sy = (float)(img_w / fm->w);
    for (int a = 0; a < num_anchors; ++a) {
        p.aw[a] = anchors_host[2 * a];
        p.ah[a] = anchors_host[2 * a + 1];
    }
    p.inv_b = 1.f / (float)fm->n;
    p.gscale = 1.f / ((float)fm->n * global_batch);
    p.present = (int*)workspace;
    p.partials = (float*)workspace + Y3_MAX_ANCHORS;
    // (a kernel, not hipMemsetAsync: as a memset NODE of a captured graph the clear was not reliably ordered against the loss kernels of
    // the previous scale, which read the same flags -- a replayed training step computed a wrong loss after the GPU had idled)
#ifdef Y3_DEV
    // development (tools/graph_dump.py): the round-3 form again, to look at the memset NODE it becomes in a captured step
    static const int use_memset = getenv("Y3_LOSS_MEMSET") ? atoi(getenv("Y3_LOSS_MEMSET")) : 0;
    if (use_memset) {
        if (hipMemsetAsync(p.present, 0, Y3_MAX_ANCHORS * sizeof(int), st) != hipSuccess) return Y3_ELAUNCH;
    } else
#endif
    hipLaunchKernelGGL(loss_clear_kernel, dim3(1), dim3(64), 0, st, p.present);
    Y3_CHECK_LAUNCH("loss_clear");
    const long long total = (long long)p.N * p.G_h * p.G_w * p.A;
    int pb = (int)((total + 255) / 256);
    if (pb > 256) pb = 256;
    hipLaunchKernelGGL(loss_present_kernel, dim3(pb), dim3(256), 0, st, gt, total, num_anchors, 5 + num_classes, p.present);
    Y3_CHECK_LAUNCH("loss_present");
    int blocks = (int)((total + 255) / 256);
    if (blocks > Y3_LOSS_BLOCKS) blocks = Y3_LOSS_BLOCKS;
    hipLaunchKernelGGL(loss_kernel, dim3(blocks), dim3(256), 0, st, p);
    Y3_CHECK_LAUNCH("loss");
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, st, (const float*)p.partials, blocks, loss4);
    Y3_CHECK_LAUNCH("loss_finalize");
    return Y3_OK;
}

// ---------------------------------------------------------------------------
// class-wise NMS: one 1024-thread workgroup per (image, class)
// ---------------------------------------------------------------------------
struct NmsArgs {
    const float* rows;
    int nb, D, K;
    float min_box, score_thr, iou_thr, clip_w, clip_h;
    int* keep_idx;
    int* keep_cnt;
    float* keep_score;
    int max_keep;
    int raw;  // 1: rows are [x0,y0,x1,y1,score] and the score is used as is (single_class_nms)
    int cap;  // capacity (power of two) of the per-block key array
    unsigned char* ws;
    size_t ws_per_block;
};

// IoU in the reference's operation order (bbox_utils.py:200-214), float32, no contraction
__device__ __forceinline__ bool nms_suppressed(float kx0, float ky0, float kx1, float ky1, float karea, float x0, float y0, float x1, float y1,
                                               float area, float thr) {
    const float xl = fmaxf(kx0, x0), yt = fmaxf(ky0, y0);
    const float xr = fminf(kx1, x1), yb = fminf(ky1, y1);
    const float inter = fmaxf(yb - yt, 0.f) * fmaxf(xr - xl, 0.f);
    const float uni = (karea + area) - inter;
    const float iou = inter / uni;
    return !(iou <= thr);  // survivors satisfy iou <= thr; NaN is dropped, as np.where(iou <= thr) drops it
}

// bbox_utils.filter_small_boxes (bbox_utils.py:274-281): indices of the rows with (x1 - x0) > min AND (y1 - y0) > min (strict),
// in row order.  One 1024-thread workgroup: per-chunk ballots + an exclusive scan over the 16 waves keep the order.
__global__ __launch_bounds__(1024) void filter_small_kernel(const float* __restrict__ rows, int m, int ld, float min_size, int* __restrict__ idx,
                                                            int* __restrict__ count) {
    __shared__ int wave_cnt[16];
    __shared__ int base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    for (int r0 = 0; r0 < m; r0 += 1024) {
        const int r = r0 + threadIdx.x;
        bool keep = false;
        if (r < m) {
            const float* b = rows + (size_t)r * ld;
            keep = (b[2] - b[0]) > min_size && (b[3] - b[1]) > min_size;
        }
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) wave_cnt[wave] = __popcll(bal);
        __syncthreads();
        int off = base;
        for (int wv = 0; wv < wave; ++wv) off += wave_cnt[wv];
        if (keep) idx[off + __popcll(bal & ((1ull << lane) - 1ull))] = r;
        __syncthreads();
        if (threadIdx.x == 0) {
            int t = 0;
            for (int wv = 0; wv < 16; ++wv) t += wave_cnt[wv];
            base += t;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *count = base;
}
extern "C" int y3_filter_small_boxes(const float* rows, int m, int ld, float min_size, int* keep_idx, int* keep_cnt, y3_stream_t stream) {
    Y3_CHECK_ARG(rows && keep_idx && keep_cnt && m > 0 && ld >= 4, "filter_small_boxes: bad args");
    hipLaunchKernelGGL(filter_small_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, rows, m, ld, min_size, keep_idx, keep_cnt);
    Y3_CHECK_LAUNCH("filter_small_boxes");
    return Y3_OK;
}

// bbox_utils.compute_iou (bbox_utils.py:200-214): IoU of one corner box against m boxes, same operation order as above
__global__ void compute_iou_kernel(const float* __restrict__ box, const float* __restrict__ boxes, int m, int ld, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const float kx0 = box[0], ky0 = box[1], kx1 = box[2], ky1 = box[3];
    const float* b = boxes + (size_t)i * ld;
    const float x0 = b[0], y0 = b[1], x1 = b[2], y1 = b[3];
    const float karea = (kx1 - kx0) * (ky1 - ky0), area = (x1 - x0) * (y1 - y0);
    const float xl = fmaxf(kx0, x0), yt = fmaxf(ky0, y0);
    const float xr = fminf(kx1, x1), yb = fminf(ky1, y1);
    const float inter = fmaxf(yb - yt, 0.f) * fmaxf(xr - xl, 0.f);
    out[i] = inter / ((karea + area) - inter);
}
extern "C" int y3_compute_iou(const float* box4, const float* boxes, int m, int ld, float* iou, y3_stream_t stream) {
    Y3_CHECK_ARG(box4 && boxes && iou && m > 0 && ld >= 4, "compute_iou: bad args");
    hipLaunchKernelGGL(compute_iou_kernel, dim3(y3_cdiv(m, 256)), dim3(256), 0, (hipStream_t)stream, box4, boxes, m, ld, iou);
    Y3_CHECK_LAUNCH("compute_iou");
    return Y3_OK;
}

template <bool LDS_KEYS>
__global__ __launch_bounds__(1024) void nms_kernel(const NmsArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int s_count, s_kept, s_nk;
    __shared__ float s_kb[5][64];

    const int tid = threadIdx.x;
    const int img = blockIdx.x / p.K, cls = blockIdx.x % p.K;
    const float* rows = p.rows + (long long)img * p.nb * p.D;
    unsigned char* wsb = p.ws + (size_t)blockIdx.x * p.ws_per_block;
    // workspace layout per block: boxes SoA [5][cap] floats | keys [cap] u64 | dead [cap] bytes  (keys/dead only when !LDS_KEYS)
    float* bx0 = (float*)wsb;
    float* by0 = bx0 + p.cap;
    float* bx1 = by0 + p.cap;
    float* by1 = bx1 + p.cap;
    float* bar = by1 + p.cap;
    unsigned long long* keys = LDS_KEYS ? (unsigned long long*)smem : (unsigned long long*)(wsb + (size_t)p.cap * 20);
    unsigned char* dead = LDS_KEYS ? (smem + (size_t)p.cap * 8) : (wsb + (size_t)p.cap * 28);

    if (tid == 0) {
        s_count = 0;
        s_kept = 0;
    }
    __syncthreads();

    // 1. candidates: small-box filter (strict >), score = sqrt(cls*obj) >= thr
    const bool clip = p.clip_w > 0.f;
    for (int i = tid; i < p.nb; i += 1024) {
        const float* r = rows + (long long)i * p.D;
        float x0 = r[0], y0 = r[1], x1 = r[2], y1 = r[3];
        if (clip) {
            x0 = fminf(fmaxf(x0, 0.f), p.clip_w);
            x1 = fminf(fmaxf(x1, 0.f), p.clip_w);
            y0 = fminf(fmaxf(y0, 0.f), p.clip_h);
            y1 = fminf(fmaxf(y1, 0.f), p.clip_h);
        }
        const float w = x1 - x0, h = y1 - y0;
        const float score = p.raw ? r[4] : sqrtf(r[5 + cls] * r[4]);
        if (w > p.min_box && h > p.min_box && score >= p.score_thr) {
            const int slot = atomicAdd(&s_count, 1);
            keys[slot] = ((unsigned long long)__float_as_uint(score) << 32) | (unsigned)i;
        }
    }
    __syncthreads();
    const int count = s_count;
    int n2 = 1;
    while (n2 < count) n2 <<= 1;
    for (int i = count + tid; i < n2; i += 1024) keys[i] = 0ull;
    for (int i = tid; i < n2; i += 1024) dead[i] = 0;
    __syncthreads();

    // 2. bitonic sort, descending on (score bits, row index): ties -> higher row index first
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (n2 >> 1); t += 1024) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));  // index with bit j clear
                const int hi = lo | j;
                const bool desc = (lo & k) == 0;
                const unsigned long long a = keys[lo], b = keys[hi];
                if ((a < b) == desc) {
                    keys[lo] = b;
                    keys[hi] = a;
                }
            }
            __syncthreads();
        }
    }

    // 3. gather the sorted boxes (SoA) and their areas
    for (int i = tid; i < count; i += 1024) {
        const float* r = rows + (long long)(unsigned)(keys[i] & 0xffffffffull) * p.D;
        float x0 = r[0], y0 = r[1], x1 = r[2], y1 = r[3];
        if (clip) {
            x0 = fminf(fmaxf(x0, 0.f), p.clip_w);
            x1 = fminf(fmaxf(x1, 0.f), p.clip_w);
            y0 = fminf(fmaxf(y0, 0.f), p.clip_h);
            y1 = fminf(fmaxf(y1, 0.f), p.clip_h);
        }
        bx0[i] = x0;
        by0[i] = y0;
        bx1[i] = x1;
        by1[i] = y1;
        bar[i] = (x1 - x0) * (y1 - y0);
    }
    __syncthreads();

    // 4. greedy suppression in sorted order, 64 candidates per round
    int* out_idx = p.keep_idx + (long long)blockIdx.x * p.max_keep;
    float* out_sc = p.keep_score + (long long)blockIdx.x * p.max_keep;
    for (int base = 0; base < count; base += 64) {
        if (tid < 64) {
            const int j = base + tid;
            const bool valid = j < count;
            bool alive = valid && !dead[j];
            float x0 = 0.f, y0 = 0.f, x1 = 0.f, y1 = 0.f, ar = 0.f;
            if (valid) {
                x0 = bx0[j];
                y0 = by0[j];
                x1 = bx1[j];
                y1 = by1[j];
                ar = bar[j];
            }
            unsigned long long mask = __ballot(alive);
            unsigned long long kept = 0ull;
            while (mask) {
                const int k = __ffsll((long long)mask) - 1;
                kept |= 1ull << k;
                const float kx0 = __shfl(x0, k), ky0 = __shfl(y0, k), kx1 = __shfl(x1, k), ky1 = __shfl(y1, k), kar = __shfl(ar, k);
                if (alive && tid > k && nms_suppressed(kx0, ky0, kx1, ky1, kar, x0, y0, x1, y1, ar, p.iou_thr)) alive = false;
                mask = __ballot(alive) & ~((2ull << k) - 1ull);
            }
            const int nk = __popcll(kept);
            if ((kept >> tid) & 1ull) {
                const int rank = __popcll(kept & ((1ull << tid) - 1ull));
                s_kb[0][rank] = x0;
                s_kb[1][rank] = y0;
                s_kb[2][rank] = x1;
                s_kb[3][rank] = y1;
                s_kb[4][rank] = ar;
                const int o = s_kept + rank;
                if (o < p.max_keep) {
                    out_idx[o] = (int)(unsigned)(keys[j] & 0xffffffffull);
                    out_sc[o] = __uint_as_float((unsigned)(keys[j] >> 32));
                }
            }
            if (tid == 0) s_nk = nk;
        }
        __syncthreads();
        const int nk = s_nk;
        if (tid == 0) s_kept += nk;
        if (nk > 0) {
            for (int j = base + 64 + tid; j < count; j += 1024) {
                if (dead[j]) continue;
                const float x0 = bx0[j], y0 = by0[j], x1 = bx1[j], y1 = by1[j], ar = bar[j];
                for (int q = 0; q < nk; ++q)
                    if (nms_suppressed(s_kb[0][q], s_kb[1][q], s_kb[2][q], s_kb[3][q], s_kb[4][q], x0, y0, x1, y1, ar, p.iou_thr)) {
                        dead[j] = 1;
                        break;
                    }
            }
        }
        __syncthreads();
    }
    if (tid == 0) p.keep_cnt[blockIdx.x] = s_kept < p.max_keep ? s_kept : p.max_keep;
}

static int nms_cap(int nb) {
    int c = 64;
    while (c < nb) c <<= 1;
    return c;
}
extern "C" size_t y3_nms_workspace_bytes(int n, int nb, int num_classes) {
    const size_t per = ((size_t)nms_cap(nb) * 29 + 255) & ~(size_t)255;
    return per * (size_t)n * (size_t)num_classes;
}

static int nms_launch(const float* rows, int n, int nb, int num_classes, int raw, float min_box, float score_thr, float iou_thr,
                      float clip_w, float clip_h, int* keep_idx, int* keep_cnt, float* keep_score, int max_keep, void* workspace,
                      size_t workspace_bytes, y3_stream_t stream) {
    Y3_CHECK_ARG(rows && keep_idx && keep_cnt && keep_score && workspace, "nms: null pointer");
    Y3_CHECK_ARG(n > 0 && nb > 0 && num_classes > 0 && max_keep > 0, "nms: bad sizes");
    Y3_CHECK_ARG(workspace_bytes >= y3_nms_workspace_bytes(n, nb, num_classes), "nms: workspace too small");
    NmsArgs p = {};
    p.rows = rows;
    p.nb = nb;
    p.D = raw ? 5 : 5 + num_classes;
    p.K = num_classes;
    p.raw = raw;
    p.min_box = min_box;
    p.score_thr = score_thr;
    p.iou_thr = iou_thr;
    p.clip_w = clip_w;
    p.clip_h = clip_h;
    p.keep_idx = keep_idx;
    p.keep_cnt = keep_cnt;
    p.keep_score = keep_score;
    p.max_keep = max_keep;
    p.cap = nms_cap(nb);
    p.ws = (unsigned char*)workspace;
    p.ws_per_block = ((size_t)p.cap * 29 + 255) & ~(size_t)255;
    hipStream_t st = (hipStream_t)stream;
    const int blocks = n * num_classes;
    if (p.cap <= 16384) {
        const size_t lds = (size_t)p.cap * 9;  // keys + dead flags, <= 144 KiB of the CU's 160 KiB
        static bool attr_set = false;
        if (!attr_set) {
            if (hipFuncSetAttribute((const void*)nms_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 9) != hipSuccess) {
                y3_set_error("nms_per_class: cannot raise dynamic LDS limit");
                return Y3_ELAUNCH;
            }
            attr_set = true;
        }
        hipLaunchKernelGGL((nms_kernel<true>), dim3(blocks), dim3(1024), lds, st, p);
    } else {
        hipLaunchKernelGGL((nms_kernel<false>), dim3(blocks), dim3(1024), 0, st, p);
    }
    Y3_CHECK_LAUNCH("nms");
    return Y3_OK;
}

extern "C" int y3_nms_per_class(const float* rows, int n, int nb, int num_classes, float min_box, float score_thr, float iou_thr,
                                float clip_w, float clip_h, int* keep_idx, int* keep_cnt, float* keep_score, int max_keep, void* workspace,
                                size_t workspace_bytes, y3_stream_t stream) {
    return nms_launch(rows, n, nb, num_classes, 0, min_box, score_thr, iou_thr, clip_w, clip_h, keep_idx, keep_cnt, keep_score, max_keep,
                      workspace, workspace_bytes, stream);
}

extern "C" int y3_nms_single_class(const float* rows5, int m, float iou_thr, int* keep_idx, int* keep_cnt, float* keep_score, void* workspace,
                                   size_t workspace_bytes, y3_stream_t stream) {
    return nms_launch(rows5, 1, m, 1, 1, -INFINITY, -INFINITY, iou_thr, -1.f, -1.f, keep_idx, keep_cnt, keep_score, m, workspace,
                      workspace_bytes, stream);
}
