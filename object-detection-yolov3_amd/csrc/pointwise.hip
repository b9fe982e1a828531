// HBM-bound kernels around the convolutions: BatchNorm (training statistics,
// apply, backward), the all-ones upsample, concat copies, gradient fan-in,
// layout changes, Adam, z-score.  All are streaming kernels: 16-byte accesses
// along the contiguous channel axis, grid-stride loops capped at ~8 blocks/CU.
#include "common.h"

static inline int stream_blocks(long long items, int threads) {
    long long b = (items + threads - 1) / threads;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

static int check_view(const y3_tensor* t, const char* name) {
    Y3_CHECK_ARG(t && t->ptr, "%s: null tensor", name);
    Y3_CHECK_ARG(t->n > 0 && t->h > 0 && t->w > 0 && t->c > 0 && t->ld >= t->c, "%s: bad dims", name);
    return 0;
}
static int check_view4(const y3_tensor* t, const char* name) {
    if (int e = check_view(t, name)) return e;
    Y3_CHECK_ARG((t->c & 3) == 0 && (t->ld & 3) == 0 && ((uintptr_t)t->ptr & 15) == 0, "%s: needs c, ld multiples of 4 and 16-byte alignment", name);
    return 0;
}
static inline long long pixels(const y3_tensor* t) { return (long long)t->n * t->h * t->w; }
static inline bool same_geom(const y3_tensor* a, const y3_tensor* b) { return a->n == b->n && a->h == b->h && a->w == b->w && a->c == b->c; }

// ---------------------------------------------------------------------------
// BatchNorm training statistics
// ---------------------------------------------------------------------------
// A block owns 4 channels and spreads the partial rows over 512 lanes (x 2 statistics = 1024 threads, one float4 each):
// C/4 blocks instead of C/32, so the 32- and 64-channel layers at full resolution (10 816 / 2 704 partial rows) are no
// longer reduced by one or two workgroups.  fp64 accumulation, fixed tree -> deterministic.
__global__ __launch_bounds__(1024) void bn_stats_finalize_kernel(const float* __restrict__ stats, int tiles, int C, double inv_count,
                                                                 double bessel, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float eps, float momentum,
                                                                 float* moving_mean, float* moving_var, float* save_mean,
                                                                 float* save_rstd, float* scale, float* shift) {
    __shared__ double sm[1024][4];
    const int which = threadIdx.x & 1, lane = threadIdx.x >> 1;
    const int c0 = blockIdx.x * 4;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int t = lane; t < tiles; t += 512) {
        const float4 v = *reinterpret_cast<const float4*>(stats + ((long long)t * 2 + which) * C + c0);
        acc[0] += (double)v.x;
        acc[1] += (double)v.y;
        acc[2] += (double)v.z;
        acc[3] += (double)v.w;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) sm[threadIdx.x][e] = acc[e];
    __syncthreads();
    for (int half = 512; half >= 2; half >>= 1) {      // threads t and t + half hold the same statistic (half is even)
        if (threadIdx.x < half)
#pragma unroll
            for (int e = 0; e < 4; ++e) sm[threadIdx.x][e] += sm[threadIdx.x + half][e];
        __syncthreads();
    }
    if (threadIdx.x < 4) {
        const int c = c0 + threadIdx.x;
        const double mean = sm[0][threadIdx.x] * inv_count;
        double var = sm[1][threadIdx.x] * inv_count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        const float fmean = (float)mean;
        const float sc = gamma[c] * rstd;
        scale[c] = sc;
        shift[c] = beta[c] - fmean * sc;
        save_mean[c] = fmean;
        save_rstd[c] = rstd;
        if (moving_mean) {
            moving_mean[c] = moving_mean[c] * momentum + fmean * (1.f - momentum);
            moving_var[c] = moving_var[c] * momentum + (float)(var * bessel) * (1.f - momentum);
        }
    }
}

extern "C" int y3_bn_stats_finalize(const float* stats, int tiles, int c, int count, const float* gamma, const float* beta, float eps,
                                    float momentum, float* moving_mean, float* moving_var, float* save_mean, float* save_rstd,
                                    float* scale, float* shift, y3_stream_t stream) {
    Y3_CHECK_ARG(stats && gamma && beta && save_mean && save_rstd && scale && shift, "bn_stats_finalize: null pointer");
    Y3_CHECK_ARG(tiles > 0 && c > 0 && count > 0, "bn_stats_finalize: bad sizes");
    Y3_CHECK_ARG((moving_mean == nullptr) == (moving_var == nullptr), "bn_stats_finalize: moving stats must both be given");
    const double bessel = count > 1 ? (double)count / (double)(count - 1) : 1.0;
    Y3_CHECK_ARG((c & 3) == 0 && ((uintptr_t)stats & 15) == 0, "bn_stats_finalize: channels must be a multiple of 4, stats 16-byte aligned");
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(c / 4), dim3(1024), 0, (hipStream_t)stream, stats, tiles, c, 1.0 / (double)count,
                       bessel, gamma, beta, eps, momentum, moving_mean, moving_var, save_mean, save_rstd, scale, shift);
    Y3_CHECK_LAUNCH("bn_stats_finalize");
    return Y3_OK;
}

__global__ void bn_fold_kernel(const float* gamma, const float* beta, const float* mean, const float* var, float eps, int C, float* scale,
                               float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        const float sc = gamma[c] * (1.f / sqrtf(var[c] + eps));
        scale[c] = sc;
        shift[c] = beta[c] - mean[c] * sc;
    }
}
extern "C" int y3_bn_fold_inference(const float* gamma, const float* beta, const float* moving_mean, const float* moving_var, float eps,
                                    int c, float* scale, float* shift, y3_stream_t stream) {
    Y3_CHECK_ARG(gamma && beta && moving_mean && moving_var && scale && shift && c > 0, "bn_fold_inference: bad args");
    hipLaunchKernelGGL(bn_fold_kernel, dim3(y3_cdiv(c, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta, moving_mean, moving_var, eps, c,
                       scale, shift);
    Y3_CHECK_LAUNCH("bn_fold_inference");
    return Y3_OK;
}

// every BatchNorm layer of the network in one launch: table rows {gamma, beta, moving mean, moving var, scale, shift (float offsets), C}
__global__ void bn_fold_batched_kernel(const float* __restrict__ params, const float* __restrict__ moving, float* __restrict__ chan,
                                       const int* __restrict__ table, float eps) {
    const int* row = table + blockIdx.x * 7;
    const float* gamma = params + row[0];
    const float* beta = params + row[1];
    const float* mean = moving + row[2];
    const float* var = moving + row[3];
    float* scale = chan + row[4];
    float* shift = chan + row[5];
    const int C = row[6];
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float sc = gamma[c] * (1.f / sqrtf(var[c] + eps));
        scale[c] = sc;
        shift[c] = beta[c] - mean[c] * sc;
    }
}
extern "C" int y3_bn_fold_inference_batched(const float* params, const float* moving, float* chan, const int* table_dev, int nlayers, float eps,
                                            y3_stream_t stream) {
    Y3_CHECK_ARG(params && moving && chan && table_dev && nlayers > 0, "bn_fold_inference_batched: bad args");
    hipLaunchKernelGGL(bn_fold_batched_kernel, dim3(nlayers), dim3(256), 0, (hipStream_t)stream, params, moving, chan, table_dev, eps);
    Y3_CHECK_LAUNCH("bn_fold_inference_batched");
    return Y3_OK;
}

// y = a*scale + shift (+ resid)
__global__ void bn_apply_kernel(const float* __restrict__ a, int a_ld, const float* __restrict__ scale, const float* __restrict__ shift,
                                const float* __restrict__ resid, int r_ld, float* __restrict__ y, int y_ld, long long npix, int c4) {
    const long long total = npix * c4;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const long long pix = i / c4;
        const int c = (int)(i - pix * c4) * 4;
        const float4 v = *reinterpret_cast<const float4*>(a + pix * a_ld + c);
        const float4 sc = *reinterpret_cast<const float4*>(scale + c);
        const float4 sf = *reinterpret_cast<const float4*>(shift + c);
        float4 o = make_float4(v.x * sc.x + sf.x, v.y * sc.y + sf.y, v.z * sc.z + sf.z, v.w * sc.w + sf.w);
        if (resid) {
            const float4 r = *reinterpret_cast<const float4*>(resid + pix * r_ld + c);
            o.x += r.x;
            o.y += r.y;
            o.z += r.z;
            o.w += r.w;
        }
        *reinterpret_cast<float4*>(y + pix * y_ld + c) = o;
    }
}
extern "C" int y3_bn_apply(const y3_tensor* a, const float* scale, const float* shift, const y3_tensor* resid, const y3_tensor* y,
                           y3_stream_t stream) {
    if (int e = check_view4(a, "bn_apply a")) return e;
    if (int e = check_view4(y, "bn_apply y")) return e;
    Y3_CHECK_ARG(same_geom(a, y) && scale && shift, "bn_apply: geometry/pointers");
    if (resid) {
        if (int e = check_view4(resid, "bn_apply resid")) return e;
        Y3_CHECK_ARG(same_geom(a, resid), "bn_apply: resid geometry");
    }
    const long long total = pixels(a) * (a->c / 4);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(stream_blocks(total, 256)), dim3(256), 0, (hipStream_t)stream, a->ptr, a->ld, scale, shift,
                       resid ? resid->ptr : nullptr, resid ? resid->ld : 0, y->ptr, y->ld, pixels(a), a->c / 4);
    Y3_CHECK_LAUNCH("bn_apply");
    return Y3_OK;
}

// ---------------------------------------------------------------------------
// BatchNorm + leaky-relu backward
// ---------------------------------------------------------------------------
#define Y3_BNB_SUMS 6
#define Y3_BNB_BLOCKS 256             // workgroups per launch (one per CU): 512 stream no faster and double the last arrival's work (measured)
#define Y3_BNB_TICKETS 1024           // bytes of tickets (one int per channel slice, <= 16 slices) at the head of the workspace

// A workgroup streams a band of rows of ONE channel slice: lc float4 lanes (<= 8: 32 channels) x 256/lc row groups.  Slicing
// the channels keeps the fp64 partials at parts * slices * 6 * 32 doubles <= 384 KiB whatever C is (whole-row workgroups
// wrote up to 25 MiB of partials for the 1024-channel layers, more than the tensors they reduce).
struct BnbPlan {
    int lc, sw, slices, parts;
};
static bool plan_bnb(long long m, int c, BnbPlan* p) {
    if (c < 4 || c > 1024 || (c & 3)) return false;
#ifdef Y3_DEV      // development switches (make DEV=1): the product library reads neither
    static const int lcap = getenv("Y3_BNB_LC") ? atoi(getenv("Y3_BNB_LC")) : 8;
    static const int blocks = getenv("Y3_BNB_BLOCKS") ? atoi(getenv("Y3_BNB_BLOCKS")) : Y3_BNB_BLOCKS;
#else
    const int lcap = 8;                  // 8 float4 lanes = one 128-byte line per row
    const int blocks = Y3_BNB_BLOCKS;
#endif
    p->lc = c / 4 < lcap ? c / 4 : lcap;
    if (p->lc & (p->lc - 1)) return false;
    p->sw = 4 * p->lc;
    if (c % p->sw) return false;
    p->slices = c / p->sw;
    long long parts = y3_cdiv(m, (long long)(256 / p->lc) * 4);
    int cap = (blocks < Y3_BNB_BLOCKS ? blocks : Y3_BNB_BLOCKS) / p->slices;
    if (cap < 1) cap = 1;
    if (parts > cap) parts = cap;
    if (parts < 1) parts = 1;
    p->parts = (int)parts;
    return true;
}
extern "C" size_t y3_bn_bwd_workspace(int m, int c) {
    BnbPlan p;
    if (!plan_bnb(m, c, &p)) return 0;
    return (size_t)Y3_BNB_TICKETS + (size_t)p.parts * p.slices * Y3_BNB_SUMS * p.sw * sizeof(double);
}

struct BnbArgs {
    const float* dy;
    const float* a;
    float* dres;          // optional: gradient of the residual input, dres (+)= dy while dy streams through
    double* partials;     // [slice][part][6][sw]
    int* tickets;
    const float* gamma;
    const float* mean;
    const float* rstd;
    float* dgamma;
    float* dbeta;
    float* dbias;
    float* coef;
    long long npix;
    double count;
    int dy_ld, a_ld, dres_ld, dres_acc;
    int C, lc, slices, parts;
    float alpha;
};

// Raw moments of (dy, a), fp64:
//   S0 = sum dy, S1 = sum dy*a, S2 = sum_{a>0} dy, S3 = sum_{a>0} a, S4 = #{a>0}, S5 = sum a
// The leaky-relu slope s is 1 for a > 0 and alpha otherwise, so with xhat = (a - mu) * r
//   sum dy*xhat = r*(S1 - mu*S0),  sum dy*s = alpha*S0 + (1-alpha)*S2,  sum s = alpha*M + (1-alpha)*S4,
//   sum xhat*s  = r*((1-alpha)*(S3 - mu*S4) + alpha*(S5 - mu*M))
// Raw moments keep the per-element work at two conversions, one fma and a few (masked) adds.  fp64 accumulation: dbias is a
// small difference of these sums (BatchNorm removes the mean shift a bias introduces), fp32 running sums lose it to
// cancellation.
// The workgroup whose partial arrives LAST for a channel slice (ticket, as in the split-K convolutions: sc1 stores, vmcnt
// drained, one agent-scope atomic add) sums the slice's partials in part order -- the result does not depend on which
// workgroup that was -- and writes dgamma / dbeta / dbias and the coefficients of bn_bwd_apply.  No second launch.
template <bool RES>
__global__ __launch_bounds__(256) void bn_bwd_stats_kernel(const BnbArgs p) {
    __shared__ double sm[256 * 4 * Y3_BNB_SUMS];
    __shared__ int last_flag;
    const int tid = threadIdx.x;
    const int lc = p.lc, rgn = 256 / lc, sw = 4 * lc;
    const int slice = blockIdx.x % p.slices, part = blockIdx.x / p.slices;
    const int cq = tid % lc, rg = tid / lc;
    const int c = slice * sw + cq * 4;
    double acc[5][4];
    int cnt[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[j][e] = 0.0;
    const long long rows_per_block = (p.npix + p.parts - 1) / p.parts;
    const long long r0 = (long long)part * rows_per_block;
    long long r1 = r0 + rows_per_block;
    if (r1 > p.npix) r1 = p.npix;
    auto accumulate = [&](const float4 d4, const float4 a4) {
        const float dv[4] = {d4.x, d4.y, d4.z, d4.w}, av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const double d = (double)dv[e], x = (double)av[e];
            const bool pos = av[e] > 0.f;
            acc[0][e] += d;
            acc[1][e] += d * x;
            acc[2][e] += pos ? d : 0.0;
            acc[3][e] += pos ? x : 0.0;
            acc[4][e] += x;
            cnt[e] += pos ? 1 : 0;
        }
    };
    auto fan_in = [&](long long row, const float4 d4) {
        float4* q = reinterpret_cast<float4*>(p.dres + row * p.dres_ld + c);
        float4 o = d4;
        if (p.dres_acc) {
            const float4 t = *q;
            o = make_float4(t.x + d4.x, t.y + d4.y, t.z + d4.z, t.w + d4.w);
        }
        *q = o;
    };
    // four rows per trip, and the next trip's eight 16-byte loads are issued before this trip's rows are accumulated (one
    // wave per SIMD cannot hide a memory round trip per trip otherwise).  Rows past the band read as zeros, which leave every
    // sum unchanged -- no serial remainder loop.  Rows are accumulated in their original order.
    const long long trip = 4LL * rgn;
    long long r = r0 + rg;
    float4 d4[2][4], a4[2][4];
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load_trip = [&](int b, long long row) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long ru = row + (long long)u * rgn;
            const bool in = ru < r1;
            d4[b][u] = in ? *reinterpret_cast<const float4*>(p.dy + ru * p.dy_ld + c) : zero4;
            a4[b][u] = in ? *reinterpret_cast<const float4*>(p.a + ru * p.a_ld + c) : zero4;
        }
    };
    auto use_trip = [&](int b, long long row) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (RES && row + (long long)u * rgn < r1) fan_in(row + (long long)u * rgn, d4[b][u]);
            accumulate(d4[b][u], a4[b][u]);
        }
    };
    const long long ntrips = r < r1 ? (r1 - r + trip - 1) / trip : 0;
    if (ntrips > 0) {
        load_trip(0, r);
        long long t = 0;
        for (; t + 2 < ntrips; t += 2) {
            load_trip(1, r + trip);
            use_trip(0, r);
            load_trip(0, r + 2 * trip);
            use_trip(1, r + trip);
            r += 2 * trip;
        }
        if (t + 1 < ntrips) {
            load_trip(1, r + trip);
            use_trip(0, r);
            use_trip(1, r + trip);
        } else {
            use_trip(0, r);
        }
    }
    // order in memory: S0, S1, S2, S3, S4 (count), S5 (sum a)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        double* q = sm + tid * (4 * Y3_BNB_SUMS) + e;
        q[0] = acc[0][e];
        q[4] = acc[1][e];
        q[8] = acc[2][e];
        q[12] = acc[3][e];
        q[16] = (double)cnt[e];
        q[20] = acc[4][e];
    }
    __syncthreads();
    // sum the row groups (fixed order) with every thread busy: output o = (sum j, channel ch of the slice)
    double* mine = p.partials + ((long long)slice * p.parts + part) * (Y3_BNB_SUMS * sw);
    for (int o = tid; o < Y3_BNB_SUMS * sw; o += 256) {
        const int j = o / sw, ch = o - j * sw;
        const double* q = sm + (ch >> 2) * (4 * Y3_BNB_SUMS) + j * 4 + (ch & 3);
        double s = q[0];
        for (int g = 1; g < rgn; ++g) s += q[(long long)g * lc * (4 * Y3_BNB_SUMS)];
        __hip_atomic_store(mine + o, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // sc1: written through to where every XCD sees it
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const int old = __hip_atomic_fetch_add(p.tickets + slice, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = old == p.parts - 1;
        if (last) __hip_atomic_store(p.tickets + slice, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_flag = last;
    }
    __syncthreads();
    if (last_flag == 0) return;

    // ---- the slice's last arrival: partials -> sums.  Thread = (channel pair, part lane); 16-byte sc1 loads, 12 in flight
    // (256 workgroups of 32 channels: two parts per thread, one round trip).
    const int half = sw >> 1, chp = tid % half, pl = tid / half, pln = 256 / half;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.partials + (long long)slice * p.parts * (Y3_BNB_SUMS * sw), 0,
                                                                         (unsigned)(p.parts * Y3_BNB_SUMS * sw * 8), 0x00020000);
    double t[Y3_BNB_SUMS][2];
#pragma unroll
    for (int j = 0; j < Y3_BNB_SUMS; ++j) t[j][0] = t[j][1] = 0.0;
    const unsigned part_bytes = (unsigned)(Y3_BNB_SUMS * sw * 8);
    int q = pl;
    for (; q + pln < p.parts; q += 2 * pln) {
        f32x4 v[2][Y3_BNB_SUMS];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int j = 0; j < Y3_BNB_SUMS; ++j)
                v[u][j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(q + u * pln) * part_bytes + (unsigned)(j * sw * 8 + chp * 16), 0, 16 /* sc1 */);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int j = 0; j < Y3_BNB_SUMS; ++j) {
                const double2 d = __builtin_bit_cast(double2, v[u][j]);
                t[j][0] += d.x;
                t[j][1] += d.y;
            }
    }
    if (q < p.parts) {
#pragma unroll
        for (int j = 0; j < Y3_BNB_SUMS; ++j) {
            const f32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)q * part_bytes + (unsigned)(j * sw * 8 + chp * 16), 0, 16 /* sc1 */);
            const double2 d = __builtin_bit_cast(double2, v);
            t[j][0] += d.x;
            t[j][1] += d.y;
        }
    }
    __syncthreads();      // the row-group sums in `sm` have been read by everybody
#pragma unroll
    for (int j = 0; j < Y3_BNB_SUMS; ++j) {
        sm[(pl * Y3_BNB_SUMS + j) * sw + 2 * chp] = t[j][0];
        sm[(pl * Y3_BNB_SUMS + j) * sw + 2 * chp + 1] = t[j][1];
    }
    __syncthreads();
    if (tid < sw) {
        double s[Y3_BNB_SUMS];
#pragma unroll
        for (int j = 0; j < Y3_BNB_SUMS; ++j) {
            s[j] = sm[j * sw + tid];
            for (int g = 1; g < pln; ++g) s[j] += sm[(g * Y3_BNB_SUMS + j) * sw + tid];
        }
        const int cc = slice * sw + tid, C = p.C;
        const double ga = p.gamma[cc], rr = p.rstd[cc], mu = p.mean[cc], al = (double)p.alpha, count = p.count, inv_count = 1.0 / count;
        const double db = s[0];                                                       // sum dy
        const double dg = rr * (s[1] - mu * s[0]);                                    // sum dy * xhat
        const double sdys = al * s[0] + (1.0 - al) * s[2];                            // sum dy * slope
        const double ss = al * count + (1.0 - al) * s[4];                             // sum slope
        const double sxs = rr * ((1.0 - al) * (s[3] - mu * s[4]) + al * (s[5] - mu * count));   // sum xhat * slope
        // da = ga*r*(dy - db/M - xhat*dg/M);  dz = da*slope;  dbias = sum dz
        p.dgamma[cc] = (float)dg;
        p.dbeta[cc] = (float)db;
        p.dbias[cc] = (float)(ga * rr * (sdys - db * inv_count * ss - dg * inv_count * sxs));
        const double k1 = ga * rr;
        const double k2 = -ga * rr * rr * dg * inv_count;
        const double k3 = -ga * rr * db * inv_count - k2 * mu;
        p.coef[cc] = (float)k1;
        p.coef[C + cc] = (float)k2;
        p.coef[2 * C + cc] = (float)k3;
    }
}

extern "C" int y3_bn_bwd_stats(const y3_tensor* dy, const y3_tensor* a, const y3_tensor* dres, int dres_accumulate, const float* gamma,
                               const float* save_mean, const float* save_rstd, float alpha, float* dgamma, float* dbeta, float* dbias,
                               float* coef, void* workspace, size_t workspace_bytes, y3_stream_t stream) {
    if (int e = check_view4(dy, "bn_bwd_stats dy")) return e;
    if (int e = check_view4(a, "bn_bwd_stats a")) return e;
    Y3_CHECK_ARG(same_geom(dy, a), "bn_bwd_stats: dy / a geometry");
    Y3_CHECK_ARG(gamma && save_mean && save_rstd && dgamma && dbeta && dbias && coef && workspace, "bn_bwd_stats: null pointer");
    if (dres) {
        if (int e = check_view4(dres, "bn_bwd_stats dres")) return e;
        Y3_CHECK_ARG(same_geom(dy, dres), "bn_bwd_stats: dres geometry");
    }
    BnbPlan pl;
    Y3_CHECK_ARG(plan_bnb(pixels(a), a->c, &pl), "bn_bwd_stats: channels %d unsupported (a multiple of 64 up to 1024, or 4 / 8 / 16 / 32)", a->c);
    const size_t need = (size_t)Y3_BNB_TICKETS + (size_t)pl.parts * pl.slices * Y3_BNB_SUMS * pl.sw * sizeof(double);
    Y3_CHECK_ARG(workspace_bytes >= need && ((uintptr_t)workspace & 15) == 0, "bn_bwd_stats: workspace %zu bytes, %zu needed (16-byte aligned)",
                 workspace_bytes, need);
    BnbArgs p;
    p.dy = dy->ptr; p.a = a->ptr; p.dres = dres ? dres->ptr : nullptr;
    p.tickets = reinterpret_cast<int*>(workspace);
    p.partials = reinterpret_cast<double*>(reinterpret_cast<char*>(workspace) + Y3_BNB_TICKETS);
    p.gamma = gamma; p.mean = save_mean; p.rstd = save_rstd;
    p.dgamma = dgamma; p.dbeta = dbeta; p.dbias = dbias; p.coef = coef;
    p.npix = pixels(a); p.count = (double)pixels(a);
    p.dy_ld = dy->ld; p.a_ld = a->ld; p.dres_ld = dres ? dres->ld : 0; p.dres_acc = dres_accumulate;
    p.C = a->c; p.lc = pl.lc; p.slices = pl.slices; p.parts = pl.parts;
    p.alpha = alpha;
    if (dres)
        hipLaunchKernelGGL(bn_bwd_stats_kernel<true>, dim3(pl.parts * pl.slices), dim3(256), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(bn_bwd_stats_kernel<false>, dim3(pl.parts * pl.slices), dim3(256), 0, (hipStream_t)stream, p);
    Y3_CHECK_LAUNCH("bn_bwd_stats");
    return Y3_OK;
}

// The same coefficients from the [row tile][6][C] fp32 partial moments that y3_conv2d_dgrad_bn leaves behind (conv.hip, BNS
// epilogue: sums over the <= 128 rows of a tile in fp32, tiles added here in fp64).  A block owns 4 channels: 64 tile lanes x
// 6 sums = 384 threads, one float4 per tile row; C/4 blocks.  LDS budget: this launch-bound kernel sits on the critical path
// while the previous layer's kernel gradient fills the CUs from the second stream (3 workgroups x 48 KB of the 160 KB): with
// 24 KB of LDS (128 lanes) its blocks waited for a kernel-gradient workgroup to retire -- 31.6 us per launch in the overlapped
// step against 8.7 us alone; 12 KB fit beside them.
#ifndef Y3_BNF_LANES
#define Y3_BNF_LANES 64
#endif
__global__ __launch_bounds__(Y3_BNF_LANES * Y3_BNB_SUMS) void bn_bwd_finalize_tiles_kernel(const float* __restrict__ partials, int tiles, int C, double count, float alpha,
                                                                    const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                    const float* __restrict__ rstd, float* dgamma, float* dbeta, float* dbias,
                                                                    float* coef) {
    __shared__ double sm[Y3_BNB_SUMS][Y3_BNF_LANES][4];
    const int j = threadIdx.x % Y3_BNB_SUMS, lane = threadIdx.x / Y3_BNB_SUMS;
    const int c0 = blockIdx.x * 4;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int t = lane; t < tiles; t += Y3_BNF_LANES) {
        const float4 v = *reinterpret_cast<const float4*>(partials + ((long long)t * Y3_BNB_SUMS + j) * C + c0);
        acc[0] += (double)v.x;
        acc[1] += (double)v.y;
        acc[2] += (double)v.z;
        acc[3] += (double)v.w;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) sm[j][lane][e] = acc[e];
    __syncthreads();
    for (int half = Y3_BNF_LANES / 2; half >= 1; half >>= 1) {
        if (lane < half)
#pragma unroll
            for (int e = 0; e < 4; ++e) sm[j][lane][e] += sm[j][lane + half][e];
        __syncthreads();
    }
    if (threadIdx.x < 4) {
        const int e = threadIdx.x, c = c0 + e;
        const double s0 = sm[0][0][e], s1 = sm[1][0][e], s2 = sm[2][0][e], s3 = sm[3][0][e], s4 = sm[4][0][e], s5 = sm[5][0][e];
        const double ga = gamma[c], r = rstd[c], mu = mean[c], al = (double)alpha, inv_count = 1.0 / count;
        const double db = s0;                                                    // sum dy
        const double dg = r * (s1 - mu * s0);                                    // sum dy * xhat
        const double sdys = al * s0 + (1.0 - al) * s2;                           // sum dy * slope
        const double ss = al * count + (1.0 - al) * s4;                          // sum slope
        const double sxs = r * ((1.0 - al) * (s3 - mu * s4) + al * (s5 - mu * count));   // sum xhat * slope
        dgamma[c] = (float)dg;
        dbeta[c] = (float)db;
        dbias[c] = (float)(ga * r * (sdys - db * inv_count * ss - dg * inv_count * sxs));
        const double k1 = ga * r;
        const double k2 = -ga * r * r * dg * inv_count;
        const double k3 = -ga * r * db * inv_count - k2 * mu;
        coef[c] = (float)k1;
        coef[C + c] = (float)k2;
        coef[2 * C + c] = (float)k3;
    }
}
extern "C" int y3_bn_bwd_finalize_tiles(const float* partials, int tiles, int c, int count, const float* gamma, const float* save_mean,
                                        const float* save_rstd, float alpha, float* dgamma, float* dbeta, float* dbias, float* coef,
                                        y3_stream_t stream) {
    Y3_CHECK_ARG(partials && gamma && save_mean && save_rstd && dgamma && dbeta && dbias && coef, "bn_bwd_finalize_tiles: null pointer");
    Y3_CHECK_ARG(tiles > 0 && c > 0 && count > 0, "bn_bwd_finalize_tiles: bad sizes");
    Y3_CHECK_ARG((c & 3) == 0 && ((uintptr_t)partials & 15) == 0, "bn_bwd_finalize_tiles: channels must be a multiple of 4, partials 16-byte aligned");
    hipLaunchKernelGGL(bn_bwd_finalize_tiles_kernel, dim3(c / 4), dim3(Y3_BNF_LANES * Y3_BNB_SUMS), 0, (hipStream_t)stream, partials, tiles, c, (double)count, alpha, gamma,
                       save_mean, save_rstd, dgamma, dbeta, dbias, coef);
    Y3_CHECK_LAUNCH("bn_bwd_finalize_tiles");
    return Y3_OK;
}

// dz = (k1*dy + k2*a + k3) * slope(a); RES: the residual fan-in dres (+)= dy rides along (dy is in registers)
template <bool RES>
__global__ void bn_bwd_apply_kernel(const float* __restrict__ dy, int dy_ld, const float* __restrict__ a, int a_ld,
                                    const float* __restrict__ coef, float alpha, float* __restrict__ dz, int dz_ld, float* __restrict__ dres,
                                    int dres_ld, int dres_acc, long long npix, int C) {
    const int c4n = C >> 2;
    const long long total = npix * c4n;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const long long pix = i / c4n;
        const int c = (int)(i - pix * c4n) * 4;
        const float4 d4 = *reinterpret_cast<const float4*>(dy + pix * dy_ld + c);
        const float4 a4 = *reinterpret_cast<const float4*>(a + pix * a_ld + c);
        const float4 k1 = *reinterpret_cast<const float4*>(coef + c);
        const float4 k2 = *reinterpret_cast<const float4*>(coef + C + c);
        const float4 k3 = *reinterpret_cast<const float4*>(coef + 2 * C + c);
        if (RES) {
            float4* q = reinterpret_cast<float4*>(dres + pix * dres_ld + c);
            float4 o = d4;
            if (dres_acc) {
                const float4 t = *q;
                o = make_float4(t.x + d4.x, t.y + d4.y, t.z + d4.z, t.w + d4.w);
            }
            *q = o;
        }
        float4 o;
        o.x = (k1.x * d4.x + k2.x * a4.x + k3.x) * (a4.x > 0.f ? 1.f : alpha);
        o.y = (k1.y * d4.y + k2.y * a4.y + k3.y) * (a4.y > 0.f ? 1.f : alpha);
        o.z = (k1.z * d4.z + k2.z * a4.z + k3.z) * (a4.z > 0.f ? 1.f : alpha);
        o.w = (k1.w * d4.w + k2.w * a4.w + k3.w) * (a4.w > 0.f ? 1.f : alpha);
        *reinterpret_cast<float4*>(dz + pix * dz_ld + c) = o;
    }
}
static int bn_bwd_apply_launch(const char* what, const y3_tensor* dy, const y3_tensor* a, const float* coef, float alpha, const y3_tensor* dz,
                               const y3_tensor* dres, int dres_accumulate, y3_stream_t stream) {
    if (int e = check_view4(dy, what)) return e;
    if (int e = check_view4(a, what)) return e;
    if (int e = check_view4(dz, what)) return e;
    Y3_CHECK_ARG(same_geom(dy, a) && same_geom(dy, dz) && coef, "%s: geometry/pointers", what);
    const long long total = pixels(a) * (a->c / 4);
    if (dres) {
        if (int e = check_view4(dres, what)) return e;
        Y3_CHECK_ARG(same_geom(dy, dres), "%s: dres geometry", what);
        hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, dim3(stream_blocks(total, 256)), dim3(256), 0, (hipStream_t)stream, dy->ptr, dy->ld, a->ptr,
                           a->ld, coef, alpha, dz->ptr, dz->ld, dres->ptr, dres->ld, dres_accumulate, pixels(a), a->c);
    } else {
        hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3(stream_blocks(total, 256)), dim3(256), 0, (hipStream_t)stream, dy->ptr, dy->ld, a->ptr,
                           a->ld, coef, alpha, dz->ptr, dz->ld, nullptr, 0, 0, pixels(a), a->c);
    }
    Y3_CHECK_LAUNCH(what);
    return Y3_OK;
}
extern "C" int y3_bn_bwd_apply(const y3_tensor* dy, const y3_tensor* a, const float* coef, float alpha, const y3_tensor* dz,
                               y3_stream_t stream) {
    return bn_bwd_apply_launch("bn_bwd_apply", dy, a, coef, alpha, dz, nullptr, 0, stream);
}
extern "C" int y3_bn_bwd_apply_fanin(const y3_tensor* dy, const y3_tensor* a, const float* coef, float alpha, const y3_tensor* dz,
                                     const y3_tensor* dres, int dres_accumulate, y3_stream_t stream) {
    Y3_CHECK_ARG(dres && dres->ptr, "bn_bwd_apply_fanin: null dres");
    return bn_bwd_apply_launch("bn_bwd_apply_fanin", dy, a, coef, alpha, dz, dres, dres_accumulate, stream);
}

// ---------------------------------------------------------------------------
// upsample_2x = frozen all-ones Conv2DTranspose(k2,s2): a channel sum, broadcast
// ---------------------------------------------------------------------------
// one wave per input pixel
__global__ __launch_bounds__(256) void upsample_fwd_kernel(const float* __restrict__ in, int in_ld, int C, float* __restrict__ out, int out_ld,
                                                           int outC, int N, int H, int W) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long npix = (long long)N * H * W;
    if (wave >= npix) return;
    const float* src = in + wave * in_ld;
    float s = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(src + c);
        s += (v.x + v.y) + (v.z + v.w);
    }
    s = y3_wave_sum(s);
    const int n = (int)(wave / ((long long)H * W));
    const int r = (int)(wave - (long long)n * H * W);
    const int i = r / W, j = r - i * W;
    const float4 o = make_float4(s, s, s, s);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            float* dst = out + (((long long)n * 2 * H + 2 * i + a) * 2 * W + 2 * j + b) * out_ld;
            for (int c = lane * 4; c < outC; c += 256) *reinterpret_cast<float4*>(dst + c) = o;
        }
}
extern "C" int y3_upsample_sum2x_fwd(const y3_tensor* in, const y3_tensor* out, y3_stream_t stream) {
    if (int e = check_view4(in, "upsample_fwd in")) return e;
    if (int e = check_view4(out, "upsample_fwd out")) return e;
    Y3_CHECK_ARG(out->n == in->n && out->h == 2 * in->h && out->w == 2 * in->w, "upsample_fwd: geometry");
    const long long npix = pixels(in);
    hipLaunchKernelGGL(upsample_fwd_kernel, dim3(y3_cdiv(npix, 4)), dim3(256), 0, (hipStream_t)stream, in->ptr, in->ld, in->c, out->ptr,
                       out->ld, out->c, in->n, in->h, in->w);
    Y3_CHECK_LAUNCH("upsample_fwd");
    return Y3_OK;
}

__global__ __launch_bounds__(256) void upsample_bwd_kernel(const float* __restrict__ dout, int do_ld, int outC, float* __restrict__ din,
                                                           int di_ld, int C, int N, int H, int W) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long npix = (long long)N * H * W;
    if (wave >= npix) return;
    const int n = (int)(wave / ((long long)H * W));
    const int r = (int)(wave - (long long)n * H * W);
    const int i = r / W, j = r - i * W;
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const float* src = dout + (((long long)n * 2 * H + 2 * i + a) * 2 * W + 2 * j + b) * do_ld;
            for (int c = lane * 4; c < outC; c += 256) {
                const float4 v = *reinterpret_cast<const float4*>(src + c);
                s += (v.x + v.y) + (v.z + v.w);
            }
        }
    s = y3_wave_sum(s);
    const float4 o = make_float4(s, s, s, s);
    float* dst = din + wave * di_ld;
    for (int c = lane * 4; c < C; c += 256) *reinterpret_cast<float4*>(dst + c) = o;
}
extern "C" int y3_upsample_sum2x_bwd(const y3_tensor* dout, const y3_tensor* din, y3_stream_t stream) {
    if (int e = check_view4(dout, "upsample_bwd dout")) return e;
    if (int e = check_view4(din, "upsample_bwd din")) return e;
    Y3_CHECK_ARG(dout->n == din->n && dout->h == 2 * din->h && dout->w == 2 * din->w, "upsample_bwd: geometry");
    const long long npix = pixels(din);
    hipLaunchKernelGGL(upsample_bwd_kernel, dim3(y3_cdiv(npix, 4)), dim3(256), 0, (hipStream_t)stream, dout->ptr, dout->ld, dout->c, din->ptr,
                       din->ld, din->c, din->n, din->h, din->w);
    Y3_CHECK_LAUNCH("upsample_bwd");
    return Y3_OK;
}

// ---------------------------------------------------------------------------
// data movement
// ---------------------------------------------------------------------------
template <bool ADD>
__global__ void copy_add_kernel(const float* __restrict__ src, int s_ld, float* __restrict__ dst, int d_ld, long long npix, int c4n) {
    const long long total = npix * c4n;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const long long pix = i / c4n;
        const int c = (int)(i - pix * c4n) * 4;
        float4 v = *reinterpret_cast<const float4*>(src + pix * s_ld + c);
        if (ADD) {
            const float4 d = *reinterpret_cast<const float4*>(dst + pix * d_ld + c);
            v.x += d.x;
            v.y += d.y;
            v.z += d.z;
            v.w += d.w;
        }
        *reinterpret_cast<float4*>(dst + pix * d_ld + c) = v;
    }
}
template <bool ADD>
static int copy_add(const y3_tensor* src, const y3_tensor* dst, y3_stream_t stream, const char* what) {
    if (int e = check_view4(src, what)) return e;
    if (int e = check_view4(dst, what)) return e;
    Y3_CHECK_ARG(same_geom(src, dst), "%s: geometry", what);
    const long long total = pixels(src) * (src->c / 4);
    hipLaunchKernelGGL((copy_add_kernel<ADD>), dim3(stream_blocks(total, 256)), dim3(256), 0, (hipStream_t)stream, src->ptr, src->ld, dst->ptr,
                       dst->ld, pixels(src), src->c / 4);
    Y3_CHECK_LAUNCH(what);
    return Y3_OK;
}
extern "C" int y3_copy(const y3_tensor* src, const y3_tensor* dst, y3_stream_t stream) { return copy_add<false>(src, dst, stream, "copy"); }
extern "C" int y3_add_inplace(const y3_tensor* src, const y3_tensor* dst, y3_stream_t stream) {
    return copy_add<true>(src, dst, stream, "add_inplace");
}

__global__ void fill_kernel(float* p, size_t count, float v) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) p[i] = v;
}
extern "C" int y3_fill(float* ptr, size_t count, float value, y3_stream_t stream) {
    Y3_CHECK_ARG(ptr || count == 0, "fill: null pointer");
    if (count == 0) return Y3_OK;
    hipLaunchKernelGGL(fill_kernel, dim3(stream_blocks((long long)count, 256)), dim3(256), 0, (hipStream_t)stream, ptr, count, value);
    Y3_CHECK_LAUNCH("fill");
    return Y3_OK;
}

// [N,C,H,W] -> NHWC with channel padding: one thread per (pixel), reads C planes (coalesced along W)
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, int C, long long hw, float* __restrict__ dst, int d_ld, int dC, long long npix) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += stride) {
        const long long n = i / hw, r = i - n * hw;
        for (int c = 0; c < dC; ++c) dst[i * d_ld + c] = c < C ? src[(n * C + c) * hw + r] : 0.f;
    }
}
extern "C" int y3_nchw_to_nhwc(const float* src, int n, int c, int h, int w, const y3_tensor* dst, y3_stream_t stream) {
    if (int e = check_view(dst, "nchw_to_nhwc dst")) return e;
    Y3_CHECK_ARG(src && dst->n == n && dst->h == h && dst->w == w && dst->c >= c, "nchw_to_nhwc: geometry");
    const long long npix = (long long)n * h * w;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(stream_blocks(npix, 256)), dim3(256), 0, (hipStream_t)stream, src, c, (long long)h * w, dst->ptr,
                       dst->ld, dst->c, npix);
    Y3_CHECK_LAUNCH("nchw_to_nhwc");
    return Y3_OK;
}
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ src, int s_ld, int C, long long hw, float* __restrict__ dst, long long total) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const long long r = i % hw;
        const long long nc = i / hw;
        const long long n = nc / C;
        const int c = (int)(nc - n * C);
        dst[i] = src[(n * hw + r) * s_ld + c];
    }
}
extern "C" int y3_nhwc_to_nchw(const y3_tensor* src, float* dst, y3_stream_t stream) {
    if (int e = check_view(src, "nhwc_to_nchw src")) return e;
    Y3_CHECK_ARG(dst, "nhwc_to_nchw: null dst");
    const long long total = pixels(src) * src->c;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(stream_blocks(total, 256)), dim3(256), 0, (hipStream_t)stream, src->ptr, src->ld, src->c,
                       (long long)src->h * src->w, dst, total);
    Y3_CHECK_LAUNCH("nhwc_to_nchw");
    return Y3_OK;
}

// out[c] = sum over pixels; one block per channel, fp64 accumulation, fixed order (deterministic)
__global__ __launch_bounds__(1024) void colsum_kernel(const float* __restrict__ src, int ld, int C, long long npix, float* __restrict__ out) {
    __shared__ double sm[1024];
    const int c = blockIdx.x;
    double s = 0.0;
    for (long long r = threadIdx.x; r < npix; r += 1024) s += (double)src[r * ld + c];
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[c] = (float)sm[0];
}
extern "C" int y3_colsum(const y3_tensor* src, float* out, y3_stream_t stream) {
    if (int e = check_view(src, "colsum src")) return e;
    Y3_CHECK_ARG(out && src->c <= 65535, "colsum: bad args");
    hipLaunchKernelGGL(colsum_kernel, dim3(src->c), dim3(1024), 0, (hipStream_t)stream, src->ptr, src->ld, src->c, pixels(src), out);
    Y3_CHECK_LAUNCH("colsum");
    return Y3_OK;
}

// ---------------------------------------------------------------------------
// Keras Adam (App. C5), fused over the whole parameter arena
// ---------------------------------------------------------------------------
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t count4,
                            size_t count, const float* __restrict__ lr_t_dev, float b1, float b2, float eps) {
    const float lr_t = *lr_t_dev;
    const float o1 = 1.f - b1, o2 = 1.f - b2;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count4; i += stride) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        const float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
#define Y3_ADAM1(f)                              \
    mm.f += (gg.f - mm.f) * o1;                  \
    vv.f += (gg.f * gg.f - vv.f) * o2;           \
    pp.f -= (mm.f * lr_t) / (sqrtf(vv.f) + eps);
        Y3_ADAM1(x) Y3_ADAM1(y) Y3_ADAM1(z) Y3_ADAM1(w)
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    // tail
    if (blockIdx.x == 0 && threadIdx.x < (count & 3)) {
        const size_t i = (count4 << 2) + threadIdx.x;
        float mm = m[i], vv = v[i];
        mm += (g[i] - mm) * o1;
        vv += (g[i] * g[i] - vv) * o2;
        p[i] -= (mm * lr_t) / (sqrtf(vv) + eps);
        m[i] = mm;
        v[i] = vv;
    }
}
extern "C" int y3_adam_step(float* param, const float* grad, float* m, float* v, size_t count, const float* lr_t_dev, float beta1, float beta2,
                            float eps, y3_stream_t stream) {
    Y3_CHECK_ARG(param && grad && m && v && lr_t_dev, "adam_step: null pointer");
    Y3_CHECK_ARG((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "adam_step: arenas must be 16-byte aligned");
    if (count == 0) return Y3_OK;
    hipLaunchKernelGGL(adam_kernel, dim3(stream_blocks((long long)(count / 4 + 1), 256)), dim3(256), 0, (hipStream_t)stream, param, grad, m, v,
                       count / 4, count, lr_t_dev, beta1, beta2, eps);
    Y3_CHECK_LAUNCH("adam_step");
    return Y3_OK;
}

// ---------------------------------------------------------------------------
// z-score (imagereader.py:34-46): per image mean / population std, fp64 partials
// ---------------------------------------------------------------------------
#define Y3_ZS_BLOCKS 128
__global__ __launch_bounds__(256) void zscore_partial_kernel(const float* __restrict__ in, size_t count, double* __restrict__ ws) {
    __shared__ double sm[2][256];
    const int img = blockIdx.y;
    const float* src = in + (size_t)img * count;
    double s = 0.0, q = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)Y3_ZS_BLOCKS * 256) {
        const double v = (double)src[i];
        s += v;
        q += v * v;
    }
    sm[0][threadIdx.x] = s;
    sm[1][threadIdx.x] = q;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            sm[0][threadIdx.x] += sm[0][threadIdx.x + o];
            sm[1][threadIdx.x] += sm[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        ws[((size_t)img * Y3_ZS_BLOCKS + blockIdx.x) * 2 + 0] = sm[0][0];
        ws[((size_t)img * Y3_ZS_BLOCKS + blockIdx.x) * 2 + 1] = sm[1][0];
    }
}
__global__ void zscore_apply_kernel(const float* __restrict__ in, float* __restrict__ out, size_t count, const double* __restrict__ ws) {
    const int img = blockIdx.y;
    double s = 0.0, q = 0.0;
    for (int b = 0; b < Y3_ZS_BLOCKS; ++b) {  // every thread re-reduces the 128 partials (L2-resident, fixed order)
        s += ws[((size_t)img * Y3_ZS_BLOCKS + b) * 2 + 0];
        q += ws[((size_t)img * Y3_ZS_BLOCKS + b) * 2 + 1];
    }
    const double mean = s / (double)count;
    double var = q / (double)count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float mv = (float)mean, sd = (float)sqrt(var);
    const float* src = in + (size_t)img * count;
    float* dst = out + (size_t)img * count;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    if (sd <= 1.0f) {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) dst[i] = src[i] - mv;
    } else {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) dst[i] = (src[i] - mv) / sd;
    }
}
// 128 partial pairs per image, and one {mean, std} float pair per image behind them (y3_tile_gather_zscore_nhwc)
extern "C" size_t y3_zscore_workspace_bytes(int n) { return (size_t)n * Y3_ZS_BLOCKS * 2 * sizeof(double) + (size_t)n * 2 * sizeof(float); }
extern "C" int y3_zscore(const float* in, float* out, int n, size_t count, void* workspace, y3_stream_t stream) {
    Y3_CHECK_ARG(in && out && workspace && n > 0 && count > 0, "zscore: bad args");
    hipLaunchKernelGGL(zscore_partial_kernel, dim3(Y3_ZS_BLOCKS, n), dim3(256), 0, (hipStream_t)stream, in, count, (double*)workspace);
    Y3_CHECK_LAUNCH("zscore_partial");
    int blocks = stream_blocks((long long)count, 256);
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(zscore_apply_kernel, dim3(blocks, n), dim3(256), 0, (hipStream_t)stream, in, out, count, (const double*)workspace);
    Y3_CHECK_LAUNCH("zscore_apply");
    return Y3_OK;
}

// ---------------------------------------------------------------------------
// inference_tiled.py:29-100 on the device: crop + np.pad(mode='reflect') + HWC -> CHW + astype(float32) for a batch of tiles.
// table rows {y0, ny, pre_y, x0, nx, pre_x}: the clamped crop img[y0:y0+ny, x0:x0+nx] and the number of reflected rows /
// columns in front of it.  numpy's reflect is relative to the CROP (period 2(n-1), no edge repeat), repeated when the
// pad is longer than the crop.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int reflect_index(int p, int n) {  // p relative to the crop, any sign
    if (n == 1) return 0;
    const int period = 2 * (n - 1);
    int q = p % period;
    if (q < 0) q += period;
    return q < n ? q : period - q;
}
template <typename T>
__global__ void tile_gather_kernel(const T* __restrict__ img, int W, int C, const int* __restrict__ table, int th, int tw, float* __restrict__ out) {
    const int* row = table + blockIdx.z * 6;
    const int y0 = row[0], ny = row[1], pre_y = row[2], x0 = row[3], nx = row[4], pre_x = row[5];
    const int y = blockIdx.y;
    const int sy = y0 + reflect_index(y - pre_y, ny);
    const T* src = img + (size_t)sy * W * C;
    float* dst = out + ((size_t)blockIdx.z * C * th + y) * tw;
    for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < tw; x += gridDim.x * blockDim.x) {
        const int sx = x0 + reflect_index(x - pre_x, nx);
        for (int c = 0; c < C; ++c) dst[(size_t)c * th * tw + x] = (float)src[(size_t)sx * C + c];
    }
}
extern "C" int y3_tile_gather(const void* img, int dtype, int height, int width, int channels, const int* table_dev, int ntiles, int tile_h,
                              int tile_w, float* out, y3_stream_t stream) {
    Y3_CHECK_ARG(img && table_dev && out, "tile_gather: null pointer");
    Y3_CHECK_ARG(height > 0 && width > 0 && channels > 0 && ntiles > 0 && tile_h > 0 && tile_w > 0, "tile_gather: bad dims");
    Y3_CHECK_ARG(tile_h <= 65535 && ntiles <= 65535, "tile_gather: grid too large");
    const dim3 grid(y3_cdiv(tile_w, 256), tile_h, ntiles), block(256);
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case 0: hipLaunchKernelGGL(tile_gather_kernel<unsigned char>, grid, block, 0, st, (const unsigned char*)img, width, channels, table_dev, tile_h, tile_w, out); break;
        case 1: hipLaunchKernelGGL(tile_gather_kernel<unsigned short>, grid, block, 0, st, (const unsigned short*)img, width, channels, table_dev, tile_h, tile_w, out); break;
        case 2: hipLaunchKernelGGL(tile_gather_kernel<float>, grid, block, 0, st, (const float*)img, width, channels, table_dev, tile_h, tile_w, out); break;
        default: Y3_CHECK_ARG(false, "tile_gather: dtype %d (0 = u8, 1 = u16, 2 = f32)", dtype);
    }
    Y3_CHECK_LAUNCH("tile_gather");
    return Y3_OK;
}

// ---------------------------------------------------------------------------
// y3_tile_gather + y3_zscore + y3_nchw_to_nhwc in two passes over the image instead of five over the tiles: statistics of every
// (reflect-padded) tile straight from the image, then gather + normalise + channel-pad into the network's NHWC input buffer.
// For float images the statistics walk the tile in the order zscore_partial_kernel walks the gathered tensor ([C][th][tw], same
// 128 x 256 strided partition, same tree); for 8- and 16-bit images the sums are exact whatever the order.  Either way mean / std
// -- and with them every output value -- are the bits the three-kernel path gives.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void tile_stats_kernel(const T* __restrict__ img, int W, int C, const int* __restrict__ table, int th, int tw,
                                                         double* __restrict__ ws) {
    __shared__ double sm[2][256];
    const int tile = blockIdx.y;
    const int* row = table + tile * 6;
    const int y0 = row[0], ny = row[1], pre_y = row[2], x0 = row[3], nx = row[4], pre_x = row[5];
    const unsigned plane = (unsigned)th * (unsigned)tw, count = plane * (unsigned)C;
    double s = 0.0, q = 0.0;
    if constexpr (sizeof(T) < 4) {
        // integer pixels: the sums of values and squares are exact in fp64 (< 2^53 for a 16-bit tile of any size the grid allows), so
        // the order is free: rows by block, pixels by lane, the channels of a pixel together (coalesced reads, no index divisions)
        for (int y = blockIdx.x; y < th; y += Y3_ZS_BLOCKS) {
            const T* src = img + (size_t)(y0 + reflect_index(y - pre_y, ny)) * W * C;
            for (int x = threadIdx.x; x < tw; x += 256) {
                const T* px = src + (size_t)(x0 + reflect_index(x - pre_x, nx)) * C;
                for (int c = 0; c < C; ++c) {
                    const double v = (double)px[c];
                    s += v;
                    q += v * v;
                }
            }
        }
    } else {
        for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < count; i += (unsigned)Y3_ZS_BLOCKS * 256u) {
            const unsigned c = i / plane, r = i - c * plane;
            const unsigned y = r / (unsigned)tw, x = r - y * (unsigned)tw;
            const int sy = y0 + reflect_index((int)y - pre_y, ny), sx = x0 + reflect_index((int)x - pre_x, nx);
            const double v = (double)(float)img[((size_t)sy * W + sx) * C + c];
            s += v;
            q += v * v;
        }
    }
    sm[0][threadIdx.x] = s;
    sm[1][threadIdx.x] = q;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            sm[0][threadIdx.x] += sm[0][threadIdx.x + o];
            sm[1][threadIdx.x] += sm[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        ws[((size_t)tile * Y3_ZS_BLOCKS + blockIdx.x) * 2 + 0] = sm[0][0];
        ws[((size_t)tile * Y3_ZS_BLOCKS + blockIdx.x) * 2 + 1] = sm[1][0];
    }
}
// mean / std of every tile from its 128 partial pairs, summed in the order of zscore_apply_kernel: {mean, std} as two floats behind the partials
__global__ void tile_stats_finalize_kernel(const double* __restrict__ ws, int ntiles, double count, float* __restrict__ mvsd) {
    const int tile = blockIdx.x * blockDim.x + threadIdx.x;
    if (tile >= ntiles) return;
    double s = 0.0, q = 0.0;
    for (int b = 0; b < Y3_ZS_BLOCKS; ++b) {
        s += ws[((size_t)tile * Y3_ZS_BLOCKS + b) * 2 + 0];
        q += ws[((size_t)tile * Y3_ZS_BLOCKS + b) * 2 + 1];
    }
    const double mean = s / count;
    double var = q / count - mean * mean;
    if (var < 0.0) var = 0.0;
    mvsd[2 * tile] = (float)mean;
    mvsd[2 * tile + 1] = (float)sqrt(var);
}
#define Y3_TG_ROWS 8       // tile rows per workgroup of the gather
template <typename T>
__global__ __launch_bounds__(256) void tile_gather_norm_kernel(const T* __restrict__ img, int W, int C, const int* __restrict__ table, int th, int tw,
                                                               const float* __restrict__ mvsd, float* __restrict__ out, int cpitch) {
    const int tile = blockIdx.z;
    const float mv = mvsd[2 * tile], sd = mvsd[2 * tile + 1];
    const bool divide = !(sd <= 1.0f);
    const int* row = table + tile * 6;
    const int y0 = row[0], ny = row[1], pre_y = row[2], x0 = row[3], nx = row[4], pre_x = row[5];
    for (int y = blockIdx.y * Y3_TG_ROWS; y < min(th, (int)(blockIdx.y + 1) * Y3_TG_ROWS); ++y) {
        const int sy = y0 + reflect_index(y - pre_y, ny);
        const T* src = img + (size_t)sy * W * C;
        float* dst = out + ((size_t)tile * th + y) * tw * cpitch;
        for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < tw; x += gridDim.x * blockDim.x) {
            const int sx = x0 + reflect_index(x - pre_x, nx);
            if (cpitch == 4 && C <= 4) {
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                for (int c = 0; c < C; ++c) {
                    const float d = (float)src[(size_t)sx * C + c] - mv;
                    v[c] = divide ? d / sd : d;
                }
                *reinterpret_cast<float4*>(dst + (size_t)x * 4) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                for (int c = 0; c < cpitch; ++c) {
                    float d = 0.f;
                    if (c < C) {
                        d = (float)src[(size_t)sx * C + c] - mv;
                        d = divide ? d / sd : d;
                    }
                    dst[(size_t)x * cpitch + c] = d;
                }
            }
        }
    }
}
extern "C" int y3_tile_gather_zscore_nhwc(const void* img, int dtype, int height, int width, int channels, const int* table_dev, int ntiles,
                                          int tile_h, int tile_w, float* out, int channel_pitch, void* workspace, y3_stream_t stream) {
    Y3_CHECK_ARG(img && table_dev && out && workspace, "tile_gather_zscore_nhwc: null pointer");
    Y3_CHECK_ARG(height > 0 && width > 0 && channels > 0 && ntiles > 0 && tile_h > 0 && tile_w > 0, "tile_gather_zscore_nhwc: bad dims");
    Y3_CHECK_ARG(channel_pitch >= channels && (channel_pitch & 3) == 0 && ((uintptr_t)out & 15) == 0, "tile_gather_zscore_nhwc: channel pitch %d (multiple of 4, >= %d channels), out 16-byte aligned", channel_pitch, channels);
    Y3_CHECK_ARG(tile_h <= 65535 && ntiles <= 65535 && (long long)channels * tile_h * tile_w < 0x7fffffffLL, "tile_gather_zscore_nhwc: grid too large");
    const dim3 sgrid(Y3_ZS_BLOCKS, ntiles), grid(1, y3_cdiv(tile_h, Y3_TG_ROWS), ntiles), block(256);
    hipStream_t st = (hipStream_t)stream;
    double* ws = (double*)workspace;
    // {mean, std} of every tile: two floats per tile behind the partials (y3_zscore_workspace_bytes counts them)
    float* mvsd = (float*)(ws + (size_t)ntiles * Y3_ZS_BLOCKS * 2);
    const double count = (double)channels * (double)tile_h * (double)tile_w;
    switch (dtype) {
        case 0:
            hipLaunchKernelGGL(tile_stats_kernel<unsigned char>, sgrid, block, 0, st, (const unsigned char*)img, width, channels, table_dev, tile_h, tile_w, ws);
            hipLaunchKernelGGL(tile_stats_finalize_kernel, dim3(y3_cdiv(ntiles, 64)), dim3(64), 0, st, (const double*)ws, ntiles, count, mvsd);
            hipLaunchKernelGGL(tile_gather_norm_kernel<unsigned char>, grid, block, 0, st, (const unsigned char*)img, width, channels, table_dev, tile_h, tile_w, (const float*)mvsd, out, channel_pitch);
            break;
        case 1:
            hipLaunchKernelGGL(tile_stats_kernel<unsigned short>, sgrid, block, 0, st, (const unsigned short*)img, width, channels, table_dev, tile_h, tile_w, ws);
            hipLaunchKernelGGL(tile_stats_finalize_kernel, dim3(y3_cdiv(ntiles, 64)), dim3(64), 0, st, (const double*)ws, ntiles, count, mvsd);
            hipLaunchKernelGGL(tile_gather_norm_kernel<unsigned short>, grid, block, 0, st, (const unsigned short*)img, width, channels, table_dev, tile_h, tile_w, (const float*)mvsd, out, channel_pitch);
            break;
        case 2:
            hipLaunchKernelGGL(tile_stats_kernel<float>, sgrid, block, 0, st, (const float*)img, width, channels, table_dev, tile_h, tile_w, ws);
            hipLaunchKernelGGL(tile_stats_finalize_kernel, dim3(y3_cdiv(ntiles, 64)), dim3(64), 0, st, (const double*)ws, ntiles, count, mvsd);
            hipLaunchKernelGGL(tile_gather_norm_kernel<float>, grid, block, 0, st, (const float*)img, width, channels, table_dev, tile_h, tile_w, (const float*)mvsd, out, channel_pitch);
            break;
        default: Y3_CHECK_ARG(false, "tile_gather_zscore_nhwc: dtype %d (0 = u8, 1 = u16, 2 = f32)", dtype);
    }
    Y3_CHECK_LAUNCH("tile_gather_zscore_nhwc");
    return Y3_OK;
}

