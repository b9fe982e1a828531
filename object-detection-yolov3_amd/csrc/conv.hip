// Implicit-GEMM convolution on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32:
// f32 in, f32 accumulate, bit-exact fmaf chain) for gfx950.
//
// One gather-GEMM kernel serves conv forward (model.py:29-39,108-120), the data
// gradient (tape.gradient, model.py:496; stride-2 layers as 4 output-parity
// launches so no MFMA work is wasted on structural zeros) and 1x1 convs:
//     dst[pix(m)][n] = epi( sum_{t<taps} sum_{c<C} src[pix(m)+off(t)][c] * wt[wsel(t)][c][n] )
// and a second kernel computes the kernel gradient
//     dw[t][c][n]    = sum_m src[pix(m)+off(t)][c] * ddst[m][n]        (split over m)
//
// Data layout: activations NHWC (channel-contiguous) so the GEMM K axis (tap, c)
// is contiguous per pixel; weights [tap][C][Nout] (Keras order) so a B row is
// Nout-contiguous.  Tiles are staged global -> registers -> LDS (double buffered,
// one barrier per K step); operands are read from LDS as
//   A: one ds_read_b128 per lane = 4 consecutive k of its pixel row (rows padded
//      to BK+4 floats: conflict-free), feeding 4 MFMAs
//   B: ds_read_b32, 32 consecutive n per half-wave (conflict-free)
// The k order inside a K step is permuted identically for A and B
// (lane half h takes k = 8*kk + 4*h + i), which only reorders the fp32 sum.
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "common.h"

struct ConvArgs {
    const float* src;
    const float* wt;
    float* dst;
    const float* bias;
    const float* scale;
    const float* shift;
    const float* resid;
    float* stats;
    unsigned long long tap_dhdw;  // 4 bits per tap: (dh+1) | (dw+1) << 2
    unsigned long long tap_wsel;  // 4 bits per tap: weight slice
    int H, W, C, logC, cmask, src_ld;
    int OH, OW, sh, sw;
    int DH, DW, dsh, dsw, doh, dow, dst_ld, dense_dst;
    int resid_ld;
    int Nout, K, M;
    unsigned flags;
    float alpha;
    int nbn;
    int src_n, wt_rows;  // host-side only: batch of src, rows of the weight matrix (descriptor sizes)
    const float* bn_a;   // data gradient with BatchNorm-backward statistics in the epilogue (y3_conv2d_dgrad_bn): the activation `a`
    float* bn_part;      // ... and the partial sums [row tile][6][Nout]
    int bn_a_ld;
    int x3;              // Y3_CONV_X3: the launch runs conv_x3.hip and `wt` is the copy with K contiguous per output column
};

template <int BM, int BN, int WM, int WN, int BK>
__global__ __launch_bounds__(64 * WM * WN) void conv_igemm_kernel(const ConvArgs p) {
    constexpr int THREADS = 64 * WM * WN;
    constexpr int LDA = BK + 4;
    constexpr int TM = BM / WM, TN = BN / WN, MB = TM / 32, NB = TN / 32;
    constexpr int KV = BK / 4;
    constexpr int A_TOTAL = BM * KV, A_LOADS = (A_TOTAL + THREADS - 1) / THREADS;
    constexpr int BN4 = BN / 4;
    constexpr int B_TOTAL = BK * BN4, B_LOADS = (B_TOTAL + THREADS - 1) / THREADS;
    static_assert(THREADS % KV == 0 && TM % 32 == 0 && TN % 32 == 0, "tile shape");

    __shared__ __attribute__((aligned(16))) float As[2][BM * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK * BN];
    __shared__ float red[2][WM][BN];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int bid = y3_xcd_remap(blockIdx.x, gridDim.x);
    const int bm = bid / p.nbn, bn = bid % p.nbn;
    const int m0 = bm * BM, n0 = bn * BN;

    // ---- per-thread A rows (pixel decomposition is K-invariant)
    int a_pix[A_LOADS], a_ih0[A_LOADS], a_iw0[A_LOADS];
    bool a_ok[A_LOADS];
    const int a_kv = tid % KV;
    const int ohw = p.OH * p.OW;
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
        const int idx = tid + i * THREADS;
        const int row = idx / KV;
        const int m = m0 + row;
        const bool ok = (idx < A_TOTAL) && (m < p.M);
        const int mm = ok ? m : 0;
        const int n = mm / ohw;
        const int r = mm - n * ohw;
        const int oh = r / p.OW;
        const int ow = r - oh * p.OW;
        a_ih0[i] = oh * p.sh;
        a_iw0[i] = ow * p.sw;
        a_pix[i] = (n * p.H + a_ih0[i]) * p.W + a_iw0[i];
        a_ok[i] = ok;
    }
    const bool n_aligned = (p.Nout & 3) == 0;

    float4 ra[A_LOADS], rb[B_LOADS];
    auto gload = [&](int k0) {
        {
            const int k = k0 + a_kv * 4;
            const int tap = k >> p.logC;
            const int c = k & p.cmask;
            const int code = (int)((p.tap_dhdw >> (4 * tap)) & 15ull);
            const int dh = (code & 3) - 1, dw = (code >> 2) - 1;
            const int doff = dh * p.W + dw;
#pragma unroll
            for (int i = 0; i < A_LOADS; ++i) {
                const int ih = a_ih0[i] + dh, iw = a_iw0[i] + dw;
                const bool ok = a_ok[i] && (k < p.K) && ((unsigned)ih < (unsigned)p.H) && ((unsigned)iw < (unsigned)p.W);
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok) {
                    v = *reinterpret_cast<const float4*>(p.src + ((long long)(a_pix[i] + doff) * p.src_ld + c));
                    if (k + 4 > p.K) {  // K tail (K % 4 != 0): zero the lanes past K
                        if (k + 1 >= p.K) v.y = 0.f;
                        if (k + 2 >= p.K) v.z = 0.f;
                        if (k + 3 >= p.K) v.w = 0.f;
                    }
                }
                ra[i] = v;
            }
        }
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) {
            const int idx = tid + i * THREADS;
            const int kr = idx / BN4, n4 = idx % BN4;
            const int k = k0 + kr;
            const int n = n0 + n4 * 4;
            const int tap = k >> p.logC;
            const int c = k & p.cmask;
            const int ws = (int)((p.tap_wsel >> (4 * tap)) & 15ull);
            const float* g = p.wt + ((long long)(ws * p.C + c) * p.Nout + n);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < B_TOTAL && k < p.K) {
                if (n_aligned) {
                    if (n < p.Nout) v = *reinterpret_cast<const float4*>(g);
                } else {
                    if (n < p.Nout) v.x = g[0];
                    if (n + 1 < p.Nout) v.y = g[1];
                    if (n + 2 < p.Nout) v.z = g[2];
                    if (n + 3 < p.Nout) v.w = g[3];
                }
            }
            rb[i] = v;
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            const int idx = tid + i * THREADS;
            if (idx < A_TOTAL) *reinterpret_cast<float4*>(&As[buf][(idx / KV) * LDA + a_kv * 4]) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) {
            const int idx = tid + i * THREADS;
            if (idx < B_TOTAL) *reinterpret_cast<float4*>(&Bs[buf][(idx / BN4) * BN + (idx % BN4) * 4]) = rb[i];
        }
    };

    f32x16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (p.K + BK - 1) / BK;
    gload(0);
    lstore(0);
    __syncthreads();
    int cur = 0;
    for (int ks = 0; ks < nk; ++ks) {
        const bool more = ks + 1 < nk;
        if (more) gload((ks + 1) * BK);
        const float* as = &As[cur][(wm * TM + l31) * LDA + lh * 4];
        const float* bs = &Bs[cur][(lh * 4) * BN + wn * TN + l31];
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            float4 av[MB];
            float bv[NB][4];
#pragma unroll
            for (int i = 0; i < MB; ++i) av[i] = *reinterpret_cast<const float4*>(as + i * 32 * LDA + kk * 8);
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) bv[j][q] = bs[(kk * 8 + q) * BN + j * 32];
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].x, bv[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].y, bv[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].z, bv[j][2], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].w, bv[j][3], acc[i][j], 0, 0, 0);
                }
        }
        if (more) lstore(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const bool do_lrelu = p.flags & Y3_EPI_LRELU;
    const bool do_accum = p.flags & Y3_EPI_ACCUM;
    float ssum[NB], ssq[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) ssum[j] = ssq[j] = 0.f;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int n = n0 + wn * TN + j * 32 + l31;
        const bool nok = n < p.Nout;
        const float bias = (p.bias && nok) ? p.bias[n] : 0.f;
        const float sc = (p.scale && nok) ? p.scale[n] : 1.f;
        const float sf = (p.scale && nok) ? p.shift[n] : 0.f;
#pragma unroll
        for (int i = 0; i < MB; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < p.M && nok) {
                    float v = acc[i][j][r] + bias;
                    if (do_lrelu) v = v > 0.f ? v : p.alpha * v;
                    ssum[j] += v;
                    ssq[j] += v * v;
                    long long pix;
                    if (p.dense_dst) {
                        pix = m;
                    } else {
                        const int nimg = m / ohw;
                        const int rr = m - nimg * ohw;
                        const int oh = rr / p.OW;
                        const int ow = rr - oh * p.OW;
                        pix = ((long long)nimg * p.DH + oh * p.dsh + p.doh) * p.DW + ow * p.dsw + p.dow;
                    }
                    if (p.scale) v = v * sc + sf;
                    if (p.resid) v += p.resid[pix * p.resid_ld + n];
                    float* d = p.dst + pix * p.dst_ld + n;
                    if (do_accum) v += *d;
                    *d = v;
                }
            }
        }
    }
    if (p.stats) {
        // per-(row tile, channel) partial sums for training-mode BatchNorm: combine the two
        // half-waves, then the WM waves stacked along M, in a fixed order (deterministic)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const float s = ssum[j] + __shfl_xor(ssum[j], 32);
            const float q = ssq[j] + __shfl_xor(ssq[j], 32);
            if (lh == 0) {
                red[0][wm][wn * TN + j * 32 + l31] = s;
                red[1][wm][wn * TN + j * 32 + l31] = q;
            }
        }
        __syncthreads();
        for (int c = tid; c < 2 * BN; c += THREADS) {
            const int which = c / BN, col = c % BN;
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) s += red[which][w][col];
            const int n = n0 + col;
            if (n < p.Nout) p.stats[((long long)bm * 2 + which) * p.Nout + n] = s;
        }
    }
}

#include "conv_fast.h"   // FastArgs, the work-item decode, the split-K hand-off and the epilogue (shared with conv_x3.hip)

template <int BM, int BN, int WM, int WN, int BK, bool DENSE, bool BNS = false>
__device__ __forceinline__ void conv_fast_body(const FastArgs& p, const int braw, const int grid) {
    constexpr int THREADS = 64 * WM * WN;
    constexpr int LDA = BK + 4;
    constexpr int TM = BM / WM, TN = BN / WN, MB = TM / 32, NB = TN / 32;
    constexpr int KV = BK / 4;
    constexpr int A_TOTAL = BM * KV, A_LOADS = A_TOTAL / THREADS;
    constexpr int BN4 = BN / 4;
    constexpr int B_TOTAL = BK * BN4, B_LOADS = (B_TOTAL + THREADS - 1) / THREADS;
    static_assert(A_TOTAL % THREADS == 0 && THREADS % KV == 0 && TM % 32 == 0 && TN % 32 == 0, "tile shape");

    __shared__ __attribute__((aligned(16))) float As[2][BM * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK * BN];
    // column sums of the epilogue (forward statistics: 2, BNS: 6 per column and wave row).  BNS: they live in the A stage, dead
    // once the K loop is over -- at 64x64 the workgroup's LDS decides whether 8 or 7 workgroups share a CU, and the split-K plan
    // (~2 000 workgroups) is sized for 8
    static_assert(!BNS || 6 * WM * BN <= 2 * BM * LDA, "BNS column sums do not fit the A stage");
    __shared__ float red_own[BNS ? 1 : 2][BNS ? 1 : WM][BNS ? 1 : BN];
    float (*red)[WM][BN] = BNS ? reinterpret_cast<float (*)[WM][BN]>(&As[0][0]) : reinterpret_cast<float (*)[WM][BN]>(&red_own[0][0][0]);

    Y3_TSTAMP(0);
    Y3_ABL_INIT();
#ifdef Y3_TIMING
    if (y3_timing_buf && threadIdx.x == 0) {
        y3_timing_buf[(size_t)blockIdx.x * 8 + 4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_ID
        y3_timing_buf[(size_t)blockIdx.x * 8 + 5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // XCC_ID
        y3_timing_buf[(size_t)blockIdx.x * 8 + 6] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    const FastWork fw = conv_fast_decode<BM, BN, WM, WN, BK>(p, braw, grid);
    const int tid = fw.tid, lane = fw.lane, l31 = fw.l31, lh = fw.lh, wm = fw.wm, wn = fw.wn;
    const int m0 = fw.m0, n0 = fw.n0, kbeg = fw.kbeg, kend = fw.kend;
    const int ohw = fw.ohw, OW = fw.OW, aM = fw.aM, aH = fw.aH, aW = fw.aW, src_ld = fw.src_ld, csh = fw.csh, csw = fw.csw, ntaps = fw.ntaps, Nout = fw.Nout;
    const Y3Div dv_ohw = fw.dv_ohw, dv_ow = fw.dv_ow;
    (void)lane;

    const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.src), 0, p.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wt), 0, p.wt_bytes, 0x00020000);

    // loop-invariant per-lane offsets
    unsigned a_voff[A_LOADS], a_mask[A_LOADS];
    const int a_kv = tid % KV;
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
        const int row = (tid + i * THREADS) / KV;
        const int m = m0 + row;
        const bool ok = m < aM;
        const int mm = ok ? m : 0;
        const int n = y3_div(mm, dv_ohw);
        const int r = mm - n * ohw;
        const int oh = y3_div(r, dv_ow);
        const int ow = r - oh * OW;
        const int ih0 = oh * csh, iw0 = ow * csw;
        a_voff[i] = (unsigned)(((n * aH + ih0) * aW + iw0) * src_ld + a_kv * 4) * 4u;
        unsigned msk = 0;
        // (decoding the taps from a packed 64-bit scalar instead of these two table loads per tap made the 3x3 layers 8 % slower: measured)
        for (int t = 0; t < ntaps; ++t) {
            const int ih = ih0 + p.tap_dh[t], iw = iw0 + p.tap_dw[t];
            if (ok && (unsigned)ih < (unsigned)aH && (unsigned)iw < (unsigned)aW) msk |= 1u << t;
        }
        a_mask[i] = msk;
    }
    unsigned b_voff[B_LOADS];
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) {
        const int idx = tid + i * THREADS;
        const int kr = idx / BN4, n = n0 + (idx % BN4) * 4;
        b_voff[i] = (idx < B_TOTAL && n < Nout) ? (unsigned)(kr * Nout + n) * 4u : Y3_OOB;
    }

    // Global -> register -> LDS staging: TWO register sets (tile of K step s lives in set s & 1), so that a tile's loads are
    // issued two K steps before its LDS stores (one step in the first version: the layers whose launches leave fewer than four
    // workgroups per CU -- every 1x1 layer -- stood at the stores' vmcnt wait).  8 VGPRs per set on the 64x64 tile.
    f32x4 ra[2][A_LOADS], rb[2][B_LOADS];
    // Tap -> (source offset, weight row) without a table: indexing the argument-segment tables with a run-time tap is a scalar
    // MEMORY load, and an SMEM load in flight forces every later LDS wait to lgkmcnt(0) (scalar loads return out of order),
    // which stalls the MFMA stream on the fragment reads just issued (a register copy of the tables was turned into a
    // scratch array by the compiler: worse).  Every tap list this kernel sees is a (rows x tg_nx) grid in row-major order
    // (3x3, 1x1, and the 1/2/2/4-tap parity classes of a stride-2 data gradient), so both quantities are affine in the
    // grid coordinates: a few scalar ALU instructions.
    // K steps at or beyond kend are DEAD (the loop below runs an even number of uniform steps and issues loads three steps ahead):
    // tap 31 selects a mask bit that is never set and `dead` pushes the weight offsets out of range, so their loads fetch nothing
    // and return zeros.
    struct Soff {
        int tap, cb, toff, wrow;
        unsigned dead;
    };
    auto soff_prep = [&](int k0) {
        Soff o;
        const bool live = k0 < kend;
        o.dead = live ? 0u : Y3_OOB;
        k0 = live ? k0 : kbeg;
        if (p.korder) {        // (wave-uniform: scalar unit)
            const int step = k0 >> 4, sh = p.korder - 1, g = y3_div(step, p.dv_taps), r = step - g * (p.ntaps << sh);
            o.tap = r >> sh;
            o.cb = (g << (4 + sh)) + ((r & sh) << 4);
        } else {
            o.tap = k0 >> p.logC;
            o.cb = k0 & p.cmask;
        }
        const int ty = (o.tap * p.tg_mul) >> 5;                  // tap / tg_nx for tap < 9
        const int tx = o.tap - ty * p.tg_nx;
        o.toff = p.tg_off0 + ty * p.tg_offy + tx * p.tg_offx;
        o.wrow = p.tg_w0 + ty * p.tg_wy + tx * p.tg_wx;
        o.tap = live ? o.tap : 31;
        return o;
    };
    auto gload_at = [&](const Soff& o, auto S) {
        constexpr int set = decltype(S)::value;
        const unsigned a_soff = (unsigned)(o.toff + o.cb * 4);
        const unsigned b_soff = (unsigned)((o.wrow + o.cb) * p.Nout) * 4u;
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            const unsigned vo = ((a_mask[i] >> o.tap) & 1u) ? a_voff[i] : Y3_OOB;
            ra[set][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_src, vo, a_soff, 0);
        }
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) rb[set][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_wt, b_voff[i] | o.dead, b_soff, 0);
    };
    auto gload = [&](int k0, auto S) { gload_at(soff_prep(k0), S); };
    auto lstore = [&](int buf) {     // set 0 -> LDS buffer `buf` (prologue only)
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) *reinterpret_cast<f32x4*>(&As[buf][((tid + i * THREADS) / KV) * LDA + a_kv * 4]) = ra[0][i];
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) {
            const int idx = tid + i * THREADS;
            if (B_TOTAL % THREADS == 0 || idx < B_TOTAL) *reinterpret_cast<f32x4*>(&Bs[buf][(idx / BN4) * BN + (idx % BN4) * 4]) = rb[0][i];
        }
    };

    f32x16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (kend - kbeg) / BK;
    {
        static_assert(BK == 16, "the hand-interleaved K loop is written for K steps of 16");
        // Software-pipelined K loop, interleaved by hand.  A wave issues MFMAs in order and cannot queue them: once an MFMA
        // has issued, only the 64 cycles it executes are free for other instructions, so memory instructions left in a block
        // between two MFMA groups idle the matrix pipe (measured with tools/probe/conv_timing: 2 758 cycles per step for
        // 2 048 of MFMA with that layout and everything else ablated).  Here every MFMA is followed by at most one or two
        // memory instructions (a "slot"), fenced with sched_barrier so that the compiler keeps the order:
        //   group 0 (fragment set 0):  slots carry the LDS reads of set 1 (this step's second half), then the LDS stores of
        //                              the NEXT step's tile (global data loaded one step earlier)
        //   LDS-only barrier            publishes the next buffer; global loads stay in flight across it
        //   group 1 (fragment set 1):  slots carry the global loads for the step THREE ahead (register set (ks + 1) & 1, the one
        //                              group 0 has just stored from), then the LDS reads of the next step's set 0
        // The steady-state body is two steps (the register set is a compile-time index) without branches; the last three K steps
        // run through the peeled forms.
        f32x4 fa[2][MB];
        float fb[2][NB][4];
        constexpr int NM = MB * NB * 4;             // MFMAs per group
        constexpr int NW = A_LOADS + B_LOADS;       // LDS stores = global loads per K step
        const float* as_base = &As[0][(wm * TM + l31) * LDA + lh * 4];
        const float* bs_base = &Bs[0][(lh * 4) * BN + wn * TN + l31];
        constexpr int ABUF = BM * LDA, BBUF = BK * BN;
        // events of a half step, in issue order.  group 0: MB reads (A), 2 reads (B, q pairs), NW stores.
        //                                        group 1: NW global loads, MB reads (A), 2 reads (B).
        constexpr int NE = MB + 2 + NW;
        Soff nxt;
        int knext = 0;
        auto event = [&](auto G, auto E, auto H1, auto H2, auto P) {
            constexpr int g = decltype(G)::value, e = decltype(E)::value;
            constexpr bool h1 = decltype(H1)::value, h2 = decltype(H2)::value;
            constexpr int cur = decltype(P)::value, rs = cur ^ 1;      // LDS buffer of this step; register set stored / reloaded
            if constexpr (g == 0) {
                // fragment set 1 <- group 1 of the current buffer; then stores of the next tile
                if constexpr (e < MB) {
                    fa[1][e] = *reinterpret_cast<const f32x4*>(as_base + cur * ABUF + e * 32 * LDA + 8);
                } else if constexpr (e < MB + 2) {
                    constexpr int q0 = (e - MB) * 2;
#pragma unroll
                    for (int q = q0; q < q0 + 2; ++q)
#pragma unroll
                        for (int j = 0; j < NB; ++j) fb[1][j][q] = bs_base[cur * BBUF + (8 + q) * BN + j * 32];
                } else if constexpr (h1) {
                    constexpr int w = e - MB - 2;
                    if (!Y3_ABL(2)) {
                        if constexpr (w < A_LOADS) {
                            *reinterpret_cast<f32x4*>(&As[cur ^ 1][((tid + w * THREADS) / KV) * LDA + a_kv * 4]) = ra[rs][w];
                        } else {
                            constexpr int wb = w - A_LOADS;
                            const int idx = tid + wb * THREADS;
                            if (B_TOTAL % THREADS == 0 || idx < B_TOTAL) *reinterpret_cast<f32x4*>(&Bs[cur ^ 1][(idx / BN4) * BN + (idx % BN4) * 4]) = rb[rs][wb];
                        }
                    }
                }
            } else {
                if constexpr (e < NW) {
                    if constexpr (h2) {
                        if (!Y3_ABL(1)) {
                            if constexpr (e < A_LOADS) {
                                const unsigned vo = ((a_mask[e] >> nxt.tap) & 1u) ? a_voff[e] : Y3_OOB;
                                ra[rs][e] = __builtin_amdgcn_raw_buffer_load_b128(rs_src, vo, (unsigned)(nxt.toff + nxt.cb * 4), 0);
                            } else {
                                rb[rs][e - A_LOADS] = __builtin_amdgcn_raw_buffer_load_b128(rs_wt, b_voff[e - A_LOADS] | nxt.dead, (unsigned)((nxt.wrow + nxt.cb) * p.Nout) * 4u, 0);
                            }
                        }
                    }
                } else if constexpr (h1) {
                    // fragment set 0 <- group 0 of the NEXT buffer
                    if constexpr (e < NW + MB) {
                        constexpr int i = e - NW;
                        fa[0][i] = *reinterpret_cast<const f32x4*>(as_base + (cur ^ 1) * ABUF + i * 32 * LDA);
                    } else {
                        constexpr int q0 = (e - NW - MB) * 2;
#pragma unroll
                        for (int q = q0; q < q0 + 2; ++q)
#pragma unroll
                            for (int j = 0; j < NB; ++j) fb[0][j][q] = bs_base[(cur ^ 1) * BBUF + q * BN + j * 32];
                    }
                }
            }
        };
        auto half = [&](auto G, auto H1, auto H2, auto P) {
            constexpr int g = decltype(G)::value;
            y3_for_each_ic(std::make_integer_sequence<int, NM>{}, [&](auto M) {
                constexpr int m = decltype(M)::value;
                constexpr int q = m / (MB * NB), i = (m / NB) % MB, j = m % NB;
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g][i][q], fb[g][j][q], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                // this slot's share of the NE events (spread evenly when there are more events than MFMAs)
                constexpr int e0 = NE <= NM ? m : m * NE / NM, e1 = NE <= NM ? (m < NE ? m + 1 : m) : (m + 1) * NE / NM;
                y3_for_each_ic(std::make_integer_sequence<int, e1 - e0>{}, [&](auto D) { event(G, std::integral_constant<int, e0 + decltype(D)::value>{}, H1, H2, P); });
                // the scalar arithmetic for the NEXT iteration's global-load offsets rides in the slot after this one's loads
                if constexpr (g == 1 && decltype(H2)::value && m == (NW < NM ? NW : NM - 1)) nxt = soff_prep(knext);
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        auto step = [&](auto H1, auto H2, auto P, int ks) {     // P = ks & 1
            knext = kbeg + (ks + 4) * BK;
            half(std::integral_constant<int, 0>{}, H1, H2, P);
            if constexpr (decltype(H1)::value) {
                if (!Y3_ABL(4)) y3_lds_barrier();
            }
            half(std::integral_constant<int, 1>{}, H1, H2, P);
        };
        gload(kbeg, std::integral_constant<int, 0>{});
        lstore(0);
        __syncthreads();
        Y3_TSTAMP(1);
        gload(kbeg + BK, std::integral_constant<int, 1>{});
        gload(kbeg + 2 * BK, std::integral_constant<int, 0>{});
        nxt = soff_prep(kbeg + 3 * BK);      // offsets of the loads issued in iteration 0 (K step 3)
        {   // fragment set 0 of step 0
#pragma unroll
            for (int i = 0; i < MB; ++i) fa[0][i] = *reinterpret_cast<const f32x4*>(as_base + i * 32 * LDA);
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) fb[0][j][q] = bs_base[q * BN + j * 32];
        }
        // Uniform steps, two per iteration: an odd step count is padded with one dead step (all-zero operands), and the stores,
        // barrier and fragment reads of the last step serve a tile nobody multiplies.
        for (int ks = 0; ks < nk; ks += 2) {
            step(std::true_type{}, std::true_type{}, std::integral_constant<int, 0>{}, ks);
            step(std::true_type{}, std::true_type{}, std::integral_constant<int, 1>{}, ks + 1);
        }
    }

    Y3_TSTAMP(2);
    conv_fast_finish<BM, BN, WM, WN, DENSE, BNS>(p, fw, acc, red);
}

// registers: at least 3 waves per SIMD; the BNS 64x64 variant must also stay at 64 VGPRs (8 waves per SIMD like its plain twin, see `red`)
template <int BM, int BN, int WM, int WN, int BK, bool DENSE, bool BNS = false>
__global__ __launch_bounds__(64 * WM * WN, (BNS && BM * BN <= 64 * 64) ? 8 : 3) void conv_igemm_fast_kernel(const FastArgs p) {
    conv_fast_body<BM, BN, WM, WN, BK, DENSE, BNS>(p, (int)blockIdx.x, (int)gridDim.x);
}

// Up to four independent gather-GEMMs in ONE launch: the (row parity, column parity) classes of a stride-2 data gradient.
// Each class has its own tap list, K and destination lattice; block ranges [first[c], first[c+1]) select the class.  As
// separate launches the four small grids ran one after the other, each with its own ramp-up and tail.
template <int BM, int BN, int WM, int WN, int BK, bool BNS = false>
__global__ __launch_bounds__(64 * WM * WN, 3) void conv_igemm_fast_multi_kernel(const FastArgs4 m) {
    int c = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) c += ((int)blockIdx.x >= m.first[i]) ? 1 : 0;
    conv_fast_body<BM, BN, WM, WN, BK, false, BNS>(m.a[c], (int)blockIdx.x - m.first[c], m.first[c + 1] - m.first[c]);
}

// ---------------------------------------------------------------------------
// kernel gradient: out[z][k][n] = sum_{m in chunk z} A[m][k] * ddst[m][n]
// ---------------------------------------------------------------------------
#ifndef Y3_WGRAD_WAVES_EU
#define Y3_WGRAD_WAVES_EU 3   // waves per SIMD the register allocation aims at (140 VGPRs; 2 lets the compiler take 204)
#endif
#ifndef Y3_WG_TABLE
#define Y3_WG_TABLE 2048
#endif
//   // pixels per split the LDS pixel table holds (plan_wgrad keeps chunks below it)
template <int BKR, int BN, int WM, int WN, int BP>
__global__ __launch_bounds__(64 * WM * WN, Y3_WGRAD_WAVES_EU) void conv_wgrad_kernel(const WgradArgs p) {
    constexpr int THREADS = 64 * WM * WN;
    constexpr int TM = BKR / WM, TN = BN / WN, MB = TM / 32, NB = TN / 32;
    constexpr int KR4 = BKR / 4, BN4 = BN / 4;
    constexpr int A_TOTAL = BP * KR4, A_LOADS = (A_TOTAL + THREADS - 1) / THREADS;
    constexpr int B_TOTAL = BP * BN4, B_LOADS = (B_TOTAL + THREADS - 1) / THREADS;
    static_assert(THREADS % KR4 == 0 && THREADS % BN4 == 0 && BP == 16, "loader shape");
    constexpr int A_PSTEP = THREADS / KR4;  // pixel distance between a thread's consecutive A loads
    constexpr int B_PSTEP = THREADS / BN4;

    __shared__ __attribute__((aligned(16))) float At[2][BP * BKR];
    __shared__ __attribute__((aligned(16))) float Bt[2][BP * BN];
    __shared__ uint2 pix[Y3_WG_TABLE];   // per pixel of this split: {byte offset of its (dh, dw) = (0, 0) source pixel, tap validity bits}

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    // 1-D grid of (splits rounded up to 8) x tiles workgroups.  Blocks are dealt round-robin over the 8 XCDs, so block
    // L lands on XCD L % 8: give every XCD whole pixel chunks (split = 8 * group + xcd) and let it walk all the
    // (k-tile, n-tile) pairs of that chunk, so the chunk's activations / gradients are fetched into ONE L2 instead of eight.
    // (With few chunks that idles XCDs or loses to the tile-major order, measured: then tiles are spread with the usual remap instead.)
    // the scalars of the index decode, fetched from the argument segment in one batch (see conv_fast_body)
    int splits = p.splits, tiles = p.tiles, nbn = p.nbn, chunk = p.chunk, aM = p.M, ohw = p.ohw, OW = p.OW, aH = p.H, aW = p.W;
    int src_ld = p.src_ld, csh = p.sh, csw = p.sw, ntaps = p.ntaps;
    unsigned dt_m = p.dv_tiles.mul, dn_m = p.dv_nbn.mul, dohw_m = p.dv_ohw.mul, dow_m = p.dv_ow.mul;
    int dt_s = p.dv_tiles.shift, dn_s = p.dv_nbn.shift, dohw_s = p.dv_ohw.shift, dow_s = p.dv_ow.shift;
    unsigned dhdw_lo = (unsigned)p.tap_dhdw, dhdw_hi = (unsigned)(p.tap_dhdw >> 32);
    Y3_PIN_S(splits); Y3_PIN_S(tiles); Y3_PIN_S(nbn); Y3_PIN_S(chunk); Y3_PIN_S(aM); Y3_PIN_S(ohw); Y3_PIN_S(OW); Y3_PIN_S(aH); Y3_PIN_S(aW);
    Y3_PIN_S(src_ld); Y3_PIN_S(csh); Y3_PIN_S(csw); Y3_PIN_S(ntaps);
    Y3_PIN_S(dt_m); Y3_PIN_S(dn_m); Y3_PIN_S(dohw_m); Y3_PIN_S(dow_m); Y3_PIN_S(dt_s); Y3_PIN_S(dn_s); Y3_PIN_S(dohw_s); Y3_PIN_S(dow_s);
    Y3_PIN_S(dhdw_lo); Y3_PIN_S(dhdw_hi);
    const Y3Div dv_tiles = {dt_m, dt_s}, dv_nbn = {dn_m, dn_s}, dv_ohw = {dohw_m, dohw_s}, dv_ow = {dow_m, dow_s};
    const unsigned long long tap_dhdw = ((unsigned long long)dhdw_hi << 32) | dhdw_lo;
    int split, bid;
    if (splits >= 32) {
        const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
        const int q = y3_div(jx, dv_tiles);
        split = q * 8 + xcd;
        bid = jx - q * tiles;
    } else {
        // every XCD takes one contiguous run of (pixel run, tile) items, tile fastest: the tiles of a pixel run read the same
        // pixels of src and ddst (each at its own tap / channel block), so a run is fetched by one or two XCDs instead of all eight
        const int item = y3_xcd_remap((int)blockIdx.x, (int)gridDim.x);
        split = y3_div(item, dv_tiles);
        bid = item - split * tiles;
    }
    if (split >= splits) return;
    const int bk = y3_div(bid, dv_nbn), bn = bid - bk * nbn;
    const int k0 = bk * BKR, n0 = bn * BN;
    const int mbeg = split * chunk;
    const int mend = min(aM, mbeg + chunk);
    const int nsteps = (mend > mbeg) ? (mend - mbeg + BP - 1) / BP : 0;

    // Pixel table: the (image, row, column) decomposition of every pixel of the split and the zero-padding test of every
    // tap are done ONCE here; the K loop then only reads {offset, mask} from LDS one step ahead.  Loads are buffer loads
    // whose per-lane offset points past the descriptor's range whenever the element does not exist (padding, pixels beyond
    // the split, K / Nout tails): the hardware returns zeros, the loop has no branches and no 64-bit address arithmetic.
    {
        for (int pl = tid; pl < nsteps * BP; pl += THREADS) {
            const int m = mbeg + pl;
            unsigned off = 0, msk = 0;
            if (m < mend) {
                const int n = y3_div(m, dv_ohw);
                const int r = m - n * ohw;
                const int oh = y3_div(r, dv_ow), ow = r - oh * OW;
                const int ih0 = oh * csh, iw0 = ow * csw;
                off = (unsigned)(((n * aH + ih0) * aW + iw0) * src_ld) * 4u;
                for (int t = 0; t < ntaps; ++t) {
                    const int code = (int)((tap_dhdw >> (4 * t)) & 15ull);
                    const int ih = ih0 + (code & 3) - 1, iw = iw0 + (code >> 2) - 1;
                    if ((unsigned)ih < (unsigned)aH && (unsigned)iw < (unsigned)aW) msk |= 1u << t;
                }
            }
            pix[pl] = make_uint2(off, msk);
        }
    }
    const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.src), 0, p.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_dd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.ddst), 0, p.dd_bytes, 0x00020000);
    // A: this thread always loads the same 4 k's (tap, c..c+3); only the pixel advances
    const int a_kv = tid % KR4;
    const int ak = k0 + a_kv * 4;
    const bool ak_ok = ak < p.K;            // K % 4 == 0 (Cin % 4 == 0): a quad is inside K or outside as a whole
    const int atap = ak_ok ? (ak >> p.logC) : 0;
    const int ac = ak & p.cmask;
    const int acode = (int)((tap_dhdw >> (4 * atap)) & 15ull);
    const int a_tapoff = ((((acode & 3) - 1) * p.W + ((acode >> 2) - 1)) * p.src_ld + ac) * 4;   // may be negative; only used where the tap is valid
    const unsigned a_bit = ak_ok ? 1u << atap : 0u;
    const int a_pp0 = tid / KR4;
    const int b_n4 = tid % BN4;
    const int bnn = n0 + b_n4 * 4;
    const bool bn_ok = bnn < p.Nout;
    const int b_pp0 = tid / BN4;
    const unsigned b_lane = (unsigned)(bnn * 4);

    // two register sets (tile of step s in set s & 1): a tile's loads are issued two steps before its LDS stores (see conv_fast_body)
    f32x4 ra[2][A_LOADS], rb[2][B_LOADS];
    uint2 pe[A_LOADS];                     // table entries of the step whose loads are issued next
    auto tload = [&](int step) {          // LDS read of the table entries for `step`
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            const int pp = a_pp0 + i * A_PSTEP;
            pe[i] = pix[step * BP + (A_TOTAL % THREADS == 0 || pp < BP ? pp : 0)];
        }
    };
    // `step` may lie beyond the split (the loop pads to an even step count and loads three steps ahead): such loads are
    // pushed out of range and return zeros
    auto gload_a = [&](int i, int step, auto S) {
        const int pp = a_pp0 + i * A_PSTEP;
        const bool ok = ((pe[i].y & a_bit) != 0) & (A_TOTAL % THREADS == 0 || pp < BP) & (step < nsteps);
        const unsigned off = pe[i].x + (unsigned)a_tapoff;
        const unsigned vo = ok ? off : Y3_OOB;
        ra[decltype(S)::value][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_src, vo, 0, 0);
    };
    auto gload_b = [&](int i, int step, auto S) {
        const int pp = b_pp0 + i * B_PSTEP;
        const int m = mbeg + step * BP + pp;
        const unsigned off = (unsigned)(m * p.dd_ld) * 4u + b_lane;      // unconditional arithmetic + select: no divergent branch in the loop
        const bool ok = (B_TOTAL % THREADS == 0 || pp < BP) & (m < mend) & bn_ok;
        const unsigned vo = ok ? off : Y3_OOB;
        rb[decltype(S)::value][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_dd, vo, 0, 0);
    };
    auto lstore_a = [&](int i, int buf, auto S) {
        const int pp = a_pp0 + i * A_PSTEP;
        if (A_TOTAL % THREADS == 0 || pp < BP) *reinterpret_cast<f32x4*>(&At[buf][pp * BKR + a_kv * 4]) = ra[decltype(S)::value][i];
    };
    auto lstore_b = [&](int i, int buf, auto S) {
        const int pp = b_pp0 + i * B_PSTEP;
        if (B_TOTAL % THREADS == 0 || pp < BP) *reinterpret_cast<f32x4*>(&Bt[buf][pp * BN + b_n4 * 4]) = rb[decltype(S)::value][i];
    };

    f32x16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Same hand-interleaved software pipeline as conv_fast_body (PIPE 2): a step is BP = 16 pixels = two groups of four
    // pixel pairs; every MFMA is followed by a slot with at most a couple of memory instructions.
    //   group 0: LDS reads of fragment set 1 (pairs 4..7 of this buffer), then the LDS stores of the next step's tile
    //   LDS-only barrier
    //   group 1: global loads for the step THREE ahead into the register set group 0 has just stored from (table entries were
    //            read a step ahead), the table entries of the step after that, then the LDS reads of set 0 from the next buffer
    // Uniform steps, two per loop iteration (the register set is a compile-time index); an odd count is padded with a dead step.
    constexpr int GP = 4;
    constexpr int NM = GP * MB * NB;
    constexpr int NW = A_LOADS + B_LOADS;
    constexpr int NE = GP + NW + 1;      // events per half: 4 fragment-pair reads + NW stores / loads (+ the table read in group 1)
    float fa[2][GP][MB], fb[2][GP][NB];
    const float* as_base = &At[0][lh * BKR + wm * TM + l31];
    const float* bs_base = &Bt[0][lh * BN + wn * TN + l31];
    auto ldf_pair = [&](int buf, int g, int u, int set) {
#pragma unroll
        for (int i = 0; i < MB; ++i) fa[set][u][i] = as_base[buf * (BP * BKR) + (g * GP + u) * 2 * BKR + i * 32];
#pragma unroll
        for (int j = 0; j < NB; ++j) fb[set][u][j] = bs_base[buf * (BP * BN) + (g * GP + u) * 2 * BN + j * 32];
    };
    auto event = [&](auto G, auto E, auto P, int st) {
        constexpr int g = decltype(G)::value, e = decltype(E)::value;
        constexpr int cur = decltype(P)::value;
        constexpr std::integral_constant<int, cur ^ 1> RS{};     // register set stored in group 0 and reloaded in group 1
        if constexpr (g == 0) {
            if constexpr (e < GP) {
                ldf_pair(cur, 1, e, 1);
            } else if constexpr (e < GP + NW) {
                constexpr int w = e - GP;
                if constexpr (w < A_LOADS)
                    lstore_a(w, cur ^ 1, RS);
                else
                    lstore_b(w - A_LOADS, cur ^ 1, RS);
            }
        } else {
            if constexpr (e < NW) {
                if constexpr (e < A_LOADS)
                    gload_a(e, st + 3, RS);
                else
                    gload_b(e - A_LOADS, st + 3, RS);
            } else if constexpr (e == NW) {
                tload(min(st + 4, nsteps - 1));
            } else {
                ldf_pair(cur ^ 1, 0, e - NW - 1, 0);
            }
        }
    };
    auto half = [&](auto G, auto P, int st) {
        constexpr int g = decltype(G)::value;
        y3_for_each_ic(std::make_integer_sequence<int, NM>{}, [&](auto Mi) {
            constexpr int m = decltype(Mi)::value;
            constexpr int u = m / (MB * NB), i = (m / NB) % MB, j = m % NB;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g][u][i], fb[g][u][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            constexpr int e0 = NE <= NM ? m : m * NE / NM, e1 = NE <= NM ? (m < NE ? m + 1 : m) : (m + 1) * NE / NM;
            y3_for_each_ic(std::make_integer_sequence<int, e1 - e0>{}, [&](auto D) { event(G, std::integral_constant<int, e0 + decltype(D)::value>{}, P, st); });
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    auto step = [&](auto P, int st) {      // P = st & 1
        half(std::integral_constant<int, 0>{}, P, st);
        y3_lds_barrier();
        half(std::integral_constant<int, 1>{}, P, st);
    };
    __syncthreads();                       // pixel table complete
    if (nsteps > 0) {
        constexpr std::integral_constant<int, 0> S0{};
        constexpr std::integral_constant<int, 1> S1{};
        tload(0);
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) gload_a(i, 0, S0);
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) gload_b(i, 0, S0);
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) lstore_a(i, 0, S0);
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) lstore_b(i, 0, S0);
        __syncthreads();
        tload(min(1, nsteps - 1));
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) gload_a(i, 1, S1);
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) gload_b(i, 1, S1);
        tload(min(2, nsteps - 1));
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) gload_a(i, 2, S0);
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) gload_b(i, 2, S0);
        tload(min(3, nsteps - 1));
#pragma unroll
        for (int u = 0; u < GP; ++u) ldf_pair(0, 0, u, 0);
        for (int st = 0; st < nsteps; st += 2) {
            step(S0, st);
            step(S1, st + 1);
        }
    }
    __syncthreads();                       // (the reduction below reuses At as a flag word)

    if (p.splits > 1 && p.tickets != nullptr) {
        // Reduction over the pixel splits inside the kernel (the host takes this path for 2 .. Y3_WG_FANIN splits only): every
        // split parks its raw accumulators (slab[tile][split][r4][thread], 16-byte sc1 stores) and takes the tile's ticket;
        // the split that draws the last ticket sums all of them in split order (bit-reproducible whichever split it is) and
        // writes dw.  With more splits a flat reduction serialises too many slab reads in one workgroup and a tree of such
        // levels cost more than the streaming slab_reduce_kernel (measured): those launches carry no tickets.
        // Hand-off rules: see conv_fast_body.
        constexpr int R4 = MB * NB * 4;
        const __amdgpu_buffer_rsrc_t rs_slab = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, 0x7ffffff0, 0x00020000);
        const unsigned item_bytes = (unsigned)(R4 * THREADS * 16);
        int* flag = reinterpret_cast<int*>(&At[0][0]);
        const int count = p.splits;
        const unsigned level0 = (unsigned)(bid * count) * item_bytes + (unsigned)tid * 16u;
        {
            const unsigned base = level0 + (unsigned)split * item_bytes;
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        f32x4 v = {acc[i][j][4 * r], acc[i][j][4 * r + 1], acc[i][j][4 * r + 2], acc[i][j][4 * r + 3]};
                        __builtin_amdgcn_raw_buffer_store_b128(v, rs_slab, base, (unsigned)(((i * NB + j) * 4 + r) * THREADS * 16), 16 /* sc1 */);
                    }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                int* tk = p.tickets + bid;
                const int old = __hip_atomic_fetch_add(tk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int last = old == count - 1;
                if (last) __hip_atomic_store(tk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *flag = last;
            }
            __syncthreads();
            if (!*flag) return;
#pragma unroll 1
            for (int z = 0; z < count; ++z) {
                const unsigned base2 = level0 + (unsigned)z * item_bytes;
#pragma unroll
                for (int i = 0; i < MB; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const f32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_slab, base2, (unsigned)(((i * NB + j) * 4 + r) * THREADS * 16), 16 /* sc1 */);
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[i][j][4 * r + e] = z == 0 ? v[e] : acc[i][j][4 * r + e] + v[e];
                        }
            }
        }
    }

    // no tickets: `out` is dw itself (one split) or the split's slab in the natural [K][Nout] layout (slab_reduce_kernel follows)
    float* out = p.tickets != nullptr ? p.dw : p.out + (p.splits > 1 ? (long long)split * p.K * p.Nout : 0ll);
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int n = n0 + wn * TN + j * 32 + l31;
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = k0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (k < p.K && n < p.Nout) out[(long long)k * p.Nout + n] = acc[i][j][r];
            }
    }
}

// dst[i] = sum_z slabs[z][i], fixed summation order (deterministic).  A block owns 64 consecutive outputs
// (16 float4 lanes) x 16 slab lanes: the slab axis is parallel too, so a few-thousand-element gradient
// split over ~1000 pixel chunks is not reduced by a handful of serial threads.  Used when a kernel gradient has more
// splits than the in-kernel reduction handles in one level (Y3_WG_FANIN): there the whole chip streams the slabs.
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dst, long long count, int nslabs) {
    __shared__ f32x4 sm[16][16];
    const int cl = threadIdx.x & 15, zl = threadIdx.x >> 4;
    const long long i = ((long long)blockIdx.x * 16 + cl) * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (i < count)
        for (int z = zl; z < nslabs; z += 16) acc += *reinterpret_cast<const f32x4*>(slabs + (long long)z * count + i);
    sm[zl][cl] = acc;
    __syncthreads();
    if (zl == 0 && i < count) {
#pragma unroll
        for (int z = 1; z < 16; ++z) acc += sm[z][cl];
        *reinterpret_cast<f32x4*>(dst + i) = acc;
    }
}

__global__ void transpose_weights_kernel(const float* __restrict__ w, float* __restrict__ wt, int cin, int cout) {
    // one 32x32 tile per block, blockIdx.z = tap;  wt[t][co][ci] = w[t][ci][co]
    __shared__ float tile[32][33];
    const int t = blockIdx.z;
    const int ci0 = blockIdx.y * 32, co0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: ty 0..7
    const float* src = w + (long long)t * cin * cout;
    float* dst = wt + (long long)t * cin * cout;
    for (int r = ty; r < 32; r += 8) {
        const int ci = ci0 + r, co = co0 + tx;
        tile[r][tx] = (ci < cin && co < cout) ? src[(long long)ci * cout + co] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int co = co0 + r, ci = ci0 + tx;
        if (ci < cin && co < cout) dst[(long long)co * cin + ci] = tile[tx][r];
    }
}

// All layers in one launch: table[l] = {arena offset, taps, cin, cout, first tile}; a block finds its layer by bisection.
__global__ void transpose_weights_batched_kernel(const float* __restrict__ params, float* __restrict__ params_t,
                                                 const int* __restrict__ table, int nlayers) {
    __shared__ float tile[32][33];
    int lo = 0, hi = nlayers - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid * 5 + 4] <= (int)blockIdx.x)
            lo = mid;
        else
            hi = mid - 1;
    }
    const int off = table[lo * 5], cin = table[lo * 5 + 2], cout = table[lo * 5 + 3];
    const int local = blockIdx.x - table[lo * 5 + 4];
    const int tco = (cout + 31) >> 5, tci = (cin + 31) >> 5;
    const int t = local / (tco * tci), rem = local % (tco * tci);
    const int ci0 = (rem / tco) * 32, co0 = (rem % tco) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* src = params + off + (long long)t * cin * cout;
    float* dst = params_t + off + (long long)t * cin * cout;
    for (int r = ty; r < 32; r += 8) {
        const int ci = ci0 + r, co = co0 + tx;
        tile[r][tx] = (ci < cin && co < cout) ? src[(long long)ci * cout + co] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int co = co0 + r, ci = ci0 + tx;
        if (ci < cin && co < cout) dst[(long long)co * cin + ci] = tile[tx][r];
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static int check_tensor(const y3_tensor* t, const char* name) {
    Y3_CHECK_ARG(t && t->ptr, "%s: null tensor", name);
    Y3_CHECK_ARG(t->n > 0 && t->h > 0 && t->w > 0 && t->c > 0 && t->ld >= t->c, "%s: bad dims n=%d h=%d w=%d c=%d ld=%d", name,
                 t->n, t->h, t->w, t->c, t->ld);
    Y3_CHECK_ARG((t->ld & 3) == 0, "%s: ld=%d must be a multiple of 4", name, t->ld);
    Y3_CHECK_ARG(((uintptr_t)t->ptr & 15) == 0, "%s: pointer must be 16-byte aligned", name);
    return 0;
}

struct TileCfg {
    int bm, bn, bk;
};

// Tuning switches.  The product library reads exactly one environment variable on this path, Y3_NO_FAST (generic kernel for
// every launch; exercised by the GPU tests).  Everything else -- tile override, split-K targets, kernel-gradient plan -- is a
// development knob of tools/: compiled in with -DY3_DEV only (make DEV=1 -> libyolo3hip_dev.so), constants otherwise.
#ifdef Y3_DEV
// Y3_TILE="bm,bn,bk" forces one configuration for every launch.
static bool tile_override(TileCfg* t) {
    static int state = 0;  // 0 unknown, 1 none, 2 set
    static TileCfg forced;
    if (state == 0) {
        const char* e = getenv("Y3_TILE");
        state = 1;
        if (e && sscanf(e, "%d,%d,%d", &forced.bm, &forced.bn, &forced.bk) == 3) state = 2;
    }
    if (state == 2) *t = forced;
    return state == 2;
}
static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
static const char* env_str(const char* name) { return getenv(name); }
#else
static bool tile_override(TileCfg*) { return false; }
static int env_int(const char*, int dflt) { return dflt; }
static const char* env_str(const char*) { return nullptr; }
#endif

// Tile choice from the measured sweep (tools/conv_tune.py, MI355X): a launch wants >= ~600 workgroups
// (256 CUs x 2-3 resident); prefer the largest tile that still gives that many, else 64x64 (+ split-K).
static TileCfg pick_tile(int M, int Nout) {
    TileCfg t;
    if (Nout <= 32)
        t = {128, 32, 16};
    else if (Nout <= 64)
        t = {128, 64, 16};
    else {
        const long long t128 = (long long)y3_cdiv(M, 128) * y3_cdiv(Nout, 128);
        const long long t64x128 = (long long)y3_cdiv(M, 64) * y3_cdiv(Nout, 128);
        if (t128 >= 600)
            t = {128, 128, 16};
        else if (t64x128 >= 600)
            t = {64, 128, 16};
        else
            t = {64, 64, 16};
    }
    TileCfg f;
    if (tile_override(&f) && f.bn <= ((Nout + 31) / 32) * 32) t = f;
    return t;
}

#define Y3_WS_HEADER (256 * 1024)   // bytes of tile tickets in front of the slabs (65 536 tiles)
#define Y3_MAX_TICKETS (Y3_WS_HEADER / 4)
struct ConvPlan {
    TileCfg t;
    int f, s0, s1, chunk0, chunk1;   // FastArgs::sk_*: tiles [0,f) in s0 slices of chunk0 K steps, the rest in s1 of chunk1
    int tiles, stats_tiles;
    size_t ws_bytes;
    int short_last = 0;              // x3 overflow plan: deal the short last slices to the blocks the dispatcher starts last (FastArgs::x3_mode bit 3)
};
// The K loop of conv_fast_body runs its steps in pairs (an odd count is padded with a dead step): slices get an even step count.
static inline int even_steps(int chunk) { return chunk + (chunk & 1); }
// The x3 kernels (conv_x3.hip): 128 x 128 tiles (128 x 64 for <= 64 output columns), two to three workgroups per CU.  Every
// launch of the layers they are used for is cut along K into slices of >= 12 K steps so that ~700 workgroups share the work
// evenly -- 2.7 per CU, dealt out as slots free up -- e.g. 338 tiles x 2, 172 x 4, 88 x 8 slices of 36 steps for the 3x3 layers
// of the 52 / 26 / 13 grids at batch 8.
static bool x3_shape_ok(int C, int Nout, int K, int ntaps) {
    return C % 16 == 0 && K % 16 == 0 && (ntaps == 1 || y3_is_pow2(C)) && Nout >= 32 && K / 16 >= 2;
}
static ConvPlan plan_conv_x3(int M, int Nout, int K, int ntaps) {
    ConvPlan pl;
    pl.t = {128, Nout <= 64 ? 64 : 128, 16};
    {
        static const int force_bn = env_int("Y3_X3_BN", 0);      // development: 64 = 128 x 64 tiles everywhere
        if (force_bn == 64 || force_bn == 128) pl.t.bn = Nout <= 64 ? 64 : force_bn;
    }
    const int tiles = y3_cdiv(M, pl.t.bm) * y3_cdiv(Nout, pl.t.bn);
    const int nk = K / 16;
    pl.tiles = pl.f = tiles;
    pl.s0 = pl.s1 = 1;
    pl.chunk0 = pl.chunk1 = even_steps(nk);
    static const int min_steps = env_int("Y3_X3_MINSTEPS", 12);
    // slices are whole units of K steps: 3x3 launches in units of 18 (two 16-channel chunks of nine taps: the patch kernel's loop
    // body, conv_x3.hip), the others in pairs of steps
    const int unit = ntaps == 9 ? 18 : 2;
    static const int rsplit_on = env_int("Y3_X3_RSPLIT", 2);      // 0 off, 1 only where the uniform split does not apply, 2 preferred above 256 tiles
    static const int force_ks = env_int("Y3_X3_KS", 0);           // development: this many K slices for every split launch
    static const int slots = env_int("Y3_X3_SLOTS", 512);         // two workgroups of the patch kernel per CU
    if (tiles <= Y3_MAX_TICKETS && tiles > 256 && rsplit_on && !force_ks) {
        // more tiles than CUs but too few to balance by themselves (338 tiles: a third of the CUs would carry two): whole rounds of
        // tiles stay whole -- no slabs for them -- and only the remainder round is cut, so that its pieces spread evenly
        const int F = tiles / 256 * 256, R = tiles - F;
        int best = 1;
        double best_cost = 1.0;
        for (int S = 2; S <= 6; ++S) {
            const int ch = y3_cdiv(y3_cdiv(nk, S), unit) * unit;
            if (ch * S != nk || ch < min_steps) continue;             // equal slices only
            const double cost = (double)y3_cdiv((long long)R * S, 256) / S + 0.03 * (S - 1);
            if (cost < best_cost - 1e-9) {
                best_cost = cost;
                best = S;
            }
        }
        if (R > 0 && best > 1) {
            pl.f = F;
            pl.chunk1 = nk / best;
            pl.s1 = best;
        }
    } else if (tiles <= Y3_MAX_TICKETS && tiles <= 256) {
        // Fewer tiles than CUs: cut along K so that the launch fills the chip's workgroup slots ONCE, two per CU -- a lone workgroup
        // (one wave per SIMD) runs its loop at a third of the MFMA rate, and a second, partly filled round costs a whole round
        // (measured, tools/probe/x3_ks_sweep.sh / x3_ks_sweep2.sh: 507 pieces 78 us, 338 pieces 97 us, 676 pieces 103 us on the same launch).
        // The slice counts a launch can have are ceil(nk / c) for c a multiple of the unit; take the largest count s_lo with
        // tiles * s_lo <= slots and give the next larger one, s_hi, to as many tiles as fill the rest of the slots.
        auto count_for = [&](int c) { return y3_cdiv(nk, c); };
        int c_lo = 0, c_hi = 0;      // chunk lengths of the counts s_lo <= slots / tiles < s_hi
        for (int c = y3_cdiv(nk, unit) * unit; c >= unit && c >= min_steps; c -= unit) {      // slice counts grow as c shrinks
            const int sc = count_for(c);
            if (sc > 16) break;
            const bool fits = force_ks ? sc <= force_ks : (long long)tiles * sc <= slots;
            if (fits)
                c_lo = c;
            else if (c_hi == 0 && !force_ks && (c_lo == 0 || sc > count_for(c_lo)))
                c_hi = c;
        }
        const int s_lo = c_lo > 0 ? count_for(c_lo) : 1;
        if (s_lo > 1) {
            pl.s0 = pl.s1 = s_lo;
            pl.chunk0 = pl.chunk1 = c_lo;
        }
        // Overflow by SHORT slices: when the next larger count leaves a last slice of at most a third of the others, every tile
        // takes that count although the launch then has a few blocks more than slots: the items are dealt so that every XCD's
        // blocks END with short slices (conv_fast_decode<SHORTLAST>), i.e. the blocks beyond the slots are short ones that start as
        // the first short ones finish -- and no tile needs the longer slices of s_lo (13x13 forward: 88 x 6 = 528 pieces of 54 / 18
        // steps instead of 80 x 6 + 8 x 4 with 72-step pieces: 92 -> 82 us).
        static const int overflow_on = env_int("Y3_X3_OVERFLOW", 1);
        if (c_hi > 0 && overflow_on && !force_ks && ntaps == 9) {
            const int s_hi = count_for(c_hi), last = nk - (s_hi - 1) * c_hi;
            if ((long long)tiles * s_hi <= slots + slots / 16 && 3 * last <= c_hi) {
                pl.f = tiles;
                pl.s0 = pl.s1 = s_hi;
                pl.chunk0 = pl.chunk1 = c_hi;
                pl.short_last = 1;
                c_hi = 0;
            }
        }
        if (c_hi > 0) {      // tiles [0, f) take the next larger count: f * s_hi + (tiles - f) * s_lo <= slots
            const int s_hi = count_for(c_hi);
            const int f = (int)((slots - (long long)tiles * s_lo) / (s_hi - s_lo));
            if (f > 0) {
                pl.f = f < tiles ? f : tiles;
                pl.s0 = s_hi;
                pl.chunk0 = c_hi;
            }
        }
    }
    pl.stats_tiles = y3_cdiv(M, pl.t.bm);
    const long long split_items = pl.s0 > 1 ? (long long)pl.f * pl.s0 + (long long)(tiles - pl.f) * pl.s1 : (pl.s1 > 1 ? (long long)(tiles - pl.f) * pl.s1 : 0);
    const long long slab_bytes = split_items * pl.t.bm * pl.t.bn * 4;
    if (slab_bytes >= 0x7ff00000LL) {
        pl.f = tiles;
        pl.s0 = pl.s1 = 1;
        pl.chunk0 = pl.chunk1 = even_steps(nk);
        pl.ws_bytes = 0;
    } else {
        pl.ws_bytes = split_items > 0 ? (size_t)Y3_WS_HEADER + (size_t)slab_bytes : 0;
    }
    return pl;
}
// fast_ok: the launch qualifies for conv_igemm_fast_kernel (the only kernel with split-K)
static ConvPlan plan_conv(int M, int Nout, int K, bool fast_ok, bool x3 = false, int ntaps = 1) {
    if (x3) return plan_conv_x3(M, Nout, K, ntaps);
    ConvPlan pl;
    pl.t = pick_tile(M, Nout);
    const int tiles = y3_cdiv(M, pl.t.bm) * y3_cdiv(Nout, pl.t.bn);
    const int nk = K / pl.t.bk;                                   // K steps (fast path: K % bk == 0)
    pl.tiles = tiles;
    pl.f = tiles;
    pl.s0 = pl.s1 = 1;
    pl.chunk0 = pl.chunk1 = nk > 0 ? nk : 1;
    static const int want = env_int("Y3_SPLITK_WGS", 2000);   // workgroups to aim for (swept with tools/fwd_time.py: 1000 / 1400 / 2000 / 2800)
    static const int min_k = env_int("Y3_SPLITK_MINK", 256);  // shortest K slice worth a launch
    static const int cus = env_int("Y3_CUS", 256);
    static const int rsplit_on = env_int("Y3_RSPLIT", 1);
    // measured (tools/fixed_cost.py, layer_times.py): a 676-tile launch (2.6 workgroups per CU) is better left whole unless
    // K is long; at <= 512 tiles the extra workgroups win over the slab round trip
    static const int few_max = env_int("Y3_FEWTILES", 512);
    const bool few_tiles = tiles <= few_max || ((long long)tiles * 2 <= want && K >= 2048);
    if (fast_ok && tiles <= Y3_MAX_TICKETS && few_tiles && K >= 2 * min_k) {
        int ks = (int)((want + tiles / 2) / tiles);
        const int maxs = K / min_k;
        if (ks > maxs) ks = maxs;
        if (ks > 16) ks = 16;
        if (ks > 1) {
            pl.chunk0 = even_steps(y3_cdiv(nk, ks));
            pl.s0 = y3_cdiv(nk, pl.chunk0);
        }
    } else if (fast_ok && rsplit_on && tiles <= Y3_MAX_TICKETS && tiles > cus && nk >= 8) {
        // whole rounds of tiles stay whole; the remainder round is cut along K so that its pieces spread evenly over the CUs:
        // load per CU = full rounds + ceil(R * S / CUs) / S tiles against tiles / CUs ideal
        const int F = tiles / cus * cus, R = tiles - F;
        if (R > 0) {
            int best = 1;
            double best_cost = 1.0;
            for (int S = 2; S <= 6 && S * 4 <= nk; ++S) {
                const double cost = (double)y3_cdiv((long long)R * S, cus) / S + 0.03 * (S - 1);
                if (cost < best_cost - 1e-9) {
                    best_cost = cost;
                    best = S;
                }
            }
            if (best > 1) {
                pl.f = F;
                pl.chunk1 = even_steps(y3_cdiv(nk, best));
                pl.s1 = y3_cdiv(nk, pl.chunk1);
            }
        }
    }
    pl.stats_tiles = y3_cdiv(M, pl.t.bm);
    const long long split_items = pl.s0 > 1 ? (long long)pl.f * pl.s0 + (long long)(tiles - pl.f) * pl.s1 : (pl.s1 > 1 ? (long long)(tiles - pl.f) * pl.s1 : 0);
    const long long slab_bytes = split_items * pl.t.bm * pl.t.bn * 4;
    if (slab_bytes >= 0x7ff00000LL) {   // slab offsets are 32-bit buffer offsets: fall back to whole tiles
        pl.f = tiles;
        pl.s0 = pl.s1 = 1;
        pl.chunk0 = pl.chunk1 = nk > 0 ? nk : 1;
        pl.ws_bytes = 0;
    } else {
        pl.ws_bytes = split_items > 0 ? (size_t)Y3_WS_HEADER + (size_t)slab_bytes : 0;
    }
    return pl;
}
// Debug switch of the product library (documented in yolo3hip.h): with Y3_CHECK_TICKETS=1 every launch that uses the ticket
// header first synchronises the stream and verifies that the header is all zero -- a workspace that was never zeroed (or was
// shared between two streams) otherwise shows up only as stale output: no slice ever draws the "last" ticket.
static int check_tickets(const char* what, const void* workspace, hipStream_t st) {
    static const int on = getenv("Y3_CHECK_TICKETS") ? atoi(getenv("Y3_CHECK_TICKETS")) : 0;
    if (!on || !workspace) return Y3_OK;
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;      // a stream under graph capture cannot be synchronised: no check there
    if (hipStreamIsCapturing(st, &capturing) != hipSuccess || capturing != hipStreamCaptureStatusNone) return Y3_OK;
    // read-back on the launch's own stream (pinned buffer, async copy, then synchronise)
    static int* host = nullptr;
    if (!host && hipHostMalloc((void**)&host, Y3_WS_HEADER, hipHostMallocDefault) != hipSuccess) host = nullptr;
    if (!host || hipMemcpyAsync(host, workspace, Y3_WS_HEADER, hipMemcpyDeviceToHost, st) != hipSuccess) return Y3_OK;
    if (hipStreamSynchronize(st) != hipSuccess) return Y3_OK;
    for (int i = 0; i < Y3_MAX_TICKETS; ++i)
        if (host[i] != 0) {
            y3_set_error("%s: ticket %d of the workspace header is %d, not 0: zero the workspace once after allocating it and keep its "
                         "launches on one stream (include/yolo3hip.h, workspace contract)", what, i, host[i]);
            return Y3_EINVAL;
        }
    return Y3_OK;
}
static bool fast_shape_ok(int C, int Nout, int K, int ntaps) {
    return !getenv("Y3_NO_FAST") && C % 16 == 0 && K % 16 == 0 && (ntaps == 1 || y3_is_pow2(C));
}

extern "C" int y3_conv2d_x3_ok(int m, int c, int ntaps, int nout) {
    return (m > 0 && (ntaps == 1 || ntaps == 9 || ntaps == 2 || ntaps == 4) && x3_shape_ok(c, nout, ntaps * c, ntaps)) ? 1 : 0;
}
extern "C" int y3_conv2d_stats_tiles_x(int m, int cin, int ksize, int cout, unsigned flags) {
    const int taps = ksize * ksize;
    const bool x3 = (flags & Y3_CONV_X3) && x3_shape_ok(cin, cout, taps * cin, taps);
    return plan_conv(m, cout, taps * cin, fast_shape_ok(cin, cout, taps * cin, taps), x3, taps).stats_tiles;
}
extern "C" size_t y3_conv2d_fwd_workspace_x(int m, int cin, int ksize, int cout, unsigned flags) {
    const int taps = ksize * ksize;
    const bool x3 = (flags & Y3_CONV_X3) && x3_shape_ok(cin, cout, taps * cin, taps);
    return plan_conv(m, cout, taps * cin, fast_shape_ok(cin, cout, taps * cin, taps), x3, taps).ws_bytes;
}
extern "C" int y3_conv2d_stats_tiles(int m, int cin, int ksize, int cout) { return y3_conv2d_stats_tiles_x(m, cin, ksize, cout, 0u); }
extern "C" size_t y3_conv2d_fwd_workspace(int m, int cin, int ksize, int cout) { return y3_conv2d_fwd_workspace_x(m, cin, ksize, cout, 0u); }

// Diagnostics (include/yolo3hip.h): the plan behind y3_conv2d_fwd / stride-1 y3_conv2d_dgrad for an M x cout x (ksize^2 cin) GEMM
extern "C" size_t y3_conv2d_plan(int m, int cin, int ksize, int cout, int* out13) { return y3_conv2d_plan_x(m, cin, ksize, cout, 0u, out13); }
extern "C" size_t y3_conv2d_plan_x(int m, int cin, int ksize, int cout, unsigned flags, int* out13) {
    const int taps = ksize * ksize, K = taps * cin;
    const bool x3 = (flags & Y3_CONV_X3) && x3_shape_ok(cin, cout, K, taps);
    const bool fast = x3 || fast_shape_ok(cin, cout, K, taps);
    const ConvPlan pl = plan_conv(m, cout, K, fast, x3, taps);
    if (out13) {
        const int v[13] = {pl.t.bm, pl.t.bn, pl.t.bk, pl.tiles, pl.f, pl.s0, pl.s1, pl.chunk0, pl.chunk1,
                           pl.f * pl.s0 + (pl.tiles - pl.f) * pl.s1, pl.stats_tiles, (fast ? 1 : 0) | (pl.short_last ? 2 : 0), K / pl.t.bk};
        for (int i = 0; i < 13; ++i) out13[i] = v[i];
    }
    return pl.ws_bytes;
}

template <int BM, int BN, int WM, int WN, int BK>
static void launch_cfg(const ConvArgs& p, int grid, hipStream_t st) {
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, BK>), dim3(grid), dim3(64 * WM * WN), 0, st, p);
}
template <int BM, int BN, int WM, int WN, int BK>
static void launch_fast(const FastArgs& p, bool dense, int grid, hipStream_t st) {
    const dim3 g(grid), b(64 * WM * WN);
    if (p.bn_a)      // launch_igemm has checked: dense destination
        hipLaunchKernelGGL((conv_igemm_fast_kernel<BM, BN, WM, WN, BK, true, true>), g, b, 0, st, p);
    else if (dense)
        hipLaunchKernelGGL((conv_igemm_fast_kernel<BM, BN, WM, WN, BK, true>), g, b, 0, st, p);
    else
        hipLaunchKernelGGL((conv_igemm_fast_kernel<BM, BN, WM, WN, BK, false>), g, b, 0, st, p);
}

// Build the fast kernel's arguments; false if the launch does not qualify.
static bool make_fast(const ConvArgs& a, int ntaps, int bk, FastArgs* f) {
    // Nout need not be a multiple of 4 (the detection heads): a weight-row load that runs past column Nout - 1 picks up the
    // head of the next row (zeros past the end of the buffer) into accumulator columns >= Nout, which the epilogue never stores
    if (a.C % bk != 0 || a.K % bk != 0) return false;
    if (ntaps > 1 && !y3_is_pow2(a.C)) return false;
    const long long src_elems = (long long)(a.M > 0 ? 1 : 0) * 0;  // (unused)
    (void)src_elems;
    FastArgs p = {};
    // bias the base pointer by the most negative tap offset so that scalar offsets stay non-negative
    int min_off = 0;
    int dh[9], dw[9];
    for (int t = 0; t < ntaps; ++t) {
        const int code = (int)((a.tap_dhdw >> (4 * t)) & 15ull);
        dh[t] = (code & 3) - 1;
        dw[t] = (code >> 2) - 1;
        const int off = (dh[t] * a.W + dw[t]) * a.src_ld;
        if (off < min_off) min_off = off;
    }
    const long long total = (long long)a.src_n * a.H * a.W * a.src_ld - min_off;
    const long long wtotal = (long long)(a.wt_rows) * a.Nout;
    if (total * 4 >= 0x7fffffffLL || wtotal * 4 >= 0x7fffffffLL) return false;
    p.src = a.src ? a.src + min_off : nullptr;      // (null in the dry runs of the *_tiles queries)
    p.src_bytes = (unsigned)(total * 4);
    p.wt = a.wt;
    p.Cper = a.C;
    p.wt_bytes = (unsigned)(wtotal * 4);
    if (a.x3) {      // three bf16 piece planes of the K-contiguous copy (y3_x3_split_weights): 6 bytes per element, same 2 GiB limit
        if (wtotal * 6 >= 0x7fffffffLL) return false;
        p.wt_bytes = (unsigned)(wtotal * 6);
        static const int mode = env_int("Y3_X3_MODE", 1);
        p.x3_mode = mode;
    }
    for (int t = 0; t < ntaps; ++t) {
        p.tap_dh[t] = dh[t];
        p.tap_dw[t] = dw[t];
        p.tap_off[t] = ((dh[t] * a.W + dw[t]) * a.src_ld - min_off) * 4;
        p.tap_wrow[t] = (int)((a.tap_wsel >> (4 * t)) & 15ull) * a.C;
    }
    p.ntaps = ntaps;
    static const int korder = env_int("Y3_KORDER", 2);      // development: 0 = taps outermost, 1 = 16-channel chunks outermost
    p.korder = ntaps > 1 ? (korder == 2 && a.C % 32 != 0 ? 1 : korder) : 0;
    p.dv_taps = y3_make_div(p.korder == 2 ? 2 * ntaps : ntaps);
    p.ohw = a.OH * a.OW;
    p.dv_ohw = y3_make_div(p.ohw);
    p.dv_ow = y3_make_div(a.OW);
    {
        // the tap list as a (rows x nx) grid: nx = length of the first run of equal dh
        int nx = 1;
        while (nx < ntaps && dh[nx] == dh[0]) ++nx;
        if (ntaps % nx != 0) return false;
        p.tg_nx = nx;
        p.tg_mul = nx == 1 ? 32 : (nx == 2 ? 16 : 11);
        p.tg_off0 = p.tap_off[0];
        p.tg_w0 = p.tap_wrow[0];
        p.tg_offx = nx > 1 ? p.tap_off[1] - p.tap_off[0] : 0;
        p.tg_wx = nx > 1 ? p.tap_wrow[1] - p.tap_wrow[0] : 0;
        p.tg_offy = ntaps > nx ? p.tap_off[nx] - p.tap_off[0] : 0;
        p.tg_wy = ntaps > nx ? p.tap_wrow[nx] - p.tap_wrow[0] : 0;
        if (nx > 3) return false;
        for (int t = 0; t < ntaps; ++t)
            if (p.tap_off[t] != p.tg_off0 + (t / nx) * p.tg_offy + (t % nx) * p.tg_offx ||
                p.tap_wrow[t] != p.tg_w0 + (t / nx) * p.tg_wy + (t % nx) * p.tg_wx)
                return false;   // not a grid: the generic kernel takes it
    }
    {
        const long long dpix = a.dense_dst ? (long long)a.M : (long long)a.src_n * a.DH * a.DW;
        const long long db = dpix * a.dst_ld * 4, rb = dpix * (long long)a.resid_ld * 4;
        if (db >= 0x7fffffffLL || rb >= 0x7fffffffLL) return false;
        p.dst_bytes = (unsigned)db;
        p.resid_bytes = (unsigned)rb;
    }
    p.dst = a.dst;
    p.bias = a.bias;
    p.scale = a.scale;
    p.shift = a.shift;
    p.resid = a.resid;
    p.stats = a.stats;
    p.bn_a = a.bn_a;
    p.bn_part = a.bn_part;
    p.bn_a_ld = a.bn_a_ld;
    p.bn_a_bytes = a.bn_a ? (unsigned)((a.dense_dst ? (long long)a.M : (long long)a.src_n * a.DH * a.DW) * a.bn_a_ld * 4) : 0u;
    p.bn_row0 = 0;
    p.H = a.H;
    p.W = a.W;
    p.logC = a.logC;
    p.cmask = a.cmask;
    p.src_ld = a.src_ld;
    p.OH = a.OH;
    p.OW = a.OW;
    p.sh = a.sh;
    p.sw = a.sw;
    p.DH = a.DH;
    p.DW = a.DW;
    p.dsh = a.dsh;
    p.dsw = a.dsw;
    p.doh = a.doh;
    p.dow = a.dow;
    p.dst_ld = a.dst_ld;
    p.resid_ld = a.resid_ld;
    p.Nout = a.Nout;
    p.K = a.K;
    p.M = a.M;
    p.flags = a.flags;
    p.alpha = a.alpha;
    *f = p;
    return true;
}

static int launch_igemm(const ConvArgs& a, void* workspace, size_t workspace_bytes, hipStream_t st) {
    ConvArgs p = a;
    const int ntaps = p.K / p.C;
    const bool x3 = p.x3 != 0;
    if (x3 && !x3_shape_ok(p.C, p.Nout, p.K, ntaps)) {
        y3_set_error("conv: Y3_CONV_X3 does not take this shape (C %d, Nout %d, K %d): ask y3_conv2d_x3_ok() first", p.C, p.Nout, p.K);
        return Y3_EINVAL;
    }
    const bool fast_ok = x3 || fast_shape_ok(p.C, p.Nout, p.K, ntaps);
    ConvPlan pl = plan_conv(p.M, p.Nout, p.K, fast_ok, x3, ntaps);
    if (pl.ws_bytes > 0 && (workspace == nullptr || workspace_bytes < pl.ws_bytes)) {  // no room for slabs: whole tiles
        pl.f = pl.tiles;
        pl.s0 = pl.s1 = 1;
        pl.chunk0 = pl.chunk1 = even_steps(p.K / pl.t.bk);
        pl.ws_bytes = 0;
    }
    const TileCfg t = pl.t;
    p.nbn = y3_cdiv(p.Nout, t.bn);
    const int tiles = y3_cdiv(p.M, t.bm) * p.nbn;
    const int key = t.bm * 10000 + t.bn * 10 + (t.bk == 32 ? 1 : 0);
    FastArgs f;
    if (fast_ok && make_fast(p, ntaps, t.bk, &f)) {
        f.nbn = p.nbn;
        f.nbm = y3_cdiv(p.M, t.bm);
        static const int colmajor_kb = env_int("Y3_COLMAJOR_KB", 2048);     // kernel matrix larger than this: column-major tile ids
        f.col_major = ((long long)p.K * p.Nout * 4 > (long long)colmajor_kb * 1024) ? 1 : 0;
        f.nb_fast = f.col_major ? f.nbm : f.nbn;
        f.dv_nb = y3_make_div(f.nb_fast);
        f.dv_s0 = y3_make_div(pl.s0);
        f.dv_s1 = y3_make_div(pl.s1);
        f.sk_f = pl.f;
        f.sk_s0 = pl.s0;
        f.sk_s1 = pl.s1;
        f.sk_n0 = pl.f * pl.s0;
        f.sk_chunk0 = pl.chunk0;
        f.sk_chunk1 = pl.chunk1;
        f.sk_slab0 = pl.s0 > 1 ? 0 : f.sk_n0;
        f.tickets = pl.ws_bytes ? (int*)workspace : nullptr;
        if (f.tickets)
            if (int e = check_tickets("conv2d", workspace, st)) return e;
        f.slab = pl.ws_bytes ? (float*)((char*)workspace + Y3_WS_HEADER) : nullptr;
        const int grid = f.sk_n0 + (tiles - pl.f) * pl.s1;
        const bool dense = p.dense_dst != 0;
        if (t.bk != 16) {
            y3_set_error("conv: the fast kernel is built for K steps of 16 (tile %dx%dx%d)", t.bm, t.bn, t.bk);
            return Y3_EINVAL;
        }
        if (p.bn_a && !dense) {
            y3_set_error("conv: BatchNorm-backward statistics need the dense fast kernel with K steps of 16");
            return Y3_EINVAL;
        }
        if (x3) {
            if (pl.short_last && pl.f == tiles) f.x3_mode |= 8;
            if (!y3_x3_launch(f, t.bm, t.bn, dense, grid, st)) {
                y3_set_error("conv: no x3 kernel for tile %dx%d", t.bm, t.bn);
                return Y3_EINVAL;
            }
            Y3_CHECK_LAUNCH("conv_x3");
            return Y3_OK;
        }
        switch (key) {
            case 128 * 10000 + 128 * 10 + 0: launch_fast<128, 128, 2, 2, 16>(f, dense, grid, st); break;
            case 128 * 10000 + 64 * 10 + 0: launch_fast<128, 64, 4, 1, 16>(f, dense, grid, st); break;
            case 128 * 10000 + 32 * 10 + 0: launch_fast<128, 32, 4, 1, 16>(f, dense, grid, st); break;
            case 64 * 10000 + 64 * 10 + 0: launch_fast<64, 64, 2, 2, 16>(f, dense, grid, st); break;
            case 64 * 10000 + 128 * 10 + 0: launch_fast<64, 128, 2, 2, 16>(f, dense, grid, st); break;
            default: y3_set_error("conv: no fast kernel for tile %dx%dx%d", t.bm, t.bn, t.bk); return Y3_EINVAL;
        }
        Y3_CHECK_LAUNCH("conv_igemm_fast");
        return Y3_OK;
    }
    if (p.bn_a || x3) {
        y3_set_error("conv: %s not available on the generic kernel (shape %d x %d x %d)", x3 ? "Y3_CONV_X3 is" : "BatchNorm-backward statistics are", p.M, p.Nout, p.K);
        return Y3_EINVAL;
    }
    const int grid = tiles;
    switch (key) {
        case 128 * 10000 + 128 * 10 + 0: launch_cfg<128, 128, 2, 2, 16>(p, grid, st); break;
        case 128 * 10000 + 64 * 10 + 0: launch_cfg<128, 64, 4, 1, 16>(p, grid, st); break;
        case 128 * 10000 + 32 * 10 + 0: launch_cfg<128, 32, 4, 1, 16>(p, grid, st); break;
        case 64 * 10000 + 64 * 10 + 0: launch_cfg<64, 64, 2, 2, 16>(p, grid, st); break;
        case 64 * 10000 + 128 * 10 + 0: launch_cfg<64, 128, 2, 2, 16>(p, grid, st); break;
        default:   // shapes only the fast kernel is instantiated for (forced tiles): the generic 64x64
            p.nbn = y3_cdiv(p.Nout, 64);
            launch_cfg<64, 64, 2, 2, 16>(p, y3_cdiv(p.M, 64) * p.nbn, st);
            break;
    }
    Y3_CHECK_LAUNCH("conv_igemm");
    return Y3_OK;
}

static int set_channels(int C, int taps, int* logC, int* cmask) {
    if (taps == 1) {
        *logC = 31;
        *cmask = 0x7fffffff;
        return 0;
    }
    Y3_CHECK_ARG(y3_is_pow2(C) && C >= 4, "3x3 conv needs power-of-two channels >= 4 (got %d)", C);
    *logC = y3_ilog2(C);
    *cmask = C - 1;
    return 0;
}

extern "C" int y3_conv2d_fwd(const y3_tensor* src, const float* wt, const float* bias, int ksize, int stride, const y3_tensor* dst,
                             unsigned flags, float alpha, const float* scale, const float* shift, const y3_tensor* resid, float* stats,
                             void* workspace, size_t workspace_bytes, y3_stream_t stream) {
    if (int e = check_tensor(src, "conv2d_fwd src")) return e;
    if (int e = check_tensor(dst, "conv2d_fwd dst")) return e;
    Y3_CHECK_ARG(wt, "conv2d_fwd: null weights");
    Y3_CHECK_ARG(ksize == 1 || ksize == 3, "conv2d_fwd: ksize %d unsupported", ksize);
    Y3_CHECK_ARG(stride == 1 || stride == 2, "conv2d_fwd: stride %d unsupported", stride);
    Y3_CHECK_ARG((src->c & 3) == 0, "conv2d_fwd: Cin=%d must be a multiple of 4", src->c);
    const int OH = (src->h + stride - 1) / stride, OW = (src->w + stride - 1) / stride;
    Y3_CHECK_ARG(dst->n == src->n && dst->h == OH && dst->w == OW, "conv2d_fwd: dst geometry %dx%dx%d != expected %dx%dx%d", dst->n,
                 dst->h, dst->w, src->n, OH, OW);
    Y3_CHECK_ARG((scale == nullptr) == (shift == nullptr), "conv2d_fwd: scale/shift must both be given");
    ConvArgs p = {};
    p.src = src->ptr;
    p.wt = wt;
    p.dst = dst->ptr;
    p.bias = bias;
    p.scale = scale;
    p.shift = shift;
    p.stats = stats;
    if (resid) {
        if (int e = check_tensor(resid, "conv2d_fwd resid")) return e;
        Y3_CHECK_ARG(resid->n == dst->n && resid->h == dst->h && resid->w == dst->w && resid->c == dst->c, "conv2d_fwd: resid geometry");
        p.resid = resid->ptr;
        p.resid_ld = resid->ld;
    }
    const int taps = ksize * ksize;
    if (int e = set_channels(src->c, taps, &p.logC, &p.cmask)) return e;
    const int pbh = y3_same_pad_before(src->h, ksize, stride), pbw = y3_same_pad_before(src->w, ksize, stride);
    for (int kh = 0; kh < ksize; ++kh)
        for (int kw = 0; kw < ksize; ++kw) {
            const int t = kh * ksize + kw;
            const int dh = kh - pbh, dw = kw - pbw;
            Y3_CHECK_ARG(dh >= -1 && dh <= 2 && dw >= -1 && dw <= 2, "conv2d_fwd: tap offset out of range");
            p.tap_dhdw |= (unsigned long long)((dh + 1) | ((dw + 1) << 2)) << (4 * t);
            p.tap_wsel |= (unsigned long long)t << (4 * t);
        }
    p.H = src->h;
    p.W = src->w;
    p.C = src->c;
    p.src_ld = src->ld;
    p.OH = OH;
    p.OW = OW;
    p.sh = p.sw = stride;
    p.DH = OH;
    p.DW = OW;
    p.dsh = p.dsw = 1;
    p.dst_ld = dst->ld;
    p.dense_dst = 1;
    p.Nout = dst->c;
    p.K = taps * src->c;
    p.M = src->n * OH * OW;
    p.flags = flags & ~Y3_CONV_X3;
    p.x3 = (flags & Y3_CONV_X3) ? 1 : 0;
    p.alpha = alpha;
    p.src_n = src->n;
    p.wt_rows = taps * src->c;
    return launch_igemm(p, workspace, workspace_bytes, (hipStream_t)stream);
}

static int conv2d_dgrad_impl(const y3_tensor* ddst, const float* wt_t, int ksize, int stride, const y3_tensor* dsrc, unsigned flags,
                             const y3_tensor* bn_a, float* bn_partials, void* workspace, size_t workspace_bytes, y3_stream_t stream,
                             int* dry_rows = nullptr, size_t* dry_ws = nullptr);
// shapes the x3 data gradient takes: stride 1 as the forward; stride 2 (3x3, the merged launch of the parity classes): >= 64 input
// channels of the layer (output columns of the GEMM), its output channels a power of two
static bool dgrad_x3(unsigned flags, const y3_tensor* ddst, int ksize, int stride, const y3_tensor* dsrc) {
    if (!(flags & Y3_CONV_X3) || !ddst || !dsrc) return false;
    if (stride == 1) return x3_shape_ok(ddst->c, dsrc->c, ksize * ksize * ddst->c, ksize * ksize);
    return stride == 2 && ksize == 3 && dsrc->c >= 64 && y3_is_pow2(ddst->c) && x3_shape_ok(ddst->c, dsrc->c, ddst->c, 1);
}
extern "C" int y3_conv2d_dgrad_x3_ok(const y3_tensor* ddst, int ksize, int stride, const y3_tensor* dsrc) {
    return dgrad_x3(Y3_CONV_X3, ddst, ksize, stride, dsrc) ? 1 : 0;
}

extern "C" size_t y3_conv2d_dgrad_workspace(const y3_tensor* ddst, int ksize, int stride, const y3_tensor* dsrc) {
    return y3_conv2d_dgrad_workspace_x(ddst, ksize, stride, dsrc, 0u);
}
extern "C" size_t y3_conv2d_dgrad_workspace_x(const y3_tensor* ddst, int ksize, int stride, const y3_tensor* dsrc, unsigned flags) {
    const int taps = ksize * ksize;
    if (stride == 1) {
        const int K = taps * ddst->c;
        const bool x3 = (flags & Y3_CONV_X3) && x3_shape_ok(ddst->c, dsrc->c, K, taps);
        return plan_conv(dsrc->n * dsrc->h * dsrc->w, dsrc->c, K, fast_shape_ok(ddst->c, dsrc->c, K, taps), x3, taps).ws_bytes;
    }
    if (dgrad_x3(flags, ddst, ksize, stride, dsrc)) {      // the merged x3 launch: slabs of all classes behind one ticket header
        int rows = 0;
        size_t ws = 0;
        if (conv2d_dgrad_impl(ddst, nullptr, ksize, 2, dsrc, Y3_CONV_X3, nullptr, nullptr, nullptr, 0, nullptr, &rows, &ws) == Y3_OK && rows > 0) return ws;
    }
    size_t best = 0;
    for (int nt = 1; nt <= 4; nt *= 2) {  // parity classes carry 1, 2, 2 and 4 taps of a 3x3 kernel
        const int K = nt * ddst->c;
        const int M = dsrc->n * ((dsrc->h + 1) / 2) * ((dsrc->w + 1) / 2);
        const size_t b = plan_conv(M, dsrc->c, K, fast_shape_ok(ddst->c, dsrc->c, K, nt)).ws_bytes;
        if (b > best) best = b;
    }
    return best;
}

// The merged stride-2 data gradient on the x3 kernels (conv_x3_multi_kernel).  The classes carry 1 / 2 / 2 / 4 taps, i.e. K steps
// in the ratio 1 : 2 : 2 : 4, and the x3 loop wants the launch to fill the 512 workgroup slots once: every class is cut along K
// into slices of about the same length L -- the smallest L for which the launch still fits the slots.
struct MultiX3Plan {
    int bn, tiles[4], s[4], chunk[4], grid, rows;
    size_t ws;
};
static bool plan_dgrad_multi_x3(const ConvArgs* cls, int ncls, MultiX3Plan* pl) {
    if (ncls < 2 || ncls > 4) return false;
    const int Nout = cls[0].Nout;
    if (Nout < 64) return false;
    pl->bn = Nout >= 128 ? 128 : 64;
    static const int slots = env_int("Y3_X3_SLOTS", 512);
    int steps[4], tsum = 0, smax = 0;
    long long total = 0;
    pl->rows = 0;
    for (int c = 0; c < ncls; ++c) {
        const int ntaps = cls[c].K / cls[c].C;
        if (cls[c].Nout != Nout || !x3_shape_ok(cls[c].C, Nout, cls[c].K, ntaps)) return false;
        pl->tiles[c] = y3_cdiv(cls[c].M, 128) * y3_cdiv(Nout, pl->bn);
        pl->rows += y3_cdiv(cls[c].M, 128);
        steps[c] = cls[c].K / 16;
        pl->s[c] = 1;
        pl->chunk[c] = even_steps(steps[c]);
        tsum += pl->tiles[c];
        total += (long long)pl->tiles[c] * steps[c];
        smax = steps[c] > smax ? steps[c] : smax;
    }
    if (tsum < slots && tsum <= Y3_MAX_TICKETS) {
        int L = even_steps(y3_cdiv(total, slots));
        if (L < 12) L = 12;
        for (; L < smax; L += 2) {
            long long g = 0;
            for (int c = 0; c < ncls; ++c) g += (long long)pl->tiles[c] * y3_cdiv(steps[c], L);
            if (g <= slots) break;
        }
        for (int c = 0; c < ncls; ++c) {
            const int sc = y3_cdiv(steps[c], L);
            if (sc > 1) {
                pl->chunk[c] = even_steps(y3_cdiv(steps[c], sc));
                pl->s[c] = y3_cdiv(steps[c], pl->chunk[c]);
            }
        }
    }
    pl->grid = 0;
    size_t slab = 0;
    for (int c = 0; c < ncls; ++c) {
        pl->grid += pl->tiles[c] * pl->s[c];
        if (pl->s[c] > 1) slab += (size_t)pl->tiles[c] * pl->s[c] * 128 * pl->bn * 4;
    }
    pl->ws = slab ? (size_t)Y3_WS_HEADER + slab : 0;
    return true;
}
static bool launch_dgrad_multi_x3(const ConvArgs* cls, int ncls, hipStream_t st, int* dry, size_t* dry_ws, void* workspace, size_t workspace_bytes) {
    MultiX3Plan pl;
    if (!plan_dgrad_multi_x3(cls, ncls, &pl)) return false;
    if (!dry && pl.ws > 0 && (workspace == nullptr || workspace_bytes < pl.ws)) {      // no room for slabs: whole tiles
        for (int c = 0; c < ncls; ++c) {
            pl.s[c] = 1;
            pl.chunk[c] = even_steps(cls[c].K / 16);
        }
        pl.ws = 0;
    }
    FastArgs4 m = {};
    int first = 0, rows = 0, tick = 0;
    size_t slab_floats = 0;
    for (int c = 0; c < ncls; ++c) {
        const int ntaps = cls[c].K / cls[c].C;
        if (!make_fast(cls[c], ntaps, 16, &m.a[c])) return false;
        FastArgs& a = m.a[c];
        a.nbn = y3_cdiv(cls[c].Nout, pl.bn);
        a.nbm = y3_cdiv(cls[c].M, 128);
        a.col_major = 0;
        a.nb_fast = a.nbn;
        a.dv_nb = y3_make_div(a.nbn);
        a.dv_s0 = a.dv_s1 = y3_make_div(pl.s[c]);
        a.sk_f = pl.tiles[c];
        a.sk_s0 = a.sk_s1 = pl.s[c];
        a.sk_n0 = pl.tiles[c] * pl.s[c];
        a.sk_chunk0 = a.sk_chunk1 = pl.chunk[c];
        a.sk_slab0 = 0;
        const bool split = pl.s[c] > 1 && pl.ws > 0;
        const bool have = split && workspace != nullptr;      // (dry runs plan without a workspace)
        a.tickets = have ? (int*)workspace + tick : nullptr;
        a.slab = have ? (float*)((char*)workspace + Y3_WS_HEADER) + slab_floats : nullptr;
        if (split) {
            tick += pl.tiles[c];
            slab_floats += (size_t)pl.tiles[c] * pl.s[c] * 128 * pl.bn;
        }
        a.bn_row0 = rows;
        rows += a.nbm;
        m.first[c] = first;
        first += pl.tiles[c] * pl.s[c];
    }
    for (int c = ncls; c <= 4; ++c) m.first[c] = first;
    if (dry) {
        *dry = rows;
        if (dry_ws) *dry_ws = pl.ws;
        return true;
    }
    if (pl.ws > 0)
        if (check_tickets("conv2d_dgrad (stride 2, x3)", workspace, st) != Y3_OK) return false;
    return y3_x3_multi_launch(m, pl.bn, cls[0].bn_a != nullptr, first, st);
}

// One launch for all parity classes of a stride-2 data gradient; false if the shapes do not qualify (the caller then
// launches the classes one by one).
// dry != nullptr: nothing is launched, *dry receives the number of partial-statistics rows (row tiles over all classes)
static bool launch_dgrad_multi(const ConvArgs* cls, int ncls, hipStream_t st, int* dry = nullptr, size_t* dry_ws = nullptr, void* workspace = nullptr,
                              size_t workspace_bytes = 0) {
    static const int off = env_int("Y3_NO_DGRAD_MULTI", 0);
    if (off || ncls < 2 || ncls > 4) return false;
    if (cls[0].x3) return launch_dgrad_multi_x3(cls, ncls, st, dry, dry_ws, workspace, workspace_bytes);
    if (dry_ws) *dry_ws = 0;
    FastArgs4 m = {};
    int mmax = 0;
    for (int c = 0; c < ncls; ++c) mmax = cls[c].M > mmax ? cls[c].M : mmax;
    const int ntaps_max = cls[0].K / cls[0].C;
    if (!fast_shape_ok(cls[0].C, cls[0].Nout, cls[0].K, ntaps_max)) return false;
    TileCfg t = pick_tile(mmax * ncls, cls[0].Nout);   // the classes share one grid: size the tile for their sum
    if (t.bm == 128 && t.bn == 128) t.bm = 64;   // the 128x128 instantiation of the merged kernel spills (four argument sets live)
    if (t.bk != 16) return false;
    int first = 0, rows = 0;
    for (int c = 0; c < ncls; ++c) {
        const int ntaps = cls[c].K / cls[c].C;
        if (!fast_shape_ok(cls[c].C, cls[c].Nout, cls[c].K, ntaps) || !make_fast(cls[c], ntaps, t.bk, &m.a[c])) return false;
        m.a[c].nbn = y3_cdiv(cls[c].Nout, t.bn);
        m.a[c].nbm = y3_cdiv(cls[c].M, t.bm);
        m.a[c].col_major = 0;
        m.a[c].nb_fast = m.a[c].nbn;
        m.a[c].dv_nb = y3_make_div(m.a[c].nbn);
        m.a[c].dv_s0 = m.a[c].dv_s1 = y3_make_div(1);
        m.a[c].sk_f = m.a[c].sk_n0 = y3_cdiv(cls[c].M, t.bm) * m.a[c].nbn;   // whole tiles only
        m.a[c].sk_s0 = m.a[c].sk_s1 = 1;
        m.a[c].sk_chunk0 = m.a[c].sk_chunk1 = cls[c].K / t.bk;
        m.a[c].sk_slab0 = 0;
        m.a[c].slab = nullptr;
        m.a[c].tickets = nullptr;
        m.a[c].bn_row0 = rows;
        rows += m.a[c].nbm;
        m.first[c] = first;
        first += y3_cdiv(cls[c].M, t.bm) * m.a[c].nbn;
    }
    for (int c = ncls; c <= 4; ++c) m.first[c] = first;
    const int key = t.bm * 1000 + t.bn;
    if (dry) {
        *dry = rows;
        return key == 128 * 1000 + 64 || key == 128 * 1000 + 32 || key == 64 * 1000 + 64 || key == 64 * 1000 + 128;
    }
    if (cls[0].bn_a) {
        switch (key) {
            case 128 * 1000 + 64: hipLaunchKernelGGL((conv_igemm_fast_multi_kernel<128, 64, 4, 1, 16, true>), dim3(first), dim3(256), 0, st, m); break;
            case 128 * 1000 + 32: hipLaunchKernelGGL((conv_igemm_fast_multi_kernel<128, 32, 4, 1, 16, true>), dim3(first), dim3(256), 0, st, m); break;
            case 64 * 1000 + 64: hipLaunchKernelGGL((conv_igemm_fast_multi_kernel<64, 64, 2, 2, 16, true>), dim3(first), dim3(256), 0, st, m); break;
            case 64 * 1000 + 128: hipLaunchKernelGGL((conv_igemm_fast_multi_kernel<64, 128, 2, 2, 16, true>), dim3(first), dim3(256), 0, st, m); break;
            default: return false;
        }
        return true;
    }
    switch (key) {
        case 128 * 1000 + 64: hipLaunchKernelGGL((conv_igemm_fast_multi_kernel<128, 64, 4, 1, 16>), dim3(first), dim3(256), 0, st, m); break;
        case 128 * 1000 + 32: hipLaunchKernelGGL((conv_igemm_fast_multi_kernel<128, 32, 4, 1, 16>), dim3(first), dim3(256), 0, st, m); break;
        case 64 * 1000 + 64: hipLaunchKernelGGL((conv_igemm_fast_multi_kernel<64, 64, 2, 2, 16>), dim3(first), dim3(256), 0, st, m); break;
        case 64 * 1000 + 128: hipLaunchKernelGGL((conv_igemm_fast_multi_kernel<64, 128, 2, 2, 16>), dim3(first), dim3(256), 0, st, m); break;
        default: return false;
    }
    return true;
}

extern "C" int y3_conv2d_dgrad(const y3_tensor* ddst, const float* wt_t, int ksize, int stride, const y3_tensor* dsrc, unsigned flags,
                               void* workspace, size_t workspace_bytes, y3_stream_t stream) {
    return conv2d_dgrad_impl(ddst, wt_t, ksize, stride, dsrc, flags, nullptr, nullptr, workspace, workspace_bytes, stream);
}

// Row tiles of the partial statistics y3_conv2d_dgrad_bn writes for this shape, 0 if the shape does not qualify (stride 2,
// channel counts off the fast path): the caller then runs y3_bn_bwd_stats on the finished gradient instead.
extern "C" int y3_conv2d_dgrad_bn_tiles(const y3_tensor* ddst, int ksize, int stride, const y3_tensor* dsrc) {
    return y3_conv2d_dgrad_bn_tiles_x(ddst, ksize, stride, dsrc, 0u);
}
extern "C" int y3_conv2d_dgrad_bn_tiles_x(const y3_tensor* ddst, int ksize, int stride, const y3_tensor* dsrc, unsigned flags) {
    if (!ddst || !dsrc || (ksize != 1 && ksize != 3)) return 0;
    if (stride == 2) {      // the merged launch of the four parity classes: rows of all classes, or 0 if it would not be taken
        if (ksize != 3 || ddst->h != (dsrc->h + 1) / 2 || ddst->w != (dsrc->w + 1) / 2 || ddst->n != dsrc->n) return 0;
        int rows = 0;
        const unsigned x3f = dgrad_x3(flags, ddst, ksize, stride, dsrc) ? Y3_CONV_X3 : 0u;
        if (conv2d_dgrad_impl(ddst, nullptr, ksize, 2, dsrc, x3f, nullptr, nullptr, nullptr, 0, nullptr, &rows) != Y3_OK) return 0;
        return rows;
    }
    if (stride != 1) return 0;
    if (ddst->h != dsrc->h || ddst->w != dsrc->w || ddst->n != dsrc->n) return 0;
    const int taps = ksize * ksize, K = taps * ddst->c, M = dsrc->n * dsrc->h * dsrc->w;
    const bool x3 = dgrad_x3(flags, ddst, ksize, stride, dsrc);
    if (!x3 && !fast_shape_ok(ddst->c, dsrc->c, K, taps)) return 0;
    const ConvPlan pl = plan_conv(M, dsrc->c, K, true, x3, taps);
    if (pl.t.bk != 16) return 0;
    // the launch itself must be accepted too (2 GiB buffer limits, tap grid): dry run of the argument builder
    int ok = 0;
    if (conv2d_dgrad_impl(ddst, nullptr, ksize, 1, dsrc, x3 ? Y3_CONV_X3 : 0u, nullptr, nullptr, nullptr, 0, nullptr, &ok) != Y3_OK || !ok) return 0;
    return y3_cdiv(M, pl.t.bm);
}

extern "C" int y3_conv2d_dgrad_bn(const y3_tensor* ddst, const float* wt_t, int ksize, int stride, const y3_tensor* dsrc, unsigned flags,
                                  const y3_tensor* bn_a, float* bn_partials, void* workspace, size_t workspace_bytes, y3_stream_t stream) {
    if (int e = check_tensor(bn_a, "conv2d_dgrad_bn bn_a")) return e;
    Y3_CHECK_ARG(bn_partials, "conv2d_dgrad_bn: null partials");
    Y3_CHECK_ARG(dsrc && bn_a->n == dsrc->n && bn_a->h == dsrc->h && bn_a->w == dsrc->w && bn_a->c == dsrc->c, "conv2d_dgrad_bn: bn_a must have dsrc's geometry");
    Y3_CHECK_ARG((long long)bn_a->n * bn_a->h * bn_a->w * bn_a->ld * 4 < 0x7fffffffLL, "conv2d_dgrad_bn: bn_a of 2 GiB or more");
    Y3_CHECK_ARG(y3_conv2d_dgrad_bn_tiles_x(ddst, ksize, stride, dsrc, flags) > 0, "conv2d_dgrad_bn: shape does not qualify (y3_conv2d_dgrad_bn_tiles() == 0)");
    return conv2d_dgrad_impl(ddst, wt_t, ksize, stride, dsrc, flags, bn_a, bn_partials, workspace, workspace_bytes, stream);
}

static int conv2d_dgrad_impl(const y3_tensor* ddst, const float* wt_t, int ksize, int stride, const y3_tensor* dsrc, unsigned flags,
                             const y3_tensor* bn_a, float* bn_partials, void* workspace, size_t workspace_bytes, y3_stream_t stream,
                             int* dry_rows, size_t* dry_ws) {
    if (!dry_rows) {      // dry run (y3_conv2d_dgrad_bn_tiles, stride 2): geometry only, pointers may be null
        if (int e = check_tensor(ddst, "conv2d_dgrad ddst")) return e;
        if (int e = check_tensor(dsrc, "conv2d_dgrad dsrc")) return e;
        Y3_CHECK_ARG(wt_t, "conv2d_dgrad: null weights");
    }
    Y3_CHECK_ARG(ksize == 1 || ksize == 3, "conv2d_dgrad: ksize %d unsupported", ksize);
    Y3_CHECK_ARG(stride == 1 || stride == 2, "conv2d_dgrad: stride %d unsupported", stride);
    const int OH = (dsrc->h + stride - 1) / stride, OW = (dsrc->w + stride - 1) / stride;
    Y3_CHECK_ARG(ddst->n == dsrc->n && ddst->h == OH && ddst->w == OW, "conv2d_dgrad: geometry mismatch");
    Y3_CHECK_ARG((flags & ~(Y3_EPI_ACCUM | Y3_CONV_X3)) == 0, "conv2d_dgrad: only Y3_EPI_ACCUM and Y3_CONV_X3 allowed");
    Y3_CHECK_ARG(!(flags & Y3_CONV_X3) || dgrad_x3(flags, ddst, ksize, stride, dsrc), "conv2d_dgrad: Y3_CONV_X3 does not take this shape (ask y3_conv2d_dgrad_x3_ok())");
    const int pbh = y3_same_pad_before(dsrc->h, ksize, stride), pbw = y3_same_pad_before(dsrc->w, ksize, stride);
    // the contraction runs over (tap, cout): channels of ddst
    ConvArgs base = {};
    base.src = ddst->ptr;
    base.wt = wt_t;
    base.dst = dsrc->ptr;
    base.H = ddst->h;
    base.W = ddst->w;
    base.C = ddst->c;
    base.src_ld = ddst->ld;
    base.dst_ld = dsrc->ld;
    base.Nout = dsrc->c;
    base.flags = flags & ~Y3_CONV_X3;
    base.x3 = (flags & Y3_CONV_X3) ? 1 : 0;
    base.DH = dsrc->h;
    base.DW = dsrc->w;
    base.sh = base.sw = 1;
    base.src_n = ddst->n;
    base.wt_rows = ksize * ksize * ddst->c;
    if (stride == 1) {
        ConvArgs p = base;
        const int taps = ksize * ksize;
        if (int e = set_channels(ddst->c, taps, &p.logC, &p.cmask)) return e;
        for (int kh = 0; kh < ksize; ++kh)
            for (int kw = 0; kw < ksize; ++kw) {
                const int t = kh * ksize + kw;
                const int dh = pbh - kh, dw = pbw - kw;  // dsrc[i] += ddst[i + pad - k] * w[k]
                p.tap_dhdw |= (unsigned long long)((dh + 1) | ((dw + 1) << 2)) << (4 * t);
                p.tap_wsel |= (unsigned long long)t << (4 * t);
            }
        p.OH = dsrc->h;
        p.OW = dsrc->w;
        p.dsh = p.dsw = 1;
        p.dense_dst = 1;
        p.K = taps * ddst->c;
        p.M = dsrc->n * p.OH * p.OW;
        if (dry_rows) {      // would launch_igemm take the fast kernel for this shape?  (pointers are not dereferenced)
            FastArgs f;
            *dry_rows = ((p.x3 || fast_shape_ok(p.C, p.Nout, p.K, taps)) && make_fast(p, taps, 16, &f)) ? 1 : 0;
            return Y3_OK;
        }
        if (bn_a) {
            p.bn_a = bn_a->ptr;
            p.bn_a_ld = bn_a->ld;
            p.bn_part = bn_partials;
        }
        return launch_igemm(p, workspace, workspace_bytes, (hipStream_t)stream);
    }
    // stride 2: forward out o reads in[2o + k - pad]; input pixel i = 2q + par receives from the taps with
    // (par + pad - k) even, at o = q + (par + pad - k)/2.  One launch per (row parity, col parity).
    ConvArgs cls[4];
    int ncls = 0;
    for (int ph = 0; ph < 2; ++ph)
        for (int pw = 0; pw < 2; ++pw) {
            ConvArgs p = base;
            int nt = 0;
            for (int kh = 0; kh < ksize; ++kh) {
                if ((ph + pbh - kh) & 1) continue;
                for (int kw = 0; kw < ksize; ++kw) {
                    if ((pw + pbw - kw) & 1) continue;
                    const int dh = (ph + pbh - kh) / 2, dw = (pw + pbw - kw) / 2;  // exact: numerator even (may be negative)
                    Y3_CHECK_ARG(dh >= -1 && dh <= 2 && dw >= -1 && dw <= 2, "conv2d_dgrad: tap offset out of range");
                    p.tap_dhdw |= (unsigned long long)((dh + 1) | ((dw + 1) << 2)) << (4 * nt);
                    p.tap_wsel |= (unsigned long long)(kh * ksize + kw) << (4 * nt);
                    ++nt;
                }
            }
            p.OH = (dsrc->h - ph + 1) / 2;
            p.OW = (dsrc->w - pw + 1) / 2;
            if (p.OH <= 0 || p.OW <= 0) continue;
            p.dsh = p.dsw = 2;
            p.doh = ph;
            p.dow = pw;
            p.dense_dst = 0;
            p.M = dsrc->n * p.OH * p.OW;
            if (nt == 0) {
                // no tap reaches this parity class (1x1 stride 2): gradient is zero there
                Y3_CHECK_ARG(false, "conv2d_dgrad: 1x1 stride-2 not supported");
            }
            if (int e = set_channels(ddst->c, nt == 1 ? 1 : 9, &p.logC, &p.cmask)) return e;
            if (nt == 1) {  // single tap: plain k = c (no power-of-two requirement)
                p.logC = 31;
                p.cmask = 0x7fffffff;
            }
            p.K = nt * ddst->c;
            if (bn_a) {
                p.bn_a = bn_a->ptr;
                p.bn_a_ld = bn_a->ld;
                p.bn_part = bn_partials;
            }
            cls[ncls++] = p;
        }
    // longest contraction first, so that the 4-tap workgroups of a merged launch start before the 1-tap ones
    for (int i = 1; i < ncls; ++i)
        for (int j = i; j > 0 && cls[j].K > cls[j - 1].K; --j) {
            const ConvArgs tmp = cls[j];
            cls[j] = cls[j - 1];
            cls[j - 1] = tmp;
        }
    if (dry_rows) {
        *dry_rows = 0;
        int rows = 0;
        if (launch_dgrad_multi(cls, ncls, (hipStream_t)stream, &rows, dry_ws)) *dry_rows = rows;
        return Y3_OK;
    }
    if (launch_dgrad_multi(cls, ncls, (hipStream_t)stream, nullptr, nullptr, workspace, workspace_bytes)) {
        Y3_CHECK_LAUNCH("conv_igemm_fast_multi");
        return Y3_OK;
    }
    Y3_CHECK_ARG(!bn_a, "conv2d_dgrad_bn: the merged stride-2 launch is not available for this shape");
    for (int c = 0; c < ncls; ++c)
        if (int e = launch_igemm(cls[c], workspace, workspace_bytes, (hipStream_t)stream)) return e;
    return Y3_OK;
}

// ---- wgrad ---------------------------------------------------------------
struct WgradPlan {
    int bkr, bn, splits, chunk, tiles;
};
static WgradPlan plan_wgrad(int K, int Nout, int M, int taps) {
    WgradPlan w;
    w.bkr = (K <= 64) ? 64 : 128;
    w.bn = (Nout <= 32) ? 32 : (Nout <= 64 ? 64 : 128);
    // Measured per shape (tools/conv_tune.py with Y3_WGRAD_TILE, batch 8 at 416^2): the 1x1 layers (8-11 K steps per split, slab
    // traffic as large as the operands) run 20-25 % faster on 64x64 tiles; the 3x3 layers with large kernel matrices (26x26 and
    // 13x13 grids: K*Nout >= 1M) 5-10 % faster on 128x64, the 104x104 layer (K = 576) 6 % faster on 64x128.
    static const int shape_rules = env_int("Y3_WGRAD_SHAPE_RULES", 1);
    if (shape_rules && w.bkr == 128 && w.bn == 128) {
        if (taps == 1) {
            w.bkr = 64;
            w.bn = 64;
        } else if ((long long)K * Nout >= (1 << 20)) {
            w.bn = 64;
        } else if (K <= 576) {
            w.bkr = 64;
        }
    }
    {
        // experiments: Y3_WGRAD_TILE=bkr,bn (64|128, 32|64|128) for the layers with K <= Y3_WGRAD_TILE_MAXK (default 1024)
        static const char* ov = env_str("Y3_WGRAD_TILE");
        static const int maxk = env_int("Y3_WGRAD_TILE_MAXK", 1024);
        int a = 0, b = 0;
        if (ov && K <= maxk && sscanf(ov, "%d,%d", &a, &b) == 2 && (a == 64 || a == 128) && (b == 32 || b == 64 || b == 128) && b <= ((Nout + 31) / 32) * 32) {
            w.bkr = a;
            w.bn = b;
        }
    }
    w.tiles = y3_cdiv(K, w.bkr) * y3_cdiv(Nout, w.bn);
    // aim at ~16 waves per CU overall (these launches are latency / HBM bound per workgroup), at least 128 pixels per split
    static const int want_waves = env_int("Y3_WGRAD_WAVES", 4096);
    const int waves_per_wg = (w.bkr == 64 && w.bn == 32) ? 2 : 4;
    int splits = y3_cdiv(want_waves, w.tiles * waves_per_wg);
    const int maxs = y3_cdiv(M, 128);
    if (splits > maxs) splits = maxs;
    if (splits < 1) splits = 1;
    if (splits < y3_cdiv(M, Y3_WG_TABLE - 32)) splits = y3_cdiv(M, Y3_WG_TABLE - 32);   // a split's pixels fit the kernel's LDS pixel table
    int chunk = y3_cdiv(M, splits);
    chunk = y3_cdiv(chunk, 32) * 32;                   // an even number of 16-pixel steps (the kernel runs its steps in pairs)
    w.splits = y3_cdiv(M, chunk);
    w.chunk = chunk;
    return w;
}

// The x3 kernel gradient (conv_x3.hip: conv_wgrad_x3_kernel): 128 x 128 tiles, two workgroups per CU (64 KB of LDS each), six
// steps of 16 pixels per loop iteration.  ~480 workgroups per launch -- one round of the 512 slots.
static bool wgrad_x3_shape_ok(int K, int Nout, int taps, int cin) {
    return cin % 4 == 0 && K >= 128 && Nout >= 128 && (taps == 1 || y3_is_pow2(cin));
}
static WgradPlan plan_wgrad_x3(int K, int Nout, int M) {
    WgradPlan w;
    w.bkr = 128;
    w.bn = 128;
    w.tiles = y3_cdiv(K, 128) * y3_cdiv(Nout, 128);
    static const int want = env_int("Y3_WGX3_WGS", 480);
    int splits = want / w.tiles;
    if (splits < 1) splits = 1;
    // more tiles than CUs (13x13 3x3 layers: 288): one pixel run per tile leaves most CUs with a lone workgroup; two runs are a
    // round and an eighth; three (864 workgroups of ~28 steps, reduced in the kernel) measured best: 98 -> 87 us
    if (w.tiles > 256 && splits < 3) splits = 3;
    const int maxs = y3_cdiv(M, 192);
    if (splits > maxs) splits = maxs;
    if (splits < y3_cdiv(M, Y3_WG_TABLE - 96)) splits = y3_cdiv(M, Y3_WG_TABLE - 96);       // a split's pixels fit the LDS pixel table
    int chunk = y3_cdiv(M, splits);
    chunk = y3_cdiv(chunk, 96) * 96;                  // whole loop iterations: six steps of 16 pixels
    w.splits = y3_cdiv(M, chunk);
    w.chunk = chunk;
    return w;
}

// splits <= Y3_WG_FANIN: the reduction runs inside the kernel (one level: tickets + fragment-order slabs behind the header);
// more splits: natural-layout slabs [split][K][Nout] + slab_reduce_kernel (measured: a multi-level in-kernel tree costs more
// than the streaming reduce when every split is only a few K steps long)
static bool wgrad_in_kernel(const WgradPlan& w) {
    static const int mode = env_int("Y3_WGRAD_INKERNEL", 1);
    return mode != 0 && w.splits > 1 && w.splits <= Y3_WG_FANIN && w.tiles <= Y3_MAX_TICKETS;
}
static size_t wgrad_ws_bytes(const WgradPlan& w, int K, int Nout) {
    if (w.splits <= 1) return 0;
    if (wgrad_in_kernel(w)) return (size_t)Y3_WS_HEADER + (size_t)w.tiles * w.splits * w.bkr * w.bn * sizeof(float);
    return (size_t)Y3_WS_HEADER + (size_t)w.splits * K * Nout * sizeof(float);
}

// Diagnostics (include/yolo3hip.h): the plan behind y3_conv2d_wgrad
extern "C" size_t y3_conv2d_wgrad_plan(int m, int cin, int ksize, int cout, int* out8) { return y3_conv2d_wgrad_plan_x(m, cin, ksize, cout, 0u, out8); }
extern "C" size_t y3_conv2d_wgrad_plan_x(int m, int cin, int ksize, int cout, unsigned flags, int* out8) {
    const int taps = ksize * ksize, K = taps * cin;
    const bool x3 = (flags & Y3_CONV_X3) && wgrad_x3_shape_ok(K, cout, taps, cin);
    const WgradPlan w = x3 ? plan_wgrad_x3(K, cout, m) : plan_wgrad(K, cout, m, taps);
    if (out8) {
        const int v[8] = {w.bkr, w.bn, w.splits, w.chunk, w.tiles, wgrad_in_kernel(w) ? 1 : 0,
                          (w.splits >= 32 ? y3_cdiv(w.splits, 8) * 8 : w.splits) * w.tiles, Y3_WG_TABLE};
        for (int i = 0; i < 8; ++i) out8[i] = v[i];
    }
    return wgrad_ws_bytes(w, K, cout);
}

extern "C" int y3_conv2d_wgrad_x3_ok(int m, int cin, int ksize, int cout) {
    return (m > 0 && wgrad_x3_shape_ok(ksize * ksize * cin, cout, ksize * ksize, cin)) ? 1 : 0;
}
extern "C" size_t y3_conv2d_wgrad_workspace_x(const y3_tensor* src, const y3_tensor* ddst, int ksize, int stride, unsigned flags) {
    (void)stride;
    const int K = ksize * ksize * src->c;
    const int M = ddst->n * ddst->h * ddst->w;
    const bool x3 = (flags & Y3_CONV_X3) && wgrad_x3_shape_ok(K, ddst->c, ksize * ksize, src->c);
    const WgradPlan w = x3 ? plan_wgrad_x3(K, ddst->c, M) : plan_wgrad(K, ddst->c, M, ksize * ksize);
    return wgrad_ws_bytes(w, K, ddst->c);
}
extern "C" size_t y3_conv2d_wgrad_workspace(const y3_tensor* src, const y3_tensor* ddst, int ksize, int stride) {
    return y3_conv2d_wgrad_workspace_x(src, ddst, ksize, stride, 0u);
}

extern "C" int y3_conv2d_wgrad(const y3_tensor* src, const y3_tensor* ddst, int ksize, int stride, float* dw, void* workspace,
                               size_t workspace_bytes, y3_stream_t stream) {
    return y3_conv2d_wgrad_x(src, ddst, ksize, stride, dw, 0u, workspace, workspace_bytes, stream);
}

extern "C" int y3_conv2d_wgrad_x(const y3_tensor* src, const y3_tensor* ddst, int ksize, int stride, float* dw, unsigned flags, void* workspace,
                                 size_t workspace_bytes, y3_stream_t stream) {
    if (int e = check_tensor(src, "conv2d_wgrad src")) return e;
    if (int e = check_tensor(ddst, "conv2d_wgrad ddst")) return e;
    Y3_CHECK_ARG(dw, "conv2d_wgrad: null dw");
    Y3_CHECK_ARG(ksize == 1 || ksize == 3, "conv2d_wgrad: ksize %d unsupported", ksize);
    Y3_CHECK_ARG(stride == 1 || stride == 2, "conv2d_wgrad: stride %d unsupported", stride);
    Y3_CHECK_ARG((src->c & 3) == 0, "conv2d_wgrad: Cin=%d must be a multiple of 4", src->c);
    const int OH = (src->h + stride - 1) / stride, OW = (src->w + stride - 1) / stride;
    Y3_CHECK_ARG(ddst->n == src->n && ddst->h == OH && ddst->w == OW, "conv2d_wgrad: geometry mismatch");
    WgradArgs p = {};
    const int taps = ksize * ksize;
    if (int e = set_channels(src->c, taps, &p.logC, &p.cmask)) return e;
    const int pbh = y3_same_pad_before(src->h, ksize, stride), pbw = y3_same_pad_before(src->w, ksize, stride);
    for (int kh = 0; kh < ksize; ++kh)
        for (int kw = 0; kw < ksize; ++kw) {
            const int t = kh * ksize + kw;
            p.tap_dhdw |= (unsigned long long)((kh - pbh + 1) | ((kw - pbw + 1) << 2)) << (4 * t);
        }
    p.src = src->ptr;
    p.ddst = ddst->ptr;
    p.H = src->h;
    p.W = src->w;
    p.C = src->c;
    p.src_ld = src->ld;
    p.OH = OH;
    p.OW = OW;
    p.sh = p.sw = stride;
    p.dd_ld = ddst->ld;
    p.Nout = ddst->c;
    p.K = taps * src->c;
    p.M = src->n * OH * OW;
    {
        const long long sb = (long long)src->n * src->h * src->w * src->ld * 4, db = (long long)p.M * ddst->ld * 4;
        Y3_CHECK_ARG(sb < 0x7fffffffLL && db < 0x7fffffffLL, "conv2d_wgrad: tensors of 2 GiB or more are not supported (32-bit buffer offsets)");
        p.src_bytes = (unsigned)sb;
        p.dd_bytes = (unsigned)db;
    }
    Y3_CHECK_ARG((flags & ~Y3_CONV_X3) == 0, "conv2d_wgrad: only Y3_CONV_X3 allowed in flags");
    const bool x3 = (flags & Y3_CONV_X3) != 0;
    Y3_CHECK_ARG(!x3 || wgrad_x3_shape_ok(p.K, p.Nout, taps, src->c), "conv2d_wgrad: Y3_CONV_X3 does not take this shape (ask y3_conv2d_wgrad_x3_ok())");
    const WgradPlan w = x3 ? plan_wgrad_x3(p.K, p.Nout, p.M) : plan_wgrad(p.K, p.Nout, p.M, taps);
    p.chunk = w.chunk;
    p.nbn = y3_cdiv(p.Nout, w.bn);
    p.ohw = OH * OW;
    p.ntaps = taps;
    p.dv_tiles = y3_make_div(w.tiles);
    p.dv_nbn = y3_make_div(p.nbn);
    p.dv_ohw = y3_make_div(p.ohw);
    p.dv_ow = y3_make_div(OW);
    const size_t need = wgrad_ws_bytes(w, p.K, p.Nout);
    Y3_CHECK_ARG(workspace_bytes >= need && (need == 0 || workspace), "conv2d_wgrad: workspace %zu < %zu", workspace_bytes, need);
    Y3_CHECK_ARG(need < 0x7ff00000ull, "conv2d_wgrad: slab area too large (%zu bytes)", need);
    const bool in_kernel = wgrad_in_kernel(w);
    p.tickets = in_kernel ? (int*)workspace : nullptr;
    if (p.tickets)
        if (int e = check_tickets("conv2d_wgrad", workspace, (hipStream_t)stream)) return e;
    p.out = w.splits > 1 ? (float*)((char*)workspace + Y3_WS_HEADER) : dw;
    p.dw = dw;
    hipStream_t st = (hipStream_t)stream;
    p.tiles = w.tiles;
    p.splits = w.splits;
    dim3 grid((unsigned)((w.splits >= 32 ? y3_cdiv(w.splits, 8) * 8 : w.splits) * w.tiles));
    // Unused dynamic LDS on top of the kernel's static 40 KB: the kernel gradients run on the second stream beside the
    // BatchNorm-backward kernels of the compute stream, and the 128x64 / 64x128 variants otherwise fill all
    // 160 KB of a CU with 4 workgroups -- the 12 KB bn_bwd_finalize_tiles kernel on the critical path then waits for one of
    // them to retire (20 us per launch in the overlapped step against 9.7 us alone).  8 KB of pad caps them at 3 workgroups
    // per CU (the occupancy their register budget aims at): step 18.21 -> 18.03 ms (tools/ab_wgrad_pad.sh, two rounds;
    // padding the 32 KB 64x64 variant as well: no further change).
    static const int pad40 = env_int("Y3_WGRAD_PAD40", 8192), pad32 = env_int("Y3_WGRAD_PAD32", 0);
    if (x3) {
        if (!y3_wgrad_x3_launch(p, w.bkr, w.bn, grid.x, st)) {
            y3_set_error("conv2d_wgrad: no x3 kernel for tile %dx%d", w.bkr, w.bn);
            return Y3_EINVAL;
        }
    } else if (w.bkr == 128 && w.bn == 128)
        hipLaunchKernelGGL((conv_wgrad_kernel<128, 128, 2, 2, 16>), grid, dim3(256), 0, st, p);
    else if (w.bkr == 128 && w.bn == 64)
        hipLaunchKernelGGL((conv_wgrad_kernel<128, 64, 4, 1, 16>), grid, dim3(256), pad40, st, p);
    else if (w.bkr == 128 && w.bn == 32)
        hipLaunchKernelGGL((conv_wgrad_kernel<128, 32, 4, 1, 16>), grid, dim3(256), 0, st, p);
    else if (w.bkr == 64 && w.bn == 128)
        hipLaunchKernelGGL((conv_wgrad_kernel<64, 128, 2, 2, 16>), grid, dim3(256), pad40, st, p);
    else if (w.bkr == 64 && w.bn == 64)
        hipLaunchKernelGGL((conv_wgrad_kernel<64, 64, 2, 2, 16>), grid, dim3(256), pad32, st, p);
    else
        hipLaunchKernelGGL((conv_wgrad_kernel<64, 32, 2, 1, 16>), grid, dim3(128), 0, st, p);
    Y3_CHECK_LAUNCH("conv_wgrad");
    if (w.splits > 1 && !in_kernel) {
        const long long count = (long long)p.K * p.Nout;  // K*Nout is a multiple of 4 (Cin % 4 == 0)
        hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((count + 63) / 64)), dim3(256), 0, st, (const float*)p.out, dw, count, w.splits);
        Y3_CHECK_LAUNCH("slab_reduce");
    }
    return Y3_OK;
}

extern "C" int y3_transpose_weights(const float* wt, float* wt_t, int taps, int cin, int cout, y3_stream_t stream) {
    Y3_CHECK_ARG(wt && wt_t && taps > 0 && cin > 0 && cout > 0, "transpose_weights: bad args");
    dim3 grid(y3_cdiv(cout, 32), y3_cdiv(cin, 32), taps);
    hipLaunchKernelGGL(transpose_weights_kernel, grid, dim3(256), 0, (hipStream_t)stream, wt, wt_t, cin, cout);
    Y3_CHECK_LAUNCH("transpose_weights");
    return Y3_OK;
}

extern "C" int y3_transpose_weights_batched(const float* params, float* params_t, const int* table_dev, int nlayers, int total_tiles,
                                            y3_stream_t stream) {
    Y3_CHECK_ARG(params && params_t && table_dev && nlayers > 0 && total_tiles > 0, "transpose_weights_batched: bad args");
    hipLaunchKernelGGL(transpose_weights_batched_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, params, params_t, table_dev, nlayers);
    Y3_CHECK_LAUNCH("transpose_weights_batched");
    return Y3_OK;
}
