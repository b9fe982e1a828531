// bf16 inference convolution (BASELINE config 5: tiled inference, "bf16 conv path + fp32 NMS").
//
// Same gather-GEMM as conv.hip's fast path, with bf16 operands and fp32 accumulation on
// v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate):
//   A: activations NHWC bf16 (K = (tap, c) contiguous per pixel)
//   B: kernels in the [tap][Cout][Cin] layout of the transposed-weight arena, converted to bf16 -- K-contiguous per
//      output channel, so BOTH operands are "rows of 32 bf16 = 64 bytes" and share one loader and one LDS image:
//      unpadded rows with XOR-swizzled 16-byte quads, fragment = one conflict-free ds_read_b128 (8 bf16).
// Epilogue: + bias -> leaky-relu -> folded BatchNorm affine -> + residual (bf16) -> bf16 (or fp32 for the heads).
// At 64 bytes of operands per 32 MFMA cycles this kernel is bound by the L2 -> LDS path, not by the matrix pipe.
#include "common.h"
#include <cstdlib>
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

struct Bf16Args {
    const u16* src;  // biased so every tap offset is >= 0
    const u16* wt;
    void* dst;
    const float* bias;
    const float* scale;
    const float* shift;
    const u16* resid;
    int tap_off[9];  // bytes
    int tap_wt[9];   // element offset of tap t in wt (t * Cout * Cin)
    int tap_dh[9], tap_dw[9];
    int tap_off0, tap_wt1;         // tap_off[0], tap_wt[1] (= Cout * Cin)
    int tg_nx, tg_offy, tg_offx;   // tap_off as an affine map of the tap grid (tap = ty * tg_nx + tx): no table lookup in the K loop
    unsigned src_bytes, wt_bytes, dst_bytes, resid_bytes;
    int ntaps;
    int H, W, C, logC, cmask, src_ld;
    int OH, OW, sh, sw;
    int dst_ld, resid_ld;
    int Nout, K, M;
    unsigned flags;
    float alpha;
    int nbn, out_f32, vec_ok;
    int ohw;
    Y3Div dv_nbn, dv_ohw, dv_ow;   // index decode without run-time divides (y3_make_div)
    // split-K of the small-M layers (SK instantiations only): block b is slice b % sk_splits of tile b / sk_splits; a slice runs
    // sk_chunk K steps; partial accumulators go through `slab` (fp32, fragment order) and the slice that draws the tile's last
    // ticket sums them in slice order and runs the epilogue (same hand-off as conv.hip's fast kernel)
    int sk_splits, sk_chunk;
    Y3Div dv_sk;
    float* slab;
    int* tickets;
};

#define Y3_OOB 0x80000000u
// buffer_load_dwordx4 ... lds.  The builtin needs a gfx950 target feature; in the HOST pass of the same compilation the
// template body would be invalid and clang then silently drops the kernel's launch stub (undefined __device_stub__ at load
// time), so the host pass sees an empty statement.
#if defined(__HIP_DEVICE_COMPILE__)
#define Y3_GLDS16(rsrc, lds_dst, voff, soff) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, lds_dst, 16, voff, soff, 0, 0)
#else
#define Y3_GLDS16(rsrc, lds_dst, voff, soff) ((void)(rsrc), (void)(lds_dst), (void)(voff), (void)(soff))
#endif

// Development instrumentation (tools/probe/bf16_timing.hip builds this file with -DY3_TIMING): per-workgroup timestamps
// of the kernel phases.  Compiled out of the product library.
#ifdef Y3_TIMING
__device__ unsigned long long* y3_timing_buf = nullptr;
#define Y3_TSTAMP(i)                                                                                               \
    do {                                                                                                           \
        if (y3_timing_buf && threadIdx.x == 0) y3_timing_buf[(size_t)blockIdx.x * 8 + (i)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define Y3_TSTAMP(i)
#endif

__device__ __forceinline__ float bf16_to_f32(u16 v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ u16 f32_to_bf16(float f) {  // round to nearest even; NaN stays NaN
    return __builtin_bit_cast(u16, (__bf16)f);
}

// largest multiple of tm that divides bm and fits cap rows
constexpr int stage_rows(int bm, int tm, int cap) {
    int best = tm;
    for (int r = tm; r <= bm; r += tm)
        if (bm % r == 0 && r <= cap) best = r;
    return best;
}

template <int BM, int BN, int WM, int WN, bool STAGED, int BK = 32, int NBUF = 3, bool SK = false>
__global__ __launch_bounds__(64 * WM * WN, (BM == 128 && BN == 128) ? 3 : 1) void conv_bf16_kernel(const Bf16Args p) {
    constexpr int THREADS = 64 * WM * WN;
    constexpr int Q = BK / 8;             // 16-byte quads per row and K step (BK = 32: 64-byte rows, 64: 128-byte rows)
    constexpr int LDR = BK;               // row pitch in u16: no padding -- the 16-byte quads of a row are XOR-swizzled instead:
    // quad q of row r lives in slot q ^ f(r), f(r) = (r >> 2) & 3 for 64-byte rows, (r >> 1) & 7 for 128-byte rows.  A
    // ds_write_b128 group (8 lanes) then covers whole rows = 32 distinct banks, and a ds_read_b128 group (16 lanes: same
    // quad of 16 rows, MI355X_MICROARCH.md LDS table) hits 16 distinct 16-byte bank slots: both conflict free.
    constexpr int SWZ_SHIFT = BK == 32 ? 2 : 1, SWZ_MASK = Q - 1;
    constexpr int TM = BM / WM, TN = BN / WN, MB = TM / 32, NB = TN / 32;
    constexpr int A_LOADS = BM * Q / THREADS, B_LOADS = (BN * Q + THREADS - 1) / THREADS;
    static_assert(BM * Q % THREADS == 0 && TM % 32 == 0 && TN % 32 == 0 && (BK == 32 || BK == 64), "tile shape");

    // ONE LDS block (a second __shared__ object beside an LDS-DMA staging array makes hipcc drain vmcnt before every
    // ds_read, cdna_hip_programming.md 5): a ring of NBUF operand stages {A rows | B rows} in the main loop, re-used as the
    // fp32 staging tile of the epilogue.  B holds at least one full wave-load of rows so that every wave issues the same
    // number of LDS-DMA loads per K step (the counted vmcnt below relies on it); surplus rows receive zeros.
    constexpr int BROWS = B_LOADS * THREADS / Q;          // >= BN
    constexpr int STG_U16 = (BM + BROWS) * LDR;            // one ring stage
    constexpr int STAGE_U16 = (BM / WM) * (BN + 4) * 2;    // one wave-row band of the fp32 staging tile of the epilogue
    constexpr int OPER_U16 = NBUF * STG_U16;
    __shared__ __attribute__((aligned(16))) u16 smem[OPER_U16 > STAGE_U16 ? OPER_U16 : STAGE_U16];

    Y3_TSTAMP(0);
#ifdef Y3_TIMING
    if (y3_timing_buf && threadIdx.x == 0) {
        y3_timing_buf[(size_t)blockIdx.x * 8 + 4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_ID
        y3_timing_buf[(size_t)blockIdx.x * 8 + 5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // XCC_ID
    }
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    // the scalars of the index decode, fetched from the argument segment in one batch (see conv_fast_body in conv.hip)
    int nbn = p.nbn, ohw = p.ohw, OW = p.OW, aM = p.M, aH = p.H, aW = p.W, src_ld = p.src_ld, csh = p.sh, csw = p.sw, ntaps = p.ntaps;
    unsigned dn_m = p.dv_nbn.mul, dohw_m = p.dv_ohw.mul, dow_m = p.dv_ow.mul;
    int dn_s = p.dv_nbn.shift, dohw_s = p.dv_ohw.shift, dow_s = p.dv_ow.shift;
    Y3_PIN_S(nbn); Y3_PIN_S(ohw); Y3_PIN_S(OW); Y3_PIN_S(aM); Y3_PIN_S(aH); Y3_PIN_S(aW); Y3_PIN_S(src_ld); Y3_PIN_S(csh); Y3_PIN_S(csw); Y3_PIN_S(ntaps);
    Y3_PIN_S(dn_m); Y3_PIN_S(dohw_m); Y3_PIN_S(dow_m); Y3_PIN_S(dn_s); Y3_PIN_S(dohw_s); Y3_PIN_S(dow_s);
    const Y3Div dv_nbn = {dn_m, dn_s}, dv_ohw = {dohw_m, dohw_s}, dv_ow = {dow_m, dow_s};
    const int bid_raw = y3_xcd_remap(blockIdx.x, gridDim.x);
    int bid = bid_raw, kz = 0;
    if constexpr (SK) {
        bid = y3_div(bid_raw, p.dv_sk);
        kz = bid_raw - bid * p.sk_splits;
    }
    const int bm = y3_div(bid, dv_nbn), bn = bid - bm * nbn;
    const int m0 = bm * BM, n0 = bn * BN;

    const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.src), 0, p.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wt = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.wt), 0, p.wt_bytes, 0x00020000);

    unsigned a_voff[A_LOADS], a_mask[A_LOADS];
    const int quad = tid & (Q - 1);
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
        const int row = (tid + i * THREADS) / Q;
        const int m = m0 + row;
        const bool ok = m < aM;
        const int mm = ok ? m : 0;
        const int n = y3_div(mm, dv_ohw);
        const int r = mm - n * ohw;
        const int oh = y3_div(r, dv_ow);
        const int ow = r - oh * OW;
        const int ih0 = oh * csh, iw0 = ow * csw;
        // LDS-DMA writes lane l of a wave to (wave-uniform base) + 16 * l: the LDS image is lane-linear, so the XOR swizzle of
        // the 16-byte quads is applied to the SOURCE address: the lane that fills slot `quad` of row `row` fetches quad
        // quad ^ f(row)
        const int qsrc = quad ^ ((row >> SWZ_SHIFT) & SWZ_MASK);
        a_voff[i] = (unsigned)(((n * aH + ih0) * aW + iw0) * src_ld + qsrc * 8) * 2u;
        unsigned msk = 0;
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
        const int row = idx / Q, n = n0 + row;
        const int qsrc = quad ^ ((row >> SWZ_SHIFT) & SWZ_MASK);
        b_voff[i] = (idx < BN * Q && n < p.Nout) ? (unsigned)(n * p.C + qsrc * 8) * 2u : Y3_OOB;
    }

    // Operand staging by LDS-DMA (buffer_load_dwordx4 ... lds): global -> LDS with no VGPR hop and no ds_write.  With 16x
    // the fp32 MFMA rate the register-staged version of this loop was bound by the LDS STORE path (4 ds_write_b128 per wave
    // and K step at ~13 cycles each against 256 cycles of MFMA, 3 workgroups per CU) and by the two register stages it needed
    // to cover an L2 round trip.  Here a K step is 2 + 2 DMA instructions per wave (1 KiB each: 16 consecutive tile rows,
    // lane l -> slot l), a ring of NBUF = 3 stages keeps two steps in flight, and out-of-range lanes (zero padding, tile
    // edges) point past the descriptor and deposit zeros (probed: tools/probe/lds_dma_probe.hip).
    auto glds = [&](int k0, int buf) {
        // tap -> offsets by arithmetic: indexing the argument-segment tables is a scalar MEMORY load in the loop, which forces
        // every later LDS wait to lgkmcnt(0) (see conv.hip)
        const int tap = k0 >> p.logC;
        const int cb = k0 & p.cmask;
        const int ty = p.tg_nx == 1 ? tap : (tap * 11) >> 5;      // tap / 3 for tap < 9
        const int tx = tap - ty * p.tg_nx;
        const unsigned a_soff = (unsigned)(p.tap_off0 + ty * p.tg_offy + tx * p.tg_offx + cb * 2);
        const unsigned b_soff = (unsigned)(tap * p.tap_wt1 + cb) * 2u;
        u16* stage = smem + buf * STG_U16;
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            u16* dst = stage + (wave * (64 / Q) + i * (THREADS / Q)) * LDR;     // wave-uniform: becomes M0
            Y3_GLDS16(rs_src, dst, ((a_mask[i] >> tap) & 1u) ? a_voff[i] : Y3_OOB, a_soff);
        }
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) {
            u16* dst = stage + (BM + wave * (64 / Q) + i * (THREADS / Q)) * LDR;
            Y3_GLDS16(rs_wt, dst, b_voff[i], b_soff);
        }
    };

    f32x16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk_all = p.K / BK;
    const int ks0 = SK ? kz * p.sk_chunk : 0;                                  // first K step of this slice
    const int nk = SK ? min(p.sk_chunk, nk_all - ks0) : nk_all;               // K steps of this slice (>= 1 by construction)
    auto compute = [&](int buf) {
        const u16* as = smem + buf * STG_U16 + (wm * TM + l31) * LDR;
        const u16* bs = smem + buf * STG_U16 + (BM + wn * TN + l31) * LDR;
        const int swz = (l31 >> SWZ_SHIFT) & SWZ_MASK;      // rows advance in multiples of 32 per fragment: f(row) = f(l31)
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            bf16x8 av[MB], bv[NB];
            const int qo = ((2 * kk + lh) ^ swz) * 8;
#pragma unroll
            for (int i = 0; i < MB; ++i) av[i] = *reinterpret_cast<const bf16x8*>(as + i * 32 * LDR + qo);
#pragma unroll
            for (int j = 0; j < NB; ++j) bv[j] = *reinterpret_cast<const bf16x8*>(bs + j * 32 * LDR + qo);
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    };
    // Every wave issues exactly L = A_LOADS + B_LOADS DMA instructions per step, steps past the end re-load the last step
    // into a stage nobody reads, so "all but the newest L are done" (s_waitcnt vmcnt(L)) always means "step ks has landed".
    // The barrier is a raw s_barrier + lgkmcnt(0): __syncthreads() would also drain vmcnt and with it the step in flight.
    constexpr int L = A_LOADS + B_LOADS, AHEAD = NBUF - 1;     // steps in flight
    static_assert(L * (AHEAD - 1) < 16, "vmcnt immediate");
    const int kfirst = ks0 * BK, klast = (ks0 + nk - 1) * BK;
#pragma unroll
    for (int a = 0; a < AHEAD; ++a) glds(min(kfirst + a * BK, klast), a);
    Y3_TSTAMP(1);
    int buf = 0, nbuf = AHEAD;
    for (int ks = 0; ks < nk; ++ks) {
        __builtin_amdgcn_s_waitcnt(0x0F70 | (L * (AHEAD - 1)));   // all but the newest AHEAD - 1 steps: this wave's part of step ks is in LDS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // this wave's fragment reads of step ks - 1 have returned
        __builtin_amdgcn_s_barrier();                       // ... and so have everybody's: stage (ks - 1) % NBUF is free
        glds(min(kfirst + (ks + AHEAD) * BK, klast), nbuf);
        compute(buf);
        buf = buf == NBUF - 1 ? 0 : buf + 1;
        nbuf = nbuf == NBUF - 1 ? 0 : nbuf + 1;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0): no DMA may land in the block the epilogue re-uses
    __syncthreads();

    if constexpr (SK) {
        // park the raw accumulators (slab[tile][slice][r4][thread], 16-byte sc1 stores), take the tile's ticket; the slice that
        // draws the last one re-reads all slices in slice order (bit-reproducible whichever it is) and goes on to the epilogue
        constexpr int R4 = MB * NB * 4;
        const __amdgpu_buffer_rsrc_t rs_slab = __builtin_amdgcn_make_buffer_rsrc(p.slab, 0, 0x7ffffff0, 0x00020000);
        const unsigned item_bytes = (unsigned)(R4 * THREADS * 16);
        const unsigned tile0 = (unsigned)(bid * p.sk_splits) * item_bytes + (unsigned)tid * 16u;
        {
            const unsigned base = tile0 + (unsigned)kz * item_bytes;
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        f32x4 v = {acc[i][j][4 * r], acc[i][j][4 * r + 1], acc[i][j][4 * r + 2], acc[i][j][4 * r + 3]};
                        __builtin_amdgcn_raw_buffer_store_b128(v, rs_slab, base, (unsigned)(((i * NB + j) * 4 + r) * THREADS * 16), 16 /* sc1 */);
                    }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* flag = reinterpret_cast<int*>(smem);
        if (tid == 0) {
            const int old = __hip_atomic_fetch_add(p.tickets + bid, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = old == p.sk_splits - 1;
            if (last) __hip_atomic_store(p.tickets + bid, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *flag = last;
        }
        __syncthreads();
        const int last = *flag;
        __syncthreads();          // smem is the epilogue's staging tile next
        if (!last) return;
#pragma unroll 1
        for (int z = 0; z < p.sk_splits; ++z) {
            const unsigned base = tile0 + (unsigned)z * item_bytes;
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const f32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_slab, base, (unsigned)(((i * NB + j) * 4 + r) * THREADS * 16), 16 /* sc1 */);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i][j][4 * r + e] = z == 0 ? v[e] : acc[i][j][4 * r + e] + v[e];
                    }
        }
    }

    Y3_TSTAMP(2);
    if constexpr (!STAGED) {
        // direct epilogue (no residual, 128x128 tiles): a lane stores its own channel of 16 x MB rows, 2 bytes at a time in
        // 64-byte runs; cheaper than staging when there is nothing to load and keeps the main loop at 3 waves / SIMD
        const bool do_lrelu = p.flags & Y3_EPI_LRELU;
        const bool has_scale = p.scale != nullptr;
        const __amdgpu_buffer_rsrc_t rs_dst = __builtin_amdgcn_make_buffer_rsrc(p.dst, 0, p.dst_bytes, 0x00020000);
        const int mrow = m0 + wm * TM + 4 * lh;
        const unsigned esz = p.out_f32 ? 4u : 2u;
        const unsigned ldb = (unsigned)p.dst_ld * esz;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int n = n0 + wn * TN + j * 32 + l31;
            const bool nok = n < p.Nout;
            const float bias = (p.bias && nok) ? p.bias[n] : 0.f;
            const float sc = (has_scale && nok) ? p.scale[n] : 1.f;
            const float sf = (has_scale && nok) ? p.shift[n] : 0.f;
            const unsigned vbase = (unsigned)mrow * ldb + (unsigned)n * esz;
#pragma unroll
            for (int i = 0; i < MB; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = i * 32 + (r & 3) + 8 * (r >> 2);
                    const bool ok = nok && mrow + dr < p.M;
                    float v = acc[i][j][r] + bias;
                    if (do_lrelu) v = v > 0.f ? v : p.alpha * v;
                    if (has_scale) v = v * sc + sf;
                    if (p.out_f32)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs_dst, ok ? vbase : Y3_OOB, (unsigned)dr * ldb, 0);
                    else
                        __builtin_amdgcn_raw_buffer_store_b16(f32_to_bf16(v), rs_dst, ok ? vbase : Y3_OOB, (unsigned)dr * ldb, 0);
                }
            }
        }
#ifdef Y3_TIMING
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        Y3_TSTAMP(3);
        return;
    }
    // ---- epilogue.  D layout: col = lane & 31 (channel), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
    // A lane owns single channels of scattered rows, so storing from registers would be 2-byte accesses in 64-byte
    // runs.  Instead the tile goes through LDS as fp32 (bias / lrelu / BN affine already applied) in passes of RPP rows,
    // and is read back 8 channels per thread: 16-byte residual loads, one rounding, 16-byte stores, full rows of the
    // tile contiguous in memory.
    constexpr int SLD = BN + 4;                                    // staged row pitch (floats)
    constexpr int CAP_ROWS = (int)(sizeof(smem) / (SLD * 4));
    constexpr int RPP = stage_rows(BM, TM, CAP_ROWS);              // rows per pass: multiple of TM, divides BM
    constexpr int PASSES = BM / RPP;
    constexpr int CG = BN / 8;                                     // 8-channel groups per row
    static_assert(RPP >= TM && BM % RPP == 0, "staging tile");
    float* stage = reinterpret_cast<float*>(smem);
    const bool do_lrelu = p.flags & Y3_EPI_LRELU;
    const bool has_scale = p.scale != nullptr, has_resid = p.resid != nullptr;
    float bias[NB], sc[NB], sf[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int n = n0 + wn * TN + j * 32 + l31;
        const bool nok = n < p.Nout;
        bias[j] = (p.bias && nok) ? p.bias[n] : 0.f;
        sc[j] = (has_scale && nok) ? p.scale[n] : 1.f;
        sf[j] = (has_scale && nok) ? p.shift[n] : 0.f;
    }
    const char* resid_b = reinterpret_cast<const char*>(p.resid);
    char* dst_b = reinterpret_cast<char*>(p.dst);
#pragma unroll 1
    for (int pass = 0; pass < PASSES; ++pass) {
        const int row0 = pass * RPP;                               // first tile row of this pass
        if (wm * TM >= row0 && wm * TM < row0 + RPP) {
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int i = 0; i < MB; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = acc[i][j][r] + bias[j];
                        if (do_lrelu) v = v > 0.f ? v : p.alpha * v;
                        if (has_scale) v = v * sc[j] + sf[j];
                        stage[(wm * TM - row0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * SLD + wn * TN + j * 32 + l31] = v;
                    }
        }
        __syncthreads();
        for (int idx = tid; idx < RPP * CG; idx += THREADS) {
            const int row = idx / CG, c8 = (idx - row * CG) * 8;
            const int m = m0 + row0 + row, n = n0 + c8;
            if (m >= p.M || n >= p.Nout) continue;
            const float* sp = stage + row * SLD + c8;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(sp), hi = *reinterpret_cast<const f32x4*>(sp + 4);
            float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            if (p.vec_ok && n + 8 <= p.Nout) {
                if (has_resid) {
                    const uint4 rv = *reinterpret_cast<const uint4*>(resid_b + ((size_t)m * p.resid_ld + n) * 2);
                    const unsigned rw[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[2 * e] += __uint_as_float(rw[e] << 16);
                        v[2 * e + 1] += __uint_as_float(rw[e] & 0xffff0000u);
                    }
                }
                if (p.out_f32) {
                    float* d = reinterpret_cast<float*>(dst_b + ((size_t)m * p.dst_ld + n) * 4);
                    *reinterpret_cast<f32x4*>(d) = f32x4{v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f32x4*>(d + 4) = f32x4{v[4], v[5], v[6], v[7]};
                } else {
                    unsigned pk[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) pk[e] = (unsigned)f32_to_bf16(v[2 * e]) | ((unsigned)f32_to_bf16(v[2 * e + 1]) << 16);
                    *reinterpret_cast<uint4*>(dst_b + ((size_t)m * p.dst_ld + n) * 2) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                }
            } else {   // ragged channel count (the detection heads) or unaligned pitches: element by element
                for (int e = 0; e < 8 && n + e < p.Nout; ++e) {
                    float x = v[e];
                    if (has_resid) x += bf16_to_f32(p.resid[(size_t)m * p.resid_ld + n + e]);
                    if (p.out_f32)
                        reinterpret_cast<float*>(dst_b)[(size_t)m * p.dst_ld + n + e] = x;
                    else
                        reinterpret_cast<u16*>(dst_b)[(size_t)m * p.dst_ld + n + e] = f32_to_bf16(x);
                }
            }
        }
        if (pass + 1 < PASSES) __syncthreads();
    }
#ifdef Y3_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    Y3_TSTAMP(3);
}

// ---------------------------------------------------------------------------
// 256 x 256 x 64 "ping-pong" kernel for the layers that carry the FLOPs of the tiled path (Cin % 64 == 0, Cout >= 256).
//
// Why another kernel: conv_bf16_kernel above is bound by operand delivery, not by the matrix pipe (MfmaUtil 14-22 %,
// DESIGN 3.4): a 128 x 128 tile moves 64 FLOP per operand byte and its loop alternates "everybody loads" / "everybody
// multiplies".  This one follows the 8-phase recipe of cdna_hip_programming.md 5 (256^2 tile = 128 FLOP / byte, 8 waves,
// one workgroup per CU, LDS-DMA with a counted vmcnt that never drains in the loop, raw s_barrier) with a schedule laid
// out for this gather-GEMM:
//
//  * waves 2 (pixels) x 4 (channels), each owns 128 pixels x 64 channels = 2 x 2 quadrants of 64 x 32 = 8 accumulator
//    tiles of v_mfma_f32_32x32x16_bf16 (128 registers).  The WEIGHTS are the MFMA's A operand (rows = output channels) and
//    the pixels its B operand (columns), so a lane ends up with 4-channel runs of ONE pixel: the epilogue stores from
//    registers (below).  32x32x16 rather than 16x16x32: it holds the SIMD's issue port for 8 of its 32 cycles instead of
//    8 of 16, and the loop lives on the partner wave issuing its loads in those gaps (measured with tools/probe/
//    bf16_pp_probe: with 16x16x32 the load and multiply sections of the two waves of a SIMD mostly serialised).
//    A K tile (64 deep) is four PHASES, one quadrant each, 8 MFMAs per phase:
//        P1 (m0, n0): reads pixels m0 (8 ds_read_b128) + weights n0 (4)      P2 (m0, n1): reads weights n1 (4)
//        P3 (m1, n1): reads pixels m1 (8)                                    P4 (m1, n0): reads nothing (n0 still in registers)
//  * a phase is  [LOAD: the phase's fragment reads, ONE 16 KB staging unit by LDS-DMA, s_waitcnt vmcnt(8)]  s_barrier
//    [8 MFMAs]  s_barrier.  Waves 4-7 run half a phase behind waves 0-3 (one extra barrier up front): on every SIMD one
//    wave multiplies while its partner loads.
//  * LDS: two K tiles x four units of 128 rows x 128 B: A0 / A1 = the m0 / m1 pixel blocks of BOTH wave rows, B0 / B1 = the
//    n0 / n1 channel blocks of all four wave columns -- a unit is exactly what ONE phase reads, so it can be refilled (for
//    the K tile after next) two phases after that phase: P3(t) stages A0(t+2), P4(t) B0(t+2), P1(t+1) B1(t+2), P2(t+1)
//    A1(t+2).  Every unit is issued 5-6 phases before its first read; vmcnt(8) = "all but the newest four units have
//    landed" is the only wait, placed one phase before the read with a barrier in between (RAW), and a unit is re-staged
//    >= 2 phases after its last read (WAR): the two rules of the guide's template, checked here for both wave groups.
//  * rows are 128 B with the 16-byte quads XOR-swizzled by (row >> 1) & 7 (applied to the SOURCE address of the DMA and to
//    the fragment reads): conflict-free ds_read_b128 for the 32-row fragments.
//  * K tiles are processed in pairs (static LDS addresses); an odd count is padded with one all-zero tile (DMA lanes
//    pointed out of range deposit zeros).
// Epilogue from registers: D[row = channel][column = pixel]: lane (pixel l & 31, half h = l >> 5) holds channels
// 8 g + 4 h + j (g, j = 0..3) of its pixel; v_permlane32_swap between the halves turns two 4-channel runs into one 8-channel
// run (cdna_hip_programming.md T21) -> bias / leaky-relu / BN affine in fp32, 16-byte residual loads, ONE rounding, 16-byte
// stores.  No LDS staging, no barrier after the K loop.
// ---------------------------------------------------------------------------
#ifdef Y3_TIMING
__device__ int y3_pp_abl = 0;       // probe only (tools/probe/bf16_pp_probe.hip): 1 no DMA in the loop, 2 no fragment reads, 4 no MFMAs, 8 no vmcnt wait
#define Y3_PPABL_INIT() const int pp_abl = __builtin_amdgcn_readfirstlane(y3_pp_abl)      /* read ONCE: a reload after every barrier would be timed too */
#define Y3_PPABL(bit) (pp_abl & (bit))
#else
#define Y3_PPABL_INIT()
#define Y3_PPABL(bit) 0
#endif
#define Y3_PP_UNIT 16384                  // 128 rows x 128 B
#define Y3_PP_TILE (4 * Y3_PP_UNIT)       // A0 | A1 | B0 | B1
#define Y3_PP_LDS (2 * Y3_PP_TILE)

#if defined(__HIP_DEVICE_COMPILE__)
#define Y3_SWAP32(lo, hi)                                                          \
    do {                                                                           \
        auto r_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo), __float_as_uint(hi), false, false); \
        lo = __uint_as_float(r_[0]);                                               \
        hi = __uint_as_float(r_[1]);                                               \
    } while (0)
#else
#define Y3_SWAP32(lo, hi) ((void)(lo), (void)(hi))
#endif

__global__ __launch_bounds__(512, 2) void conv_bf16_pp_kernel(const Bf16Args p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[Y3_PP_LDS];
    Y3_PPABL_INIT();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int l31 = lane & 31, lh = lane >> 5;
    int nbn = p.nbn, ohw = p.ohw, OW = p.OW, aM = p.M, aH = p.H, aW = p.W, src_ld = p.src_ld, csh = p.sh, csw = p.sw, ntaps = p.ntaps;
    unsigned dn_m = p.dv_nbn.mul, dohw_m = p.dv_ohw.mul, dow_m = p.dv_ow.mul;
    int dn_s = p.dv_nbn.shift, dohw_s = p.dv_ohw.shift, dow_s = p.dv_ow.shift;
    Y3_PIN_S(nbn); Y3_PIN_S(ohw); Y3_PIN_S(OW); Y3_PIN_S(aM); Y3_PIN_S(aH); Y3_PIN_S(aW); Y3_PIN_S(src_ld); Y3_PIN_S(csh); Y3_PIN_S(csw); Y3_PIN_S(ntaps);
    Y3_PIN_S(dn_m); Y3_PIN_S(dohw_m); Y3_PIN_S(dow_m); Y3_PIN_S(dn_s); Y3_PIN_S(dohw_s); Y3_PIN_S(dow_s);
    const Y3Div dv_nbn = {dn_m, dn_s}, dv_ohw = {dohw_m, dohw_s}, dv_ow = {dow_m, dow_s};
    const int bid = y3_xcd_remap(blockIdx.x, gridDim.x);
    const int bm = y3_div(bid, dv_nbn), bn = bid - bm * nbn;
    const int m0 = bm * 256, n0 = bn * 256;

    const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.src), 0, p.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wt = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.wt), 0, p.wt_bytes, 0x00020000);

    // ---- staging addresses.  Unit row R = (i * 8 + wave) * 8 + (lane >> 3), slot = lane & 7 (LDS-DMA: lane l lands at
    // wave base + 16 l); the lane fetches source quad slot ^ f(R).  A unit s: R < 64 -> tile pixel s*64 + R of wave row 0,
    // else 128 + s*64 + (R - 64); B unit s: tile channel (R >> 5) * 64 + s * 32 + (R & 31).
    unsigned a_voff[2][2], a_mask[2][2], b_voff[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int R = (i * 8 + wave) * 8 + (lane >> 3);
            const int qsrc = (lane & 7) ^ ((R >> 1) & 7);
            const int m = m0 + (R >> 6) * 128 + s * 64 + (R & 63);
            const bool ok = m < aM;
            const int mm = ok ? m : 0;
            const int n = y3_div(mm, dv_ohw);
            const int r = mm - n * ohw;
            const int oh = y3_div(r, dv_ow);
            const int ow = r - oh * OW;
            const int ih0 = oh * csh, iw0 = ow * csw;
            a_voff[s][i] = (unsigned)(((n * aH + ih0) * aW + iw0) * src_ld + qsrc * 8) * 2u;
            unsigned msk = 0;
            for (int t = 0; t < ntaps; ++t) {
                const int ih = ih0 + p.tap_dh[t], iw = iw0 + p.tap_dw[t];
                if (ok && (unsigned)ih < (unsigned)aH && (unsigned)iw < (unsigned)aW) msk |= 1u << t;
            }
            a_mask[s][i] = msk;
            const int nn = n0 + (R >> 5) * 64 + s * 32 + (R & 31);
            b_voff[s][i] = nn < p.Nout ? (unsigned)(nn * p.C + qsrc * 8) * 2u : Y3_OOB;
        }

    struct Ktile {      // scalars of one K tile (wave-uniform)
        unsigned a_soff, b_soff;
        int tap, valid;
    };
    const int nk = p.K >> 6;
    auto ktile = [&](int t) {
        Ktile k;
        const int k0 = min(t, nk - 1) << 6;
        k.tap = k0 >> p.logC;
        const int cb = k0 & p.cmask;
        const int ty = p.tg_nx == 1 ? k.tap : (k.tap * 11) >> 5;      // tap / 3 for tap < 9
        const int tx = k.tap - ty * p.tg_nx;
        k.a_soff = (unsigned)(p.tap_off0 + ty * p.tg_offy + tx * p.tg_offx + cb * 2);
        k.b_soff = (unsigned)(k.tap * p.tap_wt1 + cb) * 2u;
        k.valid = t < nk;
        return k;
    };
    // unit u of buffer b: 0 = A0, 1 = A1, 2 = B0, 3 = B1
    auto stage_a = [&](int buf, int s, const Ktile& k) {
        if (Y3_PPABL(1)) return;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            unsigned char* dst = smem + buf * Y3_PP_TILE + s * Y3_PP_UNIT + (i * 8 + wave) * 1024;
            Y3_GLDS16(rs_src, dst, (k.valid && ((a_mask[s][i] >> k.tap) & 1u)) ? a_voff[s][i] : Y3_OOB, k.a_soff);
        }
    };
    auto stage_b = [&](int buf, int s, const Ktile& k) {
        if (Y3_PPABL(1)) return;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            unsigned char* dst = smem + buf * Y3_PP_TILE + (2 + s) * Y3_PP_UNIT + (i * 8 + wave) * 1024;
            Y3_GLDS16(rs_wt, dst, k.valid ? b_voff[s][i] : Y3_OOB, k.b_soff);
        }
    };

    // ---- fragment addresses: 32 rows x (2 quads of one 16-deep K step); lane: row l31, quad 2 ks + lh
    const int swz = (l31 >> 1) & 7;
    int qo[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qo[ks] = ((2 * ks + lh) ^ swz) * 16;
    const unsigned char* a_lane = smem + (wr * 64 + l31) * 128;
    const unsigned char* b_lane = smem + (wc * 32 + l31) * 128;
    bf16x8 pf[2][4], wf[2][4];          // pixels: [pixel tile][k step]; weights: [n sub][k step]
    f32x16 acc[2][2][2];                // [m sub][pixel tile][n sub]: D[row = channel][column = pixel]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][c][r] = 0.f;

    auto read_a = [&](int buf, int s) {
        if (Y3_PPABL(2)) return;
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            const unsigned char* q = a_lane + buf * Y3_PP_TILE + s * Y3_PP_UNIT + pt * 4096;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) pf[pt][ks] = *reinterpret_cast<const bf16x8*>(q + qo[ks]);
        }
    };
    auto read_b = [&](int buf, int s) {
        if (Y3_PPABL(2)) return;
        const unsigned char* q = b_lane + buf * Y3_PP_TILE + (2 + s) * Y3_PP_UNIT;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wf[s][ks] = *reinterpret_cast<const bf16x8*>(q + qo[ks]);
    };
    auto mma = [&](auto SM, auto SN) {
        constexpr int sm = decltype(SM)::value, sn = decltype(SN)::value;
        if (Y3_PPABL(4)) return;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int pt = 0; pt < 2; ++pt)
                acc[sm][pt][sn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[sn][ks], pf[pt][ks], acc[sm][pt][sn], 0, 0, 0);
    };
    auto load_end = [&]() {       // end of a LOAD section: all but the newest four units have landed; the group barrier
        if (!Y3_PPABL(8)) __builtin_amdgcn_s_waitcnt(0x0F70 | 8);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto mma_end = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    // one K tile in buffer BUF; k1 = scalars of tile t + 1, k2 = of tile t + 2 (computed in P3)
    auto tile = [&](auto BUF, int t, Ktile& k1, Ktile& k2) {
        constexpr int buf = decltype(BUF)::value;
        // P1 (m0, n0)
        read_b(buf, 0);
        __builtin_amdgcn_sched_barrier(0);
        read_a(buf, 0);
        stage_b(buf ^ 1, 1, k1);
        load_end();
        mma(I0{}, I0{});
        mma_end();
        // P2 (m0, n1)
        read_b(buf, 1);
        stage_a(buf ^ 1, 1, k1);
        load_end();
        mma(I0{}, I1{});
        mma_end();
        // P3 (m1, n1)
        read_a(buf, 1);
        k2 = ktile(t + 2);
        stage_a(buf, 0, k2);
        load_end();
        mma(I1{}, I1{});
        mma_end();
        // P4 (m1, n0)
        stage_b(buf, 0, k2);
        load_end();
        mma(I1{}, I0{});
        mma_end();
        k1 = k2;
    };

    Y3_TSTAMP(0);
    // ---- prologue: tile 0 complete, A0 / B0 of tile 1 (the loop issues B1(1) in P1(0), A1(1) in P2(0), ...)
    Ktile k0 = ktile(0), k1 = ktile(1), k2 = k1;
    stage_a(0, 0, k0);
    stage_b(0, 0, k0);
    stage_b(0, 1, k0);
    stage_a(0, 1, k0);
    stage_a(1, 0, k1);
    stage_b(1, 0, k1);
    __builtin_amdgcn_s_waitcnt(0x0F70 | 8);      // A0(0), B0(0) have landed (this wave's share)
    __builtin_amdgcn_s_barrier();                // ... everybody's
    if (wr == 1) __builtin_amdgcn_s_barrier();   // waves 4-7 run half a phase behind
    Y3_TSTAMP(1);
    const int npairs = (nk + 1) >> 1;
    for (int it = 0; it < npairs; ++it) {
        tile(I0{}, 2 * it, k1, k2);
        tile(I1{}, 2 * it + 1, k1, k2);
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();   // barrier counts of the two groups match again
    Y3_TSTAMP(2);

    // ---- epilogue from registers (the DMA still in flight only ever targets LDS, which is not touched again).
    // Tile (sm, pt, sn): pixel = m0 + wr*128 + sm*64 + pt*32 + l31; register r = 4 g + j is channel cb + 8 g + 4 lh + j with
    // cb = n0 + wc*64 + sn*32.  Swapping (g even, lanes >= 32) <-> (g odd, lanes < 32) leaves lanes < 32 with the 8 channels
    // cb + 16 e .. + 7 and lanes >= 32 with cb + 16 e + 8 .. + 15 (e = g / 2) of their pixel.
    const bool do_lrelu = p.flags & Y3_EPI_LRELU;
    const bool has_scale = p.scale != nullptr, has_resid = p.resid != nullptr;
    const char* resid_b = reinterpret_cast<const char*>(p.resid);
    char* dst_b = reinterpret_cast<char*>(p.dst);
#pragma unroll
    for (int sn = 0; sn < 2; ++sn) {
        const int cb = n0 + wc * 64 + sn * 32;
        f32x4 bias[4], sc[4], sf[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = cb + 8 * g + 4 * lh;
            const bool cok = c < p.Nout;      // Nout % 8 == 0: a 4-channel run is inside or outside
            bias[g] = (p.bias && cok) ? *reinterpret_cast<const f32x4*>(p.bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
            sc[g] = (has_scale && cok) ? *reinterpret_cast<const f32x4*>(p.scale + c) : f32x4{1.f, 1.f, 1.f, 1.f};
            sf[g] = (has_scale && cok) ? *reinterpret_cast<const f32x4*>(p.shift + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int sm = 0; sm < 2; ++sm)
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                const int m = m0 + wr * 128 + sm * 64 + pt * 32 + l31;
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float x = acc[sm][pt][sn][r] + bias[r >> 2][r & 3];
                    if (do_lrelu) x = x > 0.f ? x : p.alpha * x;
                    if (has_scale) x = x * sc[r >> 2][r & 3] + sf[r >> 2][r & 3];
                    v[r] = x;
                }
#pragma unroll
                for (int e = 0; e < 2; ++e) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) Y3_SWAP32(v[8 * e + j], v[8 * e + 4 + j]);
                    // lanes < 32: v[8e .. 8e+3] = own channels cb+16e+0..3, v[8e+4 .. +7] = the upper half's cb+16e+4..7
                    // lanes >= 32: v[8e .. +3] = the lower half's cb+16e+8..11, v[8e+4 .. +7] = own cb+16e+12..15
                    const int n = cb + 16 * e + 8 * lh;
                    if (m < p.M && n < p.Nout) {
                        float* w = v + 8 * e;
                        if (has_resid) {
                            const uint4 rv = *reinterpret_cast<const uint4*>(resid_b + ((size_t)m * p.resid_ld + n) * 2);
                            const unsigned rw[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                w[2 * q] += __uint_as_float(rw[q] << 16);
                                w[2 * q + 1] += __uint_as_float(rw[q] & 0xffff0000u);
                            }
                        }
                        if (p.out_f32) {
                            float* d = reinterpret_cast<float*>(dst_b + ((size_t)m * p.dst_ld + n) * 4);
                            *reinterpret_cast<f32x4*>(d) = f32x4{w[0], w[1], w[2], w[3]};
                            *reinterpret_cast<f32x4*>(d + 4) = f32x4{w[4], w[5], w[6], w[7]};
                        } else {
                            unsigned pk[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) pk[q] = (unsigned)f32_to_bf16(w[2 * q]) | ((unsigned)f32_to_bf16(w[2 * q + 1]) << 16);
                            *reinterpret_cast<uint4*>(dst_b + ((size_t)m * p.dst_ld + n) * 2) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                        }
                    }
                }
            }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);          // the (zero-filling) DMA of the padded tail must not outlive the workgroup's LDS
#ifdef Y3_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    Y3_TSTAMP(3);
}

// ---------------------------------------------------------------------------
// The RGB layer (3x3, stride 1, Cin padded to 4, Cout = 32) on the matrix pipe.  (Rounds 1-2 ran it as a direct fp32 convolution
// on the vector ALU: 864 FMAs per pixel, 737 us for 45 tiles of 608^2 where the 1.06 GB of output needs 213 us.)  A workgroup walks down a 32-pixel column strip in groups of
// 4 output rows (one row per wave; the weights are split once per workgroup): the 6 x 34 input pixels of a group are loaded once (fp32, 4 channels = 16 bytes), split into THREE bf16 pieces (8 + 8 + 8
// mantissa bits: the split is exact) and parked in LDS as [piece][row][pixel][4 channels].  For one kernel row ky the K index
// is kx * 4 + c (16 deep: kx = 3 is padding, its weights are zero), so the B operand of v_mfma_f32_32x32x16_bf16 -- 8
// consecutive k of one pixel -- is the 16 contiguous bytes of two neighbouring pixels in LDS, and the weights (A operand:
// D = [channel][pixel], see conv_bf16_pp_kernel) stay in registers, split the same way.  Of the 9 piece products the 6 with
// piece indices summing to <= 2 are kept (the dropped ones are below 2^-24 of the product): fp32-accurate like the direct
// kernel, 18 MFMAs per wave.  Epilogue from registers: bias / leaky-relu / BN affine in fp32, v_permlane32_swap to 8-channel
// runs, one rounding, 16-byte stores (staging the block through LDS for 1 KB-contiguous stores was not faster: 480 vs 430 us).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void y3_split3(float x, u16& h, u16& m, u16& l) {
    h = f32_to_bf16(x);
    const float r1 = x - bf16_to_f32(h);
    m = f32_to_bf16(r1);
    l = f32_to_bf16(r1 - bf16_to_f32(m));
}

__global__ __launch_bounds__(256) void conv_first_bf16_mfma_kernel(const float* __restrict__ src, int src_ld, const float* __restrict__ wt,
                                                                   const float* __restrict__ bias, const float* __restrict__ scale,
                                                                   const float* __restrict__ shift, u16* __restrict__ dst, int dst_ld, int N,
                                                                   int H, int W, int xtiles, int rchunks, int groups, unsigned flags,
                                                                   float alpha, unsigned src_bytes) {
    __shared__ __attribute__((aligned(16))) u16 tile[2][3][6][36][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int tx = blockIdx.x % xtiles;
    const int t = blockIdx.x / xtiles;
    const int rc = t % rchunks, n = t / rchunks;
    const int x0 = tx * 32;
    const int ybeg = rc * (4 * groups), yend = min(H, ybeg + 4 * groups);      // `groups` row groups (of 4 rows) per workgroup
    const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, src_bytes, 0x00020000);
    // input staging: thread tid < 204 owns pixel (row tid / 34, column tid % 34) of the 6 x 34 window of every row group
    const int lr = tid / 34, lp = tid - lr * 34;
    const bool loader = tid < 6 * 34;
    const int ix = x0 - 1 + lp;
    auto gload = [&](int y0) {
        const int iy = y0 - 1 + lr;
        const bool ok = loader && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        const unsigned off = (unsigned)((((long long)n * H + iy) * W + ix) * src_ld * 4);
        return __builtin_amdgcn_raw_buffer_load_b128(rs_src, ok ? off : Y3_OOB, 0, 0);
    };
    auto lstore = [&](const f32x4& v, int buf) {
        if (loader) {
            u16 hh[4], mm[4], ll[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) y3_split3(v[c], hh[c], mm[c], ll[c]);
            *reinterpret_cast<uint2*>(&tile[buf][0][lr][lp][0]) = make_uint2((unsigned)hh[0] | ((unsigned)hh[1] << 16), (unsigned)hh[2] | ((unsigned)hh[3] << 16));
            *reinterpret_cast<uint2*>(&tile[buf][1][lr][lp][0]) = make_uint2((unsigned)mm[0] | ((unsigned)mm[1] << 16), (unsigned)mm[2] | ((unsigned)mm[3] << 16));
            *reinterpret_cast<uint2*>(&tile[buf][2][lr][lp][0]) = make_uint2((unsigned)ll[0] | ((unsigned)ll[1] << 16), (unsigned)ll[2] | ((unsigned)ll[3] << 16));
        }
    };
    f32x4 nxt = gload(ybeg);
    if (tid >= 6 * 34 && tid < 6 * 34 + 36) {
        // pixels 34 and 35 of every row and piece only ever meet zero weights, but 0 x NaN is NaN: clear them once (both buffers)
        const int i = tid - 6 * 34, pc = i / 12, r = (i % 12) >> 1, px = 34 + (i & 1);
        *reinterpret_cast<uint2*>(&tile[0][pc][r][px][0]) = make_uint2(0u, 0u);
        *reinterpret_cast<uint2*>(&tile[1][pc][r][px][0]) = make_uint2(0u, 0u);
    }
    // weights: A operand, lane = (channel l31, k block lh): k = kx * 4 + c for kx = 2 lh, 2 lh + 1 (kx = 3: zero)
    bf16x8 wf[3][3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        union {
            u16 u[8];
            bf16x8 v;
        } ch, cm, cl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int kx = 2 * lh + (e >> 2), c = e & 3;
            const float w = kx < 3 ? wt[((ky * 3 + kx) * 4 + c) * 32 + l31] : 0.f;
            y3_split3(w, ch.u[e], cm.u[e], cl.u[e]);
        }
        wf[ky][0] = ch.v;
        wf[ky][1] = cm.v;
        wf[ky][2] = cl.v;
    }
    const bool lrelu = flags & Y3_EPI_LRELU;
    f32x4 eb[4], es[4], ef[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int c = 8 * g + 4 * lh;
        eb[g] = bias ? *reinterpret_cast<const f32x4*>(bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        es[g] = scale ? *reinterpret_cast<const f32x4*>(scale + c) : f32x4{1.f, 1.f, 1.f, 1.f};
        ef[g] = scale ? *reinterpret_cast<const f32x4*>(shift + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    lstore(nxt, 0);
    __syncthreads();
    int buf = 0;
    for (int y0 = ybeg; y0 < yend; y0 += 4, buf ^= 1) {
        const bool more = y0 + 4 < yend;
        if (more) nxt = gload(y0 + 4);          // in flight under this group's MFMAs and stores
        f32x16 acc, acc1;                          // two chains: a dependent MFMA waits for its predecessor
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = acc1[r] = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int pc = 2; pc >= 0; --pc) {      // small pieces first
                union {
                    uint2 q[2];
                    bf16x8 v;
                } b;
                b.q[0] = *reinterpret_cast<const uint2*>(&tile[buf][pc][wave + ky][l31 + 2 * lh][0]);
                b.q[1] = *reinterpret_cast<const uint2*>(&tile[buf][pc][wave + ky][l31 + 2 * lh + 1][0]);
#pragma unroll
                for (int wp = 2 - pc; wp >= 0; --wp) {
                    if ((wp + pc + ky) & 1)
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ky][wp], b.v, acc1, 0, 0, 0);
                    else
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ky][wp], b.v, acc, 0, 0, 0);
                }
            }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += acc1[r];
        // epilogue: acc[r] = channel 8 (r / 4) + 4 lh + r % 4 of pixel (y0 + wave, x0 + l31)
        const int oy = y0 + wave, ox = x0 + l31;
        float v[16];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float x = acc[4 * g + j] + eb[g][j];
                if (lrelu) x = x > 0.f ? x : alpha * x;
                if (scale) x = x * es[g][j] + ef[g][j];
                v[4 * g + j] = x;
            }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
#pragma unroll
            for (int j = 0; j < 4; ++j) Y3_SWAP32(v[8 * e + j], v[8 * e + 4 + j]);
            // lanes < 32 now hold channels 16 e .. + 7 of their pixel, lanes >= 32 channels 16 e + 8 .. + 15
            if (oy < yend && ox < W) {
                const float* w = v + 8 * e;
                unsigned pk[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) pk[q] = (unsigned)f32_to_bf16(w[2 * q]) | ((unsigned)f32_to_bf16(w[2 * q + 1]) << 16);
                *reinterpret_cast<uint4*>(dst + (((long long)n * H + oy) * W + ox) * dst_ld + 16 * e + 8 * lh) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            }
        }
        if (more) {
            lstore(nxt, buf ^ 1);               // the other buffer was last read one iteration ago, before the barrier below
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------
// 3x3, Cin = 32 -> Cout = 64, stride 1 or 2: the 608^2 -> 304^2 stage of the tiled path, as a PATCH kernel.
//
// Why another kernel: as im2col rows of conv_bf16_kernel<128, 64> these two layers read
// every input pixel nine times through L2 -> LDS and ran 3-4 x above their HBM time (464 / 419 us for 45 tiles of 608^2 where
// the activations need 100-170 us).  Here a workgroup (4 waves) walks DOWN a strip of 32 output columns in groups of RPG
// output rows; the input patch of a group ((RPG-1) S + 3 rows x 31 S + 3 pixels x 32 channels = 64 B per pixel) is loaded
// ONCE into LDS -- the next group's patch is in flight in registers under the current group's MFMAs, two LDS buffers, one
// barrier per group -- and the nine taps are walked from LDS.  D = [channel][pixel] as in conv_bf16_pp_kernel: the weights are
// the MFMA's A operand and stay in REGISTERS for the whole launch (wave = one 32-channel block: 9 taps x 2 K halves = 18
// fragments = 72 VGPRs), the pixels its B operand: one ds_read_b128 per (patch row, kx, K half), used by every output row
// of the wave that sees this patch row through some ky (stride 1: up to three rows -> 36 reads for 72 MFMAs).
// Waves: (channel block cb = wave & 1) x (row set rs = wave >> 1, RW = RPG / 2 output rows each).
// LDS pixel layout: [row][plane][idx][4 slots of 16 B], slot ^= (idx >> 2) & 3.  Stride 1: one plane, idx = column; stride 2:
// even and odd input columns in two planes (idx = column / 2), so that the 32 pixels of a fragment read (columns 2 l + kx)
// are CONSECUTIVE idx and the reads are conflict-free for both strides; the plane pitch (34 pixels) puts the odd plane 128 B
// off in bank space, so the stores (4 slots x 4 neighbouring pixels per 16 lanes) are conflict-free too.
// The launch is persistent: min(groups, 2 per CU) workgroups take contiguous runs of the (image, strip, row group) list.
// Epilogue from registers: bias -> leaky-relu -> folded BN affine (constants parked in LDS) -> v_permlane32_swap to 8-channel
// runs -> + residual (16-byte loads) -> ONE rounding -> 16-byte stores.
// ---------------------------------------------------------------------------
struct PatchArgs {
    const u16* src;
    const u16* wt;
    u16* dst;
    const float* bias;
    const float* scale;
    const float* shift;
    const u16* resid;
    unsigned src_bytes, dst_bytes, resid_bytes;   // resid_bytes 0: no residual (every access out of range = 0)
    int H, W, OH, OW, src_ld, dst_ld, resid_ld;
    int pbh, pbw;            // SAME padding before (rows, columns)
    int xs, rg, groups;      // strips per image row, row groups per strip, groups in the launch (images x xs x rg)
    Y3Div dv_rg, dv_xs;
    unsigned flags;
    float alpha;
};

template <int S, bool RES>
__global__ __launch_bounds__(256, 2) void conv_bf16_c32_kernel(const PatchArgs p) {
    constexpr int RPG = S == 1 ? 8 : 4, RW = RPG / 2;          // output rows per group / per wave
    constexpr int IR = (RPG - 1) * S + 3, IC = 31 * S + 3;      // patch rows / columns
    constexpr int PLANEB = 34 * 64;
    constexpr int ROWB = S * PLANEB, BUFB = IR * ROWB;
    constexpr int QUADS = IR * IC * 4, NL = (QUADS + 255) / 256;
    constexpr int PL = (RW - 1) * S + 3;                        // patch rows one wave reads
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUFB + 3 * 64 * 4];
    float* epi = reinterpret_cast<float*>(smem + 2 * BUFB);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int cb = wave & 1, rs = wave >> 1;
    const int bid = y3_xcd_remap(blockIdx.x, gridDim.x);       // neighbouring runs (halo columns / rows in common) behind the same L2
    const int g_begin = (int)((long long)bid * p.groups / gridDim.x), g_end = (int)((long long)(bid + 1) * p.groups / gridDim.x);
    if (g_begin >= g_end) return;
    const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.src), 0, p.src_bytes, 0x00020000);
    const int aH = p.H, aW = p.W, src_ld = p.src_ld;

    // Staging map (a thread's share of a patch), chosen so that NOTHING of the quad decode is left in the group loop -- the first
    // version decoded q = tid + 256 i with constant divisions per load and group: with the predicates and the LDS address that was
    // ~40 vector instructions per load, and the stride-2 c32 launch was bound by the vector ALU (1 340 VALU instructions per wave and
    // group beside 36 MFMAs; `profiles/r04_pmc_bf16_patch_before_valu.txt`).  MAIN loads: stride 1: two patch rows x 32 pixels
    // per load (r2 = row parity of the thread), stride 2: one row x 64 pixels; load i covers rows RPL i (+ r2), so its global
    // offset is the thread's own offset + i x a scalar and its LDS address the thread's own + an immediate.  ONE extra load takes the
    // remaining 2 S^-1.. pixels per row (stride 1: columns 32, 33; stride 2: column 64) for all rows.
    constexpr int SL = 4;                                    // 16-byte slots per pixel
    constexpr int RPL = S == 1 ? 2 : 1, PXL = S == 1 ? 32 : 64, NM = IR / RPL, XP = IC - PXL;   // rows / pixels per main load, main loads, extra pixels per row
    static_assert(NM * RPL == IR && RPL * PXL * SL == 256 && NM + 1 == NL && IR * XP * SL <= 256, "staging map");
    const int m_r2 = S == 1 ? tid / (PXL * SL) : 0, m_px = (tid / SL) % PXL, m_slot = tid % SL;
    const int e_row = tid / (XP * SL), e_px = PXL + (tid / SL) % XP, e_slot = tid % SL;
    const bool e_live = tid < IR * XP * SL;
    auto lds_of = [&](int row, int px, int slot) {
        const int idx = S == 1 ? px : px >> 1, plane = S == 1 ? 0 : px & 1;
        return row * ROWB + plane * PLANEB + idx * 64 + ((slot ^ ((idx >> 2) & 3)) << 4);
    };
    const int m_lds = lds_of(m_r2, m_px, m_slot), e_lds = lds_of(e_live ? e_row : 0, e_px, e_slot);
    const int m_goff = ((m_r2 * aW + m_px) * src_ld + m_slot * 8) * 2, e_goff = ((e_row * aW + e_px) * src_ld + e_slot * 8) * 2;
    const int row_pitch = RPL * aW * src_ld * 2;               // bytes between the rows of consecutive main loads
    auto decode = [&](int g, int& img, int& oy0, int& ox0) {
        const int t = y3_div(g, p.dv_rg), rgi = g - t * p.rg;
        img = y3_div(t, p.dv_xs);
        oy0 = rgi * RPG;
        ox0 = (t - img * p.xs) * 32;
    };
    f32x4 nxt[NL];
    // every memory operation of the loop is UNCONDITIONAL (out-of-range offsets instead of branches): the compiler's vmcnt
    // bookkeeping is then exact and a wait for one load does not degrade to vmcnt(0), which would end the prefetch early
    auto gload = [&](int g, bool live) {
        int img, oy0, ox0;
        decode(g, img, oy0, ox0);
        const int iy0 = oy0 * S - p.pbh, ix0 = ox0 * S - p.pbw;
        const int base = ((img * aH + iy0) * aW + ix0) * src_ld * 2;   // may be negative; with a quad's own offset it is not, for a pixel inside the image
        const bool okx = live & ((unsigned)(ix0 + m_px) < (unsigned)aW);
        const int vbase = base + m_goff;
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            const bool ok = okx & ((unsigned)(iy0 + RPL * i + m_r2) < (unsigned)aH);
            nxt[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_src, ok ? (unsigned)(vbase + i * row_pitch) : Y3_OOB, 0, 0);
        }
        const bool eok = live & e_live & ((unsigned)(iy0 + e_row) < (unsigned)aH) & ((unsigned)(ix0 + e_px) < (unsigned)aW);
        nxt[NM] = __builtin_amdgcn_raw_buffer_load_b128(rs_src, eok ? (unsigned)(base + e_goff) : Y3_OOB, 0, 0);
    };
    auto lstore = [&](int buf) {
        unsigned char* mb = smem + buf * BUFB + m_lds;
#pragma unroll
        for (int i = 0; i < NM; ++i) *reinterpret_cast<f32x4*>(mb + i * RPL * ROWB) = nxt[i];
        if (e_live) *reinterpret_cast<f32x4*>(smem + buf * BUFB + e_lds) = nxt[NM];
    };
    gload(g_begin, true);

    // weights: A operand, lane = (channel cb * 32 + l31, k = lh * 8 .. + 7 of the 16-channel half kh)
    bf16x8 wf[3][3][2];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
                wf[ky][kx][kh] = *reinterpret_cast<const bf16x8*>(p.wt + ((ky * 3 + kx) * 64 + cb * 32 + l31) * 32 + kh * 16 + lh * 8);
    if (tid < 64) {
        epi[tid] = p.bias ? p.bias[tid] : 0.f;
        epi[64 + tid] = p.scale ? p.scale[tid] : 1.f;
        epi[128 + tid] = p.scale ? p.shift[tid] : 0.f;
    }
    // fragment reads: pixel column l31 * S + kx of patch row rs * 4 + pl, slot kh * 2 + lh
    int b_addr[3][2];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            const int idx = S == 1 ? l31 + kx : l31 + (kx >> 1), plane = S == 1 ? 0 : kx & 1, slot = kh * 2 + lh;
            b_addr[kx][kh] = rs * 4 * ROWB + plane * PLANEB + idx * 64 + ((slot ^ ((idx >> 2) & 3)) << 4);
        }
    const float alpha = (p.flags & Y3_EPI_LRELU) ? p.alpha : 1.f;   // flags folded into constants: no selects per value in the epilogue
    const __amdgpu_buffer_rsrc_t rs_dst = __builtin_amdgcn_make_buffer_rsrc(p.dst, 0, p.dst_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.resid ? p.resid : p.src), 0, p.resid_bytes, 0x00020000);
    const int aOH = p.OH, aOW = p.OW, dst_ld = p.dst_ld, resid_ld = p.resid_ld;

    lstore(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): the weights have landed -- no wait for them inside the loop
    __syncthreads();
    int buf = 0;
#pragma unroll 1
    for (int g = g_begin; g < g_end; ++g, buf ^= 1) {
        const bool more = g + 1 < g_end;
        gload(g + 1, more);                      // in flight under this group's MFMAs and stores
        __builtin_amdgcn_sched_barrier(0);
        int img, oy0, ox0;
        decode(g, img, oy0, ox0);
        f32x16 acc[RW];
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;
        const unsigned char* bb = smem + buf * BUFB;
        // the six fragments of patch row pl + 1 are read while the MFMAs of patch row pl issue (two register sets; the
        // sched_barrier keeps the compiler from hoisting ALL 6 PL reads to the top, which costs 100+ VGPRs and spills)
        bf16x8 bf[2][3][2];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) bf[0][kx][kh] = *reinterpret_cast<const bf16x8*>(bb + b_addr[kx][kh]);
#pragma unroll
        for (int pl = 0; pl < PL; ++pl) {
            if (pl + 1 < PL) {
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int kh = 0; kh < 2; ++kh) bf[(pl + 1) & 1][kx][kh] = *reinterpret_cast<const bf16x8*>(bb + b_addr[kx][kh] + (pl + 1) * ROWB);
            }
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                    for (int r = 0; r < RW; ++r) {
                        const int ky = pl - r * S;
                        if (ky >= 0 && ky < 3) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ky][kx][kh], bf[pl & 1][kx][kh], acc[r], 0, 0, 0);
                    }
            __builtin_amdgcn_sched_barrier(0);
        }
        // epilogue: acc[r][4 g + j] = channel cb * 32 + 8 g + 4 lh + j of pixel (oy0 + rs * RW + r, ox0 + l31).  The residual of
        // ALL rows is requested first (the fragment registers are free now): one exposed round trip per group, not one per store
        const int ox = ox0 + l31;
        unsigned o_dst[RW];
        f32x4 rv[RW][2];
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int oy = oy0 + rs * RW + r;
            const bool valid = oy < aOH && ox < aOW;
            const int m = (img * aOH + oy) * aOW + ox;
            o_dst[r] = valid ? (unsigned)(m * dst_ld + cb * 32 + 8 * lh) * 2u : Y3_OOB;
            const unsigned o_res = valid ? (unsigned)(m * resid_ld + cb * 32 + 8 * lh) * 2u : Y3_OOB;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                // without a residual (RES = false) there is NO load here: loads return in order, so even an out-of-range request
                // would make the first store wait for the patch of the next group as well
                if constexpr (RES)
                    rv[r][e] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, o_res, e * 32, 0);
                else
                    rv[r][e] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            float v[16];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int c = cb * 32 + 8 * gq + 4 * lh;           // the constants are re-read per row (broadcast reads): 48 registers otherwise
                const f32x4 eb = *reinterpret_cast<const f32x4*>(epi + c), es = *reinterpret_cast<const f32x4*>(epi + 64 + c),
                            ef = *reinterpret_cast<const f32x4*>(epi + 128 + c);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float x = acc[r][4 * gq + j] + eb[j];
                    x = __builtin_fmaxf(x, alpha * x);       // leaky-relu for 0 <= alpha <= 1 (the host checks); alpha = 1 without one
                    x = x * es[j] + ef[j];                   // scale 1, shift 0 without a folded BatchNorm
                    v[4 * gq + j] = x;
                }
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
#pragma unroll
                for (int j = 0; j < 4; ++j) Y3_SWAP32(v[8 * e + j], v[8 * e + 4 + j]);
                // lanes < 32 now hold channels cb * 32 + 16 e .. + 7 of their pixel, lanes >= 32 channels + 8 .. + 15
                float* w = v + 8 * e;
                const f32x4 rr = rv[r][e];
                unsigned pk[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned rw = __float_as_uint(rr[q]);
                    const float lo = RES ? w[2 * q] + __uint_as_float(rw << 16) : w[2 * q], hi = RES ? w[2 * q + 1] + __uint_as_float(rw & 0xffff0000u) : w[2 * q + 1];
                    pk[q] = (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16);
                }
                __builtin_amdgcn_raw_buffer_store_b128(f32x4{__uint_as_float(pk[0]), __uint_as_float(pk[1]), __uint_as_float(pk[2]), __uint_as_float(pk[3])},
                                                       rs_dst, o_dst[r], e * 32, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) {
            lstore(buf ^ 1);                     // the other buffer was last read one iteration ago, before the barrier below
            y3_lds_barrier();                    // LDS only: the epilogue's stores stay in flight across it
        }
    }
}

// ---------------------------------------------------------------------------
// The same patch scheme for 3x3, Cin = 64 -> Cout = 128 (the 304^2 -> 152^2 stage): 8 waves = 4 channel blocks (cb = wave & 3)
// x 2 K HALVES (kh2 = wave >> 2: input channels 32 kh2 .. + 31), so that a wave's weights are again 18 fragments = 72 VGPRs.
// Every wave multiplies all RPG rows of the group over its half of K; the two waves of a channel block then exchange halves
// of their accumulators through LDS (wave kh2 finishes rows kh2 * RPG / 2 ..: it parks the partial sums of the OTHER rows,
// one barrier, and adds its partner's) and run the register epilogue of their rows.  Pixels are 128 B in LDS (8 slots,
// slot ^= (idx >> 1) & 7: conflict-free fragment reads); one workgroup per CU (2 patch buffers + 64 KB exchange buffer).
// ---------------------------------------------------------------------------
template <int S>
__global__ __launch_bounds__(512) void conv_bf16_c64_kernel(const PatchArgs p) {
    constexpr int RPG = S == 1 ? 4 : 2, RF = RPG / 2;           // output rows per group / rows a wave finishes
    constexpr int IR = (RPG - 1) * S + 3, IC = 31 * S + 3;      // patch rows / columns
    constexpr int PLANEB = 34 * 128;
    constexpr int ROWB = S * PLANEB, BUFB = IR * ROWB;
    constexpr int QUADS = IR * IC * 8, NL = (QUADS + 511) / 512;
    constexpr int PL = IR;                                      // patch rows a wave reads: all of them
    constexpr int REDB = 8 * RF * 16 * 64 * 4;                  // exchange buffer: [wave][row][register][lane] floats
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUFB + REDB + 3 * 128 * 4];
    unsigned char* red = smem + 2 * BUFB;
    float* epi = reinterpret_cast<float*>(smem + 2 * BUFB + REDB);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int cb = wave & 3, kh2 = wave >> 2;
    const int bid = y3_xcd_remap(blockIdx.x, gridDim.x);       // neighbouring runs (halo columns / rows in common) behind the same L2
    const int g_begin = (int)((long long)bid * p.groups / gridDim.x), g_end = (int)((long long)(bid + 1) * p.groups / gridDim.x);
    if (g_begin >= g_end) return;
    const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.src), 0, p.src_bytes, 0x00020000);
    const int aH = p.H, aW = p.W, src_ld = p.src_ld;

    // Staging map (a thread's share of a patch), chosen so that NOTHING of the quad decode is left in the group loop -- the first
    // version decoded q = tid + 512 i with constant divisions per load and group: with the predicates and the LDS address that was
    // ~40 vector instructions per load, and the stride-2 c32 launch was bound by the vector ALU (1 340 VALU instructions per wave and
    // group beside 36 MFMAs; `profiles/r04_pmc_bf16_patch_before_valu.txt`).  MAIN loads: stride 1: two patch rows x 32 pixels
    // per load (r2 = row parity of the thread), stride 2: one row x 64 pixels; load i covers rows RPL i (+ r2), so its global
    // offset is the thread's own offset + i x a scalar and its LDS address the thread's own + an immediate.  ONE extra load takes the
    // remaining 2 S^-1.. pixels per row (stride 1: columns 32, 33; stride 2: column 64) for all rows.
    constexpr int SL = 8;                                    // 16-byte slots per pixel
    constexpr int RPL = S == 1 ? 2 : 1, PXL = S == 1 ? 32 : 64, NM = IR / RPL, XP = IC - PXL;   // rows / pixels per main load, main loads, extra pixels per row
    static_assert(NM * RPL == IR && RPL * PXL * SL == 512 && NM + 1 == NL && IR * XP * SL <= 512, "staging map");
    const int m_r2 = S == 1 ? tid / (PXL * SL) : 0, m_px = (tid / SL) % PXL, m_slot = tid % SL;
    const int e_row = tid / (XP * SL), e_px = PXL + (tid / SL) % XP, e_slot = tid % SL;
    const bool e_live = tid < IR * XP * SL;
    auto lds_of = [&](int row, int px, int slot) {
        const int idx = S == 1 ? px : px >> 1, plane = S == 1 ? 0 : px & 1;
        return row * ROWB + plane * PLANEB + idx * 128 + ((slot ^ ((idx >> 1) & 7)) << 4);
    };
    const int m_lds = lds_of(m_r2, m_px, m_slot), e_lds = lds_of(e_live ? e_row : 0, e_px, e_slot);
    const int m_goff = ((m_r2 * aW + m_px) * src_ld + m_slot * 8) * 2, e_goff = ((e_row * aW + e_px) * src_ld + e_slot * 8) * 2;
    const int row_pitch = RPL * aW * src_ld * 2;               // bytes between the rows of consecutive main loads
    auto decode = [&](int g, int& img, int& oy0, int& ox0) {
        const int t = y3_div(g, p.dv_rg), rgi = g - t * p.rg;
        img = y3_div(t, p.dv_xs);
        oy0 = rgi * RPG;
        ox0 = (t - img * p.xs) * 32;
    };
    f32x4 nxt[NL];
    // every memory operation of the loop is UNCONDITIONAL (out-of-range offsets instead of branches): the compiler's vmcnt
    // bookkeeping is then exact and a wait for one load does not degrade to vmcnt(0), which would end the prefetch early
    auto gload = [&](int g, bool live) {
        int img, oy0, ox0;
        decode(g, img, oy0, ox0);
        const int iy0 = oy0 * S - p.pbh, ix0 = ox0 * S - p.pbw;
        const int base = ((img * aH + iy0) * aW + ix0) * src_ld * 2;   // may be negative; with a quad's own offset it is not, for a pixel inside the image
        const bool okx = live & ((unsigned)(ix0 + m_px) < (unsigned)aW);
        const int vbase = base + m_goff;
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            const bool ok = okx & ((unsigned)(iy0 + RPL * i + m_r2) < (unsigned)aH);
            nxt[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_src, ok ? (unsigned)(vbase + i * row_pitch) : Y3_OOB, 0, 0);
        }
        const bool eok = live & e_live & ((unsigned)(iy0 + e_row) < (unsigned)aH) & ((unsigned)(ix0 + e_px) < (unsigned)aW);
        nxt[NM] = __builtin_amdgcn_raw_buffer_load_b128(rs_src, eok ? (unsigned)(base + e_goff) : Y3_OOB, 0, 0);
    };
    auto lstore = [&](int buf) {
        unsigned char* mb = smem + buf * BUFB + m_lds;
#pragma unroll
        for (int i = 0; i < NM; ++i) *reinterpret_cast<f32x4*>(mb + i * RPL * ROWB) = nxt[i];
        if (e_live) *reinterpret_cast<f32x4*>(smem + buf * BUFB + e_lds) = nxt[NM];
    };
    gload(g_begin, true);

    // weights: A operand, lane = (channel cb * 32 + l31, k = lh * 8 .. + 7 of the 16-channel step kq of the wave's K half)
    bf16x8 wf[3][3][2];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int kq = 0; kq < 2; ++kq)
                wf[ky][kx][kq] = *reinterpret_cast<const bf16x8*>(p.wt + ((ky * 3 + kx) * 128 + cb * 32 + l31) * 64 + kh2 * 32 + kq * 16 + lh * 8);
    if (tid < 128) {
        epi[tid] = p.bias ? p.bias[tid] : 0.f;
        epi[128 + tid] = p.scale ? p.scale[tid] : 1.f;
        epi[256 + tid] = p.scale ? p.shift[tid] : 0.f;
    }
    int b_addr[3][2];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int kq = 0; kq < 2; ++kq) {
            const int idx = S == 1 ? l31 + kx : l31 + (kx >> 1), plane = S == 1 ? 0 : kx & 1, slot = kh2 * 4 + kq * 2 + lh;
            b_addr[kx][kq] = plane * PLANEB + idx * 128 + ((slot ^ ((idx >> 1) & 7)) << 4);
        }
    const float alpha = (p.flags & Y3_EPI_LRELU) ? p.alpha : 1.f;   // flags folded into constants: no selects per value in the epilogue
    const __amdgpu_buffer_rsrc_t rs_dst = __builtin_amdgcn_make_buffer_rsrc(p.dst, 0, p.dst_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.resid ? p.resid : p.src), 0, p.resid_bytes, 0x00020000);
    const int aOH = p.OH, aOW = p.OW, dst_ld = p.dst_ld, resid_ld = p.resid_ld;

    lstore(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): the weights have landed
    __syncthreads();
    int buf = 0;
#pragma unroll 1
    for (int g = g_begin; g < g_end; ++g, buf ^= 1) {
        const bool more = g + 1 < g_end;
        gload(g + 1, more);
        __builtin_amdgcn_sched_barrier(0);
        int img, oy0, ox0;
        decode(g, img, oy0, ox0);
        f32x16 acc[RPG];
#pragma unroll
        for (int r = 0; r < RPG; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;
        const unsigned char* bb = smem + buf * BUFB;
        bf16x8 bf[2][3][2];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int kq = 0; kq < 2; ++kq) bf[0][kx][kq] = *reinterpret_cast<const bf16x8*>(bb + b_addr[kx][kq]);
#pragma unroll
        for (int pl = 0; pl < PL; ++pl) {
            if (pl + 1 < PL) {
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int kq = 0; kq < 2; ++kq) bf[(pl + 1) & 1][kx][kq] = *reinterpret_cast<const bf16x8*>(bb + b_addr[kx][kq] + (pl + 1) * ROWB);
            }
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int kq = 0; kq < 2; ++kq)
#pragma unroll
                    for (int r = 0; r < RPG; ++r) {
                        const int ky = pl - r * S;
                        if (ky >= 0 && ky < 3) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ky][kx][kq], bf[pl & 1][kx][kq], acc[r], 0, 0, 0);
                    }
            __builtin_amdgcn_sched_barrier(0);
        }
        // the residual of this wave's rows is requested before the exchange (its latency hides behind the barrier)
        const int ox = ox0 + l31;
        unsigned o_dst[RF];
        f32x4 rv[RF][2];
#pragma unroll
        for (int r = 0; r < RF; ++r) {
            const int oy = oy0 + kh2 * RF + r;
            const bool valid = oy < aOH && ox < aOW;
            const int m = (img * aOH + oy) * aOW + ox;
            o_dst[r] = valid ? (unsigned)(m * dst_ld + cb * 32 + 8 * lh) * 2u : Y3_OOB;
            const unsigned o_res = valid ? (unsigned)(m * resid_ld + cb * 32 + 8 * lh) * 2u : Y3_OOB;
#pragma unroll
            for (int e = 0; e < 2; ++e) rv[r][e] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, o_res, e * 32, 0);
        }
        // exchange: park the partial sums of the rows the PARTNER finishes (rows (1 - kh2) * RF ..), then add the partner's
        {
            unsigned char* mine = red + (wave * RF) * 4096 + lane * 16;
#pragma unroll
            for (int r = 0; r < RF; ++r)
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {                   // (a wave-uniform select between two register sets, not an indexed array)
                    const f32x4 v0 = f32x4{acc[RF + r][4 * e4], acc[RF + r][4 * e4 + 1], acc[RF + r][4 * e4 + 2], acc[RF + r][4 * e4 + 3]};
                    const f32x4 v1 = f32x4{acc[r][4 * e4], acc[r][4 * e4 + 1], acc[r][4 * e4 + 2], acc[r][4 * e4 + 3]};
                    *reinterpret_cast<f32x4*>(mine + r * 4096 + e4 * 1024) = kh2 == 0 ? v0 : v1;
                }
        }
        y3_lds_barrier();
        f32x16 fin[RF];
        {
            const unsigned char* theirs = red + ((wave ^ 4) * RF) * 4096 + lane * 16;
#pragma unroll
            for (int r = 0; r < RF; ++r)
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    const f32x4 o = *reinterpret_cast<const f32x4*>(theirs + r * 4096 + e4 * 1024);
#pragma unroll
                    for (int j = 0; j < 4; ++j) fin[r][4 * e4 + j] = (kh2 == 0 ? acc[r][4 * e4 + j] : acc[RF + r][4 * e4 + j]) + o[j];
                }
        }
        __builtin_amdgcn_sched_barrier(0);
        // epilogue: fin[r][4 g + j] = channel cb * 32 + 8 g + 4 lh + j of pixel (oy0 + kh2 * RF + r, ox0 + l31)
#pragma unroll
        for (int r = 0; r < RF; ++r) {
            float v[16];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int c = cb * 32 + 8 * gq + 4 * lh;
                const f32x4 eb = *reinterpret_cast<const f32x4*>(epi + c), es = *reinterpret_cast<const f32x4*>(epi + 128 + c),
                            ef = *reinterpret_cast<const f32x4*>(epi + 256 + c);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float x = fin[r][4 * gq + j] + eb[j];
                    x = __builtin_fmaxf(x, alpha * x);       // leaky-relu for 0 <= alpha <= 1 (the host checks); alpha = 1 without one
                    x = x * es[j] + ef[j];                   // scale 1, shift 0 without a folded BatchNorm
                    v[4 * gq + j] = x;
                }
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
#pragma unroll
                for (int j = 0; j < 4; ++j) Y3_SWAP32(v[8 * e + j], v[8 * e + 4 + j]);
                float* w = v + 8 * e;
                const f32x4 rr = rv[r][e];
                unsigned pk[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned rw = __float_as_uint(rr[q]);
                    const float lo = w[2 * q] + __uint_as_float(rw << 16), hi = w[2 * q + 1] + __uint_as_float(rw & 0xffff0000u);
                    pk[q] = (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16);
                }
                __builtin_amdgcn_raw_buffer_store_b128(f32x4{__uint_as_float(pk[0]), __uint_as_float(pk[1]), __uint_as_float(pk[2]), __uint_as_float(pk[3])},
                                                       rs_dst, o_dst[r], e * 32, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) lstore(buf ^ 1);
        y3_lds_barrier();                        // orders the patch buffers AND the exchange buffer of the next group
    }
}

// ---------------------------------------------------------------------------
// small bf16 helpers
// ---------------------------------------------------------------------------
__global__ void f32_to_bf16_kernel(const float* __restrict__ src, u16* __restrict__ dst, size_t count) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) dst[i] = f32_to_bf16(src[i]);
}

// all-ones transposed conv (model.py:94-105) on bf16: one wave per input pixel, fp32 sum over channels
__global__ __launch_bounds__(256) void upsample_bf16_kernel(const u16* __restrict__ in, int in_ld, int C, u16* __restrict__ out, int out_ld,
                                                            int outC, int N, int H, int W) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long npix = (long long)N * H * W;
    if (wave >= npix) return;
    const u16* src = in + wave * in_ld;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += bf16_to_f32(src[c]);
    s = y3_wave_sum(s);
    const u16 o = f32_to_bf16(s);
    const int n = (int)(wave / ((long long)H * W));
    const int r = (int)(wave - (long long)n * H * W);
    const int i = r / W, j = r - i * W;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            u16* dst = out + (((long long)n * 2 * H + 2 * i + a) * 2 * W + 2 * j + b) * out_ld;
            for (int c = lane; c < outC; c += 64) dst[c] = o;
        }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static int check_bf16_tensor(const y3_tensor* t, const char* name) {
    Y3_CHECK_ARG(t && t->ptr, "%s: null tensor", name);
    Y3_CHECK_ARG(t->n > 0 && t->h > 0 && t->w > 0 && t->c > 0 && t->ld >= t->c, "%s: bad dims", name);
    return 0;
}

// Experiment switches: read from the environment only in the developer build (make DEV=1); the product library uses the defaults.
static inline int dev_int(const char* name, int dflt) {
#ifdef Y3_DEV
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
#else
    (void)name;
    return dflt;
#endif
}

template <int BM, int BN, int WM, int WN, bool STAGED = true, int BK = 32>
static void launch_bf16(const Bf16Args& args, int grid, hipStream_t st) {
    Bf16Args p = args;
    p.ohw = p.OH * p.OW;
    p.dv_nbn = y3_make_div(p.nbn);
    p.dv_ohw = y3_make_div(p.ohw);
    p.dv_ow = y3_make_div(p.OW);
    static const int nbuf = dev_int("Y3_BF16_NBUF", 3);     // ring depth (experiments): 2, 3 or 4
    if constexpr (BK == 32 && BM * BN <= 128 * 128) {
        if (nbuf == 4) {
            hipLaunchKernelGGL((conv_bf16_kernel<BM, BN, WM, WN, STAGED, BK, 4>), dim3(grid), dim3(64 * WM * WN), 0, st, p);
            return;
        }
        if (nbuf == 2) {
            hipLaunchKernelGGL((conv_bf16_kernel<BM, BN, WM, WN, STAGED, BK, 2>), dim3(grid), dim3(64 * WM * WN), 0, st, p);
            return;
        }
    }
    hipLaunchKernelGGL((conv_bf16_kernel<BM, BN, WM, WN, STAGED, BK, 3>), dim3(grid), dim3(64 * WM * WN), 0, st, p);
}

// Split-K plan of the 64 x 64-tile path (the small-M layers: 13x13 / 26x26 grids at batch 8, 19x19 at 8 x 608^2): without it a
// workgroup walks the whole K = 9 Cin (144 steps of 32 at Cin = 512) on its own while most of the chip's wave slots idle.
#define Y3_BF16_SK_HEADER (256 * 1024)     // tickets: the same 256 KiB header as the fp32 entries (yolo3hip.h workspace contract), so a shared workspace has ONE layout
struct Bf16Split {
    int splits, chunk;
    size_t ws_bytes;
};
static Bf16Split plan_bf16_split(long long M, int Nout, int K) {
    Bf16Split s = {1, K / 32, 0};
    static const int on = dev_int("Y3_BF16_SPLITK", 1);
    const long long tiles = (long long)y3_cdiv(M, 64) * y3_cdiv(Nout, 64);
    const int nk = K / 32;
    static const int maxtiles = dev_int("Y3_BF16_SK_MAXTILES", 512);
    if (!on || tiles > maxtiles || tiles * 4 > Y3_BF16_SK_HEADER || nk < 32) return s;
    static const int wgs = dev_int("Y3_BF16_SK_WGS", 1024);
    static const int minsteps = dev_int("Y3_BF16_SK_MINSTEPS", 16);
    int want = (int)(wgs / tiles);                     // aim at <= wgs workgroups
    if (want > 8) want = 8;
    if (want > nk / minsteps) want = nk / minsteps;    // at least minsteps K steps per slice
    if (want < 2) return s;
    s.chunk = y3_cdiv(nk, want);
    s.splits = y3_cdiv(nk, s.chunk);
    if (s.splits < 2) {
        s.splits = 1;
        s.chunk = nk;
        return s;
    }
    s.ws_bytes = (size_t)Y3_BF16_SK_HEADER + (size_t)tiles * s.splits * 64 * 64 * sizeof(float);
    return s;
}

extern "C" size_t y3_conv2d_fwd_bf16_workspace(int m, int cin, int ksize, int cout) {
    if (cout <= 64) return 0;
    return plan_bf16_split(m, cout, ksize * ksize * cin).ws_bytes;      // (only consulted when the 64 x 64-tile path is taken)
}

static int conv2d_fwd_bf16_impl(const y3_tensor* src, const void* wt_t_bf16, const float* bias, int ksize, int stride, const y3_tensor* dst,
                                int dst_is_f32, unsigned flags, float alpha, const float* scale, const float* shift, const y3_tensor* resid,
                                void* workspace, size_t workspace_bytes, y3_stream_t stream);

extern "C" int y3_conv2d_fwd_bf16(const y3_tensor* src, const void* wt_t_bf16, const float* bias, int ksize, int stride, const y3_tensor* dst,
                                  int dst_is_f32, unsigned flags, float alpha, const float* scale, const float* shift, const y3_tensor* resid,
                                  y3_stream_t stream) {
    return conv2d_fwd_bf16_impl(src, wt_t_bf16, bias, ksize, stride, dst, dst_is_f32, flags, alpha, scale, shift, resid, nullptr, 0, stream);
}
extern "C" int y3_conv2d_fwd_bf16_ws(const y3_tensor* src, const void* wt_t_bf16, const float* bias, int ksize, int stride, const y3_tensor* dst,
                                     int dst_is_f32, unsigned flags, float alpha, const float* scale, const float* shift, const y3_tensor* resid,
                                     void* workspace, size_t workspace_bytes, y3_stream_t stream) {
    return conv2d_fwd_bf16_impl(src, wt_t_bf16, bias, ksize, stride, dst, dst_is_f32, flags, alpha, scale, shift, resid, workspace, workspace_bytes, stream);
}

static int conv2d_fwd_bf16_impl(const y3_tensor* src, const void* wt_t_bf16, const float* bias, int ksize, int stride, const y3_tensor* dst,
                                int dst_is_f32, unsigned flags, float alpha, const float* scale, const float* shift, const y3_tensor* resid,
                                void* workspace, size_t workspace_bytes, y3_stream_t stream) {
    if (int e = check_bf16_tensor(src, "conv2d_fwd_bf16 src")) return e;
    if (int e = check_bf16_tensor(dst, "conv2d_fwd_bf16 dst")) return e;
    Y3_CHECK_ARG(wt_t_bf16, "conv2d_fwd_bf16: null weights");
    Y3_CHECK_ARG(ksize == 1 || ksize == 3, "conv2d_fwd_bf16: ksize %d unsupported", ksize);
    Y3_CHECK_ARG(stride == 1 || stride == 2, "conv2d_fwd_bf16: stride %d unsupported", stride);
    Y3_CHECK_ARG(src->c % 32 == 0 && (src->ld & 7) == 0 && ((uintptr_t)src->ptr & 15) == 0, "conv2d_fwd_bf16: Cin=%d must be a multiple of 32, ld of 8, 16-byte aligned", src->c);
    Y3_CHECK_ARG(ksize == 1 || y3_is_pow2(src->c), "conv2d_fwd_bf16: 3x3 needs power-of-two channels");
    const int OH = (src->h + stride - 1) / stride, OW = (src->w + stride - 1) / stride;
    Y3_CHECK_ARG(dst->n == src->n && dst->h == OH && dst->w == OW, "conv2d_fwd_bf16: dst geometry");
    Y3_CHECK_ARG((scale == nullptr) == (shift == nullptr), "conv2d_fwd_bf16: scale/shift must both be given");
    Bf16Args p = {};
    const int taps = ksize * ksize;
    const int pbh = y3_same_pad_before(src->h, ksize, stride), pbw = y3_same_pad_before(src->w, ksize, stride);
    int min_off = 0;
    for (int kh = 0; kh < ksize; ++kh)
        for (int kw = 0; kw < ksize; ++kw) {
            const int t = kh * ksize + kw;
            p.tap_dh[t] = kh - pbh;
            p.tap_dw[t] = kw - pbw;
            const int off = (p.tap_dh[t] * src->w + p.tap_dw[t]) * src->ld;
            if (off < min_off) min_off = off;
        }
    for (int t = 0; t < taps; ++t) {
        p.tap_off[t] = ((p.tap_dh[t] * src->w + p.tap_dw[t]) * src->ld - min_off) * 2;
        p.tap_wt[t] = t * dst->c * src->c;
    }
    // the same as affine maps of the (ksize x ksize) tap grid, tap = ty * ksize + tx
    p.tg_nx = ksize;
    p.tap_off0 = p.tap_off[0];
    p.tap_wt1 = dst->c * src->c;
    p.tg_offx = ksize > 1 ? p.tap_off[1] - p.tap_off[0] : 0;
    p.tg_offy = ksize > 1 ? p.tap_off[ksize] - p.tap_off[0] : 0;
    const long long sbytes = ((long long)src->n * src->h * src->w * src->ld - min_off) * 2;
    const long long wbytes = (long long)taps * dst->c * src->c * 2;
    const long long M = (long long)src->n * OH * OW;
    const long long dbytes = M * dst->ld * (dst_is_f32 ? 4 : 2);
    Y3_CHECK_ARG(sbytes < 0x7fffffffLL && wbytes < 0x7fffffffLL && dbytes < 0x7fffffffLL, "conv2d_fwd_bf16: tensor exceeds 2 GiB (split the batch)");
    p.src = (const u16*)src->ptr + min_off;
    p.src_bytes = (unsigned)sbytes;
    p.wt = (const u16*)wt_t_bf16;
    p.wt_bytes = (unsigned)wbytes;
    p.dst = dst->ptr;
    p.dst_bytes = (unsigned)dbytes;
    p.bias = bias;
    p.scale = scale;
    p.shift = shift;
    if (resid) {
        if (int e = check_bf16_tensor(resid, "conv2d_fwd_bf16 resid")) return e;
        Y3_CHECK_ARG(resid->n == dst->n && resid->h == dst->h && resid->w == dst->w && resid->c == dst->c, "conv2d_fwd_bf16: resid geometry");
        p.resid = (const u16*)resid->ptr;
        p.resid_ld = resid->ld;
        p.resid_bytes = (unsigned)(M * resid->ld * 2);
    }
    p.ntaps = taps;
    p.H = src->h;
    p.W = src->w;
    p.C = src->c;
    if (taps == 1) {
        p.logC = 31;
        p.cmask = 0x7fffffff;
    } else {
        p.logC = y3_ilog2(src->c);
        p.cmask = src->c - 1;
    }
    p.src_ld = src->ld;
    p.OH = OH;
    p.OW = OW;
    p.sh = p.sw = stride;
    p.dst_ld = dst->ld;
    p.Nout = dst->c;
    p.K = taps * src->c;
    p.M = (int)M;
    p.flags = flags;
    p.alpha = alpha;
    p.out_f32 = dst_is_f32;
    {   // 16-byte epilogue accesses need 16-byte aligned rows
        const int eb = dst_is_f32 ? 4 : 2;
        bool ok = ((uintptr_t)dst->ptr & 15) == 0 && ((long long)dst->ld * eb) % 16 == 0;
        if (resid) ok = ok && ((uintptr_t)resid->ptr & 15) == 0 && (resid->ld & 7) == 0;
        p.vec_ok = ok ? 1 : 0;
    }
    hipStream_t st = (hipStream_t)stream;
    // Tile choice, measured in the network on MI355X (tools/infer_bench.py --layers, batch 8 of 416^2 and 608^2).  With
    // 16x the fp32 MFMA rate the kernel is bound by how well resident waves cover each other's LDS / L2 latency, not by
    // the matrix pipe, so the best shape follows the grid size: 256x128 tiles (8 waves, K steps of 64) where that gives
    // 150-300 workgroups, 128x128 for larger grids, many small 64x64 workgroups (8+ waves / SIMD) for the small-M layers.
    const bool k64 = p.K % 64 == 0 && (p.ntaps == 1 || p.C % 64 == 0);
    static const int force = dev_int("Y3_BF16_TILE", 0);   // experiments: 1 = 64x64, 2 = 128x128, 3 = 256x128
    const long long t256 = (long long)y3_cdiv(p.M, 256) * y3_cdiv(p.Nout, 128);
    const long long t128 = (long long)y3_cdiv(p.M, 128) * y3_cdiv(p.Nout, 128);
    // the 256 x 256 ping-pong kernel: whole 64-deep K tiles inside one tap, 16-byte rows for the epilogue, Cout in eights
    static const int pp_mode = dev_int("Y3_BF16_PP", 1);     // 0 = off (A/B against conv_bf16_kernel)
    const long long tpp = (long long)y3_cdiv(p.M, 256) * y3_cdiv(p.Nout, 256);
    if (pp_mode && p.C % 64 == 0 && p.Nout >= 256 && p.Nout % 8 == 0 && p.vec_ok && tpp >= (pp_mode == 2 ? 1 : 96)) {
        p.nbn = y3_cdiv(p.Nout, 256);
        p.ohw = p.OH * p.OW;
        p.dv_nbn = y3_make_div(p.nbn);
        p.dv_ohw = y3_make_div(p.ohw);
        p.dv_ow = y3_make_div(p.OW);
        hipLaunchKernelGGL(conv_bf16_pp_kernel, dim3((unsigned)tpp), dim3(512), 0, st, p);
        Y3_CHECK_LAUNCH("conv_bf16_pp");
        return Y3_OK;
    }
    // the patch kernel for the 32 -> 64 3x3 layers (conv_bf16_c32_kernel)
    static const int patch_on = dev_int("Y3_BF16_PATCH", 1);   // 0 = off (A/B against conv_bf16_kernel<128, 64>)
    const bool patch_epi_ok = !(flags & Y3_BF16_NO_PATCH) && (!(flags & Y3_EPI_LRELU) || (alpha >= 0.f && alpha <= 1.f));   // their leaky-relu is max(x, alpha x)
    if (patch_on && patch_epi_ok && ksize == 3 && p.C == 32 && p.Nout == 64 && !dst_is_f32 && p.vec_ok && ((uintptr_t)wt_t_bf16 & 15) == 0 &&
        (!bias || ((uintptr_t)bias & 3) == 0)) {
        PatchArgs q = {};
        q.src = (const u16*)src->ptr;
        q.wt = (const u16*)wt_t_bf16;
        q.dst = (u16*)dst->ptr;
        q.bias = bias;
        q.scale = scale;
        q.shift = shift;
        q.resid = p.resid;
        q.src_bytes = (unsigned)((long long)src->n * src->h * src->w * src->ld * 2);
        q.dst_bytes = p.dst_bytes;
        q.resid_bytes = p.resid ? p.resid_bytes : 0u;
        q.H = src->h;
        q.W = src->w;
        q.OH = OH;
        q.OW = OW;
        q.src_ld = src->ld;
        q.dst_ld = dst->ld;
        q.resid_ld = p.resid_ld;
        q.pbh = pbh;
        q.pbw = pbw;
        q.xs = y3_cdiv(OW, 32);
        q.rg = y3_cdiv(OH, stride == 1 ? 8 : 4);
        const long long groups = (long long)src->n * q.xs * q.rg;
        Y3_CHECK_ARG(groups < 0x7fffffffLL, "conv2d_fwd_bf16: too many row groups");
        q.groups = (int)groups;
        q.dv_rg = y3_make_div(q.rg);
        q.dv_xs = y3_make_div(q.xs);
        q.flags = flags;
        q.alpha = alpha;
        const unsigned grid = (unsigned)(groups < 512 ? groups : 512);       // two workgroups per CU, each a contiguous run of groups
        if (stride == 1 && q.resid)
            hipLaunchKernelGGL((conv_bf16_c32_kernel<1, true>), dim3(grid), dim3(256), 0, st, q);
        else if (stride == 1)
            hipLaunchKernelGGL((conv_bf16_c32_kernel<1, false>), dim3(grid), dim3(256), 0, st, q);
        else if (q.resid)
            hipLaunchKernelGGL((conv_bf16_c32_kernel<2, true>), dim3(grid), dim3(256), 0, st, q);
        else
            hipLaunchKernelGGL((conv_bf16_c32_kernel<2, false>), dim3(grid), dim3(256), 0, st, q);
        Y3_CHECK_LAUNCH("conv_bf16_c32");
        return Y3_OK;
    }
    if (patch_on && patch_epi_ok && ksize == 3 && p.C == 64 && p.Nout == 128 && !dst_is_f32 && p.vec_ok && ((uintptr_t)wt_t_bf16 & 15) == 0) {
        PatchArgs q = {};
        q.src = (const u16*)src->ptr;
        q.wt = (const u16*)wt_t_bf16;
        q.dst = (u16*)dst->ptr;
        q.bias = bias;
        q.scale = scale;
        q.shift = shift;
        q.resid = p.resid;
        q.src_bytes = (unsigned)((long long)src->n * src->h * src->w * src->ld * 2);
        q.dst_bytes = p.dst_bytes;
        q.resid_bytes = p.resid ? p.resid_bytes : 0u;
        q.H = src->h;
        q.W = src->w;
        q.OH = OH;
        q.OW = OW;
        q.src_ld = src->ld;
        q.dst_ld = dst->ld;
        q.resid_ld = p.resid_ld;
        q.pbh = pbh;
        q.pbw = pbw;
        q.xs = y3_cdiv(OW, 32);
        q.rg = y3_cdiv(OH, stride == 1 ? 4 : 2);
        const long long groups = (long long)src->n * q.xs * q.rg;
        Y3_CHECK_ARG(groups < 0x7fffffffLL, "conv2d_fwd_bf16: too many row groups");
        q.groups = (int)groups;
        q.dv_rg = y3_make_div(q.rg);
        q.dv_xs = y3_make_div(q.xs);
        q.flags = flags;
        q.alpha = alpha;
        const unsigned grid = (unsigned)(groups < 256 ? groups : 256);       // one workgroup of 8 waves per CU
        if (stride == 1)
            hipLaunchKernelGGL(conv_bf16_c64_kernel<1>, dim3(grid), dim3(512), 0, st, q);
        else
            hipLaunchKernelGGL(conv_bf16_c64_kernel<2>, dim3(grid), dim3(512), 0, st, q);
        Y3_CHECK_LAUNCH("conv_bf16_c64");
        return Y3_OK;
    }
    if (p.Nout <= 32) {
        p.nbn = 1;
        launch_bf16<128, 32, 4, 1>(p, y3_cdiv(p.M, 128), st);
    } else if (p.Nout <= 64) {
        p.nbn = 1;
        launch_bf16<128, 64, 4, 1>(p, y3_cdiv(p.M, 128), st);
    } else if (k64 && (force == 3 || (force == 0 && t256 >= 150 && t256 <= 300))) {
        p.nbn = y3_cdiv(p.Nout, 128);
        // (staged epilogue without a residual too, as for the 128 x 128 launches below: same box 8 x 608^2 2.552 -> 2.530 ms, 8 x 416^2
        // 1.729 -> 1.722, 25 / 45 tiles unchanged)
        static const int direct256 = dev_int("Y3_BF16_DIRECT256", 0);   // 1 = the direct epilogue again (A/B)
        if (p.resid || !direct256)
            launch_bf16<256, 128, 4, 2, true, 64>(p, y3_cdiv(p.M, 256) * p.nbn, st);
        else
            launch_bf16<256, 128, 4, 2, false, 64>(p, y3_cdiv(p.M, 256) * p.nbn, st);
    } else if (force == 2 || (force == 0 && t128 >= 512)) {
        p.nbn = y3_cdiv(p.Nout, 128);
        // with or without a residual through the staged epilogue (16-byte stores over whole tile rows).  Round 2 preferred the direct
        // one (2-byte stores from registers, 3 waves per SIMD) where there is nothing to load; re-measured at the batches the tiled path
        // plans (45 x 608^2, same box): the ten 256 -> 128 1x1 launches of the 76^2 stage 60 -> 55 us, 512 -> 128 88 -> 76
        static const int direct128 = dev_int("Y3_BF16_DIRECT128", 0);   // 1 = the direct epilogue again (A/B)
        if (p.resid || !direct128)
            launch_bf16<128, 128, 2, 2, true>(p, y3_cdiv(p.M, 128) * p.nbn, st);
        else
            launch_bf16<128, 128, 2, 2, false>(p, y3_cdiv(p.M, 128) * p.nbn, st);
    } else {
        p.nbn = y3_cdiv(p.Nout, 64);
        const int tiles = y3_cdiv(p.M, 64) * p.nbn;
        const Bf16Split sk = plan_bf16_split(p.M, p.Nout, p.K);
        if (sk.splits > 1 && workspace && workspace_bytes >= sk.ws_bytes) {
            p.sk_splits = sk.splits;
            p.sk_chunk = sk.chunk;
            p.dv_sk = y3_make_div(sk.splits);
            p.tickets = (int*)workspace;
            p.slab = (float*)((char*)workspace + Y3_BF16_SK_HEADER);
            p.ohw = p.OH * p.OW;
            p.dv_nbn = y3_make_div(p.nbn);
            p.dv_ohw = y3_make_div(p.ohw);
            p.dv_ow = y3_make_div(p.OW);
            hipLaunchKernelGGL((conv_bf16_kernel<64, 64, 2, 2, true, 32, 3, true>), dim3(tiles * sk.splits), dim3(256), 0, st, p);
        } else {
            launch_bf16<64, 64, 2, 2>(p, tiles, st);
        }
    }
    Y3_CHECK_LAUNCH("conv_bf16");
    return Y3_OK;
}

extern "C" int y3_conv2d_first_bf16(const y3_tensor* src, const float* wt, const float* bias, const y3_tensor* dst, unsigned flags, float alpha,
                                    const float* scale, const float* shift, y3_stream_t stream) {
    Y3_CHECK_ARG(src && src->ptr && dst && dst->ptr && wt, "conv2d_first_bf16: null pointer");
    Y3_CHECK_ARG(src->c == 4 && dst->c == 32, "conv2d_first_bf16: needs Cin (padded) = 4 and Cout = 32, got %d -> %d", src->c, dst->c);
    Y3_CHECK_ARG((src->ld & 3) == 0 && ((uintptr_t)src->ptr & 15) == 0 && (dst->ld & 7) == 0 && ((uintptr_t)dst->ptr & 15) == 0,
                 "conv2d_first_bf16: src must be 16-byte aligned per pixel, dst 16-byte aligned");
    Y3_CHECK_ARG(dst->n == src->n && dst->h == src->h && dst->w == src->w, "conv2d_first_bf16: dst geometry");
    Y3_CHECK_ARG((scale == nullptr) == (shift == nullptr), "conv2d_first_bf16: scale/shift must both be given");
    const long long npix = (long long)src->n * src->h * src->w;
    Y3_CHECK_ARG(npix * src->ld * 4 < 0x7fffffffLL, "conv2d_first_bf16: input of 2 GiB or more (split the batch)");
    static const int groups = dev_int("Y3_BF16_FIRST_GROUPS", 16);
    const int xtiles = y3_cdiv(src->w, 32), rchunks = y3_cdiv(src->h, 4 * groups);
    const long long wgs = (long long)src->n * rchunks * xtiles;
    Y3_CHECK_ARG(wgs < 0x7fffffffLL, "conv2d_first_bf16: too many tiles");
    hipLaunchKernelGGL(conv_first_bf16_mfma_kernel, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, (const float*)src->ptr, src->ld, wt, bias,
                       scale, shift, (u16*)dst->ptr, dst->ld, src->n, src->h, src->w, xtiles, rchunks, groups, flags, alpha, (unsigned)(npix * src->ld * 4));
    Y3_CHECK_LAUNCH("conv_first_bf16");
    return Y3_OK;
}

extern "C" int y3_f32_to_bf16(const float* src, void* dst, size_t count, y3_stream_t stream) {
    Y3_CHECK_ARG((src && dst) || count == 0, "f32_to_bf16: null pointer");
    if (count == 0) return Y3_OK;
    size_t blocks = (count + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, (u16*)dst, count);
    Y3_CHECK_LAUNCH("f32_to_bf16");
    return Y3_OK;
}

extern "C" int y3_upsample_sum2x_fwd_bf16(const y3_tensor* in, const y3_tensor* out, y3_stream_t stream) {
    if (int e = check_bf16_tensor(in, "upsample_bf16 in")) return e;
    if (int e = check_bf16_tensor(out, "upsample_bf16 out")) return e;
    Y3_CHECK_ARG(out->n == in->n && out->h == 2 * in->h && out->w == 2 * in->w, "upsample_bf16: geometry");
    const long long npix = (long long)in->n * in->h * in->w;
    hipLaunchKernelGGL(upsample_bf16_kernel, dim3(y3_cdiv(npix, 4)), dim3(256), 0, (hipStream_t)stream, (const u16*)in->ptr, in->ld, in->c,
                       (u16*)out->ptr, out->ld, out->c, in->n, in->h, in->w);
    Y3_CHECK_LAUNCH("upsample_bf16");
    return Y3_OK;
}
