// fp32 convolution arithmetic on the bf16 matrix pipe of gfx950 ("x3"): the forward pass and the data gradient of the
// MFMA fast path (conv_fast.h) with every operand element cut into three bf16 pieces.
//
//   x = x0 + x1 + x2 exactly:  x0 = bf16(x), x1 = bf16(x - x0), x2 = x - x0 - x1      (round to nearest even; 8 + 8 + 8
//   significant bits, the residuals are exact in fp32 and the last one is exactly a bf16)
//   a * b = sum over the piece pairs; the six pairs with piece indices i + j <= 2 are multiplied, the other three are
//   below 2^-26 of the product (fp32 rounding of the running sum is 2^-24): the result is fp32-class, not bf16-class.
//
// One K step of 16 is ONE v_mfma_f32_32x32x16_bf16 per piece pair and 32x32 block: 6 x 32 cycles where
// v_mfma_f32_32x32x2_f32 needs 8 x 64 -- 2.67x fewer matrix-pipe cycles for the same fp32 product, fp32 accumulation.
// The accumulator layout is that of the fp32 instruction, so the split-K hand-off and the epilogue are conv_fast_finish().
//
// Why a kernel of its own (round 3 measured the same arithmetic as a variant of conv_fast_body at +1 %): with 32x32 wave
// tiles a wave has 6 MFMAs (192 cycles) per barrier, too little to cover the barrier, the split (5 vector instructions per
// element) and the LDS round trip.  Here a wave owns 64 x 64 (24 MFMAs = 768 cycles per K step), its fragments are read
// one PIECE ahead (not a whole set ahead: 56 instead of 96 fragment registers), and every MFMA is followed by a slot that
// carries at most a handful of vector / LDS instructions:
//   slots of groups 0 .. GB-1  reads of this step's remaining pieces, then split + LDS stores of the NEXT step's tile
//                              (global data loaded two steps earlier), four sub-steps of <= 6 vector instructions per 16 bytes
//   LDS-only barrier           global loads stay in flight across it
//   slots of the other groups  reads of the next step's first pieces, global loads for the step THREE ahead
// The WEIGHTS arrive already split (y3_x3_split_weights: three bf16 planes of the copy of the kernel whose K axis is contiguous
// per output column -- forward: [tap][Cout][Cin], data gradient: [tap][Cin][Cout] -- written once per optimiser step): the probe
// showed the loop bound by the vector ALU, not by the matrix pipe (192 k cycles per workgroup, 106 k = the MFMA time with the
// split ablated, loads free either way), and the weight tile was 4/5 of the split work.  A weight tile is 3 x 16-byte loads
// and 3 ds_write_b128 per thread and step; only the activations are split in the kernel.  The planes are stored TILE-WISE,
// [tap][C / 16][row][piece][16 k]: the 128 rows x 3 pieces x 32 bytes a workgroup needs for one K step are 12 KB of consecutive
// addresses, every 128-byte line fully used (as [piece][tap][row][C] a step touched 32 bytes of each of 384 lines, re-fetched
// from L2 every step: 4x the bytes, and the loop ran slower than with fp32 weights split in the kernel).  Both operands share one LDS image:
// [piece][k half][row][8 k] bf16, the 32 lanes of a fragment read (one k half, consecutive rows) cover 512 contiguous bytes.
#include "conv_fast.h"

typedef __bf16 y3_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 y3_bf16x2 __attribute__((ext_vector_type(2)));
typedef float y3_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned x3_pk(float a, float b) {      // v_cvt_pk_bf16_f32: two roundings to nearest even, a in the low half
    const y3_f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, y3_bf16x2));
}
__device__ __forceinline__ float x3_lo(unsigned pk) { return __uint_as_float(pk << 16); }
__device__ __forceinline__ float x3_hi(unsigned pk) { return __uint_as_float(pk & 0xffff0000u); }

#ifndef Y3_X3_GB
#define Y3_X3_GB 4      // the barrier sits after MFMA group GB - 1 (of 6)
#endif

template <int BM, int BN, int WM, int WN, bool DENSE, bool BNS>
__device__ __forceinline__ void conv_x3_body(const FastArgs& p, const int braw, const int grid) {
    constexpr int BK = 16, KV = 4;
    constexpr int THREADS = 64 * WM * WN;
    constexpr int TM = BM / WM, TN = BN / WN, MB = TM / 32, NB = TN / 32;
    constexpr int A_LOADS = BM * KV / THREADS;
    constexpr int B_LOADS = BN * 6 / THREADS;      // weight tile: BN rows x 3 pieces x 2 k halves of 16 bytes, consecutive in memory
    static_assert((BM * KV) % THREADS == 0 && (BN * 6) % THREADS == 0 && TM % 32 == 0 && TN % 32 == 0, "tile shape");
    // k-half planes are 64 bytes longer than their rows: LDS stores are banked by (address / 4) mod 32, and without the pad the
    // two halves a 16-lane store group writes fall on the same banks
    constexpr int AH = BM * 8 + 32, BH = BN * 8 + 32;       // u16 per k half
    constexpr int A3 = 6 * AH, B3 = 6 * BH, BUF = A3 + B3;  // u16 per piece set / per buffer
    constexpr int RED = (BNS ? 6 : 2) * WM * BN;            // floats: column sums of the epilogue
    __shared__ __attribute__((aligned(16))) unsigned short lds[2 * BUF + 2 * RED];
    float (*red)[WM][BN] = reinterpret_cast<float (*)[WM][BN]>(&lds[2 * BUF]);

    Y3_TSTAMP(0);
    Y3_ABL_INIT();
#ifdef Y3_TIMING
    if (y3_timing_buf && threadIdx.x == 0) {
        y3_timing_buf[(size_t)blockIdx.x * 8 + 4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_ID
        y3_timing_buf[(size_t)blockIdx.x * 8 + 5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // XCC_ID
        y3_timing_buf[(size_t)blockIdx.x * 8 + 6] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    const FastWork fw = conv_fast_decode<BM, BN, WM, WN, BK, false, true>(p, braw, grid);
    const int tid = fw.tid, l31 = fw.l31, lh = fw.lh, wm = fw.wm, wn = fw.wn;
    const int m0 = fw.m0, n0 = fw.n0, kbeg = fw.kbeg, kend = fw.kend;
    const int ohw = fw.ohw, OW = fw.OW, aM = fw.aM, aH = fw.aH, aW = fw.aW, src_ld = fw.src_ld, csh = fw.csh, csw = fw.csw, ntaps = fw.ntaps, Nout = fw.Nout;
    const Y3Div dv_ohw = fw.dv_ohw, dv_ow = fw.dv_ow;

    const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.src), 0, p.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wt), 0, p.wt_bytes, 0x00020000);

    // loop-invariant per-lane offsets: thread -> (row idx / 4, k quad idx % 4) of both operand tiles
    unsigned a_voff[A_LOADS], a_mask[A_LOADS], b_voff[B_LOADS];
    int b_st[B_LOADS];
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
        for (int t = 0; t < ntaps; ++t) {
            const int ih = ih0 + p.tap_dh[t], iw = iw0 + p.tap_dw[t];
            if (ok && (unsigned)ih < (unsigned)aH && (unsigned)iw < (unsigned)aW) msk |= 1u << t;
        }
        a_mask[i] = msk;
    }
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) {      // weight tile: 16-byte chunk j of the step's block -> row j / 6, piece (j % 6) / 2, k half j % 2
        const int j = tid + i * THREADS, row = j / 6, part = j % 6;
        b_voff[i] = n0 + row < Nout ? (unsigned)(n0 * 96 + j * 16) : Y3_OOB;
        b_st[i] = (part >> 1) * 2 * BH + (part & 1) * BH + row * 8;
    }
    // LDS addresses (u16 units): this thread's 8-byte store slot of an activation piece plane, and its 16-byte fragment slots
    const int a_st = (a_kv >> 1) * AH + (tid / KV) * 8 + (a_kv & 1) * 4;
    const int a_fr = lh * AH + (wm * TM + l31) * 8;
    const int b_fr = lh * BH + (wn * TN + l31) * 8;

    // Global -> register staging: TWO register sets (tile of K step s in set s & 1), loads issued three steps ahead
    // (conv_fast_body); dead steps at or beyond kend fetch nothing and return zeros.
    f32x4 ra[2][A_LOADS], rb[2][B_LOADS];      // (rb: 16 bytes of bf16 each, carried as four dwords)
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
        o.wrow = p.tg_w0 + ty * p.tg_wy + tx * p.tg_wx;         // (weight tap) * C
        o.tap = live ? o.tap : 31;
        return o;
    };
    auto gload_one = [&](const Soff& o, auto S, auto E) {
        constexpr int set = decltype(S)::value, e = decltype(E)::value;
        if constexpr (e < A_LOADS) {
            const unsigned vo = ((a_mask[e] >> o.tap) & 1u) ? a_voff[e] : Y3_OOB;
            ra[set][e] = __builtin_amdgcn_raw_buffer_load_b128(rs_src, vo, (unsigned)(o.toff + o.cb * 4), 0);
        } else {
            // the step's block [(weight tap) * C / 16 + chunk][Nout rows][3 pieces][16 k]: 96 bytes per row
            constexpr int eb = e - A_LOADS;
            rb[set][eb] = __builtin_amdgcn_raw_buffer_load_b128(rs_wt, b_voff[eb] | o.dead, (unsigned)(((o.wrow + o.cb) >> 4) * p.Nout) * 96u, 0);
        }
    };
    constexpr int NW = A_LOADS + B_LOADS;
    auto gload = [&](int k0, auto S) {
        const Soff o = soff_prep(k0);
        y3_for_each_ic(std::make_integer_sequence<int, NW>{}, [&](auto E) { gload_one(o, S, E); });
    };

    // Split + store of one 16-byte load (4 consecutive k of one row) in four sub-steps of <= 6 vector instructions; the
    // residual overwrites the staging register.
    unsigned pk0 = 0, pk1 = 0;
    auto st_addr = [&](auto W, int buf, int piece) -> unsigned short* {
        constexpr int w = decltype(W)::value;
        return &lds[buf * BUF + piece * 2 * AH + a_st + w * (THREADS / KV) * 8];
    };
    auto split_sub = [&](auto S, auto W, auto SUB, int buf) {      // activation load W (< A_LOADS), sub-step SUB
        constexpr int set = decltype(S)::value, w = decltype(W)::value, sub = decltype(SUB)::value;
        f32x4& v = ra[set][w];
        if constexpr (sub == 0) {
            pk0 = x3_pk(v[0], v[1]);
            pk1 = x3_pk(v[2], v[3]);
            *reinterpret_cast<uint2*>(st_addr(W, buf, 0)) = make_uint2(pk0, pk1);
            v[0] -= x3_lo(pk0);
            v[1] -= x3_hi(pk0);
        } else if constexpr (sub == 1) {
            v[2] -= x3_lo(pk1);
            v[3] -= x3_hi(pk1);
            pk0 = x3_pk(v[0], v[1]);
            pk1 = x3_pk(v[2], v[3]);
            *reinterpret_cast<uint2*>(st_addr(W, buf, 1)) = make_uint2(pk0, pk1);
        } else if constexpr (sub == 2) {
            v[0] -= x3_lo(pk0);
            v[1] -= x3_hi(pk0);
            v[2] -= x3_lo(pk1);
        } else {
            v[3] -= x3_hi(pk1);
            *reinterpret_cast<uint2*>(st_addr(W, buf, 2)) = make_uint2(x3_pk(v[0], v[1]), x3_pk(v[2], v[3]));
        }
    };
    auto b_store = [&](auto S, auto EB, int buf) {      // weight load EB: 16 bytes straight into its (piece, k half, row) slot
        constexpr int set = decltype(S)::value, eb = decltype(EB)::value;
        *reinterpret_cast<f32x4*>(&lds[buf * BUF + A3 + b_st[eb]]) = rb[set][eb];
    };
    constexpr int NS = A_LOADS * 4 + B_LOADS;            // LDS-store events of a K step: the split sub-steps of the activations, then the weight stores
    auto store_event = [&](auto S, auto E, int buf) {
        constexpr int e = decltype(E)::value;
        if constexpr (e < A_LOADS * 4)
            split_sub(S, std::integral_constant<int, e / 4>{}, std::integral_constant<int, e % 4>{}, buf);
        else
            b_store(S, std::integral_constant<int, e - A_LOADS * 4>{}, buf);
    };
    auto split_store_all = [&](auto S, int buf) {     // prologue: the whole tile of register set S
        y3_for_each_ic(std::make_integer_sequence<int, NS>{}, [&](auto E) { store_event(S, E, buf); });
    };

    // fragments: [step parity][piece][block]; a step multiplies the piece pairs in the order (0,0) (0,1) (0,2) (1,0) (1,1) (2,0)
    y3_bf16x8 FA[2][3][MB], FB[2][3][NB];
    auto read_a = [&](auto F, auto PC, int i, int buf) {
        FA[decltype(F)::value][decltype(PC)::value][i] =
            *reinterpret_cast<const y3_bf16x8*>(&lds[buf * BUF + decltype(PC)::value * 2 * AH + a_fr + i * 32 * 8]);
    };
    auto read_b = [&](auto F, auto PC, int j, int buf) {
        FB[decltype(F)::value][decltype(PC)::value][j] =
            *reinterpret_cast<const y3_bf16x8*>(&lds[buf * BUF + A3 + decltype(PC)::value * 2 * BH + b_fr + j * 32 * 8]);
    };

    f32x16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    using C2 = std::integral_constant<int, 2>;
    const int nk = (kend - kbeg) / BK;
    {
        constexpr int NMG = MB * NB;          // MFMAs per piece pair
        constexpr int NM = 6 * NMG;           // MFMAs per K step
        constexpr int SB = Y3_X3_GB * NMG;    // slots in front of the barrier
        // events in front of the barrier: reads of pieces B2 (needed by group 2), A1 (group 3), A2 (group 5), one per slot from
        // slot 0; and the NW * 4 split sub-steps, spread evenly over the SB slots
        constexpr int NR1 = NB + MB + MB;
        // events behind it: reads of the next step's A0, B0, B1, two per slot; then the NW global loads, one per slot
        constexpr int NR2 = MB + NB + NB;
        constexpr int SP = NM - SB;
        constexpr int RS2 = (NR2 + 1) / 2;    // slots taken by the reads
        static_assert(NR1 <= SB && RS2 < SP, "not enough MFMA slots for the events of a K step");
        constexpr int LPS = (NW + SP - RS2 - 1) / (SP - RS2);     // global loads per slot
        constexpr int LS = (NW + LPS - 1) / LPS;                  // slots taken by the loads
        Soff nxt;
        int knext = 0;
        auto slot = [&](auto Mi, auto P) {
            constexpr int m = decltype(Mi)::value, cur = decltype(P)::value;
            using F = std::integral_constant<int, cur>;
            using G = std::integral_constant<int, cur ^ 1>;      // register set stored / reloaded; fragment set of the next step
            if constexpr (m < SB) {
                if constexpr (m < NR1) {
                    if constexpr (m < NB)
                        read_b(F{}, C2{}, m, cur);
                    else if constexpr (m < NB + MB)
                        read_a(F{}, C1{}, m - NB, cur);
                    else
                        read_a(F{}, C2{}, m - NB - MB, cur);
                }
                constexpr int s0 = m * NS / SB, s1 = (m + 1) * NS / SB;
                if (!Y3_ABL(2))
                    y3_for_each_ic(std::make_integer_sequence<int, s1 - s0>{}, [&](auto D) { store_event(G{}, std::integral_constant<int, s0 + decltype(D)::value>{}, cur ^ 1); });
            } else {
                constexpr int q = m - SB;
                if constexpr (q < RS2) {
                    y3_for_each_ic(std::make_integer_sequence<int, 2>{}, [&](auto D) {
                        constexpr int r = q * 2 + decltype(D)::value;
                        if constexpr (r < MB)
                            read_a(G{}, C0{}, r, cur ^ 1);
                        else if constexpr (r < MB + NB)
                            read_b(G{}, C0{}, r - MB, cur ^ 1);
                        else if constexpr (r < NR2)
                            read_b(G{}, C1{}, r - MB - NB, cur ^ 1);
                    });
                } else if constexpr (q < RS2 + LS) {
                    if (!Y3_ABL(1))
                        y3_for_each_ic(std::make_integer_sequence<int, LPS>{}, [&](auto D) {
                            constexpr int e = (q - RS2) * LPS + decltype(D)::value;
                            if constexpr (e < NW) gload_one(nxt, G{}, std::integral_constant<int, e>{});
                        });
                } else if constexpr (q == RS2 + LS) {
                    nxt = soff_prep(knext);       // the scalar arithmetic of the NEXT step's loads
                }
            }
        };
        auto step = [&](auto P, int ks) {      // P = ks & 1: LDS buffer and fragment set of tile ks
            constexpr int cur = decltype(P)::value;
            knext = kbeg + (ks + 4) * BK;
            y3_for_each_ic(std::make_integer_sequence<int, NM>{}, [&](auto Mi) {
                constexpr int m = decltype(Mi)::value;
                constexpr int g = m / NMG, i = (m % NMG) / NB, j = m % NB;
                constexpr int pa = g < 3 ? 0 : (g < 5 ? 1 : 2), pb = g < 3 ? g : (g == 3 ? 0 : (g == 4 ? 1 : 0));
                if constexpr (m == SB) {
                    if (!Y3_ABL(4)) y3_lds_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[cur][pa][i], FB[cur][pb][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                slot(Mi, P);
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (RS2 + LS >= SP) nxt = soff_prep(knext);
        };
        gload(kbeg, C0{});
        split_store_all(C0{}, 0);
        __syncthreads();
        Y3_TSTAMP(1);
        // (pinned: issued the other way round -- the scheduler is free to -- the set the loop consumes first is the YOUNGER one and
        // the loop's first wait becomes vmcnt(0) instead of "all but the newest tile")
        __builtin_amdgcn_sched_barrier(0);
        gload(kbeg + BK, C1{});
        __builtin_amdgcn_sched_barrier(0);
        gload(kbeg + 2 * BK, C0{});
        __builtin_amdgcn_sched_barrier(0);
        nxt = soff_prep(kbeg + 3 * BK);      // offsets of the loads issued in step 0 (tile 3)
#pragma unroll
        for (int i = 0; i < MB; ++i) read_a(C0{}, C0{}, i, 0);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            read_b(C0{}, C0{}, j, 0);
            read_b(C0{}, C1{}, j, 0);
        }
        // Uniform steps, two per iteration (buffer and register set are compile-time indices); an odd step count is padded with one
        // dead step (all-zero operands), and the stores, barrier and reads of the last step serve a tile nobody multiplies.
        for (int ks = 0; ks < nk; ks += 2) {
            step(C0{}, ks);
            step(C1{}, ks + 1);
        }
    }
    Y3_TSTAMP(2);
    conv_fast_finish<BM, BN, WM, WN, DENSE, BNS, true>(p, fw, acc, red);
}

// ---------------------------------------------------------------------------
// The same arithmetic for the stride-1 3x3 layers with the A operand staged as a PATCH ("x3p").  An implicit-GEMM loop over
// (tap, channel) fetches and splits every activation nine times, once per tap; conv_x3_body above spends more time on its
// global loads and on the split than on its matrix instructions (probe: 97.8 us per launch, 79.6 without the loads, 86.8 without
// split + stores, 37 us of MFMA at the clock the chip holds).  Here the K loop runs channel chunk outermost: the 16 channels of
// a chunk are staged ONCE for all pixels the tile's nine taps touch -- the BM output pixels plus W + 1 pixels of halo on either
// side in the flattened (image, row, column) order, <= Y3_X3P_ROWS rows -- and the nine taps of the chunk read their A
// fragments from that one patch at a row shift of dh * W + dw.  Taps that fall outside the image (zero padding; rows of
// another image) are a per-lane bit test that points the fragment read at an all-zero row.  Per K step the workgroup then
// loads and splits only its weight tile; the activations cost 4 loads and 88 vector instructions per thread per NINE steps.
// The loop body is 18 steps (two chunks: LDS buffer, register set and tap are compile-time indices).
// ---------------------------------------------------------------------------
#define Y3_X3P_ROWS 240      // patch rows held in LDS: BM + 2 * (W + 1) <= 240, i.e. image widths up to 55 with 128-row tiles

struct X3Pk {
    unsigned p0, p1;
};
// one of the four sub-steps of splitting 4 consecutive k (16 bytes of fp32) into the three piece planes (8-byte slots d0 / d1 / d2)
template <int SUB>
__device__ __forceinline__ void x3_split_sub(f32x4& v, X3Pk& st, unsigned short* d0, unsigned short* d1, unsigned short* d2) {
    if constexpr (SUB == 0) {
        st.p0 = x3_pk(v[0], v[1]);
        st.p1 = x3_pk(v[2], v[3]);
        *reinterpret_cast<uint2*>(d0) = make_uint2(st.p0, st.p1);
        v[0] -= x3_lo(st.p0);
        v[1] -= x3_hi(st.p0);
    } else if constexpr (SUB == 1) {
        v[2] -= x3_lo(st.p1);
        v[3] -= x3_hi(st.p1);
        st.p0 = x3_pk(v[0], v[1]);
        st.p1 = x3_pk(v[2], v[3]);
        *reinterpret_cast<uint2*>(d1) = make_uint2(st.p0, st.p1);
    } else if constexpr (SUB == 2) {
        v[0] -= x3_lo(st.p0);
        v[1] -= x3_hi(st.p0);
        v[2] -= x3_lo(st.p1);
    } else {
        v[3] -= x3_hi(st.p1);
        *reinterpret_cast<uint2*>(d2) = make_uint2(x3_pk(v[0], v[1]), x3_pk(v[2], v[3]));
    }
}

template <int BM, int BN, int WM, int WN, bool BNS>
__device__ __forceinline__ void conv_x3p_body(const FastArgs& p, const int braw, const int grid) {
    constexpr int BK = 16, KV = 4, NT = 9;
    constexpr int THREADS = 64 * WM * WN;
    constexpr int TM = BM / WM, TN = BN / WN, MB = TM / 32, NB = TN / 32;
    constexpr int PR = Y3_X3P_ROWS;
    constexpr int A4 = (PR * KV + THREADS - 1) / THREADS;      // patch loads per thread and chunk
    constexpr int B_LOADS = BN * 6 / THREADS;      // weight tile: BN rows x 3 pieces x 2 k halves of 16 bytes, consecutive in memory
    static_assert((BN * 6) % THREADS == 0 && TM % 32 == 0 && TN % 32 == 0, "tile shape");
    constexpr int AHP = (PR + 1) * 8 + 32;      // u16 per k half of a patch plane: PR rows, the all-zero row (index PR), 64 bytes of pad
    constexpr int AP = 6 * AHP;                 // one patch buffer: 3 pieces x 2 k halves
    constexpr int BH = BN * 8 + 32, B3 = 6 * BH;
    constexpr int RED = (BNS ? 6 : 2) * WM * BN;
    __shared__ __attribute__((aligned(16))) unsigned short lds[2 * AP + 2 * B3 + 2 * RED];     // [patch 0][patch 1][B 0][B 1][column sums]
    constexpr int BOFF = 2 * AP;
    float (*red)[WM][BN] = reinterpret_cast<float (*)[WM][BN]>(&lds[2 * AP + 2 * B3]);

    Y3_TSTAMP(0);
    Y3_ABL_INIT();
#ifdef Y3_TIMING
    if (y3_timing_buf && threadIdx.x == 0) {
        y3_timing_buf[(size_t)blockIdx.x * 8 + 4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_ID
        y3_timing_buf[(size_t)blockIdx.x * 8 + 5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // XCC_ID
        y3_timing_buf[(size_t)blockIdx.x * 8 + 6] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    const FastWork fw = conv_fast_decode<BM, BN, WM, WN, BK, true, true>(p, braw, grid);
    const int tid = fw.tid, l31 = fw.l31, lh = fw.lh, wm = fw.wm, wn = fw.wn;
    const int m0 = fw.m0, n0 = fw.n0, kbeg = fw.kbeg, kend = fw.kend;
    const int ohw = fw.ohw, OW = fw.OW, aM = fw.aM, aH = fw.aH, aW = fw.aW, src_ld = fw.src_ld, Nout = fw.Nout;
    const Y3Div dv_ohw = fw.dv_ohw, dv_ow = fw.dv_ow;

    const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.src), 0, p.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wt), 0, p.wt_bytes, 0x00020000);

    // patch row r <-> input pixel m0 - halo + r of the flattened (image, row, column) index (stride 1, SAME: output pixel m reads
    // input pixels m + dh * W + dw); p.src is biased by -halo pixels (make_fast), rows before the tensor / past its end read zeros
    const int halo = aW + 1;
    const int a_kv = tid % KV;
    unsigned pa_voff[A4];
#pragma unroll
    for (int j = 0; j < A4; ++j) {
        const int row = (tid + j * THREADS) / KV;
        const int q = m0 - halo + row;
        const bool ok = row < BM + 2 * halo && row < PR && q >= 0 && q < aM;
        pa_voff[j] = ok ? (unsigned)((q + halo) * src_ld + a_kv * 4) * 4u : Y3_OOB;
    }
    unsigned b_voff[B_LOADS];
    int b_st[B_LOADS];
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) {      // weight tile: 16-byte chunk j of the step's block -> row j / 6, piece (j % 6) / 2, k half j % 2
        const int j = tid + i * THREADS, row = j / 6, part = j % 6;
        b_voff[i] = n0 + row < Nout ? (unsigned)(n0 * 96 + j * 16) : Y3_OOB;
        b_st[i] = (part >> 1) * 2 * BH + (part & 1) * BH + row * 8;
    }
    // per-tap scalars: row shift of the patch, and the byte offset of the tap's block of the K-contiguous kernel copy
    int s_shift[NT];
    unsigned s_wtap[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        s_shift[t] = (p.tap_dh[t] * aW + p.tap_dw[t]) * 8;                                   // u16 units (8 per row)
        s_wtap[t] = (unsigned)((p.tg_w0 + (t / 3) * p.tg_wy + (t % 3) * p.tg_wx) >> 4);      // (weight tap) * C / 16: first 16-channel block of the tap
    }
    // which taps of this lane's output pixels stay inside the image (bit t), per 32-row block
    unsigned fmask[MB];
    int a_fr0[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        const int m = m0 + wm * TM + i * 32 + l31;
        const bool ok = m < aM;
        const int mm = ok ? m : 0;
        const int n = y3_div(mm, dv_ohw);
        const int r = mm - n * ohw;
        const int oh = y3_div(r, dv_ow);
        const int ow = r - oh * OW;
        unsigned msk = 0;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int ih = oh + p.tap_dh[t], iw = ow + p.tap_dw[t];
            if (ok && (unsigned)ih < (unsigned)aH && (unsigned)iw < (unsigned)aW) msk |= 1u << t;
        }
        fmask[i] = msk;
        a_fr0[i] = lh * AHP + (halo + wm * TM + i * 32 + l31) * 8;
    }
    const int zero_fr = lh * AHP + PR * 8;
    const int pa_st = (a_kv >> 1) * AHP + (tid / KV) * 8 + (a_kv & 1) * 4;
    const int b_fr = lh * BH + (wn * TN + l31) * 8;

    // K bookkeeping: a slice covers whole chunks; step s of the slice is (chunk c0 + s / 9, tap s % 9)
    const int nk = (kend - kbeg) / BK;            // multiple of 18 (host)
    const int c0 = (kbeg / BK) / NT;
    const int nchunks = nk / NT;

    // Staging registers.  A K step is only 768 cycles of MFMA per wave, so a tile loaded "two steps ahead" has about one step
    // (~1 us) to arrive -- less than an L2 round trip under load, and the loop stood at its vmcnt waits (probe: 83 k cycles per
    // workgroup against 56 k with the loads ablated).  The weight tiles therefore go through a ring of THREE sets (tile s in set
    // s % 3, loaded in step s - 4, stored in step s - 1: two full steps in flight), and the patch of the next chunk, loaded at
    // tap 0, is not touched before tap 3.
    // The activations stream through an XCD's L2 once per launch (each patch row is read by one workgroup of the XCD), the weight
    // block of the XCD's column tiles is re-read by every row tile: the patch loads are non-temporal so that they do not evict it.
    const bool a_nt = (p.x3_mode & 1) != 0;
    f32x4 rp[A4];                                  // patch staging (one set: loaded at tap 0 of a chunk, stored over its taps 3..8)
    f32x4 rb[3][B_LOADS];
    X3Pk pk_a = {0, 0};
    auto patch_load = [&](int chunk_rel) {         // chunk c0 + chunk_rel; beyond the slice: dead (zeros)
        const bool live = chunk_rel < nchunks;
        const unsigned soff = (unsigned)((c0 + (live ? chunk_rel : 0)) * BK) * 4u;
#pragma unroll
        for (int j = 0; j < A4; ++j)
            rp[j] = a_nt ? __builtin_amdgcn_raw_buffer_load_b128(rs_src, live ? pa_voff[j] : Y3_OOB, soff, 2 /* nt */)
                         : __builtin_amdgcn_raw_buffer_load_b128(rs_src, live ? pa_voff[j] : Y3_OOB, soff, 0);
    };
    auto patch_load_one = [&](auto J, int chunk_rel) {
        constexpr int j = decltype(J)::value;
        const bool live = chunk_rel < nchunks;
        const unsigned soff = Y3_ABL(128) ? 0u : (unsigned)((c0 + (live ? chunk_rel : 0)) * BK) * 4u;
        rp[j] = a_nt ? __builtin_amdgcn_raw_buffer_load_b128(rs_src, live ? pa_voff[j] : Y3_OOB, soff, 2 /* nt */)
                     : __builtin_amdgcn_raw_buffer_load_b128(rs_src, live ? pa_voff[j] : Y3_OOB, soff, 0);
    };
    auto patch_sub = [&](auto E, int pb) {         // sub-step E (0 .. 4 * A4 - 1) of the patch in rp -> patch buffer pb
        constexpr int e = decltype(E)::value, j = e / 4, sub = e % 4;
        const int row = (tid + j * THREADS) / KV;
        if (A4 * THREADS <= PR * KV || row < PR) {
            unsigned short* d = &lds[pb * AP + pa_st + j * (THREADS / KV) * 8];
            x3_split_sub<sub>(rp[j], pk_a, d, d + 2 * AHP, d + 4 * AHP);
        }
    };
    auto b_load_one = [&](auto S, auto E, int step, auto T) {      // step (relative to the slice) with compile-time tap T
        constexpr int set = decltype(S)::value, e = decltype(E)::value, t = decltype(T)::value;
        const bool live = step < nk;
        const int chunk = c0 + (live ? step : 0) / NT;
        // the step's block [(weight tap) * C / 16 + chunk][Nout rows][3 pieces][16 k]: 96 bytes per row
        rb[set][e] = __builtin_amdgcn_raw_buffer_load_b128(rs_wt, b_voff[e] | (live ? 0u : Y3_OOB), Y3_ABL(64) ? 0u : (s_wtap[t] + (unsigned)chunk) * (unsigned)p.Nout * 96u, 0);
    };
    auto b_sub = [&](auto S, auto E, int buf) {      // weight load E: 16 bytes straight into its (piece, k half, row) slot
        constexpr int set = decltype(S)::value, e = decltype(E)::value;
        *reinterpret_cast<f32x4*>(&lds[BOFF + buf * B3 + b_st[e]]) = rb[set][e];
    };

    y3_bf16x8 FA[2][3][MB], FB[2][3][NB];
    int a_addr[2][MB];                             // fragment address (u16 index inside a patch buffer, piece 0) of the step's tap, per block
    auto set_addr = [&](auto F, auto T) {
        constexpr int f = decltype(F)::value, t = decltype(T)::value;
#pragma unroll
        for (int i = 0; i < MB; ++i) a_addr[f][i] = ((fmask[i] >> t) & 1u) ? a_fr0[i] + s_shift[t] : zero_fr;
    };
    auto read_a = [&](auto F, auto PC, int i, int pb) {
        FA[decltype(F)::value][decltype(PC)::value][i] =
            *reinterpret_cast<const y3_bf16x8*>(&lds[pb * AP + decltype(PC)::value * 2 * AHP + a_addr[decltype(F)::value][i]]);
    };
    auto read_b = [&](auto F, auto PC, int j, int buf) {
        FB[decltype(F)::value][decltype(PC)::value][j] =
            *reinterpret_cast<const y3_bf16x8*>(&lds[BOFF + buf * B3 + decltype(PC)::value * 2 * BH + b_fr + j * 32 * 8]);
    };

    f32x16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    using C2 = std::integral_constant<int, 2>;
    {
        constexpr int NMG = MB * NB, NM = 6 * NMG, SB = Y3_X3_GB * NMG, SP = NM - SB;
        constexpr int NR1 = NB + MB + MB;          // reads in front of the barrier: B2, A1, A2
        constexpr int NSB = B_LOADS;               // weight stores per step
        constexpr int NSA = A4 * 4;                // patch sub-steps per chunk, spread over taps TA0 .. 8
        constexpr int TA0 = 3;
        constexpr int SA_PER = (NSA + NT - TA0 - 1) / (NT - TA0);
        constexpr int NR2 = MB + NB + NB, RS2 = (NR2 + 1) / 2;
        static_assert(NR1 <= SB && RS2 + B_LOADS + (A4 + 1) / 2 <= SP, "not enough MFMA slots for the events of a K step");
        // step U of the 18-step body (two chunks), chunk pair starting at relative chunk cp
        auto step = [&](auto U, int cp) {
            constexpr int u = decltype(U)::value, cr = u / NT, t = u % NT, cur = u & 1;
            constexpr int un = (u + 1) % (2 * NT), crn = un / NT, tn = un % NT;          // the next step: patch buffer, tap
            using F = std::integral_constant<int, cur>;
            using G = std::integral_constant<int, cur ^ 1>;
            using RS = std::integral_constant<int, (u + 1) % 3>;      // weight register set stored (tile u + 1) and reloaded (tile u + 4) in this step
            const int srel = cp * NT + u;           // step relative to the slice
            y3_for_each_ic(std::make_integer_sequence<int, NM>{}, [&](auto Mi) {
                constexpr int m = decltype(Mi)::value;
                constexpr int g = m / NMG, i = (m % NMG) / NB, j = m % NB;
                constexpr int pa = g < 3 ? 0 : (g < 5 ? 1 : 2), pb = g < 3 ? g : (g == 3 ? 0 : (g == 4 ? 1 : 0));
                if constexpr (m == SB) {
                    if (!Y3_ABL(4)) y3_lds_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[cur][pa][i], FB[cur][pb][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (m < SB) {
                    if constexpr (m < NB)
                        read_b(F{}, C2{}, m, cur);
                    else if constexpr (m < NB + MB)
                        read_a(F{}, C1{}, m - NB, cr);
                    else if constexpr (m < NR1)
                        read_a(F{}, C2{}, m - NB - MB, cr);
                    // weight tile of the next step: split + store
                    constexpr int s0 = m * NSB / SB, s1 = (m + 1) * NSB / SB;
                    if (!Y3_ABL(2))
                        y3_for_each_ic(std::make_integer_sequence<int, s1 - s0>{}, [&](auto D) { b_sub(RS{}, std::integral_constant<int, s0 + decltype(D)::value>{}, cur ^ 1); });
                    // the NEXT chunk's patch: its sub-steps ride in the second half of the slots of taps TA0 .. 8
                    if constexpr (t >= TA0) {
                        constexpr int a0 = (t - TA0) * SA_PER, a1 = ((t - TA0 + 1) * SA_PER < NSA) ? (t - TA0 + 1) * SA_PER : NSA;
                        constexpr int na = a1 > a0 ? a1 - a0 : 0;
                        constexpr int half = SB / 2;
                        if constexpr (m >= half) {
                            constexpr int q0 = a0 + (m - half) * na / (SB - half), q1 = a0 + (m - half + 1) * na / (SB - half);
                            if (!Y3_ABL(2))
                                y3_for_each_ic(std::make_integer_sequence<int, q1 - q0>{}, [&](auto D) { patch_sub(std::integral_constant<int, q0 + decltype(D)::value>{}, cr ^ 1); });
                        }
                    }
                } else {
                    constexpr int q = m - SB;
                    if constexpr (q == 0) set_addr(G{}, std::integral_constant<int, tn>{});
                    if constexpr (q < RS2) {
                        y3_for_each_ic(std::make_integer_sequence<int, 2>{}, [&](auto D) {
                            constexpr int r = q * 2 + decltype(D)::value;
                            if constexpr (r < MB)
                                read_a(G{}, C0{}, r, crn);
                            else if constexpr (r < MB + NB)
                                read_b(G{}, C0{}, r - MB, cur ^ 1);
                            else if constexpr (r < NR2)
                                read_b(G{}, C1{}, r - MB - NB, cur ^ 1);
                        });
                    } else if constexpr (q < RS2 + B_LOADS) {
                        if (!Y3_ABL(1 | 32)) b_load_one(RS{}, std::integral_constant<int, q - RS2>{}, srel + 4, std::integral_constant<int, (u + 4) % NT>{});
                    } else if constexpr (t == 0 && q < RS2 + B_LOADS + (A4 + 1) / 2) {
                        if (!Y3_ABL(1 | 16))
                            y3_for_each_ic(std::make_integer_sequence<int, 2>{}, [&](auto D) {
                                constexpr int jj = (q - RS2 - B_LOADS) * 2 + decltype(D)::value;
                                if constexpr (jj < A4) patch_load_one(std::integral_constant<int, jj>{}, cp + cr + 1);
                            });
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        // prologue: zero rows, the first chunk's patch and the first weight tile
        if (tid < 12) {       // the all-zero row of both patch buffers, every piece and k half: 16 bytes each
            const int pbuf = tid / 6, pl = tid % 6;
            *reinterpret_cast<uint4*>(&lds[pbuf * AP + pl * AHP + PR * 8]) = make_uint4(0u, 0u, 0u, 0u);
        }
        patch_load(0);
        y3_for_each_ic(std::make_integer_sequence<int, B_LOADS>{}, [&](auto E) { b_load_one(C0{}, E, 0, C0{}); });
        y3_for_each_ic(std::make_integer_sequence<int, NSA>{}, [&](auto E) { patch_sub(E, 0); });
        y3_for_each_ic(std::make_integer_sequence<int, NSB>{}, [&](auto E) { b_sub(C0{}, E, 0); });
        __syncthreads();
        Y3_TSTAMP(1);
        __builtin_amdgcn_sched_barrier(0);       // (pinned order: the set the loop consumes first must be the older one, see conv_x3_body)
        y3_for_each_ic(std::make_integer_sequence<int, B_LOADS>{}, [&](auto E) { b_load_one(C1{}, E, 1, C1{}); });
        __builtin_amdgcn_sched_barrier(0);
        y3_for_each_ic(std::make_integer_sequence<int, B_LOADS>{}, [&](auto E) { b_load_one(C2{}, E, 2, C2{}); });
        __builtin_amdgcn_sched_barrier(0);
        y3_for_each_ic(std::make_integer_sequence<int, B_LOADS>{}, [&](auto E) { b_load_one(C0{}, E, 3, std::integral_constant<int, 3>{}); });
        __builtin_amdgcn_sched_barrier(0);
        set_addr(C0{}, C0{});
#pragma unroll
        for (int i = 0; i < MB; ++i) read_a(C0{}, C0{}, i, 0);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            read_b(C0{}, C0{}, j, 0);
            read_b(C0{}, C1{}, j, 0);
        }
        for (int cp = 0; cp < nchunks; cp += 2)
            y3_for_each_ic(std::make_integer_sequence<int, 2 * NT>{}, [&](auto U) { step(U, cp); });
    }
    Y3_TSTAMP(2);
    conv_fast_finish<BM, BN, WM, WN, true, BNS, true>(p, fw, acc, red);
}

template <int BM, int BN, int WM, int WN, bool BNS>
__global__ __launch_bounds__(64 * WM * WN, 2) void conv_x3p_kernel(const FastArgs p) {
    conv_x3p_body<BM, BN, WM, WN, BNS>(p, (int)blockIdx.x, (int)gridDim.x);
}

// registers: two waves per SIMD (64 accumulators + 56 fragment + 32 staging registers per lane)
template <int BM, int BN, int WM, int WN, bool DENSE, bool BNS>
__global__ __launch_bounds__(64 * WM * WN, 2) void conv_x3_kernel(const FastArgs p) {
    conv_x3_body<BM, BN, WM, WN, DENSE, BNS>(p, (int)blockIdx.x, (int)gridDim.x);
}

// The four parity classes of a stride-2 data gradient in one launch (conv.hip: conv_igemm_fast_multi_kernel), on the x3
// arithmetic: every class brings its own K slices, tickets and slabs.
template <int BM, int BN, int WM, int WN, bool BNS>
__global__ __launch_bounds__(64 * WM * WN, 2) void conv_x3_multi_kernel(const FastArgs4 m) {
    int c = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) c += ((int)blockIdx.x >= m.first[i]) ? 1 : 0;
    conv_x3_body<BM, BN, WM, WN, false, BNS>(m.a[c], (int)blockIdx.x - m.first[c], m.first[c + 1] - m.first[c]);
}
bool y3_x3_multi_launch(const FastArgs4& m, int bn, bool bns, int grid, hipStream_t st) {
    const dim3 g(grid);
    if (bn == 128) {
        if (bns)
            hipLaunchKernelGGL((conv_x3_multi_kernel<128, 128, 2, 2, true>), g, dim3(256), 0, st, m);
        else
            hipLaunchKernelGGL((conv_x3_multi_kernel<128, 128, 2, 2, false>), g, dim3(256), 0, st, m);
    } else if (bn == 64) {
        if (bns)
            hipLaunchKernelGGL((conv_x3_multi_kernel<128, 64, 2, 1, true>), g, dim3(128), 0, st, m);
        else
            hipLaunchKernelGGL((conv_x3_multi_kernel<128, 64, 2, 1, false>), g, dim3(128), 0, st, m);
    } else {
        return false;
    }
    return true;
}

template <int BM, int BN, int WM, int WN>
static void x3_launch_tile(const FastArgs& p, bool dense, int grid, hipStream_t st) {
    const dim3 g(grid), b(64 * WM * WN);
    if (p.bn_a)
        hipLaunchKernelGGL((conv_x3_kernel<BM, BN, WM, WN, true, true>), g, b, 0, st, p);
    else if (dense)
        hipLaunchKernelGGL((conv_x3_kernel<BM, BN, WM, WN, true, false>), g, b, 0, st, p);
    else
        hipLaunchKernelGGL((conv_x3_kernel<BM, BN, WM, WN, false, false>), g, b, 0, st, p);
}

// ---------------------------------------------------------------------------
// Kernel gradient on the same arithmetic:  dw[k][n] = sum_m A[m + off(tap)][c] * dz[m][n],  k = (tap, c).
// The contraction runs over PIXELS, so both operands are activations (both are split in the kernel) and the MFMA wants, per
// lane, 8 consecutive pixels of ONE channel -- the transpose of how NHWC data arrives.  The tiles are stored as they come,
// [16 pixels][128 channels] bf16 per piece (256-byte rows, 16-byte chunks XOR-swizzled by the row: cdna_hip_programming.md T10
// image (b)), and read with ds_read_b64_tr_b16, which hands every lane 4 pixels of its channel: two reads = one fragment.
// One step = 16 pixels = one v_mfma_f32_32x32x16_bf16 per piece pair and block.  Everything around the loop -- the pixel table,
// the split over pixel runs, the two reduction forms -- is conv_wgrad_kernel's (conv.hip).
// Loop body: SIX steps (tile s: LDS buffer s & 1, register set s % 3: loads four steps ahead of their stores' step).
// ---------------------------------------------------------------------------
typedef short y3_s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ y3_bf16x8 x3_tr_pair(const unsigned short* a0, const unsigned short* a1) {
    const y3_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((y3_s16x4 __attribute__((address_space(3)))*)(a0));
    const y3_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((y3_s16x4 __attribute__((address_space(3)))*)(a1));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(y3_bf16x8, v);
}

template <int BKR, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN, 2) void conv_wgrad_x3_kernel(const WgradArgs p) {
    constexpr int BP = 16;
    constexpr int THREADS = 64 * WM * WN;
    constexpr int TM = BKR / WM, TN = BN / WN, MB = TM / 32, NB = TN / 32;
    constexpr int KR4 = BKR / 4, BN4 = BN / 4;
    static_assert(BKR == 128 && BN == 128 && THREADS == 256, "the swizzled [16][128] images and the loader are written for 128 x 128 tiles");
    constexpr int A_LOADS = BP * KR4 / THREADS, B_LOADS = BP * BN4 / THREADS;      // 2 + 2
    constexpr int PSTEP = THREADS / KR4;                                             // 8: pixel distance between a thread's loads
    constexpr int PLANE = BP * 128;                  // u16 per piece plane (16 rows of 256 bytes)
    constexpr int OPB = 3 * PLANE;                   // one operand, three pieces
    constexpr int BUF = 2 * OPB;                     // A then B
    __shared__ __attribute__((aligned(16))) unsigned short lds[2 * BUF];
    __shared__ uint2 pix[Y3_WG_TABLE];               // per pixel of this split: {byte offset of its (dh, dw) = (0, 0) source pixel, tap validity bits}

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
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
    if (splits >= 32) {       // (block -> (pixel run, tile): conv_wgrad_kernel)
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
    // loaders: thread -> (pixel tid / 32 (+ 8 per load), channel quad tid % 32) of both [16][128] tiles
    const int a_kv = tid % KR4;
    const int ak = k0 + a_kv * 4;
    const bool ak_ok = ak < p.K;
    const int atap = ak_ok ? (ak >> p.logC) : 0;
    const int ac = ak & p.cmask;
    const int acode = (int)((tap_dhdw >> (4 * atap)) & 15ull);
    const int a_tapoff = ((((acode & 3) - 1) * p.W + ((acode >> 2) - 1)) * p.src_ld + ac) * 4;
    const unsigned a_bit = ak_ok ? 1u << atap : 0u;
    const int pp0 = tid / KR4;
    const int bnn = n0 + a_kv * 4;
    const bool bn_ok = bnn < p.Nout;
    const unsigned b_lane = (unsigned)(bnn * 4);
    // LDS image of a tile: 256-byte pixel rows, 16-byte chunk ch of row r at 16 * (ch ^ (((r & 3) << 2) | ((r >> 2) & 3)))
    auto swz = [](int row) { return ((row & 3) << 2) | ((row >> 2) & 3); };
    int st_off[A_LOADS];       // u16 index of this thread's 8-byte store slot (same for both operands)
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
        const int row = pp0 + i * PSTEP;
        st_off[i] = row * 128 + 8 * ((a_kv >> 1) ^ swz(row)) + 4 * (a_kv & 1);
    }
    // transposed fragment reads: 16-lane group g = lane / 16 takes pixels 8 (g / 2) + 4 j .. + 3 of channels 16 (g & 1) .. + 15 of
    // its 32-channel block; lane 4 q + pc of the group supplies row q, columns 4 pc .. 4 pc + 3
    int fr_a[MB][2], fr_b[NB][2];
    {
        const int g = lane >> 4, li = lane & 15, q = li >> 2, pc = li & 3, h = g >> 1;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = 8 * h + 4 * j + q;
#pragma unroll
            for (int i = 0; i < MB; ++i) {
                const int c = wm * TM + i * 32 + 16 * (g & 1) + 4 * pc;
                fr_a[i][j] = row * 128 + 8 * ((c >> 3) ^ swz(row)) + (c & 4);
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int c = wn * TN + i * 32 + 16 * (g & 1) + 4 * pc;
                fr_b[i][j] = row * 128 + 8 * ((c >> 3) ^ swz(row)) + (c & 4);
            }
        }
    }

    f32x4 ra[3][A_LOADS], rb[3][B_LOADS];
    uint2 pe[A_LOADS];
    auto tload = [&](int step) {
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) pe[i] = pix[step * BP + pp0 + i * PSTEP];
    };
    auto gload_a = [&](auto I, int step, auto S) {
        constexpr int i = decltype(I)::value;
        const bool ok = ((pe[i].y & a_bit) != 0) & (step < nsteps);
        ra[decltype(S)::value][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_src, ok ? pe[i].x + (unsigned)a_tapoff : Y3_OOB, 0, 0);
    };
    auto gload_b = [&](auto I, int step, auto S) {
        constexpr int i = decltype(I)::value;
        const int m = mbeg + step * BP + pp0 + i * PSTEP;
        const bool ok = (m < mend) & bn_ok;
        rb[decltype(S)::value][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_dd, ok ? (unsigned)(m * p.dd_ld) * 4u + b_lane : Y3_OOB, 0, 0);
    };
    X3Pk pk = {0, 0};
    auto sub = [&](auto S, auto E, int buf) {      // sub-step E (0 .. 15) of splitting tile set S into LDS buffer buf
        constexpr int set = decltype(S)::value, e = decltype(E)::value, w = e / 4, sb = e % 4;
        if constexpr (w < A_LOADS) {
            unsigned short* d = &lds[buf * BUF + st_off[w]];
            x3_split_sub<sb>(ra[set][w], pk, d, d + PLANE, d + 2 * PLANE);
        } else {
            unsigned short* d = &lds[buf * BUF + OPB + st_off[w - A_LOADS]];
            x3_split_sub<sb>(rb[set][w - A_LOADS], pk, d, d + PLANE, d + 2 * PLANE);
        }
    };
    y3_bf16x8 FA[2][3][MB], FB[2][3][NB];
    auto read_a = [&](auto F, auto PC, int i, int buf) {
        const unsigned short* b = &lds[buf * BUF + decltype(PC)::value * PLANE];
        FA[decltype(F)::value][decltype(PC)::value][i] = x3_tr_pair(b + fr_a[i][0], b + fr_a[i][1]);
    };
    auto read_b = [&](auto F, auto PC, int j, int buf) {
        const unsigned short* b = &lds[buf * BUF + OPB + decltype(PC)::value * PLANE];
        FB[decltype(F)::value][decltype(PC)::value][j] = x3_tr_pair(b + fr_b[j][0], b + fr_b[j][1]);
    };

    f32x16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    using C2 = std::integral_constant<int, 2>;
    constexpr int NW = A_LOADS + B_LOADS;
    __syncthreads();                       // pixel table complete
    if (nsteps > 0) {
        constexpr int NMG = MB * NB, NM = 6 * NMG, SB = Y3_X3_GB * NMG, SP = NM - SB;
        constexpr int NR1 = NB + MB + MB, NS = NW * 4;
        constexpr int NR2 = MB + NB + NB, RS2 = (NR2 + 1) / 2;
        static_assert(NR1 <= SB && RS2 + (NW + 1) / 2 + 1 <= SP, "not enough MFMA slots for the events of a step");
        auto step = [&](auto U, int s6) {       // step U of the six-step body; s6 = first step of the body
            constexpr int u = decltype(U)::value, cur = u & 1;
            using F = std::integral_constant<int, cur>;
            using G = std::integral_constant<int, cur ^ 1>;
            using RS = std::integral_constant<int, (u + 1) % 3>;      // register set stored (tile u + 1) and reloaded (tile u + 4) in this step
            const int st = s6 + u;
            y3_for_each_ic(std::make_integer_sequence<int, NM>{}, [&](auto Mi) {
                constexpr int m = decltype(Mi)::value;
                constexpr int g = m / NMG, i = (m % NMG) / NB, j = m % NB;
                constexpr int pa = g < 3 ? 0 : (g < 5 ? 1 : 2), pb = g < 3 ? g : (g == 3 ? 0 : (g == 4 ? 1 : 0));
                if constexpr (m == SB) {
                    y3_lds_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[cur][pa][i], FB[cur][pb][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (m < SB) {
                    if constexpr (m < NB)
                        read_b(F{}, C2{}, m, cur);
                    else if constexpr (m < NB + MB)
                        read_a(F{}, C1{}, m - NB, cur);
                    else if constexpr (m < NR1)
                        read_a(F{}, C2{}, m - NB - MB, cur);
                    constexpr int s0 = m * NS / SB, s1 = (m + 1) * NS / SB;
                    y3_for_each_ic(std::make_integer_sequence<int, s1 - s0>{}, [&](auto D) { sub(RS{}, std::integral_constant<int, s0 + decltype(D)::value>{}, cur ^ 1); });
                } else {
                    constexpr int q = m - SB;
                    if constexpr (q < RS2) {
                        y3_for_each_ic(std::make_integer_sequence<int, 2>{}, [&](auto D) {
                            constexpr int r = q * 2 + decltype(D)::value;
                            if constexpr (r < MB)
                                read_a(G{}, C0{}, r, cur ^ 1);
                            else if constexpr (r < MB + NB)
                                read_b(G{}, C0{}, r - MB, cur ^ 1);
                            else if constexpr (r < NR2)
                                read_b(G{}, C1{}, r - MB - NB, cur ^ 1);
                        });
                    } else if constexpr (q < RS2 + (NW + 1) / 2) {
                        y3_for_each_ic(std::make_integer_sequence<int, 2>{}, [&](auto D) {
                            constexpr int e = (q - RS2) * 2 + decltype(D)::value;
                            if constexpr (e < A_LOADS)
                                gload_a(std::integral_constant<int, e>{}, st + 4, RS{});
                            else if constexpr (e < NW)
                                gload_b(std::integral_constant<int, e - A_LOADS>{}, st + 4, RS{});
                        });
                    } else if constexpr (q == RS2 + (NW + 1) / 2) {
                        tload(min(st + 5, nsteps - 1));       // table entries of the next step's loads
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        auto load_tile = [&](int step, auto S) {
            tload(min(step, nsteps - 1));
            y3_for_each_ic(std::make_integer_sequence<int, A_LOADS>{}, [&](auto I) { gload_a(I, step, S); });
            y3_for_each_ic(std::make_integer_sequence<int, B_LOADS>{}, [&](auto I) { gload_b(I, step, S); });
        };
        load_tile(0, C0{});
        y3_for_each_ic(std::make_integer_sequence<int, NW * 4>{}, [&](auto E) { sub(C0{}, E, 0); });
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);       // (pinned order: the set the loop consumes first is the oldest, see conv_x3_body)
        load_tile(1, C1{});
        __builtin_amdgcn_sched_barrier(0);
        load_tile(2, C2{});
        __builtin_amdgcn_sched_barrier(0);
        load_tile(3, C0{});
        __builtin_amdgcn_sched_barrier(0);
        tload(min(4, nsteps - 1));
#pragma unroll
        for (int i = 0; i < MB; ++i) read_a(C0{}, C0{}, i, 0);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            read_b(C0{}, C0{}, j, 0);
            read_b(C0{}, C1{}, j, 0);
        }
        // six uniform steps per iteration; steps beyond the split's pixels are dead (their loads return zeros)
        for (int s6 = 0; s6 < nsteps; s6 += 6)
            y3_for_each_ic(std::make_integer_sequence<int, 6>{}, [&](auto U) { step(U, s6); });
    }
    __syncthreads();                       // (the reduction below reuses the stage as a flag word)

    constexpr int R4 = MB * NB * 4;
    const int l31 = lane & 31, lh = lane >> 5;
    if (p.splits > 1 && p.tickets != nullptr) {
        // in-kernel reduction over <= Y3_WG_FANIN pixel runs: conv_wgrad_kernel's hand-off (slab[tile][split][r4][thread], tickets)
        const __amdgpu_buffer_rsrc_t rs_slab = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, 0x7ffffff0, 0x00020000);
        const unsigned item_bytes = (unsigned)(R4 * THREADS * 16);
        int* flag = reinterpret_cast<int*>(&lds[0]);
        const int count = p.splits;
        const unsigned level0 = (unsigned)(bid * count) * item_bytes + (unsigned)tid * 16u;
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

bool y3_wgrad_x3_launch(const WgradArgs& p, int bkr, int bn, unsigned grid, hipStream_t st) {
    if (bkr != 128 || bn != 128) return false;
    hipLaunchKernelGGL((conv_wgrad_x3_kernel<128, 128, 2, 2>), dim3(grid), dim3(256), 0, st, p);
    return true;
}

// ---------------------------------------------------------------------------
// Weight planes for the kernels above.  Input: a kernel with K contiguous per row, w[tap][row][C] fp32 (forward: row = output
// channel, C = Cin -- the transposed copy; data gradient: row = input channel, C = Cout -- the Keras copy).  Output, bf16:
//     planes[(((tap * C/16 + c/16) * rows + row) * 3 + piece) * 16 + c % 16] = piece `piece` of w[tap][row][c]
// (x = x0 + x1 + x2 exactly, each piece the round-to-nearest bf16 of what the earlier ones leave): the rows x 3 x 16 block one K
// step reads is contiguous.  HBM-bound pass, once per optimiser step over both copies (batched: every layer in one launch).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void x3_split_store(const float* __restrict__ w, unsigned short* __restrict__ planes, long long e, int rows, int C) {
    // e: flat index of 4 consecutive c of one (tap, row)
    const long long tr = e / C;                 // tap * rows + row
    const int c = (int)(e - tr * C);
    const long long tap = tr / rows;
    const int row = (int)(tr - tap * rows);
    f32x4 r = *reinterpret_cast<const f32x4*>(w + e);
    const unsigned a0 = x3_pk(r[0], r[1]), a1 = x3_pk(r[2], r[3]);
    r[0] -= x3_lo(a0); r[1] -= x3_hi(a0); r[2] -= x3_lo(a1); r[3] -= x3_hi(a1);
    const unsigned b0 = x3_pk(r[0], r[1]), b1 = x3_pk(r[2], r[3]);
    r[0] -= x3_lo(b0); r[1] -= x3_hi(b0); r[2] -= x3_lo(b1); r[3] -= x3_hi(b1);
    unsigned short* d = planes + (((tap * (C >> 4) + (c >> 4)) * rows + row) * 3) * 16 + (c & 15);
    *reinterpret_cast<uint2*>(d) = make_uint2(a0, a1);
    *reinterpret_cast<uint2*>(d + 16) = make_uint2(b0, b1);
    *reinterpret_cast<uint2*>(d + 32) = make_uint2(x3_pk(r[0], r[1]), x3_pk(r[2], r[3]));
}
__global__ __launch_bounds__(256) void x3_split_weights_kernel(const float* __restrict__ w, unsigned short* __restrict__ planes, long long count, int rows, int C) {
    const long long e = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e < count) x3_split_store(w, planes, e, rows, C);
}
// table[l] = {arena offset of the layer's kernel (floats), taps, rows, C, first block}; the layer's planes start at 3 * offset (bf16 elements)
__global__ __launch_bounds__(256) void x3_split_weights_batched_kernel(const float* __restrict__ arena, unsigned short* __restrict__ planes_arena,
                                                                       const int* __restrict__ table, int nlayers) {
    int lo = 0, hi = nlayers - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid * 5 + 4] <= (int)blockIdx.x)
            lo = mid;
        else
            hi = mid - 1;
    }
    const long long off = table[lo * 5];
    const int taps = table[lo * 5 + 1], rows = table[lo * 5 + 2], C = table[lo * 5 + 3];
    const long long count = (long long)taps * rows * C;
    const long long e = ((long long)((int)blockIdx.x - table[lo * 5 + 4]) * 256 + threadIdx.x) * 4;
    if (e < count) x3_split_store(arena + off, planes_arena + 3 * off, e, rows, C);
}
extern "C" int y3_x3_split_weights(const float* w, void* planes, int taps, int rows, int k_per_row, y3_stream_t stream) {
    Y3_CHECK_ARG(w && planes && taps > 0 && rows > 0 && k_per_row > 0 && k_per_row % 16 == 0, "x3_split_weights: null pointer, or K per row (%d) not a multiple of 16", k_per_row);
    Y3_CHECK_ARG((((uintptr_t)w) & 15) == 0 && (((uintptr_t)planes) & 15) == 0, "x3_split_weights: w and planes must be 16-byte aligned");
    const long long count = (long long)taps * rows * k_per_row;
    hipLaunchKernelGGL(x3_split_weights_kernel, dim3((unsigned)((count / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, (unsigned short*)planes, count, rows, k_per_row);
    Y3_CHECK_LAUNCH("x3_split_weights");
    return Y3_OK;
}
extern "C" int y3_x3_split_weights_batched(const float* arena, void* planes_arena, const int* table_dev, int nlayers, int total_blocks, y3_stream_t stream) {
    Y3_CHECK_ARG(arena && planes_arena && table_dev && nlayers > 0 && total_blocks > 0, "x3_split_weights_batched: bad args");
    hipLaunchKernelGGL(x3_split_weights_batched_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, arena, (unsigned short*)planes_arena, table_dev, nlayers);
    Y3_CHECK_LAUNCH("x3_split_weights_batched");
    return Y3_OK;
}

// Everything the optimiser step leaves to refresh, in ONE pass over the parameter arena: the transposed fp32 copy (operand of the
// fp32 data gradients, y3_transpose_weights_batched's job) and the piece planes of both copies (the two y3_x3_split_weights_batched
// launches) -- the arena is read once instead of three times and the transposed copy is not read back.  A block owns one 32 x 32
// (Cin x Cout) tile of one tap of one layer; table[l] = {arena offset, taps, cin, cout, first tile} as for the transpose.  The
// planes of a copy are written for the layers whose K per row is a multiple of 16 (the x3 kernels take no others).
__device__ __forceinline__ void x3_split4(f32x4 r, unsigned short* d) {      // d: piece 0 of the four k; pieces 1, 2 follow 16 and 32 bf16 further
    const unsigned a0 = x3_pk(r[0], r[1]), a1 = x3_pk(r[2], r[3]);
    r[0] -= x3_lo(a0); r[1] -= x3_hi(a0); r[2] -= x3_lo(a1); r[3] -= x3_hi(a1);
    const unsigned b0 = x3_pk(r[0], r[1]), b1 = x3_pk(r[2], r[3]);
    r[0] -= x3_lo(b0); r[1] -= x3_hi(b0); r[2] -= x3_lo(b1); r[3] -= x3_hi(b1);
    *reinterpret_cast<uint2*>(d) = make_uint2(a0, a1);
    *reinterpret_cast<uint2*>(d + 16) = make_uint2(b0, b1);
    *reinterpret_cast<uint2*>(d + 32) = make_uint2(x3_pk(r[0], r[1]), x3_pk(r[2], r[3]));
}
__global__ __launch_bounds__(256) void x3_prepare_weights_kernel(const float* __restrict__ params, float* __restrict__ params_t, unsigned short* __restrict__ planes,
                                                                 unsigned short* __restrict__ planes_t, const int* __restrict__ table, int nlayers) {
    __shared__ float tile[32][33];
    int lo = 0, hi = nlayers - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid * 5 + 4] <= (int)blockIdx.x)
            lo = mid;
        else
            hi = mid - 1;
    }
    const long long off = table[lo * 5];
    const int cin = table[lo * 5 + 2], cout = table[lo * 5 + 3];
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
    const int row = threadIdx.x >> 3, q = (threadIdx.x & 7) * 4;
    if ((cout & 15) == 0) {      // planes of the Keras copy [tap][Cin][Cout]: row = input channel, k = output channel
        const int ci = ci0 + row, co = co0 + q;
        if (ci < cin && co < cout) {
            const f32x4 v = {tile[row][q], tile[row][q + 1], tile[row][q + 2], tile[row][q + 3]};
            x3_split4(v, planes + 3 * off + ((((long long)t * (cout >> 4) + (co >> 4)) * cin + ci) * 3) * 16 + (co & 15));
        }
    }
    if ((cin & 15) == 0) {       // planes of the transposed copy [tap][Cout][Cin]: row = output channel, k = input channel
        const int co = co0 + row, ci = ci0 + q;
        if (co < cout && ci < cin) {
            const f32x4 v = {tile[q][row], tile[q + 1][row], tile[q + 2][row], tile[q + 3][row]};
            x3_split4(v, planes_t + 3 * off + ((((long long)t * (cin >> 4) + (ci >> 4)) * cout + co) * 3) * 16 + (ci & 15));
        }
    }
}
extern "C" int y3_x3_prepare_weights_batched(const float* params, float* params_t, void* planes, void* planes_t, const int* table_dev, int nlayers,
                                             int total_tiles, y3_stream_t stream) {
    Y3_CHECK_ARG(params && params_t && planes && planes_t && table_dev && nlayers > 0 && total_tiles > 0, "x3_prepare_weights_batched: bad args");
    hipLaunchKernelGGL(x3_prepare_weights_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, params, params_t, (unsigned short*)planes,
                       (unsigned short*)planes_t, table_dev, nlayers);
    Y3_CHECK_LAUNCH("x3_prepare_weights_batched");
    return Y3_OK;
}

// Tiles this file is built for (conv.hip plans with them): false if (bm, bn) is not one of them.
bool y3_x3_tile_ok(int bm, int bn) { return bm == 128 && (bn == 128 || bn == 64); }

// The patch kernel takes a launch when: 3x3 stride 1 with the standard tap grid (dense destination), the patch fits the LDS rows
// (BM + 2 (W + 1) <= Y3_X3P_ROWS), and every K slice is a whole, even number of 16-channel chunks (18 K steps).
bool y3_x3p_ok(const FastArgs& p, int bm, int bn, bool dense) {
    if (!(bm == 128 && bn == 128) || !dense || p.ntaps != 9 || p.sh != 1 || p.sw != 1) return false;
    for (int t = 0; t < 9; ++t)
        if (p.tap_dh[t] < -1 || p.tap_dh[t] > 1 || p.tap_dw[t] < -1 || p.tap_dw[t] > 1) return false;
    if (bm + 2 * (p.W + 1) > Y3_X3P_ROWS) return false;
    bool biased = false;                                                // src biased by exactly -(W + 1) pixels: tap (-1, -1) at offset 0
    for (int t = 0; t < 9; ++t) biased |= p.tap_dh[t] == -1 && p.tap_dw[t] == -1 && p.tap_off[t] == 0;
    if (!biased) return false;
    if ((p.Cper % 32) != 0 || (p.K / 16) % 18 != 0) return false;
    if ((p.sk_s0 > 1 && p.sk_chunk0 % 18 != 0) || (p.sk_s1 > 1 && p.sk_chunk1 % 18 != 0)) return false;
    return true;
}

bool y3_x3_launch(const FastArgs& p, int bm, int bn, bool dense, int grid, hipStream_t st) {
#ifdef Y3_DEV
    static const int no_patch = getenv("Y3_X3_NO_PATCH") ? atoi(getenv("Y3_X3_NO_PATCH")) : 0;      // development: the im2col-order kernel for every launch
#else
    const int no_patch = 0;
#endif
    if (!no_patch && y3_x3p_ok(p, bm, bn, dense)) {
        const dim3 g(grid), b(256);
        if (p.bn_a)
            hipLaunchKernelGGL((conv_x3p_kernel<128, 128, 2, 2, true>), g, b, 0, st, p);
        else
            hipLaunchKernelGGL((conv_x3p_kernel<128, 128, 2, 2, false>), g, b, 0, st, p);
        return true;
    }
    if (bm == 128 && bn == 128)
        x3_launch_tile<128, 128, 2, 2>(p, dense, grid, st);
    else if (bm == 128 && bn == 64)
        x3_launch_tile<128, 64, 2, 1>(p, dense, grid, st);
    else
        return false;
    return true;
}
