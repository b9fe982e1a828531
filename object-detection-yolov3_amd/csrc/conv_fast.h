// Shared pieces of the gather-GEMM "fast path" (conv.hip: fp32 matrix instructions; conv_x3.hip: fp32 arithmetic as three bf16
// pieces per operand on the bf16 matrix pipe): the kernel arguments, the work-item decode (tile, K slice, split-K bookkeeping)
// and everything after the K loop -- the in-kernel split-K reduction and the epilogue.  Both kernels produce their accumulators
// in the C/D layout of the 32x32 MFMA shapes (col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)), which is the same
// for v_mfma_f32_32x32x2_f32 and v_mfma_f32_32x32x16_bf16, so the code below serves both.
#pragma once
#include <type_traits>
#include <utility>

#include "common.h"

// Fast path of the same gather-GEMM (taken when C % BK == 0, i.e. every layer except the first conv; the 14-channel
// heads included -- see make_fast): a K step never straddles a tap, so the tap and the channel
// base are wave-uniform and travel in the scalar offset of buffer loads; per-lane offsets are loop
// invariant; padding / tile-edge lanes are pointed past the descriptor's range and read zeros from the
// hardware bounds check instead of branching.
// ---------------------------------------------------------------------------
struct FastArgs {
    const float* src;  // biased so that every tap offset is >= 0
    const float* wt;   // [tap][C][Nout]; the x3 kernels (conv_x3.hip): the bf16 piece planes y3_x3_split_weights makes of the K-contiguous copy
    int Cper;          // channels per tap (K = ntaps * Cper)
    int x3_mode;       // x3 kernels: bit 0 = non-temporal activation loads, bit 3 = short-last deal of the items (conv_fast_decode)
    // Order of the K steps of a multi-tap launch.  korder 2: 32-channel group (one 128-byte line per pixel) outermost, taps inside,
    // the two 16-channel halves of the line innermost: the taps of a group re-read the same few image rows, and so do the
    // neighbouring row tiles -- all within 2 * ntaps steps, while the rows are still in L2.  korder 0 (taps outermost) spaces the
    // three uses of an image row a third of a workgroup's life apart: the big early layers fetched their input 3-4x (rocprofv3
    // FETCH_SIZE, profiles/r04_traffic_by_kernel.txt).  korder 1 (16-channel chunks outermost; C % 32 != 0) splits the two halves
    // of a line by ntaps steps: worse than 0 for the 32-channel layers.  dv_taps divides by the steps of a group (ntaps << (korder - 1)).
    int korder;
    Y3Div dv_taps;
    float* dst;
    const float* bias;
    const float* scale;
    const float* shift;
    const float* resid;
    float* stats;
    int tap_off[9];   // byte offset of tap t from the pixel base (biased, >= 0)
    int tap_wrow[9];  // weight row of tap t, channel 0
    int tg_nx, tg_off0, tg_offy, tg_offx, tg_w0, tg_wy, tg_wx;   // the same two tables as affine maps of the tap grid (tap = ty * tg_nx + tx)
    int tg_mul;          // tap / tg_nx = (tap * tg_mul) >> 5 for tap < 9: 32, 16, 11 for tg_nx = 1, 2, 3 (no branch in the K loop)
    int tap_dh[9], tap_dw[9];
    unsigned src_bytes, wt_bytes, dst_bytes, resid_bytes;  // extents for the buffer descriptors
    int ntaps;
    int H, W, logC, cmask, src_ld;
    int OH, OW, sh, sw;
    int DH, DW, dsh, dsw, doh, dow, dst_ld;
    int resid_ld;
    int Nout, K, M;
    unsigned flags;
    float alpha;
    int nbn, nbm, col_major;
    // divisions of the index decode as multiply-high + shift (y3_make_div): tile id by the fastest-varying tile count, work
    // item by the slice counts, output pixel by OH*OW and by OW
    int nb_fast, ohw;
    Y3Div dv_nb, dv_s0, dv_s1, dv_ohw, dv_ow;
    // Split-K with the reduction inside the kernel.  Work items: tiles [0, sk_f) are cut into sk_s0 K slices each, tiles
    // [sk_f, tiles) into sk_s1 (the remainder of a launch whose tile count is not a multiple of the CU count is split
    // finer so that every CU ends up with the same amount of MFMA work).  Item i < sk_n0 = sk_f * sk_s0 is slice
    // i % sk_s0 of tile i / sk_s0; item i >= sk_n0 is slice (i - sk_n0) % sk_s1 of tile sk_f + (i - sk_n0) / sk_s1.  A slice
    // covers sk_chunk{0,1} K steps.  Slices of a split tile park their raw accumulators in `slab` (item-major, fragment
    // order) and take a ticket; the slice that draws the last ticket re-reads ALL of them in slice order (fixed order ->
    // bit-reproducible), runs the normal epilogue and leaves the ticket at zero for the next launch.
    int sk_f, sk_n0, sk_s0, sk_s1, sk_chunk0, sk_chunk1;
    int sk_slab0;        // first item that owns a slab slot (0, or sk_n0 when only the remainder tiles are split)
    float* slab;
    int* tickets;        // one per tile, zero before the launch
    const float* bn_a;   // BNS kernels: activation of the BatchNorm layer whose output gradient this launch completes (dst geometry)
    float* bn_part;      // BNS kernels: [row tile][6][Nout] partial raw moments of (dst, bn_a), see bn_bwd_stats_kernel
    unsigned bn_a_bytes;
    int bn_a_ld;
    int bn_row0;         // BNS: first partial row of this launch / parity class (rows are bn_row0 + row tile)
};

#define Y3_OOB 0x80000000u

// Development instrumentation (tools/probe/conv_timing.hip builds this file with -DY3_TIMING): per-workgroup s_memtime
// stamps of the kernel phases + the CU the workgroup ran on.  Compiled out of the product library.
#ifdef Y3_TIMING
__device__ int y3_abl_dev = 0;   // ablation mask for the probe: 1 = no global loads in the K loop, 2 = no LDS stores, 4 = no barrier
// read ONCE per workgroup into an SGPR (Y3_ABL_INIT at the top of the kernel body): read inside the loop, the word was re-fetched
// after every barrier and the probe timed that
#define Y3_ABL_INIT() const int y3_abl = __builtin_amdgcn_readfirstlane(y3_abl_dev)
#define Y3_ABL(bit) (y3_abl & (bit))
__device__ unsigned long long* y3_timing_buf = nullptr;
#define Y3_TSTAMP(i)                                                                                                    \
    do {                                                                                                                \
        if (y3_timing_buf && threadIdx.x == 0) y3_timing_buf[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define Y3_TSTAMP(i)
#define Y3_ABL_INIT()
#define Y3_ABL(bit) 0
#endif

template <int... I, class F>
__device__ __forceinline__ void y3_for_each_ic(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
// Instruction-mix directives for the machine scheduler (masks: 0x008 MFMA, 0x020 VMEM read, 0x100 DS read, 0x200 DS write)
template <int MASK, int N>
__device__ __forceinline__ void y3_sgb() {
    if constexpr (N > 0) __builtin_amdgcn_sched_group_barrier(MASK, N, 0);
}
template <int COUNT, int MASK, int MFMAS = 1>
__device__ __forceinline__ void y3_sgb_pairs() {   // COUNT x { MFMAS matrix instructions, then one instruction of MASK }
    if constexpr (COUNT > 0) {
        y3_sgb<0x008, MFMAS>();
        y3_sgb<MASK, 1>();
        y3_sgb_pairs<COUNT - 1, MASK, MFMAS>();
    }
}

// kernel gradient (conv.hip: conv_wgrad_kernel; conv_x3.hip: conv_wgrad_x3_kernel)
#define Y3_WG_FANIN 8   // kernel-gradient slab reduction: fan-in of the in-kernel tree
struct WgradArgs {
    const float* src;
    const float* ddst;
    float* out;
    unsigned long long tap_dhdw;
    int H, W, C, logC, cmask, src_ld;
    int OH, OW, sh, sw;
    int dd_ld, Nout, K, M;
    int chunk;  // pixels per split (multiple of BP)
    int nbn, tiles, splits;
    unsigned src_bytes, dd_bytes;  // extents for the buffer descriptors (out-of-range lanes read zeros)
    int* tickets;  // splits > 1: one per (k-tile, n-tile), zero before the launch; `out` is then the slab area
    float* dw;     // final destination [K][Nout]
    int ohw, ntaps;
    Y3Div dv_tiles, dv_nbn, dv_ohw, dv_ow;   // index decode without run-time divides (y3_make_div)
};

#ifndef Y3_WG_TABLE
#define Y3_WG_TABLE 2048      // pixels per split the LDS pixel table holds (plan_wgrad keeps chunks below it)
#endif
bool y3_wgrad_x3_launch(const WgradArgs& p, int bkr, int bn, unsigned grid, hipStream_t st);

// conv_x3.hip: launch of the x3 kernel for a planned tile (false: no kernel built for it)
bool y3_x3_launch(const FastArgs& p, int bm, int bn, bool dense, int grid, hipStream_t st);
// Up to four independent gather-GEMMs in ONE launch: the (row parity, column parity) classes of a stride-2 data gradient.
// Each class has its own tap list, K, destination lattice and (x3) K slices; block ranges [first[c], first[c+1]) select the class.
struct FastArgs4 {
    FastArgs a[4];
    int first[5];
};
bool y3_x3_multi_launch(const FastArgs4& m, int bn, bool bns, int grid, hipStream_t st);
bool y3_x3_tile_ok(int bm, int bn);

// One workgroup's share of a launch, decoded from its (XCD-remapped) block index.
struct FastWork {
    int tid, lane, wave, l31, lh, wm, wn;
    int bid0, bid, kz, nz;            // work item, tile, K slice of the tile, slices of the tile
    int bm, bn, m0, n0, kbeg, kend;   // row / column tile and the K range [kbeg, kend) of this slice
    int ohw, OW, aK, aM, aH, aW, src_ld, csh, csw, ntaps, Nout;
    Y3Div dv_ohw, dv_ow;
};

// KZMAJOR (the x3 patch kernel): with column-major tile ids the items of a column are dealt K slice by K slice (row tile fastest)
// instead of tile by tile: the run of items an XCD takes then holds ALL row tiles of a few (column, K slice) pairs -- they load
// the same weight blocks step by step, so a block of the planes is fetched into one L2 instead of two or three.  Only who
// computes which item changes; item ids (slab slots, tickets) stay tile-major.  SHORTLAST (the x3 kernels): see below.
template <int BM, int BN, int WM, int WN, int BK, bool KZMAJOR = false, bool SHORTLAST = false>
__device__ __forceinline__ FastWork conv_fast_decode(const FastArgs& p, const int braw, const int grid) {
    FastWork w;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    // Every scalar of the index decode is fetched from the argument segment HERE, in one batch (the pins keep the compiler from
    // sinking each load to its first use): left alone it emitted ~17 load / wait / branch rounds of ~200 cycles each before
    // the first global load of the workgroup was issued.
    int sk_n0 = p.sk_n0, sk_s0 = p.sk_s0, sk_s1 = p.sk_s1, sk_f = p.sk_f, sk_c0 = p.sk_chunk0, sk_c1 = p.sk_chunk1;
    int col_major = p.col_major, nb_fast = p.nb_fast, ohw = p.ohw, OW = p.OW, aK = p.K, aM = p.M, aH = p.H, aW = p.W;
    int src_ld = p.src_ld, csh = p.sh, csw = p.sw, ntaps = p.ntaps, Nout = p.Nout;
    unsigned dnb_m = p.dv_nb.mul, ds0_m = p.dv_s0.mul, ds1_m = p.dv_s1.mul, dohw_m = p.dv_ohw.mul, dow_m = p.dv_ow.mul;
    int dnb_s = p.dv_nb.shift, ds0_s = p.dv_s0.shift, ds1_s = p.dv_s1.shift, dohw_s = p.dv_ohw.shift, dow_s = p.dv_ow.shift;
    Y3_PIN_S(sk_n0); Y3_PIN_S(sk_s0); Y3_PIN_S(sk_s1); Y3_PIN_S(sk_f); Y3_PIN_S(sk_c0); Y3_PIN_S(sk_c1);
    Y3_PIN_S(col_major); Y3_PIN_S(nb_fast); Y3_PIN_S(ohw); Y3_PIN_S(OW); Y3_PIN_S(aK); Y3_PIN_S(aM); Y3_PIN_S(aH); Y3_PIN_S(aW);
    Y3_PIN_S(src_ld); Y3_PIN_S(csh); Y3_PIN_S(csw); Y3_PIN_S(ntaps); Y3_PIN_S(Nout);
    Y3_PIN_S(dnb_m); Y3_PIN_S(ds0_m); Y3_PIN_S(ds1_m); Y3_PIN_S(dohw_m); Y3_PIN_S(dow_m);
    Y3_PIN_S(dnb_s); Y3_PIN_S(ds0_s); Y3_PIN_S(ds1_s); Y3_PIN_S(dohw_s); Y3_PIN_S(dow_s);
    const Y3Div dv_nb = {dnb_m, dnb_s}, dv_s0 = {ds0_m, ds0_s}, dv_s1 = {ds1_m, ds1_s}, dv_ohw = {dohw_m, dohw_s}, dv_ow = {dow_m, dow_s};

    // work item: ids are contiguous per XCD inside the two ranges [0, sk_n0) and [sk_n0, grid)
    int bid0 = braw < sk_n0 ? y3_xcd_remap(braw, sk_n0) : sk_n0 + y3_xcd_remap(braw - sk_n0, grid - sk_n0);
    if (SHORTLAST && (p.x3_mode & 8) && sk_n0 == grid && sk_s0 > 1) {
        // SHORT-LAST deal (the planner's overflow plans: every tile in s slices, the last one short, a few blocks more than slots):
        // an XCD takes its eighth of the LONG items (K-slice-major as below) first and fills up with SHORT ones, so the blocks the
        // dispatcher starts last -- the ones beyond the slots -- are short everywhere
        const int s = sk_s0, tiles = sk_f, nl = tiles * (s - 1);
        const int x = braw & 7, j = braw >> 3;
        const int ql = nl >> 3, rl = nl & 7;
        const int nlx = ql + (x < rl ? 1 : 0), l0 = x * ql + min(x, rl);      // this XCD's share of the long items
        if (j < nlx) {
            const int a = l0 + j, col = a / (nb_fast * (s - 1)), r = a - col * nb_fast * (s - 1), kzp = r / nb_fast;
            bid0 = (col * nb_fast + (r - kzp * nb_fast)) * s + kzp;
        } else {
            int s0x = 0;                                                      // short items the XCDs in front of this one take
            for (int y = 0; y < x; ++y) s0x += ((grid - y + 7) >> 3) - (ql + (y < rl ? 1 : 0));
            bid0 = (s0x + j - nlx) * s + (s - 1);
        }
    } else if (KZMAJOR && col_major) {
        const bool ra = bid0 < sk_n0;
        const int sl = ra ? sk_s0 : sk_s1;
        if (sl > 1) {
            const int t0 = ra ? 0 : sk_f, t1 = ra ? sk_f : sk_f + y3_div(grid - sk_n0, dv_s1);      // tiles [t0, t1) of the range
            int a = ra ? bid0 : bid0 - sk_n0, base = t0;
            int n = min(t1, (y3_div(t0, dv_nb) + 1) * nb_fast) - t0;      // tiles of the range in its first column
            if (a >= n * sl) {
                a -= n * sl;
                base += n;
                const int col = a / (nb_fast * sl);                      // whole columns in front of this item's
                a -= col * nb_fast * sl;
                base += col * nb_fast;
                n = min(nb_fast, t1 - base);
            }
            const int kzp = a / n;
            bid0 = (ra ? 0 : sk_n0) + (base + (a - kzp * n) - t0) * sl + kzp;
        }
    }
    int bid, kz, nz, kchunk;
    if (bid0 < sk_n0) {
        bid = y3_div(bid0, dv_s0);
        kz = bid0 - bid * sk_s0;
        nz = sk_s0;
        kchunk = sk_c0;
    } else {
        const int t = bid0 - sk_n0;
        const int q = y3_div(t, dv_s1);
        bid = sk_f + q;
        kz = t - q * sk_s1;
        nz = sk_s1;
        kchunk = sk_c1;
    }
    // tile id -> (row tile, column tile).  Ids are contiguous per XCD (y3_xcd_remap), so the fastest-varying coordinate decides
    // which operand an XCD's private 4 MB L2 keeps: row-major ids walk all column tiles of a few row tiles (the activation
    // rows stay, the whole kernel matrix streams through once per row tile), column-major ids walk all row tiles of a few
    // column tiles (a slice of the kernel matrix stays, the activations stream).  The host picks column-major when the kernel
    // matrix is too large to stay resident (> 2 MB): 13x13 512->1024 3x3 fetched its 18.9 MB of weights ~24 times per launch.
    // nb_fast is the count of the fastest-varying coordinate (nbm when column-major, else nbn).
    const int tq = y3_div(bid, dv_nb), tr = bid - tq * nb_fast;
    const int bm = col_major ? tr : tq;
    const int bn = col_major ? tq : tr;
    const int m0 = bm * BM, n0 = bn * BN;
    const int kbeg = kz * kchunk * BK;
    const int kend = min(aK, kbeg + kchunk * BK);
    w.tid = tid; w.lane = lane; w.wave = wave; w.l31 = l31; w.lh = lh; w.wm = wm; w.wn = wn;
    w.bid0 = bid0; w.bid = bid; w.kz = kz; w.nz = nz;
    w.bm = bm; w.bn = bn; w.m0 = m0; w.n0 = n0; w.kbeg = kbeg; w.kend = kend;
    w.ohw = ohw; w.OW = OW; w.aK = aK; w.aM = aM; w.aH = aH; w.aW = aW; w.src_ld = src_ld; w.csh = csh; w.csw = csw; w.ntaps = ntaps; w.Nout = Nout;
    w.dv_ohw = dv_ohw; w.dv_ow = dv_ow;
    return w;
}

// Everything after the K loop: the split-K hand-off (slices of a tile park their raw accumulators, the last arriver sums them
// in slice order) and the epilogue (bias, leaky-relu, BatchNorm statistics / folded affine, residual, accumulate, stores).
// `red`: LDS for the column sums, [2 or 6][WM][BN] floats (BNS kernels pass their A stage, dead after the loop).
template <int BM, int BN, int WM, int WN, bool DENSE, bool BNS, bool PEEK = false>
__device__ __forceinline__ void conv_fast_finish(const FastArgs& p, const FastWork& w, f32x16 (&acc)[BM / WM / 32][BN / WN / 32], float (*red)[WM][BN]) {
    constexpr int THREADS = 64 * WM * WN;
    constexpr int TM = BM / WM, TN = BN / WN, MB = TM / 32, NB = TN / 32;
    const int tid = w.tid, l31 = w.l31, lh = w.lh, wm = w.wm, wn = w.wn;
    const int bid0 = w.bid0, bid = w.bid, kz = w.kz, nz = w.nz, bm = w.bm, m0 = w.m0, n0 = w.n0;
    const int ohw = w.ohw, OW = w.OW;
    const Y3Div dv_ohw = w.dv_ohw, dv_ow = w.dv_ow;
    Y3_ABL_INIT();
    if (nz > 1) {
        // park the raw accumulators: slab[item][r4][thread] as 16-byte stores, one KiB per wave instruction.  The hand-off to the
        // slice that finishes last follows MI355X_MICROARCH.md (workgroup dispatch, measured hand-offs, row 1): every byte is
        // stored sc1 (written through, no L2 write-back fence needed) and loaded sc1, every storing wave drains vmcnt before the
        // workgroup barrier, ONE lane then adds to the tile's ticket with an agent-scope atomic and the workgroup whose add
        // came last (told by the value returned) loads after a second barrier.
        // PEEK (the x3 kernels): before parking anything a slice LOOKS at the ticket (one sc1 load -- the "poll of that counter"
        // form of the same table row).  If the other nz - 1 slices have all arrived it is the last one for certain: it neither
        // stores nor re-reads its own 64 KB but sums the others' slabs around its registers, IN SLICE ORDER as always (slabs
        // 0 .. kz-1, then its own accumulators, then kz+1 .. nz-1: the sum is bit for bit the one the ticket path produces,
        // whichever slice ends up last).  Otherwise it parks and takes a ticket as before.  The last finisher of a tile -- the one
        // the launch waits for -- thus skips a 64 KB write-through store, its drain and a barrier, and 1 / nz of the slab bytes
        // never exist (nz = 2: half).
        constexpr int R4 = MB * NB * 4;
        const __amdgpu_buffer_rsrc_t rs_slab = __builtin_amdgcn_make_buffer_rsrc(p.slab, 0, 0x7ffffff0, 0x00020000);
        const unsigned item_bytes = (unsigned)(R4 * THREADS * 16);
        int* flag = reinterpret_cast<int*>(&red[0][0][0]);
        bool sure_last = false;
        if constexpr (PEEK) {
            if (tid == 0) *flag = __hip_atomic_load(p.tickets + bid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nz - 1 ? 1 : 0;
            __syncthreads();
            sure_last = *flag != 0;
            __syncthreads();      // (flag is written again below)
        }
        const unsigned first = (unsigned)(bid0 - kz - p.sk_slab0) * item_bytes + (unsigned)tid * 16u;   // slice 0 of this tile
        auto add_slab = [&](f32x16 (&dst)[MB][NB], int z, bool init) {
            const unsigned base = first + (unsigned)z * item_bytes;
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const f32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_slab, base, (unsigned)(((i * NB + j) * 4 + r) * THREADS * 16), 16 /* sc1 */);
#pragma unroll
                        for (int e = 0; e < 4; ++e) dst[i][j][4 * r + e] = init ? v[e] : dst[i][j][4 * r + e] + v[e];
                    }
        };
        if (PEEK && sure_last) {
            if (tid == 0) __hip_atomic_store(p.tickets + bid, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (kz > 0) {
                f32x16 head[MB][NB];
#pragma unroll 1
                for (int z = 0; z < kz; ++z) add_slab(head, z, z == 0);
#pragma unroll
                for (int i = 0; i < MB; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] = head[i][j][r] + acc[i][j][r];
            }
#pragma unroll 1
            for (int z = kz + 1; z < nz; ++z) add_slab(acc, z, false);
        } else {
            {
                const unsigned base = (unsigned)(bid0 - p.sk_slab0) * item_bytes + (unsigned)tid * 16u;
#pragma unroll
                for (int i = 0; i < MB; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            f32x4 v = {acc[i][j][4 * r], acc[i][j][4 * r + 1], acc[i][j][4 * r + 2], acc[i][j][4 * r + 3]};
                            if (Y3_ABL(8))      // probe only: default cache policy (the hand-off is then not guaranteed; timing / clock experiment)
                                __builtin_amdgcn_raw_buffer_store_b128(v, rs_slab, base, (unsigned)(((i * NB + j) * 4 + r) * THREADS * 16), 0);
                            else
                                __builtin_amdgcn_raw_buffer_store_b128(v, rs_slab, base, (unsigned)(((i * NB + j) * 4 + r) * THREADS * 16), 16 /* sc1 */);
                        }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                const int old = __hip_atomic_fetch_add(p.tickets + bid, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int last = old == nz - 1;
                if (last) __hip_atomic_store(p.tickets + bid, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *flag = last;
            }
            __syncthreads();
            if (*flag == 0) return;
            __syncthreads();  // `red` is reused by the statistics below
#pragma unroll 1
            for (int z = 0; z < nz; ++z) add_slab(acc, z, z == 0);
        }
    }

    // ---- epilogue (same contract as conv_igemm_kernel)
    const bool do_lrelu = p.flags & Y3_EPI_LRELU;
    const bool do_accum = p.flags & Y3_EPI_ACCUM;
    float ssum[NB], ssq[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) ssum[j] = ssq[j] = 0.f;
    float bsum[BNS ? 6 : 1][NB];
#pragma unroll
    for (int q = 0; q < (BNS ? 6 : 1); ++q)
#pragma unroll
        for (int j = 0; j < NB; ++j) bsum[q][j] = 0.f;
    if constexpr (DENSE) {
        // dense destination (pixel index == m): buffer stores with the row part of the offset in the scalar operand and
        // tile-edge lanes pointed out of range -- no per-element 64-bit address math, no divergent branches.  All 676
        // workgroups of a layer reach their epilogue together, so its instruction count is exposed, not hidden.
        const __amdgpu_buffer_rsrc_t rs_dst = __builtin_amdgcn_make_buffer_rsrc(p.dst, 0, p.dst_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.resid ? p.resid : p.dst), 0,
                                                                               p.resid ? p.resid_bytes : 0u, 0x00020000);
        const int mrow = m0 + wm * TM + 4 * lh;
        const unsigned ld4 = (unsigned)p.dst_ld * 4u, rld4 = (unsigned)p.resid_ld * 4u;
        const bool full = m0 + BM <= p.M;  // wave-uniform: only the last row tile needs per-row masking
        const bool has_scale = p.scale != nullptr, has_resid = p.resid != nullptr;
        // BNS: this launch completes the output gradient dy of a BatchNorm layer, so the six raw moments of (dy, a) that its
        // backward needs (pointwise.hip, bn_bwd_stats_kernel) are summed here, while dy is in registers -- the separate pass
        // over dy and a is gone.  `a` has the geometry of dst; lanes outside the tile read zeros (no contribution).
        const __amdgpu_buffer_rsrc_t rs_bna = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(BNS ? p.bn_a : p.dst), 0, BNS ? p.bn_a_bytes : 0u, 0x00020000);
        const unsigned ald4 = (unsigned)p.bn_a_ld * 4u;
        const bool one_extra = has_resid != do_accum;
        const __amdgpu_buffer_rsrc_t rs_ext = has_resid ? rs_res : rs_dst;
        const unsigned xld4 = has_resid ? rld4 : ld4;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int n = n0 + wn * TN + j * 32 + l31;
            const bool nok = n < p.Nout;
            const float bias = (p.bias && nok) ? p.bias[n] : 0.f;
            const float sc = (has_scale && nok) ? p.scale[n] : 1.f;
            const float sf = (has_scale && nok) ? p.shift[n] : 0.f;
            const unsigned vbase = nok ? (unsigned)mrow * ld4 + (unsigned)n * 4u : Y3_OOB;
            const unsigned rbase = nok ? (unsigned)mrow * rld4 + (unsigned)n * 4u : Y3_OOB;
            const unsigned abase = nok ? (unsigned)mrow * ald4 + (unsigned)n * 4u : Y3_OOB;
            const unsigned xbase = has_resid ? rbase : vbase;
            // Exactly one extra operand per element (the residual of an inference layer, or the gradient a data gradient adds
            // to): its loads are issued eight at a time from the one descriptor in use, then consumed.  As single loads inside the
            // arithmetic they were sixteen dependent memory round trips per 32x32 block.  (Eight, not sixteen, in flight: the 64x64
            // kernel must stay within 64 VGPRs -- 8 workgroups per CU -- see `red`.)
            if (!BNS && one_extra) {
#pragma unroll
                for (int i = 0; i < MB; ++i) {
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        float ext[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const int r = g * 8 + q;
                            const int dr = i * 32 + (r & 3) + 8 * (r >> 2);
                            const bool ok = full ? nok : (nok && mrow + dr < p.M);
                            ext[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_ext, ok ? xbase : Y3_OOB, (unsigned)dr * xld4, 0));
                        }
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const int r = g * 8 + q;
                            const int dr = i * 32 + (r & 3) + 8 * (r >> 2);
                            const bool ok = full ? nok : (nok && mrow + dr < p.M);
                            const unsigned vo = ok ? vbase : Y3_OOB;
                            float v = acc[i][j][r] + bias;
                            if (do_lrelu) v = v > 0.f ? v : p.alpha * v;
                            const float vs = ok ? v : 0.f;
                            ssum[j] += vs;
                            ssq[j] += vs * vs;
                            if (has_scale) v = v * sc + sf;
                            v += ext[q];
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs_dst, vo, (unsigned)dr * ld4, 0);
                        }
                    }
                }
                continue;
            }
#pragma unroll
            for (int i = 0; i < MB; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = i * 32 + (r & 3) + 8 * (r >> 2);
                    const bool ok = full ? nok : (nok && mrow + dr < p.M);
                    const unsigned vo = ok ? vbase : Y3_OOB;
                    float v = acc[i][j][r] + bias;
                    if (do_lrelu) v = v > 0.f ? v : p.alpha * v;
                    const float vs = ok ? v : 0.f;
                    if constexpr (!BNS) {
                        ssum[j] += vs;
                        ssq[j] += vs * vs;
                    }
                    if (has_scale) v = v * sc + sf;
                    if (has_resid) v += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_res, ok ? rbase : Y3_OOB, (unsigned)dr * rld4, 0));
                    if (do_accum) v += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_dst, vo, (unsigned)dr * ld4, 0));
                    if constexpr (BNS) {
                        const float av = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_bna, ok ? abase : Y3_OOB, (unsigned)dr * ald4, 0));
                        const float dv = ok ? v : 0.f;
                        const bool pos = av > 0.f;
                        bsum[0][j] += dv;
                        bsum[1][j] += dv * av;
                        bsum[2][j] += pos ? dv : 0.f;
                        bsum[3][j] += pos ? av : 0.f;
                        bsum[4][j] += pos ? 1.f : 0.f;
                        bsum[5][j] += av;
                    }
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs_dst, vo, (unsigned)dr * ld4, 0);
                }
            }
        }
    } else {
        // strided destination (the four parity launches of a stride-2 data gradient): decompose each accumulator row
        // once (not once per column block), then the same buffer-store epilogue as above
        const __amdgpu_buffer_rsrc_t rs_dst = __builtin_amdgcn_make_buffer_rsrc(p.dst, 0, p.dst_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.resid ? p.resid : p.dst), 0,
                                                                               p.resid ? p.resid_bytes : 0u, 0x00020000);
        const bool has_scale = p.scale != nullptr, has_resid = p.resid != nullptr;
        const __amdgpu_buffer_rsrc_t rs_bna = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(BNS ? p.bn_a : p.dst), 0, BNS ? p.bn_a_bytes : 0u, 0x00020000);
        unsigned rowpix[MB][16];
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int mm = m < p.M ? m : 0;
                const int nimg = y3_div(mm, dv_ohw);
                const int rr = mm - nimg * ohw;
                const int oh = y3_div(rr, dv_ow);
                const int ow = rr - oh * OW;
                const unsigned pix = (unsigned)((nimg * p.DH + oh * p.dsh + p.doh) * p.DW + ow * p.dsw + p.dow);
                rowpix[i][r] = m < p.M ? pix : 0xffffffffu;
            }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int n = n0 + wn * TN + j * 32 + l31;
            const bool nok = n < p.Nout;
            const float bias = (p.bias && nok) ? p.bias[n] : 0.f;
            const float sc = (has_scale && nok) ? p.scale[n] : 1.f;
            const float sf = (has_scale && nok) ? p.shift[n] : 0.f;
#pragma unroll
            for (int i = 0; i < MB; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool ok = nok && rowpix[i][r] != 0xffffffffu;
                    const unsigned vo = ok ? (rowpix[i][r] * (unsigned)p.dst_ld + (unsigned)n) * 4u : Y3_OOB;
                    float v = acc[i][j][r] + bias;
                    if (do_lrelu) v = v > 0.f ? v : p.alpha * v;
                    const float vs = ok ? v : 0.f;
                    ssum[j] += vs;
                    ssq[j] += vs * vs;
                    if (has_scale) v = v * sc + sf;
                    if (has_resid)
                        v += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_res, ok ? (rowpix[i][r] * (unsigned)p.resid_ld + (unsigned)n) * 4u : Y3_OOB, 0, 0));
                    if (do_accum) v += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_dst, vo, 0, 0));
                    if constexpr (BNS) {      // the same six moments as in the dense epilogue; `a` has the geometry of the strided destination
                        const float av = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_bna, ok ? (rowpix[i][r] * (unsigned)p.bn_a_ld + (unsigned)n) * 4u : Y3_OOB, 0, 0));
                        const float dv = ok ? v : 0.f;
                        const bool pos = av > 0.f;
                        bsum[0][j] += dv;
                        bsum[1][j] += dv * av;
                        bsum[2][j] += pos ? dv : 0.f;
                        bsum[3][j] += pos ? av : 0.f;
                        bsum[4][j] += pos ? 1.f : 0.f;
                        bsum[5][j] += av;
                    }
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs_dst, vo, 0, 0);
                }
            }
        }
    }
    if constexpr (BNS) {
        // fixed order: lane halves, then the WM waves of a column -- deterministic, no atomics (as the forward statistics below)
        __syncthreads();      // `red` aliases the A stage: every wave must be done with its last fragment reads
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const float s = bsum[q][j] + __shfl_xor(bsum[q][j], 32);
                if (lh == 0) red[q][wm][wn * TN + j * 32 + l31] = s;
            }
        __syncthreads();
        for (int c = tid; c < 6 * BN; c += THREADS) {
            const int which = c / BN, col = c % BN;
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) s += red[which][w][col];
            const int n = n0 + col;
            if (n < p.Nout) p.bn_part[((long long)(p.bn_row0 + bm) * 6 + which) * p.Nout + n] = s;
        }
    } else if (p.stats) {
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
#ifdef Y3_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (y3_timing_buf && threadIdx.x == 0) y3_timing_buf[(size_t)blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memrealtime();
#endif
    Y3_TSTAMP(3);
}
