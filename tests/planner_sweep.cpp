// Host-only sweep of the convolution launch planners (conv.hip: plan_conv, plan_wgrad, the *_workspace / *_tiles queries),
// built by tests/test_cpu_planner.py from the HOST side of conv.hip + core.hip with -fsanitize=address,undefined
// (hipcc --cuda-host-only: no device code, no GPU needed) and run once per environment-switch combination.
//
// For every convolution of the network (any batch / image size given on the command line) it asks the
// library for the plan of the forward, the data gradient (stride 1, and the 1/2/4-tap parity classes of stride 2) and the
// kernel gradient, and REPLAYS on the host what the kernels do with those numbers: work item -> (tile, K slice), slab slot,
// ticket index, K range -- checking every index against the workspace the query function told the caller to allocate.
// Prints one line per plan ("key=value ...") so that the Python side can compare with the product library; exits non-zero
// on the first violated invariant.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>

#include "../include/yolo3hip.h"

static int fails = 0;
#define REQUIRE(cond, ...)                         \
    do {                                           \
        if (!(cond)) {                             \
            ++fails;                               \
            fprintf(stderr, "VIOLATION: ");        \
            fprintf(stderr, __VA_ARGS__);          \
            fprintf(stderr, "  [%s]\n", #cond);    \
        }                                          \
    } while (0)

static int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// common.h y3_xcd_remap, restated (the kernel's block id -> work item permutation)
static int xcd_remap(int orig, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = orig & 7, j = orig >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

static const size_t HEADER = 256 * 1024;      // yolo3hip.h workspace contract: 65 536 tickets in front of the slabs

static void check_conv_plan(const char* what, int m, int cin, int k, int cout, unsigned flags = 0) {
    int o[13];
    const size_t ws = y3_conv2d_plan_x(m, cin, k, cout, flags, o);
    const int bm = o[0], bn = o[1], bk = o[2], tiles = o[3], f = o[4], s0 = o[5], s1 = o[6], c0 = o[7], c1 = o[8], grid = o[9],
              stats_tiles = o[10], fast = o[11], nk = o[12];
    printf("conv %s m=%d cin=%d k=%d cout=%d bm=%d bn=%d bk=%d tiles=%d f=%d s0=%d s1=%d c0=%d c1=%d grid=%d stats=%d fast=%d nk=%d ws=%zu\n", what, m,
           cin, k, cout, bm, bn, bk, tiles, f, s0, s1, c0, c1, grid, stats_tiles, fast, nk, ws);
    REQUIRE(ws == y3_conv2d_fwd_workspace_x(m, cin, k, cout, flags), "%s: plan and workspace query disagree", what);
    REQUIRE(stats_tiles == y3_conv2d_stats_tiles_x(m, cin, k, cout, flags), "%s: plan and stats_tiles query disagree", what);
    if (flags & Y3_CONV_X3) {
        REQUIRE(bm == 128 && (bn == 128 || bn == 64), "%s: x3 tile %dx%d", what, bm, bn);
        // the patch kernel's loop body is 18 K steps (two 16-channel chunks of nine taps): split 3x3 launches are cut in such units
        if (k == 3 && s0 > 1) REQUIRE(c0 % 18 == 0, "%s: x3 slice of %d K steps", what, c0);
        if (k == 3 && s1 > 1 && f < tiles) REQUIRE(c1 % 18 == 0, "%s: x3 remainder slice of %d K steps", what, c1);
        REQUIRE(c0 % 2 == 0 && c1 % 2 == 0, "%s: x3 slices are pairs of steps (%d, %d)", what, c0, c1);
    }
    REQUIRE(bk == 16 && (bm == 64 || bm == 128 || bm == 256) && (bn == 32 || bn == 64 || bn == 128), "%s: tile %dx%dx%d", what, bm, bn, bk);
    REQUIRE(tiles == cdiv(m, bm) * cdiv(cout, bn) && stats_tiles == cdiv(m, bm), "%s: tile count", what);
    REQUIRE(f >= 0 && f <= tiles && s0 >= 1 && s1 >= 1 && c0 >= 1 && c1 >= 1, "%s: split parameters", what);
    REQUIRE((long long)f * s0 + (long long)(tiles - f) * s1 == grid && grid > 0, "%s: grid", what);
    if (!(fast & 1)) {
        REQUIRE(s0 == 1 && s1 == 1 && ws == 0, "%s: the generic kernel has no split-K", what);
        return;
    }
    // every slice non-empty, the slices cover the K steps exactly
    if (s0 > 1) REQUIRE(cdiv(nk, c0) == s0 && (long long)(s0 - 1) * c0 < nk, "%s: s0=%d chunk0=%d nk=%d", what, s0, c0, nk);
    if (s1 > 1) REQUIRE(cdiv(nk, c1) == s1 && (long long)(s1 - 1) * c1 < nk, "%s: s1=%d chunk1=%d nk=%d", what, s1, c1, nk);
    if (s0 == 1 && f > 0) REQUIRE(c0 >= nk, "%s: unsplit tiles must run all K steps", what);
    if (s1 == 1 && f < tiles) REQUIRE(c1 >= nk, "%s: unsplit remainder tiles must run all K steps", what);
    const bool split = s0 > 1 || (s1 > 1 && f < tiles);
    REQUIRE(split == (ws > 0), "%s: workspace %zu for split=%d", what, ws, (int)split);
    if (!split) return;
    REQUIRE(tiles <= (int)(HEADER / 4), "%s: %d tiles exceed the ticket header", what, tiles);
    const int n0 = f * s0;
    const int slab0 = s0 > 1 ? 0 : n0;                        // launch_igemm: first item that owns a slab slot
    const size_t item_bytes = (size_t)bm * bn * 4;
    REQUIRE(ws >= HEADER, "%s: workspace smaller than its header", what);
    std::vector<unsigned char> seen((size_t)grid, 0), slices((size_t)tiles, 0);
    for (int b = 0; b < grid; ++b) {
        int item = b < n0 ? xcd_remap(b, n0) : n0 + xcd_remap(b - n0, grid - n0);
        if ((flags & Y3_CONV_X3) && (fast & 2) && n0 == grid && s0 > 1) {
            // conv_fast_decode<SHORTLAST>: an XCD takes its eighth of the long items first and fills up with short ones
            const int nb = cdiv(m, bm), nl = tiles * (s0 - 1), x = b & 7, j = b >> 3, ql = nl >> 3, rl = nl & 7;
            const int nlx = ql + (x < rl ? 1 : 0), l0 = x * ql + std::min(x, rl);
            if (j < nlx) {
                const int a = l0 + j, col = a / (nb * (s0 - 1)), r = a - col * nb * (s0 - 1), kzp = r / nb;
                item = (col * nb + (r - kzp * nb)) * s0 + kzp;
            } else {
                int s0x = 0;
                for (int y = 0; y < x; ++y) s0x += ((grid - y + 7) >> 3) - (ql + (y < rl ? 1 : 0));
                REQUIRE(s0x + j - nlx >= 0 && s0x + j - nlx < tiles, "%s: short item %d of %d (block %d)", what, s0x + j - nlx, tiles, b);
                item = (s0x + j - nlx) * s0 + (s0 - 1);
            }
        } else if ((flags & Y3_CONV_X3) && k == 3) {
            // conv_fast_decode<KZMAJOR> (x3 patch kernel, column-major tile ids): the items of a column are dealt K slice by K slice
            const bool ra = item < n0;
            const int sl = ra ? s0 : s1, nb = cdiv(m, bm);
            if (sl > 1) {
                const int t0 = ra ? 0 : f, t1 = ra ? f : tiles;
                int a = ra ? item : item - n0, base = t0;
                int n = std::min(t1, (t0 / nb + 1) * nb) - t0;
                if (a >= n * sl) {
                    a -= n * sl;
                    base += n;
                    const int col = a / (nb * sl);
                    a -= col * nb * sl;
                    base += col * nb;
                    n = std::min(nb, t1 - base);
                }
                REQUIRE(n > 0, "%s: empty column in the K-slice-major deal (block %d)", what, b);
                if (n <= 0) continue;
                const int kzp = a / n;
                item = (ra ? 0 : n0) + (base + (a - kzp * n) - t0) * sl + kzp;
            }
        }
        REQUIRE(item >= 0 && item < grid, "%s: item %d of %d", what, item, grid);
        if (item < 0 || item >= grid) continue;
        REQUIRE(!seen[item], "%s: item %d drawn twice", what, item);
        seen[item] = 1;
        int tile, kz, nz, chunk;
        if (item < n0) {
            tile = item / s0, kz = item % s0, nz = s0, chunk = c0;
        } else {
            const int t = item - n0;
            tile = f + t / s1, kz = t % s1, nz = s1, chunk = c1;
        }
        REQUIRE(tile >= 0 && tile < tiles, "%s: tile %d of %d", what, tile, tiles);
        REQUIRE((long long)kz * chunk < nk, "%s: slice %d of tile %d starts at K step %lld >= %d", what, kz, tile, (long long)kz * chunk, nk);
        if (tile >= 0 && tile < tiles) ++slices[tile];
        if (nz > 1) {
            const long long slot = (long long)item - slab0;
            REQUIRE(slot >= 0 && HEADER + (size_t)(slot + 1) * item_bytes <= ws, "%s: slab slot %lld (item %d) outside the %zu-byte workspace", what, slot, item, ws);
            REQUIRE((size_t)(slot + 1) * item_bytes < 0x7ffffff0ull, "%s: slab offset overflows the 32-bit buffer offset", what);
        }
    }
    for (int t = 0; t < tiles; ++t) REQUIRE(slices[t] == (t < f ? s0 : s1), "%s: tile %d got %d slices", what, t, (int)slices[t]);
}

static void check_wgrad_plan(int m, int cin, int k, int cout, unsigned flags = 0) {
    int o[8];
    const size_t ws = y3_conv2d_wgrad_plan_x(m, cin, k, cout, flags, o);
    const int bkr = o[0], bn = o[1], splits = o[2], chunk = o[3], tiles = o[4], in_kernel = o[5], grid = o[6], table = o[7];
    const long long K = (long long)k * k * cin;
    printf("wgrad%s m=%d cin=%d k=%d cout=%d bkr=%d bn=%d splits=%d chunk=%d tiles=%d in_kernel=%d grid=%d ws=%zu\n", flags ? "_x3" : "", m, cin, k, cout, bkr, bn, splits,
           chunk, tiles, in_kernel, grid, ws);
    y3_tensor src = {nullptr, 1, 1, m, cin, cin}, dd = {nullptr, 1, 1, m, cout, cout};
    REQUIRE(ws == y3_conv2d_wgrad_workspace_x(&src, &dd, k, 1, flags), "wgrad: plan and workspace query disagree");
    if (flags & Y3_CONV_X3) REQUIRE(bkr == 128 && bn == 128 && chunk % 96 == 0, "wgrad x3: tile %dx%d, %d pixels per split (six steps of 16 per loop iteration)", bkr, bn, chunk);
    REQUIRE((bkr == 64 || bkr == 128) && (bn == 32 || bn == 64 || bn == 128), "wgrad: tile %dx%d", bkr, bn);
    REQUIRE(tiles == cdiv(K, bkr) * cdiv(cout, bn), "wgrad: tile count");
    REQUIRE(splits >= 1 && chunk >= 16 && chunk % 16 == 0, "wgrad: splits=%d chunk=%d", splits, chunk);
    REQUIRE(cdiv(m, chunk) == splits && (long long)(splits - 1) * chunk < m, "wgrad: an empty split (m=%d chunk=%d splits=%d)", m, chunk, splits);
    REQUIRE(chunk <= table, "wgrad: %d pixels per split do not fit the %d-entry LDS pixel table", chunk, table);
    REQUIRE(grid >= splits * tiles && grid > 0, "wgrad: grid %d < %d", grid, splits * tiles);
    if (splits == 1) {
        REQUIRE(ws == 0 && !in_kernel, "wgrad: one split needs no workspace");
        return;
    }
    REQUIRE(ws < 0x7ff00000ull, "wgrad: workspace %zu too large for 32-bit slab offsets", ws);
    if (in_kernel) {
        REQUIRE(splits <= 8 && tiles <= (int)(HEADER / 4), "wgrad: in-kernel reduction with %d splits / %d tiles", splits, tiles);
        // slab[tile][split] in fragment order: bkr x bn floats per item
        const size_t item = (size_t)bkr * bn * 4;
        REQUIRE(HEADER + (size_t)tiles * splits * item <= ws, "wgrad: slabs outside the workspace");
    } else {
        REQUIRE(HEADER + (size_t)splits * K * cout * 4 <= ws, "wgrad: natural-layout slabs outside the workspace");
    }
}

struct Layer {
    int cin, cout, k, s, div;      // div: output spatial size = img / div
};

int main(int argc, char** argv) {
    const int batch = argc > 1 ? atoi(argv[1]) : 8, img = argc > 2 ? atoi(argv[2]) : 416, heads = argc > 3 ? atoi(argv[3]) : 14;
    std::vector<Layer> L;
    auto conv = [&](int cin, int cout, int k, int s, int div) { L.push_back({cin, cout, k, s, div}); };
    auto block = [&](int c, int reps, int div) {
        for (int i = 0; i < reps; ++i) {
            conv(c, c / 2, 1, 1, div);
            conv(c / 2, c, 3, 1, div);
        }
    };
    // model.py:383-421 (backbone), :51-59 / :356-380 (yolo blocks, laterals, heads)
    conv(4, 32, 3, 1, 1);
    conv(32, 64, 3, 2, 2);
    block(64, 1, 2);
    conv(64, 128, 3, 2, 4);
    block(128, 2, 4);
    conv(128, 256, 3, 2, 8);
    block(256, 8, 8);
    conv(256, 512, 3, 2, 16);
    block(512, 8, 16);
    conv(512, 1024, 3, 2, 32);
    block(1024, 4, 32);
    auto yolo = [&](int cin, int fc, int div) {
        for (int i = 0; i < 3; ++i) {
            conv(i == 0 ? cin : fc, fc / 2, 1, 1, div);
            conv(fc / 2, fc, 3, 1, div);
        }
        conv(fc, heads, 1, 1, div);
    };
    yolo(1024, 1024, 32);
    conv(512, 512, 1, 1, 32);
    yolo(1024, 512, 16);
    conv(256, 256, 1, 1, 16);
    yolo(512, 256, 8);
    for (size_t i = 0; i < L.size(); ++i) {
        const Layer& l = L[i];
        const int o = img / l.div, m = batch * o * o;
        char name[64];
        snprintf(name, sizeof name, "fwd[%zu]", i);
        check_conv_plan(name, m, l.cin, l.k, l.cout);
        check_wgrad_plan(m, l.cin, l.k, l.cout);
        if (y3_conv2d_x3_ok(m, l.cin, l.k * l.k, l.cout)) {      // the same launch on the x3 kernels (Y3_CONV_X3)
            snprintf(name, sizeof name, "fwd_x3[%zu]", i);
            check_conv_plan(name, m, l.cin, l.k, l.cout, Y3_CONV_X3);
        }
        if (y3_conv2d_wgrad_x3_ok(m, l.cin, l.k, l.cout)) check_wgrad_plan(m, l.cin, l.k, l.cout, Y3_CONV_X3);
        if (i == 0) continue;                    // the first layer has no data gradient
        const int in = o * l.s, mi = batch * in * in;
        y3_tensor dd = {nullptr, batch, o, o, l.cout, (l.cout + 3) / 4 * 4}, ds = {nullptr, batch, in, in, l.cin, l.cin};
        if (l.s == 1) {
            snprintf(name, sizeof name, "dgrad[%zu]", i);
            check_conv_plan(name, mi, l.cout, l.k, l.cin);
            if (y3_conv2d_x3_ok(mi, l.cout, l.k * l.k, l.cin)) {
                snprintf(name, sizeof name, "dgrad_x3[%zu]", i);
                check_conv_plan(name, mi, l.cout, l.k, l.cin, Y3_CONV_X3);
                REQUIRE(y3_conv2d_dgrad_workspace_x(&dd, l.k, l.s, &ds, Y3_CONV_X3) == y3_conv2d_fwd_workspace_x(mi, l.cout, l.k, l.cin, Y3_CONV_X3), "dgrad_x3[%zu]: workspace queries disagree", i);
            }
        } else {
            for (int nt = 1; nt <= 4; nt *= 2) {          // parity classes of a 3x3 stride-2 data gradient: 1, 2, 2, 4 taps
                snprintf(name, sizeof name, "dgrad[%zu]/taps%d", i, nt);
                check_conv_plan(name, batch * ((in + 1) / 2) * ((in + 1) / 2), nt * l.cout, 1, l.cin);
            }
            if (y3_conv2d_dgrad_x3_ok(&dd, l.k, l.s, &ds)) {      // the merged launch on the x3 kernels: per-class K slices behind one ticket header
                const size_t xws = y3_conv2d_dgrad_workspace_x(&dd, l.k, l.s, &ds, Y3_CONV_X3);
                const int xrows = y3_conv2d_dgrad_bn_tiles_x(&dd, l.k, l.s, &ds, Y3_CONV_X3);
                int want = 0;
                for (int ph = 0; ph < 2; ++ph)
                    for (int pw = 0; pw < 2; ++pw) want += cdiv(batch * ((in - ph + 1) / 2) * ((in - pw + 1) / 2), 128);
                printf("dgrad_x3_s2[%zu] ws=%zu bn_tiles=%d\n", i, xws, xrows);
                REQUIRE(xrows == want || xrows == 0 /* merged launch switched off: Y3_NO_DGRAD_MULTI */, "dgrad_x3_s2[%zu]: %d statistics rows, the four classes have %d row tiles of 128", i, xrows, want);
                // at most one slice per workgroup slot (512; the development switch goes up to 1000) of 128 x 128 floats behind the header, or no split at all
                REQUIRE(xws == 0 || (xws > HEADER && xws <= HEADER + (size_t)(1024 + 4 * cdiv(l.cin, 64)) * 128 * 128 * 4), "dgrad_x3_s2[%zu]: workspace %zu", i, xws);
            }
        }
        const size_t dws = y3_conv2d_dgrad_workspace(&dd, l.k, l.s, &ds);
        const int rows = y3_conv2d_dgrad_bn_tiles(&dd, l.k, l.s, &ds);
        printf("dgrad_q[%zu] ws=%zu bn_tiles=%d\n", i, dws, rows);
        REQUIRE(rows >= 0 && rows <= 4 * cdiv(mi, 64) + 4, "dgrad_bn_tiles[%zu] = %d", i, rows);
        if (l.s == 1 && rows > 0) {
            int p[13];
            y3_conv2d_plan(mi, dd.c, l.k, l.cin, p);
            REQUIRE(rows == cdiv(mi, p[0]), "dgrad_bn_tiles[%zu]: %d rows but the launch has %d row tiles", i, rows, cdiv(mi, p[0]));
        }
    }
    if (fails) {
        fprintf(stderr, "%d violation(s)\n", fails);
        return 1;
    }
    printf("ok %zu layers batch=%d img=%d\n", L.size(), batch, img);
    return 0;
}
