// Probe: where does an fp32 conv workgroup spend its time, and how is the work spread over the CUs?
// Builds conv.hip with -DY3_TIMING (per-workgroup s_memtime stamps: start, after prologue, after main loop, end; HW_ID,
// XCC_ID, s_memrealtime at both ends) and prints the distributions, the workgroups per CU and the clock the chip held.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DY3_TIMING -I include -I object-detection-yolov3_amd/csrc \
//         tools/probe/conv_timing.hip object-detection-yolov3_amd/csrc/core.hip -o tools/probe/conv_timing
//   tools/probe/conv_timing [n h cin cout k [x3]]   (forward, stride 1; x3 = 1: the Y3_CONV_X3 kernel; env Y3_TILE / Y3_RSPLIT / Y3_ABL apply)
// The stamped launch is the last of 200 back-to-back stamped launches, so the clock it reports is the clock the chip HOLDS under
// this kernel (and under the ablation Y3_ABL selects), not the clock of a first launch on an idle chip.
#include "../../object-detection-yolov3_amd/csrc/conv.hip"
#include "../../object-detection-yolov3_amd/csrc/conv_x3.hip"
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 8, h = argc > 2 ? atoi(argv[2]) : 52, w = h, cin = argc > 3 ? atoi(argv[3]) : 128,
              cout = argc > 4 ? atoi(argv[4]) : 256, k = argc > 5 ? atoi(argv[5]) : 3;
    const unsigned x3 = (argc > 6 && atoi(argv[6])) ? Y3_CONV_X3 : 0u;
    const size_t xs = (size_t)n * h * w * cin, ws = (size_t)k * k * cout * cin, ys = (size_t)n * h * w * cout;
    std::vector<float> hx(xs), hw(ws);
    unsigned seed = 1;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((int)(seed >> 9) & 0xffff) / 65536.f - 0.5f; };
    for (auto& v : hx) v = rnd();
    for (auto& v : hw) v = rnd() * 0.1f;
    float *dx, *dw, *dy, *db, *dstats;
    void* dws;
    unsigned long long* dt;
    const size_t wsb = y3_conv2d_fwd_workspace_x(n * h * w, cin, k, cout, x3) + 16;
    hipMalloc(&dx, xs * 4); hipMalloc(&dw, ws * 8);      /* (x3: three bf16 planes = 6 bytes per element) */ hipMalloc(&dy, ys * 4); hipMalloc(&db, cout * 4);
    hipMalloc(&dstats, (size_t)n * h * w / 16 * cout * 4 + 65536); hipMalloc(&dws, wsb);
    hipMemset(dws, 0, wsb);
    hipMemcpy(dx, hx.data(), xs * 4, hipMemcpyHostToDevice);
    hipMemcpy(dw, hw.data(), ws * 4, hipMemcpyHostToDevice);
    hipMemset(db, 0, cout * 4);
    y3_tensor src{dx, n, h, w, cin, cin}, dst{dy, n, h, w, cout, cout};
    auto run = [&]() { return y3_conv2d_fwd(&src, dw, db, k, 1, &dst, Y3_EPI_LRELU | x3, 0.2f, nullptr, nullptr, nullptr, dstats, dws, wsb, nullptr); };
    if (run() != 0) { printf("launch failed: %s\n", y3_last_error()); return 1; }
    for (int i = 0; i < 20; ++i) run();
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) run();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.0 * n * h * w * k * k * cin * cout;
    printf("layer %dx%dx%d %d->%d k%d%s: %.1f us per launch = %.1f TFLOP/s (stamps off)\n", n, h, w, cin, cout, k, x3 ? " x3" : "", ms * 50.f, flop / (ms * 50e-6) / 1e12);
    const int maxwg = 1 << 16;
    hipMalloc(&dt, (size_t)maxwg * 8 * 8);
    hipMemset(dt, 0, (size_t)maxwg * 8 * 8);
    const int abl = getenv("Y3_ABL") ? atoi(getenv("Y3_ABL")) : 0;     // ablation (results become wrong, timing stays meaningful)
    hipMemcpyToSymbol(HIP_SYMBOL(y3_abl_dev), &abl, sizeof(abl));
    if (abl) printf("ablation mask %d (1 = no global loads in the K loop, 2 = no LDS stores (x3: no split either), 4 = no barrier, 8 = split-K slabs stored with the default cache policy instead of sc1, x3 patch kernel: 16 = no activation loads, 32 = no weight loads, 64 = every weight load reads the first K block (always cached), 128 = every activation load reads the first channel chunk)\n", abl);
    {   // the whole launch under the ablation, stamps still off (the buffer pointer is set below)
        for (int i = 0; i < 5; ++i) run();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) run();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("launch under ablation %d: %.1f us (stamps off)\n", abl, ms * 50.f);
    }
    hipMemcpyToSymbol(HIP_SYMBOL(y3_timing_buf), &dt, sizeof(dt));
    for (int i = 0; i < 200; ++i) run();      // sustained load: the stamps of the last launch are the ones read back
    hipMemsetAsync(dt, 0, (size_t)maxwg * 8 * 8, nullptr);      // (slices that are not last leave before their end stamp: no stale ones from earlier launches)
    run();
    hipDeviceSynchronize();
    std::vector<unsigned long long> t((size_t)maxwg * 8);
    hipMemcpy(t.data(), dt, t.size() * 8, hipMemcpyDeviceToHost);
    // s_memtime is a per-XCD counter: only differences inside ONE workgroup are meaningful.  Slices of a split tile that do
    // not finish last leave before stamp 3: they are counted but not timed.
    int launched = 0;
    std::vector<int> done;
    for (int i = 0; i < maxwg; ++i) {
        if (t[(size_t)i * 8] == 0) continue;
        ++launched;
        if (t[(size_t)i * 8 + 3] != 0) done.push_back(i);
    }
    const int nwg = (int)done.size();
    if (nwg == 0) { printf("no workgroup reached its end stamp\n"); return 1; }
    std::vector<double> clk;
    for (int i : done) {
        const double usec = (double)(t[(size_t)i * 8 + 7] - t[(size_t)i * 8 + 6]) / 100.0;     // s_memrealtime: 100 MHz
        if (usec > 1.0) clk.push_back((double)(t[(size_t)i * 8 + 3] - t[(size_t)i * 8]) / usec / 1e3);
    }
    std::sort(clk.begin(), clk.end());
    printf("workgroups launched %d, timed to the end %d; shader clock (median over workgroups) %.2f GHz\n", launched, nwg,
           clk.empty() ? 0.0 : clk[clk.size() / 2]);
    auto stat = [&](const char* name, int a, int b) {
        std::vector<unsigned long long> d;
        for (int i : done) d.push_back(t[(size_t)i * 8 + b] - t[(size_t)i * 8 + a]);
        std::sort(d.begin(), d.end());
        printf("  %-22s min %8llu  p10 %8llu  median %8llu  p90 %8llu  max %8llu\n", name, d.front(), d[d.size() / 10], d[d.size() / 2], d[d.size() * 9 / 10], d.back());
    };
    stat("prologue", 0, 1);
    stat("main loop", 1, 2);
    stat("epilogue", 2, 3);
    stat("whole workgroup", 0, 3);
    // workgroups per CU (all launched ones): HW_ID bits [11:8] CU, [12] SH, [15:13] SE; XCC_ID low bits
    std::map<unsigned long long, int> per_cu;
    for (int i = 0; i < maxwg; ++i) {
        if (t[(size_t)i * 8] == 0) continue;
        const unsigned hw = (unsigned)t[(size_t)i * 8 + 4], xcc = (unsigned)t[(size_t)i * 8 + 5] & 0xf;
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per_cu[((unsigned long long)xcc << 16) | (se << 8) | (sh << 4) | cu]++;
    }
    std::map<int, int> hist;
    for (auto& kv : per_cu) hist[kv.second]++;
    printf("CUs used %zu; workgroups per CU:", per_cu.size());
    for (auto& kv : hist) printf("  %d x%d", kv.first, kv.second);
    printf("\n");
    if (getenv("Y3_PAIRS")) {      // which workgroup ids share a CU (two resident per CU): distance of their ids, and id % 8 against XCC_ID
        std::map<unsigned long long, std::vector<int>> ids;
        int xcc_match = 0, n = 0;
        for (int i = 0; i < maxwg; ++i) {
            if (t[(size_t)i * 8] == 0) continue;
            const unsigned hw = (unsigned)t[(size_t)i * 8 + 4], xcc = (unsigned)t[(size_t)i * 8 + 5] & 0xf;
            const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            ids[((unsigned long long)xcc << 16) | (se << 8) | (sh << 4) | cu].push_back(i);
            xcc_match += (unsigned)(i % 8) == xcc;
            ++n;
        }
        std::map<int, int> dist;
        for (auto& kv : ids)
            if (kv.second.size() == 2) dist[kv.second[1] - kv.second[0]]++;
        printf("id %% 8 == XCC_ID for %d of %d workgroups; id distance of the two workgroups of a CU:", xcc_match, n);
        for (auto& kv : dist) printf("  %d x%d", kv.first, kv.second);
        printf("\n");
    }
    // SIMD-cycles of MFMA in the whole launch: 4096 flop per 64 cycles (v_mfma_f32_32x32x2_f32); x3: 6 x 32768 flop-equivalents per 32 cycles each
    const double total_mfma_cycles = x3 ? flop * 6.0 / 32768.0 * 32.0 : flop / 4096.0 * 64.0;
    printf("MFMA floor: %.0f cycles per SIMD if spread evenly over %zu CUs x 4 SIMDs\n", total_mfma_cycles / (per_cu.size() * 4.0), per_cu.size());
    return 0;
}
