// Probe: where does a bf16 conv workgroup spend its time?  Builds conv_bf16.hip with -DY3_TIMING (per-workgroup
// s_memtime stamps: start, after prologue, after main loop, after epilogue; HW_ID, XCC_ID) and prints the distribution.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DY3_TIMING -I include -I object-detection-yolov3_amd/csrc \
//         tools/probe/bf16_timing.hip object-detection-yolov3_amd/csrc/core.hip -o tools/probe/bf16_timing
#include "../../object-detection-yolov3_amd/csrc/conv_bf16.hip"
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 8, h = argc > 2 ? atoi(argv[2]) : 76, w = h, cin = argc > 3 ? atoi(argv[3]) : 128,
              cout = argc > 4 ? atoi(argv[4]) : 256, k = argc > 5 ? atoi(argv[5]) : 3;
    const size_t xs = (size_t)n * h * w * cin, ws = (size_t)k * k * cout * cin, ys = (size_t)n * h * w * cout;
    std::vector<unsigned short> hx(xs), hw(ws);
    unsigned seed = 1;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return (unsigned short)(0x3c00 + ((seed >> 16) & 0x1ff) + ((seed >> 30) << 15)); };  // ~ +-[0.0078, 0.03]
    for (auto& v : hx) v = rnd();
    for (auto& v : hw) v = rnd();
    void *dx, *dw, *dy;
    float* db;
    unsigned long long* dt;
    hipMalloc(&dx, xs * 2); hipMalloc(&dw, ws * 2); hipMalloc(&dy, ys * 2); hipMalloc(&db, cout * 4);
    hipMemcpy(dx, hx.data(), xs * 2, hipMemcpyHostToDevice);
    hipMemcpy(dw, hw.data(), ws * 2, hipMemcpyHostToDevice);
    hipMemset(db, 0, cout * 4);
    y3_tensor src{(float*)dx, n, h, w, cin, cin}, dst{(float*)dy, n, h, w, cout, cout};
    for (int i = 0; i < 20; ++i) y3_conv2d_fwd_bf16(&src, dw, db, k, 1, &dst, 0, Y3_EPI_LRELU, 0.2f, nullptr, nullptr, nullptr, nullptr);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) y3_conv2d_fwd_bf16(&src, dw, db, k, 1, &dst, 0, Y3_EPI_LRELU, 0.2f, nullptr, nullptr, nullptr, nullptr);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("layer %dx%dx%d %d->%d k%d: %.1f us per launch (stamps off)\n", n, h, w, cin, cout, k, ms * 50.f);
    const int maxwg = 1 << 16;
    hipMalloc(&dt, (size_t)maxwg * 8 * 8);
    hipMemset(dt, 0, (size_t)maxwg * 8 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(y3_timing_buf), &dt, sizeof(dt));
    y3_conv2d_fwd_bf16(&src, dw, db, k, 1, &dst, 0, Y3_EPI_LRELU, 0.2f, nullptr, nullptr, nullptr, nullptr);
    hipDeviceSynchronize();
    std::vector<unsigned long long> t((size_t)maxwg * 8);
    hipMemcpy(t.data(), dt, t.size() * 8, hipMemcpyDeviceToHost);
    int nwg = 0;
    while (nwg < maxwg && t[(size_t)nwg * 8 + 3] != 0) ++nwg;
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int i = 0; i < nwg; ++i) { tmin = std::min(tmin, t[i * 8]); tmax = std::max(tmax, t[i * 8 + 3]); }
    printf("workgroups %d, kernel span %llu ticks (s_memtime)\n", nwg, tmax - tmin);
    auto stat = [&](const char* name, int a, int b) {
        std::vector<unsigned long long> d;
        for (int i = 0; i < nwg; ++i) d.push_back(t[i * 8 + b] - t[i * 8 + a]);
        std::sort(d.begin(), d.end());
        printf("  %-22s min %8llu  median %8llu  max %8llu\n", name, d.front(), d[d.size() / 2], d.back());
    };
    stat("prologue", 0, 1);
    stat("main loop", 1, 2);
    stat("epilogue", 2, 3);
    stat("whole workgroup", 0, 3);
    {
        std::vector<unsigned long long> s;
        for (int i = 0; i < nwg; ++i) s.push_back(t[i * 8] - tmin);
        std::sort(s.begin(), s.end());
        printf("  start offsets: p0 %llu p25 %llu p50 %llu p75 %llu p100 %llu\n", s[0], s[nwg / 4], s[nwg / 2], s[3 * nwg / 4], s[nwg - 1]);
        std::vector<unsigned long long> e;
        for (int i = 0; i < nwg; ++i) e.push_back(t[i * 8 + 3] - tmin);
        std::sort(e.begin(), e.end());
        printf("  end offsets:   p0 %llu p25 %llu p50 %llu p75 %llu p100 %llu\n", e[0], e[nwg / 4], e[nwg / 2], e[3 * nwg / 4], e[nwg - 1]);
    }
    std::map<unsigned, int> per_cu;
    for (int i = 0; i < nwg; ++i) {
        const unsigned hw = (unsigned)t[i * 8 + 4], xcc = (unsigned)t[i * 8 + 5] & 15;
        const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu]++;
    }
    std::map<int, int> hist;
    for (auto& kv : per_cu) hist[kv.second]++;
    printf("  distinct (xcc, se, sh, cu) slots used: %zu;  workgroups per slot histogram:", per_cu.size());
    for (auto& kv : hist) printf("  %d wg x %d", kv.first, kv.second);
    printf("\n");
    return 0;
}
