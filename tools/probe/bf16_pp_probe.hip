// Probe for conv_bf16_pp_kernel: launch time per shape, per-workgroup phase stamps, and the K loop with parts ablated
// (results become wrong, timing stays meaningful): 1 = no LDS-DMA in the loop, 2 = no fragment reads, 4 = no MFMAs,
// 8 = no vmcnt wait.  Builds conv_bf16.hip with -DY3_TIMING:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DY3_TIMING -I include -I object-detection-yolov3_amd/csrc \
//         tools/probe/bf16_pp_probe.hip object-detection-yolov3_amd/csrc/core.hip -o tools/probe/bf16_pp_probe
//   tools/probe/bf16_pp_probe [n h cin cout k]
#include "../../object-detection-yolov3_amd/csrc/conv_bf16.hip"
#include <algorithm>
#include <cstdio>
#include <vector>

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 25, h = argc > 2 ? atoi(argv[2]) : 76, w = h, cin = argc > 3 ? atoi(argv[3]) : 128,
              cout = argc > 4 ? atoi(argv[4]) : 256, k = argc > 5 ? atoi(argv[5]) : 3;
    const size_t xs = (size_t)n * h * w * cin, ws = (size_t)k * k * cout * cin, ys = (size_t)n * h * w * cout;
    std::vector<unsigned short> hx(xs), hw(ws);
    unsigned seed = 1;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return (unsigned short)(0x3c00 + ((seed >> 16) & 0x1ff) + ((seed >> 30) << 15)); };
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
    auto run = [&]() { return y3_conv2d_fwd_bf16(&src, dw, db, k, 1, &dst, 0, Y3_EPI_LRELU, 0.2f, nullptr, nullptr, nullptr, nullptr); };
    const double flop = 2.0 * n * h * w * k * k * cin * cout;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int masks[] = {0, 1, 2, 4, 8, 3, 5, 6, 7, 9};
    for (int abl : masks) {
        hipMemcpyToSymbol(HIP_SYMBOL(y3_pp_abl), &abl, sizeof(abl));
        if (run() != 0) { printf("launch failed: %s\n", y3_last_error()); return 1; }
        for (int i = 0; i < 10; ++i) run();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) run();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("layer %dx%dx%d %d->%d k%d ablation %d: %.1f us per launch = %.1f TFLOP/s\n", n, h, w, cin, cout, k, abl, ms * 50.f, flop / (ms * 50e-6) / 1e12);
    }
    int abl = 0;
    hipMemcpyToSymbol(HIP_SYMBOL(y3_pp_abl), &abl, sizeof(abl));
    const int maxwg = 1 << 14;
    hipMalloc(&dt, (size_t)maxwg * 8 * 8);
    hipMemset(dt, 0, (size_t)maxwg * 8 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(y3_timing_buf), &dt, sizeof(dt));
    run();
    hipDeviceSynchronize();
    std::vector<unsigned long long> t((size_t)maxwg * 8);
    hipMemcpy(t.data(), dt, t.size() * 8, hipMemcpyDeviceToHost);
    int nwg = 0;
    while (nwg < maxwg && t[(size_t)nwg * 8 + 3] != 0) ++nwg;
    if (nwg == 0) { printf("no stamps\n"); return 0; }
    auto stat = [&](const char* name, int a, int b) {
        std::vector<unsigned long long> d;
        for (int i = 0; i < nwg; ++i) d.push_back(t[(size_t)i * 8 + b] - t[(size_t)i * 8 + a]);
        std::sort(d.begin(), d.end());
        printf("  %-22s min %8llu  median %8llu  max %8llu\n", name, d.front(), d[d.size() / 2], d.back());
    };
    printf("workgroups %d; K tiles per workgroup %d (64 deep, 4 phases each)\n", nwg, k * k * cin / 64);
    stat("prologue", 0, 1);
    stat("main loop", 1, 2);
    stat("epilogue", 2, 3);
    stat("whole workgroup", 0, 3);
    return 0;
}
