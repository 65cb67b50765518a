// Phase timing of csrc/xattn2.hip's rows kernel (scores + softmax) from inside: every workgroup stamps the shader clock at its start,
// after the prologue, at the top of every k step before / after the wait + barrier, after the k loop, after the softmax and at the end.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DX2_PROBE -I spn4cir_amd/csrc -I include tools/x2probe/x2probe.hip -o tools/x2probe/x2probe
//   tools/x2probe/x2probe [B=128] [L=32]
#include "../../spn4cir_amd/csrc/xattn2.hip"
#include <algorithm>
#include <cstdio>
#include <vector>
const char* spn_env(const char*) { return nullptr; }
namespace spn { int xattn_sp(int S) { return S <= 256 ? 256 : 640; } }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 128, L = argc > 2 ? atoi(argv[2]) : 32, H = 12, S = 577, E = 768, R = L * H, SP = 640;
    const size_t nq = (size_t)B * R * E, nx = (size_t)B * S * E, np = (size_t)B * R * SP;
    std::vector<uint16_t> h(std::max(nq, nx));
    uint32_t st = 12345;
    for (auto& v : h) { st = st * 1664525u + 1013904223u; v = (uint16_t)(0x3c00 + ((st >> 16) & 0x1ff) - (((st >> 8) & 1) << 15)); }
    bf16_t *q, *x, *p;
    CK(hipMalloc(&q, nq * 2)); CK(hipMalloc(&x, nx * 2)); CK(hipMalloc(&p, np * 2));
    CK(hipMemcpy(q, h.data(), nq * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(x, h.data(), nx * 2, hipMemcpyHostToDevice));
    const int wgs = B * ((R + 95) / 96);
    uint64_t* buf;
    CK(hipMalloc(&buf, (size_t)wgs * X2_PROBE_SLOTS * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(spn::x2_probe_buf), &buf, sizeof(buf)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 3; ++it) {
        CK(hipEventRecord(e0, 0));
        int rc = spn::xattn2_scores_softmax(q, x, p, B, R, S, E, 0, nullptr, H);
        if (rc) { printf("launch rc %d\n", rc); return 1; }
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("launch %d: %.1f us\n", it, ms * 1000);
    }
    std::vector<uint64_t> t((size_t)wgs * X2_PROBE_SLOTS);
    CK(hipMemcpy(t.data(), buf, t.size() * 8, hipMemcpyDeviceToHost));
    const int nk = E / 32;
    auto med = [&](auto f) { std::vector<double> v; for (int w = 0; w < wgs; ++w) v.push_back(f(&t[(size_t)w * X2_PROBE_SLOTS])); std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    uint64_t tmin = ~0ull, tmax = 0;
    for (int w = 0; w < wgs; ++w) { tmin = std::min(tmin, t[(size_t)w * X2_PROBE_SLOTS]); tmax = std::max(tmax, t[(size_t)w * X2_PROBE_SLOTS + 4]); }
    printf("workgroups %d; first start -> last end %.0f cycles\n", wgs, (double)(tmax - tmin));
    printf("median per workgroup (shader cycles): prologue %.0f  k loop %.0f  softmax %.0f  stores %.0f  total %.0f\n",
           med([](const uint64_t* s) { return (double)(s[1] - s[0]); }), med([](const uint64_t* s) { return (double)(s[2] - s[1]); }),
           med([](const uint64_t* s) { return (double)(s[3] - s[2]); }), med([](const uint64_t* s) { return (double)(s[4] - s[3]); }),
           med([](const uint64_t* s) { return (double)(s[4] - s[0]); }));
    printf("k step (median cycles), wave 0 | wave 7:  vmcnt wait, lgkm + barrier, compute\n");
    for (int kt = 0; kt < nk; ++kt) {
        double v[6];
        for (int w = 0; w < 2; ++w) {
            const int o = w * 96;
            v[3 * w + 0] = med([&](const uint64_t* s) { return (double)(s[o + 9 + 3 * kt] - s[o + 8 + 3 * kt]); });
            v[3 * w + 1] = med([&](const uint64_t* s) { return (double)(s[o + 10 + 3 * kt] - s[o + 9 + 3 * kt]); });
            v[3 * w + 2] = med([&](const uint64_t* s) { return (double)((kt + 1 < nk ? s[o + 11 + 3 * kt] : s[o + 2]) - s[o + 10 + 3 * kt]); });
        }
        printf("  %2d: %6.0f %6.0f %6.0f | %6.0f %6.0f %6.0f\n", kt, v[0], v[1], v[2], v[3], v[4], v[5]);
    }
    printf("wave 7 - wave 0 arrival skew at the top of k step 12 (median): %.0f\n", med([&](const uint64_t* s) { return (double)((int64_t)(s[96 + 8 + 36] - s[8 + 36])); }));
    return 0;
}
