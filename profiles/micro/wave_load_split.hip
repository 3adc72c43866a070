// wave_load_split.hip -- how long does the wave that holds the 16 covariance quads of a quarter-tile workgroup (kw_tick, cfg 2) wait
// for its 15 KiB, and would the four waves of the workgroup fetching a quarter each be faster?
// 256 workgroups of 256 threads, one per CU, each reads (or writes) the 60 rows x 256 B of its quarter of a 64-filter wave tile
// (row pitch 1 KiB, fp64 quads of 16 B per filter), non-temporal like the engine's; a store launch precedes every load launch, as in
// the engine (the previous tick's stores).  Stamps (s_memtime) by lane 0 of every wave.
//   MODE 0  rows    : wave 0 alone, lane l -> 16 B of filter l % 16 in row 4 k + l / 16      (15 instructions, 4 x 256 B runs each)
//   MODE 1  rows/4  : the same rows dealt round-robin over the four waves                     (4 + 4 + 4 + 3 instructions)
//   MODE 2  quads   : wave 0 alone in the engine's pattern, lane 4 f + j (j < 3) -> filter f, row 6 m + 2 j + h   (20 instructions,
//                     3 x 256 B runs each, every fourth lane idle)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int kRows = 60;

template <int MODE>
__global__ __launch_bounds__(256) void k_probe(float* buf, unsigned long long* clk, int store)
{
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    float* tile = buf + (size_t)(blockIdx.x >> 2) * (kRows * 256) + (blockIdx.x & 3) * 64;   // floats: row pitch 256, quarter offset 16 x 4
    constexpr int N = MODE == 2 ? 20 : 15;
    f4 v[N];
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = f4{(float)k, 1.f, 2.f, 3.f};
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int k = 0; k < N; ++k) {
        bool mine; int row, f;
        if (MODE == 2) { const int j = l & 3; f = l >> 2; row = 6 * (k >> 1) + 2 * j + (k & 1); mine = w == 0 && j < 3; }
        else { f = l & 15; row = 4 * k + (l >> 4); mine = MODE == 0 ? w == 0 : (k & 3) == w; }
        f4* p = reinterpret_cast<f4*>(tile + row * 256 + f * 4);
        if (mine) { if (store) __builtin_nontemporal_store(v[k], p); else v[k] = __builtin_nontemporal_load(p); }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < N; ++k) s += v[k].x + v[k].w;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_readcyclecounter();
    if (l == 0) {
        unsigned long long* c = clk + ((size_t)blockIdx.x * 4 + w) * 4;
        c[0] = t0; c[1] = t1; c[2] = t2; c[3] = (unsigned long long)(s != 12345.f);
    }
}

template <int MODE> static void launch(float* buf, unsigned long long* clk, int store) { k_probe<MODE><<<256, 256>>>(buf, clk, store); }

int main()
{
    const int G = 256;
    float* buf; unsigned long long *clk, *clk2;
    (void)hipMalloc(&buf, (size_t)(G / 4) * kRows * 1024); (void)hipMemset(buf, 0, (size_t)(G / 4) * kRows * 1024);
    (void)hipMalloc(&clk, (size_t)G * 16 * 8); (void)hipMalloc(&clk2, (size_t)G * 16 * 8);
    std::vector<unsigned long long> h(G * 16);
    const char* names[3] = {"rows, wave 0 alone (15 x 1 KiB)", "rows, four waves (4+4+4+3 x 1 KiB)", "quads, wave 0 alone (20 x 768 B, the engine's pattern)"};
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 30; ++rep) {
            if (mode == 0) { launch<0>(buf, clk2, 1); launch<0>(buf, clk, 0); }
            else if (mode == 1) { launch<1>(buf, clk2, 1); launch<1>(buf, clk, 0); }
            else { launch<2>(buf, clk2, 1); launch<2>(buf, clk, 0); }
        }
        (void)hipDeviceSynchronize();
        for (int store = 0; store < 2; ++store) {
            (void)hipMemcpy(h.data(), store ? clk2 : clk, h.size() * 8, hipMemcpyDeviceToHost);
            std::vector<double> issue, done, span;
            for (int g = 0; g < G; ++g) {
                unsigned long long first = ~0ull, last = 0;
                for (int w = 0; w < 4; ++w) { first = std::min(first, h[(g * 4 + w) * 4]); last = std::max(last, h[(g * 4 + w) * 4 + 2]); }
                issue.push_back((double)(h[g * 16 + 1] - h[g * 16]));
                done.push_back((double)(h[g * 16 + 2] - h[g * 16]));
                span.push_back((double)(last - first));
            }
            auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
            printf("%-6s %-58s wave 0: issued after %5.0f, %s after %5.0f; first entry -> last wave done %5.0f [s_memtime ticks]\n",
                   store ? "stores" : "loads", names[mode], med(issue), store ? "written" : "arrived", med(done), med(span));
        }
    }
    return 0;
}
