// valu_issue.hip -- what does one fp32 / fp64 FMA cost a wave that is ALONE on its SIMD, as a function of how many independent
// chains it interleaves?  1024 waves of 64 lanes (one per SIMD), K chains of dependent FMAs each, N rounds.
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T, int K>
__global__ __launch_bounds__(64) void k_chain(T* out, T a, T b, int n)
{
    T v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = (T)(threadIdx.x + k);
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int k = 0; k < K; ++k) v[k] = fma(v[k], a, b);
        }
    }
    T s = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) s += v[k];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <typename T, int K>
static void run(const char* name, T* out, int waves_per_simd)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int n = 2000;
    k_chain<T, K><<<1024 * waves_per_simd, 64>>>(out, (T)0.999, (T)0.001, 10);
    (void)hipEventRecord(e0);
    k_chain<T, K><<<1024 * waves_per_simd, 64>>>(out, (T)0.999, (T)0.001, n);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)n * 16 * K;
    printf("%s K=%d waves/SIMD=%d: %.2f ns per FMA per wave  (%.2f cycles at 2.4 GHz; SIMD-level %.2f cycles per FMA)\n", name, K, waves_per_simd,
           ms * 1e6 / instr, ms * 1e6 / instr * 2.4, ms * 1e6 / instr * 2.4 / waves_per_simd);
}
int main()
{
    void* out;
    (void)hipMalloc(&out, 8 * 64 * 1024 * 8);
    for (int w : {1, 2, 4}) {
        run<float, 1>("f32", (float*)out, w); run<float, 2>("f32", (float*)out, w); run<float, 4>("f32", (float*)out, w); run<float, 8>("f32", (float*)out, w);
        run<double, 1>("f64", (double*)out, w); run<double, 2>("f64", (double*)out, w); run<double, 4>("f64", (double*)out, w);
    }
    return 0;
}
