// pk_issue.hip -- what does a v_pk_fma_f32 cost a wave that is ALONE on its SIMD (1024 waves of 64 lanes, one per SIMD), next to
// v_fma_f32, as a function of the number of independent accumulators K it interleaves and of the operand form (plain pair, scalar
// broadcast through op_sel)?  Times are per INSTRUCTION; the in-kernel clock is reported from s_memtime / s_memrealtime.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ unsigned long long g_clk[2];
template <int K, int MODE>   // MODE 0: v_fma_f32; 1: v_pk_fma_f32 pair * pair; 2: v_pk_fma_f32 with a broadcast half of another register
__global__ __launch_bounds__(64) void k_chain(float* out, float a, float b, int n)
{
    f2 v[K], c[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { v[k] = f2{(float)(threadIdx.x + k), (float)k}; c[k] = f2{a + k * 1e-3f, a - k * 1e-3f}; }
    const f2 bb = {b, b * 0.5f};
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (MODE == 0) v[k].x = fmaf(v[k].x, c[k].x, bb.x);
                else if (MODE == 1) v[k] = v[k] * c[k] + bb;
                else { const f2 s = {c[(k + 1) % K].y, c[(k + 1) % K].y}; v[k] = v[k] * s + bb; }
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) s += v[k].x + v[k].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (blockIdx.x == 17 && threadIdx.x == 0) { g_clk[0] = t1 - t0; g_clk[1] = r1 - r0; }
}
template <int K, int MODE>
static void run(float* out, int wps)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int n = 4000;
    k_chain<K, MODE><<<1024 * wps, 64>>>(out, 0.999f, 0.001f, 10);
    (void)hipEventRecord(e0);
    k_chain<K, MODE><<<1024 * wps, 64>>>(out, 0.999f, 0.001f, n);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long clk[2];
    (void)hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clk), sizeof(clk));
    const double instr = (double)n * 16 * K, ghz = (double)clk[0] / ((double)clk[1] * 10.0);
    const char* nm[3] = {"v_fma_f32           ", "v_pk_fma_f32        ", "v_pk_fma_f32 op_sel "};
    printf("%s K=%2d waves/SIMD=%d: %.2f ns per instruction per wave = %.2f cycles at the in-kernel clock %.2f GHz\n", nm[MODE], K, wps, ms * 1e6 / instr,
           ms * 1e6 / instr * ghz, ghz);
}
int main()
{
    float* out;
    (void)hipMalloc(&out, 4 * 64 * 1024 * 8);
    for (int w : {1, 2}) {
        run<1, 0>(out, w); run<2, 0>(out, w); run<4, 0>(out, w); run<8, 0>(out, w); run<16, 0>(out, w);
        run<1, 1>(out, w); run<2, 1>(out, w); run<4, 1>(out, w); run<8, 1>(out, w); run<16, 1>(out, w);
        run<2, 2>(out, w); run<4, 2>(out, w); run<8, 2>(out, w); run<16, 2>(out, w);
    }
    return 0;
}
