// valu_issue2.hip -- fp32 FMA with three VGPR operands (acc = fma(x, y, acc), x / y from other registers) for one wave per SIMD:
// independent accumulators, operands shared or distinct.  Complements valu_issue.hip (FMA with scalar operands).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int K, int MODE>
__global__ __launch_bounds__(64) void k_fma3(float* out, const float* in, int n)
{
    float v[K], x[K], y[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { v[k] = in[threadIdx.x + k]; x[k] = in[64 + threadIdx.x + k] ; y[k] = in[128 + threadIdx.x + k]; }
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (MODE == 0) v[k] = fmaf(x[k], y[k], v[k]);            // three distinct VGPRs per FMA
                else if (MODE == 1) v[k] = fmaf(x[0], y[k], v[k]);       // one operand shared by all (rank-1 update shape)
                else v[k] = fmaf(-x[k % 4], y[(k + r) % K], v[k]);       // rank-1 update shape with rotating operands
            }
        }
        asm volatile("" ::: "memory");
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) s += v[k];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int K, int MODE>
static void run(float* out, float* in)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int n = 4000;
    k_fma3<K, MODE><<<1024, 64>>>(out, in, 10);
    (void)hipEventRecord(e0);
    k_fma3<K, MODE><<<1024, 64>>>(out, in, n);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)n * 8 * K;
    printf("K=%d mode=%d: %.2f ns per FMA (%.2f cycles at 2.4 GHz)\n", K, MODE, ms * 1e6 / instr, ms * 1e6 / instr * 2.4);
}
int main()
{
    float *out, *in;
    (void)hipMalloc(&out, 4 * 64 * 1024);
    (void)hipMalloc(&in, 4 * 1024);
    (void)hipMemset(in, 0, 4 * 1024);
    run<8, 0>(out, in); run<16, 0>(out, in); run<32, 0>(out, in);
    run<8, 1>(out, in); run<16, 1>(out, in); run<32, 1>(out, in);
    run<8, 2>(out, in); run<16, 2>(out, in); run<32, 2>(out, in);
    return 0;
}
