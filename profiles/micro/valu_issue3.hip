// valu_issue3.hip -- is a lone wave limited by instruction SUPPLY on long straight-line code?  The same 32-accumulator fp32 FMA
// pattern (three VGPR operands, VOP3) as a small loop body (64 FMAs per iteration) and as one straight-line stream of `LEN` FMAs
// executed `n` times (so the code footprint is LEN x 8 bytes and every instruction is fetched again on each pass).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LEN>
__global__ __launch_bounds__(64) void k_line(float* out, const float* in, int n)
{
    float v[32], x[8], y[8];
#pragma unroll
    for (int k = 0; k < 32; ++k) v[k] = in[threadIdx.x + k];
#pragma unroll
    for (int k = 0; k < 8; ++k) { x[k] = in[64 + threadIdx.x + k]; y[k] = in[128 + threadIdx.x + k]; }
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int r = 0; r < LEN; ++r) v[r % 32] = fmaf(-x[(r / 32) % 8], y[(r * 7 / 32) % 8], v[r % 32]);
        asm volatile("" ::: "memory");
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < 32; ++k) s += v[k];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int LEN>
static void run(float* out, float* in, int waves)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int n = 256 * 1024 / LEN * 8;
    k_line<LEN><<<1024 * waves, 64>>>(out, in, 4);
    (void)hipEventRecord(e0);
    k_line<LEN><<<1024 * waves, 64>>>(out, in, n);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)n * LEN;
    printf("straight-line %5d FMAs (%3d KiB of code), %d wave(s)/SIMD: %.2f ns per FMA per wave (%.2f cycles at 2.4 GHz)\n", LEN, LEN * 8 / 1024, waves,
           ms * 1e6 / instr, ms * 1e6 / instr * 2.4);
}
int main()
{
    float *out, *in;
    (void)hipMalloc(&out, 4 * 64 * 1024 * 4);
    (void)hipMalloc(&in, 4 * 1024);
    (void)hipMemset(in, 0, 4 * 1024);
    for (int w : {1, 2}) {
        run<64>(out, in, w); run<512>(out, in, w); run<2048>(out, in, w); run<4096>(out, in, w); run<8192>(out, in, w);
    }
    return 0;
}
