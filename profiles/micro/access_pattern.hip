// access_pattern.hip -- how fast can a wave stream the 144-word state records (wave tiles) with different lane -> address maps?
//   lane : one lane per filter, lane l reads quad row k at (k*64 + l)*16           (k_predict's pattern)
//   quad : four lanes per filter, lane 4f+j (j<3) reads quad row 4+3m+j at filter f  (first cooperative kernel: 64-byte runs)
//   pack : four lanes per filter on a layout whose rows are [filter][lane j] (48 contiguous bytes per filter)
// Each variant reads the P part (30 quads per filter) and writes it back (+1.0f), in place, B filters.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int kSW = 144, kTile = 64;

template <int MODE>
__global__ __launch_bounds__(256) void k_copy(float* st, long B)
{
    if (MODE == 0) {
        const long i = (long)blockIdx.x * 256 + threadIdx.x;
        if (i >= B) return;
        float* tb = st + (i >> 6) * (long)(kSW * kTile);
        const int l = (int)(i & 63);
        f4 v[30];
#pragma unroll
        for (int k = 0; k < 30; ++k) v[k] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(tb + ((4 + k) * kTile + l) * 4));
#pragma unroll
        for (int k = 0; k < 30; ++k) __builtin_nontemporal_store(v[k] + 1.0f, reinterpret_cast<f4*>(tb + ((4 + k) * kTile + l) * 4));
    } else {
        const long tile = blockIdx.x;
        const int f = threadIdx.x >> 2, j = threadIdx.x & 3;
        if (tile * 64 + f >= B || j == 3) return;
        float* tb = st + tile * (long)(kSW * kTile);
        f4 v[10];
#pragma unroll
        for (int m = 0; m < 10; ++m) {
            const long off = MODE == 1 ? ((4 + 3 * m + j) * kTile + f) * 4 : (4 * kTile * 4) + ((long)m * kTile * 3 + f * 3 + j) * 4;
            v[m] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(tb + off));
        }
#pragma unroll
        for (int m = 0; m < 10; ++m) {
            const long off = MODE == 1 ? ((4 + 3 * m + j) * kTile + f) * 4 : (4 * kTile * 4) + ((long)m * kTile * 3 + f * 3 + j) * 4;
            __builtin_nontemporal_store(v[m] + 1.0f, reinterpret_cast<f4*>(tb + off));
        }
    }
}

int main(int argc, char** argv)
{
    const long B = argc > 1 ? atol(argv[1]) : 65536;
    float* st;
    hipMalloc(&st, sizeof(float) * kSW * B);
    hipMemset(st, 0, sizeof(float) * kSW * B);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[3] = {"lane", "quad", "pack"};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 3; ++mode) {
            const int N = 300;
            dim3 g(mode == 0 ? (unsigned)((B + 255) / 256) : (unsigned)(B / 64));
            for (int k = 0; k < 20; ++k) {
                if (mode == 0) k_copy<0><<<g, 256>>>(st, B); else if (mode == 1) k_copy<1><<<g, 256>>>(st, B); else k_copy<2><<<g, 256>>>(st, B);
            }
            hipEventRecord(e0);
            for (int k = 0; k < N; ++k) {
                if (mode == 0) k_copy<0><<<g, 256>>>(st, B); else if (mode == 1) k_copy<1><<<g, 256>>>(st, B); else k_copy<2><<<g, 256>>>(st, B);
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("B=%ld %s: %.2f us per launch, %.0f GB/s (120 words read + written per filter)\n", B, names[mode], ms / N * 1e3, 2.0 * 480 * B / (ms / N * 1e-3) / 1e9);
        }
    return 0;
}
