// ekf_kernels.hpp -- HIP kernels of the batched EKF engine (gfx950).
//
// Data layout in HBM ("quad rows"): a per-filter record of W words of type T
// is stored as rows of 16-byte quads, row k holding words [k*VW, (k+1)*VW) of
// every filter (VW = 4 for fp32, 2 for fp64):
//     addr(word w, filter i) = ((w / VW) * B + i) * VW + (w % VW)
// so lane i of a wave reads one aligned 16-byte quad per row and a wave reads
// 1 KiB contiguous (global_load_dwordx4 per lane, fully coalesced).  A record
// whose length is not a multiple of VW (fp32 u: 6 words) ends in one row of
// 8-byte halves.  One lane owns one filter; x (16 words) and the packed
// symmetric P (120 words) live in VGPRs for the whole tick.
#pragma once

#include "ekf_device.hpp"

namespace qle {

constexpr int kBlock = 256;
constexpr int kXW = 16;   // state words
constexpr int kPW = 120;  // packed covariance words
constexpr int kUW = 6;    // IMU words
constexpr int kZW = 8;    // tag pose 7 words + mask word
constexpr int kFW = 24;   // per-filter parameter words

template <typename T> struct Quad;
template <> struct Quad<float> { using type = float4; static constexpr int VW = 4; };
template <> struct Quad<double> { using type = double2; static constexpr int VW = 2; };

__device__ __forceinline__ void unpack_quad(const float4& v, float* r) { r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w; }
__device__ __forceinline__ void unpack_quad(const double2& v, double* r) { r[0] = v.x; r[1] = v.y; }
__device__ __forceinline__ float4 pack_quad(const float* r) { return make_float4(r[0], r[1], r[2], r[3]); }
__device__ __forceinline__ double2 pack_quad(const double* r) { return make_double2(r[0], r[1]); }

// Offset (in words) of word w of filter i in a W-word record array.
template <typename T>
__host__ __device__ inline int64_t word_off(int w, int64_t i, int64_t B, int W)
{
    constexpr int VW = 16 / (int)sizeof(T);
    const int nf = W / VW;
    if (w < nf * VW) return ((int64_t)(w / VW) * B + i) * VW + (w % VW);
    const int rem = W - nf * VW;
    return (int64_t)nf * VW * B + i * rem + (w - nf * VW);
}

template <typename T, int W>
__device__ __forceinline__ void load_rec(const T* __restrict__ base, int64_t B, int64_t i, T (&r)[W])
{
    using Q = typename Quad<T>::type;
    constexpr int VW = Quad<T>::VW;
    constexpr int NF = W / VW;
    constexpr int REM = W % VW;
    const Q* q = reinterpret_cast<const Q*>(base);
#pragma unroll
    for (int k = 0; k < NF; ++k) {
        Q v = q[(int64_t)k * B + i];
        unpack_quad(v, &r[k * VW]);
    }
    if (REM == 2) {  // fp32 only: trailing row of 8-byte halves
        const float2* h = reinterpret_cast<const float2*>(base + (int64_t)NF * VW * B);
        float2 v = h[i];
        r[NF * VW] = v.x;
        r[NF * VW + 1] = v.y;
    }
    static_assert(REM == 0 || REM == 2, "record tail must be empty or one 8-byte half");
}

template <typename T, int W>
__device__ __forceinline__ void store_rec(T* __restrict__ base, int64_t B, int64_t i, const T (&r)[W])
{
    using Q = typename Quad<T>::type;
    constexpr int VW = Quad<T>::VW;
    constexpr int NF = W / VW;
    static_assert(W % VW == 0, "stored records are whole quads");
    Q* q = reinterpret_cast<Q*>(base);
#pragma unroll
    for (int k = 0; k < NF; ++k) q[(int64_t)k * B + i] = pack_quad(&r[k * VW]);
}

template <typename T, bool PFP>
__device__ __forceinline__ void load_noise(const DevParams<T>& p, const T* __restrict__ pfp, int64_t B, int64_t i, Noise<T>& nz)
{
    if (PFP) {
        T f[kFW];
        load_rec<T, kFW>(pfp, B, i, f);
#pragma unroll
        for (int k = 0; k < 12; ++k) nz.Q[k] = f[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) { nz.ab_static[k] = f[12 + k]; nz.wb_static[k] = f[15 + k]; }
#pragma unroll
        for (int k = 0; k < 6; ++k) nz.R[k] = f[18 + k];
    } else {
#pragma unroll
        for (int k = 0; k < 12; ++k) nz.Q[k] = p.Q[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) { nz.ab_static[k] = p.ab_static[k]; nz.wb_static[k] = p.wb_static[k]; }
#pragma unroll
        for (int k = 0; k < 6; ++k) nz.R[k] = p.R[k];
    }
}

// ------------------------------------------------------------- hot kernels
// Predict tick: reads x16 + P120 + u6, writes x16 + P120 (278 words/filter).
template <typename T, bool PFP, bool AUX>
__global__ __launch_bounds__(kBlock) void k_predict(DevParams<T> p, T* __restrict__ xs, T* __restrict__ Ps,
                                                    const T* __restrict__ us, const T* __restrict__ pfp,
                                                    T* __restrict__ aux_accel, int64_t B)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    T x[kXW], P[kPW], u[kUW], accel[3];
    load_rec<T, kUW>(us, B, i, u);
    load_rec<T, kXW>(xs, B, i, x);
    load_rec<T, kPW>(Ps, B, i, P);
    Noise<T> nz;
    load_noise<T, PFP>(p, pfp, B, i, nz);
    ekf_predict<T>(p, nz, x, P, u, accel);
    store_rec<T, kXW>(xs, B, i, x);
    store_rec<T, kPW>(Ps, B, i, P);
    if (AUX) {  // side output, AoS [B][3] in the compute dtype
#pragma unroll
        for (int k = 0; k < 3; ++k) aux_accel[i * 3 + k] = accel[k];
    }
}

// Fused tick (filter_update single-rate branch, EKF.cpp:238-249,265-290):
// predict, then correct where the record's mask word is non-zero.
// Reads x16 + P120 + u6 + z7 (+mask), writes x16 + P120 (285 words/filter).
template <typename T, bool DIRECT, bool PFP, bool AUX>
__global__ __launch_bounds__(kBlock) void k_step(DevParams<T> p, T* __restrict__ xs, T* __restrict__ Ps,
                                                 const T* __restrict__ us, const T* __restrict__ zs,
                                                 const T* __restrict__ pfp, T* __restrict__ aux_accel,
                                                 T* __restrict__ aux_obs, int64_t B)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    T x[kXW], P[kPW], u[kUW], zr[kZW], accel[3];
    load_rec<T, kUW>(us, B, i, u);
    load_rec<T, kZW>(zs, B, i, zr);
    load_rec<T, kXW>(xs, B, i, x);
    load_rec<T, kPW>(Ps, B, i, P);
    Noise<T> nz;
    load_noise<T, PFP>(p, pfp, B, i, nz);
    ekf_predict<T>(p, nz, x, P, u, accel);
    T obs[7] = {T(0), T(0), T(0), T(0), T(0), T(0), T(1)};
    const bool corr = zr[7] != T(0);
    if (corr) {
        T z[7] = {zr[0], zr[1], zr[2], zr[3], zr[4], zr[5], zr[6]};
        ekf_update<T, DIRECT>(p, nz, x, P, z, obs);
    }
    store_rec<T, kXW>(xs, B, i, x);
    store_rec<T, kPW>(Ps, B, i, P);
    if (AUX) {
#pragma unroll
        for (int k = 0; k < 3; ++k) aux_accel[i * 3 + k] = accel[k];
        if (corr) {
#pragma unroll
            for (int k = 0; k < 7; ++k) aux_obs[i * 7 + k] = obs[k];
        }
    }
}

// Stand-alone correction (correction_step, EKF.cpp:417-502) where mask != 0.
template <typename T, bool DIRECT, bool PFP, bool AUX>
__global__ __launch_bounds__(kBlock) void k_update(DevParams<T> p, T* __restrict__ xs, T* __restrict__ Ps,
                                                   const T* __restrict__ zs, const T* __restrict__ pfp,
                                                   T* __restrict__ aux_obs, int64_t B)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    T zr[kZW];
    load_rec<T, kZW>(zs, B, i, zr);
    if (zr[7] == T(0)) return;
    T x[kXW], P[kPW];
    load_rec<T, kXW>(xs, B, i, x);
    load_rec<T, kPW>(Ps, B, i, P);
    Noise<T> nz;
    load_noise<T, PFP>(p, pfp, B, i, nz);
    T z[7] = {zr[0], zr[1], zr[2], zr[3], zr[4], zr[5], zr[6]};
    T obs[7];
    ekf_update<T, DIRECT>(p, nz, x, P, z, obs);
    store_rec<T, kXW>(xs, B, i, x);
    store_rec<T, kPW>(Ps, B, i, P);
    if (AUX) {
#pragma unroll
        for (int k = 0; k < 7; ++k) aux_obs[i * 7 + k] = obs[k];
    }
}

// ------------------------------------------------ layout conversion kernels
// Host-facing AoS fp64 <-> device quad rows, one chunk [i0, i0+n) of the batch
// per launch (the AoS side is a staging buffer holding only that chunk).
// Not on the hot path.
template <typename T>
__global__ void k_pack_off(const double* __restrict__ aos, int stride, int W, T* __restrict__ dst, int64_t B, int64_t i0, int64_t n)
{
    const int64_t li = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= n) return;
    for (int w = 0; w < W; ++w) dst[word_off<T>(w, i0 + li, B, W)] = (T)aos[li * stride + w];
}
template <typename T>
__global__ void k_unpack_off(const T* __restrict__ src, int stride, int W, double* __restrict__ aos, int64_t B, int64_t i0, int64_t n)
{
    const int64_t li = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= n) return;
    for (int w = 0; w < W; ++w) aos[li * stride + w] = (double)src[word_off<T>(w, i0 + li, B, W)];
}
// z (7) + mask -> 8-word record; z == nullptr writes an identity pose, mask == nullptr means "all".
template <typename T>
__global__ void k_pack_z_off(const double* __restrict__ z, const uint8_t* __restrict__ mask, T* __restrict__ dst, int64_t B, int64_t i0, int64_t n)
{
    const int64_t li = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= n) return;
    for (int w = 0; w < 7; ++w) dst[word_off<T>(w, i0 + li, B, kZW)] = z ? (T)z[li * 7 + w] : (w == 6 ? T(1) : T(0));
    dst[word_off<T>(7, i0 + li, B, kZW)] = (mask == nullptr || mask[li]) ? T(1) : T(0);
}
template <typename T>
__global__ void k_unpack_z_off(const T* __restrict__ src, double* __restrict__ z, uint8_t* __restrict__ mask, int64_t B, int64_t i0, int64_t n)
{
    const int64_t li = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= n) return;
    for (int w = 0; w < 7; ++w) z[li * 7 + w] = (double)src[word_off<T>(w, i0 + li, B, kZW)];
    mask[li] = src[word_off<T>(7, i0 + li, B, kZW)] != T(0) ? 1 : 0;
}
// Full n x n row-major covariance -> packed symmetric part (P + P^T)/2.
template <typename T>
__global__ void k_pack_P_off(const double* __restrict__ Pf, int n, T* __restrict__ dst, int64_t B, int64_t i0, int64_t m)
{
    const int64_t li = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= m) return;
    const double* Pi = Pf + li * n * n;
    for (int a = 0; a < 15; ++a)
        for (int b = a; b < 15; ++b) {
            double v = (a < n && b < n) ? 0.5 * (Pi[a * n + b] + Pi[b * n + a]) : 0.0;
            dst[word_off<T>(sidx(a, b), i0 + li, B, kPW)] = (T)v;
        }
}
template <typename T>
__global__ void k_unpack_P_off(const T* __restrict__ src, int n, double* __restrict__ Pf, int64_t B, int64_t i0, int64_t m)
{
    const int64_t li = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= m) return;
    double* Pi = Pf + li * n * n;
    for (int a = 0; a < n; ++a)
        for (int b = 0; b < n; ++b) Pi[a * n + b] = (double)src[word_off<T>(sidx(a, b), i0 + li, B, kPW)];
}

// initialize_state, EKF.cpp:305-344, one filter per lane.
template <typename T>
__global__ void k_seed(DevParams<T> p, const T* __restrict__ zs, T* __restrict__ xs, T* __restrict__ Ps, T cov0, T cov1,
                       T cov2, T cov3, T cov4, int reinit_bias, int64_t B)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    T zr[kZW], x[kXW], P[kPW];
    load_rec<T, kZW>(zs, B, i, zr);
    load_rec<T, kXW>(xs, B, i, x);
    T qct[4] = {zr[3], zr[4], zr[5], zr[6]}, t[4], qn[4], C[9], pv[3];
    quat_mul(p.q_vc, qct, t);                       // EKF.cpp:310
    qn[0] = -t[0]; qn[1] = -t[1]; qn[2] = -t[2]; qn[3] = t[3];
    quat_norm(qn);                                  // EKF.cpp:311
    quat_to_rot(qn, C);
#pragma unroll
    for (int k = 0; k < 3; ++k) pv[k] = (p.C_vc[3 * k] * zr[0] + p.C_vc[3 * k + 1] * zr[1] + p.C_vc[3 * k + 2] * zr[2]) + p.r_v_cv[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        x[k] = -(C[3 * k] * pv[0] + C[3 * k + 1] * pv[1] + C[3 * k + 2] * pv[2]);  // EKF.cpp:313
        x[3 + k] = T(0);                                                        // EKF.cpp:315
        if (reinit_bias) { x[10 + k] = T(0); x[13 + k] = T(0); }                // EKF.cpp:317-321
        x[10 + k] *= p.bias_on; x[13 + k] *= p.bias_on;
    }
    x[6] = qn[0]; x[7] = qn[1]; x[8] = qn[2]; x[9] = qn[3];
#pragma unroll
    for (int k = 0; k < kPW; ++k) P[k] = T(0);
#pragma unroll
    for (int k = 0; k < 3; ++k) {                                               // EKF.cpp:323
        P[sidx(k, k)] = cov0; P[sidx(3 + k, 3 + k)] = cov1; P[sidx(6 + k, 6 + k)] = cov2;
        P[sidx(9 + k, 9 + k)] = cov3; P[sidx(12 + k, 12 + k)] = cov4;
    }
    store_rec<T, kXW>(xs, B, i, x);
    store_rec<T, kPW>(Ps, B, i, P);
}

// What the node publishes after a tick (NODE.cpp:192-220), AoS fp64.
template <typename T>
__global__ void k_report_off(DevParams<T> p, const T* __restrict__ xs, const T* __restrict__ Ps, const T* __restrict__ pfp,
                             double* __restrict__ pose, double* __restrict__ pose_cov, double* __restrict__ vel,
                             double* __restrict__ bias, int64_t B, int64_t i0, int64_t n)
{
    const int64_t li = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= n) return;
    const int64_t i = i0 + li;
    auto X = [&](int w) { return (double)xs[word_off<T>(w, i, B, kXW)]; };
    for (int k = 0; k < 3; ++k) pose[li * 7 + k] = X(k);
    for (int k = 0; k < 4; ++k) pose[li * 7 + 3 + k] = X(6 + k);
    {  // rows/cols {0-2, 6-8}, row-major (NODE.cpp:203-210)
        const int sel[6] = {0, 1, 2, 6, 7, 8};
        for (int a = 0; a < 6; ++a)
            for (int b = 0; b < 6; ++b) pose_cov[li * 36 + a * 6 + b] = (double)Ps[word_off<T>(sidx(sel[a], sel[b]), i, B, kPW)];
    }
    for (int k = 0; k < 3; ++k) vel[li * 3 + k] = X(3 + k);
    for (int k = 0; k < 3; ++k) {  // ab_nom + ab_static, wb_nom + wb_static (NODE.cpp:215-220)
        double as = pfp ? (double)pfp[word_off<T>(12 + k, i, B, kFW)] : (double)p.ab_static[k];
        double ws = pfp ? (double)pfp[word_off<T>(15 + k, i, B, kFW)] : (double)p.wb_static[k];
        bias[li * 6 + k] = X(10 + k) + as;
        bias[li * 6 + 3 + k] = X(13 + k) + ws;
    }
}

template <typename T>
__global__ void k_count_nonfinite(const T* __restrict__ xs, const T* __restrict__ Ps, unsigned long long* __restrict__ out, int64_t B)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    bool bad = false;
    for (int w = 0; w < kXW; ++w) bad |= !isfinite((double)xs[word_off<T>(w, i, B, kXW)]);
    for (int w = 0; w < kPW; ++w) bad |= !isfinite((double)Ps[word_off<T>(w, i, B, kPW)]);
    if (bad) atomicAdd(out, 1ULL);
}

}  // namespace qle
