// ekf_kernels.hpp -- HIP kernels of the batched EKF engine (gfx950).
//
// Data layout in HBM: "wave tiles".  The batch is cut into tiles of 64
// consecutive filters (one wavefront).  A per-filter record of WT words of
// type T is stored tile by tile; inside a tile it is stored as rows of 16-byte
// quads, row k holding words [k*VW, (k+1)*VW) of the tile's 64 filters
// (VW = 4 for fp32, 2 for fp64):
//     off(word w, filter i) = (i/64)*WT*64 + ((w/VW)*64 + i%64)*VW + w%VW
// Lane l of a wave reads one aligned 16-byte quad per row (global_load_dwordx4),
// a row is 1 KiB contiguous, and the whole record of a wave is one contiguous
// block (state: 144 words -> 36 KiB fp32 / 72 KiB fp64 per tile), so a wave
// touches a handful of DRAM pages / TLB entries instead of one per row.
// A record whose length is not a multiple of VW (fp32 u: 6 words) ends in one
// row of 8-byte halves.
//
// The filter state is ONE record of 144 words: x (16), the packed upper triangle of P (120), and
// the IMU sample that produced it (6 + 2 pad; only written by the multirate EKF, whose history
// entries are exactly these records).  One lane owns one filter; x and P live in VGPRs for the
// whole tick and the state is updated in place.  The multirate filter keeps its history next to it (IMU ring, checkpoints,
// anchors: see k_step_mr); the last 8 words of the record are padding.
#pragma once

#include "ekf_device.hpp"
#include "ekf_quad.hpp"
#include "ekf_fused.hpp"
#include "ekf_packed.hpp"
#include "ekf_split.hpp"

namespace qle {

// Cache policy of the stores into the IMU ring of the multirate history (0 cached, 2 non-temporal).  Measured on cfg3mr
// (profiles/r03_tuning.md): cached ring stores make the correcting tick's sample loads cheaper and every predict tick dearer
// (10.1 -> 10.7 us); the whole schedule moves by +0.6 %, inside the box-to-box spread: the ring stays streamed.
#ifndef QLE_RING_POLICY
#define QLE_RING_POLICY 2
#endif
// How many replayed ticks ahead k_step_mr requests its stored IMU samples (1..4).  Measured at 1 / 2 / 3 / 4 (profiles/r04_tuning.md section 8):
// k_step_mr<float> 40.2 / 41.6 / 40.5 / 41.6 us, <double> 89.6 / 90.9 / 91.4 / 91.8 -- one is enough, the anchor's stores do not hold the
// replay up.
#ifndef QLE_MR_PREFETCH_F32
#define QLE_MR_PREFETCH_F32 1
#endif
#ifndef QLE_MR_PREFETCH_F64
#define QLE_MR_PREFETCH_F64 1
#endif
constexpr int kBlock = 256;
constexpr int kTile = 64;   // filters per tile = wavefront size

// Minimum waves per SIMD the predict kernel is compiled for (register budget 512/waves).
// Measured on MI355X (profiles/r01_sweep.md): forcing two waves for fp32 costs 16 spilled VGPRs
// (68 B/lane of scratch traffic) and is slower at every batch size, so the default is one.
#ifndef QLE_PREDICT_WAVES_F32
#define QLE_PREDICT_WAVES_F32 1
#endif
template <typename T> struct PredictWaves { static constexpr int value = sizeof(T) == 4 ? QLE_PREDICT_WAVES_F32 : 1; };
constexpr int kXW = 16;     // state words
constexpr int kPW = 120;    // packed covariance words
constexpr int kSW = kXW + kPW + 8;  // state record: x, P, 8 words of padding (36 / 72 KiB per tile)
constexpr int kUW = 6;      // IMU words
constexpr int kZW = 8;      // tag pose 7 words + mask word
constexpr int kFW = 24;     // per-filter parameter words
constexpr int kHW = 8;      // IMU sample kept in the multirate history: 6 words + 2 pad

// 16-byte quads as native vectors (global_load/store_dwordx4).  NT selects the cache policy of the hot kernels'
// state accesses: 0 = cached loads and stores (the state lives in the 256 MiB Infinity Cache from tick to tick),
// 1 = non-temporal loads, cached stores, 2 = non-temporal loads and stores.  Every state byte is read once and
// written once per launch; which policy sustains the highest rate depends on the state size (chosen per handle,
// see ekf_capi.hip).  The input records are always read non-temporally.
typedef float qle_f4 __attribute__((ext_vector_type(4)));
typedef double qle_d2 __attribute__((ext_vector_type(2)));
typedef float qle_f2 __attribute__((ext_vector_type(2)));
template <typename T> struct Quad;
template <> struct Quad<float> { using type = qle_f4; static constexpr int VW = 4; };
template <> struct Quad<double> { using type = qle_d2; static constexpr int VW = 2; };

// Which accesses of a hot kernel are non-temporal under policy NT (profiles/r01_tuning.md section 5, sustained rates):
//   IMU / tag records (read once, never again): always non-temporal, so the input stream does not displace the state
//   in the Infinity Cache; state and per-filter parameter records: loads non-temporal for NT >= 1, stores for NT >= 2.
template <int NT, int WT> struct NtLd { static constexpr int value = (WT == kUW || WT == kZW || NT >= 1) ? 2 : 0; };
template <int NT> struct NtSt { static constexpr int value = NT >= 2 ? 2 : 0; };

template <int NT, typename Q>
__device__ __forceinline__ Q ld_quad(const Q* ptr)
{
    if (NT >= 1) return __builtin_nontemporal_load(ptr);
    return *ptr;
}
template <int NT, typename Q>
__device__ __forceinline__ void st_quad(Q* ptr, Q v)
{
    if (NT >= 1) __builtin_nontemporal_store(v, ptr);
    else *ptr = v;
}
__device__ __forceinline__ void unpack_quad(const qle_f4& v, float* r) { r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w; }
__device__ __forceinline__ void unpack_quad(const qle_d2& v, double* r) { r[0] = v.x; r[1] = v.y; }
__device__ __forceinline__ qle_f4 pack_quad(const float* r) { qle_f4 v = {r[0], r[1], r[2], r[3]}; return v; }
__device__ __forceinline__ qle_d2 pack_quad(const double* r) { qle_d2 v = {r[0], r[1]}; return v; }

// Number of filters a record array must be allocated for (whole tiles).
__host__ __device__ inline int64_t padded_filters(int64_t B) { return (B + kTile - 1) / kTile * kTile; }

// Offset (in words) of word w of filter i in an array of WT-word records.
template <typename T>
__host__ __device__ inline int64_t word_off(int w, int64_t i, int WT)
{
    constexpr int VW = 16 / (int)sizeof(T);
    const int64_t tile = i / kTile;
    const int lane = (int)(i % kTile);
    const int nf = WT / VW;
    const int64_t base = tile * WT * kTile;
    if (w < nf * VW) return base + ((int64_t)(w / VW) * kTile + lane) * VW + (w % VW);
    const int rem = WT - nf * VW;
    return base + (int64_t)nf * VW * kTile + lane * rem + (w - nf * VW);
}

// Tile index of filter i, as a wave-uniform (SGPR) value: the 64 lanes of a wave always belong to one
// tile (blocks are multiples of 64 threads), so the tile base can live in scalar registers and the
// loads/stores use the scalar-base + per-lane-offset addressing form instead of 64-bit VALU adds.
// Block index -> position in the batch (XCD-aware).  Workgroups are dispatched round-robin over the 8 XCDs
// (block b runs on XCD b % 8), so with the identity map every XCD touches every 8th 4-tile group of the state.
// Giving each XCD one contiguous eighth of the batch instead measured +3 % on k_predict at 65 536 filters,
// +3 % at 262 144, +1-2 % at 1 M, -1 % at 131 072 (profiles/r01_tuning.md section 4).  The map is a bijection
// on [0, gridDim.x): the first 8*floor(n/8) blocks are transposed, the ragged rest keeps its index.
// -DQLE_XCD_CHUNK=0 restores the identity map.
#ifndef QLE_XCD_CHUNK
#define QLE_XCD_CHUNK 1
#endif
// The kernel arguments the first loads depend on, all requested at the kernel's entry.  Left alone the compiler fetches an argument
// where it is first needed and waits there: grid size -> (wait) -> block size, batch size -> (wait) -> record pointers -> (wait) ->
// first load, three scalar-cache misses one after the other in front of every launch's first byte; with this they are one.
// (The per-tick kernels no longer fetch these arguments at all: they are among the 14 dwords the dispatch preloads into SGPRs, see
// k_predict; for them this only pins the order, for k_run_resident -- one launch per run -- it is the single fetch.)
#ifndef QLE_EARLY_ARGS
#define QLE_EARLY_ARGS 1
#endif
template <typename... A> __device__ __forceinline__ void args_early(A... a)
{
#if QLE_EARLY_ARGS
    (..., [](auto v) { asm volatile("" ::"s"(v)); }(a));
#endif
}
#define QLE_ARGS_EARLY(...) args_early(__VA_ARGS__)
// nothing is scheduled across this point: the loads in front of it are all issued before the arithmetic behind it starts
#ifndef QLE_LOADS_FIRST_ON
#define QLE_LOADS_FIRST_ON 1
#endif
#if QLE_LOADS_FIRST_ON
#define QLE_LOADS_FIRST() __builtin_amdgcn_sched_barrier(0)
#else
#define QLE_LOADS_FIRST() ((void)0)
#endif
__device__ __forceinline__ int64_t batch_block(unsigned grid)
{
#if QLE_XCD_CHUNK
    const unsigned b = blockIdx.x, n8 = grid & ~7u;
    return b < n8 ? (int64_t)((b & 7u) * (n8 >> 3) + (b >> 3)) : (int64_t)b;
#else
    return (int64_t)blockIdx.x;
#endif
}
__device__ __forceinline__ int64_t batch_block() { return batch_block(gridDim.x); }

__device__ __forceinline__ int64_t wave_tile(int64_t i) { return (int64_t)__builtin_amdgcn_readfirstlane((int)(i >> 6)); }

// Load words [W0, W0+W) of filter i's WT-word record.  W0 and W are whole quads,
// except that the load may end with the record's 8-byte tail row (fp32 only).
template <typename T, int WT, int W0, int W, int NT = 0>
__device__ __forceinline__ void load_rec(const T* __restrict__ base, int64_t i, T (&r)[W])
{
    using Q = typename Quad<T>::type;
    constexpr int VW = Quad<T>::VW;
    constexpr int NFT = WT / VW;       // full quad rows in the record
    constexpr int NF = W / VW;         // full quad rows in this load
    constexpr int REM = W % VW;
    static_assert(W0 % VW == 0, "loads start on a quad row");
    static_assert(REM == 0 || (REM == 2 && W0 + W == WT && W0 / VW + NF == NFT), "only the record's own 8-byte tail may be partial");
    const int64_t tile = wave_tile(i);
    const int lane = (int)(i & 63);
    const T* tb = base + tile * (int64_t)(WT * kTile);
#pragma unroll
    for (int k = 0; k < NF; ++k) {
        Q v = ld_quad<NtLd<NT, WT>::value>(reinterpret_cast<const Q*>(tb + ((W0 / VW + k) * kTile + lane) * VW));
        unpack_quad(v, &r[k * VW]);
    }
    if (REM == 2) {
        qle_f2 v = ld_quad<NtLd<NT, WT>::value>(reinterpret_cast<const qle_f2*>(tb + NFT * VW * kTile + lane * 2));
        r[NF * VW] = v.x;
        r[NF * VW + 1] = v.y;
    }
}

template <typename T, int WT, int W0, int W, int NT = 0>
__device__ __forceinline__ void store_rec(T* __restrict__ base, int64_t i, const T (&r)[W])
{
    using Q = typename Quad<T>::type;
    constexpr int VW = Quad<T>::VW;
    constexpr int NF = W / VW;
    static_assert(W % VW == 0 && W0 % VW == 0, "stored ranges are whole quads");
    const int64_t tile = wave_tile(i);
    const int lane = (int)(i & 63);
    T* tb = base + tile * (int64_t)(WT * kTile);
#pragma unroll
    for (int k = 0; k < NF; ++k) st_quad<NtSt<NT>::value>(reinterpret_cast<Q*>(tb + ((W0 / VW + k) * kTile + lane) * VW), pack_quad(&r[k * VW]));
}

// Compact records: est_bias = false (EKF.cpp:92, num_states = 9) without the multirate history.  The bias blocks of such a filter's P are
// identically zero (no process noise, no coupling: EKF.cpp:405-409), so its record keeps only the 45 words of the 9 x 9 pose block, as
// their own row-major triangle in record words 16..60 (3 words of padding): a tick moves 16 + 48 words per direction instead of 136.
// The arithmetic runs on the same 15-state register image (zeros in the bias blocks), which is what the full-record path computes too.
constexpr int kPWc = 48;
__host__ __device__ constexpr int sidx9(int i, int j) { return i * 9 - i * (i - 1) / 2 + (j - i); }   // i <= j < 9
// record word of P(a, b), a <= b, or -1 when a compact record does not hold it
__host__ __device__ constexpr int p_word(int a, int b, bool compact)
{
    return compact ? (b < 9 ? kXW + sidx9(a, b) : -1) : kXW + sidx(a, b);
}
template <typename T, int NT = 0>
__device__ __forceinline__ void load_P_compact(const T* __restrict__ st, int64_t i, T (&P)[kPW])
{
    T t[kPWc];
    load_rec<T, kSW, kXW, kPWc, NT>(st, i, t);
#pragma unroll
    for (int a = 0; a < 15; ++a)
#pragma unroll
        for (int b = a; b < 15; ++b) P[sidx(a, b)] = b < 9 ? t[sidx9(a, b)] : T(0);
}
template <typename T, int NT = 0>
__device__ __forceinline__ void store_P_compact(T* __restrict__ st, int64_t i, const T (&P)[kPW])
{
    T t[kPWc];
#pragma unroll
    for (int k = 45; k < kPWc; ++k) t[k] = T(0);
#pragma unroll
    for (int a = 0; a < 9; ++a)
#pragma unroll
        for (int b = a; b < 9; ++b) t[sidx9(a, b)] = P[sidx(a, b)];
    store_rec<T, kSW, kXW, kPWc, NT>(st, i, t);
}
// the P part of a state record, either layout (wave-uniform choice)
template <typename T, int NT = 0>
__device__ __forceinline__ void load_P_any(const T* __restrict__ st, int64_t i, T (&P)[kPW], bool compact)
{
    if (compact) load_P_compact<T, NT>(st, i, P);
    else load_rec<T, kSW, kXW, kPW, NT>(st, i, P);
}
template <typename T, int NT = 0>
__device__ __forceinline__ void store_P_any(T* __restrict__ st, int64_t i, const T (&P)[kPW], bool compact)
{
    if (compact) store_P_compact<T, NT>(st, i, P);
    else store_rec<T, kSW, kXW, kPW, NT>(st, i, P);
}

template <typename T, bool PFP>
__device__ __forceinline__ void load_noise(const DevParams<T>& p, const T* __restrict__ pfp, int64_t i, Noise<T>& nz)
{
    if (PFP) {
        T f[kFW];
        load_rec<T, kFW, 0, kFW>(pfp, i, f);
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

// Load / store a range of whole quad rows [Q0, Q1) of the packed P (record words kXW + 4q ..),
// rows taken in DESCENDING order so that the bias rows (end of the row-major triangle) come first.
template <typename T, int Q0, int Q1, int NT = 0>
__device__ __forceinline__ void load_P_quads_desc(const T* __restrict__ st, int64_t i, T (&P)[kPW])
{
    using Q = typename Quad<T>::type;
    constexpr int VW = Quad<T>::VW;
    const int64_t tile = wave_tile(i);
    const int lane = (int)(i & 63);
    const T* tb = st + tile * (int64_t)(kSW * kTile);
#pragma unroll
    for (int k = Q1 - 1; k >= Q0; --k) {
        Q v = ld_quad<NtLd<NT, kSW>::value>(reinterpret_cast<const Q*>(tb + ((kXW / VW + k) * kTile + lane) * VW));
        unpack_quad(v, &P[k * VW]);
    }
}
template <typename T, int Q0, int Q1, int NT = 0>
__device__ __forceinline__ void store_P_quads_desc(T* __restrict__ st, int64_t i, const T (&P)[kPW])
{
    using Q = typename Quad<T>::type;
    constexpr int VW = Quad<T>::VW;
    const int64_t tile = wave_tile(i);
    const int lane = (int)(i & 63);
    T* tb = st + tile * (int64_t)(kSW * kTile);
#pragma unroll
    for (int k = Q1 - 1; k >= Q0; --k) st_quad<NtSt<NT>::value>(reinterpret_cast<Q*>(tb + ((kXW / VW + k) * kTile + lane) * VW), pack_quad(&P[k * VW]));
}

// A filter is "not initialised" (state_initialized == false, EKF.cpp:73,129-130) while its stored quaternion is all
// zero -- the state memory starts zeroed, initialize_state / set_state write a unit quaternion -- and every tick kernel
// leaves such a filter untouched: no predict, no counters, no history entry, exactly the reference's early return.
// The flag lives in the record the tick reads anyway, so it costs no traffic.
template <typename T>
__device__ __forceinline__ bool filter_uninitialised(const T (&x)[kXW])
{
    return x[6] == T(0) && x[7] == T(0) && x[8] == T(0) && x[9] == T(0);
}

// --------------------------------------------------------- measurement gate
// Decision logic of filter_update, EKF.cpp:147-186, per filter on the device:
//   consume  = measurement_ready && (!limit_measurement_freq || upds_since_correction + 1 >= upd_per_meas)
//   perform  = consume && (!corner_margin_enbl || some tag of the bundle projects inside the image margins)
// upds_since_correction is kept implicitly: last_corr[i] is the index of the filter's last correcting
// tick (-1 = never), so upds_since_correction before tick n is n - last_corr[i] - 1 and predict-only
// ticks never touch the array.  The projection runs in fp64 whatever the compute dtype, so the
// discrete decision matches the fp64 reference for the same (dtype-rounded) tag pose.
struct GateParams {
    int32_t limit;              // limit_measurement_freq (EKF.hpp:75)
    int32_t upd_per_meas;       // EKF.cpp:91
    int32_t corner_enbl;        // corner_margin_enbl (EKF.hpp:76)
    int32_t n_tags;             // EKF.hpp:117
    int32_t tick;               // index of this tick
    double K[9];                // camera_K row-major
    double x_lo, x_hi, y_lo, y_hi;  // camera_width*margin, camera_width*(1-margin), same for height (EKF.cpp:175-178)
    double hw[16], px[16], py[16];  // tag_widths/2, tag_positions x,y (EKF.cpp:163-164)
};

__device__ inline bool corner_gate(const GateParams& g, const double (&z)[7])
{
    double q[4] = {z[3], z[4], z[5], z[6]}, C[9];
    quat_to_rot<double>(q, C);  // T_ct = Translation(r_c_tc) * q_ct, EKF.cpp:154
    for (int t = 0; t < g.n_tags; ++t) {
        const double hw = g.hw[t];
        const double cx[4] = {hw + g.px[t], -hw + g.px[t], -hw + g.px[t], hw + g.px[t]};
        const double cy[4] = {hw + g.py[t], hw + g.py[t], -hw + g.py[t], -hw + g.py[t]};
        double mnx = 0, mny = 0, mxx = 0, mxy = 0;
        for (int k = 0; k < 4; ++k) {
            double pc[3];
            for (int r = 0; r < 3; ++r) pc[r] = C[3 * r] * cx[k] + C[3 * r + 1] * cy[k] + C[3 * r + 2] * 0.0 + z[r];
            const double iz = 1.0 / pc[2];                                   // EKF.cpp:168
            const double nx = pc[0] * iz, ny = pc[1] * iz, nz = pc[2] * iz;  // EKF.cpp:169
            const double u = g.K[0] * nx + g.K[1] * ny + g.K[2] * nz;        // EKF.cpp:170
            const double v = g.K[3] * nx + g.K[4] * ny + g.K[5] * nz;
            if (k == 0) { mnx = mxx = u; mny = mxy = v; }
            else { mnx = fmin(mnx, u); mxx = fmax(mxx, u); mny = fmin(mny, v); mxy = fmax(mxy, v); }
        }
        if (mnx > g.x_lo && mny > g.y_lo && mxx < g.x_hi && mxy < g.y_hi) return true;  // EKF.cpp:175-180
    }
    return false;
}

// ------------------------------------------------------------- hot kernels
// Predict tick: reads x16 + P120 + u6, writes x16 + P120 (278 words/filter).
// Packed P (sidx in ekf_device.hpp): the words of block-row r come first, then v, th, ab, wb.
// Loads are issued bottom-up and each block-row of the new P is stored as soon as it is final
// (ekf_predict_levels), so the stores overlap the loads of the rows above inside the one wave a
// SIMD holds at B = 65 536.  QLE_PREDICT_LEVELS=0 selects the in-place variant (ekf_predict).
#ifndef QLE_PREDICT_LEVELS
#define QLE_PREDICT_LEVELS 1
#endif
// `src` is the state at tick n-1, `dst` the state at tick n: the same array (in place: every load of a lane is issued
// before its first store).  MR (multirate filter): the tick also appends to the history -- the IMU sample goes to its slot
// of the IMU ring (hist_u, EKF.cpp:254-256) and on checkpoint ticks the new state is copied to its checkpoint slot
// (hist_ck != nullptr, wave-uniform); see k_step_mr for the history scheme.
template <typename T, bool PFP, int NT, bool MR, bool COMPACT = false, bool LF = false>
__device__ __forceinline__ void predict_tick(const DevParams<T>& p, const T* src, T* dst, const T* __restrict__ us,
                                             const T* __restrict__ pfp, T* __restrict__ aux_accel, T* __restrict__ hist_u,
                                             T* __restrict__ hist_ck, bool ck_cached, int64_t i)
{
    T x[kXW], P[kPW], u[kUW], accel[3];
    load_rec<T, kUW, 0, kUW, NT>(us, i, u);
    load_rec<T, kSW, 0, kXW, NT>(src, i, x);
    Noise<T> nz;
    load_noise<T, PFP>(p, pfp, i, nz);
#if QLE_PREDICT_LEVELS
    constexpr int VW = Quad<T>::VW;
    constexpr int NQ = kPW / VW;
    if constexpr (COMPACT) load_P_compact<T, NT>(src, i, P);
    else load_P_quads_desc<T, 0, NQ, NT>(src, i, P);
    // Every load of the tick is requested before the first instruction that needs one of them: left alone the scheduler issued ten of
    // the 36 record loads, waited for x and u, ran the first dozen instructions of the nominal predict (they feed the wave-uniform
    // branch of the half-angle series that ends the block) and only then requested the other 26 quads of the covariance -- one memory
    // round trip later.  LF is chosen per launch (tu_predict.hip): +1 % where every SIMD holds ONE wave (65 536 filters: 9.39 -> 9.30 us),
    // nothing below, and -3 % with two waves per SIMD (131 072: 20.8 -> 21.4 us), where the staggered requests are the gentler pattern
    // for the caches (profiles/r04_tuning.md section 10).
    if constexpr (LF) QLE_LOADS_FIRST();
    // A filter that is not initialised is left untouched.  No early exit: the compiler would sink the covariance loads below such a
    // branch and every wave would wait for x before it even issues them (+1 us per tick at 65 536 filters, profiles/r02_tuning.md).
    // Instead the lane computes on (with a unit quaternion, so that its arithmetic stays finite) and only its stores are masked.
    const bool dead = filter_uninitialised(x);
    if (dead) x[9] = T(1);
    T Pn[kPW];
    // a quad is final once every word in it is: first quad that holds only block-rows >= ab / th / v (ekf_device.hpp)
    constexpr int q_ab = level_first_word(3, VW) / VW, q_th = level_first_word(2, VW) / VW, q_v = level_first_word(1, VW) / VW;
    ekf_predict_levels<T>(p, nz, x, P, u, accel, Pn, [&](int level) {
        if (dead) return;
        if (level == -1) store_rec<T, kSW, 0, kXW, NT>(dst, i, x);
        else if (COMPACT) { if (level == 3) store_P_compact<T, NT>(dst, i, Pn); }
        else if (level == 0) store_P_quads_desc<T, q_ab, NQ, NT>(dst, i, Pn);
        else if (level == 1) store_P_quads_desc<T, q_th, q_ab, NT>(dst, i, Pn);
        else if (level == 2) store_P_quads_desc<T, q_v, q_th, NT>(dst, i, Pn);
        else store_P_quads_desc<T, 0, q_v, NT>(dst, i, Pn);
        if (MR && hist_ck) {
            // checkpoint copy of the same words: a grid checkpoint is streamed past the caches (it is rarely read again); the extra
            // checkpoint at the expected entry of the next tag pose is read back a dozen ticks later and is written CACHED, so that it
            // waits in the Infinity Cache next to the state (wave-uniform choice)
            if (ck_cached) {
                if (level == -1) store_rec<T, kSW, 0, kXW, 0>(hist_ck, i, x);
                else if (level == 0) store_P_quads_desc<T, q_ab, NQ, 0>(hist_ck, i, Pn);
                else if (level == 1) store_P_quads_desc<T, q_th, q_ab, 0>(hist_ck, i, Pn);
                else if (level == 2) store_P_quads_desc<T, q_v, q_th, 0>(hist_ck, i, Pn);
                else store_P_quads_desc<T, 0, q_v, 0>(hist_ck, i, Pn);
            } else {
                if (level == -1) store_rec<T, kSW, 0, kXW, 2>(hist_ck, i, x);
                else if (level == 0) store_P_quads_desc<T, q_ab, NQ, 2>(hist_ck, i, Pn);
                else if (level == 1) store_P_quads_desc<T, q_th, q_ab, 2>(hist_ck, i, Pn);
                else if (level == 2) store_P_quads_desc<T, q_v, q_th, 2>(hist_ck, i, Pn);
                else store_P_quads_desc<T, 0, q_v, 2>(hist_ck, i, Pn);
            }
        }
    });
#else
    load_P_any<T, NT>(src, i, P, COMPACT);
    const bool dead = filter_uninitialised(x);
    if (dead) x[9] = T(1);
    ekf_predict<T>(p, nz, x, P, u, accel);
    if (!dead) {
        store_rec<T, kSW, 0, kXW, NT>(dst, i, x);
        store_P_any<T, NT>(dst, i, P, COMPACT);
        if (MR && hist_ck) {
            store_rec<T, kSW, 0, kXW, 2>(hist_ck, i, x);
            store_rec<T, kSW, kXW, kPW, 2>(hist_ck, i, P);
        }
    }
#endif
    if (dead) return;
    if (MR) {
        const T uk[kHW] = {u[0], u[1], u[2], u[3], u[4], u[5], T(0), T(0)};
        store_rec<T, kHW, 0, kHW, QLE_RING_POLICY>(hist_u, i, uk);
    }
    if (aux_accel) {  // optional side output (wave-uniform), AoS [B][3] in the compute dtype
#pragma unroll
        for (int k = 0; k < 3; ++k) aux_accel[i * 3 + k] = accel[k];
    }
}

// NT == 3 ("split", states larger than the Infinity Cache): the workgroups selected by `split` keep their tiles
// cached (policy 0), all others stream (policy 2), so a fixed part of the state that fits the cache stays resident
// from tick to tick.  split >= 0: the first `split` dispatched workgroups (spread over all XCDs by batch_block());
// split < 0: interleaved, workgroups with ((blockIdx.x >> 3) & 63) < -split, i.e. -split/64 of every XCD's share.
__device__ __forceinline__ bool cached_workgroup(int32_t split)
{
    return split >= 0 ? blockIdx.x < (unsigned)split : ((blockIdx.x >> 3) & 63u) < (unsigned)(-split);
}

template <typename T, bool PFP, int NT, bool MR, bool COMPACT = false, bool LF = false>
__global__ __launch_bounds__(kBlock, PredictWaves<T>::value) void k_predict(const T* src, T* dst, const T* __restrict__ us, int64_t B, int64_t i0,
                                                       int32_t grid_x, int32_t block_x, int32_t split, int32_t ck_cached,
                                                       const T* __restrict__ pfp, T* __restrict__ aux_accel, T* __restrict__ hist_u,
                                                       T* __restrict__ hist_ck, DevParams<T> p)
{
    // Argument order: what the first loads depend on comes first, 14 dwords of it, so that the wave finds them in its SGPRs when it starts
    // (the translation units are built with -amdgpu-kernarg-preload-count; grid and block size are passed explicitly because the
    // hidden arguments are not among the preloaded ones) instead of fetching them from the kernel-argument segment -- a memory round
    // trip in front of every launch's first load.  The parameter block, needed when the first data arrive, comes last.
    QLE_ARGS_EARLY(src, dst, us, B, i0, grid_x, block_x);
    const int64_t i = i0 + batch_block((unsigned)grid_x) * block_x + threadIdx.x;   // i0: first filter of this launch (a tick may be launched in chunks)
    if (i >= B) return;
    if (NT == 3) {
        if (cached_workgroup(split)) predict_tick<T, PFP, 0, MR, COMPACT, LF>(p, src, dst, us, pfp, aux_accel, hist_u, hist_ck, ck_cached != 0, i);
        else predict_tick<T, PFP, 2, MR, COMPACT, LF>(p, src, dst, us, pfp, aux_accel, hist_u, hist_ck, ck_cached != 0, i);
    } else {
        predict_tick<T, PFP, NT, MR, COMPACT, LF>(p, src, dst, us, pfp, aux_accel, hist_u, hist_ck, ck_cached != 0, i);
    }
}

// The nominal state of a lane in / out of its column of an LDS array (step_tick: fp64 keeps x there during the covariance sweep).
template <typename T, bool ON, bool LOAD>
struct LdsPark {
    static constexpr bool parked = ON;
    T (*slot)[ON ? kBlock : 1];
    __device__ __forceinline__ void operator()(T (&xx)[kXW]) const
    {
        if constexpr (ON) {
#pragma unroll
            for (int k = 0; k < kXW; ++k) {
                if (LOAD) xx[k] = slot[k][threadIdx.x];
                else slot[k][threadIdx.x] = xx[k];
            }
        }
    }
};

// Fused tick (filter_update single-rate branch, EKF.cpp:238-249,265-290): predict, then correct where the record's mask word is
// non-zero, as one straight-line schedule (ekf_step_fused, ekf_fused.hpp).
// Reads x16 + P120 + u6 + z7 (+mask), writes x16 + P120 (285 words/filter).
template <typename T, bool DIRECT, bool PFP, bool GATE, int NT, bool COMPACT = false>
__device__ __forceinline__ void step_tick(const DevParams<T>& p, const GateParams& gp, T* st, const T* __restrict__ us,
                                          const T* __restrict__ zs, const T* __restrict__ pfp,
                                          T* __restrict__ aux_accel, T* __restrict__ aux_obs,
                                          int32_t* __restrict__ last_corr, uint8_t* __restrict__ flags, int64_t i)
{
    using Q = typename Quad<T>::type;
    constexpr int VW = Quad<T>::VW;
    T x[kXW], Po[kPW], u[kUW], zr[kZW];
    load_rec<T, kUW, 0, kUW, NT>(us, i, u);
    load_rec<T, kZW, 0, kZW, NT>(zs, i, zr);
    load_rec<T, kSW, 0, kXW, NT>(st, i, x);
    load_P_any<T, NT>(st, i, Po, COMPACT);   // ascending: the fused schedule starts with rows r
    const bool dead = filter_uninitialised(x);   // left untouched; no early exit (see predict_tick)
    if (dead) x[9] = T(1);
    bool corr = !dead && zr[7] != T(0);
    if (GATE) {  // the mask word means "measurement_ready"; decide here (EKF.cpp:147-186)
        const bool consume = corr && (!gp.limit || (gp.tick - last_corr[i]) >= gp.upd_per_meas);
        bool ok = consume;
        if (consume && gp.corner_enbl) {
            const double zd[7] = {(double)zr[0], (double)zr[1], (double)zr[2], (double)zr[3], (double)zr[4], (double)zr[5], (double)zr[6]};
            ok = corner_gate(gp, zd);
        }
        corr = ok;
        if (ok) last_corr[i] = gp.tick;
        flags[i] = (uint8_t)((ok ? 1 : 0) | (consume ? 2 : 0));
    }
    Noise<T> nz;
    load_noise<T, PFP>(p, pfp, i, nz);
    const T z[7] = {zr[0], zr[1], zr[2], zr[3], zr[4], zr[5], zr[6]};
    T* tb = st + wave_tile(i) * (int64_t)(kSW * kTile);
    const int lane = (int)(i & 63);
    T Pc[kPWc];   // compact records only: written group by group in the final sweep (every pose-block word is in one), never otherwise
    auto tag_pose = [&](T (&zz)[7]) {
        if constexpr (sizeof(T) == 8) {   // fp64: read again where the innovation needs it (an L2 hit) instead of 14 registers held through the sweep
            T zq[kZW];
            load_rec<T, kZW, 0, kZW, NT>(zs, i, zq);
#pragma unroll
            for (int k = 0; k < 7; ++k) zz[k] = zq[k];
        } else {
#pragma unroll
            for (int k = 0; k < 7; ++k) zz[k] = z[k];
        }
    };
    // fp64, conventional orientation method: the nominal state (16 values) waits out the covariance sweep in the LDS (32 KiB per workgroup)
    // and r and q are read back from there for each block-row's Gx -- with that the kernel needs no scratch (was 140-250 B per lane)
    constexpr bool kPark = sizeof(T) == 8 && !DIRECT;   // the direct method fits without (256 + 215 registers) and is 1.7 us faster so (28.5 vs 26.8 us)
    __shared__ T parked[kPark ? kXW : 1][kPark ? kBlock : 1];
    const LdsPark<T, kPark, false> park{parked};
    const LdsPark<T, kPark, true> unpark{parked};
    ekf_step_fused_z<T, DIRECT>(p, nz, x, Po, u, tag_pose, corr, !dead,
        [&](const T (&accel)[3]) {
            if (aux_accel && !dead) {   // optional side outputs (wave-uniform), written as soon as they exist
#pragma unroll
                for (int k = 0; k < 3; ++k) aux_accel[i * 3 + k] = accel[k];
            }
        },
        [&](const T (&obs)[7]) {
            if (aux_accel) {
#pragma unroll
                for (int k = 0; k < 7; ++k) aux_obs[i * 7 + k] = obs[k];
            }
        },
        [&]() { store_rec<T, kSW, 0, kXW, NT>(st, i, x); },
        [&](auto qc, const T* w4) {   // q4 = index of a 4-word group of P; one 16-byte quad in fp32, two in fp64
            constexpr int q4 = decltype(qc)::value;
            if constexpr (COMPACT) {   // compact records: the words of the pose block are collected and stored once, below
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (word_col(4 * q4 + k) < 9) Pc[sidx9(word_row(4 * q4 + k), word_col(4 * q4 + k))] = w4[k];
                return;
            }
#pragma unroll
            for (int h = 4 / VW - 1; h >= 0; --h) {
                const int qr = kXW / VW + q4 * (4 / VW) + h;
                st_quad<NtSt<NT>::value>(reinterpret_cast<Q*>(tb + (qr * kTile + lane) * VW), pack_quad(w4 + h * VW));
            }
        }, park, unpark);
    if (COMPACT && !dead) {
#pragma unroll
        for (int k = 45; k < kPWc; ++k) Pc[k] = T(0);
        store_rec<T, kSW, kXW, kPWc, NT>(st, i, Pc);
    }
}

template <typename T, bool DIRECT, bool PFP, bool GATE, int NT, bool COMPACT = false>
__global__ __launch_bounds__(kBlock, sizeof(T) == 8 ? 1 : 2) void k_step(T* st, const T* __restrict__ us, const T* __restrict__ zs, int64_t B, int64_t i0,
                                                 int32_t grid_x, int32_t block_x, int32_t split,   // (argument order: see k_predict)
                                                 const T* __restrict__ pfp, T* __restrict__ aux_accel, T* __restrict__ aux_obs,
                                                 int32_t* __restrict__ last_corr, uint8_t* __restrict__ flags, DevParams<T> p, GateParams gp)
{
    QLE_ARGS_EARLY(st, us, zs, B, i0, grid_x, block_x);
    const int64_t i = i0 + batch_block((unsigned)grid_x) * block_x + threadIdx.x;
    if (i >= B) return;
    if (NT == 3) {   // see k_predict
        if (cached_workgroup(split)) step_tick<T, DIRECT, PFP, GATE, 0, COMPACT>(p, gp, st, us, zs, pfp, aux_accel, aux_obs, last_corr, flags, i);
        else step_tick<T, DIRECT, PFP, GATE, 2, COMPACT>(p, gp, st, us, zs, pfp, aux_accel, aux_obs, last_corr, flags, i);
    } else {
        step_tick<T, DIRECT, PFP, GATE, NT, COMPACT>(p, gp, st, us, zs, pfp, aux_accel, aux_obs, last_corr, flags, i);
    }
}

// ------------------------------------------------------------ multirate EKF
// filter_update with multirate_ekf = true (EKF.cpp:196-236, 251-264): a tag pose that was taken `step` ticks ago is fused
// into the state the filter held THEN, and the predictions since are replayed with the stored IMU samples.
//
// History.  The reference keeps per-filter vectors x_hist / u_hist / P_hist (EKF.hpp:62-64) with one entry per tick; only
// the entry `step` ticks back (at most step_max) and its successors are ever read again, and after a correction the history
// starts at the corrected entry (the trim of EKF.cpp:214-219).  "State after tick t" is a pure function of an earlier state of
// the same chain and the IMU samples in between, so the engine stores
//   cur      the state after the newest tick, in place (one record array, cache-resident exactly like the single-rate filter);
//   u ring   the IMU sample of every tick, slot t % Cu (8 words per filter and tick);
//   ckpt     a copy of the state after every k-th tick, slot (t/k) % Nc (Cu = k Nc >= step_max + k + 1);
//   anchor   per filter the corrected entry of its last correction (tick hist_first[i]) -- the start of its history;
//   extra    one more checkpoint slot (index Nc) that the host places where it expects the NEXT measurement's entry: tag poses come at
//            a regular cadence with a near-constant latency, so after a correcting tick n the next entry will be about
//            n + (ticks between the last two correcting ticks) - (nominal step delay); the predict launch of that tick copies the state
//            there as well.  A filter whose entry is at or just after it starts from it and replays nothing (or a tick or two) instead of
//            (k-1)/2 ticks from the grid; a filter it does not fit (another phase, an early pose) never looks at it;
// and rebuilds the entry a measurement belongs to by replaying at most k-1 predictions from the newest checkpoint in
// (hist_first, mt], or from the anchor.  The replay towards "now" rewrites the checkpoints it passes, so every checkpoint
// newer than hist_first always holds the current chain.  Same arithmetic on the same stored samples as the reference's
// rewritten history entries, hence the same values; a predict-only tick costs 8 + 136/k extra words instead of a second copy
// of the state, and the history of 65 536 fp32 filters at 400 Hz with a 200 ms window is 0.75 GB instead of 6 GB.
struct MrParams {
    int32_t k;            // checkpoint period in ticks
    int32_t Nc;           // checkpoint slots
    int32_t Cu;           // IMU ring slots = k * Nc
    int32_t tick;         // index n of this tick; the newest history entry is tick n-1 (= cur)
    int32_t fixed_step;   // measurement_step_delay (EKF.cpp:93) when !dynamic
    int32_t dynamic;      // dynamic_meas_delay (EKF.hpp:79)
    int32_t gate;         // 1: mask word = measurement_ready, decide on device; 0: mask word = perform
    int32_t e_tick;       // tick whose state the EXTRA checkpoint slot (index Nc) holds; far negative = none (see k_step_mr)
    int64_t slot_words;   // words per state slot (kSW x padded batch)
    int64_t u_words;      // words per IMU ring slot (kHW x padded batch)
    double dT, offset, delay_max, t_curr, uniform_age;  // EKF.cpp:199-200
};

__host__ __device__ inline int32_t floor_div(int32_t a, int32_t b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }
template <typename T>
__device__ __forceinline__ T* mr_u_slot(T* uring, const MrParams& m, int32_t tick)
{
    int32_t s = tick % m.Cu;
    if (s < 0) s += m.Cu;
    return uring + (int64_t)s * m.u_words;
}
template <typename T>
__device__ __forceinline__ T* mr_ck_slot(T* ckpt, const MrParams& m, int32_t tick)   // tick is a multiple of k, >= 0
{
    return ckpt + (int64_t)((tick / m.k) % m.Nc) * m.slot_words;
}

// Per-wave timeline of k_step_mr (diagnostic build only: make dbg, -DQLE_MR_STAMPS; profiles/r03_scripts/mr_timeline.py reads it back
// through qle_debug_clocks).  Lane 0 of every wave writes s_memtime at the marked points; each stamp takes a value of the phase before it
// as an input so that it cannot move.  Slots: 0 entry, 1 inputs and x arrived, 2 chain start decided, 3 chain state arrived, 4 first
// IMU sample arrived, 5 / 6 correction begin / end, 7 end; 8 + 2 j / 9 + 2 j: IMU sample of loop iteration j ready / its predict done.
#ifdef QLE_MR_STAMPS
constexpr int kDbgSlots = 128, kDbgWaves = 4096;
static __device__ unsigned long long qle_dbg_clock[kDbgWaves * kDbgSlots];
#define QLE_STAMP(k, dep)                                                                                                  \
    do {                                                                                                                   \
        unsigned long long t_;                                                                                             \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : "v"(dep) : "memory");                              \
        if ((i & 63) == 0 && (i >> 6) < kDbgWaves && (k) < kDbgSlots) qle_dbg_clock[(i >> 6) * kDbgSlots + (k)] = t_;       \
    } while (0)
// the same for a kernel that names its wave and its writing lane itself (kw_tick: one workgroup per tile)
#define QLE_STAMPW(wave, writer, k, dep)                                                                                   \
    do {                                                                                                                   \
        unsigned long long t_;                                                                                             \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : "v"(dep) : "memory");                              \
        if ((writer) && (wave) < kDbgWaves && (k) < kDbgSlots) qle_dbg_clock[(wave) * kDbgSlots + (k)] = t_;                \
    } while (0)
#else
#define QLE_STAMP(k, dep) do { } while (0)
#define QLE_STAMPW(wave, writer, k, dep) do { } while (0)
#endif

__device__ __forceinline__ int32_t wave_max_i32(int32_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int32_t o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return __builtin_amdgcn_readfirstlane(v);
}
// Minimum of a value over the 64 lanes of the wave, as a wave-uniform (SGPR) value.  Every lane must be active.
__device__ __forceinline__ int32_t wave_min_i32(int32_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int32_t o = __shfl_xor(v, off, 64);
        v = o < v ? o : v;
    }
    return __builtin_amdgcn_readfirstlane(v);
}

// The stored IMU sample of tick ts (wave-uniform) into dst: from the ring, or -- the current tick's own -- from the input record; a tick
// beyond the current one has none.
template <typename T>
__device__ __forceinline__ void mr_request_sample(const MrParams& m, T* uring, const T* __restrict__ us, int64_t i, int32_t ts, T (&dst)[kUW])
{
    if (ts < m.tick) {
        T ur[kHW];
        load_rec<T, kHW, 0, kHW>(mr_u_slot(uring, m, ts), i, ur);
#pragma unroll
        for (int k = 0; k < kUW; ++k) dst[k] = ur[k];
    } else if (ts == m.tick) {
        T uc[kUW];
        load_rec<T, kUW, 0, kUW>(us, i, uc);
#pragma unroll
        for (int k = 0; k < kUW; ++k) dst[k] = uc[k];
    }
}

// A multirate tick that carries tag poses.  Lanes that correct: load the newest checkpoint at or before the entry the
// measurement belongs to (or the anchor), replay up to that entry, fuse the measurement there (the corrected entry becomes the
// anchor), replay the predictions up to n-1 from the stored IMU samples -- rewriting the checkpoints on the way -- then predict
// tick n.  Lanes that do not: plain predict of `cur`.
//
// The covariance stays in registers for the whole chain (12-35 predictions), so the kernel is bound by the number of instructions per
// replayed tick, not by bytes (profiles/r02_tuning.md section 10): the chain runs on the register-block form of ekf_packed.hpp
// (packed fp32 FMAs, no libm call per tick), and the loop counter is WAVE-UNIFORM -- the wave walks from the earliest entry any of its
// lanes starts from, a lane joins at its own entry -- so that the history addresses (IMU ring slot, checkpoint slot) are scalar and
// the "is this the current tick" selects are scalar branches.  One predict call site serves the replay and the current tick.
// The covariance of a replayed chain as it is held between the ticks of the loop.  fp32: the register blocks of ekf_packed.hpp (two FMAs per
// instruction), the correction on the unpacked triangle.  fp64: SPLIT (ekf_split.hpp) -- the packed triangle alone is 240 of a wave's 512
// registers and with the correction's gain vectors next to it the kernel spilled 1.2-1.5 KB per lane (240 us per launch); the two top
// block-rows (75 of the 120 values) live in the wave's 37.5 KiB window of the LDS instead, the three bottom ones in registers, and predict
// and correction stream the top rows through registers a 3 x 3 block at a time: no scratch.
extern __shared__ unsigned char qle_dyn_lds[];
constexpr size_t kMrLdsPerWave = (size_t)kTopWords * kTile * sizeof(double);   // fp64 only (fp32 launches with no dynamic LDS)
template <typename T, bool BLOCKS = (sizeof(T) == 4)> struct MrChain;
template <typename T> struct MrChain<T, true> {
    PackedCov<T> S;
    __device__ __forceinline__ void init() {}
    __device__ __forceinline__ void from_flat(const T (&P)[kPW]) { cov_pack<T>(P, S); }
    __device__ __forceinline__ void load_cov(const T* __restrict__ src, int64_t i)
    {
        T P[kPW];
        load_rec<T, kSW, kXW, kPW>(src, i, P);
        from_flat(P);
    }
    template <int NT> __device__ __forceinline__ void store_cov(T* __restrict__ dst, int64_t i)
    {
        with_flat<false>([&](const T (&P)[kPW]) { store_rec<T, kSW, kXW, kPW, NT>(dst, i, P); });
    }
    // f(P) on the packed triangle; MODIFIES: f changes P
    template <bool MODIFIES, typename F> __device__ __forceinline__ void with_flat(F&& f)
    {
        T P[kPW];
        cov_unpack<T>(S, P);
        f(P);
        if (MODIFIES) cov_pack<T>(P, S);
    }
    __device__ __forceinline__ void predict(const DevParams<T>& p, const Noise<T>& nz, T (&x)[kXW], const T (&u)[kUW], T (&accel)[3])
    {
        ekf_predict_packed<T>(p, nz, x, S, u, accel);
    }
    // correction_step at the entry the measurement belongs to; done(P): the corrected triangle (the new anchor)
    template <bool DIRECT, typename Emit, typename Done>
    __device__ __forceinline__ void correct(const DevParams<T>& p, const Noise<T>& nz, T (&x)[kXW], const T (&z)[7], Emit&& emit, Done&& done)
    {
        with_flat<true>([&](T (&P)[kPW]) {
            ekf_update_emit<T, DIRECT>(p, nz, x, P, z, emit);
            done([&](T* __restrict__ dst, int64_t i) { store_rec<T, kSW, kXW, kPW, 2>(dst, i, P); });
        });
    }
    __device__ __forceinline__ T probe() const { return S.blk(2, 2).d + S.blk(0, 1).c.x; }   // values a predict forms last (diagnostic stamps)
};
template <typename T> struct MrChain<T, false> {
    T lo[kLoWords];
    LdsTop<T> top;
    __device__ __forceinline__ void init()
    {
        top.p = reinterpret_cast<T*>(qle_dyn_lds) + (size_t)(threadIdx.x >> 6) * (kTopWords * kTile) + (threadIdx.x & 63);
    }
    // Record words [W0, W0 + W) of the packed triangle <-> their homes.  The record is moved in two parts with a fence between them
    // (the first 84 words hold block-rows r and v, which go to the LDS): all 120 words at once would be 240 registers in flight next to
    // everything else the kernel holds at that point.
    template <int W0, int W> __device__ __forceinline__ void load_part(const T* __restrict__ src, int64_t i)
    {
        T t[W];
        load_rec<T, kSW, kXW + W0, W>(src, i, t);
        static_for<W0, W0 + W>([&](auto wc) {   // a compile-time loop: every index must be a constant (no array may reach scratch)
            constexpr int w = decltype(wc)::value, hw = split_word(word_row(w), word_col(w));
            if constexpr (word_row(w) < 6) top.st(hw, t[w - W0]);
            else lo[hw] = t[w - W0];
        });
    }
    template <int NT, int W0, int W> __device__ __forceinline__ void store_part(T* __restrict__ dst, int64_t i)
    {
        T t[W];
        static_for<W0, W0 + W>([&](auto wc) {
            constexpr int w = decltype(wc)::value, hw = split_word(word_row(w), word_col(w));
            if constexpr (word_row(w) < 6) t[w - W0] = top.ld(hw);
            else t[w - W0] = lo[hw];
        });
        store_rec<T, kSW, kXW + W0, W, NT>(dst, i, t);
    }
    // the whole triangle at once (callers with little else live: k_run_resident, compact records)
    __device__ __forceinline__ void from_flat(const T (&Pf)[kPW]) { split_from_flat<T>(Pf, top, lo); }
    template <bool MODIFIES, typename F> __device__ __forceinline__ void with_flat(F&& f)
    {
        T P[kPW];
        split_to_flat<T>(top, lo, P);
        f(P);
        if (MODIFIES) split_from_flat<T>(P, top, lo);
    }
    static constexpr int kTopPart = 84;   // words 0..83: block-rows r, v and the first words of row th (sidx order, ekf_device.hpp)
    __device__ __forceinline__ void load_cov(const T* __restrict__ src, int64_t i)
    {
        load_part<0, kTopPart>(src, i);
        QLE_PHASE_FENCE();
        load_part<kTopPart, kPW - kTopPart>(src, i);
        QLE_PHASE_FENCE();
    }
    template <int NT> __device__ __forceinline__ void store_cov(T* __restrict__ dst, int64_t i)
    {
        QLE_PHASE_FENCE();
        store_part<NT, 0, kTopPart>(dst, i);
        QLE_PHASE_FENCE();
        store_part<NT, kTopPart, kPW - kTopPart>(dst, i);
        QLE_PHASE_FENCE();
    }
    __device__ __forceinline__ void predict(const DevParams<T>& p, const Noise<T>& nz, T (&x)[kXW], const T (&u)[kUW], T (&accel)[3])
    {
        ekf_predict_split<T>(p, nz, x, top, lo, u, accel);
    }
    template <bool DIRECT, typename Emit, typename Done>
    __device__ __forceinline__ void correct(const DevParams<T>& p, const Noise<T>& nz, T (&x)[kXW], const T (&z)[7], Emit&& emit, Done&& done)
    {
        ekf_update_split<T, DIRECT>(p, nz, x, top, lo, z, emit);
        done([&](T* __restrict__ dst, int64_t i) { store_cov<2>(dst, i); });
    }
    __device__ __forceinline__ T probe() const { return lo[L_TT] + top.ld(T_RV); }
};

template <typename T, bool DIRECT, bool PFP>
__global__ __launch_bounds__(kBlock) void k_step_mr(T* cur, const T* __restrict__ us, const T* __restrict__ zs, int64_t B, int32_t grid_x, int32_t block_x,
                                                    int32_t* __restrict__ hist_first, T* uring,   // (argument order: see k_predict; the first loads need these 14 dwords)
                                                    T* ckpt, T* anchor, const T* __restrict__ pfp,
                                                    const double* __restrict__ stamp, T* __restrict__ aux_accel, T* __restrict__ aux_obs,
                                                    int32_t* __restrict__ last_corr, uint8_t* __restrict__ flags,
                                                    double* __restrict__ delay_out, DevParams<T> p, GateParams gp, MrParams m)
{
    QLE_ARGS_EARLY(cur, us, zs, B, grid_x, block_x, hist_first, uring, ckpt, anchor);
    const int64_t i = batch_block((unsigned)grid_x) * block_x + threadIdx.x;
    if ((i & ~(int64_t)63) >= B) return;             // the whole wave lies beyond the batch (wave-uniform)
    // from here on all 64 lanes stay active (the record arrays are allocated in whole tiles; a lane beyond B sees a zeroed,
    // i.e. not initialised, filter and never touches the per-filter scalar arrays)
    T x[kXW], u[kUW], accel[3] = {T(0), T(0), T(0)};
    QLE_STAMP(0, (T)(i & 63));
    load_rec<T, kUW, 0, kUW>(us, i, u);
    Noise<T> nz;
    load_noise<T, PFP>(p, pfp, i, nz);
    T zr[kZW];
    load_rec<T, kZW, 0, kZW>(zs, i, zr);
    load_rec<T, kSW, 0, kXW>(cur, i, x);
    // the per-filter history indices are requested with the records above, not after them (they depend on nothing but i)
    const int32_t first_i = i < B ? hist_first[i] : 0;
    const int32_t lastc_i = (m.gate && i < B) ? last_corr[i] : 0;
    QLE_STAMP(1, x[9] + zr[7] + u[5]);
    const bool valid = i < B && !filter_uninitialised(x);   // EKF.cpp:129-130
    bool corr = valid && zr[7] != T(0);
    if (m.gate && valid) {  // EKF.cpp:147-186
        const bool consume = corr && (!gp.limit || (gp.tick - lastc_i) >= gp.upd_per_meas);
        bool ok = consume;
        if (consume && gp.corner_enbl) {
            const double zd[7] = {(double)zr[0], (double)zr[1], (double)zr[2], (double)zr[3], (double)zr[4], (double)zr[5], (double)zr[6]};
            ok = corner_gate(gp, zd);
        }
        corr = ok;
        if (ok) last_corr[i] = gp.tick;
        flags[i] = (uint8_t)((ok ? 1 : 0) | (consume ? 2 : 0));
    }
    int32_t start = m.tick - 1, mt = 0;    // entry the chain starts from; tick the measurement belongs to (if corr)
    const T* sp = cur;
    if (corr) {
        // EKF.cpp:199-201: delay -> step delay -> history entry the measurement belongs to
        int32_t step = m.fixed_step;
        if (m.dynamic) {
            const double age = stamp ? (m.t_curr - stamp[i]) : m.uniform_age;
            const double dcur = fmin(age + m.offset, m.delay_max);
            delay_out[i] = dcur;
            step = (int32_t)(dcur / m.dT + 0.5);
            if (step < 1) step = 1;
        }
        const int32_t first = first_i;
        const int32_t len = m.tick - first;    // entries first .. n-1
        int32_t ind = len - step;
        if (ind < 0) ind = 0;
        mt = first + ind;
        const int32_t c0 = floor_div(mt, m.k) * m.k;   // newest checkpoint tick <= mt
        if (c0 > first) { start = c0; sp = mr_ck_slot(ckpt, m, c0); }
        else { start = first; sp = anchor; }
        // the extra checkpoint, written where the host expected this measurement's entry (a regular cadence: no pre-replay at all)
        if (m.e_tick > start && m.e_tick <= mt) { start = m.e_tick; sp = ckpt + (int64_t)m.Nc * m.slot_words; }
        hist_first[i] = mt;                    // EKF.cpp:214-219
    }
    const int32_t t_lo = wave_min_i32(valid ? start : 0x7fffffff);
    if (t_lo == 0x7fffffff) return;            // no initialised filter in this wave (wave-uniform)
    QLE_STAMP(2, (T)start);
    // The correction of one lane at the entry its measurement belongs to (EKF.cpp:202-211): fuse, then the corrected entry is the anchor.
    auto emit = [&](const T (&o)[7]) {                    // EKF.cpp:209
        if (aux_accel) {
#pragma unroll
            for (int k = 0; k < 7; ++k) aux_obs[i * 7 + k] = o[k];
        }
    };
    auto new_anchor = [&](auto&& store_cov_to) {          // EKF.cpp:210-211: the history now starts here
        store_rec<T, kSW, 0, kXW, 2>(anchor, i, x);
        store_cov_to(anchor, i);
    };
    MrChain<T> S;
    S.init();
    auto correct_chain = [&]() {
        T z[7];
        if constexpr (sizeof(T) == 8) {   // fp64: the tag pose is read again here instead of occupying 16 registers through the pre-replay
            T zq[kZW];
            load_rec<T, kZW, 0, kZW>(zs, i, zq);
#pragma unroll
            for (int k = 0; k < 7; ++k) z[k] = zq[k];
        } else {
#pragma unroll
            for (int k = 0; k < 7; ++k) z[k] = zr[k];
        }
        if constexpr (sizeof(T) == 8 && PFP) load_noise<T, PFP>(p, pfp, i, nz);
        S.template correct<DIRECT>(p, nz, x, z, emit, new_anchor);
    };
    // fp32, regular cadence: every lane's chain starts AT its measurement's entry (the extra checkpoint).  The correction then runs on the
    // loaded triangle directly -- its scalar chains (innovation, R_k) under the tail of the 36 MB load, no pack / unpack round trip through
    // the register blocks in front of it -- and the loop below finds nothing left to correct (wave-uniform choice).
    // The IMU samples of the next kQ replayed ticks are requested ahead (wave-uniform slot addresses), the first kQ in front of the
    // correction.  kQ = 1 ships: one step of arithmetic (~1.8 us) covers the latency of the ring, which was streamed to HBM; deeper queues
    // (the idea: a sample requested behind the 36 / 72 MB of anchor stores is not delivered before they have drained) measured no gain, and
    // requesting the whole window up front (LDS-DMA, profiles/r03_tuning.md) made the prologue 15 000 cycles longer.
    // A sample index beyond the current tick has no request; the current tick's own sample comes from `us` (it is asked for again here so
    // that it is not carried in registers through the whole replay).
    constexpr int kQ = sizeof(T) == 4 ? QLE_MR_PREFETCH_F32 : QLE_MR_PREFETCH_F64;
    static_assert(kQ >= 1 && kQ <= 4, "the sample queue is four named register arrays");
    T un0[kUW], un1[kUW], un2[kUW], un3[kUW];   // separate arrays: a [kQ][kHW] array was left in scratch by the backend
    auto request_sample = [&](int32_t ts, T (&dst)[kUW]) { mr_request_sample<T>(m, uring, us, i, ts, dst); };   // ts is wave-uniform
    request_sample(t_lo + 1, un0);
    if constexpr (kQ > 1) request_sample(t_lo + 2, un1);
    if constexpr (kQ > 2) request_sample(t_lo + 3, un2);
    if constexpr (kQ > 3) request_sample(t_lo + 4, un3);
    bool early = false;
    if (sp != cur) load_rec<T, kSW, 0, kXW>(sp, i, x);
    if constexpr (sizeof(T) == 4) {
        T P[kPW];
        load_rec<T, kSW, kXW, kPW>(sp, i, P);
        early = __ballot(valid && !(corr && start == mt)) == 0;
        if (early && corr) {
            QLE_STAMP(5, x[0]);
            const T z[7] = {zr[0], zr[1], zr[2], zr[3], zr[4], zr[5], zr[6]};
            ekf_update_emit<T, DIRECT>(p, nz, x, P, z, emit);
            new_anchor([&](T* __restrict__ dst, int64_t ii) { store_rec<T, kSW, kXW, kPW, 2>(dst, ii, P); });
            QLE_STAMP(6, x[0]);
        }
        S.from_flat(P);
        QLE_STAMP(3, P[0] + P[119] + x[0]);
    } else {
        S.load_cov(sp, i);
        QLE_STAMP(3, S.probe() + x[0]);
    }
    // The chain.  Two copies of the loop: the first runs up to the last entry any lane of the wave corrects at (wave-uniform t_cmax) with
    // the correction inside; the second takes the rest -- after that tick nothing of the correction (the tag pose, R, its temporaries) is
    // live across the replayed ticks, which is what the 256-VGPR kernel is short of.  On a regular cadence (`early`) the first has nothing to do.
    int32_t t = t_lo;                          // wave-uniform
    int dbg_j = 0;
    (void)dbg_j;
    auto chain = [&](auto corr_in_loop, int32_t t_stop) {
    for (;;) {
        if constexpr (decltype(corr_in_loop)::value) {
            if (corr && t == mt) {                            // the entry the measurement belongs to
                QLE_STAMP(5, x[0]);
                correct_chain();
                QLE_STAMP(6, x[0]);
            }
        }
        if (t == t_stop) break;
        ++t;                                                  // EKF.cpp:222-226, then :249
        const bool now = t == m.tick;                         // wave-uniform
        T u6[kUW];
#pragma unroll
        for (int k = 0; k < kUW; ++k) u6[k] = un0[k];
        QLE_STAMP(8 + 2 * dbg_j, u6[0] + u6[5]);
#pragma unroll
        for (int k = 0; k < kUW; ++k) {
            if constexpr (kQ > 1) un0[k] = un1[k];
            if constexpr (kQ > 2) un1[k] = un2[k];
            if constexpr (kQ > 3) un2[k] = un3[k];
        }
        if constexpr (kQ == 1) request_sample(t + 1, un0);
        else if constexpr (kQ == 2) request_sample(t + 2, un1);
        else if constexpr (kQ == 3) request_sample(t + 3, un2);
        else request_sample(t + 4, un3);
        if (valid && t > start) {
            if constexpr (sizeof(T) == 8 && PFP) load_noise<T, PFP>(p, pfp, i, nz);   // fp64: 24 values read again (L2) rather than 48 registers held through the loop
            S.predict(p, nz, x, u6, accel);
            QLE_STAMP(9 + 2 * dbg_j, x[0] + x[9] + S.probe());
            const bool extra = t == m.e_tick;                 // wave-uniform
            const bool ck = (t % m.k == 0 || extra) && (now || (corr && t > mt));   // checkpoints of the rewritten part of the chain
            if constexpr (sizeof(T) == 4) {
                if (now || ck) {
                    S.template with_flat<false>([&](const T (&P)[kPW]) {
                        if (now) {
                            store_rec<T, kSW, 0, kXW>(cur, i, x);
                            store_rec<T, kSW, kXW, kPW>(cur, i, P);
                        }
                        if (ck) {
                            T* ckp = extra ? ckpt + (int64_t)m.Nc * m.slot_words : mr_ck_slot(ckpt, m, t);
                            store_rec<T, kSW, 0, kXW, 2>(ckp, i, x);
                            store_rec<T, kSW, kXW, kPW, 2>(ckp, i, P);
                        }
                    });
                }
            } else {
                if (now) {
                    store_rec<T, kSW, 0, kXW>(cur, i, x);
                    S.template store_cov<0>(cur, i);
                }
                if (ck) {
                    T* ckp = extra ? ckpt + (int64_t)m.Nc * m.slot_words : mr_ck_slot(ckpt, m, t);
                    store_rec<T, kSW, 0, kXW, 2>(ckp, i, x);
                    S.template store_cov<2>(ckp, i);
                }
            }
            if (now) {
                const T uk[kHW] = {u6[0], u6[1], u6[2], u6[3], u6[4], u6[5], T(0), T(0)};
                store_rec<T, kHW, 0, kHW, QLE_RING_POLICY>(mr_u_slot(uring, m, t), i, uk);   // EKF.cpp:254-256
            }
        }
#ifdef QLE_MR_STAMPS
        ++dbg_j;
#endif
    }
    };
    {
        const int32_t t_cmax = early ? (int32_t)0x80000000 : wave_max_i32(corr ? mt : (int32_t)0x80000000);
        if (t_cmax >= t_lo) chain(std::true_type{}, t_cmax);       // mt >= start >= t_lo for every correcting lane
        chain(std::false_type{}, m.tick);
    }
    QLE_STAMP(7, x[0]);
    if (aux_accel && valid) {
#pragma unroll
        for (int k = 0; k < 3; ++k) aux_accel[i * 3 + k] = accel[k];
    }
}

template <typename I>   // a template only so that every translation unit may include this header
__global__ void k_fill_i32(I* __restrict__ dst, I v, int64_t B)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) dst[i] = v;
}

// On-chip-resident multi-tick kernel: x and P stay in registers for n ticks of the single-rate
// filter (predict every tick, correct where the tick has a tag record whose mask word is set).
// Per tick only the 6-word IMU record (and the 8-word tag record on measurement ticks) is read;
// the next tick's IMU record is loaded before the current tick's arithmetic.
template <typename T, bool DIRECT, bool PFP, bool COMPACT = false>
__global__ __launch_bounds__(kBlock) void k_run_resident(DevParams<T> p, T* st, const T* __restrict__ us, const T* __restrict__ zs,
                                                         const int32_t* __restrict__ slot, int64_t pitch_u, int64_t pitch_z, int64_t T_seq,
                                                         int64_t t0, int64_t n, const T* __restrict__ pfp, int64_t B)
{
    QLE_ARGS_EARLY(st, us, zs, slot, B, gridDim.x, blockDim.x);
    const int64_t i = batch_block() * blockDim.x + threadIdx.x;
    if (i >= B) return;
    T x[kXW], u[kUW], un[kUW], accel[3];
    load_rec<T, kSW, 0, kXW>(st, i, x);
    // fp64: the covariance split between the LDS and registers (ekf_split.hpp; launched with kMrLdsPerWave of dynamic LDS per wave) -- the flat
    // triangle with the sequential update next to it spilled 0.8-1.2 KB per lane here as it did in k_step_mr; fp32: the flat triangle
    constexpr bool kSplit = sizeof(T) == 8;
    MrChain<T, !kSplit> S;      // (fp32 instantiates the block form's type only to keep one declaration; it is not used there)
    T P[kSplit ? 1 : kPW];
    if constexpr (kSplit) {
        S.init();
        T Pf[kPW];
        load_P_any<T>(st, i, Pf, COMPACT);
        S.from_flat(Pf);
    } else {
        load_P_any<T>(st, i, P, COMPACT);
    }
    if (filter_uninitialised(x)) return;
    Noise<T> nz;
    load_noise<T, PFP>(p, pfp, i, nz);
    int64_t t = t0 % T_seq;
    load_rec<T, kUW, 0, kUW>(us + t * pitch_u, i, u);
    for (int64_t k = 0; k < n; ++k) {
        const int64_t tn = (t + 1 == T_seq) ? 0 : t + 1;
        if (k + 1 < n) load_rec<T, kUW, 0, kUW>(us + tn * pitch_u, i, un);
        const int32_t s = slot[t];  // wave-uniform
        T zr[kZW];
        if (s >= 0) load_rec<T, kZW, 0, kZW>(zs + (int64_t)s * pitch_z, i, zr);
        if constexpr (kSplit) {
            if constexpr (PFP) load_noise<T, PFP>(p, pfp, i, nz);   // read again per tick (L2) rather than 48 registers held through the loop
            S.predict(p, nz, x, u, accel);
        } else {
            ekf_predict<T>(p, nz, x, P, u, accel);
        }
        if (s >= 0 && zr[7] != T(0)) {
            const T z[7] = {zr[0], zr[1], zr[2], zr[3], zr[4], zr[5], zr[6]};
            if constexpr (kSplit) ekf_update_split<T, DIRECT>(p, nz, x, S.top, S.lo, z, [](const T (&)[7]) {});
            else ekf_update_emit<T, DIRECT>(p, nz, x, P, z, [](const T (&)[7]) {});
        }
#pragma unroll
        for (int c = 0; c < kUW; ++c) u[c] = un[c];
        t = tn;
    }
    store_rec<T, kSW, 0, kXW>(st, i, x);
    if constexpr (kSplit) S.template with_flat<false>([&](const T (&Pf)[kPW]) { store_P_any<T>(st, i, Pf, COMPACT); });
    else store_P_any<T>(st, i, P, COMPACT);
}

// Shift the tick origin: subtract `shift` from every filter's last-correction index so that the
// 32-bit tick arithmetic never wraps in a long-running service.  "Never / long ago" saturates.
template <typename I>
__global__ void k_rebase_ticks(I* __restrict__ last_corr, I shift, int64_t B)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    // saturate far in the past: such a filter has not corrected / has no history entry for longer than any
    // rate limit or ring capacity, which is all the consumers of these indices distinguish
    const int64_t v = (int64_t)last_corr[i] - shift;
    last_corr[i] = (int32_t)(v < -(int64_t)(1 << 30) ? -(int64_t)(1 << 30) : v);
}

// upds_since_correction (EKF.hpp:128) per filter from the implicit counter: ticks since the filter's last correction,
// 0 for a filter that is not initialised yet (the reference never advances it, EKF.cpp:129-130).
template <typename T>
__global__ void k_upds_since(const T* __restrict__ st, const int32_t* __restrict__ last_corr, int32_t tick, int32_t* __restrict__ out, int64_t B)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    bool init = false;
    for (int w = 6; w < 10; ++w) init |= st[word_off<T>(w, i, kSW)] != T(0);
    out[i] = init ? tick - 1 - last_corr[i] : 0;
}

// Stand-alone correction (correction_step, EKF.cpp:417-502) where mask != 0.
template <typename T, bool DIRECT, bool PFP, bool COMPACT = false>
__global__ __launch_bounds__(kBlock) void k_update(T* __restrict__ st, const T* __restrict__ zs, int64_t B, int32_t grid_x, int32_t block_x,   // (argument order: see k_predict)
                                                   const T* __restrict__ pfp, T* __restrict__ aux_obs, DevParams<T> p)
{
    QLE_ARGS_EARLY(st, zs, B, grid_x, block_x);
    const int64_t i = batch_block((unsigned)grid_x) * block_x + threadIdx.x;
    if (i >= B) return;
    T zr[kZW];
    load_rec<T, kZW, 0, kZW>(zs, i, zr);
    if (zr[7] == T(0)) return;
    T x[kXW];
    load_rec<T, kSW, 0, kXW>(st, i, x);
    Noise<T> nz;
    T z[7] = {zr[0], zr[1], zr[2], zr[3], zr[4], zr[5], zr[6]};
    T obs[7];
    if constexpr (sizeof(T) == 8) {   // fp64: the split covariance and the batch-form correction (no scratch; see k_run_resident)
        MrChain<T, false> S;
        S.init();
        if constexpr (COMPACT) {
            T Pf[kPW];
            load_P_compact<T>(st, i, Pf);
            S.from_flat(Pf);
        } else {
            S.load_cov(st, i);
        }
        if (filter_uninitialised(x)) return;
        load_noise<T, PFP>(p, pfp, i, nz);
        ekf_update_split<T, DIRECT>(p, nz, x, S.top, S.lo, z, [&](const T (&o)[7]) {
#pragma unroll
            for (int k = 0; k < 7; ++k) obs[k] = o[k];
        });
        store_rec<T, kSW, 0, kXW>(st, i, x);
        if constexpr (COMPACT) S.template with_flat<false>([&](const T (&Pf)[kPW]) { store_P_compact<T>(st, i, Pf); });
        else S.template store_cov<0>(st, i);
    } else {
        T P[kPW];
        load_P_any<T>(st, i, P, COMPACT);
        if (filter_uninitialised(x)) return;
        load_noise<T, PFP>(p, pfp, i, nz);
        ekf_update<T, DIRECT>(p, nz, x, P, z, obs);
        store_rec<T, kSW, 0, kXW>(st, i, x);
        store_P_any<T>(st, i, P, COMPACT);
    }
    if (aux_obs) {
#pragma unroll
        for (int k = 0; k < 7; ++k) aux_obs[i * 7 + k] = obs[k];
    }
}

// ------------------------------------------------ layout conversion kernels
// Host-facing AoS fp64 <-> device tiles, one chunk [i0, i0+n) of the batch per
// launch (the AoS side is a staging buffer holding only that chunk).  W words
// of the host row go to words [w0, w0+W) of the WT-word device record.
// Not on the hot path.
template <typename T>
__global__ void k_pack_off(const double* __restrict__ aos, int stride, int W, T* __restrict__ dst, int WT, int w0, int64_t i0, int64_t n)
{
    const int64_t li = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= n) return;
    for (int w = 0; w < W; ++w) dst[word_off<T>(w0 + w, i0 + li, WT)] = (T)aos[li * stride + w];
}
template <typename T>
__global__ void k_unpack_off(const T* __restrict__ src, int stride, int W, double* __restrict__ aos, int WT, int w0, int64_t i0, int64_t n)
{
    const int64_t li = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= n) return;
    for (int w = 0; w < W; ++w) aos[li * stride + w] = (double)src[word_off<T>(w0 + w, i0 + li, WT)];
}
// z (7) + mask -> 8-word record; z == nullptr writes an identity pose, mask == nullptr means "all".
template <typename T>
__global__ void k_pack_z_off(const double* __restrict__ z, const uint8_t* __restrict__ mask, T* __restrict__ dst, int64_t i0, int64_t n)
{
    const int64_t li = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= n) return;
    for (int w = 0; w < 7; ++w) dst[word_off<T>(w, i0 + li, kZW)] = z ? (T)z[li * 7 + w] : (w == 6 ? T(1) : T(0));
    dst[word_off<T>(7, i0 + li, kZW)] = (mask == nullptr || mask[li]) ? T(1) : T(0);
}
template <typename T>
__global__ void k_unpack_z_off(const T* __restrict__ src, double* __restrict__ z, uint8_t* __restrict__ mask, int64_t i0, int64_t n)
{
    const int64_t li = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= n) return;
    for (int w = 0; w < 7; ++w) z[li * 7 + w] = (double)src[word_off<T>(w, i0 + li, kZW)];
    mask[li] = src[word_off<T>(7, i0 + li, kZW)] != T(0) ? 1 : 0;
}
// Full n x n row-major covariance -> packed symmetric part (P + P^T)/2 of the state record.
template <typename T>
__global__ void k_pack_P_off(const double* __restrict__ Pf, int n, T* __restrict__ st, int64_t i0, int64_t m, int compact)
{
    const int64_t li = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= m) return;
    const double* Pi = Pf + li * n * n;
    for (int a = 0; a < 15; ++a)
        for (int b = a; b < 15; ++b) {
            double v = (a < n && b < n) ? 0.5 * (Pi[a * n + b] + Pi[b * n + a]) : 0.0;
            const int w = p_word(a, b, compact != 0);
            if (w >= 0) st[word_off<T>(w, i0 + li, kSW)] = (T)v;
        }
}
template <typename T>
__global__ void k_unpack_P_off(const T* __restrict__ st, int n, double* __restrict__ Pf, int64_t i0, int64_t m, int compact)
{
    const int64_t li = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= m) return;
    double* Pi = Pf + li * n * n;
    for (int a = 0; a < n; ++a)
        for (int b = 0; b < n; ++b) {
            const int w = a <= b ? p_word(a, b, compact != 0) : p_word(b, a, compact != 0);
            Pi[a * n + b] = w >= 0 ? (double)st[word_off<T>(w, i0 + li, kSW)] : 0.0;
        }
}

// The covariance part of every record from one layout to the other (a handle re-configured with the other est_bias, qle_set_params).
template <typename T>
__global__ void k_relayout_P(T* __restrict__ st, int from_compact, int to_compact, int64_t B)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    T P[kPW];
    load_P_any<T>(st, i, P, from_compact != 0);
    store_P_any<T>(st, i, P, to_compact != 0);
}

// initialize_state, EKF.cpp:305-344, one filter per lane, for the filters whose tag record's mask word is set
// (the node seeds a filter on ITS first detection, NODE.cpp:169-174).  A filter that was not initialised before starts
// its counters here: upds_since_correction = 0 (EKF.cpp:77), i.e. last_corr = tick - 1.  Every seeded filter restarts
// its multirate history with the single entry "state now" (EKF.cpp:337-339).
template <typename T>
__global__ void k_seed(DevParams<T> p, const T* __restrict__ zs, T* __restrict__ st, T cov0, T cov1, T cov2, T cov3, T cov4,
                       int reinit_bias, int32_t tick, int32_t* __restrict__ last_corr, int32_t* __restrict__ hist_first,
                       T* __restrict__ anchor, int64_t B)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    T zr[kZW], x[kXW], P[kPW];
    load_rec<T, kZW, 0, kZW>(zs, i, zr);
    if (zr[7] == T(0)) return;
    load_rec<T, kSW, 0, kXW>(st, i, x);
    const bool fresh = filter_uninitialised(x);
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
    store_rec<T, kSW, 0, kXW>(st, i, x);
    store_P_any<T>(st, i, P, p.compact != 0);
    if (fresh && last_corr) last_corr[i] = tick - 1;
    if (hist_first) {   // multirate: the history is the single entry "state now" (EKF.cpp:337-339)
        hist_first[i] = tick - 1;
        store_rec<T, kSW, 0, kXW>(anchor, i, x);
        store_rec<T, kSW, kXW, kPW>(anchor, i, P);
    }
}

// What the node publishes after a tick (NODE.cpp:192-220), AoS fp64, one chunk.
template <typename T>
__global__ void k_report_off(DevParams<T> p, const T* __restrict__ st, const T* __restrict__ pfp, double* __restrict__ pose,
                             double* __restrict__ pose_cov, double* __restrict__ vel, double* __restrict__ bias, int64_t i0, int64_t n)
{
    const int64_t li = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= n) return;
    const int64_t i = i0 + li;
    auto X = [&](int w) { return (double)st[word_off<T>(w, i, kSW)]; };
    for (int k = 0; k < 3; ++k) pose[li * 7 + k] = X(k);
    for (int k = 0; k < 4; ++k) pose[li * 7 + 3 + k] = X(6 + k);
    {  // rows/cols {0-2, 6-8}, row-major (NODE.cpp:203-210)
        const int sel[6] = {0, 1, 2, 6, 7, 8};
        for (int a = 0; a < 6; ++a)
            for (int b = 0; b < 6; ++b)
                pose_cov[li * 36 + a * 6 + b] = X(sel[a] <= sel[b] ? p_word(sel[a], sel[b], p.compact != 0) : p_word(sel[b], sel[a], p.compact != 0));
    }
    for (int k = 0; k < 3; ++k) vel[li * 3 + k] = X(3 + k);
    for (int k = 0; k < 3; ++k) {  // ab_nom + ab_static, wb_nom + wb_static (NODE.cpp:215-220)
        double as = pfp ? (double)pfp[word_off<T>(12 + k, i, kFW)] : (double)p.ab_static[k];
        double ws = pfp ? (double)pfp[word_off<T>(15 + k, i, kFW)] : (double)p.wb_static[k];
        bias[li * 6 + k] = X(10 + k) + as;
        bias[li * 6 + 3 + k] = X(13 + k) + ws;
    }
}

template <typename T>
__global__ void k_count_nonfinite(const T* __restrict__ st, unsigned long long* __restrict__ out, int64_t B, int record_words)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    bool bad = false;
    for (int w = 0; w < record_words; ++w) bad |= !isfinite((double)st[word_off<T>(w, i, kSW)]);   // the words a tick reads (64 in compact records)
    if (bad) atomicAdd(out, 1ULL);
}

}  // namespace qle
