// q3_wave_sort.h — 64-lane bitonic sorting network on registers (one element per lane), descending.
// Partner exchange never touches LDS: quad_perm for lane^1 / lane^2, bank-masked row shifts for lane^4, row_ror:8 for lane^8,
// and gfx950's v_permlane16_swap / v_permlane32_swap for lane^16 / lane^32.  21 compare-exchange stages, ~0.3 us for a lone wave
// (the 64-broadcast rank count it replaces in the sampler took 1.4 us).
#pragma once
#include <hip/hip_runtime.h>

namespace q3 {

template <int J>
static __device__ __forceinline__ int wave_xor_lane_i(int v, int lane) {
    if (J == 1) return __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false);          // quad_perm:[1,0,3,2]
    if (J == 2) return __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false);          // quad_perm:[2,3,0,1]
    if (J == 4) {   // banks 0 and 2 of each row take lane+4 (row_shl:4), banks 1 and 3 take lane-4 (row_shr:4)
        int t = __builtin_amdgcn_update_dpp(v, v, 0x104, 0xF, 0x5, false);
        return __builtin_amdgcn_update_dpp(t, v, 0x114, 0xF, 0xA, false);
    }
    if (J == 8) return __builtin_amdgcn_update_dpp(v, v, 0x128, 0xF, 0xF, false);         // row_ror:8 == lane^8 inside a row of 16
    if (J == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
        return (lane & 16) ? (int)r[0] : (int)r[1];
    }
    const auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    return (lane & 32) ? (int)r[0] : (int)r[1];
}
template <int J>
static __device__ __forceinline__ float wave_xor_lane_f(float v, int lane) {
    return __builtin_bit_cast(float, wave_xor_lane_i<J>(__builtin_bit_cast(int, v), lane));
}

// keys only
template <int K, int J>
static __device__ __forceinline__ float wave_sort_step(float v, int lane) {
    const float pv = wave_xor_lane_f<J>(v, lane);
    const bool first = ((lane & J) == 0) == ((lane & K) == 0);   // this lane keeps the larger element of the pair
    return first ? fmaxf(v, pv) : fminf(v, pv);
}
static __device__ __forceinline__ float wave_sort_desc(float v, int lane) {
    v = wave_sort_step<2, 1>(v, lane);
    v = wave_sort_step<4, 2>(v, lane); v = wave_sort_step<4, 1>(v, lane);
    v = wave_sort_step<8, 4>(v, lane); v = wave_sort_step<8, 2>(v, lane); v = wave_sort_step<8, 1>(v, lane);
    v = wave_sort_step<16, 8>(v, lane); v = wave_sort_step<16, 4>(v, lane); v = wave_sort_step<16, 2>(v, lane); v = wave_sort_step<16, 1>(v, lane);
    v = wave_sort_step<32, 16>(v, lane); v = wave_sort_step<32, 8>(v, lane); v = wave_sort_step<32, 4>(v, lane); v = wave_sort_step<32, 2>(v, lane);
    v = wave_sort_step<32, 1>(v, lane);
    v = wave_sort_step<64, 32>(v, lane); v = wave_sort_step<64, 16>(v, lane); v = wave_sort_step<64, 8>(v, lane); v = wave_sort_step<64, 4>(v, lane);
    v = wave_sort_step<64, 2>(v, lane); v = wave_sort_step<64, 1>(v, lane);
    return v;
}

// Four sorted runs of 16 (lanes 0-15 descending, 16-31 ASCENDING, 32-47 descending, 48-63 ASCENDING: two bitonic blocks of 32) -> all 64
// descending: the two bitonic merges of the network only, 11 stages instead of the full sort's 21.
static __device__ __forceinline__ float wave_merge4x16_desc(float v, int lane) {
    v = wave_sort_step<32, 16>(v, lane); v = wave_sort_step<32, 8>(v, lane); v = wave_sort_step<32, 4>(v, lane); v = wave_sort_step<32, 2>(v, lane);
    v = wave_sort_step<32, 1>(v, lane);
    v = wave_sort_step<64, 32>(v, lane); v = wave_sort_step<64, 16>(v, lane); v = wave_sort_step<64, 8>(v, lane); v = wave_sort_step<64, 4>(v, lane);
    v = wave_sort_step<64, 2>(v, lane); v = wave_sort_step<64, 1>(v, lane);
    return v;
}

// (key, tag) pairs: descending key, ascending tag among equal keys (tags distinct)
template <int K, int J>
static __device__ __forceinline__ void wave_sort_step_kv(float& k, int& t, int lane) {
    const float pk = wave_xor_lane_f<J>(k, lane);
    const int pt = wave_xor_lane_i<J>(t, lane);
    const bool first = ((lane & J) == 0) == ((lane & K) == 0);
    const bool partner_precedes = pk > k || (pk == k && pt < t);
    const bool take = first == partner_precedes;
    k = take ? pk : k;
    t = take ? pt : t;
}
static __device__ __forceinline__ void wave_sort_desc_kv(float& k, int& t, int lane) {
    wave_sort_step_kv<2, 1>(k, t, lane);
    wave_sort_step_kv<4, 2>(k, t, lane); wave_sort_step_kv<4, 1>(k, t, lane);
    wave_sort_step_kv<8, 4>(k, t, lane); wave_sort_step_kv<8, 2>(k, t, lane); wave_sort_step_kv<8, 1>(k, t, lane);
    wave_sort_step_kv<16, 8>(k, t, lane); wave_sort_step_kv<16, 4>(k, t, lane); wave_sort_step_kv<16, 2>(k, t, lane); wave_sort_step_kv<16, 1>(k, t, lane);
    wave_sort_step_kv<32, 16>(k, t, lane); wave_sort_step_kv<32, 8>(k, t, lane); wave_sort_step_kv<32, 4>(k, t, lane); wave_sort_step_kv<32, 2>(k, t, lane);
    wave_sort_step_kv<32, 1>(k, t, lane);
    wave_sort_step_kv<64, 32>(k, t, lane); wave_sort_step_kv<64, 16>(k, t, lane); wave_sort_step_kv<64, 8>(k, t, lane); wave_sort_step_kv<64, 4>(k, t, lane);
    wave_sort_step_kv<64, 2>(k, t, lane); wave_sort_step_kv<64, 1>(k, t, lane);
}

// inclusive prefix sums in lane order on the DPP path (row_shr 1/2/4/8 with zero fill, then row_bcast15 / row_bcast31)
static __device__ __forceinline__ float wave_scan_incl_f(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, false));
    return v;
}
static __device__ __forceinline__ int wave_scan_incl_i(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);
    return v;
}

} // namespace q3
