// q3_gemm_kernels.hip — skinny-M projections for batched decode on the bf16 matrix cores.
//
// out[M][N] = epi(X[M][K] . W[N][K]^T): the weights (bf16, nn.Linear layout) are streamed exactly once per
// launch and every M row reuses them — the batch dimension is what lifts the decode step off the
// launch-latency floor of the b=1 path.  Parity with the fp32 oracle is kept by feeding the activations
// as TWO bf16 planes, x = hi + lo (|x - hi - lo| <= 2^-18 |x|): bf16 x bf16 products are exact in the
// fp32 accumulator, so two v_mfma_f32_16x16x32_bf16 per tile give fp32-grade dot products at 1/8 of the
// fp32-MFMA cost.  Fragment layout of both kernels:
//   A (x):  lane l -> row m0 + (l&15), k0 + 8*(l>>4) .. +7
//   B (W):  lane l -> weight row n0 + (l&15), same k              (HBM stream)
//   k_gemv16 (3..16 rows):   one launch per projection, fp32 rows in, RMSNorm and epilogue fused, hi/lo split in registers
//   k_gemm2  (17..128 rows): planes written by the producers (k_finish*, k_attn), activation slice staged through LDS and
//                            shared by 64 columns, split-K slabs reduced in fixed order by k_finish / k_finish_swiglu
// Summation order is fixed everywhere: results are bit-reproducible run to run.
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <unordered_map>

#include "q3_common.h"

namespace q3 {

// -DQ3_SAMPLE_PROF: wall-clock stamps (100 MHz) inside the gate/up k_gemm2 launch and k_finish, workgroup 0 / thread 0; tools/ only
#ifdef Q3_SAMPLE_PROF
__device__ long long g_gemm_prof[32];
void gemm_prof_read(long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gemm_prof), sizeof(long long) * 32); }
#define GP_MARK(cond, k) do { if ((cond) && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_gemm_prof[k] = wall_clock64(); } while (0)
// split-K seam: every K-slice workgroup of column tile 0 / row block 0 stamps its own row [kind = 0 o_proj, 1 gate/up, 2 down][slice][mark]
__device__ long long g_seam_prof[3][16][8];
void seam_prof_read(long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_seam_prof), sizeof(long long) * 3 * 16 * 8); }
#define SP_SEAM(k) do { if (blockIdx.x == 0 && blockIdx.z == 0 && threadIdx.x == 0) g_seam_prof[SEAM == 2 ? 1 : (gridDim.y > 8 ? 2 : 0)][blockIdx.y & 15][k] = wall_clock64(); } while (0)
#else
#define GP_MARK(cond, k) do { } while (0)
#define SP_SEAM(k) do { } while (0)
#endif

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static __device__ __forceinline__ float silu_g(float x) { return x / (1.0f + expf(-x)); }
static __device__ __forceinline__ bf16_t bf16_rne(float f) {
    uint32_t u = __float_as_uint(f);
    return (bf16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
static __device__ __forceinline__ void split_store(float v, bf16_t* hi, bf16_t* lo) {
    const bf16_t h = bf16_rne(v);
    *hi = h;
    *lo = bf16_rne(v - __uint_as_float((uint32_t)h << 16)); // v - hi is exact in fp32
}

static __device__ __forceinline__ void split_store4(const float (&y)[4], bf16_t* hi, bf16_t* lo) {   // 4 values -> one 8-byte store per plane
    bf16_t h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { h[j] = bf16_rne(y[j]); l[j] = bf16_rne(y[j] - __uint_as_float((uint32_t)h[j] << 16)); }
    *reinterpret_cast<uint2*>(hi) = make_uint2((uint32_t)h[0] | (uint32_t)h[1] << 16, (uint32_t)h[2] | (uint32_t)h[3] << 16);
    *reinterpret_cast<uint2*>(lo) = make_uint2((uint32_t)l[0] | (uint32_t)l[1] << 16, (uint32_t)l[2] | (uint32_t)l[3] << 16);
}
// ---- fragment-packed weight copies (q3_common.h) ----
__global__ void k_pack_mfma_b(const bf16_t* W, bf16_t* P, int N, int K) {
    const int KS = K / 32;
    const size_t n_frag = (size_t)((N + 15) / 16) * KS * 64;            // one 16-byte fragment per (tile, k-step, lane)
    for (size_t f = (size_t)blockIdx.x * blockDim.x + threadIdx.x; f < n_frag; f += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(f & 63);
        const size_t ts = f >> 6;
        const int s = (int)(ts % KS), t = (int)(ts / KS);
        int row = t * 16 + (lane & 15);
        row = row < N ? row : N - 1;
        const uint4 v = *reinterpret_cast<const uint4*>(W + (size_t)row * K + s * 32 + (lane >> 4) * 8);
        *reinterpret_cast<uint4*>(P + f * 8) = v;
    }
}
size_t packed_mfma_b_elems(int N, int K) { return (size_t)((N + 15) / 16) * (K / 32) * 64 * 8; }
void launch_pack_mfma_b(const bf16_t* W, bf16_t* P, int N, int K, hipStream_t s) {
    if (K % 32 != 0) throw Error("pack_mfma_b: K must be a multiple of 32");
    hipLaunchKernelGGL(k_pack_mfma_b, dim3(2048), dim3(256), 0, s, W, P, N, K);
}
static std::mutex g_packed_mu;
static std::unordered_map<const bf16_t*, const bf16_t*> g_packed;
void register_packed_weight(const bf16_t* W, const bf16_t* P) { std::lock_guard<std::mutex> lk(g_packed_mu); g_packed[W] = P; }
void unregister_packed_weight(const bf16_t* W) { std::lock_guard<std::mutex> lk(g_packed_mu); g_packed.erase(W); }
const bf16_t* find_packed_weight(const bf16_t* W) {
    if (W == nullptr) return nullptr;
    if (const char* k = knob("Q3TTS_PACKED_W")) if (atoi(k) == 0) return nullptr;
    std::lock_guard<std::mutex> lk(g_packed_mu);
    const auto it = g_packed.find(W);
    return it == g_packed.end() ? nullptr : it->second;
}

// ================================================================================================
// k_gemm2 — second-generation batched-decode GEMM.  Workgroup = (n-group of NW*16 columns, K slice):
// its NW waves each own one 16-column tile and SHARE the activation slice, staged once per 128-wide
// K chunk into LDS (row stride padded by 16 B so the ds_read_b128 fragment reads are conflict-free).
// K slices of different workgroups meet through fp32 partial slabs [ks][M][N] that k_finish reduces in
// fixed order (deterministic) — this is what lets the N=1024 projections (o_proj, down) use
// 128-192 workgroups instead of 64, and cuts the L2 traffic of the activation planes 4x.
// ================================================================================================
#define G2_KC 128
#define G2_LD (G2_KC + 8)

template <int MTILES, int EPI, int NW>
__global__ __launch_bounds__(NW * 64) void k_gemm2(const bf16_t* pW, const bf16_t* pW2, const bf16_t* pxh, const bf16_t* pxl, int pldx, int pM, int pN, int pK,
                                                     GemmArgs a) {   // leading scalars: kernarg-preloaded, see k_gemv1 in q3_decode_kernels.hip
    constexpr bool DUAL = EPI == EPI_SWIGLU || EPI == EPI_SLAB2;
    constexpr int NT = NW * 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, q = lane >> 4;
    const int n0 = (blockIdx.x * NW + wave) * 16;
    const int K = pK, M = pM;
    const int kslice = K / gridDim.y, kbeg = blockIdx.y * kslice;
    __shared__ __attribute__((aligned(16))) bf16_t xs[2][MTILES * 16][G2_LD];

    int nrow = n0 + r16;
    nrow = nrow < pN ? nrow : pN - 1;
    const bf16_t* wp = pW + (size_t)nrow * K + q * 8;
    const bf16_t* wp2 = DUAL ? pW2 + (size_t)nrow * K + q * 8 : nullptr;
    f32x4 acc[MTILES], acc2[MTILES];
#pragma unroll
    for (int mt = 0; mt < MTILES; ++mt) { acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[mt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    // staging assignment: 16 threads per row (8 bf16 = 16 B each), NT/16 rows per pass
    constexpr int ROWS = MTILES * 16, RPP = NT / 16, PASSES = (ROWS + RPP - 1) / RPP;
    const int srow = tid >> 4, scol = (tid & 15) * 8;

    // Two register sets: while chunk c is multiplied out of LDS, chunk c+1's weight fragments (HBM) and
    // activation slice (L2) are already in flight.
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // native vectors: arrays of the HIP uint4 struct end up in scratch (prefetch spilled behind its loads)
    struct Stage { bf16x8 b[G2_KC / 32], b2[G2_KC / 32]; u32x4 sh[PASSES], sl[PASSES]; };
    auto issue = [&](Stage& S, int k0) {
#pragma unroll
        for (int st = 0; st < G2_KC / 32; ++st) {
            S.b[st] = *reinterpret_cast<const bf16x8*>(wp + k0 + st * 32);
            if (DUAL) S.b2[st] = *reinterpret_cast<const bf16x8*>(wp2 + k0 + st * 32);
        }
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            int row = srow + p * RPP;
            row = row < M ? row : M - 1;
            S.sh[p] = *reinterpret_cast<const u32x4*>(pxh + (size_t)row * pldx + k0 + scol);
            S.sl[p] = *reinterpret_cast<const u32x4*>(pxl + (size_t)row * pldx + k0 + scol);
        }
    };
    int pm = 2;   // Q3_SAMPLE_PROF mark cursor
    (void)pm;
    auto consume = [&](Stage& S) {
        __syncthreads(); // previous chunk's fragment reads are done
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int row = srow + p * RPP;
            if (row < ROWS) {
                *reinterpret_cast<u32x4*>(&xs[0][row][scol]) = S.sh[p];
                *reinterpret_cast<u32x4*>(&xs[1][row][scol]) = S.sl[p];
            }
        }
        __syncthreads();
        GP_MARK(EPI == EPI_SLAB2, pm++);   // this chunk's operands have landed and sit in LDS
#pragma unroll
        for (int st = 0; st < G2_KC / 32; ++st) {
#pragma unroll
            for (int mt = 0; mt < MTILES; ++mt) {
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&xs[0][mt * 16 + r16][st * 32 + q * 8]);
                const bf16x8 al = *reinterpret_cast<const bf16x8*>(&xs[1][mt * 16 + r16][st * 32 + q * 8]);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, S.b[st], acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, S.b[st], acc[mt], 0, 0, 0);
                if (DUAL) {
                    acc2[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, S.b2[st], acc2[mt], 0, 0, 0);
                    acc2[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, S.b2[st], acc2[mt], 0, 0, 0);
                }
            }
        }
        GP_MARK(EPI == EPI_SLAB2, pm++);   // its MFMAs are issued
    };
    Stage s0, s1;
    const int kend = kbeg + kslice;
    GP_MARK(EPI == EPI_SLAB2, 0);
    issue(s0, kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += 2 * G2_KC) {
        if (k0 + G2_KC < kend) issue(s1, k0 + G2_KC);
        __builtin_amdgcn_sched_barrier(0);
        GP_MARK(EPI == EPI_SLAB2 && k0 == kbeg, 1);   // both chunks' loads issued
        consume(s0);
        if (k0 + G2_KC < kend) {
            if (k0 + 2 * G2_KC < kend) issue(s0, k0 + 2 * G2_KC);
            __builtin_amdgcn_sched_barrier(0);
            consume(s1);
        }
    }

    const int n = n0 + r16;
    if (n >= a.N) return;
#pragma unroll
    for (int mt = 0; mt < MTILES; ++mt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = mt * 16 + q * 4 + r; // D layout: col = lane&15, row = (lane>>4)*4 + reg
            if (m >= M) continue;
            float o = acc[mt][r];
            if (EPI == EPI_SWIGLU) o = silu_g(o) * acc2[mt][r];
            if (EPI == EPI_SLAB) { a.out[((size_t)blockIdx.y * a.slab_rows + m) * a.ldo + n] = o; continue; }
            if (EPI == EPI_SLAB2) { a.out[((size_t)blockIdx.y * a.slab_rows + m) * a.ldo + n] = o; a.out2[((size_t)blockIdx.y * a.slab_rows + m) * a.ldo + n] = acc2[mt][r]; continue; }
            if (a.out) a.out[(size_t)m * a.ldo + n] = o;
            if (a.oh) split_store(o, a.oh + (size_t)m * a.ldp + n, a.ol + (size_t)m * a.ldp + n);
        }
    }
    GP_MARK(EPI == EPI_SLAB2, 15);   // epilogue stores issued
}

template <int MTILES, int NW>
static void gemm2_epi(const GemmArgs& a, int ksplit, hipStream_t s) {
    const dim3 grid((a.N + NW * 16 - 1) / (NW * 16), ksplit), block(NW * 64);
    switch (a.epi) {
    case EPI_STORE: hipLaunchKernelGGL((k_gemm2<MTILES, EPI_STORE, NW>), grid, block, 0, s, a.W, a.W2, a.xh, a.xl, a.ldx, a.M, a.N, a.K, a); break;
    case EPI_SWIGLU: hipLaunchKernelGGL((k_gemm2<MTILES, EPI_SWIGLU, NW>), grid, block, 0, s, a.W, a.W2, a.xh, a.xl, a.ldx, a.M, a.N, a.K, a); break;
    case EPI_SLAB: hipLaunchKernelGGL((k_gemm2<MTILES, EPI_SLAB, NW>), grid, block, 0, s, a.W, a.W2, a.xh, a.xl, a.ldx, a.M, a.N, a.K, a); break;
    case EPI_SLAB2: hipLaunchKernelGGL((k_gemm2<MTILES, EPI_SLAB2, NW>), grid, block, 0, s, a.W, a.W2, a.xh, a.xl, a.ldx, a.M, a.N, a.K, a); break;
    default: throw Error("gemm2: unsupported epilogue");
    }
}
static bool gemm3_ok(const GemmArgs& a, int ksplit);                       // third generation of the slab GEMM, below
static void launch_gemm3(const GemmArgs& a, int ksplit, hipStream_t s);
static void launch_gemm3_seam(const GemmArgs& a, int ksplit, hipStream_t s);
static GemmArgs gemm3_with_packed(const GemmArgs& a0);
template <int MTILES>
static void gemm2_nw(const GemmArgs& a, int ksplit, int nw, hipStream_t s) {
    if constexpr (MTILES == 8) {   // 128 rows on two-wave workgroups spills (196 bytes per lane) and has no caller: not instantiated
        if (nw != 4) throw Error("gemm2: 65..128 rows run on four-wave workgroups");
        gemm2_epi<MTILES, 4>(a, ksplit, s);
    } else { if (nw == 2) gemm2_epi<MTILES, 2>(a, ksplit, s); else gemm2_epi<MTILES, 4>(a, ksplit, s); }
}
// ksplit: number of K slices (1 = complete sums, direct epilogue; >1 requires EPI_SLAB); nw: waves (= 16-column tiles) per workgroup
void launch_gemm2(const GemmArgs& a0, int ksplit, int nw, hipStream_t s) {
    if (a0.M < 1 || a0.K % (G2_KC * ksplit) != 0 || a0.ldx % 8 != 0) throw Error("gemm2: unsupported shape");
    if (ksplit > 1 && a0.epi != EPI_SLAB && a0.epi != EPI_SLAB2) throw Error("gemm2: split-K needs a slab epilogue");
    // more than 128 rows: 128-row blocks, each streaming the weights again (L2 / MALL resident after the first); slabs keep their
    // [slice][all rows][N] shape, a block writes its rows of every slice
    for (int m0 = 0; m0 < a0.M; m0 += 128) {
        GemmArgs a = a0;
        a.M = std::min(128, a0.M - m0);
        a.slab_rows = a0.slab_rows > 0 ? a0.slab_rows : a0.M;
        a.xh = a0.xh + (size_t)m0 * a0.ldx; a.xl = a0.xl + (size_t)m0 * a0.ldx;
        if (a0.out) a.out = a0.out + (size_t)m0 * a0.ldo;
        if (a0.out2) a.out2 = a0.out2 + (size_t)m0 * a0.ldo;
        if (a0.oh) { a.oh = a0.oh + (size_t)m0 * a0.ldp; a.ol = a0.ol + (size_t)m0 * a0.ldp; }
        if (a0.res) a.res = a0.res + (size_t)m0 * a0.ldres;
        if (a.seam != 0) {   // the caller checked gemm_seam_ok: there is no finish launch behind this GEMM
            if (a0.M > 128 || nw != 4 || !gemm_seam_ok(a, ksplit)) throw Error("gemm: split-K seam requested for a shape it does not cover");
            launch_gemm3_seam(a, ksplit, s);
            continue;
        }
        if (nw == 4 && gemm3_ok(a, ksplit)) { launch_gemm3(a, ksplit, s); continue; }
        if (a.M <= 16) gemm2_nw<1>(a, ksplit, nw, s);
        else if (a.M <= 32) gemm2_nw<2>(a, ksplit, nw, s);
        else if (a.M <= 64) gemm2_nw<4>(a, ksplit, nw, s);
        else gemm2_nw<8>(a, ksplit, nw, s);
    }
}

// ================================================================================================
// k_gemm3 — the split-K slab GEMM of the batched decode step (17..128 rows), third generation: k_gemm2's decomposition (workgroup = 64
// columns x one K slice, (hi, lo) activation planes through LDS, fp32 partial slabs reduced in fixed order by the consumer) with the
// instruction stream laid out by hand where hipcc's schedule had cost microseconds per launch (profiles/r02_gemm3_notes.txt):
//  * straight-line code, no load under a runtime condition: k_gemm2 issued its second K chunk inside `if (k0 + KC < kend)`, and at the
//    join hipcc waits for the FEWEST loads either path could have outstanding — in practice vmcnt(0..4): the first MFMA waited for
//    every byte of the slice (chunk 0 "landed" at 3.0 us of a 5.4 us launch);
//  * the K slice is cut into NCH chunks of 64: weights of the whole slice first (the HBM stream, one latency deep), then the
//    activation chunks in order; chunk c is multiplied while chunks c+1.. are still arriving (in-order vmcnt counts do the rest);
//  * a two-buffer LDS ring with ONE barrier per chunk (a wave that writes chunk c+2 has passed the barrier of chunk c+1, which every
//    wave reaches only after its reads of chunk c);
//  * a chunk's A fragments are all read before its first MFMA, the second k-step's behind the first one's MFMAs (k_gemm2 waited ~100
//    cycles of LDS latency in front of every MFMA pair: 0.75 us per 128-wide chunk for 0.23 us of matrix work);
//  * rows padded by 32 B: every ds_read_b128 lane group hits 16 distinct 16-byte slots (8 bytes of padding left one 2-way conflict);
//  * the 64 x 64 fp32 tile leaves through LDS as 16-byte row-contiguous stores (k_gemm2: sixteen 4-byte stores per lane).
// Same operations on the same operands in the same order as k_gemm2: bit-identical slabs.
// ================================================================================================
#define G3_CH 64
#define G3_LD (G3_CH + 16)
#define G3_LDE 68

#define G3_AUX_SC1 16      // raw-buffer cache policy: sc1 (agent scope: write-through stores, L1-bypassing loads)
// SEAM / NS: the split-K seam (GemmArgs::seam), NS = slab registers of the reducer (>= the launch's K slices)
template <int MTILES, int EPI, int NCH, int LA, bool NT, int SEAM = 0, int NS = 1>
__global__ __launch_bounds__(256) void k_gemm3(const bf16_t* pW, const bf16_t* pW2, const bf16_t* pxh, const bf16_t* pxl, int pldx, int pM, int pN, int pK,
                                                GemmArgs a) {   // leading scalars: kernarg-preloaded
    static_assert(EPI == EPI_SLAB || EPI == EPI_SLAB2, "k_gemm3 writes split-K slabs");
    constexpr bool DUAL = EPI == EPI_SLAB2;
    constexpr int ROWS = MTILES * 16, RPP = 32, PASSES = (ROWS + RPP - 1) / RPP;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, q = lane >> 4;
    const int n0 = (blockIdx.x * 4 + wave) * 16;
    const int K = pK, M = pM;
    const int kbeg = blockIdx.y * (NCH * G3_CH);
    if constexpr (SEAM != 0) SP_SEAM(0);
    // seam launches: the grid's extents come from the preloaded K / N instead of the hidden kernel arguments (a scalar-cache miss behind the body)
    const unsigned grid_ks = SEAM != 0 ? (unsigned)(pK / (NCH * G3_CH)) : gridDim.y, grid_nt = SEAM != 0 ? (unsigned)(pN / 64) : gridDim.x;
    const int m0 = blockIdx.z * (MTILES * 16);   // row block (grid z): the rows may be cut over workgroups to halve each one's activation ingest
    // ring of two chunk buffers [plane][row][k]; the epilogue reuses the same bytes as a [row][64 + 4] fp32 tile (x 2 for the dual product)
    constexpr int XS_BYTES = 2 * 2 * ROWS * G3_LD * 2, EP_BYTES = (DUAL ? 2 : 1) * ROWS * G3_LDE * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[XS_BYTES > EP_BYTES ? XS_BYTES : EP_BYTES];
    bf16_t (*xs)[2][ROWS][G3_LD] = reinterpret_cast<bf16_t (*)[2][ROWS][G3_LD]>(smem);

    // 1. every weight fragment of the slice: lane (r16, q) = weight row n0 + r16, k = 8 q .. 8 q + 7 of each 32-wide k-step
    int nrow = n0 + r16;
    nrow = nrow < pN ? nrow : pN - 1;
    // a.w_packed: pW / pW2 are the fragment-packed copies (q3_common.h): this wave's tile, k-step s = 1 KB at ((tile K/32 + s) 64 + lane) 8
    const bool wpk = a.w_packed;
    const int wstep = wpk ? 512 : 32;                                 // elements between consecutive k-steps of this lane's fragment
    const size_t woff = wpk ? ((size_t)(blockIdx.x * 4 + wave) * (K >> 5) + (kbeg >> 5)) * 512 + (size_t)lane * 8 : (size_t)nrow * K + kbeg + q * 8;
    const bf16_t* wp = pW + woff;
    const bf16_t* wp2 = DUAL ? pW2 + woff : nullptr;
    bf16x8 b[NCH][2], b2[DUAL ? NCH : 1][2];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            // NT: weights one launch reads once per step (the talker's 887 MB) leave no footprint in L2 / the Infinity Cache, so the
            // predictor's 161 MB, re-read 15 times per frame, stay resident there (MI355X guide, nt-weights)
            if (NT) {
                b[c][st] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp + (c * 2 + st) * wstep));
                if (DUAL) b2[DUAL ? c : 0][st] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp2 + (c * 2 + st) * wstep));
            } else {
                b[c][st] = *reinterpret_cast<const bf16x8*>(wp + (c * 2 + st) * wstep);
                if (DUAL) b2[DUAL ? c : 0][st] = *reinterpret_cast<const bf16x8*>(wp2 + (c * 2 + st) * wstep);
            }
        }
    // 2. activation chunks: 8 threads per row (16 B each), 32 rows per pass; rows past M repeat row M - 1 (their products are never stored)
    const int srow = tid >> 3, scol = (tid & 7) * 8;
    u32x4 sh[NCH][PASSES], sl[NCH][PASSES];
    auto issue = [&](int c) {
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            int row = m0 + srow + p * RPP;
            row = row < M ? row : M - 1;
            sh[c][p] = *reinterpret_cast<const u32x4*>(pxh + (size_t)row * pldx + kbeg + c * G3_CH + scol);
            sl[c][p] = *reinterpret_cast<const u32x4*>(pxl + (size_t)row * pldx + kbeg + c * G3_CH + scol);
        }
    };
#pragma unroll
    for (int c = 0; c < LA; ++c) issue(c);
    __builtin_amdgcn_sched_barrier(0);   // the loads above stay above everything below, in this order
    // seam launches: the step's generation word is requested HERE, behind the weight and activation requests and long before its use (read
    // at the seam it was a dependent memory round trip, ~1 us, between the body and the flag store)
    unsigned seam_gen_v = 0;
    if constexpr (SEAM != 0) seam_gen_v = __builtin_amdgcn_readfirstlane(*a.seam_gen);
    const int seam_spin_v = a.seam_spin;

    f32x4 acc[MTILES], acc2[DUAL ? MTILES : 1];
#pragma unroll
    for (int mt = 0; mt < MTILES; ++mt) { acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f}; if (DUAL) acc2[DUAL ? mt : 0] = f32x4{0.f, 0.f, 0.f, 0.f}; }

#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int buf = c & 1;
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int row = srow + p * RPP;
            if (ROWS % RPP == 0 || row < ROWS) {
                *reinterpret_cast<u32x4*>(&xs[buf][0][row][scol]) = sh[c][p];
                *reinterpret_cast<u32x4*>(&xs[buf][1][row][scol]) = sl[c][p];
            }
        }
        if (c + LA < NCH) issue(c + LA);     // compile-time condition: the freed registers take a later chunk
        __syncthreads();
        // both k-steps' A fragments are read before the first MFMA (pinned: left alone, hipcc re-reads one register set per MFMA pair and
        // waits ~100 cycles of LDS latency in front of each); the second k-step's reads land behind the first one's MFMAs
        bf16x8 fh[2][MTILES], fl[2][MTILES];
#pragma unroll
        for (int st = 0; st < 2; ++st) {
#pragma unroll
            for (int mt = 0; mt < MTILES; ++mt) {
                fh[st][mt] = *reinterpret_cast<const bf16x8*>(&xs[buf][0][mt * 16 + r16][st * 32 + q * 8]);
                fl[st][mt] = *reinterpret_cast<const bf16x8*>(&xs[buf][1][mt * 16 + r16][st * 32 + q * 8]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
#pragma unroll
            for (int mt = 0; mt < MTILES; ++mt) {
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh[st][mt], b[c][st], acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fl[st][mt], b[c][st], acc[mt], 0, 0, 0);
                if (DUAL) {
                    acc2[DUAL ? mt : 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh[st][mt], b2[DUAL ? c : 0][st], acc2[DUAL ? mt : 0], 0, 0, 0);
                    acc2[DUAL ? mt : 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fl[st][mt], b2[DUAL ? c : 0][st], acc2[DUAL ? mt : 0], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // 3. the tile leaves through LDS: D layout (col = lane & 15, row = 4 q + reg) -> [row][64 columns] -> one 16-byte store per thread and row pass
    __syncthreads();   // the last chunk's fragment reads are done
    float (*ep)[ROWS][G3_LDE] = reinterpret_cast<float (*)[ROWS][G3_LDE]>(smem);
#pragma unroll
    for (int mt = 0; mt < MTILES; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            ep[0][mt * 16 + q * 4 + r][wave * 16 + r16] = acc[mt][r];
            if (DUAL) ep[DUAL ? 1 : 0][mt * 16 + q * 4 + r][wave * 16 + r16] = acc2[DUAL ? mt : 0][r];
        }
    __syncthreads();
    const int erow = tid >> 4, ecol = (tid & 15) * 4, ng = blockIdx.x * 64 + ecol;
    const size_t sbase = (size_t)blockIdx.y * a.slab_rows;
    if constexpr (SEAM != 0) {
        // ---- split-K seam (tools/microbench_seam.hip, variant C; MI355X guide, hand-off table row 1).  Slab tiles leave write-through
        // (sc1), every wave drains its stores, the workgroup meets, ONE lane takes a ticket on the tile's counter.  Whoever finds the tile
        // complete — the last arriver at once, an earlier one within a bounded wait — claims 16-row chunks of the reduction and sums the
        // K slices in slab order (bit-reproducible) with sc1 loads.  An arriver whose wait runs out just leaves: the last arriver claims
        // whatever is left, so every chunk is reduced exactly once whatever the placement or residency of the workgroups.
        __shared__ unsigned seam_flag;
        SP_SEAM(1);   // body done, tile in LDS
        const size_t slab_bytes = (size_t)grid_ks * a.slab_rows * a.ldo * sizeof(float);
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)slab_bytes, 0x00020000);
        __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(DUAL ? a.out2 : a.out, 0, (int)slab_bytes, 0x00020000);
#pragma unroll
        for (int p = 0; p < ROWS / 16; ++p) {
            const int ml = erow + p * 16, m = m0 + ml;
            const unsigned off = (unsigned)(((sbase + (m < M ? m : M - 1)) * a.ldo + ng) * sizeof(float));   // rows past M rewrite row M - 1 of this slice with its own value
            const int mls = m < M ? ml : M - 1 - m0;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, *reinterpret_cast<const f32x4*>(&ep[0][mls][ecol])), rs, off, 0, G3_AUX_SC1);
            if (DUAL) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, *reinterpret_cast<const f32x4*>(&ep[DUAL ? 1 : 0][mls][ecol])), rs2, off, 0, G3_AUX_SC1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        SP_SEAM(2);   // slab tile written through and drained
        // Second protocol (the first one — arrival ticket + claim counter, two RETURNING atomics of 0.7-2 us each on the critical path —
        // lost to the finish launch: profiles/r03_negative_results.txt item 1).  No returning atomic on the fast path: every slice sets its
        // word of the unit's 64-byte line (sc1 store behind the drain); slice s < NCHK OWNS chunk s and polls the line (one sc1 load per
        // poll) until all KS words are set, then reduces.  Liveness does not rest on co-residency: an owner whose bounded wait runs out
        // marks its chunk abandoned (words 12..15) and leaves; whoever later observes the line complete — the late slice itself, after
        // its own flag store — takes abandoned chunks by compare-and-swap (1 -> 2), the owner included after one last look (the two
        // orders of {mark, last flag} are both covered because each side's store is acknowledged before its load is issued).
        const unsigned KS = grid_ks;
        constexpr int NCHK = ROWS / 16;
        unsigned* line = a.seam_cnt + ((size_t)blockIdx.z * grid_nt + blockIdx.x) * 16;
        const int mine = (int)blockIdx.y < NCHK ? (int)blockIdx.y : -1;
        // every word carries the step's generation: flag set = 4 gen + 3, chunk abandoned = 4 gen + 1, taken = 4 gen + 2.  Nothing is reset
        // between steps; whatever an earlier step (or the allocation) left in the line never equals this step's values.
        const unsigned gbase = seam_gen_v << 2;
        if (wave == 0) {
            if (lane == 0) __hip_atomic_store(line + blockIdx.y, gbase | 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // Everybody looks only AFTER its own flag store is acknowledged (the rescue protocol's store -> load order: of two slices that
            // each miss the other's last word, one must have looked before its own store landed).  Round 3 let the chunk owners poll at once;
            // they are the low-y workgroups, dispatched first and usually waiting for the later slices anyway, so the acknowledgement is
            // hidden: b=64 step 4.586 / 4.607 ms with the wait against 4.637 / 4.603 without (same box, profiles/r04_negative_results.txt).
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            auto look = [&](unsigned& marks) -> bool {   // one sc1 load of the line: all slices in? which chunks are abandoned?
                const unsigned v = __hip_atomic_load(line + (lane & 15), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long in = __ballot((lane < (int)KS && lane != (int)blockIdx.y) ? v == (gbase | 3u) : true);
                const unsigned long long ab = __ballot(lane >= 12 && lane < 16 && v == (gbase | 1u));
                marks = (unsigned)(ab >> 12) & 0xFu;
                return in == ~0ull;
            };
            unsigned take = 0, marks = 0;     // bit c: this workgroup reduces chunk c
            bool complete = false;
            const int spins = mine >= 0 ? seam_spin_v : 1;
            for (int i = 0; i < spins && !complete; ++i) complete = look(marks);
            if (mine >= 0) {
                if (complete) take |= 1u << mine;
                else {
                    if (lane == 0) __hip_atomic_exchange(line + 12 + mine, gbase | 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    complete = look(marks);
                }
            }
            if (complete && marks) {          // rescue abandoned chunks (never on a chip that runs the whole grid at once)
                for (int c = 0; c < NCHK; ++c)
                    if (marks >> c & 1u) {
                        unsigned won = 0;
                        if (lane == 0) { unsigned expect = gbase | 1u; won = __hip_atomic_compare_exchange_strong(line + 12 + c, &expect, gbase | 2u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1u : 0u; }
                        won = __builtin_amdgcn_readfirstlane(won);
                        if (won) take |= 1u << c;
                    }
            }
            if (lane == 0) seam_flag = take;
        }
        __syncthreads();
        SP_SEAM(3);   // tile seen complete (or the wait ran out)
        const unsigned take_all = seam_flag;
        if (take_all == 0) return;
        const size_t slab_stride = (size_t)a.slab_rows * a.ldo * sizeof(float);
        for (int chunk = 0; chunk < NCHK; ++chunk) {
            if (!(take_all >> chunk & 1u)) continue;
            SP_SEAM(4);
            const int mr = m0 + chunk * 16 + erow;
            const bool live = mr < M;
            const int m = live ? mr : M - 1;                       // clamped: loads stay unconditional, stores are skipped
            const unsigned off0 = (unsigned)(((size_t)m * a.ldo + ng) * sizeof(float));
            u32x4 pp[NS], qq[DUAL ? NS : 1];
#pragma unroll
            for (int sidx = 0; sidx < NS; ++sidx) {
                const unsigned so = off0 + (unsigned)((sidx < (int)KS ? sidx : (int)KS - 1) * slab_stride);
                pp[sidx] = __builtin_amdgcn_raw_buffer_load_b128(rs, so, 0, G3_AUX_SC1);
                if (DUAL) qq[DUAL ? sidx : 0] = __builtin_amdgcn_raw_buffer_load_b128(rs2, so, 0, G3_AUX_SC1);
            }
            if constexpr (SEAM == 1) {
                float* xr = a.sx + (size_t)m * a.sldx + ng;
                f32x4 t = *reinterpret_cast<const f32x4*>(xr);
                const f32x4 g = *reinterpret_cast<const f32x4*>(a.sgamma + ng);
#pragma unroll
                for (int sidx = 0; sidx < NS; ++sidx) if (sidx < (int)KS) t += __builtin_bit_cast(f32x4, pp[sidx]);
                float ss = 0.f;
                ss = fmaf(t.x, t.x, ss); ss = fmaf(t.y, t.y, ss); ss = fmaf(t.z, t.z, ss); ss = fmaf(t.w, t.w, ss);
                // the row's 16 lanes (64 columns) in a fixed butterfly order: deterministic
                ss += __shfl_xor(ss, 1, 16); ss += __shfl_xor(ss, 2, 16); ss += __shfl_xor(ss, 4, 16); ss += __shfl_xor(ss, 8, 16);
                if (live) {
                    *reinterpret_cast<f32x4*>(xr) = t;
                    const float y[4] = { g.x * t.x, g.y * t.y, g.z * t.z, g.w * t.w };
                    split_store4(y, a.oh + (size_t)m * a.ldp + ng, a.ol + (size_t)m * a.ldp + ng);
                    if ((tid & 15) == 0) a.ssq_out[(size_t)m * a.ssq_nt + blockIdx.x] = ss;
                }
            } else {
                // row scale of the input planes: 1 / rms from the producer's per-tile partials, summed in tile order
                float rsc = 1.0f;
                if (a.ssq_in != nullptr) {
                    const float* sq = a.ssq_in + (size_t)m * a.ssq_in_nt;
                    float tot = 0.f;
                    for (int t4 = 0; t4 < a.ssq_in_nt; t4 += 4) { const f32x4 v = *reinterpret_cast<const f32x4*>(sq + t4); tot += v.x; tot += v.y; tot += v.z; tot += v.w; }
                    rsc = 1.0f / sqrtf(tot / (float)K + a.seps);
                }
                f32x4 gs = { 0.f, 0.f, 0.f, 0.f }, us = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
                for (int sidx = 0; sidx < NS; ++sidx) if (sidx < (int)KS) { gs += __builtin_bit_cast(f32x4, pp[sidx]); us += __builtin_bit_cast(f32x4, qq[DUAL ? sidx : 0]); }
                gs *= rsc; us *= rsc;
                if (live) {
                    const float o[4] = { silu_g(gs.x) * us.x, silu_g(gs.y) * us.y, silu_g(gs.z) * us.z, silu_g(gs.w) * us.w };
                    split_store4(o, a.oh + (size_t)m * a.ldp + ng, a.ol + (size_t)m * a.ldp + ng);
                }
            }
            SP_SEAM(5);   // the chunk's sums are stored
        }
        return;
    }
    // Slabs are read by ANOTHER launch (the attention prologue, k_finish*, the slab sampler), in general on other XCDs: stored write-through
    // (sc1) they reach the fabric while the launch still runs; as plain stores they sit dirty in this XCD's L2 until the kernel boundary
    // writes them back (MI355X guide, boundary row: + B / 6 TB/s behind B dirty bytes — 0.7 us behind the QKV projection's 4.2 MB at 64
    // rows).  Q3TTS_GEMM_PLAIN_SLABS=1 (a.plain_slabs) is the A/B knob.
    if (!a.plain_slabs && (a.N & 3) == 0) {
        const size_t slab_bytes = (size_t)gridDim.y * a.slab_rows * a.ldo * sizeof(float);
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)slab_bytes, 0x00020000);
        __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(DUAL ? a.out2 : a.out, 0, (int)slab_bytes, 0x00020000);
#pragma unroll
        for (int p = 0; p < ROWS / 16; ++p) {
            const int ml = erow + p * 16, m = m0 + ml;
            if (m < M && ng < a.N) {
                const unsigned off = (unsigned)(((sbase + m) * a.ldo + ng) * sizeof(float));
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, *reinterpret_cast<const f32x4*>(&ep[0][ml][ecol])), rs, off, 0, G3_AUX_SC1);
                if (DUAL) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, *reinterpret_cast<const f32x4*>(&ep[DUAL ? 1 : 0][ml][ecol])), rs2, off, 0, G3_AUX_SC1);
            }
        }
        return;
    }
#pragma unroll
    for (int p = 0; p < ROWS / 16; ++p) {
        const int ml = erow + p * 16, m = m0 + ml;
        if (m < M && ng < a.N) {
            const float4 v = *reinterpret_cast<const float4*>(&ep[0][ml][ecol]);
            float* dst = a.out + (sbase + m) * a.ldo + ng;
            if (ng + 3 < a.N) *reinterpret_cast<float4*>(dst) = v;
            else { dst[0] = v.x; if (ng + 1 < a.N) dst[1] = v.y; if (ng + 2 < a.N) dst[2] = v.z; }
            if (DUAL) {
                const float4 v2 = *reinterpret_cast<const float4*>(&ep[DUAL ? 1 : 0][ml][ecol]);
                float* dst2 = a.out2 + (sbase + m) * a.ldo + ng;
                if (ng + 3 < a.N) *reinterpret_cast<float4*>(dst2) = v2;
                else { dst2[0] = v2.x; if (ng + 1 < a.N) dst2[1] = v2.y; if (ng + 2 < a.N) dst2[2] = v2.z; }
            }
        }
    }
}

// seam launches: the shapes the batched decode step produces (64 / 128 rows -> 32- / 64-row blocks, K slices of 256, N a multiple of 64)
template <int MTILES, int EPI, int SEAM, int NS>
static void gemm3_seam_go(const GemmArgs& a, int ksplit, hipStream_t s) {
    const dim3 grid(a.N / 64, ksplit, (a.M + MTILES * 16 - 1) / (MTILES * 16)), block(256);
    const int nch = a.K / ksplit / G3_CH;
    if constexpr (SEAM == 2 && NS == 4 && MTILES == 2) {   // Q3TTS_SEAM_GU_KS=2 (A/B knob): 512-wide K slices for gate/up — two slab pairs per seam instead of four
        if (nch == 8) { hipLaunchKernelGGL((k_gemm3<MTILES, EPI, 8, 2, false, SEAM, NS>), grid, block, 0, s, a.W, a.W2, a.xh, a.xl, a.ldx, a.M, a.N, a.K, a); return; }
    }
    if (nch == 4) hipLaunchKernelGGL((k_gemm3<MTILES, EPI, 4, 2, false, SEAM, NS>), grid, block, 0, s, a.W, a.W2, a.xh, a.xl, a.ldx, a.M, a.N, a.K, a);
    else hipLaunchKernelGGL((k_gemm3<MTILES, EPI, 2, 2, false, SEAM, NS>), grid, block, 0, s, a.W, a.W2, a.xh, a.xl, a.ldx, a.M, a.N, a.K, a);
}
bool gemm_seam_ok(const GemmArgs& a, int ksplit) {
    if (a.seam != 1 && a.seam != 2) return false;
    if ((a.seam == 1) != (a.epi == EPI_SLAB) || (a.seam == 2) != (a.epi == EPI_SLAB2)) return false;
    const int ksl = ksplit > 0 && a.K % ksplit == 0 ? a.K / ksplit : 0;
    if (!(ksl == 128 || ksl == 256 || (ksl == 512 && a.seam == 2 && a.M <= 64)) || a.N % 64 != 0 || a.ldo % 4 != 0 || a.ldx % 8 != 0 || a.M < 1 || a.M > 128) return false;
    // chunk c of a row block is reduced by K slice c (blockIdx.y == c): a launch with fewer slices than 16-row chunks per block (2 up to 64
    // rows, 4 beyond) would leave chunks without an owner — those shapes (K = 128 .. 768 with 256-wide slices) keep the finish launches
    if (ksplit < (a.M > 64 ? 4 : 2)) return false;
    if (a.seam == 1 && (ksplit > 12 || !a.sx || !a.sgamma || !a.ssq_out || a.ssq_nt != a.N / 64 || a.sldx % 4 != 0)) return false;
    if (a.seam == 2 && (ksplit > 8 || (a.ssq_in && a.ssq_in_nt % 4 != 0))) return false;
    return a.seam_cnt != nullptr && a.seam_gen != nullptr && a.oh != nullptr && a.ol != nullptr && a.ldp % 4 == 0 && (a.slab_rows == 0 || a.slab_rows == a.M);
}
static void launch_gemm3_seam(const GemmArgs& a0, int ksplit, hipStream_t s) {
    const GemmArgs a = gemm3_with_packed(a0);
    // 32-row blocks up to 64 rows, 64-row blocks beyond.  Q3TTS_SEAM_BIG (A/B knob, bit 0: the residual seams, bit 1: gate/up): 64-row
    // blocks from 33 rows on — half the workgroups, the weights fetched once instead of once per row block
    const int bigk = knob("Q3TTS_SEAM_BIG") ? atoi(knob("Q3TTS_SEAM_BIG")) : 0;
    const bool big = a.M > 64 || (a.M > 32 && ksplit >= 4 && ((bigk >> (a.seam - 1)) & 1));
    if (a.seam == 1) {
        if (ksplit <= 8) { if (big) gemm3_seam_go<4, EPI_SLAB, 1, 8>(a, ksplit, s); else gemm3_seam_go<2, EPI_SLAB, 1, 8>(a, ksplit, s); }
        else { if (big) gemm3_seam_go<4, EPI_SLAB, 1, 12>(a, ksplit, s); else gemm3_seam_go<2, EPI_SLAB, 1, 12>(a, ksplit, s); }
    } else {
        if (ksplit <= 4) { if (big) gemm3_seam_go<4, EPI_SLAB2, 2, 4>(a, ksplit, s); else gemm3_seam_go<2, EPI_SLAB2, 2, 4>(a, ksplit, s); }
        else { if (big) gemm3_seam_go<4, EPI_SLAB2, 2, 8>(a, ksplit, s); else gemm3_seam_go<2, EPI_SLAB2, 2, 8>(a, ksplit, s); }
    }
}

template <int MTILES, int EPI, bool NT>
static void gemm3_go(const GemmArgs& a, int ksplit, hipStream_t s) {
    const dim3 grid((a.N + 63) / 64, ksplit, (a.M + MTILES * 16 - 1) / (MTILES * 16)), block(256);
    const int nch = a.K / ksplit / G3_CH;
    constexpr int LA4 = MTILES <= 4 ? 4 : 2;
    // activation chunks two ahead (default) or all four at once (Q3TTS_GEMM3_LA=4, the A/B knob): issuing 24-32 loads per lane before the
    // first ds_write keeps the wave in its issue queue for ~2.5 us (the CU takes ~50 GB/s); two ahead measured 5.49 vs 5.63 ms per b=64 step
    const bool la2 = !(knob("Q3TTS_GEMM3_LA") && atoi(knob("Q3TTS_GEMM3_LA")) == 4);
    if (nch == 4 && la2) { hipLaunchKernelGGL((k_gemm3<MTILES, EPI, 4, 2, NT>), grid, block, 0, s, a.W, a.W2, a.xh, a.xl, a.ldx, a.M, a.N, a.K, a); return; }
    if (nch == 2) hipLaunchKernelGGL((k_gemm3<MTILES, EPI, 2, 2, NT>), grid, block, 0, s, a.W, a.W2, a.xh, a.xl, a.ldx, a.M, a.N, a.K, a);
    else hipLaunchKernelGGL((k_gemm3<MTILES, EPI, 4, LA4, NT>), grid, block, 0, s, a.W, a.W2, a.xh, a.xl, a.ldx, a.M, a.N, a.K, a);
}
// the shapes k_gemm3 is built for: slab epilogues, K slices of 128 or 256, 16-byte aligned slab rows
static bool gemm3_ok(const GemmArgs& a, int ksplit) {
    if (a.epi != EPI_SLAB && a.epi != EPI_SLAB2) return false;
    if (knob("Q3TTS_GEMM2")) return false;   // A/B knob: the second-generation kernel
    const int ksl = a.K / ksplit;
    return a.K % ksplit == 0 && (ksl == 128 || ksl == 256) && a.ldo % 4 == 0 && a.ldx % 8 == 0 && a.M >= 1 && a.M <= 128;
}
// the fragment-packed copies of this launch's weights, when the engine registered them (N a multiple of 64: every column tile of the grid
// then exists in the packed buffer; both matrices of a dual launch, or neither)
static GemmArgs gemm3_with_packed(const GemmArgs& a0) {
    GemmArgs a = a0;
    if (a.N % 64 != 0 || a.K % 32 != 0) return a;
    const bf16_t* Wp = find_packed_weight(a.W);
    const bf16_t* W2p = a.W2 ? find_packed_weight(a.W2) : nullptr;
    if (Wp != nullptr && (a.W2 == nullptr || W2p != nullptr)) { a.W = Wp; if (a.W2) a.W2 = W2p; a.w_packed = true; }
    return a;
}
static void launch_gemm3(const GemmArgs& a0, int ksplit, hipStream_t s) {
    GemmArgs a = gemm3_with_packed(a0);
    a.plain_slabs = knob("Q3TTS_GEMM_PLAIN_SLABS") != nullptr;
    const bool dual = a.epi == EPI_SLAB2;
    // nt weight loads measured on the b=64 step (graph replay, same box): 4.961 ms with, 4.917 ms without — the slab GEMM's launches are
    // bound by their latency chain, not by where the weights come from, and a replayed GEMM body loses what default-policy loads leave in
    // L2 / MALL (MI355X guide, nt-weights: "replayed back to back 0-34 % longer").  Off by default; Q3TTS_GEMM_NT=1 is the A/B knob.
    const bool want_nt = knob("Q3TTS_GEMM_NT") != nullptr;
    const bool nt = a.nt && want_nt;
#define Q3_G3(MT) do { if (dual) { if (nt) gemm3_go<MT, EPI_SLAB2, true>(a, ksplit, s); else gemm3_go<MT, EPI_SLAB2, false>(a, ksplit, s); } \
                       else { if (nt) gemm3_go<MT, EPI_SLAB, true>(a, ksplit, s); else gemm3_go<MT, EPI_SLAB, false>(a, ksplit, s); } } while (0)
    // rows over workgroups (grid z): a 64-row workgroup ingests 32 KB of weights + 64 KB of planes; two 32-row workgroups ingest 32 + 32 KB
    // each (the second one's weights are L2 hits: same XCD under round-robin placement since N / 64 is a multiple of 8) and there are
    // twice as many of them.  Q3TTS_GEMM_ROWSPLIT=0 keeps one workgroup per column tile and K slice (the A/B knob).
    const int rowsplit = knob("Q3TTS_GEMM_ROWSPLIT") ? atoi(knob("Q3TTS_GEMM_ROWSPLIT")) : 1;
    if (rowsplit && a.M > 32) { if (a.M <= 64) Q3_G3(2); else Q3_G3(4); return; }
    if (a.M <= 16) Q3_G3(1); else if (a.M <= 32) Q3_G3(2); else if (a.M <= 64) Q3_G3(4); else Q3_G3(8);
#undef Q3_G3
}

// ================================================================================================
// k_gemv16 — 3..16 activation rows, the GEMV family's contract (fp32 rows in, fused RMSNorm, fused epilogue, ONE launch per
// projection) on the matrix cores.  Workgroup = one 16-column tile of N over the full K; its NWV waves split K and meet in LDS in
// wave order (deterministic).  A wave issues every weight fragment of its K slice before anything else (HBM stream, one latency
// deep), then walks its slice of the fp32 activation rows (L2) in batches of AB k-steps: gamma, sum of squares, split into
// (hi, lo) bf16 with the hardware RNE convert, two MFMAs per k-step.  The row scale 1/rms is applied to the finished sums
// (sum_k (x*gamma)*W * inv == sum_k ((x*inv)*gamma)*W up to fp32 rounding), so no pass over x precedes the products and the
// split-K slabs + finish kernels of k_gemm2 (3 extra launches per layer) are not needed at these row counts.
// ================================================================================================
typedef unsigned g16_u32x4 __attribute__((ext_vector_type(4)));
typedef float g16_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 g16_bf16x2 __attribute__((ext_vector_type(2)));
typedef float g16_f32x2 __attribute__((ext_vector_type(2)));

static __device__ __forceinline__ g16_u32x4 g16_ldw(const bf16_t* p, bool nt) {
    if (nt) return __builtin_nontemporal_load(reinterpret_cast<const g16_u32x4*>(p));
    return *reinterpret_cast<const g16_u32x4*>(p);
}
// 8 fp32 -> hi and lo planes (4 dwords each); v - hi is exact in fp32
static __device__ __forceinline__ void g16_split8(const float (&y)[8], bf16x8& hi, bf16x8& lo) {
    g16_u32x4 h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const g16_f32x2 v = { y[2 * j], y[2 * j + 1] };
        const uint32_t hp = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, g16_bf16x2));
        const g16_f32x2 r = { y[2 * j] - __uint_as_float(hp << 16), y[2 * j + 1] - __uint_as_float(hp & 0xFFFF0000u) };
        h[j] = hp;
        l[j] = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, g16_bf16x2));
    }
    hi = __builtin_bit_cast(bf16x8, h);
    lo = __builtin_bit_cast(bf16x8, l);
}

// R8 (M <= 8): the activation rows arrive as 8 rows x 128 bytes per load instruction — lane (r8 = lane & 7, c = lane >> 3) takes floats
// [4c, 4c + 4) of row r8's 32-float k-step — instead of 16 rows x 2 x 64 bytes with the lanes of rows 8..15 switched off.  The
// launch is bound by the ISSUE of its loads (3-5 us of 6.8: profiles/r03_gemv16_phase_stamps.txt; the CU's address path takes one
// 128-byte line per clock and the operand layout made every instruction touch 8-16 lines half used): one instruction per k-step
// instead of two, every lane active, every line touched once.  Lane (r16 < 8, q) of the MFMA layout already holds its floats
// [8q, 8q + 4) (its own load: c = 2q) and takes [8q + 4, 8q + 8) from lane + 8 with one row_ror:8 DPP move per dword; lanes of
// rows 8..15 carry other rows' chunks into the product, which only reaches output rows 8..15 (never stored).  The values each
// product sees and their order are unchanged: results are bit-identical to the R8 = false kernel.
#define G16_AB 4
// GL (NORM kernels): the RMSNorm gains of a wave's K slice arrive as ONE (or two) whole-line load instructions into the wave's own LDS
// slice and reach the lanes through ds_read (broadcast over the 16 rows) instead of two 16-byte global loads per k-step and lane — 16
// of a K = 1024 wave's 40 load instructions were gains, every row's lanes asking for the same bytes.  Same values: bit-identical.
template <int KWMAX, int NWV, int EPI, bool NORM, bool R8 = false, bool GL = false>
__global__ __launch_bounds__(NWV * 64) void k_gemv16(const bf16_t* pW, const bf16_t* pW2, const float* px, const float* pgamma, const float* pepi,
                                                      int pN, int pM, int pldx, int pldepi, uint32_t pKnt /* K | nt << 31 */, GemvArgs a) {   // leading scalars: kernarg-preloaded
    constexpr bool DUAL = EPI == EPI_SWIGLU;
    constexpr int NG = (KWMAX + G16_AB - 1) / G16_AB;       // groups of AB k-steps
    constexpr int PREG = NG < (NORM ? 2 : 3) ? NG : (NORM ? 2 : 3);   // groups whose activation loads are issued BEFORE the weights
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, q = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int K = (int)(pKnt & 0x3FFFFFFFu), M = pM, N = pN;
    const bool nt = (pKnt >> 31) != 0, wpk = ((pKnt >> 30) & 1u) != 0;
    const int kslice = K / NWV, kw = kslice >> 5;   // k-steps of 32 this wave owns (1..KWMAX)
    const int kbeg = wave * kslice;
#ifdef Q3_SAMPLE_PROF
#define G16_MARK(k) do { if (KWMAX == 12 && EPI == EPI_RESIDUAL && blockIdx.x == 0 && (tid == 0 || tid == NWV * 64 - 64)) g_gemm_prof[(tid == 0 ? 16 : 24) + (k)] = wall_clock64(); } while (0)
#else
#define G16_MARK(k) do { } while (0)
#endif
    G16_MARK(0);
    __shared__ float red[DUAL ? 2 : 1][NWV][16][17];
    __shared__ float ssq[NWV][16];
    __shared__ __attribute__((aligned(16))) float gam_s[(NORM && GL) ? NWV : 1][(NORM && GL) ? KWMAX * 32 : 4];

    // Loads return in issue order, so what is issued before the weight stream can be converted while the weights are in flight:
    // the first PREG groups of activation k-steps go first, the rest follow the weights one group at a time into a freed buffer.
    const int r8 = lane & 7, c8 = lane >> 3;
    const float* xr = R8 ? px + (size_t)(r8 < M ? r8 : M - 1) * pldx + kbeg + c8 * 4     // lane (r8, c): row r8, floats [4c, 4c + 4) of each k-step
                         : px + (size_t)(r16 < M ? r16 : M - 1) * pldx + kbeg + q * 8;   // lane (r16, q): row r16, 8 consecutive k per k-step
    const float* gr = NORM ? pgamma + kbeg + q * 8 : nullptr;
    g16_f32x4 xa[PREG][G16_AB][R8 ? 1 : 2], ga[(NORM && !GL) ? PREG : 1][G16_AB][2];
    // GL: the slice's gains, 4 floats per lane and pass (a pass = 256 floats = one 1 KB load instruction), requested before everything else
    constexpr int GPASS = (KWMAX * 32 + 255) / 256;
    g16_f32x4 gq[(NORM && GL) ? GPASS : 1];
    if constexpr (NORM && GL) {
#pragma unroll
        for (int gp = 0; gp < GPASS; ++gp) {
            const int o = gp * 256 + lane * 4;
            gq[gp] = *reinterpret_cast<const g16_f32x4*>(pgamma + kbeg + (o < kslice ? o : kslice - 4));
        }
    }
    const bool row_live = R8 ? r8 < M : r16 < M;   // lanes of rows past M issue no activation loads (their outputs are never stored)
    auto issue = [&](int buf, int g) {
#pragma unroll
        for (int j = 0; j < G16_AB; ++j) {
            const int ks = g * G16_AB + j;
            const int kk = ks < kw ? ks : kw - 1;   // k-steps past kw repeat the last one; their fragment is zeroed below
            xa[buf][j][0] = g16_f32x4{ 0.f, 0.f, 0.f, 0.f };
            if (!R8) xa[buf][j][R8 ? 0 : 1] = g16_f32x4{ 0.f, 0.f, 0.f, 0.f };
            if (row_live) {
                xa[buf][j][0] = *reinterpret_cast<const g16_f32x4*>(xr + kk * 32);
                if (!R8) xa[buf][j][R8 ? 0 : 1] = *reinterpret_cast<const g16_f32x4*>(xr + kk * 32 + 4);
            }
            if (NORM && !GL) {
                ga[GL ? 0 : buf][j][0] = *reinterpret_cast<const g16_f32x4*>(gr + kk * 32);
                ga[GL ? 0 : buf][j][1] = *reinterpret_cast<const g16_f32x4*>(gr + kk * 32 + 4);
            }
        }
    };
#pragma unroll
    for (int g = 0; g < PREG; ++g) issue(g, g);
    __builtin_amdgcn_sched_barrier(0);
    // weights: the whole K slice of this wave's 16 rows in flight at once
    const int nrow = n0 + r16 < N ? n0 + r16 : N - 1;
    // bit 30 of pKnt: pW / pW2 are the fragment-packed copies (q3_common.h): tile blockIdx.x, k-step s = 1 KB at ((tile K/32 + s) 64 + lane) 8
    const int wstep = wpk ? 512 : 32;
    const size_t woff = wpk ? ((size_t)blockIdx.x * (K >> 5) + (kbeg >> 5)) * 512 + (size_t)lane * 8 : (size_t)nrow * K + kbeg + q * 8;
    const bf16_t* wp = pW + woff;
    const bf16_t* wp2 = DUAL ? pW2 + woff : nullptr;
    g16_u32x4 w[KWMAX], w2[DUAL ? KWMAX : 1];
#pragma unroll
    for (int ks = 0; ks < KWMAX; ++ks) {
        const int kk = ks < kw ? ks : kw - 1;
        w[ks] = g16_ldw(wp + kk * wstep, nt);
        if (DUAL) w2[ks] = g16_ldw(wp2 + kk * wstep, nt);
    }
    // epilogue operand (residual / bias) of the output this thread finishes: (m, n) = (tid / 16, tid % 16)
    const int em = tid >> 4, en = n0 + (tid & 15);
    float epi_in = 0.f;
    if (EPI == EPI_RESIDUAL || EPI == EPI_BIAS || EPI == EPI_BIAS_SILU) {
        if (tid < 256 && em < M && en < N) epi_in = EPI == EPI_RESIDUAL ? pepi[(size_t)em * pldepi + en] : pepi[en];
    }
    __builtin_amdgcn_sched_barrier(0);
    G16_MARK(1);   // every load issued

    if constexpr (NORM && GL) {   // gains into the wave's LDS slice (wave-private: the wave's in-order LDS queue is the synchronisation)
#pragma unroll
        for (int gp = 0; gp < GPASS; ++gp) {
            const int o = gp * 256 + lane * 4;
            if (o < KWMAX * 32) *reinterpret_cast<g16_f32x4*>(&gam_s[wave][o]) = gq[gp];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    f32x4 acc = { 0.f, 0.f, 0.f, 0.f }, acc2 = { 0.f, 0.f, 0.f, 0.f };
    float ss = 0.f;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g == 1) G16_MARK(2);   // first group's activations and weights arrived, its MFMAs issued
#pragma unroll
        for (int j = 0; j < G16_AB; ++j) {
            const int ks = g * G16_AB + j;
            if (ks >= KWMAX) continue;
            const float live = ks < kw ? 1.0f : 0.0f;
            float y[8], xin[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xin[e] = xa[g % PREG][j][0][e];
                if (R8) xin[4 + e] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, xin[e]), 0x128 /* row_ror:8 */, 0xF, 0xF, false));
                else xin[4 + e] = xa[g % PREG][j][R8 ? 0 : 1][e];
            }
            g16_f32x4 gl0 = { 0.f, 0.f, 0.f, 0.f }, gl1 = { 0.f, 0.f, 0.f, 0.f };
            if constexpr (NORM && GL) {
                const int kk = ks < kw ? ks : kw - 1;
                gl0 = *reinterpret_cast<const g16_f32x4*>(&gam_s[wave][kk * 32 + q * 8]);
                gl1 = *reinterpret_cast<const g16_f32x4*>(&gam_s[wave][kk * 32 + q * 8 + 4]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xv = (R8 && r16 >= 8 ? 0.f : xin[e]) * live;   // R8: the lanes of rows 8..15 hold other rows' chunks: zero, as the R8 = false kernel has there
                if (NORM) {
                    ss = fmaf(xv, xv, ss);
                    y[e] = xv * (GL ? (e < 4 ? gl0[e & 3] : gl1[e & 3]) : ga[GL ? 0 : g % PREG][j][e >> 2][e & 3]);
                }
                else y[e] = xv;
            }
            bf16x8 ah, al;
            g16_split8(y, ah, al);
            const bf16x8 bw = __builtin_bit_cast(bf16x8, w[ks]);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bw, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bw, acc, 0, 0, 0);
            if (DUAL) {
                const bf16x8 bw2 = __builtin_bit_cast(bf16x8, w2[ks]);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bw2, acc2, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bw2, acc2, 0, 0, 0);
            }
        }
        if (g + PREG < NG) {   // refill the buffer just consumed; the next group's conversion covers this L2 round trip
            __builtin_amdgcn_sched_barrier(0);
            issue(g % PREG, g + PREG);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    G16_MARK(3);   // all MFMAs issued
    // meet in LDS: D layout col = lane & 15 (n), row = q * 4 + reg (m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        red[0][wave][q * 4 + r][r16] = acc[r];
        if (DUAL) red[DUAL ? 1 : 0][wave][q * 4 + r][r16] = acc2[r];
    }
    if (NORM) {   // the four lanes of a row (q = 0..3) hold disjoint k: sum them, one value per (wave, row)
        ss += __shfl_xor(ss, 16);
        ss += __shfl_xor(ss, 32);
        if (q == 0) ssq[wave][r16] = ss;
    }
    __syncthreads();
    G16_MARK(4);   // all waves met
    if (tid < 256) {
        float v = 0.f, v2 = 0.f;
#pragma unroll
        for (int wv = 0; wv < NWV; ++wv) { v += red[0][wv][em][tid & 15]; if (DUAL) v2 += red[DUAL ? 1 : 0][wv][em][tid & 15]; }
        if (NORM) {
            float t = 0.f;
#pragma unroll
            for (int wv = 0; wv < NWV; ++wv) t += ssq[wv][em];
            const float inv = 1.0f / sqrtf(t / (float)K + a.eps);
            v *= inv; v2 *= inv;
        }
        if (em < M && en < N) {
            float o;
            if (EPI == EPI_STORE) o = v;
            else if (EPI == EPI_RESIDUAL) o = epi_in + v;
            else if (EPI == EPI_SWIGLU) o = silu_g(v) * v2;
            else if (EPI == EPI_BIAS) o = v + epi_in;
            else o = silu_g(v + epi_in);
            a.out[(size_t)em * a.ldo + en] = o;
        }
    }
    G16_MARK(5);   // outputs stored
    // optional fp32 copy of the normalised rows (the talker head keeps them as the predictor's first input row)
    if (NORM && a.xn_out != nullptr && blockIdx.x == 0) {
        for (int m = 0; m < M; ++m) {
            float t = 0.f;
#pragma unroll
            for (int wv = 0; wv < NWV; ++wv) t += ssq[wv][m];
            const float im = 1.0f / sqrtf(t / (float)K + a.eps);
            for (int k = tid; k < K; k += NWV * 64) a.xn_out[(size_t)m * a.ld_xn + k] = pgamma[k] * (px[(size_t)m * pldx + k] * im);
        }
    }
}

bool gemv16_ok(const GemvArgs& a) {
    if (a.M < 3 || a.M > 16 || a.po != nullptr) return false;
    if (a.K % 128 != 0 || a.K > 6144 || a.ldx % 4 != 0) return false;
    if (a.K > 1024 && a.K % 256 != 0) return false;   // 8 waves from there on
    if (a.gamma && a.epi != EPI_STORE && a.epi != EPI_SWIGLU) return false;
    // K > 3072 (24 k-steps per wave: the 1.7B talker's down projection) covers the residual / bias / plain-store epilogues only: with a
    // fused RMSNorm or two weight matrices on top the kernel spills (20 .. 524 bytes per lane), and no shipped config has such a shape
    if (a.K > 3072 && (a.gamma || a.epi == EPI_SWIGLU)) return false;
    return true;
}

template <int KWMAX, int NWV>
static void gemv16_epi(const GemvArgs& a, hipStream_t s) {
    const dim3 grid((a.N + 15) / 16), block(NWV * 64);
    const bool norm = a.gamma != nullptr;
    const bool no_r8 = knob("Q3TTS_GEMV16_R8") && atoi(knob("Q3TTS_GEMV16_R8")) == 0;   // A/B knob
    const bool r8 = a.M <= 8 && !no_r8;
    const bool no_gl = knob("Q3TTS_GEMV16_GL") && atoi(knob("Q3TTS_GEMV16_GL")) == 0;
    const bool gl = norm && !no_gl && (a.K / NWV) % 4 == 0;
    // fragment-packed weight copies, when the engine registered them for these matrices (both of a dual launch, or neither)
    const bf16_t* Wp = find_packed_weight(a.W);
    const bf16_t* W2p = a.W2 ? find_packed_weight(a.W2) : nullptr;
    const bool pk = Wp != nullptr && (a.W2 == nullptr || W2p != nullptr);
    const bf16_t* Wk = pk ? Wp : a.W;
    const bf16_t* W2k = pk && a.W2 ? W2p : a.W2;
#define Q3_G16_(EPI, NORM, R8_, GL_) hipLaunchKernelGGL((k_gemv16<KWMAX, NWV, EPI, NORM, R8_, GL_>), grid, block, 0, s, Wk, W2k, a.x, a.gamma, \
        (a.epi == EPI_RESIDUAL ? a.res : a.bias), a.N, a.M, a.ldx, a.ldres, (uint32_t)a.K | (a.nt ? 0x80000000u : 0u) | (pk ? 0x40000000u : 0u), a)
#define Q3_G16(EPI, NORM) do { if (NORM && gl) { if (r8) Q3_G16_(EPI, NORM, true, NORM); else Q3_G16_(EPI, NORM, false, NORM); } \
                               else { if (r8) Q3_G16_(EPI, NORM, true, false); else Q3_G16_(EPI, NORM, false, false); } } while (0)
    switch (a.epi) {
    case EPI_STORE:
        if constexpr (KWMAX <= 12) { if (norm) Q3_G16(EPI_STORE, true); else Q3_G16(EPI_STORE, false); }
        else { if (norm) throw Error("gemv16: K > 3072 has no fused-norm variant"); Q3_G16(EPI_STORE, false); }
        break;
    case EPI_SWIGLU:
        if constexpr (KWMAX <= 12) { if (norm) Q3_G16(EPI_SWIGLU, true); else Q3_G16(EPI_SWIGLU, false); }
        else throw Error("gemv16: K > 3072 has no SwiGLU variant");
        break;
    case EPI_RESIDUAL: Q3_G16(EPI_RESIDUAL, false); break;
    case EPI_BIAS: Q3_G16(EPI_BIAS, false); break;
    case EPI_BIAS_SILU: Q3_G16(EPI_BIAS_SILU, false); break;
    default: throw Error("gemv16: bad epilogue");
    }
#undef Q3_G16
#undef Q3_G16_
}
void launch_gemv16(const GemvArgs& a, hipStream_t s) {
    if (!gemv16_ok(a)) throw Error("gemv16: unsupported shape");
    // K <= 1024: 4 waves x <= 8 k-steps.  Wider K: 8 waves, so that at K <= 3072 every activation group is issued ahead of the weights
    // (converted while they stream) and the serial tail per wave after the last weight fragment stays at a few MFMAs.
    if (a.K <= 1024) gemv16_epi<8, 4>(a, s);
    else if (a.K <= 2048) gemv16_epi<8, 8>(a, s);
    else if (a.K <= 3072) gemv16_epi<12, 8>(a, s);
    else gemv16_epi<24, 8>(a, s);
}

// x[m][:] += sum_ks slab[ks][m][:] (fixed order), then RMSNorm(gamma) -> (hi, lo) planes (+ optional fp32 rows).
// One workgroup per row.  nslab == 0: plain RMSNorm + split.  gamma == null: residual update only.
static __device__ __forceinline__ float wave_sum_dpp(float v) {   // DPP path (see q3_decode_kernels.hip); __shfl_xor costs an LDS round trip per step
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, false));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// All slab loads of a thread are issued before the first add (clamped slab index, the sum itself runs in slab order over the first
// `nslab` only): a runtime-count loop of load-then-add had cost one L2 round trip per slab, 12 in a row after the down projection.
#define FIN_MAXS 16
template <int NS>   // slabs held in registers at once (>= nslab)
__global__ __launch_bounds__(256) void k_finish(float* x, int ldx, const float* slab, int nslab, size_t slab_stride, int ld_slab,
                                                 const float* gamma, float eps, int K, bf16_t* oh, bf16_t* ol, int ldp,
                                                 float* xn_out, int ld_xn) {
    __shared__ float red[4];
    const int m = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* xr = x + (size_t)m * ldx;
    float4 v[4], gv[4]; // K <= 4096
    float ss = 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        int k = (it * 256 + threadIdx.x) * 4;
        gv[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (it * 1024 >= K) { v[it] = make_float4(0.f, 0.f, 0.f, 0.f); continue; }   // workgroup-uniform
        const bool inr = k < K;
        k = inr ? k : K - 4;
        float4 t = *reinterpret_cast<const float4*>(xr + k);
        gv[it] = *reinterpret_cast<const float4*>((gamma != nullptr ? gamma : xr) + k);   // with the slabs, not behind the block barrier: one memory round per launch (address select, no conditional load)
        float4 p[NS > 0 ? NS : 1];
#pragma unroll
        for (int sidx = 0; sidx < NS; ++sidx) {
            const int sc = sidx < nslab ? sidx : nslab - 1;
            p[sidx] = *reinterpret_cast<const float4*>(slab + sc * slab_stride + (size_t)m * ld_slab + k);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int sidx = 0; sidx < NS; ++sidx)
            if (sidx < nslab) { t.x += p[sidx].x; t.y += p[sidx].y; t.z += p[sidx].z; t.w += p[sidx].w; }
        if (NS > 0 && inr) *reinterpret_cast<float4*>(xr + k) = t;
        if (!inr) t = make_float4(0.f, 0.f, 0.f, 0.f);
        v[it] = t;
        ss = fmaf(t.x, t.x, ss); ss = fmaf(t.y, t.y, ss); ss = fmaf(t.z, t.z, ss); ss = fmaf(t.w, t.w, ss);
    }
    if (gamma == nullptr) return;
    ss = wave_sum_dpp(ss);
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    const float r = 1.0f / sqrtf((((red[0] + red[1]) + red[2]) + red[3]) / (float)K + eps);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int k = (it * 256 + threadIdx.x) * 4;
        if (k < K) {
            const float4 g = gv[it];
            const float y[4] = { g.x * (v[it].x * r), g.y * (v[it].y * r), g.z * (v[it].z * r), g.w * (v[it].w * r) };
            split_store4(y, oh + (size_t)m * ldp + k, ol + (size_t)m * ldp + k);
            if (xn_out) *reinterpret_cast<float4*>(xn_out + (size_t)m * ld_xn + k) = make_float4(y[0], y[1], y[2], y[3]);
        }
    }
}
void launch_finish(float* x, int ldx, const float* slab, int nslab, size_t slab_stride, int ld_slab, const float* gamma, float eps,
                   int rows, int K, bf16_t* oh, bf16_t* ol, int ldp, float* xn_out, int ld_xn, hipStream_t s) {
    if (K % 4 || K > 4096 || nslab > FIN_MAXS) throw Error("finish: K must be a multiple of 4 and <= 4096, at most 16 slabs");
    if (rows <= 0) return;
#define Q3_FIN(NS) hipLaunchKernelGGL(k_finish<NS>, dim3(rows), dim3(256), 0, s, x, ldx, slab, nslab, slab_stride, ld_slab, gamma, eps, K, oh, ol, ldp, xn_out, ld_xn)
    if (nslab <= 0) Q3_FIN(0); else if (nslab <= 4) Q3_FIN(4); else if (nslab <= 8) Q3_FIN(8); else if (nslab <= 12) Q3_FIN(12); else Q3_FIN(16);
#undef Q3_FIN
}

// act = silu(sum gate slabs) * (sum up slabs) -> (hi, lo) planes; one workgroup per row
__global__ __launch_bounds__(256) void k_finish_swiglu(const float* gs, const float* us, int nslab, size_t slab_stride, int N,
                                                        bf16_t* oh, bf16_t* ol, int ldp) {
    const int m = blockIdx.x;
    // blockIdx.y = 1024-column chunk of the row: one memory round per workgroup (a loop over the chunks ran them back to back, three
    // dependent round trips per launch at ffn = 3072)
    for (int n0 = (blockIdx.y * 256 + threadIdx.x) * 4; n0 < N; n0 += 1024 * gridDim.y) {
        float4 pg[4], pu[4];   // split-K of gate/up is at most 4 (Engine::run_layers): every load first, sums in slab order
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) {
            const int sc = sidx < nslab ? sidx : nslab - 1;
            pg[sidx] = *reinterpret_cast<const float4*>(gs + sc * slab_stride + (size_t)m * N + n0);
            pu[sidx] = *reinterpret_cast<const float4*>(us + sc * slab_stride + (size_t)m * N + n0);
        }
        __builtin_amdgcn_sched_barrier(0);
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f), u = g;
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx)
            if (sidx < nslab) {
                g.x += pg[sidx].x; g.y += pg[sidx].y; g.z += pg[sidx].z; g.w += pg[sidx].w;
                u.x += pu[sidx].x; u.y += pu[sidx].y; u.z += pu[sidx].z; u.w += pu[sidx].w;
            }
        const float o[4] = { silu_g(g.x) * u.x, silu_g(g.y) * u.y, silu_g(g.z) * u.z, silu_g(g.w) * u.w };
        split_store4(o, oh + (size_t)m * ldp + n0, ol + (size_t)m * ldp + n0);
    }
}
void launch_finish_swiglu(const float* gs, const float* us, int nslab, size_t slab_stride, int rows, int N,
                          bf16_t* oh, bf16_t* ol, int ldp, hipStream_t s) {
    if (N % 4 || nslab < 1 || nslab > 4) throw Error("finish_swiglu: N must be a multiple of 4 and 1 <= nslab <= 4");
    if (rows > 0) hipLaunchKernelGGL(k_finish_swiglu, dim3(rows, (N + 1023) / 1024), dim3(256), 0, s, gs, us, nslab, slab_stride, N, oh, ol, ldp);
}

} // namespace q3
