// microbench_seam.hip — what does it cost to reduce split-K partial slabs INSIDE the GEMM launch instead of in a finish launch?
//
// Shapes of the b=64 decode step's three slab GEMMs per layer pass (64 rows, 64-column tiles, K slices across workgroups):
//   o_proj   N 1024, 8 slices  (128 workgroups)      gate/up  N 3072 x 2 products, 4 slices (192)      down  N 1024, 12 slices (192)
// A launch = a body that ingests what k_gemm3 ingests (32 KB of "weights" from a big buffer + 64 KB of the previous launch's planes)
// and produces a 64 x 64 fp32 tile of small exact integers, then one of the tails:
//   A  plain slab stores; a finish launch (one workgroup per row) sums the slabs in slab order, adds the residual, writes x and the
//      (hi, lo) planes                                                                    [today's structure]
//   B  sc1 (write-through) slab stores, drain, ticket on a per-tile counter; the LAST arriver sums the tile's slabs (sc1 loads) and
//      writes x, planes and a per-(row, tile) sum of squares                               [MI355X guide, hand-off table row 1]
//   C  as B, but every arriver that sees the tile complete within a bounded spin claims 16-row chunks of the reduction
//   D  as B with PLAIN slab stores and sc1 loads: correct only if every slice of a tile shares the reducer's XCD L2 — speed bound of an
//      XCD-local seam, with a count of the elements that came out wrong
// The chain (o_proj -> gate/up -> down, repeated) is captured in a hipGraph; x must be bit-identical across the variants.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mb_seam tools/microbench_seam.hip ; ./tools/mb_seam [layers] [replays]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short bf16_t;

#define ROWS 64
#define AUX_SC1 16
#define SPIN_MAX 400

struct Args {
    const unsigned* wbuf; size_t wwords;       // "weights": 32 KB per workgroup, rotating window of a big buffer
    const bf16_t* xh; const bf16_t* xl; int ldp; // previous launch's planes (the activation slice this workgroup stages)
    float* slab;                                // [KS][dual][ROWS][N]
    float* x;                                   // residual stream [ROWS][Nx] (single) / unused (dual)
    bf16_t* oh; bf16_t* ol;                     // planes out [ROWS][ldp]
    float* ssq;                                 // [ROWS][NT]
    unsigned* cnt;                              // per tile: arrival counter + 4 chunk claims, zeroed once per replay
    int* xcc_log;                               // [workgroup] XCC id
    int NT, KS, dual, N, it;
};

static __device__ __forceinline__ bf16_t bf16_rne(float f) {
    unsigned u = __float_as_uint(f);
    return (bf16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
static __device__ __forceinline__ float pattern(int s, int m, int n, int it, int d) { return (float)((s * 131 + m * 7 + n * 3 + it * 5 + d * 11) % 17); }

// what a k_gemm3 workgroup ingests before it can produce its tile; returns 0.0f that depends on every loaded byte
static __device__ __forceinline__ float body_ingest(const Args& a, int tile, int slice, unsigned char* smem) {
    const int tid = threadIdx.x;
    const unsigned* w = a.wbuf + (((size_t)(a.it * 977 + tile * a.KS + slice) * 8192) % (a.wwords - 8192));
    u32x4 b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) b[i] = *reinterpret_cast<const u32x4*>(w + (i * 256 + tid) * 4);           // 32 KB
    const int srow = tid >> 3, scol = (tid & 7) * 8;
    u32x4 sh[8], sl[8];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int p = 0; p < 2; ++p) {                                                                          // 64 rows x 256 k x (hi, lo) = 64 KB
            const size_t off = (size_t)(srow + 32 * p) * a.ldp + (slice * 256 + c * 64 + scol) % (a.ldp - 8);
            sh[c * 2 + p] = *reinterpret_cast<const u32x4*>(a.xh + off);
            sl[c * 2 + p] = *reinterpret_cast<const u32x4*>(a.xl + off);
        }
    u32x4* lds = reinterpret_cast<u32x4*>(smem);
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { lds[tid] = sh[i] ^ sl[i] ^ b[i]; __syncthreads(); const u32x4 t = lds[(tid * 7 + i) & 255]; acc |= (t.x & t.y & t.z & t.w); __syncthreads(); }
    return (float)(acc & 0u);
}

// finish launch of variant A: one workgroup per row (k_finish's shape)
__global__ __launch_bounds__(256) void k_finish(Args a) {
    const int m = blockIdx.x, Nout = a.N;
    for (int n0 = (blockIdx.y * 256 + threadIdx.x) * 4; n0 < Nout; n0 += 1024 * gridDim.y) {
        f32x4 t = a.dual ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(a.x + (size_t)m * Nout + n0);
        f32x4 u = {0.f, 0.f, 0.f, 0.f};
        f32x4 p[12], q[12];
#pragma unroll
        for (int s = 0; s < 12; ++s) {
            const int sc = s < a.KS ? s : a.KS - 1;
            p[s] = *reinterpret_cast<const f32x4*>(a.slab + (((size_t)sc * (a.dual + 1)) * ROWS + m) * Nout + n0);
            if (a.dual) q[s] = *reinterpret_cast<const f32x4*>(a.slab + (((size_t)sc * 2 + 1) * ROWS + m) * Nout + n0);
        }
#pragma unroll
        for (int s = 0; s < 12; ++s) if (s < a.KS) { t += p[s]; if (a.dual) u += q[s]; }
        if (a.dual) t = t * 0.25f + u;
        else *reinterpret_cast<f32x4*>(a.x + (size_t)m * Nout + n0) = t;
        bf16_t h[4], l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { h[j] = bf16_rne(t[j]); l[j] = bf16_rne(t[j] - __uint_as_float((unsigned)h[j] << 16)); }
        *reinterpret_cast<uint2*>(a.oh + (size_t)m * a.ldp + n0) = make_uint2(h[0] | (unsigned)h[1] << 16, h[2] | (unsigned)h[3] << 16);
        *reinterpret_cast<uint2*>(a.ol + (size_t)m * a.ldp + n0) = make_uint2(l[0] | (unsigned)l[1] << 16, l[2] | (unsigned)l[3] << 16);
    }
}

// MODE 0: plain stores, no seam (variant A's GEMM).  1: B.  2: C.  3: D.
template <int MODE, int CR = 16>
__global__ __launch_bounds__(256) void k_body(Args a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * ROWS * 68 * 4];
    __shared__ unsigned flag_s;
    const int tid = threadIdx.x;
    const int tile = blockIdx.x % a.NT, slice = blockIdx.x / a.NT;      // linear id = tile + NT * slice: NT % 8 == 0 puts a tile's slices on one XCD (round robin)
    if (tid == 0 && a.xcc_log) { unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); a.xcc_log[blockIdx.x] = (int)(xcc & 0xF); }
    const float zero = body_ingest(a, tile, slice, smem);
    // tile epilogue like k_gemm3's: [row][64 + 4] fp32 in LDS -> 16-byte row-contiguous stores
    float (*ep)[ROWS][68] = reinterpret_cast<float (*)[ROWS][68]>(smem);
    const int erow = tid >> 4, ecol = (tid & 15) * 4, ng = tile * 64 + ecol;
    const int nd = a.dual + 1;
#pragma unroll
    for (int p = 0; p < 4; ++p)
        for (int d = 0; d < nd; ++d)
#pragma unroll
            for (int j = 0; j < 4; ++j) ep[d][erow + 16 * p][ecol + j] = pattern(slice, erow + 16 * p, ng + j, a.it, d) + zero;
    __syncthreads();
    const size_t slab_bytes = (size_t)a.KS * nd * ROWS * a.N * 4;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.slab, 0, (int)slab_bytes, 0x00020000);
#pragma unroll
    for (int p = 0; p < 4; ++p)
        for (int d = 0; d < nd; ++d) {
            const int m = erow + 16 * p;
            const f32x4 v = *reinterpret_cast<const f32x4*>(&ep[d][m][ecol]);
            const unsigned off = (unsigned)(((((size_t)slice * nd + d) * ROWS + m) * a.N + ng) * 4);
            if (MODE == 1 || MODE == 2) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, off, 0, AUX_SC1);
            else *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(a.slab) + off) = v;
        }
    if (MODE == 0) return;
    // ---- seam: every storing wave drains, the workgroup meets, one lane takes the ticket ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned* cnt = a.cnt + (size_t)tile * 8;
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned f = old + 1 == (unsigned)a.KS ? 1u : 0u;
        if (MODE == 2 && !f) {
            for (int i = 0; i < SPIN_MAX; ++i) {
                if (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)a.KS) { f = 2u; break; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        flag_s = f;
    }
    __syncthreads();
    if (flag_s == 0) return;
    // ---- reduction of 16-row chunks: all four by the last arriver (B, D), claimed one at a time (C) ----
    constexpr int NCHK = ROWS / CR;
    for (int c0 = 0; c0 < NCHK; ++c0) {
        int chunk = c0;
        if (MODE == 2) {
            __syncthreads();
            if (tid == 0) flag_s = __hip_atomic_fetch_add(cnt + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            chunk = (int)flag_s;
            if (chunk >= NCHK) return;
        }
        if (erow >= CR) continue;   // uniform per 16-lane row group
        const int m = chunk * CR + erow;
        f32x4 t = a.dual ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(a.x + (size_t)m * a.N + ng);
        f32x4 u = {0.f, 0.f, 0.f, 0.f};
        u32x4 pp[12], qq[12];
#pragma unroll
        for (int s = 0; s < 12; ++s) {
            const int sc = s < a.KS ? s : a.KS - 1;
            pp[s] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(((((size_t)sc * nd) * ROWS + m) * a.N + ng) * 4), 0, AUX_SC1);
            if (a.dual) qq[s] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(((((size_t)sc * nd + 1) * ROWS + m) * a.N + ng) * 4), 0, AUX_SC1);
        }
#pragma unroll
        for (int s = 0; s < 12; ++s) if (s < a.KS) { t += __builtin_bit_cast(f32x4, pp[s]); if (a.dual) u += __builtin_bit_cast(f32x4, qq[s]); }
        if (a.dual) t = t * 0.25f + u;
        else *reinterpret_cast<f32x4*>(a.x + (size_t)m * a.N + ng) = t;
        bf16_t h[4], l[4];
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { h[j] = bf16_rne(t[j]); l[j] = bf16_rne(t[j] - __uint_as_float((unsigned)h[j] << 16)); ss = fmaf(t[j], t[j], ss); }
        *reinterpret_cast<uint2*>(a.oh + (size_t)m * a.ldp + ng) = make_uint2(h[0] | (unsigned)h[1] << 16, h[2] | (unsigned)h[3] << 16);
        *reinterpret_cast<uint2*>(a.ol + (size_t)m * a.ldp + ng) = make_uint2(l[0] | (unsigned)l[1] << 16, l[2] | (unsigned)l[3] << 16);
        // 16 lanes of a row: DPP-free butterfly through shuffles is fine here (a few dozen cycles)
        for (int off = 8; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 16);
        if ((tid & 15) == 0) a.ssq[(size_t)m * a.NT + tile] = ss;
    }
}

int main(int argc, char** argv) {
    const int layers = argc > 1 ? atoi(argv[1]) : 30, replays = argc > 2 ? atoi(argv[2]) : 30;
    const int H = 1024, FFN = 3072, ldp = 3072 + 8;
    hipStream_t st; CK(hipStreamCreate(&st));
    const size_t wwords = (size_t)64 << 20;   // 256 MB of "weights"
    unsigned* wbuf; CK(hipMalloc(&wbuf, wwords * 4)); CK(hipMemset(wbuf, 0, wwords * 4));
    bf16_t *pl[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) { CK(hipMalloc(&pl[i][j], (size_t)ROWS * ldp * 2)); CK(hipMemset(pl[i][j], 0, (size_t)ROWS * ldp * 2)); }
    float *slab, *x, *ssq; unsigned* cnt; int* xlog;
    CK(hipMalloc(&slab, (size_t)12 * 2 * ROWS * FFN * 4)); CK(hipMalloc(&x, (size_t)ROWS * H * 4)); CK(hipMalloc(&ssq, (size_t)ROWS * 64 * 4));
    const int n_launch = layers * 3;
    CK(hipMalloc(&cnt, (size_t)n_launch * 64 * 8 * 4)); CK(hipMalloc(&xlog, 256 * 4));
    std::vector<float> xref((size_t)ROWS * H), xv((size_t)ROWS * H);
    std::vector<bf16_t> href((size_t)ROWS * ldp), hv((size_t)ROWS * ldp);

    for (int variant = 0; variant < 6; ++variant) {
        CK(hipMemset(x, 0, (size_t)ROWS * H * 4));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        CK(hipMemsetAsync(cnt, 0, (size_t)n_launch * 64 * 8 * 4, st));
        int li = 0;
        for (int l = 0; l < layers; ++l) {
            for (int k = 0; k < 3; ++k, ++li) {
                Args a;
                a.wbuf = wbuf; a.wwords = wwords; a.ldp = ldp; a.slab = slab; a.x = x; a.ssq = ssq; a.cnt = cnt + (size_t)li * 64 * 8; a.xcc_log = (l == 0 && k == 0) ? xlog : nullptr;
                a.it = li;
                const int in = k & 1, out = in ^ 1;               // planes ping-pong: o_proj reads 1 writes 0, gate/up reads 0 writes 1, down reads 1 writes 0
                a.xh = pl[k == 1 ? 0 : 1][0]; a.xl = pl[k == 1 ? 0 : 1][1]; a.oh = pl[k == 1 ? 1 : 0][0]; a.ol = pl[k == 1 ? 1 : 0][1];
                (void)in; (void)out;
                if (k == 0) { a.NT = 16; a.KS = 8; a.dual = 0; a.N = H; }
                else if (k == 1) { a.NT = 48; a.KS = 4; a.dual = 1; a.N = FFN; }
                else { a.NT = 16; a.KS = 12; a.dual = 0; a.N = H; }
                const dim3 grid(a.NT * a.KS);
                if (variant == 0) { hipLaunchKernelGGL(k_body<0>, grid, dim3(256), 0, st, a); hipLaunchKernelGGL(k_finish, dim3(ROWS, a.dual ? 3 : 1), dim3(256), 0, st, a); }
                else if (variant == 1) hipLaunchKernelGGL(k_body<1>, grid, dim3(256), 0, st, a);
                else if (variant == 2) hipLaunchKernelGGL(k_body<2>, grid, dim3(256), 0, st, a);
                else if (variant == 3) hipLaunchKernelGGL(k_body<3>, grid, dim3(256), 0, st, a);
                else if (variant == 4) hipLaunchKernelGGL((k_body<2, 8>), grid, dim3(256), 0, st, a);
                else hipLaunchKernelGGL(k_body<0>, grid, dim3(256), 0, st, a);
            }
        }
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        // correctness after ONE replay from x = 0
        CK(hipMemcpy(xv.data(), x, xv.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hv.data(), pl[0][0], hv.size() * 2, hipMemcpyDeviceToHost));
        size_t bad = 0, badp = 0;
        if (variant == 0) { xref = xv; href = hv; }
        else {
            for (size_t i = 0; i < xv.size(); ++i) bad += xv[i] != xref[i];
            for (int m = 0; m < ROWS; ++m) for (int n = 0; n < H; ++n) badp += hv[(size_t)m * ldp + n] != href[(size_t)m * ldp + n];
        }
        if (variant == 0) {   // the reference itself against the closed form
            size_t badr = 0;
            for (int m = 0; m < ROWS; ++m) for (int n = 0; n < H; ++n) {
                float e = 0.f; int l2 = 0;
                for (int l = 0; l < layers; ++l) for (int k = 0; k < 3; ++k, ++l2) if (k != 1) { const int KS = k == 0 ? 8 : 12; for (int s = 0; s < KS; ++s) e += (float)((s * 131 + m * 7 + n * 3 + l2 * 5) % 17); }
                badr += e != xv[(size_t)m * H + n];
            }
            printf("variant A vs closed form: %zu wrong\n", badr);
        }
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < replays; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        static const char* names[6] = { "A  plain slabs + finish launch       ", "B  sc1 slabs, last arriver reduces    ", "C  sc1 slabs, arrivers claim chunks   ", "D  PLAIN slabs + last arriver (UNSAFE)",
                                        "C8 as C with 8-row chunks             ", "E  GEMM bodies only (no reduction)    " };
        printf("%s  %7.2f us per layer triple (o_proj, gate/up, down)   x mismatches %zu, plane mismatches %zu\n", names[variant], ms * 1e3 / replays / layers, bad, badp);
        if (variant == 0) {
            std::vector<int> xl(256); CK(hipMemcpy(xl.data(), xlog, 128 * 4, hipMemcpyDeviceToHost));
            int same = 0; for (int t = 0; t < 16; ++t) { bool s = true; for (int k = 1; k < 8; ++k) s = s && xl[t + 16 * k] == xl[t]; same += s; }
            printf("o_proj tiles whose 8 K slices share one XCD: %d of 16  (XCC of workgroups 0..15:", same);
            for (int t = 0; t < 16; ++t) printf(" %d", xl[t]);
            printf(")\n");
        }
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
