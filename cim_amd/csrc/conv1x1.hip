// Backbone 1x1 convolutions as small-tile fp32-MFMA GEMMs with the BatchNorm / residual / ReLU chain in the epilogue.
//
// Replaces, for the ResNet-50 C4 body of /root/reference/lib/modeling/resnet50.py:17-91 (torchvision Bottleneck:
// conv1 / conv3 / downsample.0 are 1 x 1 convolutions, each followed by a BatchNorm kept in eval mode :53-77), the
// ATen -> MIOpen -> rocBLAS path those layers took (31 GEMM launches + 31 BatchNorm launches forward at cfg2, on
// 64 x 32 ... 128 x 128 Tensile tiles at ~47 TF forward / ~26 TF backward) by ONE launch per layer and direction.
//
// In NCHW a 1 x 1 convolution of one image is  Y[Cout][HW] = W[Cout][Cin] . X[Cin][HW]:
//   forward          A = W   (row-major, K-contiguous)        B = X  ([K][N], N-contiguous)
//   data gradient    A = W^T (element (m,k) at W[k*Cin + m])  B = dY ([K][N], N-contiguous)
//   weight gradient  A = dY  (K = HW contiguous)              B = X^T (element (k,n) at X[n*HW + k]: K-contiguous)
// The contraction sizes are small (M, K = 64 ... 1024, N = HW = 1.4k ... 22k), so the tile is 64 x 64 x 16 with four
// waves of one 32 x 32 MFMA tile each (v_mfma_f32_32x32x2_f32: true fp32 products, no operand splitting), k-major LDS
// slabs ([k][m]: a lane's MFMA operand is one conflict-free ds_read_b32), double buffered, the next slab's global loads
// in flight while the current one is multiplied; up to 8 workgroups per CU hide the rest.  Short-K / long-K products
// (weight gradients: K = HW) take split-K through a workspace + one reduce pass (no float atomics: they run at
// ~90 G/s on MI355X).
// Epilogue (forward): x = acc (stored when the backward needs the convolution output), y = relu?(x * a[m] + b[m] (+ res)),
// a = gamma * rsqrt(var + eps), b = beta - mean * a: the BatchNorm (+ identity) (+ ReLU) of the bottleneck costs no pass.
#include "common.h"
#include "../../include/cim_hip.h"

// The backbone's kernels run BESIDE MaskFuse's late weight-gradient products in the last phase of the backward pass (the co-resident form
// of the pair GEMM, csrc/gemm_pair.hip: both on the same CUs) and they are the longer chain: their waves ask for the higher issue priority.
#ifndef CIM_BODY_PRIO_LEVEL
#define CIM_BODY_PRIO_LEVEL 3
#endif
#if CIM_BODY_PRIO_LEVEL > 0
#define CIM_BODY_PRIO() __builtin_amdgcn_s_setprio(CIM_BODY_PRIO_LEVEL)
#else
#define CIM_BODY_PRIO()
#endif
namespace {

#ifndef CIM_SMALL_BK
#define CIM_SMALL_BK 32
#endif
#ifndef CIM_SMALL_ABL
#define CIM_SMALL_ABL 0          // ablation builds (tools/build_alt.sh): 1 one MFMA per slab, 2 no stores, 3 no global loads (3 x 3: no gathers), 4 empty kernel, 5 (3 x 3) gathers without address arithmetic
#endif
constexpr int SBM = 64, SBN = 64, SBK = CIM_SMALL_BK;
constexpr int SLD = 68;                      // padded row stride of a k-major slab (floats); % 4 == 0 for b128 stores
typedef float f32x16 __attribute__((ext_vector_type(16)));
// 16-byte global loads from 4-byte aligned addresses: rows of an NCHW activation [C][HW] start wherever HW puts them
// (HW = 1419, 5590 at cfg2), and gfx9's global_load_dwordx4 only needs dword alignment (unaligned access mode is on under
// ROCm); the type tells the compiler not to assume more.  The first version fell back to four 4-byte loads per lane for
// such operands and ran 2x slower than the aligned one.
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ float4 ld4(const float* p) {
    const f32x4u v = *reinterpret_cast<const f32x4u*>(p);
    return make_float4(v.x, v.y, v.z, v.w);
}

struct SmallArgs {
    const float* A; const float* B; float* C; float* Xraw;
    const float* gamma; const float* beta; const float* mean; const float* var; const float* res;
    float eps;
    int M, N, K, lda, ldb, ldc;
    int a_mcontig, b_kcontig, relu, bn;
    int splits; float* ws;
    const float* colbias;    // [N] added per output COLUMN (nn.Linear bias: C = X . W^T + b); null for the convolutions
    // data gradients only: the BatchNorm + ReLU backward of the layer that PRODUCED this product's input, applied to the
    // result - C(row, col) = mask(row, col) > 0 ? C * mgamma[row] rsqrt(mvar[row] + meps) : 0 with mask = that layer's output
    // (this layer's input, same layout as C): the producer's own backward then starts from the gradient of its convolution
    const float* mask; const float* mgamma; const float* mvar; float meps;
    // ... and, when that layer's gamma / beta TRAIN (the reference's configuration, lib/modeling/resnet50.py:59-60), the per-channel
    // sums its BatchNorm backward would have made, as partial sums over the 32-column group `col / 32` of every row:
    //   mpart[0][col / 32][row] = sum dz,   mpart[1][col / 32][row] = sum dz (xraw - mean)     (dz = mask > 0 ? C : 0)
    // with mxr = that layer's convolution output (layout of C) and mmean its running mean; bn_part_finish_kernel sums the
    // groups in index order (deterministic) and scales the second by rsqrt(var + eps).  mparts = ceil(N / 32).
    const float* mxr; const float* mmean; float* mpart; int mparts;
    // res_w > 0: `res` is a [M][ceil(H / 2)][ceil(W / 2)] tensor that belongs to every SECOND pixel of the H x W = N output pixels
    // (W = res_w): the data gradient of a stride-2 1 x 1 convolution of the same input, added where it lands (zero elsewhere)
    int res_w, res_ws, res_hw;
};
struct InputBn { const float* y; const float* gamma; const float* var; float eps;           // (the fields above as arguments)
                 const float* xr; const float* mean; float* part; };

// Tile loaders.  An operand tile is ROWS x 32 (k) floats per slab, moved as 16-byte pieces: piece index p ->
//   K-contiguous operand (element (r, k) at P[r*ld + k]):  r = p % ROWS, k = (p / ROWS) * 4   (lanes along the rows: the
//     transposing LDS stores are conflict-free; with lanes along k they were 4-way conflicted)
//   row-contiguous operand (element (r, k) at P[k*ld + r]): k = p / (ROWS/4), r = (p % (ROWS/4)) * 4
// LDS slabs are k-major ([k][row], stride SLD): row-contiguous pieces are one ds_write_b128, K-contiguous ones transpose.
// Round 4: the loads are BRANCH-FREE buffer loads.  The first loaders guarded every piece (row / k range, 16-byte or three
// scalar tail loads): hipcc gave each guarded load its own basic block - ~30 branches and four waits per slab in front of 16
// MFMAs (found in the ISA: 351 s_cbranch in the kernel).  A raw buffer resource over the operand's exact extent makes the
// hardware return 0 for everything behind it - the k >= K rows of a row-contiguous operand, rows >= `rows` of a K-contiguous
// one - the K tail INSIDE a K-contiguous row (the next row's data) is zeroed by four selects, and what a row-contiguous piece
// reads past `rows` (the next k-row's first elements) only reaches output rows / columns that are never stored.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const float* p, long long floats) {       // (p, floats: wave-uniform kernel arguments)
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)(unsigned)(floats * 4), 0x00020000);
}
__device__ __forceinline__ float4 bld4(rsrc_t r, unsigned byte_off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ float bld1(rsrc_t r, unsigned byte_off) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}
constexpr unsigned OOB = 0x7fffffffu;          // a byte offset behind every operand (extents are < 2^31 bytes: checked by the launchers)
// extent of an operand in floats: rows x K elements, leading dimension ld
__host__ __device__ __forceinline__ long long tile_extent(bool RC, int rows, int K, int ld) {
    return RC ? (long long)(K - 1) * ld + rows : (long long)(rows - 1) * ld + K;
}
// byte offset of piece p of the slab at k0 (tile rows from r0)
template <bool RC, int ROWS>
__device__ __forceinline__ unsigned tile_off(int ld, int r0, int k0, int p) {
    int r, k;
    if (RC) { k = k0 + p / (ROWS / 4); r = r0 + (p % (ROWS / 4)) * 4; return ((unsigned)k * ld + r) * 4u; }
    r = r0 + p % ROWS; k = k0 + (p / ROWS) * 4;
    return ((unsigned)r * ld + k) * 4u;
}
// (the K tail of a K-contiguous piece is zeroed when the piece is STORED to LDS: a select right behind the load would make the
// wave wait for the load it has just issued)
template <bool RC, int ROWS>
__device__ __forceinline__ float4 tile_ktail(float4 v, int K, int k0, int p) {
    if (!RC) {
        const int left = K - (k0 + (p / ROWS) * 4);
        v.x = left > 0 ? v.x : 0.0f; v.y = left > 1 ? v.y : 0.0f; v.z = left > 2 ? v.z : 0.0f; v.w = left > 3 ? v.w : 0.0f;
    }
    return v;
}

template <bool RC, int ROWS>
__device__ __forceinline__ void tile_store(float* __restrict__ S, float4 v, int p) {
    if (RC) {
        const int k = p / (ROWS / 4), r = (p % (ROWS / 4)) * 4;
        *reinterpret_cast<float4*>(S + k * SLD + r) = v;
    } else {
        const int r = p % ROWS, k = (p / ROWS) * 4;      // a wave's lanes hold consecutive rows: conflict-free transposing stores
        S[(k + 0) * SLD + r] = v.x;
        S[(k + 1) * SLD + r] = v.y;
        S[(k + 2) * SLD + r] = v.z;
        S[(k + 3) * SLD + r] = v.w;
    }
}

// Split-K: every split stores its partial tile to the workspace, a separate reduce launch (small_splitk_reduce*_kernel) sums them in
// split order and applies the epilogue.  (Round 3 built the combine INSIDE the launch - the tile's last arriving workgroup sums the
// partials, arrival counters in device memory - and measured it slower in the whole step: 17.92 / 16.41 ms against 14.87 ms with the
// separate reduce; removed in round 5 together with its library-owned counter ring)
// Epilogue of one output value (shared by the single-pass kernels, the in-kernel split-K combine and the reduce kernel)
// (small_value: the epilogue's value without its stores - the vectorised reduce stores four of them at once)
__device__ __forceinline__ float small_value(const SmallArgs& g, int row, int col, float v, float& s1, float& s2) {
    const size_t o = (size_t)row * g.ldc + col;
    float y = v;
    if (g.bn) {
        const float a = g.gamma[row] * rsqrtf(g.var[row] + g.eps);
        y = fmaf(v, a, g.beta[row] - g.mean[row] * a);
    }
    if (g.res) {
        if (g.res_w == 0) y += g.res[o];
        else {
            const int py = col / g.res_w, px = col - py * g.res_w;
            if (!((py | px) & 1)) y += g.res[(size_t)row * g.res_hw + (py >> 1) * g.res_ws + (px >> 1)];
        }
    }
    if (g.colbias) y += g.colbias[col];
    if (g.relu) y = fmaxf(y, 0.0f);
    if (g.mask) {
        const bool on = g.mask[o] > 0.0f;
        if (g.mpart && on) { s1 += y; s2 = fmaf(y, g.mxr[o] - g.mmean[row], s2); }
        y = on ? y * (g.mgamma[row] * rsqrtf(g.mvar[row] + g.meps)) : 0.0f;
    }
    return y;
}
__device__ __forceinline__ void small_finish(const SmallArgs& g, int row, int col, float v, float& s1, float& s2) {
    const size_t o = (size_t)row * g.ldc + col;
    if (g.Xraw) g.Xraw[o] = v;
    g.C[o] = small_value(g, row, col, v, s1, s2);
}
__device__ __forceinline__ void small_finish(const SmallArgs& g, int row, int col, float v) {
    float s1 = 0.0f, s2 = 0.0f;
    small_finish(g, row, col, v, s1, s2);
}
// the two sums of a row over the 32 lanes of a half-wave (lanes = consecutive columns) -> mpart, written by the group's first lane
__device__ __forceinline__ void small_put_part(const SmallArgs& g, int row, int group, int lane, float s1, float s2) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
    }
    if ((lane & 31) == 0 && row < g.M && group < g.mparts) {
        g.mpart[(size_t)group * g.M + row] = s1;
        g.mpart[((size_t)g.mparts + group) * g.M + row] = s2;
    }
}
// Epilogue of a wave's 32 x 32 accumulator tile: lane holds rows 8*(r/4) + 4*(lane/32) + r%4, column lane%32
__device__ __forceinline__ void small_epilogue(const SmallArgs& g, const f32x16& acc, int row0, int col0, int lane, int split) {
    const int col = col0 + (lane & 31);
    if (g.splits > 1) {
        if (col < g.N) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
                if (row < g.M) g.ws[((size_t)split * g.M + row) * g.N + col] = acc[r];
            }
        }
        return;
    }
    if (g.mpart == nullptr) {
        if (col < g.N) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
                if (row < g.M) small_finish(g, row, col, acc[r]);
            }
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {          // (every lane takes part in the row sums)
        const int row = row0 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
        float s1 = 0.0f, s2 = 0.0f;
        if (col < g.N && row < g.M) small_finish(g, row, col, acc[r], s1, s2);
        small_put_part(g, row, col0 >> 5, lane, s1, s2);
    }
}


// WN = waves along N: 2 -> 64 x 64 tile, 256 threads; 1 -> 64 x 32 tile, 128 threads (twice the workgroups for the
// smallest problems).  Tried and measured slower (tools/bench_gemm_small.py, 7 layer shapes, us forward / dX / dW:
// 158 / 137 / 183 with this kernel): 128- and 256-row tiles (4 waves of 1 x 2 / 2 x 2 MFMA tiles, 134-170 VGPRs) to re-read
// the activation operand less often: 251 / 207 / 290 - these products are bound by the latency chain of a tile's few
// slabs and by how many tiles are in flight, not by operand traffic.
template <bool AM, bool BKc, int WN>
__global__ __launch_bounds__(128 * WN) void gemm_small_kernel(const SmallArgs g) {
    CIM_BODY_PRIO();
    constexpr int NT = 128 * WN, BNT = 32 * WN;
    constexpr int PA = SBM * 8 / NT, PB = BNT * 8 / NT;          // 16-byte pieces per thread and slab: A 4 / 2, B 2 / 2
    __shared__ __attribute__((aligned(16))) float As[2][SBK * SLD];
    __shared__ __attribute__((aligned(16))) float Bs[2][SBK * SLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // tile order: N fastest (neighbouring workgroups share the A = weight panel through L2)
    const int tiles_n = (g.N + BNT - 1) / BNT;
    const int tile = blockIdx.x;
    const int m0 = (tile / tiles_n) * SBM, n0 = (tile % tiles_n) * BNT;
    const int split = blockIdx.y;
    const int kper = ((g.K + g.splits - 1) / g.splits + SBK - 1) / SBK * SBK;      // (a multiple of the slab depth)
    const int kbeg = split * kper, kend = min(g.K, kbeg + kper);
#if CIM_SMALL_ABL == 4
    if (g.M > 0) return;
#endif

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;

    // Two register sets of pieces: the loads of slabs s + 1 and s + 2 are in flight while slab s is multiplied (one global
    // round trip is ~2 MFMA blocks of 16 x 64 cycles), LDS double buffered, ONE barrier per slab.  The LDS operands of a slab
    // are read in one batch in front of its 16 MFMAs (the first version read two values, waited, multiplied: eight exposed LDS
    // latencies per slab).
    const rsrc_t RA_ = make_rsrc(g.A, tile_extent(AM, g.M, g.K, g.lda)), RB_ = make_rsrc(g.B, tile_extent(!BKc, g.N, g.K, g.ldb));
    unsigned oa[PA], ob[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) oa[i] = tile_off<AM, SBM>(g.lda, m0, kbeg, tid + i * NT);
#pragma unroll
    for (int i = 0; i < PB; ++i) ob[i] = tile_off<!BKc, BNT>(g.ldb, n0, kbeg, tid + i * NT);
    const unsigned sa = (AM ? (unsigned)g.lda * SBK : (unsigned)SBK) * 4u, sb = (!BKc ? (unsigned)g.ldb * SBK : (unsigned)SBK) * 4u;
    float4 ra0[PA], rb0[PB], ra1[PA], rb1[PB];
#if CIM_SMALL_ABL == 3      /* ablation: no global loads */
#define SM_GLOAD(RA, RB, S)                                                                                        \
    {                                                                                                              \
        _Pragma("unroll") for (int i = 0; i < PA; ++i) asm volatile("" : "+v"(RA[i].x), "+v"(RA[i].y), "+v"(RA[i].z), "+v"(RA[i].w)); \
        _Pragma("unroll") for (int i = 0; i < PB; ++i) asm volatile("" : "+v"(RB[i].x), "+v"(RB[i].y), "+v"(RB[i].z), "+v"(RB[i].w)); \
    }
#else
#define SM_GLOAD(RA, RB, S)                                                                                        \
    {                                                                                                              \
        _Pragma("unroll") for (int i = 0; i < PA; ++i) RA[i] = bld4(RA_, oa[i] + (unsigned)(S) * sa);               \
        _Pragma("unroll") for (int i = 0; i < PB; ++i) RB[i] = bld4(RB_, ob[i] + (unsigned)(S) * sb);               \
    }
#endif
#define SM_PUT(RA, RB, BUF, S)                                                                                     \
    {                                                                                                              \
        _Pragma("unroll") for (int i = 0; i < PA; ++i)                                                             \
            tile_store<AM, SBM>(As[BUF], tile_ktail<AM, SBM>(RA[i], g.K, kbeg + (S) * SBK, tid + i * NT), tid + i * NT);     \
        _Pragma("unroll") for (int i = 0; i < PB; ++i)                                                             \
            tile_store<!BKc, BNT>(Bs[BUF], tile_ktail<!BKc, BNT>(RB[i], g.K, kbeg + (S) * SBK, tid + i * NT), tid + i * NT); \
    }
#define SM_MMA(BUF)                                                                                                \
    {                                                                                                              \
        const float* __restrict__ a = As[BUF] + (lane >> 5) * SLD + wm * 32 + (lane & 31);                         \
        const float* __restrict__ b = Bs[BUF] + (lane >> 5) * SLD + wn * 32 + (lane & 31);                         \
        float av[SBK / 2], bv[SBK / 2];                                                                            \
        _Pragma("unroll") for (int t = 0; t < SBK / 2; ++t) { av[t] = a[2 * t * SLD]; bv[t] = b[2 * t * SLD]; }    \
        _Pragma("unroll") for (int t = 0; t < (CIM_SMALL_ABL == 1 ? 1 : SBK / 2); ++t)                             \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[t], acc, 0, 0, 0);                                \
        asm volatile("" : "+a"(acc));      /* (the accumulator stays in AGPRs over the loop: hipcc moved it out and back per slab) */ \
    }
    if (kbeg < kend) {
        const int nslab = (kend - kbeg + SBK - 1) / SBK;
        // (the loads are issued unconditionally - behind the last slab they fetch it again: behind a uniform branch hipcc's wait
        // counts assume the path WITHOUT the new loads and wait for them as if they were the old ones)
        const int last = nslab - 1;
        SM_GLOAD(ra0, rb0, 0)
        SM_GLOAD(ra1, rb1, min(1, last))
        SM_PUT(ra0, rb0, 0, 0)
        asm volatile("" : "+a"(acc));
        __syncthreads();
        for (int s = 0; s < nslab; s += 2) {
            SM_GLOAD(ra0, rb0, min(s + 2, last))
            SM_MMA(0)
            if (s + 1 < nslab) SM_PUT(ra1, rb1, 1, s + 1)
            __syncthreads();
            if (s + 1 >= nslab) break;
            SM_GLOAD(ra1, rb1, min(s + 3, last))
            SM_MMA(1)
            if (s + 2 < nslab) SM_PUT(ra0, rb0, 0, s + 2)
            __syncthreads();
        }
    }
#undef SM_GLOAD
#undef SM_PUT
#undef SM_MMA

#if CIM_SMALL_ABL == 2       /* ablation: no stores (one lane of a tile that does not exist keeps the product alive) */
    if (m0 + wm * 32 + lane < 0) small_epilogue(g, acc, m0 + wm * 32, n0 + wn * 32, lane, split);
#else
    small_epilogue(g, acc, m0 + wm * 32, n0 + wn * 32, lane, split);
#endif
}

// Round 6, measured and dropped: a TILE-WALKING form of this kernel - R workgroups per CU (the LDS request padded so that exactly R
// fit) that walk over the (tile, split) items b, b + G, ... with the operand pipeline running across item boundaries (three cursors:
// loads two slabs ahead, LDS stores one ahead, MFMAs), epilogue stores draining under the next item's MFMAs.  Bit-identical to this
// kernel on every shape and epilogue, and SLOWER (tools/bench_gemm_small.py, 7 layer shapes, us forward / dX / dW): 205 / 159 / 153
// at R = 1, 177 / 143 / 149 at R = 2, 169 / 137 / 148 at R = 4 against 160 / 130 / 148 - these launches live on the NUMBER of
// workgroups a CU holds (memory operations in flight), not on a long pipeline per workgroup.  The phase ablations of the same
// round (profiles/r6/gemm_small_ablation_kernel_trace.txt; res3.conv3 forward, 704 workgroups x 4 slabs, 17.8 us of GPU time):
// without the stores 10.2 us, with one MFMA per slab instead of 16: 14.3 us, without the global loads 15.8 us - the output
// (y + the convolution output the BatchNorm backward needs, 22.9 MB at ~3 TB/s) is the largest single part, the MFMAs the smallest.
// Round 4, measured and dropped: a REGISTER-DIRECT form of this product - every wave fetches its MFMA operands straight from memory
// (lane (r, h) takes the 16 k of its half of a slab: four 16-byte loads per K-contiguous row, 16 coalesced 4-byte loads per
// row-contiguous one; no LDS staging, no barrier in the loop), one 32 x 32 tile per workgroup whose four waves split K and meet in
// LDS, so that long-K products need no split-K over workgroups and no reduce launch.  Alone (tools/bench_gemm_small.py, 7 layer
// shapes, us forward / dX / dW) 147 / 126 / 121 against this kernel's 159 / 128 / 147 - but INSIDE the step slower: body forward
// 2.10 vs 1.87 ms, body backward phase 4.62 vs 4.27 ms, step 14.6-14.7 vs 13.8-14.2 ms (same box, interleaved): its operands are
// re-read from L2 by every 32 x 32 tile (~90 MB per res4 layer) and its 4x more workgroups queue behind the 256-workgroup
// launches of MaskFuse's late weight gradients, which is where the body's backward actually runs (bench.py --phases).
// split-K: sum of the partial products in a fixed order (deterministic) + the same epilogue as the single-pass kernel.
// The partials of an element are loaded eight at a time (independent loads in flight) and added in split order: with a
// load-add-load-add loop a thread paid one memory round trip per split (8-64 of them), and the ~90 reduce launches per step
// were bound by exactly that chain.
__device__ __forceinline__ float splitk_sum(const float* __restrict__ ws, size_t i, size_t mn, int splits) {
    float v = ws[i];
    for (int k0 = 1; k0 < splits; k0 += 8) {
        float p[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) p[j] = k0 + j < splits ? ws[(size_t)(k0 + j) * mn + i] : 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (k0 + j < splits) v += p[j];
    }
    return v;
}
__global__ __launch_bounds__(256) void small_splitk_reduce_kernel(const SmallArgs g) {
    CIM_BODY_PRIO();
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t mn = (size_t)g.M * g.N;
    if (i >= mn) return;
    small_finish(g, (int)(i / g.N), (int)(i % g.N), splitk_sum(g.ws, i, mn, g.splits));
}
// Round 6: FOUR consecutive elements of the flat [M][N] result per thread (16-byte loads of the partial tiles, four of them in flight
// per split batch) - the launch above has one 4-byte load per thread and split in flight and ran at 2.5 TB/s on 20 MB
// (profiles/r6/gemm_small_ablation_kernel_trace.txt: 4.4-8.5 us).  Same sums in the same order: bit-identical.  Needs ldc == N,
// M N % 4 == 0 and 16-byte aligned workspace (the launcher checks); the four elements may straddle a row.
__global__ __launch_bounds__(256) void small_splitk_reduce4_kernel(const SmallArgs g) {
    CIM_BODY_PRIO();
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t mn = (size_t)g.M * g.N;
    const size_t i = q * 4;
    if (i >= mn) return;
    float4 v = *reinterpret_cast<const float4*>(g.ws + i);
    for (int k0 = 1; k0 < g.splits; k0 += 4) {
        float4 p[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            p[j] = k0 + j < g.splits ? *reinterpret_cast<const float4*>(g.ws + (size_t)(k0 + j) * mn + i) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (k0 + j < g.splits) { v.x += p[j].x; v.y += p[j].y; v.z += p[j].z; v.w += p[j].w; }
    }
    int row = (int)(i / g.N), col = (int)(i - (size_t)row * g.N);
    const float s[4] = {v.x, v.y, v.z, v.w};
    float y[4], s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        y[j] = small_value(g, row, col, s[j], s1, s2);
        if (++col == g.N) { col = 0; ++row; }
    }
    if (g.Xraw) *reinterpret_cast<float4*>(g.Xraw + i) = v;               // (ldc == N: element (row, col) is element i of the flat array)
    *reinterpret_cast<float4*>(g.C + i) = make_float4(y[0], y[1], y[2], y[3]);
}

// the same with the producer's affine-gradient partial sums (SmallArgs.mpart): a block owns 256 consecutive columns of ONE row,
// so that every half-wave is one 32-column group of that row.  grid (ceil(N / 256), M)
__global__ __launch_bounds__(256) void small_splitk_reduce_rows_kernel(const SmallArgs g) {
    CIM_BODY_PRIO();
    const int col = blockIdx.x * 256 + threadIdx.x, row = blockIdx.y;
    const size_t mn = (size_t)g.M * g.N;
    float s1 = 0.0f, s2 = 0.0f;
    if (col < g.N) {
        const size_t i = (size_t)row * g.N + col;
        small_finish(g, row, col, splitk_sum(g.ws, i, mn, g.splits), s1, s2);
    }
    small_put_part(g, row, col >> 5, threadIdx.x & 63, s1, s2);
}

// Finishing the affine gradients of up to 24 chained layers in ONE launch (at the end of the backward pass):
// dbeta[c] = sum over images and column groups of part[b][0][p][c]; dgamma[c] = rsqrt(var[c] + eps) * sum part[b][1][p][c], in
// index order.  A block owns 32 channels of one layer: 8 thread groups take every 8th (image, group) pair - four loads in
// flight each -, then one fixed-order sum in LDS.
struct BnPartTable { cim_bn_part_desc d[24]; int first[25]; int n; };
__global__ __launch_bounds__(256) void bn_part_finish_kernel(const BnPartTable t) {
    CIM_BODY_PRIO();
    __shared__ float sh[2][8][32];
    int e = 0;
    while (e + 1 < t.n && (int)blockIdx.x >= t.first[e + 1]) ++e;
    const cim_bn_part_desc& d = t.d[e];
    const int C = d.channels, parts = d.parts, total = d.images * d.parts;
    const int c = ((int)blockIdx.x - t.first[e]) * 32 + (threadIdx.x & 31), grp = threadIdx.x >> 5;
    float t1 = 0.0f, t2 = 0.0f;
    if (c < C) {
        for (int i0 = grp; i0 < total; i0 += 32) {
            float a[4], b[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = i0 + 8 * j;
                const int img = i / parts, p = i - img * parts;
                const float* q = d.part + ((size_t)img * 2 * parts + p) * C + c;
                a[j] = i < total ? q[0] : 0.0f;
                b[j] = i < total ? q[(size_t)parts * C] : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) { t1 += a[j]; t2 += b[j]; }
        }
    }
    sh[0][grp][threadIdx.x & 31] = t1;
    sh[1][grp][threadIdx.x & 31] = t2;
    __syncthreads();
    if (grp == 0 && c < C) {
        float a1 = 0.0f, a2 = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) { a1 += sh[0][k][threadIdx.x]; a2 += sh[1][k][threadIdx.x]; }
        if (d.dbeta) d.dbeta[c] = a1;
        if (d.dgamma) d.dgamma[c] = a2 * rsqrtf(d.var[c] + d.eps);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// 3 x 3 convolutions (padding 1, stride 1 or 2) of the bottlenecks as IMPLICIT GEMMs on the same tile, MFMA loop and
// epilogue: the im2col matrix is never built, the B loader (and, for the data gradient, the A loader) computes the
// shifted / strided address and the zero-padding predicate per element.  In NCHW, one image, X [Cin][H][W] -> Y
// [Cout][Ho][Wo], W [Cout][Cin][3][3] (the module's own tensor, no repacking), k-order (channel, tap):
//   forward          M = Cout, N = Ho Wo, K = 9 Cin   A = W as [Cout][9 Cin] (K-contiguous)    B(k, n) = X[ci][s yo + dy][s xo + dx]
//   data gradient    M = Cin,  N = H W,   K = 9 Cout  A(m, k) = W[co][m][tap], from a transposed copy  B(k, n) = dY[co][(yi - dy) / s][(xi - dx) / s]
//   weight gradient  M = Cout, N = 9 Cin, K = Ho Wo   A = dY as [Cout][Ho Wo] (K-contiguous)   B(k, n) = X[ci][s yo + dy][s xo + dx]
// (tap = 3 (dy + 1) + (dx + 1); the weight gradient's C is the weight tensor's own layout).  Gathered elements are 4-byte
// loads, 8 per thread and slab, issued together; interior float4 pieces of stride-1 convolutions take one 16-byte load.
enum { CONV_FWD = 0, CONV_DX = 1, CONV_DW = 2 };
struct ConvGeom { int H, W, Ho, Wo, stride, cin, cout; float inv_w; int dil; };      // dil: dilation (padding = dil * (k / 2): "same")

#ifndef CIM_CONV3_BK
#define CIM_CONV3_BK 32            // slab depth of the 3 x 3 kernel (K = 9 C is long: 64 halves the barriers and load round trips per k)
#endif
constexpr int CBK = CIM_CONV3_BK;
template <int MODE, int KS = 3>
__global__ __launch_bounds__(256) void conv3x3_small_kernel(const SmallArgs g, const ConvGeom c) {
    CIM_BODY_PRIO();
    constexpr int TAPS = KS * KS;
    constexpr int NT = 256, PA = SBM * (CBK / 4) / NT, PB = SBN * (CBK / 4) / NT;        // sixteen-byte pieces per thread and slab
    extern __shared__ __attribute__((aligned(16))) float c3_smem[];
    float (*As)[CBK * SLD] = reinterpret_cast<float (*)[CBK * SLD]>(c3_smem);
    float (*Bs)[CBK * SLD] = reinterpret_cast<float (*)[CBK * SLD]>(c3_smem + 2 * CBK * SLD);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (g.N + SBN - 1) / SBN;
    const int m0 = (blockIdx.x / tiles_n) * SBM, n0 = (blockIdx.x % tiles_n) * SBN;
    const int split = blockIdx.y;
    const int kper = ((g.K + g.splits - 1) / g.splits + CBK - 1) / CBK * CBK;
    const int kbeg = split * kper, kend = min(g.K, kbeg + kper);

    // ---- B pieces.  Round 4: every gathered element is ONE branch-free 4-byte buffer load - an element that falls into the
    // zero padding (or behind N / K) takes an offset behind the tensor and the hardware returns 0.  (The first loaders chose
    // per piece between a 16-byte load and four guarded scalar loads: ~400 branches in the kernel, every load in its own basic
    // block with its own wait; the 3 x 3 kernels ran at a fifth of their MFMA time.)
    // forward / data gradient: B is row-contiguous in n (pixels): piece p -> k = k0 + p / 16, n = n0 + (p % 16) * 4: the
    // four pixels are fixed for the whole K loop, (channel, tap) changes per slab.
    // weight gradient: B is K-contiguous: piece p -> n = n0 + p % 64 (a (channel, tap) pair, fixed), k = k0 + (p / 64) * 4 pixels.
    const rsrc_t RA_ = make_rsrc(g.A, tile_extent(MODE == CONV_DX, g.M, g.K, g.lda));
    const rsrc_t RB_ = make_rsrc(g.B, (long long)(MODE == CONV_DX ? c.cout * c.Ho * c.Wo : c.cin * c.H * c.W));
    const int hw_src = MODE == CONV_DX ? c.Ho * c.Wo : c.H * c.W;         // channel stride of the gathered tensor
    int py[PB][4], px[PB][4];            // forward / dX: pixel coordinates of the piece's 4 columns (y = -2^20: column >= N)
    int boff[PB], bdy[PB], bdx[PB];      // dW: channel offset and tap displacement of the piece's row (boff < 0: row >= N)
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int p = tid + i * NT;
        if (MODE == CONV_DW) {
            const int n = n0 + p % SBN;
            const int ch = n / TAPS, tap = n - ch * TAPS;
            boff[i] = n < g.N ? ch * hw_src : -1;
            bdy[i] = (tap / KS - KS / 2) * c.dil;
            bdx[i] = (tap - (tap / KS) * KS - KS / 2) * c.dil;
        } else {
            const int wrow = (MODE == CONV_DX) ? c.W : c.Wo;             // n runs over input pixels (dX) / output pixels (forward)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + (p % 16) * 4 + j;
                const int y = (int)(((float)n + 0.5f) * c.inv_w);       // n / wrow (exact: see the launcher)
                py[i][j] = n < g.N ? y : -(1 << 20);
                px[i][j] = n - y * wrow;
            }
        }
    }
    unsigned oa[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) oa[i] = tile_off<MODE == CONV_DX, SBM>(g.lda, m0, kbeg, tid + i * NT);
    const unsigned sa = (MODE == CONV_DX ? (unsigned)g.lda * CBK : (unsigned)CBK) * 4u;

    auto load_b = [&](int k0, float4 (&rb)[PB]) {
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int p = tid + i * NT;
            unsigned off[4];
            if (MODE == CONV_DW) {
                const int k = k0 + (p / SBN) * 4;                        // 4 consecutive output pixels
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int po = k + j;
                    const int yo = (int)(((float)po + 0.5f) * c.inv_w), xo = po - yo * c.Wo;
                    const int yi = yo * c.stride + bdy[i], xi = xo * c.stride + bdx[i];
                    const bool ok = boff[i] >= 0 && po < g.K && (unsigned)yi < (unsigned)c.H && (unsigned)xi < (unsigned)c.W;
                    off[j] = ok ? (unsigned)(boff[i] + yi * c.W + xi) * 4u : OOB;
                }
            } else {
                const int k = k0 + p / 16;                               // (k >= K: channel >= the tensor's -> behind its extent)
                const int ch = k / TAPS, tap = k - ch * TAPS;
                const int dy = (tap / KS - KS / 2) * c.dil, dx = (tap - (tap / KS) * KS - KS / 2) * c.dil;
                const int base = ch * hw_src;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (MODE == CONV_DX) {         // (y, x) = input pixel; the output pixel that reads it through this tap
                        const int ty = py[i][j] - dy, tx = px[i][j] - dx;
                        const int yo = c.stride == 2 ? ty >> 1 : ty, xo = c.stride == 2 ? tx >> 1 : tx;
                        const bool ok = ty >= 0 && tx >= 0 && yo < c.Ho && xo < c.Wo && !(c.stride == 2 && ((ty | tx) & 1));
                        off[j] = ok ? (unsigned)(base + yo * c.Wo + xo) * 4u : OOB;
                    } else {
                        const int yi = py[i][j] * c.stride + dy, xi = px[i][j] * c.stride + dx;
                        const bool ok = (unsigned)yi < (unsigned)c.H && (unsigned)xi < (unsigned)c.W;
                        off[j] = ok ? (unsigned)(base + yi * c.W + xi) * 4u : OOB;
                    }
                }
            }
#if CIM_SMALL_ABL == 3       /* ablation: no gather loads (the address arithmetic stays) */
            asm volatile("" : "+v"(rb[i].x), "+v"(rb[i].y), "+v"(rb[i].z), "+v"(rb[i].w) : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]));
#elif CIM_SMALL_ABL == 5     /* ablation: gather loads without their address arithmetic */
            rb[i] = make_float4(bld1(RB_, (unsigned)(p * 16 + k0 * 64)), bld1(RB_, (unsigned)(p * 16 + k0 * 64 + 4)),
                                bld1(RB_, (unsigned)(p * 16 + k0 * 64 + 8)), bld1(RB_, (unsigned)(p * 16 + k0 * 64 + 12)));
#else
            rb[i] = make_float4(bld1(RB_, off[0]), bld1(RB_, off[1]), bld1(RB_, off[2]), bld1(RB_, off[3]));
#endif
        }
    };
    auto load_a = [&](int s, float4 (&ra)[PA]) {
#pragma unroll
        for (int i = 0; i < PA; ++i)     // dX: A = the weight transposed to [(co, tap)][ci] (M-contiguous rows); else K-contiguous
            ra[i] = bld4(RA_, oa[i] + (unsigned)s * sa);
    };

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    float4 ra0[PA], rb0[PB], ra1[PA], rb1[PB];
#define C3_PUT(RA, RB, BUF, S)                                                                                      \
    {                                                                                                               \
        _Pragma("unroll") for (int i = 0; i < PA; ++i)                                                              \
            tile_store<MODE == CONV_DX, SBM>(As[BUF], tile_ktail<MODE == CONV_DX, SBM>(RA[i], g.K, kbeg + (S) * CBK, tid + i * NT), tid + i * NT); \
        _Pragma("unroll") for (int i = 0; i < PB; ++i) tile_store<MODE != CONV_DW, SBN>(Bs[BUF], RB[i], tid + i * NT); \
    }
#define C3_MMA(BUF)                                                                                                 \
    {                                                                                                               \
        const float* __restrict__ a = As[BUF] + (lane >> 5) * SLD + wm * 32 + (lane & 31);                          \
        const float* __restrict__ b = Bs[BUF] + (lane >> 5) * SLD + wn * 32 + (lane & 31);                          \
        float av[CBK / 2], bv[CBK / 2];                                                                             \
        _Pragma("unroll") for (int t = 0; t < CBK / 2; ++t) { av[t] = a[2 * t * SLD]; bv[t] = b[2 * t * SLD]; }     \
        _Pragma("unroll") for (int t = 0; t < (CIM_SMALL_ABL == 1 ? 1 : CBK / 2); ++t)                              \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[t], acc, 0, 0, 0);                                 \
        asm volatile("" : "+a"(acc));                                                                               \
    }
    if (kbeg < kend) {         // two register sets of pieces in flight, as gemm_small_kernel
        const int nslab = (kend - kbeg + CBK - 1) / CBK;
        const int last = nslab - 1;      // (loads unconditional, the slab index clamped: see gemm_small_kernel)
        load_a(0, ra0); load_b(kbeg, rb0);
        load_a(min(1, last), ra1); load_b(kbeg + min(1, last) * CBK, rb1);
        C3_PUT(ra0, rb0, 0, 0)
        asm volatile("" : "+a"(acc));
        __syncthreads();
        for (int s = 0; s < nslab; s += 2) {
            load_a(min(s + 2, last), ra0); load_b(kbeg + min(s + 2, last) * CBK, rb0);
            C3_MMA(0)
            if (s + 1 < nslab) C3_PUT(ra1, rb1, 1, s + 1)
            __syncthreads();
            if (s + 1 >= nslab) break;
            load_a(min(s + 3, last), ra1); load_b(kbeg + min(s + 3, last) * CBK, rb1);
            C3_MMA(1)
            if (s + 2 < nslab) C3_PUT(ra0, rb0, 0, s + 2)
            __syncthreads();
        }
    }
#undef C3_PUT
#undef C3_MMA
#if CIM_SMALL_ABL == 2
    if (m0 + wm * 32 + lane < 0)
#endif
    small_epilogue(g, acc, m0 + wm * 32, n0 + wn * 32, lane, split);
}

// ------------------------------------------------------------------------------------------------------------------
// Round 6: the data gradient of a STRIDE-2 3 x 3 convolution (the first bottleneck of res3 / res4: torchvision v1.5 puts the
// stride on conv2) by PARITY CLASSES of the input pixels.  conv3x3_small_kernel<CONV_DX> walks all 9 taps for every input pixel
// and gathers zeros for the taps whose parity does not match: (yi + 1 - ky) must be even, so an even row only sees ky = 1 and an
// odd row ky = 0, 2 (likewise columns) - 9 of the 36 (tap, pixel class) pairs carry data, the other three quarters of the
// MFMAs multiplied zeros (profiles/r6: 85-104 us against 33 us for the stride-1 layers of the same size).  Here the pixels are
// enumerated class by class (py, px = parity of the row / column; tiles never mix classes) and a class contracts over ITS taps
// only:  K' = cout x {1, 2, 2, 4},  k' = co T + j,  tap j of class -> (ky, kx);  the pixel (yc, xc) of a class is the input
// pixel (2 yc + py, 2 xc + px) and reads dY[co][yc + oy][xc + ox] with oy = [py and ky == 0], ox = [px and kx == 0].
// No split-K (the longest class walks 4 cout / 32 slabs: 16 / 32 at res3 / res4) and no reduce launch; the epilogue (producer's
// BatchNorm + ReLU backward, its affine partial sums) is small_epilogue's with the column mapped back to the map:
// partial-sum groups are numbered class by class (cim_conv3x3_dx_parts).
struct Dx2Geom { int H, W, Ho, Wo, cin, cout; int first[5]; int tiles_n[4]; int group0[4]; };
struct ColMap { int Wc, W, py, px, Nc, group0; };

__device__ __forceinline__ void small_epilogue_mapped(const SmallArgs& g, const f32x16& acc, int row0, int col0, int lane, const ColMap& cm) {
    const int nc = col0 + (lane & 31);
    const bool in = nc < cm.Nc;
    const int yc = nc / cm.Wc, xc = nc - yc * cm.Wc;
    const int col = (2 * yc + cm.py) * cm.W + 2 * xc + cm.px;
    if (g.mpart == nullptr) {
        if (in) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
                if (row < g.M) small_finish(g, row, col, acc[r]);
            }
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = row0 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
        float s1 = 0.0f, s2 = 0.0f;
        if (in && row < g.M) small_finish(g, row, col, acc[r], s1, s2);
        small_put_part(g, row, cm.group0 + (col0 >> 5), lane, s1, s2);
    }
}

__global__ __launch_bounds__(256) void conv3x3_dx2_kernel(const SmallArgs g, const Dx2Geom c) {
    CIM_BODY_PRIO();
    constexpr int NT = 256, PA = SBM * (CBK / 4) / NT, PB = SBN * (CBK / 4) / NT;
    extern __shared__ __attribute__((aligned(16))) float d2_smem[];
    float (*As)[CBK * SLD] = reinterpret_cast<float (*)[CBK * SLD]>(d2_smem);
    float (*Bs)[CBK * SLD] = reinterpret_cast<float (*)[CBK * SLD]>(d2_smem + 2 * CBK * SLD);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int cls = 0;
    while (cls < 3 && (int)blockIdx.x >= c.first[cls + 1]) ++cls;
    const int py = cls >> 1, px = cls & 1;
    const int Wc = (c.W - px + 1) >> 1, Nc = ((c.H - py + 1) >> 1) * Wc;
    const int local = (int)blockIdx.x - c.first[cls], tn = c.tiles_n[cls];
    const int m0 = (local / tn) * SBM, n0 = (local % tn) * SBN;
    const int nx = px ? 2 : 1, lgT = py + px, T = 1 << lgT;
    const int K = c.cout << lgT;                       // k' = co T + j
    const int nslab = (K + CBK - 1) / CBK;
    const int how = c.Ho * c.Wo;
    const rsrc_t RA_ = make_rsrc(g.A, tile_extent(true, g.M, 9 * c.cout, g.lda));
    const rsrc_t RB_ = make_rsrc(g.B, (long long)c.cout * how);
    // the four class pixels of each B piece (fixed over the K loop); yc < 0: behind the class
    int yc[PB][4], xc[PB][4];
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int p = tid + i * NT;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nc = n0 + (p % 16) * 4 + j;
            const int y = nc / Wc;
            yc[i][j] = nc < Nc ? y : -(1 << 20);
            xc[i][j] = nc - y * Wc;
        }
    }
    // tap j of this class: (ky, kx) and the displacement of the output pixel it reads
    auto tap_of = [&](int j, int& tap, int& oy, int& ox) {
        const int jy = j / nx, jx = j - jy * nx;
        const int ky = py ? 2 * jy : 1, kx = px ? 2 * jx : 1;
        tap = ky * 3 + kx;
        oy = (py && jy == 0) ? 1 : 0;
        ox = (px && jx == 0) ? 1 : 0;
    };
    auto load_a = [&](int k0, float4 (&ra)[PA]) {
#pragma unroll
        for (int i = 0; i < PA; ++i) {               // A = the weight transposed to [(co, tap)][ci]: rows of ci, k' -> row co 9 + tap
            const int p = tid + i * NT;
            const int k = k0 + p / (SBM / 4), r = m0 + (p % (SBM / 4)) * 4;
            const int co = k >> lgT;
            int tap, oy, ox;
            tap_of(k & (T - 1), tap, oy, ox);
            ra[i] = bld4(RA_, co < c.cout ? ((unsigned)(co * 9 + tap) * g.lda + r) * 4u : OOB);
        }
    };
    auto load_b = [&](int k0, float4 (&rb)[PB]) {
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int p = tid + i * NT;
            const int k = k0 + p / 16;
            const int co = k >> lgT;
            int tap, oy, ox;
            tap_of(k & (T - 1), tap, oy, ox);
            const int base = co * how;
            unsigned off[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int yo = yc[i][j] + oy, xo = xc[i][j] + ox;
                const bool ok = co < c.cout && yo >= 0 && yo < c.Ho && xo < c.Wo;
                off[j] = ok ? (unsigned)(base + yo * c.Wo + xo) * 4u : OOB;
            }
            rb[i] = make_float4(bld1(RB_, off[0]), bld1(RB_, off[1]), bld1(RB_, off[2]), bld1(RB_, off[3]));
        }
    };
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    float4 ra0[PA], rb0[PB], ra1[PA], rb1[PB];
#define D2_PUT(RA, RB, BUF)                                                                                         \
    {                                                                                                               \
        _Pragma("unroll") for (int i = 0; i < PA; ++i) tile_store<true, SBM>(As[BUF], RA[i], tid + i * NT);          \
        _Pragma("unroll") for (int i = 0; i < PB; ++i) tile_store<true, SBN>(Bs[BUF], RB[i], tid + i * NT);          \
    }
#define D2_MMA(BUF)                                                                                                 \
    {                                                                                                               \
        const float* __restrict__ a = As[BUF] + (lane >> 5) * SLD + wm * 32 + (lane & 31);                          \
        const float* __restrict__ b = Bs[BUF] + (lane >> 5) * SLD + wn * 32 + (lane & 31);                          \
        float av[CBK / 2], bv[CBK / 2];                                                                             \
        _Pragma("unroll") for (int t = 0; t < CBK / 2; ++t) { av[t] = a[2 * t * SLD]; bv[t] = b[2 * t * SLD]; }     \
        _Pragma("unroll") for (int t = 0; t < CBK / 2; ++t)                                                         \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[t], acc, 0, 0, 0);                                 \
        asm volatile("" : "+a"(acc));                                                                               \
    }
    {
        const int last = nslab - 1;      // (loads unconditional, the slab index clamped: see gemm_small_kernel)
        load_a(0, ra0); load_b(0, rb0);
        load_a(min(1, last) * CBK, ra1); load_b(min(1, last) * CBK, rb1);
        D2_PUT(ra0, rb0, 0)
        asm volatile("" : "+a"(acc));
        __syncthreads();
        for (int s = 0; s < nslab; s += 2) {
            load_a(min(s + 2, last) * CBK, ra0); load_b(min(s + 2, last) * CBK, rb0);
            D2_MMA(0)
            if (s + 1 < nslab) D2_PUT(ra1, rb1, 1)
            __syncthreads();
            if (s + 1 >= nslab) break;
            load_a(min(s + 3, last) * CBK, ra1); load_b(min(s + 3, last) * CBK, rb1);
            D2_MMA(1)
            if (s + 2 < nslab) D2_PUT(ra0, rb0, 0)
            __syncthreads();
        }
    }
#undef D2_PUT
#undef D2_MMA
    const ColMap cm{Wc, c.W, py, px, Nc, c.group0[cls]};
    small_epilogue_mapped(g, acc, m0 + wm * 32, n0 + wn * 32, lane, cm);
}

// w [Cout][Cin][9] -> wt [Cout][9][Cin] (the data gradient's A operand, M-contiguous): one workgroup per output channel,
// through LDS so that both sides are coalesced.  (Gathering W[co][ci][tap] in the GEMM's loader - 36-byte strides between
// lanes - made the data gradient 1.7x slower than the forward: 69 vs 40 us.)
__global__ __launch_bounds__(256) void conv3x3_wt_kernel(const float* __restrict__ w, float* __restrict__ wt, int cin) {
    extern __shared__ float wt_s[];
    const int co = blockIdx.x, n = cin * 9;
    for (int i = threadIdx.x; i < n; i += 256) wt_s[i] = w[(size_t)co * n + i];
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) {
        const int tap = i / cin, ci = i - tap * cin;
        wt[(size_t)co * n + i] = wt_s[ci * 9 + tap];
    }
}

// The same for up to 16 layers in ONE launch (the transposes of a whole body, made ahead of the backward pass): workgroup ->
// (layer, output channel) through the running channel counts `first`.
struct WtTable { cim_wt_desc d[16]; int first[17]; int n; };
__global__ __launch_bounds__(256) void conv3x3_wt_multi_kernel(const WtTable t) {
    extern __shared__ float wt_s[];
    int e = 0;
    while (e + 1 < t.n && (int)blockIdx.x >= t.first[e + 1]) ++e;
    const int co = blockIdx.x - t.first[e], cin = t.d[e].cin, n = cin * 9;
    const float* __restrict__ w = t.d[e].w;
    float* __restrict__ wt = t.d[e].wt;
    for (int i = threadIdx.x; i < n; i += 256) wt_s[i] = w[(size_t)co * n + i];
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) {
        const int tap = i / cin, ci = i - tap * cin;
        wt[(size_t)co * n + i] = wt_s[ci * 9 + tap];
    }
}

static void set_input_bn(SmallArgs& g, const InputBn* in_bn) {
    if (in_bn == nullptr) return;
    g.mask = in_bn->y; g.mgamma = in_bn->gamma; g.mvar = in_bn->var; g.meps = in_bn->eps;
    g.mxr = in_bn->xr; g.mmean = in_bn->mean; g.mpart = in_bn->part; g.mparts = (g.N + 31) / 32;
}
static void launch_splitk_reduce(const SmallArgs& g, hipStream_t st) {
    if (g.mpart) {
        hipLaunchKernelGGL(small_splitk_reduce_rows_kernel, dim3((unsigned)((g.N + 255) / 256), (unsigned)g.M), dim3(256), 0, st, g);
    } else {
        const size_t n = (size_t)g.M * g.N;
        // (four elements per thread pay from ~0.5 M elements: 8.5 -> 6.8 us at 1.4 M, 7.5 -> 5.6 us at 1.45 M; small results with
        // many splits - the weight gradients: 16 K elements x 64 splits - are a latency chain per thread and want MORE threads:
        // 5.6 -> 9.5 us with four elements each)
        if (n >= (1u << 19) && g.ldc == g.N && n % 4 == 0 && (((size_t)g.ws | (size_t)g.C | (size_t)g.Xraw) & 15) == 0)
            hipLaunchKernelGGL(small_splitk_reduce4_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, g);
        else
            hipLaunchKernelGGL(small_splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g);
    }
}

template <bool AM, bool BKc>
static void launch_small(const SmallArgs& g, int splits, hipStream_t st, bool narrow) {
    const long long tm = (g.M + SBM - 1) / SBM;
    if (narrow)
        hipLaunchKernelGGL((gemm_small_kernel<AM, BKc, 1>), dim3((unsigned)(tm * ((g.N + 31) / 32)), (unsigned)splits), dim3(128), 0, st, g);
    else
        hipLaunchKernelGGL((gemm_small_kernel<AM, BKc, 2>), dim3((unsigned)(tm * ((g.N + 63) / 64)), (unsigned)splits), dim3(256), 0, st, g);
}

}  // namespace

extern "C" int cim_gemm_small_splits(int M, int N, int K) {
    // A workgroup walks its K range slab by slab, one global-load latency (~1.5 us) per 32-k slab when it is alone on its CU:
    // long-K products with few output tiles (res4 conv1: 92 tiles x 32 slabs = 41 us) are cut until there are ~2
    // workgroups per CU, never below 4 slabs per workgroup.
    const long long tiles = (long long)((M + SBM - 1) / SBM) * ((N + SBN - 1) / SBN);
    // (sweep, tools/bench_gemm_small.py, us over 7 layer shapes forward / dX / dW: 157 / 136 / 182 with these limits; 2 slabs
    // per workgroup or up to 768-1024 workgroups: 188 / 178 / 219 - more partial products than the latency chain gains)
#ifndef CIM_SMALL_WGS
#define CIM_SMALL_WGS 512          // (compile-time sweep switch, tools/build_alt.sh)
#endif
#ifndef CIM_SMALL_MINSLABS
#define CIM_SMALL_MINSLABS 4
#endif
    constexpr int min_k = CIM_SMALL_MINSLABS * SBK, want = CIM_SMALL_WGS;
    int s = 1;
    while (tiles * s < want && K / (s * 2) >= min_k && s < 64) s *= 2;
    return s;
}

static int gemm_small_impl(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                           int a_mcontig, int b_kcontig, float* x_raw, const float* gamma, const float* beta,
                           const float* mean, const float* var, float eps, const float* residual, int relu, int splits,
                           float* workspace, void* stream, const InputBn* in_bn, int res_w = 0);

extern "C" int cim_gemm_small_f32(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                                  int a_mcontig, int b_kcontig, float* x_raw, const float* gamma, const float* beta,
                                  const float* mean, const float* var, float eps, const float* residual, int relu, int splits,
                                  float* workspace, void* stream) {
    return gemm_small_impl(A, B, C, M, N, K, lda, ldb, ldc, a_mcontig, b_kcontig, x_raw, gamma, beta, mean, var, eps, residual, relu,
                           splits, workspace, stream, nullptr);
}

static int gemm_small_impl(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                           int a_mcontig, int b_kcontig, float* x_raw, const float* gamma, const float* beta,
                           const float* mean, const float* var, float eps, const float* residual, int relu, int splits,
                           float* workspace, void* stream, const InputBn* in_bn, int res_w) {
    CIM_CHECK_ARG(res_w == 0 || (residual && res_w > 0 && N % res_w == 0 && ldc == N));
    CIM_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0 && lda > 0 && ldb > 0 && ldc >= N && splits >= 1);
    CIM_CHECK_ARG((gamma == nullptr) == (beta == nullptr) && (gamma == nullptr) == (mean == nullptr) && (gamma == nullptr) == (var == nullptr));
    CIM_CHECK_ARG(splits == 1 || workspace);
    SmallArgs g{};
    g.A = A; g.B = B; g.C = C; g.Xraw = x_raw;
    g.gamma = gamma; g.beta = beta; g.mean = mean; g.var = var; g.res = residual; g.eps = eps;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.a_mcontig = a_mcontig; g.b_kcontig = b_kcontig; g.relu = relu; g.bn = gamma != nullptr;
    g.splits = splits; g.ws = workspace; g.colbias = nullptr;
    g.res_w = res_w;
    if (res_w) { g.res_ws = (res_w + 1) / 2; g.res_hw = ((N / res_w + 1) / 2) * g.res_ws; }
    set_input_bn(g, in_bn);
    const long long tiles = (long long)((M + SBM - 1) / SBM) * ((N + SBN - 1) / SBN);
    CIM_CHECK_ARG(tiles * 2 < (1ll << 31) && splits <= 65535 && M <= 65535);
    // (the loaders address the operands through buffer resources with 32-bit byte offsets; OOB = 2^31 - 1 must lie behind them)
    CIM_CHECK_ARG(tile_extent(a_mcontig != 0, M, K, lda) < (1ll << 29) && tile_extent(b_kcontig == 0, N, K, ldb) < (1ll << 29));
    const bool narrow = tiles * splits < 128;           // 64 x 32 tiles only for problems that cannot fill the chip otherwise
    hipStream_t st = cim::as_stream(stream);
    if (a_mcontig) {
        if (b_kcontig) launch_small<true, true>(g, splits, st, narrow); else launch_small<true, false>(g, splits, st, narrow);
    } else {
        if (b_kcontig) launch_small<false, true>(g, splits, st, narrow); else launch_small<false, false>(g, splits, st, narrow);
    }
    if (splits > 1) launch_splitk_reduce(g, st);
    CIM_CHECK_LAUNCH();
    return 0;
}

// Y[M][N] = X[M][K] . W[N][K]^T + bias[N]  (nn.Linear on the small-tile fp32-MFMA GEMM): the eight scoring heads of
// lib/modeling/heads.py:194-219 as ONE product against their concatenated weights (N = 8 (C + 1) = 168 / 648 columns).
extern "C" int cim_linear_bias_f32(const float* X, const float* W, const float* bias, float* Y, int M, int N, int K, int splits,
                                   float* workspace, void* stream) {
    CIM_CHECK_ARG(X && W && Y && M > 0 && N > 0 && K > 0 && splits >= 1 && splits <= 65535 && (splits == 1 || workspace));
    SmallArgs g{};
    g.A = X; g.B = W; g.C = Y; g.Xraw = nullptr;
    g.gamma = g.beta = g.mean = g.var = g.res = nullptr; g.eps = 0.f;
    g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.ldc = N;
    g.a_mcontig = 0; g.b_kcontig = 1; g.relu = 0; g.bn = 0;
    g.splits = splits; g.ws = workspace; g.colbias = bias;
    hipStream_t st = cim::as_stream(stream);
    launch_small<false, true>(g, splits, st, false);
    if (splits > 1) {
        const size_t n = (size_t)M * N;
        hipLaunchKernelGGL(small_splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g);
    }
    CIM_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// The whole backward of conv1x1 -> BatchNorm (+ residual) (+ ReLU) for a batch of images in ONE call: BatchNorm / ReLU
// backward (bn_act.hip), data gradient and weight gradient GEMMs with their split-K reduces - 3 to 5 launches enqueued by
// one host call instead of a dozen Python-level operations (the backbone backward was host-bound: 1.07 ms of idle GPU
// time in 3.6 ms at cfg2 with the operations issued one by one).
extern "C" long long cim_conv1x1_bwd_workspace(int B, int cin, int cout, int hw) {
    const long long dx = (long long)cim_gemm_small_splits(cin, hw, cout) * cin * hw;
    const long long dw = (long long)cim_gemm_small_splits(cout, cin, hw) * cout * cin + (B > 1 ? (long long)cout * cin : 0);
    const long long dconv = (long long)B * cout * hw;
    return (long long)sizeof(float) * (dconv + dx + dw);      // (dX and dW run side by side: separate split-K areas)
}

namespace {
// Fork / join between the caller's stream and a side stream inside one host call: the weight-gradient GEMM of a layer
// runs on the side stream next to the data-gradient GEMM (both only read the BatchNorm backward's output; each fills a
// fraction of the chip - they are latency bound).  The two events are the CALLER's (hipEvent_t created with
// hipEventDisableTiming, passed as void*): the library creates nothing.  They are recorded and waited on inside the call, so the
// caller may hand the same pair to its next call on the same streams.
struct ForkJoin {
    hipStream_t main, side;
    hipEvent_t ev_fork, ev_join;
    bool on;
    ForkJoin(hipStream_t m, hipStream_t s, void* fork_event, void* join_event, bool want)
        : main(m), side(s), ev_fork(static_cast<hipEvent_t>(fork_event)), ev_join(static_cast<hipEvent_t>(join_event)),
          on(want && s != nullptr && s != m) {}
    hipStream_t fork() {                 // side stream, ordered after everything enqueued on main so far
        if (!on) return main;
        (void)hipEventRecord(ev_fork, main);
        (void)hipStreamWaitEvent(side, ev_fork, 0);
        return side;
    }
    void join() {                        // main waits for the side stream's work
        if (!on) return;
        (void)hipEventRecord(ev_join, side);
        (void)hipStreamWaitEvent(main, ev_join, 0);
    }
};

__global__ __launch_bounds__(256) void small_axpy_kernel(float* __restrict__ y, const float* __restrict__ x, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] += x[i];
}
}  // namespace

extern "C" int cim_conv1x1_bn_act_bwd(const float* dy, const float* y, const float* x_raw, const float* x, const float* w,
                                      const float* gamma, const float* mean, const float* var, float eps, int relu,
                                      float* dres, float* dgamma, float* dbeta, float* dx, float* dw, int B, int cin, int cout,
                                      int hw, float* workspace, void* stream, void* side_stream, void* fork_event, void* join_event, int join,
                                      int dy_is_dconv, const float* in_gamma, const float* in_var, float in_eps,
                                      const float* in_xr, const float* in_mean, float* in_part, const float* dx_add, int dx_add_w) {
    CIM_CHECK_ARG(dx_add_w == 0 || (dx_add != nullptr && dx_add_w > 0 && hw % dx_add_w == 0));
    CIM_CHECK_ARG(side_stream == nullptr || side_stream == stream || (fork_event != nullptr && (!join || join_event != nullptr)));
    const size_t add_bs = dx_add_w ? (size_t)cin * ((hw / dx_add_w + 1) / 2) * ((dx_add_w + 1) / 2) : (size_t)cin * hw;
    CIM_CHECK_ARG(dy && x_raw && x && w && gamma && mean && var && workspace && B > 0 && cin > 0 && cout > 0 && hw > 0);
    CIM_CHECK_ARG(dx_add == nullptr || (dx != nullptr && in_gamma == nullptr));      // (a second branch's gradient of x, added in the epilogue)
    CIM_CHECK_ARG((y != nullptr || !relu) && ((dgamma == nullptr) == (dbeta == nullptr)));
    // (dy_is_dconv: the consumer's data gradient already applied this layer's BatchNorm + ReLU backward; its affine gradients
    // come from the partial sums that product left: cim_bn_part_finish)
    CIM_CHECK_ARG(!dy_is_dconv || (dres == nullptr && dgamma == nullptr));
    CIM_CHECK_ARG((in_gamma == nullptr) == (in_var == nullptr));
    CIM_CHECK_ARG(in_part == nullptr || (in_gamma && in_xr && in_mean));
    float* dconv = dy_is_dconv ? const_cast<float*>(dy) : workspace;             // [B][cout][hw]: dz * a, the gradient of the convolution output
    float* ws_dx = workspace + (size_t)B * cout * hw;
    float* ws_dw = ws_dx + (size_t)cim_gemm_small_splits(cin, hw, cout) * cin * hw;
    int rc = dy_is_dconv ? 0 : cim_bn_act_bwd(dy, y, x_raw, gamma, mean, var, eps, dconv, dres, dgamma, dbeta, B, cout, hw, relu, stream);
    if (rc) return rc;
    ForkJoin fj(cim::as_stream(stream), cim::as_stream(side_stream), fork_event, join_event, dx && dw);
    void* st_dw = fj.fork();                                   // the weight gradient next to the data gradient
    for (int b = 0; b < B && dw; ++b) {                        // dW[cout, cin] = dconv . X^T  (K = hw)
        const int sp = cim_gemm_small_splits(cout, cin, hw);
        float* out = b == 0 ? dw : ws_dw + (size_t)sp * cout * cin;
        rc = cim_gemm_small_f32(dconv + (size_t)b * cout * hw, x + (size_t)b * cin * hw, out, cout, cin, hw, hw, hw, cin, 0, 1,
                                nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, nullptr, 0, sp, ws_dw, st_dw);
        if (rc) return rc;
        if (b) {
            const size_t n = (size_t)cout * cin;
            hipLaunchKernelGGL(small_axpy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, cim::as_stream(st_dw), dw, out, n);
        }
    }
    for (int b = 0; b < B && dx; ++b) {                        // dX[cin, hw] = W^T . dconv  (x the BatchNorm + ReLU backward of the layer that made x)
        const InputBn ib{x + (size_t)b * cin * hw, in_gamma, in_var, in_eps, in_xr ? in_xr + (size_t)b * cin * hw : nullptr, in_mean,
                         in_part ? in_part + (size_t)b * 2 * ((hw + 31) / 32) * cin : nullptr};
        rc = gemm_small_impl(w, dconv + (size_t)b * cout * hw, dx + (size_t)b * cin * hw, cin, hw, cout, cin, hw, hw, 1, 0,
                             nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, dx_add ? dx_add + (size_t)b * add_bs : nullptr, 0,
                             cim_gemm_small_splits(cin, hw, cout), ws_dx, stream, in_gamma ? &ib : nullptr, dx_add_w);
        if (rc) return rc;
    }
    if (join) fj.join();                                       // else the caller joins the side stream before the weight gradient is used
    CIM_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// 3 x 3 convolution (padding 1, stride 1 / 2, no bias) + BatchNorm (eval statistics) (+ residual) (+ ReLU), one image per
// launch; see conv3x3_small_kernel.
namespace {
int conv3x3_launch(int mode, const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldc, const ConvGeom& c,
                   float* x_raw, const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                   const float* residual, int relu, int splits, float* ws, hipStream_t st, int ksize = 3, const InputBn* in_bn = nullptr) {
    SmallArgs g{};
    g.A = A; g.B = B; g.C = C; g.Xraw = x_raw;
    g.gamma = gamma; g.beta = beta; g.mean = mean; g.var = var; g.res = residual; g.eps = eps;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = 0; g.ldc = ldc;
    g.a_mcontig = 0; g.b_kcontig = mode == CONV_DW; g.relu = relu; g.bn = gamma != nullptr;
    g.splits = splits; g.ws = ws; g.colbias = nullptr;
    set_input_bn(g, in_bn);
    const dim3 grid((unsigned)(((M + SBM - 1) / SBM) * ((N + SBN - 1) / SBN)), (unsigned)splits);
    const size_t lds = sizeof(float) * 4 * CBK * SLD;
    auto kern = ksize == 7 ? conv3x3_small_kernel<CONV_FWD, 7>
                : mode == CONV_FWD ? conv3x3_small_kernel<CONV_FWD> : mode == CONV_DX ? conv3x3_small_kernel<CONV_DX> : conv3x3_small_kernel<CONV_DW>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, g, c);
    if (splits > 1) launch_splitk_reduce(g, st);
    return 0;
}
ConvGeom conv_geom(int cin, int cout, int H, int W, int stride, int mode, int dil = 1) {
    ConvGeom c;
    c.H = H; c.W = W; c.stride = stride; c.cin = cin; c.cout = cout; c.dil = dil;
    c.Ho = (H - 1) / stride + 1; c.Wo = (W - 1) / stride + 1;            // odd kernel k, padding k / 2
    c.inv_w = 1.0f / (float)(mode == CONV_DX ? W : c.Wo);
    return c;
}
}  // namespace

#define CONV3_ARGS_OK(CIN_MULT)                                                                                      \
    CIM_CHECK_ARG(cin > 0 && cout > 0 && H > 0 && W > 0 && (stride == 1 || stride == 2) && W <= 4096 &&               \
                  dilation >= 1 && dilation <= 8 && (dilation == 1 || stride == 1) &&                                 \
                  (long long)H * W < (1ll << 20) && (long long)cin * H * W < (1ll << 29) && (long long)cout * H * W < (1ll << 29) && \
                  (long long)cin * cout * 9 < (1ll << 29) && cin % (CIN_MULT) == 0)       /* (2^29 floats: 32-bit byte offsets of the buffer loads) */

extern "C" int cim_conv3x3_nchw_splits(int cin, int cout, int H, int W, int stride) {
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    return cim_gemm_small_splits(cout, Ho * Wo, 9 * cin);
}

extern "C" int cim_conv3x3_nchw_f32(const float* x, const float* w, float* y, int cin, int cout, int H, int W, int stride,
                               int dilation, float* x_raw, const float* gamma, const float* beta, const float* mean, const float* var,
                               float eps, const float* residual, int relu, int splits, float* workspace, void* stream) {
    CIM_CHECK_ARG(x && w && y && splits >= 1 && splits <= 65535 && (splits == 1 || workspace));
    CONV3_ARGS_OK(1);                 // (forward: any cin - the RGB stems of VGG16 / HRNet are frozen, forward only)
    CIM_CHECK_ARG((gamma == nullptr) == (beta == nullptr) && (gamma == nullptr) == (mean == nullptr) && (gamma == nullptr) == (var == nullptr));
    const ConvGeom c = conv_geom(cin, cout, H, W, stride, CONV_FWD, dilation);
    conv3x3_launch(CONV_FWD, w, x, y, cout, c.Ho * c.Wo, 9 * cin, 9 * cin, c.Ho * c.Wo, c, x_raw, gamma, beta, mean, var, eps,
                   residual, relu, splits, workspace, cim::as_stream(stream));
    CIM_CHECK_LAUNCH();
    return 0;
}

// The stem: 7 x 7 convolution (padding 3, stride 1 / 2, no bias, any cin) -> frozen BatchNorm (+ ReLU), forward only (the
// reference freezes it: FREEZE_AT >= 1); the same implicit GEMM with 49 taps (K = 49 cin; A = the weight as it is).
extern "C" int cim_conv7x7_nchw_f32(const float* x, const float* w, float* y, int cin, int cout, int H, int W, int stride,
                                    const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                                    int relu, void* stream) {
    CIM_CHECK_ARG(x && w && y && cin > 0 && cout > 0 && H > 0 && W > 0 && (stride == 1 || stride == 2) && W <= 4096);
    CIM_CHECK_ARG((long long)H * W < (1ll << 20) && (long long)cin * H * W < (1ll << 29) && (long long)cout * H * W < (1ll << 29));
    CIM_CHECK_ARG((gamma == nullptr) == (beta == nullptr) && (gamma == nullptr) == (mean == nullptr) && (gamma == nullptr) == (var == nullptr));
    const ConvGeom c = conv_geom(cin, cout, H, W, stride, CONV_FWD);
    conv3x3_launch(CONV_FWD, w, x, y, cout, c.Ho * c.Wo, 49 * cin, 49 * cin, c.Ho * c.Wo, c, nullptr, gamma, beta, mean, var, eps,
                   nullptr, relu, 1, nullptr, cim::as_stream(stream), 7);
    CIM_CHECK_LAUNCH();
    return 0;
}

namespace {
// pixel classes of the stride-2 data gradient (conv3x3_dx2_kernel): class = 2 py + px
Dx2Geom dx2_geom(int cin, int cout, int H, int W) {
    Dx2Geom c{};
    c.H = H; c.W = W; c.Ho = (H - 1) / 2 + 1; c.Wo = (W - 1) / 2 + 1; c.cin = cin; c.cout = cout;
    const int tiles_m = (cin + SBM - 1) / SBM;
    int first = 0, group = 0;
    for (int cls = 0; cls < 4; ++cls) {
        const int py = cls >> 1, px = cls & 1;
        const int nc = ((H - py + 1) / 2) * ((W - px + 1) / 2);
        c.first[cls] = first;
        c.tiles_n[cls] = (nc + SBN - 1) / SBN;
        c.group0[cls] = group;
        first += tiles_m * c.tiles_n[cls];
        group += 2 * c.tiles_n[cls];                // two 32-column groups per tile
    }
    c.first[4] = first;
    return c;
}
}  // namespace

// 32-pixel groups of the affine partial sums a data gradient's epilogue leaves per channel (in_part: [B][2][parts][cin]): stride 1 -
// ceil(H W / 32) consecutive pixels; stride 2 - the groups of conv3x3_dx2_kernel's pixel classes (tile padded).
extern "C" int cim_conv3x3_dx_parts(int H, int W, int stride) {
    if (stride != 2) return (H * W + 31) / 32;
    int parts = 0;
    for (int cls = 0; cls < 4; ++cls) parts += 2 * ((((H - (cls >> 1) + 1) / 2) * ((W - (cls & 1) + 1) / 2) + SBN - 1) / SBN);
    return parts;
}

extern "C" long long cim_conv3x3_nchw_bwd_workspace(int B, int cin, int cout, int H, int W, int stride) {
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const long long dx = (long long)cim_gemm_small_splits(cin, H * W, 9 * cout) * cin * H * W;
    const long long dw = (long long)cim_gemm_small_splits(cout, 9 * cin, Ho * Wo) * cout * cin * 9 + (B > 1 ? (long long)cout * cin * 9 : 0);
    const long long dconv = (long long)B * cout * Ho * Wo;
    return (long long)sizeof(float) * (dconv + (long long)cout * cin * 9 + dx + dw);
}

extern "C" int cim_bn_part_finish(const cim_bn_part_desc* descs, int n, void* stream) {
    CIM_CHECK_ARG(descs && n > 0);
    hipStream_t st = cim::as_stream(stream);
    for (int base = 0; base < n; base += 24) {
        BnPartTable t{};
        t.n = n - base < 24 ? n - base : 24;
        int total = 0;
        for (int i = 0; i < t.n; ++i) {
            const cim_bn_part_desc& d = descs[base + i];
            CIM_CHECK_ARG(d.part && d.var && (d.dgamma || d.dbeta) && d.images > 0 && d.parts > 0 && d.channels > 0);
            t.d[i] = d;
            t.first[i] = total;
            total += (d.channels + 31) / 32;
        }
        t.first[t.n] = total;
        hipLaunchKernelGGL(bn_part_finish_kernel, dim3((unsigned)total), dim3(256), 0, st, t);
    }
    CIM_CHECK_LAUNCH();
    return 0;
}

extern "C" int cim_conv3x3_wt_multi(const cim_wt_desc* descs, int n, void* stream) {
    CIM_CHECK_ARG(descs && n > 0);
    hipStream_t st = cim::as_stream(stream);
    for (int base = 0; base < n; base += 16) {
        WtTable t{};
        t.n = n - base < 16 ? n - base : 16;
        int total = 0, max_cin = 0;
        for (int i = 0; i < t.n; ++i) {
            const cim_wt_desc& d = descs[base + i];
            CIM_CHECK_ARG(d.w && d.wt && d.cin > 0 && d.cout > 0 && (size_t)d.cin * 9 * sizeof(float) <= 64 * 1024);
            t.d[i] = d;
            t.first[i] = total;
            total += d.cout;
            max_cin = d.cin > max_cin ? d.cin : max_cin;
        }
        t.first[t.n] = total;
        const size_t lds = sizeof(float) * max_cin * 9;
        if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wt_multi_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL(conv3x3_wt_multi_kernel, dim3((unsigned)total), dim3(256), lds, st, t);
    }
    CIM_CHECK_LAUNCH();
    return 0;
}

// The whole backward of conv3x3 -> BatchNorm (+ residual) (+ ReLU): BatchNorm / ReLU backward (bn_act.hip), data gradient
// and weight gradient implicit GEMMs with their split-K reduces, enqueued by one host call (as cim_conv1x1_bn_act_bwd).
extern "C" int cim_conv3x3_nchw_bn_act_bwd(const float* dy, const float* y, const float* x_raw, const float* x, const float* w,
                                      const float* gamma, const float* mean, const float* var, float eps, int relu,
                                      float* dres, float* dgamma, float* dbeta, float* dx, float* dw, int B, int cin, int cout,
                                      int H, int W, int stride, int dilation, float* workspace, void* stream, void* side_stream,
                                      void* fork_event, void* join_event, int join, int dy_is_dconv, const float* in_gamma, const float* in_var, float in_eps,
                                      const float* in_xr, const float* in_mean, float* in_part, const float* wt_ready) {
    CIM_CHECK_ARG(dy && x_raw && x && w && gamma && mean && var && workspace && B > 0);
    CIM_CHECK_ARG(side_stream == nullptr || side_stream == stream || (fork_event != nullptr && (!join || join_event != nullptr)));
    CONV3_ARGS_OK(4);
    CIM_CHECK_ARG((y != nullptr || !relu) && ((dgamma == nullptr) == (dbeta == nullptr)) && cout % 4 == 0);
    CIM_CHECK_ARG(!dy_is_dconv || (dres == nullptr && dgamma == nullptr));
    CIM_CHECK_ARG((in_gamma == nullptr) == (in_var == nullptr));
    CIM_CHECK_ARG(in_part == nullptr || (in_gamma && in_xr && in_mean));
    const ConvGeom cx = conv_geom(cin, cout, H, W, stride, CONV_DX, dilation), cw = conv_geom(cin, cout, H, W, stride, CONV_DW, dilation);
    const int hwo = cx.Ho * cx.Wo, hw = H * W;
    hipStream_t st = cim::as_stream(stream);
    float* dconv = dy_is_dconv ? const_cast<float*>(dy) : workspace;      // [B][cout][Ho Wo]: the gradient of the convolution output
    float* wt_own = workspace + (size_t)B * cout * hwo;        // [cout][9][cin]
    const float* wt = wt_ready ? wt_ready : wt_own;            // (made ahead of the pass by cim_conv3x3_wt_multi, or here)
    float* ws_dx = wt_own + (size_t)cout * cin * 9;
    float* ws_dw = ws_dx + (size_t)cim_gemm_small_splits(cin, hw, 9 * cout) * cin * hw;
    CIM_CHECK_ARG((size_t)cin * 9 * sizeof(float) <= 64 * 1024);
    int rc = dy_is_dconv ? 0 : cim_bn_act_bwd(dy, y, x_raw, gamma, mean, var, eps, dconv, dres, dgamma, dbeta, B, cout, hwo, relu, stream);
    if (rc) return rc;
    ForkJoin fj(st, cim::as_stream(side_stream), fork_event, join_event, dx && dw);
    hipStream_t st_dw = fj.fork();                             // the weight gradient next to the data gradient
    for (int b = 0; b < B && dw; ++b) {                        // dW[cout][cin 9] = dconv . im2col(X)^T  (K = Ho Wo)
        const int sp = cim_gemm_small_splits(cout, 9 * cin, hwo);
        float* out = b == 0 ? dw : ws_dw + (size_t)sp * cout * cin * 9;
        conv3x3_launch(CONV_DW, dconv + (size_t)b * cout * hwo, x + (size_t)b * cin * hw, out, cout, 9 * cin, hwo, hwo, 9 * cin, cw,
                       nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, nullptr, 0, sp, ws_dw, st_dw);
        if (b) {
            const size_t n = (size_t)cout * cin * 9;
            hipLaunchKernelGGL(small_axpy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st_dw, dw, out, n);
        }
    }
    if (dx && !wt_ready) hipLaunchKernelGGL(conv3x3_wt_kernel, dim3(cout), dim3(256), sizeof(float) * cin * 9, st, w, wt_own, cin);
    const int parts = cim_conv3x3_dx_parts(H, W, dilation == 1 ? stride : 1);
    for (int b = 0; b < B && dx; ++b) {                        // dX[cin][H W] = sum over (co, tap) W[co][ci][tap] dconv[co][shifted]
        const InputBn ib{x + (size_t)b * cin * hw, in_gamma, in_var, in_eps, in_xr ? in_xr + (size_t)b * cin * hw : nullptr, in_mean,
                         in_part ? in_part + (size_t)b * 2 * parts * cin : nullptr};
        if (stride == 2 && dilation == 1) {                    // by parity classes of the input pixels: a quarter of the MFMAs, no split-K
            SmallArgs g{};
            g.A = wt; g.B = dconv + (size_t)b * cout * hwo; g.C = dx + (size_t)b * cin * hw;
            g.M = cin; g.N = hw; g.K = 9 * cout; g.lda = cin; g.ldb = 0; g.ldc = hw;
            g.splits = 1;
            set_input_bn(g, in_gamma ? &ib : nullptr);
            g.mparts = parts;
            const Dx2Geom c2 = dx2_geom(cin, cout, H, W);
            const size_t lds = sizeof(float) * 4 * CBK * SLD;
            if (lds > 48 * 1024) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_dx2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) return (int)e;
            }
            hipLaunchKernelGGL(conv3x3_dx2_kernel, dim3((unsigned)c2.first[4]), dim3(256), lds, st, g, c2);
            continue;
        }
        conv3x3_launch(CONV_DX, wt, dconv + (size_t)b * cout * hwo, dx + (size_t)b * cin * hw, cin, hw, 9 * cout, cin, hw, cx, nullptr,
                       nullptr, nullptr, nullptr, nullptr, 0.f, nullptr, 0, cim_gemm_small_splits(cin, hw, 9 * cout), ws_dx, st, 3,
                       in_gamma ? &ib : nullptr);
    }
    if (join) fj.join();                                       // else the caller joins the side stream before the weight gradient is used
    CIM_CHECK_LAUNCH();
    return 0;
}
