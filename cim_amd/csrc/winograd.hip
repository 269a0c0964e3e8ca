// Winograd-domain stages of the MaskFuse 3 x 3 convolution on R x 7 x 7 ROI maps in the MIXED 4 + 3 tiling, and the (c, h, w)
// flatten in front of seg_fc.0 - the producers and consumers of the pair engine's operands (gemm_pair.hip).
//
// Replaces (together with cim_gemm_pair_batched) the evaluation of nn.Conv2d(2C, C, 3, padding=1) of
// /root/reference/lib/modeling/resnet50.py:104 and its two gradients:
//   forward        :  y  = A^T [ (G g G^T) . (B^T d B) ] A          per tile (Lavin & Gray), 121 multiplies per 49 outputs
//   data gradient  :  dx = overlap-add of B [ (A dy A^T) . U^T ] B^T   (the ADJOINT of the forward: reuses the forward's U)
//   weight gradient:  dW = AW [ (B^T d B)^T . (GD dy GD^T) ] AW^T
// Layouts: activations channels-last [R,7,7,C]; transformed operands position-major, [121][rows][C], as pair images (written by
// the transforms here, or by csrc/roi_align.hip's fused forward); GEMM results fp32.  HBM-bound kernels: lanes along C, 16 bytes
// per lane.  The superseded F(2x2,3x3) / F(4x4,3x3) algorithms and the fp32 / bf16x3 / f16x2 engines they fed are test
// infrastructure (experiments/csrc/winograd_all.hip, round 5).
#include "common.h"
#include "../../include/cim_hip.h"
#include "wino43_mats.h"

namespace {

__device__ __forceinline__ float4 f4(float v) { return make_float4(v, v, v, v); }
__device__ __forceinline__ float2 f2(float v) { return make_float2(v, v); }
__device__ __forceinline__ void fma2(float2& a, float s, float2 v) { a.x = fmaf(s, v.x, a.x); a.y = fmaf(s, v.y, a.y); }
__device__ __forceinline__ void fma4(float4& a, float s, float4 v) {
    a.x = fmaf(s, v.x, a.x); a.y = fmaf(s, v.y, a.y); a.z = fmaf(s, v.z, a.z); a.w = fmaf(s, v.w, a.w);
}

// =============================================================================================
// Mixed tiling of the 7 x 7 ROI map: one 4-wide and one 3-wide tile per axis (4 + 3 = 7) instead of two 4-wide tiles
// (8 > 7: 49/64 of the F(4x4,3x3) tile grid is used).  Tile type (ka, kb), k = 0: the 4-wide tile, F(4,3) on
// {0,1,-1,2,-1/2,inf}, 6 positions; k = 1: the 3-wide tile, F(3,3) on {0,1,-1,2,inf}, 5 positions.  36 + 30 + 30 + 25 =
// 121 positions, ONE tile of each type per ROI: 121 multiplies per 49 outputs instead of 144 (-16 % flops in all three
// contractions, and V / D / M shrink by the same factor).  `tile == 7` in the C entry points; P must be 7.
// Matrices (wino43_mats.h, W7_*: [axis kind][..][..], zero padded): B^T, G, A^T; weight gradient: GD (dy transform), AW.
// Layouts: V, D, M: [121][R][C];  U, dU: [121][K][N].
#ifndef CIM_W7_NT
#define CIM_W7_NT 1             // 1 = nontemporal stores of the transform outputs (streamed once, consumed by the next kernel)
#endif
typedef float w7_v2 __attribute__((ext_vector_type(2)));
typedef float w7_v4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void w7_store(float* p, float v) {
#if CIM_W7_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
__device__ __forceinline__ void w7_store(float* p, float2 v) {
#if CIM_W7_NT
    __builtin_nontemporal_store(w7_v2{v.x, v.y}, reinterpret_cast<w7_v2*>(p));
#else
    *reinterpret_cast<float2*>(p) = v;
#endif
}
__device__ __forceinline__ void w7_store(float* p, float4 v) {
#if CIM_W7_NT
    __builtin_nontemporal_store(w7_v4{v.x, v.y, v.z, v.w}, reinterpret_cast<w7_v4*>(p));
#else
    *reinterpret_cast<float4*>(p) = v;
#endif
}

// ---- pair-image output (f16x2p GEMM engine, gemm_pair.hip) ------------------------------------------------------------
// A lane holds 4 consecutive channels c .. c+3 (c = 4 x its index along C), its neighbour (lane ^ 1) the other half of the
// 8-channel chunk [h: 8 x f16 | l: 8 x f16].  The even lane hands its two l words to the odd lane and receives the odd lane's
// two h words (one quad_perm DPP move each way), so both store 16 contiguous bytes - the even lane the h half, the odd lane
// the l half - at the byte offset the fp32 float4 would have gone to: pair images keep the fp32 tensor's addressing.
typedef unsigned w7_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned w7_swap1(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);      // quad_perm [1,0,3,2]
}
__device__ __forceinline__ void w7_store_pair(float* p, const float4& v, float s) {
    unsigned h0, l0, h1, l1;
    cim::pair_split2(v.x * s, v.y * s, h0, l0);
    cim::pair_split2(v.z * s, v.w * s, h1, l1);
    const bool odd = (threadIdx.x & 1) != 0;
    const unsigned r0 = w7_swap1(odd ? h0 : l0), r1 = w7_swap1(odd ? h1 : l1);
    const w7_u4 o = odd ? w7_u4{r0, r1, l0, l1} : w7_u4{h0, h1, r0, r1};
    __builtin_nontemporal_store(o, reinterpret_cast<w7_u4*>(p));
}
__device__ __forceinline__ void w7_store_pair(float*, const float2&, float) {}      // (only 4-channel lanes write pair images)
__device__ __forceinline__ void w7_store_pair(float*, float, float) {}

// lane vector width of the transform kernels (channels per lane): 16-byte accesses stream at 6.6-6.8 TB/s where the
// 4- / 8-byte ones reach 4.3-5.0 (tools/bench_wino.py), as long as the tile still fits the register file
#ifndef CIM_W7_VIN
#define CIM_W7_VIN 4            // input transform  x -> V      (2 | 4)
#endif
#ifndef CIM_W7_VOUT
#define CIM_W7_VOUT 4           // output transform M -> y      (2 | 4)
#endif
#ifndef CIM_W7_VMF
#define CIM_W7_VMF 1            // mask-folding adjoint output   Md -> dbox    (1 | 2): one channel per lane keeps both halves'
                                // transforms at ~120 VGPRs (two per lane: 314, one wave per SIMD - 0.266 vs 0.235 ms at cfg2)
#endif
#ifndef CIM_W7_MF_WAVES
#define CIM_W7_MF_WAVES 1       // register cap of that kernel: waves per SIMD the compiler must allow
#endif
#ifndef CIM_W7_VDX
#define CIM_W7_VDX 2            // adjoint output   Md -> dx    (1 | 2 | 4)
#endif
#ifndef CIM_W7_VWG
#define CIM_W7_VWG 1            // weight-gradient output dU -> dW (1 | 4; measured 0.254 | 0.297 ms)
#endif
template <int N> struct w7_vec;
template <> struct w7_vec<1> { typedef float T; };
template <> struct w7_vec<2> { typedef float2 T; };
template <> struct w7_vec<4> { typedef float4 T; };
__device__ __forceinline__ void vzero(float& a) { a = 0.0f; }
__device__ __forceinline__ void vzero(float2& a) { a = make_float2(0.f, 0.f); }
__device__ __forceinline__ void vzero(float4& a) { a = make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void vfma(float& a, float s, float v) { a = fmaf(s, v, a); }
__device__ __forceinline__ void vfma(float2& a, float s, float2 v) { fma2(a, s, v); }
__device__ __forceinline__ void vfma(float4& a, float s, float4 v) { fma4(a, s, v); }
__device__ __forceinline__ float vamax(float v) { return fabsf(v); }
__device__ __forceinline__ float vamax(float2 v) { return fmaxf(fabsf(v.x), fabsf(v.y)); }
__device__ __forceinline__ float vamax(float4 v) { return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))); }
__device__ __forceinline__ void vrelu(float& v) { v = fmaxf(v, 0.f); }
__device__ __forceinline__ void vrelu(float2& v) { v = make_float2(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f)); }
__device__ __forceinline__ void vrelu(float4& v) { v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)); }
__device__ __forceinline__ float vget(float v, int) { return v; }
__device__ __forceinline__ float vget(const float4& v, int l) { return l == 0 ? v.x : l == 1 ? v.y : l == 2 ? v.z : v.w; }
template <typename T> __device__ __forceinline__ T vload(const float* p) { return *reinterpret_cast<const T*>(p); }
#ifndef CIM_W7_NTL
#define CIM_W7_NTL 1            // 1 = nontemporal loads of the transform inputs that are read exactly once (M, Md, dU)
#endif
template <typename T> __device__ __forceinline__ T vload_once(const float* p) {
#if CIM_W7_NTL
    if constexpr (sizeof(T) == 4) return __builtin_nontemporal_load(p);
    else if constexpr (sizeof(T) == 8) { const w7_v2 v = __builtin_nontemporal_load(reinterpret_cast<const w7_v2*>(p)); return make_float2(v.x, v.y); }
    else { const w7_v4 v = __builtin_nontemporal_load(reinterpret_cast<const w7_v4*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
#else
    return vload<T>(p);
#endif
}

struct W7 {
    static constexpr int NP[2] = {6, 5};      // positions per axis
    static constexpr int OUT[2] = {4, 3};     // outputs per axis
    static constexpr int IN0[2] = {-1, 3};    // first input row / column of the tile's patch
    static constexpr int OUT0[2] = {0, 4};    // first output row / column
    static constexpr int QOFF[4] = {0, 36, 66, 96};   // first position of tile type ka * 2 + kb (121 in total)
};

__host__ __device__ constexpr float w7_abs_row_sum(const float (&M)[6], int n) {
    float s = 0.0f;
    for (int k = 0; k < n; ++k) s += M[k] < 0 ? -M[k] : M[k];
    return s;
}

template <int KA, int KB, bool AMAX, int VW, bool PAIR = false>
__device__ __forceinline__ void w7_input_tile(const float* __restrict__ x, float* __restrict__ V, int r, int R, int C,
                                              unsigned* __restrict__ row_amax, const float* __restrict__ scale = nullptr) {
    // PAIR: V is a pair image [121][R (position stride, padded)][C], scale [121]; x rows are indexed by r as before
    constexpr int NA = W7::NP[KA], NB = W7::NP[KB], P = 7, Q0 = W7::QOFF[KA * 2 + KB];
    typedef typename w7_vec<VW>::T VT;
    const size_t MC = (size_t)R * C;
    float dmax = 0.0f;
    for (int c = threadIdx.x * VW; c < C; c += 256 * VW) {
        VT d[NA][NB];
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int iy = W7::IN0[KA] + i;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int ix = W7::IN0[KB] + j;
                if ((unsigned)iy < (unsigned)P && (unsigned)ix < (unsigned)P) d[i][j] = vload<VT>(x + (((size_t)r * P + iy) * P + ix) * C + c);
                else vzero(d[i][j]);
                if constexpr (AMAX) dmax = fmaxf(dmax, vamax(d[i][j]));
            }
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            VT trow[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                vzero(trow[j]);
#pragma unroll
                for (int k = 0; k < NA; ++k)
                    if (W7_BT[KA][i][k] != 0.0f) vfma(trow[j], W7_BT[KA][i][k], d[k][j]);
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                VT v;
                vzero(v);
#pragma unroll
                for (int k = 0; k < NB; ++k)
                    if (W7_BT[KB][j][k] != 0.0f) vfma(v, W7_BT[KB][j][k], trow[k]);
                if constexpr (PAIR) w7_store_pair(V + (size_t)(Q0 + i * NB + j) * MC + (size_t)r * C + c, v, scale[Q0 + i * NB + j]);
                else w7_store(V + (size_t)(Q0 + i * NB + j) * MC + (size_t)r * C + c, v);
            }
        }
    }
    if constexpr (AMAX) {       // row-scale bounds of this tile's positions (see wino43_input_kernel)
        __shared__ float red[4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, o));
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dmax;
        __syncthreads();
        if (threadIdx.x < NA * NB) {
            const float tile_max = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
            const int i = threadIdx.x / NB, j = threadIdx.x % NB;
            float fi = 0.0f, fj = 0.0f;
#pragma unroll
            for (int p = 0; p < 6; ++p) {
                if (p == i) fi = w7_abs_row_sum(W7_BT[KA][p], NA);
                if (p == j) fj = w7_abs_row_sum(W7_BT[KB][p], NB);
            }
            row_amax[(size_t)(Q0 + threadIdx.x) * R + r] = __float_as_uint(fi * fj * tile_max * 1.0001f);
        }
    }
}

// zero rows of a pair image: positions [q0, q0 + nq) of row r (the rows that pad R up to a multiple of 32)
__device__ __forceinline__ void w7_zero_rows(float* __restrict__ V, int q0, int nq, int r, int Rs, int C) {
    for (int q = q0; q < q0 + nq; ++q)
        for (int c = threadIdx.x * 4; c < C; c += 1024)
            *reinterpret_cast<float4*>(V + ((size_t)q * Rs + r) * C + c) = make_float4(0.f, 0.f, 0.f, 0.f);
}

// x [R][7][7][C] fp32 -> pair image V [121][Rs][C] (Rs = R padded to 32 rows, pad rows zeroed); grid = (Rs, 4)
__global__ __launch_bounds__(256) void wino7_input_pair_kernel(const float* __restrict__ x, float* __restrict__ V, int R, int Rs,
                                                               int C, const float* __restrict__ scale) {
    const int r = blockIdx.x;
    if (r >= R) {
        w7_zero_rows(V, W7::QOFF[blockIdx.y], W7::NP[blockIdx.y >> 1] * W7::NP[blockIdx.y & 1], r, Rs, C);
        return;
    }
    switch (blockIdx.y) {
        case 0: w7_input_tile<0, 0, false, 4, true>(x, V, r, Rs, C, nullptr, scale); break;
        case 1: w7_input_tile<0, 1, false, 4, true>(x, V, r, Rs, C, nullptr, scale); break;
        case 2: w7_input_tile<1, 0, false, 4, true>(x, V, r, Rs, C, nullptr, scale); break;
        default: w7_input_tile<1, 1, false, 4, true>(x, V, r, Rs, C, nullptr, scale); break;
    }
}

// Filter transform into a pair image U' [121][Cout][Cin] (ci contiguous: the B operand of the forward product read
// K-contiguously, of the data gradient read N-contiguously).  One lane = one (co, ci), lanes along ci (a wave reads 64 x 36
// contiguous bytes of W); the 8 lanes of a chunk exchange their packed (h, l) halves through two ds_bpermute so that lane j
// stores word j of the chunk [h0h1 h2h3 h4h5 h6h7 | l0l1 l2l3 l4l5 l6l7] - 4 bytes per lane at the fp32 tensor's byte offset,
// 256 contiguous bytes per wave.  (The first version - one thread per chunk, 72 floats in, two 16-byte stores out per
// position - ran 0.60 ms: its loads touched 64 cache lines per instruction.)
template <int KA, int KB>
__device__ __forceinline__ void w7_filter_pair_tile(const float (&w)[3][3], float* __restrict__ U, size_t KN, size_t idx,
                                                    const float* __restrict__ scale, int src_a, int src_b, bool lo_half) {
    constexpr int NA = W7::NP[KA], NB = W7::NP[KB], Q0 = W7::QOFF[KA * 2 + KB];
    float t[NA][3];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int b = 0; b < 3; ++b) t[i][b] = W7_G[KA][i][0] * w[0][b] + W7_G[KA][i][1] * w[1][b] + W7_G[KA][i][2] * w[2][b];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const float v = (t[i][0] * W7_G[KB][j][0] + t[i][1] * W7_G[KB][j][1] + t[i][2] * W7_G[KB][j][2]) * scale[Q0 + i * NB + j];
            unsigned h, l;
            cim::pair_split2(v, 0.0f, h, l);
            const unsigned mine = (h & 0xffffu) | (l << 16);                                  // (h, l) of this lane's element
            const unsigned pa = (unsigned)__builtin_amdgcn_ds_bpermute(src_a, (int)mine);   // elements 2 (j & 3), 2 (j & 3) + 1
            const unsigned pb = (unsigned)__builtin_amdgcn_ds_bpermute(src_b, (int)mine);
            const unsigned word = lo_half ? ((pa & 0xffffu) | (pb << 16)) : ((pa >> 16) | (pb & 0xffff0000u));
            __builtin_nontemporal_store(word, reinterpret_cast<unsigned*>(U + (size_t)(Q0 + i * NB + j) * KN + idx));
        }
}

__global__ __launch_bounds__(256) void wino7_filter_pair_kernel(const float* __restrict__ W, float* __restrict__ U, int Cout,
                                                                int Cin, const float* __restrict__ scale) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;          // (co, ci), ci fastest; Cin % 8 == 0: chunks never straddle rows
    const size_t KN = (size_t)Cout * Cin;
    const bool live = idx < KN;
    const float* g = W + (live ? idx : 0) * 9;
    float w[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) w[a][b] = live ? g[a * 3 + b] : 0.0f;
    const int lane = threadIdx.x & 63, j = lane & 7;
    const int src_a = ((lane & ~7) + 2 * (j & 3)) * 4, src_b = src_a + 4;
    if (!live) return;               // (KN is a multiple of 8 and idx is chunk-aligned per 8 lanes: whole chunks leave together)
    w7_filter_pair_tile<0, 0>(w, U, KN, idx, scale, src_a, src_b, j < 4);
    w7_filter_pair_tile<0, 1>(w, U, KN, idx, scale, src_a, src_b, j < 4);
    w7_filter_pair_tile<1, 0>(w, U, KN, idx, scale, src_a, src_b, j < 4);
    w7_filter_pair_tile<1, 1>(w, U, KN, idx, scale, src_a, src_b, j < 4);
}

template <int KA, int KB, int VW>
__device__ __forceinline__ void w7_output_tile(const float* __restrict__ M, const float* __restrict__ bias,
                                               float* __restrict__ y, int r, int R, int C, int relu, float& ymax) {
    constexpr int NA = W7::NP[KA], NB = W7::NP[KB], OA = W7::OUT[KA], OB = W7::OUT[KB], P = 7, Q0 = W7::QOFF[KA * 2 + KB];
    typedef typename w7_vec<VW>::T VT;
    const size_t MC = (size_t)R * C;
    for (int c = threadIdx.x * VW; c < C; c += 256 * VW) {
        VT s[OA][NB];
#pragma unroll
        for (int a = 0; a < OA; ++a)
#pragma unroll
            for (int j = 0; j < NB; ++j) vzero(s[a][j]);
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const VT q = vload_once<VT>(M + (size_t)(Q0 + i * NB + j) * MC + (size_t)r * C + c);
#pragma unroll
                for (int a = 0; a < OA; ++a)
                    if (W7_AT[KA][a][i] != 0.0f) vfma(s[a][j], W7_AT[KA][a][i], q);
            }
        VT bv;
        if (bias) bv = vload<VT>(bias + c);
        else vzero(bv);
#pragma unroll
        for (int a = 0; a < OA; ++a)
#pragma unroll
            for (int b = 0; b < OB; ++b) {
                VT v = bv;
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    if (W7_AT[KB][b][j] != 0.0f) vfma(v, W7_AT[KB][b][j], s[a][j]);
                if (relu) vrelu(v);
                ymax = fmaxf(ymax, vamax(v));
                w7_store(y + (((size_t)r * P + W7::OUT0[KA] + a) * P + W7::OUT0[KB] + b) * C + c, v);
            }
    }
}

// y_amax (optional): max |y| as a bit pattern, atomicMax into a caller-zeroed word - the scale source of the pair image
// the flatten kernel writes next
__global__ __launch_bounds__(256) void wino7_output_kernel(const float* __restrict__ M, const float* __restrict__ bias,
                                                           float* __restrict__ y, int R, int C, int relu,
                                                           unsigned* __restrict__ y_amax) {
    const int r = blockIdx.x;
    float ymax = 0.0f;
    switch (blockIdx.y) {
        case 0: w7_output_tile<0, 0, CIM_W7_VOUT>(M, bias, y, r, R, C, relu, ymax); break;
        case 1: w7_output_tile<0, 1, CIM_W7_VOUT>(M, bias, y, r, R, C, relu, ymax); break;
        case 2: w7_output_tile<1, 0, CIM_W7_VOUT>(M, bias, y, r, R, C, relu, ymax); break;
        default: w7_output_tile<1, 1, CIM_W7_VOUT>(M, bias, y, r, R, C, relu, ymax); break;
    }
    if (y_amax != nullptr) {             // (one atomicMax per workgroup: see pair_masked_stats_kernel)
        __shared__ float s_ymax[4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ymax = fmaxf(ymax, __shfl_xor(ymax, o));
        if ((threadIdx.x & 63) == 0) s_ymax[threadIdx.x >> 6] = ymax;
        __syncthreads();
        if (threadIdx.x == 0)
            cim::amax_publish(y_amax, __float_as_uint(fmaxf(fmaxf(s_ymax[0], s_ymax[1]), fmaxf(s_ymax[2], s_ymax[3]))));
    }
}

// Output-gradient tile transform.  ADJ = false: D = GD dy GD^T (weight gradient, F(3,4) / F(3,3)).
// ADJ = true: E = A dy A^T with A = (A^T)^T - the first stage of the data gradient written as the ADJOINT of the forward,
//   y = A^T [U . (B^T d B)] A   =>   dx (+)= B [U^T . (A dy A^T)] B^T      (overlap-add over the tiles' patches),
// which reuses the forward's U (contracted over the other channel index) instead of transforming a rotated filter;
// with AMAX it also stores the row-scale bounds of E (column abs sums of A^T times the tile's max |dy|).
template <int KA, int KB, bool ADJ, bool AMAX, bool PAIR = false>
__device__ __forceinline__ void w7_dy_tile(const float* __restrict__ dy, float* __restrict__ D, int r, int R, int C,
                                           unsigned* __restrict__ row_amax, const float* __restrict__ scale = nullptr) {
    constexpr int NA = W7::NP[KA], NB = W7::NP[KB], OA = W7::OUT[KA], OB = W7::OUT[KB], P = 7, Q0 = W7::QOFF[KA * 2 + KB];
    const size_t MC = (size_t)R * C;
    float dmax = 0.0f;
    for (int c = threadIdx.x * 4; c < C; c += 256 * 4) {      // 16 B per lane: the 4 x 4 tile fits the register budget
        float4 d[OA][OB];
#pragma unroll
        for (int a = 0; a < OA; ++a)
#pragma unroll
            for (int b = 0; b < OB; ++b) {
                d[a][b] = *reinterpret_cast<const float4*>(dy + (((size_t)r * P + W7::OUT0[KA] + a) * P + W7::OUT0[KB] + b) * C + c);
                if constexpr (AMAX)
                    dmax = fmaxf(dmax, fmaxf(fmaxf(fabsf(d[a][b].x), fabsf(d[a][b].y)), fmaxf(fabsf(d[a][b].z), fabsf(d[a][b].w))));
            }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            float4 trow[OB];
#pragma unroll
            for (int b = 0; b < OB; ++b) {
                trow[b] = f4(0.f);
#pragma unroll
                for (int a = 0; a < OA; ++a) {
                    const float m = ADJ ? W7_AT[KA][a][i] : W7_GD[KA][i][a];
                    if (m != 0.0f) fma4(trow[b], m, d[a][b]);
                }
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                float4 v = f4(0.f);
#pragma unroll
                for (int b = 0; b < OB; ++b) {
                    const float m = ADJ ? W7_AT[KB][b][j] : W7_GD[KB][j][b];
                    if (m != 0.0f) fma4(v, m, trow[b]);
                }
                if constexpr (PAIR) w7_store_pair(D + (size_t)(Q0 + i * NB + j) * MC + (size_t)r * C + c, v, scale[Q0 + i * NB + j]);
                else w7_store(D + (size_t)(Q0 + i * NB + j) * MC + (size_t)r * C + c, v);
            }
        }
    }
    if constexpr (AMAX) {
        __shared__ float red[4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, o));
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dmax;
        __syncthreads();
        if (threadIdx.x < NA * NB) {
            const float tile_max = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
            const int i = threadIdx.x / NB, j = threadIdx.x % NB;
            float fi = 0.0f, fj = 0.0f;
#pragma unroll
            for (int p = 0; p < 6; ++p) {
                float sa = 0.0f, sb = 0.0f;
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    sa += fabsf(ADJ ? W7_AT[KA][a][p] : W7_GD[KA][p][a]);
                    sb += fabsf(ADJ ? W7_AT[KB][a][p] : W7_GD[KB][p][a]);
                }
                if (p == i) fi = sa;
                if (p == j) fj = sb;
            }
            row_amax[(size_t)(Q0 + threadIdx.x) * R + r] = __float_as_uint(fi * fj * tile_max * 1.0001f);
        }
    }
}

// dy [R][7][7][C] fp32 -> pair image D / E [121][Rs][C]; grid = (Rs, 4)
template <bool ADJ>
__global__ __launch_bounds__(256) void wino7_dy_pair_kernel(const float* __restrict__ dy, float* __restrict__ D, int R, int Rs,
                                                            int C, const float* __restrict__ scale) {
    const int r = blockIdx.x;
    if (r >= R) {
        w7_zero_rows(D, W7::QOFF[blockIdx.y], W7::NP[blockIdx.y >> 1] * W7::NP[blockIdx.y & 1], r, Rs, C);
        return;
    }
    switch (blockIdx.y) {
        case 0: w7_dy_tile<0, 0, ADJ, false, true>(dy, D, r, Rs, C, nullptr, scale); break;
        case 1: w7_dy_tile<0, 1, ADJ, false, true>(dy, D, r, Rs, C, nullptr, scale); break;
        case 2: w7_dy_tile<1, 0, ADJ, false, true>(dy, D, r, Rs, C, nullptr, scale); break;
        default: w7_dy_tile<1, 1, ADJ, false, true>(dy, D, r, Rs, C, nullptr, scale); break;
    }
}

// ---- flatten backward + ReLU mask + BOTH output-gradient transforms of the convolution in one launch -------------------------------
// Replaces flatten_chw_kernel<false> + wino7_dy_pair_kernel<true> + wino7_dy_pair_kernel<false> (the backward of `.view(N, -1)`
// and mask_branch's ReLU, /root/reference/lib/modeling/resnet50.py:135,104-105, in front of the convolution's two gradient products):
// dX [R][C * 49] ((c, h, w) order, seg_fc.0's data gradient) is read ONCE, masked with the saved conv output, and leaves as the
// pair images E = A dy A^T (adjoint data gradient) and D = GD dy GD^T (weight gradient) - the masked gradient dy [R,7,7,C] is
// never stored (was: 200 MB written, read twice).  Workgroup = (ROI, 256-channel slice): the slice's 49 x 256 gradients are 12544
// CONSECUTIVE floats of dX; they are transposed through LDS ([49][260] floats), masked in place (a lane owns 4 channels, a wave
// every fourth pixel: the bias partial sums keep flatten_chw_kernel's order), then wave t transforms tile type t from LDS exactly as
// w7_dy_tile does from memory - bit-identical images.  grid = (Rs, C / 256); rows R .. Rs-1 are zeroed.
constexpr int W7_FB_LDW = 260;

template <int KA, int KB, bool ADJ>
__device__ __forceinline__ void w7_dy_regs_pair(const float4 (&d)[W7::OUT[KA]][W7::OUT[KB]], float* __restrict__ D, size_t MC, size_t rc,
                                                const float* __restrict__ scale) {
    constexpr int NA = W7::NP[KA], NB = W7::NP[KB], OA = W7::OUT[KA], OB = W7::OUT[KB], Q0 = W7::QOFF[KA * 2 + KB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        float4 trow[OB];
#pragma unroll
        for (int b = 0; b < OB; ++b) {
            trow[b] = f4(0.f);
#pragma unroll
            for (int a = 0; a < OA; ++a) {
                const float m = ADJ ? W7_AT[KA][a][i] : W7_GD[KA][i][a];
                if (m != 0.0f) fma4(trow[b], m, d[a][b]);
            }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            float4 v = f4(0.f);
#pragma unroll
            for (int b = 0; b < OB; ++b) {
                const float m = ADJ ? W7_AT[KB][b][j] : W7_GD[KB][j][b];
                if (m != 0.0f) fma4(v, m, trow[b]);
            }
            w7_store_pair(D + (size_t)(Q0 + i * NB + j) * MC + rc, v, scale[Q0 + i * NB + j]);
        }
    }
}

template <int KA, int KB>
__device__ __forceinline__ void w7_fb_tile(const float* __restrict__ s, int lane, float* __restrict__ E, float* __restrict__ D, size_t MC,
                                           size_t rc, const float* __restrict__ sE, const float* __restrict__ sD) {
    constexpr int OA = W7::OUT[KA], OB = W7::OUT[KB];
    float4 d[OA][OB];
#pragma unroll
    for (int a = 0; a < OA; ++a)
#pragma unroll
        for (int b = 0; b < OB; ++b)
            d[a][b] = *reinterpret_cast<const float4*>(s + ((W7::OUT0[KA] + a) * 7 + W7::OUT0[KB] + b) * W7_FB_LDW + 4 * lane);
    if (E != nullptr) w7_dy_regs_pair<KA, KB, true>(d, E, MC, rc, sE);
    if (D != nullptr) w7_dy_regs_pair<KA, KB, false>(d, D, MC, rc, sD);
}

#ifndef CIM_W7_FB_WAVES
#define CIM_W7_FB_WAVES 3
#endif
__global__ __launch_bounds__(256, CIM_W7_FB_WAVES) void wino7_flatten_bwd_dy_pair_kernel(const float* __restrict__ dX, const float* __restrict__ y,
                                                                        float* __restrict__ E, float* __restrict__ D,
                                                                        float* __restrict__ bsum, int R, int Rs, int C,
                                                                        const float* __restrict__ sE, const float* __restrict__ sD) {
    __shared__ __attribute__((aligned(16))) float s[49 * W7_FB_LDW];
    __shared__ float4 red[3][64];          // (waves 1 .. 3; with the tile 54032 B: three workgroups per CU)
    const int r = blockIdx.x, c0 = blockIdx.y * 256, tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t MC = (size_t)Rs * C, rc = (size_t)r * C + c0 + 4 * lane;
    if (r >= R) {                          // pad rows of both images
        const w7_u4 z = {0u, 0u, 0u, 0u};
        for (int q = w; q < 121; q += 4) {
            if (E != nullptr) __builtin_nontemporal_store(z, reinterpret_cast<w7_u4*>(E + (size_t)q * MC + rc));
            if (D != nullptr) __builtin_nontemporal_store(z, reinterpret_cast<w7_u4*>(D + (size_t)q * MC + rc));
        }
        return;
    }
    // the slice's (c, p) run of dX -> s[p][c]
    const float4* __restrict__ src4 = reinterpret_cast<const float4*>(dX + ((size_t)r * C + c0) * 49);
    for (int i = tid; i < 49 * 64; i += 256) {
        const float4 v = vload_once<float4>(reinterpret_cast<const float*>(src4 + i));
        int c = (4 * i) / 49, p = 4 * i - 49 * c;
        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s[p * W7_FB_LDW + c] = e[k];
            if (++p == 49) { p = 0; ++c; }
        }
    }
    __syncthreads();
    // ReLU mask of the saved conv output, in place; per-channel sums over the ROI's pixels (the conv's bias gradient, partial)
    float4 part = f4(0.f);
    for (int p = w; p < 49; p += 4) {
        float4* sp = reinterpret_cast<float4*>(s + p * W7_FB_LDW + 4 * lane);
        float4 d = *sp;
        if (y != nullptr) {
            const float4 yv = *reinterpret_cast<const float4*>(y + ((size_t)r * 49 + p) * C + c0 + 4 * lane);
            d.x = yv.x > 0.0f ? d.x : 0.0f;
            d.y = yv.y > 0.0f ? d.y : 0.0f;
            d.z = yv.z > 0.0f ? d.z : 0.0f;
            d.w = yv.w > 0.0f ? d.w : 0.0f;
            *sp = d;
        }
        part.x += d.x; part.y += d.y; part.z += d.z; part.w += d.w;
    }
    if (bsum != nullptr && w > 0) red[w - 1][lane] = part;
    __syncthreads();
    if (bsum != nullptr && w == 0) {
        const float4 a = part, b = red[0][lane], c = red[1][lane], e = red[2][lane];
        *reinterpret_cast<float4*>(bsum + rc) = make_float4((a.x + b.x) + (c.x + e.x), (a.y + b.y) + (c.y + e.y),
                                                            (a.z + b.z) + (c.z + e.z), (a.w + b.w) + (c.w + e.w));
    }
    // wave t: the adjoint image of tile type t and the weight-gradient image of tile type 3 - t (36 + 25, 30 + 30, 30 + 30, 25 + 36
    // positions: the four waves carry the same number of stores)
    switch (w) {
        case 0: w7_fb_tile<0, 0>(s, lane, E, nullptr, MC, rc, sE, sD); w7_fb_tile<1, 1>(s, lane, nullptr, D, MC, rc, sE, sD); break;
        case 1: w7_fb_tile<0, 1>(s, lane, E, nullptr, MC, rc, sE, sD); w7_fb_tile<1, 0>(s, lane, nullptr, D, MC, rc, sE, sD); break;
        case 2: w7_fb_tile<1, 0>(s, lane, E, nullptr, MC, rc, sE, sD); w7_fb_tile<0, 1>(s, lane, nullptr, D, MC, rc, sE, sD); break;
        default: w7_fb_tile<1, 1>(s, lane, E, nullptr, MC, rc, sE, sD); w7_fb_tile<0, 0>(s, lane, nullptr, D, MC, rc, sE, sD); break;
    }
}

// scale[q] = 2^(14 - exponent(bound_q)), bound_q = (abs row sum)_i (abs row sum)_j max|d| 1.0001 >= max |transformed value| at
// position q.  kind 0: B^T (input), 1: G (filter), 2: GD (dy, weight gradient), 3: A (dy, adjoint data gradient).
__global__ void wino7_pair_scales_kernel(const unsigned* __restrict__ amax, int n_amax, const unsigned* __restrict__ amax_mul, int kind,
                                         float* __restrict__ scale) {
    const int q = threadIdx.x;
    if (q >= 121) return;
    unsigned mbits = 0;                     // (non-negative floats order like their bit patterns)
    for (int t = 0; t < n_amax; ++t) mbits = max(mbits, amax[t] & 0x7fffffffu);
    float dmax = __uint_as_float(mbits);
    if (amax_mul != nullptr) dmax *= fmaxf(1.0f, __uint_as_float(amax_mul[0] & 0x7fffffffu));
    const int type = q < 36 ? 0 : q < 66 ? 1 : q < 96 ? 2 : 3;
    const int ka = type >> 1, kb = type & 1, nb = kb ? 5 : 6, pl = q - W7::QOFF[type];
    const int pi = pl / nb, pj = pl % nb;
    float fa = 0.0f, fb = 0.0f;
    if (kind == 0) {
        for (int t = 0; t < 6; ++t) { fa += fabsf(W7_BT[ka][pi][t]); fb += fabsf(W7_BT[kb][pj][t]); }
    } else if (kind == 1) {
        for (int t = 0; t < 3; ++t) { fa += fabsf(W7_G[ka][pi][t]); fb += fabsf(W7_G[kb][pj][t]); }
    } else if (kind == 2) {
        for (int t = 0; t < 4; ++t) { fa += fabsf(W7_GD[ka][pi][t]); fb += fabsf(W7_GD[kb][pj][t]); }
    } else {
        for (int t = 0; t < 4; ++t) { fa += fabsf(W7_AT[ka][t][pi]); fb += fabsf(W7_AT[kb][t][pj]); }
    }
    scale[q] = cim::pair_scale_of(__float_as_uint(fa * fb * dmax * 1.0001f));
}

// Last stage of the adjoint data gradient: dx[r, y, x, c] = sum over the four tile types of (B M B^T)[y - y0][x - x0]
// (overlap-add: the 6- and 5-row patches share rows / columns 3 and 4).  A workgroup owns one ROI, a lane one channel
// (all 49 output pixels accumulate in registers; 4 B per lane keeps that at ~110 VGPRs).
template <int KA, int KB, typename VT>
__device__ __forceinline__ void w7_dx_tile(const float* __restrict__ M, size_t MC, size_t rc, VT (&acc)[7][7]) {
    constexpr int NA = W7::NP[KA], NB = W7::NP[KB], P = 7, Q0 = W7::QOFF[KA * 2 + KB];
    VT q[NA][NB];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) q[i][j] = vload_once<VT>(M + (size_t)(Q0 + i * NB + j) * MC + rc);
#pragma unroll
    for (int k = 0; k < NA; ++k) {                 // patch row k = sum_i B^T[i][k] q[i][:]
        const int yy = W7::IN0[KA] + k;
        if (yy < 0 || yy >= P) continue;
        VT t[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            vzero(t[j]);
#pragma unroll
            for (int i = 0; i < NA; ++i)
                if (W7_BT[KA][i][k] != 0.0f) vfma(t[j], W7_BT[KA][i][k], q[i][j]);
        }
#pragma unroll
        for (int l = 0; l < NB; ++l) {
            const int xx = W7::IN0[KB] + l;
            if (xx < 0 || xx >= P) continue;
#pragma unroll
            for (int j = 0; j < NB; ++j)
                if (W7_BT[KB][j][l] != 0.0f) vfma(acc[yy][xx], W7_BT[KB][j][l], t[j]);
        }
    }
}

// grid = (R, channel chunks); block = 256 (CIM_W7_VDX channels per lane)
__global__ __launch_bounds__(256) void wino7_dx_kernel(const float* __restrict__ M, float* __restrict__ dx, int R, int C) {
    constexpr int VW = CIM_W7_VDX;
    typedef typename w7_vec<VW>::T VT;
    const int r = blockIdx.x;
    const size_t MC = (size_t)R * C;
    for (int c = (blockIdx.y * 256 + threadIdx.x) * VW; c < C; c += gridDim.y * 256 * VW) {
        VT acc[7][7];
#pragma unroll
        for (int y = 0; y < 7; ++y)
#pragma unroll
            for (int x = 0; x < 7; ++x) vzero(acc[y][x]);
        const size_t rc = (size_t)r * C + c;
        w7_dx_tile<0, 0, VT>(M, MC, rc, acc);
        w7_dx_tile<0, 1, VT>(M, MC, rc, acc);
        w7_dx_tile<1, 0, VT>(M, MC, rc, acc);
        w7_dx_tile<1, 1, VT>(M, MC, rc, acc);
#pragma unroll
        for (int y = 0; y < 7; ++y)
#pragma unroll
            for (int x = 0; x < 7; ++x) w7_store(dx + (((size_t)r * 7 + y) * 7 + x) * C + c, acc[y][x]);
    }
}

// The same stage with MaskFuse's prologue folded back in (round 5): Md holds the gradient of cat = [box, box * mask] (2 Cb channels);
// what the ROIAlign backward needs is  dbox = dcat[:, :Cb] + mask * dcat[:, Cb:]  (lib/modeling/resnet50.py:131-134 differentiated).
// A lane transforms channel c of the MASKED half first (all four tile types into the 49 accumulators), scales by the ROI's 7 x 7
// mask, then accumulates the plain half on top: dbox [R,7,7,Cb] is written instead of dcat [R,7,7,2Cb] - half the bytes out of
// this launch and half the bytes into the ROIAlign backward (which re-reads them 1.5x).  grid = (R, chunks); block = 256.
__global__ __launch_bounds__(256, CIM_W7_MF_WAVES) void wino7_dx_maskfold_kernel(const float* __restrict__ M, const float* __restrict__ masks,
                                                                float* __restrict__ dbox, int R, int Cb) {
    constexpr int VW = CIM_W7_VMF;
    typedef typename w7_vec<VW>::T VT;
    const int r = blockIdx.x;
    const size_t C2 = 2 * (size_t)Cb, MC = (size_t)R * C2;
    const float* __restrict__ mk = masks + (size_t)r * 49;
    for (int c = (blockIdx.y * 256 + threadIdx.x) * VW; c < Cb; c += gridDim.y * 256 * VW) {
        VT acc[7][7];
#pragma unroll
        for (int y = 0; y < 7; ++y)
#pragma unroll
            for (int x = 0; x < 7; ++x) vzero(acc[y][x]);
        const size_t rc = (size_t)r * C2 + c;
        w7_dx_tile<0, 0, VT>(M, MC, rc + Cb, acc);
        w7_dx_tile<0, 1, VT>(M, MC, rc + Cb, acc);
        w7_dx_tile<1, 0, VT>(M, MC, rc + Cb, acc);
        w7_dx_tile<1, 1, VT>(M, MC, rc + Cb, acc);
#pragma unroll
        for (int y = 0; y < 7; ++y)
#pragma unroll
            for (int x = 0; x < 7; ++x) {
                const float m = mk[y * 7 + x];
                VT z;
                vzero(z);
                vfma(z, m, acc[y][x]);
                acc[y][x] = z;
            }
        w7_dx_tile<0, 0, VT>(M, MC, rc, acc);
        w7_dx_tile<0, 1, VT>(M, MC, rc, acc);
        w7_dx_tile<1, 0, VT>(M, MC, rc, acc);
        w7_dx_tile<1, 1, VT>(M, MC, rc, acc);
#pragma unroll
        for (int y = 0; y < 7; ++y)
#pragma unroll
            for (int x = 0; x < 7; ++x) w7_store(dbox + (((size_t)r * 7 + y) * 7 + x) * Cb + c, acc[y][x]);
    }
}

template <int KA, int KB, typename VT>
__device__ __forceinline__ void w7_wgrad_tile(const float* __restrict__ dU, size_t KN, size_t idx, VT (&acc)[3][3]) {
    constexpr int NA = W7::NP[KA], NB = W7::NP[KB], Q0 = W7::QOFF[KA * 2 + KB];
    VT s[3][NB];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int j = 0; j < NB; ++j) vzero(s[a][j]);
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const VT q = vload_once<VT>(dU + (size_t)(Q0 + i * NB + j) * KN + idx);
#pragma unroll
            for (int a = 0; a < 3; ++a)
                if (W7_AW[KA][a][i] != 0.0f) vfma(s[a][j], W7_AW[KA][a][i], q);
        }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int j = 0; j < NB; ++j)
                if (W7_AW[KB][b][j] != 0.0f) vfma(acc[a][b], W7_AW[KB][b][j], s[a][j]);
}

// one lane = VW consecutive output channels of one input channel (VW = 4 needs Cout % 4 == 0, else the launcher takes VW = 1)
template <int VW>
__global__ __launch_bounds__(256) void wino7_wgrad_out_kernel(const float* __restrict__ dU, float* __restrict__ dW, int Cout,
                                                              int Cin) {
    typedef typename w7_vec<VW>::T VT;
    const size_t idx = ((size_t)blockIdx.x * 256 + threadIdx.x) * VW;     // (ci, co), co fastest
    const size_t KN = (size_t)Cin * Cout;
    if (idx >= KN) return;
    const int ci = (int)(idx / Cout), co = (int)(idx % Cout);
    VT acc[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) vzero(acc[a][b]);
    w7_wgrad_tile<0, 0, VT>(dU, KN, idx, acc);
    w7_wgrad_tile<0, 1, VT>(dU, KN, idx, acc);
    w7_wgrad_tile<1, 0, VT>(dU, KN, idx, acc);
    w7_wgrad_tile<1, 1, VT>(dU, KN, idx, acc);
#pragma unroll
    for (int l = 0; l < VW; ++l) {
        float* dst = dW + ((size_t)(co + l) * Cin + ci) * 9;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) dst[a * 3 + b] = vget(acc[a][b], l);
    }
}

// The same transform with COALESCED stores: the lanes of a wave run along co (dU's contiguous index), but dW [Cout][Cin][3][3] keeps
// co OUTERMOST - a lane's nine results are 36 bytes, the next lane's 73 KB away: 576 scattered 4-byte stores per wave cost half of
// the kernel (0.213 ms; 0.104 with the stores removed).  Workgroup = 64 co x CIM_W7_WG_CI ci (wave w takes ci0 + w): the results of a
// co row pass through LDS and leave as one run of 9 CIM_W7_WG_CI floats per row.  Bit-identical results.
// grid = (Cout / 64, Cin / CIM_W7_WG_CI); block = 64 CIM_W7_WG_CI.
#ifndef CIM_W7_WG_CI
#define CIM_W7_WG_CI 4
#endif
__global__ __launch_bounds__(64 * CIM_W7_WG_CI) void wino7_wgrad_out_rows_kernel(const float* __restrict__ dU, float* __restrict__ dW,
                                                                                 int Cout, int Cin) {
    constexpr int NCI = CIM_W7_WG_CI, ROW = 9 * NCI;
    __shared__ float t[64][ROW + 1];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int co0 = blockIdx.x * 64, ci0 = blockIdx.y * NCI;
    const size_t KN = (size_t)Cin * Cout;
    const size_t idx = (size_t)(ci0 + w) * Cout + co0 + lane;
    float acc[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) acc[a][b] = 0.0f;
    w7_wgrad_tile<0, 0, float>(dU, KN, idx, acc);
    w7_wgrad_tile<0, 1, float>(dU, KN, idx, acc);
    w7_wgrad_tile<1, 0, float>(dU, KN, idx, acc);
    w7_wgrad_tile<1, 1, float>(dU, KN, idx, acc);
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) t[lane][w * 9 + a * 3 + b] = acc[a][b];
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * ROW; e += 64 * NCI) {
        const int row = e / ROW, col = e - row * ROW;
        dW[((size_t)(co0 + row) * Cin + ci0) * 9 + col] = t[row][col];
    }
}

// ---------------------------------------------------------------------------------------------
// (c, h, w) flatten of the conv output and its adjoint, fused with the ReLU mask.
// The reference flattens MaskFuse's NCHW conv output with .view(N, -1) (resnet50.py:135): seg_fc.0's weight columns
// are in (c, h, w) order.  The kernels here keep activations channels-last, so the flatten is a per-ROI [PP][C] ->
// [C][PP] transpose (forward) and, in the backward, the transpose back fused with the conv's ReLU mask
// (dy = dflat^T * (y > 0)): one pass each instead of a strided copy + compare + multiply.
// Workgroup = (ROI, 64-channel chunk), staged through a [PP][65] LDS tile: both global sides are contiguous runs
// (256 B along C, 64 * PP floats along (c, p)).
// bsum (backward, optional): [R][C] per-ROI sums over the PP pixels of the masked gradient - the conv's bias gradient is
// their sum over R (one small deterministic reduce instead of a pass over the 200 MB gradient).
template <bool FWD>
__global__ __launch_bounds__(256) void flatten_chw_kernel(const float* __restrict__ src, const float* __restrict__ y,
                                                          float* __restrict__ dst, int PP, int C, float* __restrict__ bsum) {
    __shared__ float t[64][65];
    const int r = blockIdx.x, c0 = blockIdx.y * 64, tid = threadIdx.x;
    const size_t base = (size_t)r * PP * C;
    if (FWD) {
        for (int e = tid; e < PP * 64; e += 256) {
            const int p = e >> 6, c = e & 63;
            t[p][c] = src[base + (size_t)p * C + c0 + c];
        }
        __syncthreads();
        for (int e = tid; e < PP * 64; e += 256) {
            const int c = e / PP, p = e - c * PP;
            dst[base + (size_t)c0 * PP + e] = t[p][c];
        }
    } else {
        for (int e = tid; e < PP * 64; e += 256) {
            const int c = e / PP, p = e - c * PP;
            t[p][c] = src[base + (size_t)c0 * PP + e];
        }
        __syncthreads();
        float part = 0.0f;
        for (int e = tid; e < PP * 64; e += 256) {
            const int p = e >> 6, c = e & 63;
            const size_t o = base + (size_t)p * C + c0 + c;
            const float v = (y == nullptr || y[o] > 0.0f) ? t[p][c] : 0.0f;
            dst[o] = v;
            part += v;                       // this thread's channel is tid & 63 in every iteration
        }
        if (bsum != nullptr) {
            __syncthreads();
            t[tid >> 6][tid & 63] = part;
            __syncthreads();
            if (tid < 64) bsum[(size_t)r * C + c0 + tid] = (t[0][tid] + t[1][tid]) + (t[2][tid] + t[3][tid]);
        }
    }
}

// Forward flatten straight into a pair image: src [R][PP][C] fp32 -> dst [Rs][C * PP] pair image in (c, p) order (seg_fc.0's
// A operand and, contracted over its rows, the weight gradient's B operand).  A thread packs 8 consecutive (c, p) elements
// = one 32-byte chunk.  Rows R .. Rs are zeroed.  grid = (Rs, C / 64)
__global__ __launch_bounds__(256) void flatten_chw_pair_kernel(const float* __restrict__ src, float* __restrict__ dst, int R,
                                                               int PP, int C, const float* __restrict__ scale) {
    __shared__ float t[64][65];
    const int r = blockIdx.x, c0 = blockIdx.y * 64, tid = threadIdx.x;
    float* out = dst + (size_t)r * PP * C + (size_t)c0 * PP;
    const int chunks = PP * 8;                  // 64 * PP / 8
    if (r >= R) {
        for (int q = tid; q < chunks * 2; q += 256) *reinterpret_cast<float4*>(out + q * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const size_t base = (size_t)r * PP * C;
    for (int e = tid; e < PP * 64; e += 256) {
        const int p = e >> 6, c = e & 63;
        t[p][c] = src[base + (size_t)p * C + c0 + c];
    }
    __syncthreads();
    const float s = scale[0];
    for (int q = tid; q < chunks; q += 256) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int e = q * 8 + k, c = e / PP, p = e - c * PP;
            v[k] = t[p][c] * s;
        }
        unsigned hh[4], ll[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) cim::pair_split2(v[2 * e], v[2 * e + 1], hh[e], ll[e]);
        const w7_u4 h = {hh[0], hh[1], hh[2], hh[3]}, l = {ll[0], ll[1], ll[2], ll[3]};
        *reinterpret_cast<w7_u4*>(out + q * 8) = h;
        *reinterpret_cast<w7_u4*>(out + q * 8 + 4) = l;
    }
}

}  // namespace

#define WINO_GEOM_OK() CIM_CHECK_ARG(R > 0 && P > 0 && P <= 64 && C > 0 && C % 4 == 0)

extern "C" int cim_wino_wgrad_output(const float* dU, float* dW, int Cout, int Cin, int tile, void* stream) {
    CIM_CHECK_ARG(dU && dW && Cout > 0 && Cin > 0);
    CIM_CHECK_ARG(tile == 7);         // (the F(2x2,3x3) / F(4x4,3x3) stages are test infrastructure: experiments/csrc/winograd_all.hip)
    const size_t n = (size_t)Cout * Cin;
    if (Cout % 64 == 0 && Cin % CIM_W7_WG_CI == 0 && Cin / CIM_W7_WG_CI <= 65535) {
        hipLaunchKernelGGL(wino7_wgrad_out_rows_kernel, dim3(Cout / 64, Cin / CIM_W7_WG_CI), dim3(64 * CIM_W7_WG_CI), 0, cim::as_stream(stream),
                           dU, dW, Cout, Cin);
        CIM_CHECK_LAUNCH();
        return 0;
    }
    if (CIM_W7_VWG == 4 && Cout % 4 == 0)
        hipLaunchKernelGGL(wino7_wgrad_out_kernel<4>, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, cim::as_stream(stream), dU, dW, Cout, Cin);
    else hipLaunchKernelGGL(wino7_wgrad_out_kernel<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, cim::as_stream(stream), dU, dW, Cout, Cin);
    CIM_CHECK_LAUNCH();
    return 0;
}

extern "C" int cim_wino_dx_adjoint_output(const float* M, float* dx, int R, int P, int C, int tile, void* stream) {
    WINO_GEOM_OK();
    CIM_CHECK_ARG(M && dx && tile == 7 && P == 7 && R <= 2147483647);
    int chunks = (C + 256 * CIM_W7_VDX - 1) / (256 * CIM_W7_VDX);
    if (chunks > 4) chunks = 4;
    hipLaunchKernelGGL(wino7_dx_kernel, dim3(R, chunks), dim3(256), 0, cim::as_stream(stream), M, dx, R, C);
    CIM_CHECK_LAUNCH();
    return 0;
}

extern "C" int cim_wino7_dx_maskfold(const float* M, const float* masks, float* dbox, int R, int Cb, void* stream) {
    CIM_CHECK_ARG(M && masks && dbox && R > 0 && Cb > 0 && Cb % 4 == 0);
    int chunks = (Cb + 256 * CIM_W7_VMF - 1) / (256 * CIM_W7_VMF);
    if (chunks > 4) chunks = 4;
    hipLaunchKernelGGL(wino7_dx_maskfold_kernel, dim3(R, chunks), dim3(256), 0, cim::as_stream(stream), M, masks, dbox, R, Cb);
    CIM_CHECK_LAUNCH();
    return 0;
}

// ---- pair-image producers of the f16x2p engine (tile = 7 geometry: 121 positions) ---------------------------------------
extern "C" int cim_wino7_pair_scales(const uint32_t* amax, int n_amax, const uint32_t* amax_mul, int kind, float* scale, void* stream) {
    CIM_CHECK_ARG(amax && scale && kind >= 0 && kind <= 3 && n_amax >= 1 && n_amax <= 65536);
    hipLaunchKernelGGL(wino7_pair_scales_kernel, dim3(1), dim3(128), 0, cim::as_stream(stream), amax, n_amax, amax_mul, kind, scale);
    CIM_CHECK_LAUNCH();
    return 0;
}

extern "C" int cim_wino7_input_pair(const float* x, void* V, const float* scale, int R, int Rs, int C, void* stream) {
    CIM_CHECK_ARG(x && V && scale && R > 0 && Rs >= R && C > 0 && C % 8 == 0);
    hipLaunchKernelGGL(wino7_input_pair_kernel, dim3(Rs, 4), dim3(256), 0, cim::as_stream(stream), x, (float*)V, R, Rs, C, scale);
    CIM_CHECK_LAUNCH();
    return 0;
}

extern "C" int cim_wino7_filter_pair(const float* W, void* U, const float* scale, int Cout, int Cin, void* stream) {
    CIM_CHECK_ARG(W && U && scale && Cout > 0 && Cin > 0 && Cin % 8 == 0);
    const size_t n = (size_t)Cout * Cin;
    hipLaunchKernelGGL(wino7_filter_pair_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, cim::as_stream(stream), W,
                       (float*)U, Cout, Cin, scale);
    CIM_CHECK_LAUNCH();
    return 0;
}

extern "C" int cim_wino7_dy_pair(const float* dy, void* D, const float* scale, int R, int Rs, int C, int adjoint, void* stream) {
    CIM_CHECK_ARG(dy && D && scale && R > 0 && Rs >= R && C > 0 && C % 8 == 0);
    if (adjoint) hipLaunchKernelGGL(wino7_dy_pair_kernel<true>, dim3(Rs, 4), dim3(256), 0, cim::as_stream(stream), dy, (float*)D, R, Rs, C, scale);
    else hipLaunchKernelGGL(wino7_dy_pair_kernel<false>, dim3(Rs, 4), dim3(256), 0, cim::as_stream(stream), dy, (float*)D, R, Rs, C, scale);
    CIM_CHECK_LAUNCH();
    return 0;
}

extern "C" int cim_wino7_flatten_bwd_dy_pair(const float* dX, const float* relu_y, void* E, const float* scale_e, void* D,
                                             const float* scale_d, float* bias_partial, int R, int Rs, int C, void* stream) {
    CIM_CHECK_ARG(dX && (E || D) && (!E || scale_e) && (!D || scale_d) && R > 0 && Rs >= R && C > 0 && C % 256 == 0 && C / 256 <= 65535);
    hipLaunchKernelGGL(wino7_flatten_bwd_dy_pair_kernel, dim3(Rs, C / 256), dim3(256), 0, cim::as_stream(stream), dX, relu_y, (float*)E,
                       (float*)D, bias_partial, R, Rs, C, scale_e, scale_d);
    CIM_CHECK_LAUNCH();
    return 0;
}

extern "C" int cim_wino7_output_amax(const float* M, const float* bias, float* y, int R, int C, int relu, uint32_t* y_amax,
                                     void* stream) {
    CIM_CHECK_ARG(M && y && R > 0 && C > 0 && C % 4 == 0);
    hipLaunchKernelGGL(wino7_output_kernel, dim3(R, 4), dim3(256), 0, cim::as_stream(stream), M, bias, y, R, C, relu, y_amax);
    CIM_CHECK_LAUNCH();
    return 0;
}

extern "C" int cim_flatten_chw_pair(const float* src, void* dst, const float* scale, int R, int Rs, int PP, int C, void* stream) {
    CIM_CHECK_ARG(src && dst && scale && R > 0 && Rs >= R && PP > 0 && PP <= 64 && C > 0 && C % 64 == 0 && C / 64 <= 65535);
    hipLaunchKernelGGL(flatten_chw_pair_kernel, dim3(Rs, C / 64), dim3(256), 0, cim::as_stream(stream), src, (float*)dst, R, PP, C, scale);
    CIM_CHECK_LAUNCH();
    return 0;
}

extern "C" int cim_flatten_chw_bwd_bias(const float* src, const float* relu_y, float* dst, float* bias_partial, int R, int PP,
                                        int C, void* stream) {
    CIM_CHECK_ARG(src && dst && R > 0 && PP > 0 && PP <= 64 && C > 0 && C % 64 == 0 && C / 64 <= 65535);
    hipLaunchKernelGGL(flatten_chw_kernel<false>, dim3(R, C / 64), dim3(256), 0, cim::as_stream(stream), src, relu_y, dst, PP, C, bias_partial);
    CIM_CHECK_LAUNCH();
    return 0;
}
