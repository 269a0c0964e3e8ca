// fp32 MFMA GEMM family for the MaskFuse head (~98 % of the step's FLOPs).
//
// Replaces the dense contractions of MaskFuse.forward / backward,
// /root/reference/lib/modeling/resnet50.py:104-110,135-136: the 3x3 conv 2C->C on N x 7 x 7
// (as an implicit GEMM: the im2col matrix is never materialised), Linear(49C->4096),
// Linear(4096->4096), and their data / weight gradients.
//
// Arithmetic: fp32 in, fp32 accumulate, fp32 out.  Three engines execute the multiplies:
//   gemm_f32_kernel     v_mfma_f32_32x32x2_f32 (f32 products; bitwise a k-ordered fmaf chain), 157.3 TFLOP/s peak
//   gemm_bf16x3_kernel  exact three-term bf16 split of both operands, 6 bf16 MFMA products per multiply-add
//   gemm_f16x2_kernel   scaled two-term fp16 split (23 of 24 significand bits), 3 f16 MFMA products (default)
// The split engines are in the error class of the f32-multiply engine (tests/test_gpu_gemm.py, tests/test_gpu_tolerance.py).
//
// Tiling (64-wide waves): workgroup tile BM x BN = 256 x 256, 8 waves as 2(M) x 4(N), each wave
// 128 x 64 = 4 x 2 MFMA tiles of 32 x 32 (128 accumulator VGPRs).  K is consumed in slabs of
// BK = 16 staged through LDS as [k][m] / [k][n] (row stride +4 floats), so each MFMA operand
// fetch is one conflict-free ds_read_b32 of 32 consecutive floats per half-wave.  Global->LDS is
// register-staged and double-buffered: the loads of slab t+1 are issued before the MFMAs of
// slab t; three LDS slabs rotate so the LDS write + barrier sit in the middle of a slab's MFMAs.
// Operand layouts are template parameters (no runtime dispatch inside the loop):
//   A: K-contiguous rows (activations [M,K], optionally gathered as 3x3-conv patches) or
//      M-contiguous rows ([K,M], e.g. dY^T / im2col^T for weight gradients)
//   B: N-contiguous rows ([K,N]) or K-contiguous rows ([N,K], nn.Linear weights)
#include "common.h"
#include "../../include/cim_hip.h"
#include "../include/cim_exp.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

#ifndef CIM_GEMM_BK
#define CIM_GEMM_BK 16
#endif
#ifndef CIM_GEMM_EXP
#define CIM_GEMM_EXP 0   // ablation switches (tools/bench_gemm_ab.py); 0 = product
#endif
#ifndef CIM_GEMM_BN
#define CIM_GEMM_BN 256
#endif
#ifndef CIM_GEMM_WAVES_M
#define CIM_GEMM_WAVES_M 2
#endif
#ifndef CIM_GEMM_MINW
#define CIM_GEMM_MINW 2      // waves per SIMD the register allocator must leave room for (2 = one workgroup per CU)
#endif
constexpr int BM = 256, BN = CIM_GEMM_BN, BK = CIM_GEMM_BK;
constexpr int WAVES_M = CIM_GEMM_WAVES_M, WAVES_N = 8 / CIM_GEMM_WAVES_M;
constexpr int RESIDENT = CIM_GEMM_MINW / 2;   // co-resident 512-thread workgroups per CU
constexpr int NT = 64 * WAVES_M * WAVES_N;      // 512 threads
constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;   // 128 x 64 per wave
constexpr int MI = WM / 32, NI = WN / 32;        // 4 x 2 MFMA tiles
constexpr int LDS_A = BM + 4, LDS_B = BN + 4;    // padded row strides (floats)
constexpr int SLAB = BK * (LDS_A + LDS_B);       // floats per LDS buffer

enum ALayout { A_KCONTIG = 0, A_MCONTIG = 1, A_CONV_K = 2, A_CONV_M = 3 };
enum BLayout { B_NCONTIG = 0, B_KCONTIG = 1 };

#ifndef CIM_X2_NT
#define CIM_X2_NT 0             // 1 = nontemporal stores of the f16x2 engine's final outputs
#endif
struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    const float* bias;   // [N] or null
    int M, N, K;
    int lda, ldb, ldc;
    int relu;
    int k_per_split;     // multiple of BK
    long long c_split_stride;   // elements between split-K partial outputs
    // implicit 3x3 conv geometry (A_CONV_*): A is X[R, P, P, Cin] (NHWC)
    int P, Cin;
    // batched mode (batch > 1, no split-K): blockIdx.z selects the problem; strides in elements
    int batch;
    long long a_bs, b_bs, c_bs;
    // f16x2 engine: |max| bit patterns per A row [batch][M] and per B column [batch][N] (cim_amax_rowcol)
    const unsigned* a_amax;
    const unsigned* b_amax;
};


// ---- bf16x3 engine: LDS layout + operand split -------------------------------------------------
// Every fp32 operand element x is split EXACTLY into three bf16 terms x = h + m + l
// (h = rne_bf16(x), m = rne_bf16(x - h), l = rne_bf16(x - h - m): 3 x 8 significant bits), and
// a*b is evaluated as the six MFMA products hh + hm + mh + mm + hl + lh (every bf16 x bf16 product
// is exact in the fp32 accumulator; the dropped ml + lm + ll terms are < 2^-23 |a*b|, measured
// 2e-8 * sqrt(K) against 1.6e-6 * sqrt(K) of an fp32 accumulation chain - tools/gen_winograd.py
// style check in tests/test_gpu_gemm.py::test_bf16x3_error_class).  The products run on
// v_mfma_f32_32x32x16_bf16 (2.5 PFLOP/s dense; 6 products -> 419 TFLOP/s fp32-equivalent peak
// against 157 TFLOP/s of v_mfma_f32_32x32x2_f32).
//
// LDS, per operand and 16-k slab: [plane h|m|l][k-group of 8][256 rows][8 bf16 = 16 B], so one
// ds_read_b128 is a lane's whole MFMA operand (row = lane & 31, k-group = lane >> 5).  Rows are stored
// at slot(r) = (r & ~3) | ((r + (r >> 4)) & 3): the 16-lane groups of ds_read_b128 and the 32-lane
// groups of the M-contiguous writers' ds_write_b32 (lanes 4 rows apart) both cover all 16 bank quads.
constexpr int X3_KG = BM * 16 + 64;        // bytes per (plane, k-group): 256 rows x 16 B (+64: the two k-groups land 16 banks apart)
constexpr int X3_PLANE = 2 * X3_KG;
constexpr int X3_OPER = 3 * X3_PLANE;
constexpr int X3_SLAB = 2 * X3_OPER;       // A + B (BM == BN)
static_assert(BM == BN && BK == 16, "bf16x3 engine: 256 x 256 x 16 slabs");

__device__ __forceinline__ int x3_slot(int r) { return (r & ~3) | ((r + (r >> 4)) & 3); }
typedef __bf16 x3_bf16x2 __attribute__((ext_vector_type(2)));
typedef float x3_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned x3_cvt_pk(float lo, float hi) {      // v_cvt_pk_bf16_f32 (round to nearest even)
    const x3_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, x3_bf16x2));
}
// (a, b) -> packed bf16 pairs of the three terms; a in the low half
#ifndef CIM_X3_EXP
#define CIM_X3_EXP 0     // ablation switches for tools/bench_gemm_ab.py; 0 = product
#endif
__device__ __forceinline__ void x3_split(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
#if CIM_X3_EXP == 1
    h = m = l = __float_as_uint(a) ^ __float_as_uint(b);
    return;
#endif
    h = x3_cvt_pk(a, b);
    a -= __uint_as_float(h << 16);
    b -= __uint_as_float(h & 0xffff0000u);
    m = x3_cvt_pk(a, b);
    a -= __uint_as_float(m << 16);
    b -= __uint_as_float(m & 0xffff0000u);
    l = x3_cvt_pk(a, b);
}
// K-contiguous piece: row r, k = 4q .. 4q+3
__device__ __forceinline__ void x3_store_k(char* op, int r, int q, const float4& v) {
    unsigned h0, m0, l0, h1, m1, l1;
    x3_split(v.x, v.y, h0, m0, l0);
    x3_split(v.z, v.w, h1, m1, l1);
    char* d = op + (q >> 1) * X3_KG + x3_slot(r) * 16 + (q & 1) * 8;
    *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(d + X3_PLANE) = make_uint2(m0, m1);
    *reinterpret_cast<uint2*>(d + 2 * X3_PLANE) = make_uint2(l0, l1);
}
// M-contiguous pieces: rows 4mq .. 4mq+3 at k = 2kr (v[0]) and 2kr + 1 (v[1])
__device__ __forceinline__ void x3_store_m(char* op, int mq, int kr, const float4 (&v)[2]) {
    const float a[4] = {v[0].x, v[0].y, v[0].z, v[0].w}, b[4] = {v[1].x, v[1].y, v[1].z, v[1].w};
    char* d = op + (kr >> 2) * X3_KG + (kr & 3) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned h, m, l;
        x3_split(a[j], b[j], h, m, l);
        char* dj = d + x3_slot(mq * 4 + j) * 16;
        *reinterpret_cast<unsigned*>(dj) = h;
        *reinterpret_cast<unsigned*>(dj + X3_PLANE) = m;
        *reinterpret_cast<unsigned*>(dj + 2 * X3_PLANE) = l;
    }
}


// ---- f16x2 engine: scaled two-term fp16 split ---------------------------------------------------
// Each fp32 operand element x is scaled by a power of two per A row / per B column (s = 2^(14-e),
// e = exponent of the row's / column's largest magnitude over K, so |x*s| < 2^15: exact, no overflow)
// and split into two fp16 terms  x*s = h + l + d,  h = rne_f16(x*s), l = rne_f16(x*s - h):
// 11 + 11 significant bits + the sign of l = 23 bits, |d| <= 2^-23 |x*s| (zero for 3 operands in 4).
// a*b is evaluated as the three MFMA products  hl + lh + hh  on v_mfma_f32_32x32x16_f16 (each fp16 x
// fp16 product is exact in the fp32 accumulator); the dropped l*l term is <= 2^-22 |a*b| (rms 2^-25.6).
// The accumulator is rescaled by 2^-(ea+eb) in the epilogue (exact).  Elements more than 2^16 below
// their row / column maximum fall into fp16's subnormal range and keep an ABSOLUTE accuracy of
// 2^-40 of that maximum - so the result carries the error bound of an fp32 GEMM relative to
// |a_row| * |b_col| (tests/test_gpu_gemm.py::test_f16x2_error_class).  Half the MFMA work of bf16x3.
//
// LDS per operand and 16-k slab: [plane h|l][k-group of 8][256 rows][8 f16 = 16 B], rows swizzled by
// x3_slot() exactly as in the bf16x3 engine.
#ifndef CIM_X2_EXP
#define CIM_X2_EXP 0     // ablation switches for tools/bench_gemm_ab.py; 0 = product
#endif
constexpr int X2_KG = BM * 16 + 64;
constexpr int X2_PLANE = 2 * X2_KG;
constexpr int X2_OPER = 2 * X2_PLANE;
constexpr int X2_SLAB = 2 * X2_OPER;       // 33280 B; three buffers = 99840 B
typedef _Float16 x2_f16x2 __attribute__((ext_vector_type(2)));

// amax bit pattern -> power-of-two scale (biased exponent clamped to [14, 253] so the inverse is normal)
__device__ __forceinline__ float x2_scale(unsigned amax_bits) {
    const int e = (int)((amax_bits >> 23) & 0xffu);
    int b = min(268 - e, 253);
    if (e == 0 || e == 255) b = 127;        // all-zero (or non-finite) row: scale 1
    return __uint_as_float((unsigned)b << 23);
}
__device__ __forceinline__ float x2_inv(float s) { return __uint_as_float((254u << 23) - __float_as_uint(s)); }

// (a, b) already scaled -> packed f16 pairs (a in the low half) of the two terms
// (tried: l = f16(a * s - h) as ONE v_fma_mixlo_f16 / v_fma_mixhi_f16 per element - scaling, f16 -> f32 of h, the exact
// subtraction and the rounding fused, 4 VALU per pair instead of 8, bit-identical: 1.807 vs 1.801 ms on the Winograd
// forward GEMM, no change on the others - the split arithmetic is not what co-limits these kernels; with only the MFMAs
// left (CIM_X2_EXP=7) the same launch takes 1.204 ms)
__device__ __forceinline__ void x2_split(float a, float b, unsigned& h, unsigned& l) {
#if CIM_X2_EXP == 6
    h = l = __float_as_uint(a) ^ __float_as_uint(b);
    return;
#endif
    const x3_f32x2 v = {a, b};
    const x2_f16x2 hv = __builtin_convertvector(v, x2_f16x2);          // v_cvt_pk_f16_f32 (RNE)
    const x3_f32x2 r = {a - (float)hv.x, b - (float)hv.y};               // exact
    h = __builtin_bit_cast(unsigned, hv);
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(r, x2_f16x2));
}
// K-contiguous piece: row r, k = 4q .. 4q+3, one scale for the row
__device__ __forceinline__ void x2_store_k(char* op, int r, int q, const float4& v, float s) {
    unsigned h0, l0, h1, l1;
    x2_split(v.x * s, v.y * s, h0, l0);
    x2_split(v.z * s, v.w * s, h1, l1);
    char* d = op + (q >> 1) * X2_KG + x3_slot(r) * 16 + (q & 1) * 8;
    *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(d + X2_PLANE) = make_uint2(l0, l1);
}
// M-contiguous pieces: rows 4mq .. 4mq+3 (scales s.x .. s.w) at k = 2kr (v[0]) and 2kr + 1 (v[1])
__device__ __forceinline__ void x2_store_m(char* op, int mq, int kr, const float4 (&v)[2], const float4& s, float m0, float m1) {
    // m0 / m1 = 1 (k row inside the problem) or 0 (K tail: the loader re-read a valid row; its product with 0 is 0)
    const float a[4] = {v[0].x * (s.x * m0), v[0].y * (s.y * m0), v[0].z * (s.z * m0), v[0].w * (s.w * m0)};
    const float b[4] = {v[1].x * (s.x * m1), v[1].y * (s.y * m1), v[1].z * (s.z * m1), v[1].w * (s.w * m1)};
    char* d = op + (kr >> 2) * X2_KG + (kr & 3) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned h, l;
        x2_split(a[j], b[j], h, l);
        char* dj = d + x3_slot(mq * 4 + j) * 16;
        *reinterpret_cast<unsigned*>(dj) = h;
        *reinterpret_cast<unsigned*>(dj + X2_PLANE) = l;
    }
}
__device__ __forceinline__ float4 x2_scale4(const unsigned* amax, int i0, int limit) {
    return make_float4(x2_scale(amax[min(i0, limit - 1)]), x2_scale(amax[min(i0 + 1, limit - 1)]),
                       x2_scale(amax[min(i0 + 2, limit - 1)]), x2_scale(amax[min(i0 + 3, limit - 1)]));
}

// Thread -> (row quad, k pair) of the M/N-contiguous loaders.  A lane writes 4 bytes (its k pair) into 16 B row slots,
// so the LDS bank of a write is 4*(slot mod 8) + (k pair & 3) (+16 for the second k-group): for the 32 lanes of a
// ds_write_b32 group to cover all 32 banks they must span 4 k pairs x 8 slot residues - k pair from lane bits 0-1, the
// quad's bit 0 from lane bit 2 and its bits 2-3 (which rotate the slot swizzle) from lane bits 3-4.  With one k pair per
// wave (the first layout) all lanes wrote the same 4-byte column of slots 64 B apart: 8 banks, 4-way conflicts,
// SQ_LDS_BANK_CONFLICT = 37-54 % of the LDS cycles of the kernels with such an operand (0 for K-contiguous ones).
// Global side: a wave-load covers 4 k rows x 256 contiguous bytes.
static_assert(BM == 256 && BN == 256 && NT == 512 && BK == 16, "mn_lane_map: 64 row quads x 8 k pairs over 512 threads");
__device__ __forceinline__ void mn_lane_map(int tid, int& quad, int& kpair) {
    const int l = tid & 63, w = tid >> 6;
    kpair = (l & 3) | ((w & 1) << 2);
    quad = ((l >> 2) & 1) | (((l >> 5) & 1) << 1) | (((l >> 3) & 3) << 2) | ((w >> 1) << 4);
}

// ---- A operand -------------------------------------------------------------------------------
// K-contiguous: each thread owns float4 pieces (row, 4 consecutive k); BM*BK/4/NT = 2 pieces.
template <int AL>
struct ALoaderK {
    static constexpr int PIECES = BM * BK / 4 / NT;
    const float* ptr[PIECES];   // row base (+ q*4), null when the row is out of range
    int row[PIECES];
    int oh[PIECES], ow[PIECES];
    int q;
    float sc[PIECES];      // f16x2 engine: row scales

    __device__ __forceinline__ void init_scale(const GemmArgs& g, int m0) {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) sc[i] = x2_scale(g.a_amax[min(m0 + row[i], g.M - 1)]);
    }
    // raw loads (load<true>) carry no K-tail select, so nothing waits on them until this store: the tail
    // is zeroed here through the scale (k0 = first k of the slab held in v)
    __device__ __forceinline__ void store2(char* op, const float4 (&v)[PIECES], int k0, int kend) const {
        const float m = (k0 + q * 4 < kend) ? 1.0f : 0.0f;
#pragma unroll
        for (int i = 0; i < PIECES; ++i) x2_store_k(op, row[i], q, v[i], sc[i] * m);
    }
    __device__ __forceinline__ void init(const GemmArgs& g, int m0, int tid) {
        q = tid & 3;
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            row[i] = (tid >> 2) + i * (NT / 4);
            const int m = m0 + row[i];
            if (AL == A_CONV_K) {
                const int pp = g.P * g.P;
                const int p = m % pp;
                oh[i] = p / g.P;
                ow[i] = p % g.P;
                ptr[i] = (m < g.M) ? g.A + (size_t)m * g.Cin + q * 4 : nullptr;
            } else {
                // branch-free: rows >= M re-read row M-1 (their C rows are never stored)
                ptr[i] = g.A + (size_t)min(m, g.M - 1) * g.lda;
            }
        }
    }
    template <bool RAW = false>
    __device__ __forceinline__ void load(const GemmArgs& g, int k0, float4 (&v)[PIECES]) const {
        if constexpr (AL != A_CONV_K) {
            const int k = k0 + q * 4;
            const int kc = min(k, g.K - 4);
            const bool in = RAW || k < g.K;
#pragma unroll
            for (int i = 0; i < PIECES; ++i) {
                const float4 t = *reinterpret_cast<const float4*>(ptr[i] + kc);
                v[i] = make_float4(in ? t.x : 0.f, in ? t.y : 0.f, in ? t.z : 0.f, in ? t.w : 0.f);
            }
            return;
        }
        int dy = 0, dx = 0, ci0 = k0;
        if (AL == A_CONV_K) {   // slab = 16 channels of one tap (Cin % BK == 0)
            const int tap = k0 / g.Cin;
            ci0 = k0 - tap * g.Cin;
            dy = tap / 3 - 1;
            dx = tap % 3 - 1;
        }
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ptr[i] == nullptr) continue;
            if (AL == A_CONV_K) {
                const int y = oh[i] + dy, x = ow[i] + dx;
                if ((unsigned)y < (unsigned)g.P && (unsigned)x < (unsigned)g.P)
                    v[i] = *reinterpret_cast<const float4*>(ptr[i] + (ptrdiff_t)(dy * g.P + dx) * g.Cin + ci0);
            } else if (k0 + q * 4 < g.K) {
                v[i] = *reinterpret_cast<const float4*>(ptr[i] + k0);
            }
        }
    }
    __device__ __forceinline__ void store(float* as, const float4 (&v)[PIECES]) const {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            float* d = as + (q * 4) * LDS_A + row[i];
            d[0 * LDS_A] = v[i].x;
            d[1 * LDS_A] = v[i].y;
            d[2 * LDS_A] = v[i].z;
            d[3 * LDS_A] = v[i].w;
        }
    }
    __device__ __forceinline__ void store3(char* op, const float4 (&v)[PIECES]) const {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) x3_store_k(op, row[i], q, v[i]);
    }
};

// M-contiguous: element (m, k) at A[k*lda + m]; thread owns float4 pieces (k row, 4 consecutive m).
template <int AL>
struct ALoaderM {
    static constexpr int COLS4 = BM / 4, ROWS = NT / COLS4, PIECES = BK / ROWS;
    int mq, krow0;
    const float* base;   // column base (A + m), null when out of range
    int dy, dx;
    float4 sc;             // f16x2 engine: scales of the 4 owned rows

    __device__ __forceinline__ void init_scale(const GemmArgs& g, int m0) { sc = x2_scale4(g.a_amax, m0 + mq * 4, g.M); }
    __device__ __forceinline__ void store2(char* op, const float4 (&v)[PIECES], int k0, int kend) const {
        const int k = k0 + krow0 * PIECES;
        x2_store_m(op, mq, krow0, v, sc, k < kend ? 1.0f : 0.0f, k + 1 < kend ? 1.0f : 0.0f);
    }
    __device__ __forceinline__ void init(const GemmArgs& g, int m0, int tid) {
        mn_lane_map(tid, mq, krow0);
        const int m = m0 + mq * 4;
        if (AL == A_CONV_M) {   // m = (tap, ci); 4 consecutive m share the tap (Cin % 4 == 0)
            const int tap = m / g.Cin;
            dy = tap / 3 - 1;
            dx = tap % 3 - 1;
            base = (m < g.M) ? g.A + (m - tap * g.Cin) : nullptr;
        } else {
            dy = dx = 0;
            base = g.A + min(m, g.M - 4);       // branch-free: columns >= M re-read the last quad
        }
    }
    template <bool RAW = false>
    __device__ __forceinline__ void load(const GemmArgs& g, int k0, int kend, float4 (&v)[PIECES]) const {
        if constexpr (AL != A_CONV_M) {
#pragma unroll
            for (int i = 0; i < PIECES; ++i) {
                const int k = k0 + krow0 * PIECES + i;
                const float4 t = *reinterpret_cast<const float4*>(base + (size_t)min(k, g.K - 1) * g.lda);
                const bool in = RAW || k < kend;
                v[i] = make_float4(in ? t.x : 0.f, in ? t.y : 0.f, in ? t.z : 0.f, in ? t.w : 0.f);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int k = k0 + krow0 * PIECES + i;          // a thread owns PIECES consecutive k rows
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (base == nullptr || k >= kend) continue;
            if (AL == A_CONV_M) {   // k = output pixel row (roi, oh, ow); gather the tap-shifted input pixel
                const int pp = g.P * g.P;
                const int p = k % pp;
                const int y = p / g.P + dy, x = p % g.P + dx;
                if ((unsigned)y < (unsigned)g.P && (unsigned)x < (unsigned)g.P)
                    v[i] = *reinterpret_cast<const float4*>(base + (size_t)(k + dy * g.P + dx) * g.Cin);
            } else {
                v[i] = *reinterpret_cast<const float4*>(base + (size_t)k * g.lda);
            }
        }
    }
    __device__ __forceinline__ void store(float* as, const float4 (&v)[PIECES]) const {
#pragma unroll
        for (int i = 0; i < PIECES; ++i)
            *reinterpret_cast<float4*>(as + (krow0 * PIECES + i) * LDS_A + mq * 4) = v[i];
    }
    __device__ __forceinline__ void store3(char* op, const float4 (&v)[PIECES]) const { x3_store_m(op, mq, krow0, v); }
};

// ---- B operand -------------------------------------------------------------------------------
struct BLoaderN {   // element (k, n) at B[k*ldb + n]
    static constexpr int COLS4 = BN / 4, ROWS = NT / COLS4, PIECES = BK / ROWS;
    int nq, krow0;
    const float* base;
    float4 sc;
    __device__ __forceinline__ void init_scale(const GemmArgs& g, int n0) { sc = x2_scale4(g.b_amax, n0 + nq * 4, g.N); }
    __device__ __forceinline__ void store2(char* op, const float4 (&v)[PIECES], int k0, int kend) const {
        const int k = k0 + krow0 * PIECES;
        x2_store_m(op, nq, krow0, v, sc, k < kend ? 1.0f : 0.0f, k + 1 < kend ? 1.0f : 0.0f);
    }
    __device__ __forceinline__ void init(const GemmArgs& g, int n0, int tid) {
        mn_lane_map(tid, nq, krow0);
        const int n = n0 + nq * 4;
        base = g.B + min(n, g.N - 4);           // branch-free: columns >= N re-read the last quad (never stored)
    }
    template <bool RAW = false>
    __device__ __forceinline__ void load(const GemmArgs& g, int k0, int kend, float4 (&v)[PIECES]) const {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int k = k0 + krow0 * PIECES + i;
            const float4 t = *reinterpret_cast<const float4*>(base + (size_t)min(k, g.K - 1) * g.ldb);
            const bool in = RAW || k < kend;
            v[i] = make_float4(in ? t.x : 0.f, in ? t.y : 0.f, in ? t.z : 0.f, in ? t.w : 0.f);
        }
    }
    __device__ __forceinline__ void store(float* bs, const float4 (&v)[PIECES]) const {
#pragma unroll
        for (int i = 0; i < PIECES; ++i)
            *reinterpret_cast<float4*>(bs + (krow0 * PIECES + i) * LDS_B + nq * 4) = v[i];
    }
    __device__ __forceinline__ void store3(char* op, const float4 (&v)[PIECES]) const { x3_store_m(op, nq, krow0, v); }
};

struct BLoaderK {   // element (k, n) at B[n*ldb + k]  (nn.Linear weight [N, K])
    static constexpr int PIECES = BN * BK / 4 / NT;
    const float* ptr[PIECES];
    int row[PIECES];
    int q;
    float sc[PIECES];
    __device__ __forceinline__ void init_scale(const GemmArgs& g, int n0) {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) sc[i] = x2_scale(g.b_amax[min(n0 + row[i], g.N - 1)]);
    }
    __device__ __forceinline__ void store2(char* op, const float4 (&v)[PIECES], int k0, int kend) const {
        const float m = (k0 + q * 4 < kend) ? 1.0f : 0.0f;
#pragma unroll
        for (int i = 0; i < PIECES; ++i) x2_store_k(op, row[i], q, v[i], sc[i] * m);
    }
    __device__ __forceinline__ void init(const GemmArgs& g, int n0, int tid) {
        q = tid & 3;
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            row[i] = (tid >> 2) + i * (NT / 4);
            const int n = n0 + row[i];
            ptr[i] = g.B + (size_t)min(n, g.N - 1) * g.ldb;     // branch-free: rows >= N re-read row N-1
        }
    }
    template <bool RAW = false>
    __device__ __forceinline__ void load(const GemmArgs& g, int k0, int kend, float4 (&v)[PIECES]) const {
        const int k = k0 + q * 4;
        const int kc = min(k, g.K - 4);
        const bool in = RAW || k < kend;
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const float4 t = *reinterpret_cast<const float4*>(ptr[i] + kc);
            v[i] = make_float4(in ? t.x : 0.f, in ? t.y : 0.f, in ? t.z : 0.f, in ? t.w : 0.f);
        }
    }
    __device__ __forceinline__ void store(float* bs, const float4 (&v)[PIECES]) const {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            float* d = bs + (q * 4) * LDS_B + row[i];
            d[0 * LDS_B] = v[i].x;
            d[1 * LDS_B] = v[i].y;
            d[2 * LDS_B] = v[i].z;
            d[3 * LDS_B] = v[i].w;
        }
    }
    __device__ __forceinline__ void store3(char* op, const float4 (&v)[PIECES]) const {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) x3_store_k(op, row[i], q, v[i]);
    }
};

template <int AL> struct ASel { using type = ALoaderK<AL>; };
template <> struct ASel<A_MCONTIG> { using type = ALoaderM<A_MCONTIG>; };
template <> struct ASel<A_CONV_M> { using type = ALoaderM<A_CONV_M>; };
template <int BL> struct BSel { using type = BLoaderN; };
template <> struct BSel<B_KCONTIG> { using type = BLoaderK; };

template <int AL, bool RAW = false, class L>
__device__ __forceinline__ void a_load(const L& l, const GemmArgs& g, int k0, int kend, float4 (&v)[L::PIECES]) {
    if constexpr (AL == A_KCONTIG || AL == A_CONV_K) l.template load<RAW>(g, k0, v);
    else l.template load<RAW>(g, k0, kend, v);
}

// grid = (tiles_n, tiles_m, splits); block = 512
// XCD-aware work order (speed only).  Workgroups are dealt to the 8 XCDs round robin in dispatch order (x fastest, then y,
// then z), and every XCD has its own 4 MB L2: an operand panel that several tiles share is fetched through the fabric once
// per XCD that runs one of them.  The calibrated FETCH_SIZE of the Winograd-forward launch (121 positions x 4 x 4 tiles)
// was 6.5 GB for 2.0 GB of operands: with each position's 16 tiles spread over all 8 XCDs its A panels crossed the fabric
// twice and its B panels four times.
//   * z >= 8 slices (batch entries = Winograd positions, or K-splits): each XCD takes WHOLE slices - slice z = 8 j + xcd -
//     so all tiles of a slice share ONE L2 and run at the same time (the first T workgroups an XCD receives are the T
//     tiles of its first slice); the z % 8 left-over slices are spread as before.
//   * otherwise each XCD gets a contiguous run of tiles of the slice.
//   * tile order inside a run: the index that walks the SMALLER operand's panels runs fastest, so the tiles that share a
//     panel of the bigger operand are neighbours (N > M: B = [K,N] is the bigger one -> m fastest).
#ifndef CIM_GEMM_XCD_MODE
#define CIM_GEMM_XCD_MODE 1          // 0 = round 1's order (contiguous runs inside a slice, n fastest)
#endif
__device__ __forceinline__ void xcd_tile_map(const GemmArgs& g, int& tile_m, int& tile_n, int& z) {
    const int tn = gridDim.x, tm = gridDim.y, T = tn * tm, Z = gridDim.z;
    tile_m = blockIdx.y;
    tile_n = blockIdx.x;
    z = blockIdx.z;
#ifdef CIM_GEMM_NO_XCD
    return;
#endif
    int b = blockIdx.y * tn + blockIdx.x;                       // index inside the slice
    bool remap_in_slice = true;
#if CIM_GEMM_XCD_MODE == 1
    const bool m_fast = g.N > g.M;
    if (Z >= 8) {
        const long long L = (long long)blockIdx.z * T + b;
        const int zfull = Z & ~7;
        if (L < (long long)zfull * T) {
            const int xcd = (int)(L & 7);
            const long long s = L >> 3;
            z = (int)(s / T) * 8 + xcd;
            b = (int)(s % T);
            remap_in_slice = false;
        }
    }
#else
    const bool m_fast = false;
#endif
    int t = b;
    if (remap_in_slice) {
        const int q = T >> 3, r = T & 7, xcd = b & 7, i = b >> 3;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;   // bijective for any T
    }
    if (m_fast) {
        tile_n = t / tm;
        tile_m = t - tile_n * tm;
    } else {
        tile_m = t / tn;
        tile_n = t - tile_m * tn;
    }
}

template <int AL, int BL>
__global__ __launch_bounds__(NT, CIM_GEMM_MINW) void gemm_f32_kernel(const GemmArgs g_in) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    int tile_m, tile_n, zidx;
    xcd_tile_map(g_in, tile_m, tile_n, zidx);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    GemmArgs gb = g_in;
    int zsplit = zidx;
    if (gb.batch > 1) {          // batched: one independent GEMM per z (Winograd positions)
        gb.A += (size_t)zidx * gb.a_bs;
        gb.B += (size_t)zidx * gb.b_bs;
        gb.C += (size_t)zidx * gb.c_bs;
        zsplit = 0;
    }
    const GemmArgs& g = gb;
    const int kbeg = zsplit * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);

    typename ASel<AL>::type la;
    typename BSel<BL>::type lb;
    la.init(g, m0, tid);
    lb.init(g, n0, tid);

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // Three LDS slabs in rotation.  Per slab t: first half of the k-steps, then the registers holding
    // slab t+1 (loaded from HBM/L2 during slab t-1) are written to LDS buffer (t+1)%3 and the loads
    // of slab t+2 are issued, ONE barrier, second half of the k-steps.  The LDS write, the barrier
    // skew and the next slab's first operand reads are therefore covered by MFMAs still in flight;
    // a buffer is rewritten only two barriers after its last read.
    float4 ra[ASel<AL>::type::PIECES], rb[BSel<BL>::type::PIECES];
    const int nslab = (kend - kbeg + BK - 1) / BK;
    a_load<AL>(la, g, kbeg, kend, ra);
    lb.load(g, kbeg, kend, rb);
    la.store(smem, ra);
    lb.store(smem + BK * LDS_A, rb);
    if (nslab > 1) {
        a_load<AL>(la, g, kbeg + BK, kend, ra);
        lb.load(g, kbeg + BK, kend, rb);
    }
    __syncthreads();

    const int lk = lane >> 5, l31 = lane & 31;
    const int a_off = wm * WM + l31 + lk * LDS_A;            // this lane's A fragment column, k-parity row
    const int b_off = BK * LDS_A + wn * WN + l31 + lk * LDS_B;
    // MFMA operand fragments are double-buffered in registers: the ds_reads of k-step s+1 (for the
    // last step: of the NEXT slab's first step, whose buffer is complete since the mid-slab barrier)
    // are issued before the MFMAs of step s, so LDS latency never sits between two MFMA groups.
    float af[2][MI], bf[2][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) af[0][i] = smem[a_off + i * 32];
#pragma unroll
    for (int j = 0; j < NI; ++j) bf[0][j] = smem[b_off + j * 32];
    int ic = 0;                                   // LDS buffer of the current slab (t % 3)
    constexpr int STEPS = BK / 2;
    static_assert(STEPS % 2 == 0, "fragment double-buffer parity assumes an even number of k-steps per slab");
    for (int t = 0; t < nslab; ++t) {
        const int in_ = (ic == 2) ? 0 : ic + 1;
        const float* cur = smem + ic * SLAB;
        float* nxt = smem + in_ * SLAB;
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            if (st == STEPS / 2) {
#if CIM_GEMM_EXP != 1 && CIM_GEMM_EXP != 3
                if (t + 1 < nslab) {
                    la.store(nxt, ra);
                    lb.store(nxt + BK * LDS_A, rb);
                }
#endif
#if CIM_GEMM_EXP != 1 && CIM_GEMM_EXP != 2
                if (t + 2 < nslab) {
                    a_load<AL>(la, g, kbeg + (t + 2) * BK, kend, ra);
                    lb.load(g, kbeg + (t + 2) * BK, kend, rb);
                }
#endif
#if CIM_GEMM_EXP != 4
                __syncthreads();
#endif
            }
            const int pc = st & 1, pn = pc ^ 1;
            if (st + 1 < STEPS) {
#pragma unroll
                for (int i = 0; i < MI; ++i) af[pn][i] = cur[a_off + (st + 1) * 2 * LDS_A + i * 32];
#pragma unroll
                for (int j = 0; j < NI; ++j) bf[pn][j] = cur[b_off + (st + 1) * 2 * LDS_B + j * 32];
            } else if (t + 1 < nslab) {
#pragma unroll
                for (int i = 0; i < MI; ++i) af[pn][i] = nxt[a_off + i * 32];
#pragma unroll
                for (int j = 0; j < NI; ++j) bf[pn][j] = nxt[b_off + j * 32];
            }
            __builtin_amdgcn_sched_barrier(0);     // keep the prefetch ABOVE this step's MFMAs (hipcc would sink it)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[pc][i], bf[pc][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        ic = in_;
    }

    // epilogue: C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    float* C = g.C + (size_t)zsplit * g.c_split_stride;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int n = n0 + wn * WN + j * 32 + l31;
        if (n >= g.N) continue;
        const float bv = (g.bias != nullptr) ? g.bias[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (m >= g.M) continue;
                float v = acc[i][j][r] + bv;
                if (g.relu) v = fmaxf(v, 0.0f);
                C[(size_t)m * g.ldc + n] = v;
            }
        }
    }
}


using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

// bf16x3 engine.  Same tiling, loaders, split-K / batched modes and epilogue as gemm_f32_kernel; a slab is
// one 16-k MFMA step (8 output tiles x 6 products = 48 MFMAs per wave).  Two LDS buffers: slab t+1 is
// split and written (and slab t+2's global loads issued) between the small-term and the large-term
// products of slab t, one barrier per slab.
template <int AL, int BL>
__global__ __launch_bounds__(NT, 2) void gemm_bf16x3_kernel(const GemmArgs g_in) {
    extern __shared__ __attribute__((aligned(16))) char smem3[];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    int tile_m, tile_n, zidx;
    xcd_tile_map(g_in, tile_m, tile_n, zidx);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    GemmArgs gb = g_in;
    int zsplit = zidx;
    if (gb.batch > 1) {
        gb.A += (size_t)zidx * gb.a_bs;
        gb.B += (size_t)zidx * gb.b_bs;
        gb.C += (size_t)zidx * gb.c_bs;
        zsplit = 0;
    }
    const GemmArgs& g = gb;
    const int kbeg = zsplit * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);

    typename ASel<AL>::type la;
    typename BSel<BL>::type lb;
#if CIM_X3_EXP == 8
    la.init(g, 0, tid);      // ablation: every tile streams the same (cache-resident) operand panels
    lb.init(g, 0, tid);
#else
    la.init(g, m0, tid);
    lb.init(g, n0, tid);
#endif

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    float4 ra[ASel<AL>::type::PIECES], rb[BSel<BL>::type::PIECES];
    const int nslab = (kend - kbeg + BK - 1) / BK;
    a_load<AL>(la, g, kbeg, kend, ra);
    lb.load(g, kbeg, kend, rb);
    la.store3(smem3, ra);
    lb.store3(smem3 + X3_OPER, rb);
#pragma unroll 1
    for (int pre = 1; pre <= 2; ++pre) {     // slab 1 -> LDS buffer 1, slab 2 -> registers
        const int kn = kbeg + min(pre, nslab - 1) * BK;
        a_load<AL>(la, g, kn, kend, ra);
        lb.load(g, kn, kend, rb);
        if (pre == 1) {
            la.store3(smem3 + X3_SLAB, ra);
            lb.store3(smem3 + X3_SLAB + X3_OPER, rb);
        }
    }
    __syncthreads();

    const int lk = lane >> 5, l31 = lane & 31;
    int a_off[MI], b_off[NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) a_off[i] = lk * X3_KG + x3_slot(wm * WM + i * 32 + l31) * 16;
#pragma unroll
    for (int j = 0; j < NI; ++j) b_off[j] = X3_OPER + lk * X3_KG + x3_slot(wn * WN + j * 32 + l31) * 16;

#define X3_FRAG(base, off, plane) (*reinterpret_cast<const bf16x8*>((base) + (off) + (plane) * X3_PLANE))
#define X3_MMA(AF, BF)                                                                                   \
    _Pragma("unroll") for (int i = 0; i < MI; ++i) _Pragma("unroll") for (int j = 0; j < NI; ++j)        \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF[i], BF[j], acc[i][j], 0, 0, 0)

    // Three LDS buffers, one barrier per slab.  During slab t a wave (1) multiplies the fragments of slab t,
    // (2) splits + writes slab t+2 (its fp32 values were loaded from HBM/L2 during slab t-1) and issues the
    // global loads of slab t+3, (3) prefetches the first fragments of slab t+1 (complete since the barrier
    // that ended slab t-1) - so neither the barrier nor LDS latency sits in front of the next slab's MFMAs.
    // Buffer (t+2)%3 was last read in slab t-1.  sched_barrier fences keep each third of the MFMAs with
    // its share of the VALU / LDS work (hipcc otherwise bunches the conversions).
    bf16x8 ah[MI], ax[MI], bh[NI], bx[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) bh[j] = X3_FRAG(smem3, b_off[j], 0);
#pragma unroll
    for (int i = 0; i < MI; ++i) ax[i] = X3_FRAG(smem3, a_off[i], 2);
    int ic = 0;
    for (int t = 0; t < nslab; ++t) {
        const int i1 = (ic == 2) ? 0 : ic + 1, i2 = (i1 == 2) ? 0 : i1 + 1;
        const char* cur = smem3 + ic * X3_SLAB;
        const char* nx1 = smem3 + i1 * X3_SLAB;
        char* nx2 = smem3 + i2 * X3_SLAB;
        // ---- third 1: l*h, h*l; fetch the remaining fragments of this slab
#pragma unroll
        for (int i = 0; i < MI; ++i) ah[i] = X3_FRAG(cur, a_off[i], 0);
#pragma unroll
        for (int j = 0; j < NI; ++j) bx[j] = X3_FRAG(cur, b_off[j], 2);
        X3_MMA(ax, bh);
#pragma unroll
        for (int i = 0; i < MI; ++i) ax[i] = X3_FRAG(cur, a_off[i], 1);
        X3_MMA(ah, bx);
#pragma unroll
        for (int j = 0; j < NI; ++j) bx[j] = X3_FRAG(cur, b_off[j], 1);
        __builtin_amdgcn_sched_barrier(0);
        // ---- third 2: m*m, h*m; A of slab t+2 -> LDS, A of slab t+3 <- global
        X3_MMA(ax, bx);
#if CIM_X3_EXP != 2 && CIM_X3_EXP != 5
        la.store3(nx2, ra);
#endif
        const int kn = kbeg + min(t + 3, nslab - 1) * BK;
#if CIM_X3_EXP != 3 && CIM_X3_EXP != 5
        a_load<AL>(la, g, kn, kend, ra);
#endif
        X3_MMA(ah, bx);
        __builtin_amdgcn_sched_barrier(0);
        // ---- third 3: m*h, h*h; B likewise; prefetch l(A), h(B) of slab t+1
        X3_MMA(ax, bh);
#if CIM_X3_EXP != 2 && CIM_X3_EXP != 5
        lb.store3(nx2 + X3_OPER, rb);
#endif
#if CIM_X3_EXP != 3 && CIM_X3_EXP != 5
        lb.load(g, kn, kend, rb);
#endif
#pragma unroll
        for (int i = 0; i < MI; ++i) ax[i] = X3_FRAG(nx1, a_off[i], 2);
        X3_MMA(ah, bh);
#pragma unroll
        for (int j = 0; j < NI; ++j) bh[j] = X3_FRAG(nx1, b_off[j], 0);
        __builtin_amdgcn_sched_barrier(0);
#if CIM_X3_EXP != 4 && CIM_X3_EXP != 5
        __syncthreads();
#endif
        ic = i1;
    }
#undef X3_FRAG
#undef X3_MMA

    float* C = g.C + (size_t)zsplit * g.c_split_stride;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int n = n0 + wn * WN + j * 32 + l31;
        if (n >= g.N) continue;
        const float bv = (g.bias != nullptr) ? g.bias[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (m >= g.M) continue;
                float v = acc[i][j][r] + bv;
                if (g.relu) v = fmaxf(v, 0.0f);
                C[(size_t)m * g.ldc + n] = v;
            }
        }
    }
}

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

// f16x2 engine.  Same tiling, loaders, split-K / batched modes as gemm_bf16x3_kernel; a slab is one 16-k
// MFMA step (8 output tiles x 3 products = 24 MFMAs per wave), three LDS buffers, one barrier per slab.
template <int AL, int BL>
__global__ __launch_bounds__(NT, 2) void gemm_f16x2_kernel(const GemmArgs g_in) {
    extern __shared__ __attribute__((aligned(16))) char smem2[];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    int tile_m, tile_n, zidx;
    xcd_tile_map(g_in, tile_m, tile_n, zidx);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    GemmArgs gb = g_in;
    int zsplit = zidx;
    if (gb.batch > 1) {
        gb.A += (size_t)zidx * gb.a_bs;
        gb.B += (size_t)zidx * gb.b_bs;
        gb.C += (size_t)zidx * gb.c_bs;
        gb.a_amax += (size_t)zidx * gb.M;
        gb.b_amax += (size_t)zidx * gb.N;
        zsplit = 0;
    }
    const GemmArgs& g = gb;
    const int kbeg = zsplit * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);

    typename ASel<AL>::type la;
    typename BSel<BL>::type lb;
    la.init(g, m0, tid);
    lb.init(g, n0, tid);
    la.init_scale(g, m0);
    lb.init_scale(g, n0);

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // Global loads run TWO slabs ahead of their LDS store (two register sets, the slab loop is unrolled by
    // two): a slab lasts ~2 us and the L2/HBM latency under load is of that order - with one set the split +
    // store of slab t+2 waited on loads issued only one slab earlier (measured: 282 -> 392 TF with the loads
    // removed, 422 with the stores removed, i.e. a latency chain, not a throughput limit).
    float4 ra0[ASel<AL>::type::PIECES], rb0[BSel<BL>::type::PIECES];
    float4 ra1[ASel<AL>::type::PIECES], rb1[BSel<BL>::type::PIECES];
    const int nslab = (kend - kbeg + BK - 1) / BK;
    a_load<AL, true>(la, g, kbeg, kend, ra0);
    lb.template load<true>(g, kbeg, kend, rb0);
    a_load<AL, true>(la, g, kbeg + min(1, nslab - 1) * BK, kend, ra1);
    lb.template load<true>(g, kbeg + min(1, nslab - 1) * BK, kend, rb1);
    la.store2(smem2, ra0, kbeg, kend);
    lb.store2(smem2 + X2_OPER, rb0, kbeg, kend);
    a_load<AL, true>(la, g, kbeg + min(2, nslab - 1) * BK, kend, ra0);       // slab 2 -> set 0
    lb.template load<true>(g, kbeg + min(2, nslab - 1) * BK, kend, rb0);
    la.store2(smem2 + X2_SLAB, ra1, kbeg + BK, kend);
    lb.store2(smem2 + X2_SLAB + X2_OPER, rb1, kbeg + BK, kend);
    a_load<AL, true>(la, g, kbeg + min(3, nslab - 1) * BK, kend, ra1);       // slab 3 -> set 1
    lb.template load<true>(g, kbeg + min(3, nslab - 1) * BK, kend, rb1);
    __syncthreads();

    const int lk = lane >> 5, l31 = lane & 31;
    int a_off[MI], b_off[NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) a_off[i] = lk * X2_KG + x3_slot(wm * WM + i * 32 + l31) * 16;
#pragma unroll
    for (int j = 0; j < NI; ++j) b_off[j] = X2_OPER + lk * X2_KG + x3_slot(wn * WN + j * 32 + l31) * 16;

#if CIM_X2_EXP == 7    /* ablation: MFMA only (fragments stay in registers) */
#define X2_FRAG(base, off, plane) (al[0])
#else
#define X2_FRAG(base, off, plane) (*reinterpret_cast<const f16x8*>((base) + (off) + (plane) * X2_PLANE))
#endif
    // (tried: skipping the MFMAs of a wave's 32-row sub-tiles that lie past M in the ragged last M-tile - 800 ... 1200
    // proposals against 256-row tiles - behind a wave-uniform branch: the branches break the MFMA / VALU interleave,
    // 16.8 -> 17.8 ms per step)
#define X2_MMA(AF, BF)                                                                                   \
    _Pragma("unroll") for (int i = 0; i < MI; ++i) _Pragma("unroll") for (int j = 0; j < NI; ++j)        \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AF[i], BF[j], acc[i][j], 0, 0, 0)

    // During slab t a wave (1) multiplies slab t (the two cross terms first, then h*h), (2) splits + writes
    // slab t+2 (register set t%2, loaded during slab t-2) and issues the global loads of slab t+4 into the
    // same set, (3) prefetches l(A) and h(B) of slab t+1 (complete since the barrier that ended slab t-1).
    // LDS buffer (t+2)%3 was last read in slab t-1.
    f16x8 ah[MI], al[MI], bh[NI], bl[NI];
#if CIM_X2_EXP == 7
    al[0] = *reinterpret_cast<const f16x8*>(smem2 + a_off[0]);
#endif
#pragma unroll
    for (int j = 0; j < NI; ++j) bh[j] = X2_FRAG(smem2, b_off[j], 0);
#pragma unroll
    for (int i = 0; i < MI; ++i) al[i] = X2_FRAG(smem2, a_off[i], 1);
    int ic = 0;
#if CIM_X2_EXP == 2 || CIM_X2_EXP == 5 || CIM_X2_EXP == 7
#define X2_STORE_A(RA)
#define X2_STORE_B(RB)
#else
#define X2_STORE_A(RA) la.store2(nx2, RA, ks, kend)
#define X2_STORE_B(RB) lb.store2(nx2 + X2_OPER, RB, ks, kend)
#endif
#if CIM_X2_EXP == 3 || CIM_X2_EXP == 5 || CIM_X2_EXP == 7
#define X2_LOAD_A(RA)
#define X2_LOAD_B(RB)
#else
#define X2_LOAD_A(RA) a_load<AL, true>(la, g, kn, kend, RA)
#define X2_LOAD_B(RB) lb.template load<true>(g, kn, kend, RB)
#endif
#if CIM_X2_EXP == 4 || CIM_X2_EXP == 5 || CIM_X2_EXP == 7
#define X2_SYNC()
#else
#define X2_SYNC() __syncthreads()
#endif
#define X2_SLAB_BODY(T, RA, RB)                                                                          \
    {                                                                                                    \
        const int i1 = (ic == 2) ? 0 : ic + 1, i2 = (i1 == 2) ? 0 : i1 + 1;                              \
        const char* cur = smem2 + ic * X2_SLAB;                                                          \
        const char* nx1 = smem2 + i1 * X2_SLAB;                                                          \
        char* nx2 = smem2 + i2 * X2_SLAB;                                                                \
        /* third 1: l*h; fetch h(A), l(B) of this slab */                                                \
        _Pragma("unroll") for (int i = 0; i < MI; ++i) ah[i] = X2_FRAG(cur, a_off[i], 0);                \
        _Pragma("unroll") for (int j = 0; j < NI; ++j) bl[j] = X2_FRAG(cur, b_off[j], 1);                \
        X2_MMA(al, bh);                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        /* third 2: h*l; A of slab t+2 -> LDS, A of slab t+4 <- global */                                \
        const int kn = kbeg + min((T) + 4, nslab - 1) * BK;                                              \
        const int ks = kbeg + ((T) + 2) * BK;      /* first k of the slab held in RA / RB */             \
        X2_STORE_A(RA);                                                                                  \
        X2_LOAD_A(RA);                                                                                   \
        X2_MMA(ah, bl);                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        /* third 3: h*h; B likewise; prefetch l(A), h(B) of slab t+1 */                                  \
        X2_STORE_B(RB);                                                                                  \
        X2_LOAD_B(RB);                                                                                   \
        _Pragma("unroll") for (int i = 0; i < MI; ++i) al[i] = X2_FRAG(nx1, a_off[i], 1);                \
        X2_MMA(ah, bh);                                                                                  \
        _Pragma("unroll") for (int j = 0; j < NI; ++j) bh[j] = X2_FRAG(nx1, b_off[j], 0);                \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        X2_SYNC();                                                                                       \
        ic = i1;                                                                                         \
    }
    int t = 0;
    for (; t + 1 < nslab; t += 2) {
        X2_SLAB_BODY(t, ra0, rb0)
        X2_SLAB_BODY(t + 1, ra1, rb1)
    }
    if (t < nslab) X2_SLAB_BODY(t, ra0, rb0)
#undef X2_SLAB_BODY
#undef X2_STORE_A
#undef X2_STORE_B
#undef X2_LOAD_A
#undef X2_LOAD_B
#undef X2_SYNC
#undef X2_FRAG
#undef X2_MMA

    // epilogue: undo the operand scales (powers of two: exact), bias, ReLU.  The 256 row factors are staged through
    // LDS (free after the last slab): fetched per accumulator register they were 128 dependent global loads per lane.
    float* s_isa = reinterpret_cast<float*>(smem2);
    __syncthreads();
    if (tid < BM) s_isa[tid] = x2_inv(x2_scale(g.a_amax[min(m0 + tid, g.M - 1)]));
    __syncthreads();
    float* C = g.C + (size_t)zsplit * g.c_split_stride;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int n = n0 + wn * WN + j * 32 + l31;
        if (n >= g.N) continue;
        const float bv = (g.bias != nullptr) ? g.bias[n] : 0.0f;
        const float isb = x2_inv(x2_scale(g.b_amax[n]));
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                const int m = m0 + ml;
                if (m >= g.M) continue;
                float v = acc[i][j][r] * s_isa[ml] * isb + bv;
                if (g.relu) v = fmaxf(v, 0.0f);
#if CIM_X2_NT
                if (g.c_split_stride == 0) __builtin_nontemporal_store(v, &C[(size_t)m * g.ldc + n]);      // final output: streamed
                else C[(size_t)m * g.ldc + n] = v;                                                          // split-K partial: re-read at once
#else
                C[(size_t)m * g.ldc + n] = v;
#endif
            }
        }
    }
}

// |max| bit patterns per row and per column of X [batch][rows][ld] (cols used), accumulated with atomicMax
// into caller-zeroed arrays; either output may be null.  A lane owns 4 consecutive columns: column maxima stay in
// registers, row maxima are wave-reduced per row.  The kernel is a pure HBM stream.
// wave-wide maximum on the VALU (v_max_u32_dpp: row_shr 1,2,4,8 leave each 16-lane row's maximum in its lane 15, then
// row_bcast:15 / row_bcast:31 carry it up): valid in LANE 63.
__device__ __forceinline__ unsigned amax_wave(unsigned v) {
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true));
    return v;
}
// rows_per_wg rows (a multiple of AMAX_UNROLL) x 1024 columns per workgroup; AMAX_UNROLL rows (16 B loads) in flight per lane
constexpr int AMAX_UNROLL = 4;
__global__ __launch_bounds__(256) void amax_rowcol_kernel(const float* __restrict__ X, int rows, int cols, int ld,
                                                          long long bs, unsigned* __restrict__ row_amax,
                                                          unsigned* __restrict__ col_amax, int rows_per_wg) {
    X += (size_t)blockIdx.z * bs;
    const int c = blockIdx.x * 1024 + threadIdx.x * 4;
    const int r0 = blockIdx.y * rows_per_wg, r1 = min(rows, r0 + rows_per_wg);
    const bool cin = c < cols;
    const float* p = X + (size_t)r0 * ld + (cin ? c : 0);
    unsigned* ra = row_amax ? row_amax + (size_t)blockIdx.z * rows : nullptr;
    const bool lead = (threadIdx.x & 63) == 63;
    uint4 cm = make_uint4(0, 0, 0, 0);
    int r = r0;
    for (; r + AMAX_UNROLL <= r1; r += AMAX_UNROLL, p += AMAX_UNROLL * (size_t)ld) {
        uint4 v[AMAX_UNROLL];
#pragma unroll
        for (int i = 0; i < AMAX_UNROLL; ++i) v[i] = *reinterpret_cast<const uint4*>(p + (size_t)i * ld);
#pragma unroll
        for (int i = 0; i < AMAX_UNROLL; ++i) {
            v[i].x &= 0x7fffffffu; v[i].y &= 0x7fffffffu; v[i].z &= 0x7fffffffu; v[i].w &= 0x7fffffffu;
            cm.x = max(cm.x, v[i].x); cm.y = max(cm.y, v[i].y); cm.z = max(cm.z, v[i].z); cm.w = max(cm.w, v[i].w);
        }
        if (ra != nullptr) {
#pragma unroll
            for (int i = 0; i < AMAX_UNROLL; ++i) {
                const unsigned m = amax_wave(cin ? max(max(v[i].x, v[i].y), max(v[i].z, v[i].w)) : 0u);
                if (lead) atomicMax(ra + r + i, m);
            }
        }
    }
    for (; r < r1; ++r, p += ld) {
        uint4 v = *reinterpret_cast<const uint4*>(p);
        v.x &= 0x7fffffffu; v.y &= 0x7fffffffu; v.z &= 0x7fffffffu; v.w &= 0x7fffffffu;
        cm.x = max(cm.x, v.x); cm.y = max(cm.y, v.y); cm.z = max(cm.z, v.z); cm.w = max(cm.w, v.w);
        if (ra != nullptr) {
            const unsigned m = amax_wave(cin ? max(max(v.x, v.y), max(v.z, v.w)) : 0u);
            if (lead) atomicMax(ra + r, m);
        }
    }
    if (col_amax != nullptr && cin) {
        unsigned* d = col_amax + (size_t)blockIdx.z * cols + c;
        atomicMax(d, cm.x); atomicMax(d + 1, cm.y); atomicMax(d + 2, cm.z); atomicMax(d + 3, cm.w);
    }
}

// sum split-K partials (fixed order: deterministic), add bias, optional ReLU.  One float4 per lane.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ C,
                                                            const float* __restrict__ bias, int M, int N, int ldc,
                                                            int splits, long long stride, int relu) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= (long long)M * N) return;
    const int m = (int)(i / N), n = (int)(i % N);
    float4 s = *reinterpret_cast<const float4*>(ws + i);
    for (int k = 1; k < splits; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(ws + k * stride + i);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (bias) {
        const float4 b = *reinterpret_cast<const float4*>(bias + n);
        s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
    }
    if (relu) {
        s.x = fmaxf(s.x, 0.f); s.y = fmaxf(s.y, 0.f); s.z = fmaxf(s.z, 0.f); s.w = fmaxf(s.w, 0.f);
    }
    *reinterpret_cast<float4*>(C + (size_t)m * ldc + n) = s;
}

// engine: 0 = v_mfma_f32_32x32x2_f32;  1 = bf16x3 split on v_mfma_f32_32x32x16_bf16;  2 = f16x2 (operand scales given).
// An ARGUMENT of every entry point: the library keeps no engine state.
template <int AL, int BL>
int launch(GemmArgs g, int splits, float* workspace, hipStream_t st, int engine) {
    const int tm = (g.M + BM - 1) / BM, tn = (g.N + BN - 1) / BN;
    const size_t lds = engine == 2 ? (size_t)3 * X2_SLAB : engine == 1 ? (size_t)3 * X3_SLAB : sizeof(float) * 3 * SLAB;
    auto kern = engine == 1 ? gemm_bf16x3_kernel<AL, BL> : gemm_f32_kernel<AL, BL>;
    if constexpr (AL == A_KCONTIG || AL == A_MCONTIG) {
        if (engine == 2) kern = gemm_f16x2_kernel<AL, BL>;
    } else if (engine == 2) {
        return -1;
    }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    const int slabs = (g.K + BK - 1) / BK;
    if (splits < 1) splits = 1;
    if (splits > slabs) splits = slabs;
    g.k_per_split = ((slabs + splits - 1) / splits) * BK;
    splits = (g.K + g.k_per_split - 1) / g.k_per_split;
    float* out = g.C;
    const float* bias = g.bias;
    const int relu = g.relu, ldc = g.ldc;
    if (splits > 1) {
        g.C = workspace;
        g.ldc = g.N;
        g.c_split_stride = (long long)g.M * g.N;
        g.bias = nullptr;
        g.relu = 0;
    } else {
        g.c_split_stride = 0;
    }
    if (g.batch > 1 && splits > 1) return -1;
    hipLaunchKernelGGL(kern, dim3(tn, tm, g.batch > 1 ? g.batch : splits), dim3(NT), lds, st, g);
    if (splits > 1) {
        const long long quads = ((long long)g.M * g.N + 3) / 4;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, workspace, out,
                           bias, g.M, g.N, ldc, splits, g.c_split_stride, relu);
    }
    return 0;
}

}  // namespace

static int pick_splits(int M, int N, int K, int engine) {
    // RESIDENT 512-thread workgroups fit per CU, so a launch runs in rounds of 256*RESIDENT tiles;
    // split-K fills the last round (e.g. direct conv wgrad: 288 tiles -> 2 rounds at 56 %; x8 -> 9 full
    // rounds) but costs a workspace round trip of (2s+1)*M*N floats.  Pick the split with the least
    // estimated time: flops / (125 TF * round efficiency) + workspace bytes / 4 TB/s.
    const double CUS = 256.0 * RESIDENT;
    const double tiles = (double)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    const int slabs = (K + BK - 1) / BK;
    const double flops = 2.0 * M * (double)N * K;
    const double rate = engine == 2 ? 400e12 : engine == 1 ? 250e12 : 125e12;
    int best = 1;
    double best_t = 1e30;
    for (int s = 1; s <= 16; ++s) {
        if (s > 1 && slabs / s < 16) break;            // keep >= 16 slabs (256 k) per split
        const double units = tiles * s;
        const double rounds = (double)(long long)((units + CUS - 1) / CUS);
        const double eff = units / (rounds * CUS);
        const double t = flops / (rate * eff) + (s > 1 ? (2.0 * s + 1.0) * M * (double)N * 4.0 / 4e12 : 0.0);
        if (t < best_t) { best_t = t; best = s; }
    }
    // (tried: forcing 8 K-splits where 2-7 were picked, so that every XCD takes whole splits in xcd_tile_map: 16.93-17.02
    // vs 16.82-16.87 ms per step - the extra workspace round trip costs more than the saved fabric traffic)
    return best;
}

extern "C" int cim_gemm_f32_splits(int M, int N, int K, int engine) { return pick_splits(M, N, K, engine == 0 ? 0 : 1); }
extern "C" int cim_gemm_f16x2_splits(int M, int N, int K) { return pick_splits(M, N, K, 2); }

extern "C" int cim_gemm_f32(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, int lda,
                            int ldb, int ldc, int a_mcontig, int b_kcontig, int relu, int splits, float* workspace,
                            int engine, void* stream) {
    CIM_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0 && (engine == 0 || engine == 1));
    CIM_CHECK_ARG(N % 4 == 0 && ldc % 4 == 0 && ldc >= N);
    CIM_CHECK_ARG(a_mcontig ? (M % 4 == 0 && lda % 4 == 0 && lda >= M) : (K % 4 == 0 && lda % 4 == 0 && lda >= K));
    CIM_CHECK_ARG(b_kcontig ? (K % 4 == 0 && ldb % 4 == 0 && ldb >= K) : (ldb % 4 == 0 && ldb >= N));
    CIM_CHECK_ARG(splits <= 1 || workspace != nullptr);
    GemmArgs g{A, B, C, bias, M, N, K, lda, ldb, ldc, relu, 0, 0, 0, 0, 1, 0, 0, 0, nullptr, nullptr};
    hipStream_t st = cim::as_stream(stream);
    int rc;
    if (!a_mcontig && !b_kcontig) rc = launch<A_KCONTIG, B_NCONTIG>(g, splits, workspace, st, engine);
    else if (!a_mcontig && b_kcontig) rc = launch<A_KCONTIG, B_KCONTIG>(g, splits, workspace, st, engine);
    else if (a_mcontig && !b_kcontig) rc = launch<A_MCONTIG, B_NCONTIG>(g, splits, workspace, st, engine);
    else rc = launch<A_MCONTIG, B_KCONTIG>(g, splits, workspace, st, engine);
    if (rc) { cim::set_error("cim_gemm_f32: launch setup failed (%d)", rc); return rc; }
    CIM_CHECK_LAUNCH();
    return 0;
}

extern "C" int cim_conv3x3_f32(const float* X, const float* Whwio, const float* bias, float* Y, int R, int P, int Cin,
                               int Cout, int relu, int engine, void* stream) {
    CIM_CHECK_ARG(X && Whwio && Y && R > 0 && P > 0 && Cin > 0 && Cout > 0 && (engine == 0 || engine == 1));
    CIM_CHECK_ARG(Cin % BK == 0 && Cout % 4 == 0);
    GemmArgs g{X, Whwio, Y, bias, R * P * P, Cout, 9 * Cin, Cin, Cout, Cout, relu, 0, 0, P, Cin, 1, 0, 0, 0, nullptr, nullptr};
    int rc = launch<A_CONV_K, B_NCONTIG>(g, 1, nullptr, cim::as_stream(stream), engine);
    if (rc) { cim::set_error("cim_conv3x3_f32: launch setup failed (%d)", rc); return rc; }
    CIM_CHECK_LAUNCH();
    return 0;
}

extern "C" int cim_conv3x3_wgrad_f32(const float* X, const float* dY, float* dWhwio, int R, int P, int Cin, int Cout,
                                     int splits, float* workspace, int engine, void* stream) {
    CIM_CHECK_ARG(X && dY && dWhwio && R > 0 && P > 0 && Cin > 0 && Cout > 0 && (engine == 0 || engine == 1));
    CIM_CHECK_ARG(Cin % 4 == 0 && Cout % 4 == 0);
    CIM_CHECK_ARG(splits <= 1 || workspace != nullptr);
    GemmArgs g{X, dY, dWhwio, nullptr, 9 * Cin, Cout, R * P * P, Cin, Cout, Cout, 0, 0, 0, P, Cin, 1, 0, 0, 0, nullptr, nullptr};
    int rc = launch<A_CONV_M, B_NCONTIG>(g, splits, workspace, cim::as_stream(stream), engine);
    if (rc) { cim::set_error("cim_conv3x3_wgrad_f32: launch setup failed (%d)", rc); return rc; }
    CIM_CHECK_LAUNCH();
    return 0;
}

extern "C" int cim_gemm_f32_batched(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb,
                                    int ldc, int a_mcontig, int b_kcontig, int batch, long long a_bs, long long b_bs,
                                    long long c_bs, int engine, void* stream) {
    CIM_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0 && batch > 0 && batch <= 65535 && (engine == 0 || engine == 1));
    CIM_CHECK_ARG(N % 4 == 0 && ldc % 4 == 0 && ldc >= N && a_bs % 4 == 0 && b_bs % 4 == 0 && c_bs % 4 == 0);
    CIM_CHECK_ARG(a_mcontig ? (M % 4 == 0 && lda % 4 == 0 && lda >= M) : (K % 4 == 0 && lda % 4 == 0 && lda >= K));
    CIM_CHECK_ARG(b_kcontig ? (K % 4 == 0 && ldb % 4 == 0 && ldb >= K) : (ldb % 4 == 0 && ldb >= N));
    GemmArgs g{A, B, C, nullptr, M, N, K, lda, ldb, ldc, 0, 0, 0, 0, 0, batch, a_bs, b_bs, c_bs, nullptr, nullptr};
    hipStream_t st = cim::as_stream(stream);
    int rc;
    if (!a_mcontig && !b_kcontig) rc = launch<A_KCONTIG, B_NCONTIG>(g, 1, nullptr, st, engine);
    else if (!a_mcontig && b_kcontig) rc = launch<A_KCONTIG, B_KCONTIG>(g, 1, nullptr, st, engine);
    else if (a_mcontig && !b_kcontig) rc = launch<A_MCONTIG, B_NCONTIG>(g, 1, nullptr, st, engine);
    else rc = launch<A_MCONTIG, B_KCONTIG>(g, 1, nullptr, st, engine);
    if (rc) { cim::set_error("cim_gemm_f32_batched: launch setup failed (%d)", rc); return rc; }
    CIM_CHECK_LAUNCH();
    return 0;
}

extern "C" int cim_amax_rowcol(const float* X, int rows, int cols, int ld, int batch, long long bs,
                               uint32_t* row_amax, uint32_t* col_amax, void* stream) {
    CIM_CHECK_ARG(X && rows > 0 && cols > 0 && batch > 0 && batch <= 65535);
    CIM_CHECK_ARG(cols % 4 == 0 && ld % 4 == 0 && ld >= cols && bs % 4 == 0);
    CIM_CHECK_ARG(row_amax != nullptr || col_amax != nullptr);
    // 64 rows x 1024 columns per workgroup (measured: smaller row blocks lose more to the column atomics than they
    // gain in parallelism, 8 rows in flight per lane no better than 4)
    const long long strips = (cols + 1023) / 1024;
    const int rpw = 64;
    const dim3 grid((unsigned)strips, (rows + rpw - 1) / rpw, batch);
    CIM_CHECK_ARG(grid.y <= 65535);
    hipLaunchKernelGGL(amax_rowcol_kernel, grid, dim3(256), 0, cim::as_stream(stream), X, rows, cols, ld, bs, row_amax,
                       col_amax, rpw);
    CIM_CHECK_LAUNCH();
    return 0;
}

extern "C" int cim_gemm_f16x2(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, int lda,
                              int ldb, int ldc, int a_mcontig, int b_kcontig, int relu, int splits, float* workspace,
                              const uint32_t* a_amax, const uint32_t* b_amax, void* stream) {
    CIM_CHECK_ARG(A && B && C && a_amax && b_amax && M > 0 && N > 0 && K > 0);
    CIM_CHECK_ARG(N % 4 == 0 && ldc % 4 == 0 && ldc >= N);
    CIM_CHECK_ARG(a_mcontig ? (M % 4 == 0 && lda % 4 == 0 && lda >= M) : (K % 4 == 0 && lda % 4 == 0 && lda >= K));
    CIM_CHECK_ARG(b_kcontig ? (K % 4 == 0 && ldb % 4 == 0 && ldb >= K) : (ldb % 4 == 0 && ldb >= N));
    CIM_CHECK_ARG(splits <= 1 || workspace != nullptr);
    GemmArgs g{A, B, C, bias, M, N, K, lda, ldb, ldc, relu, 0, 0, 0, 0, 1, 0, 0, 0, a_amax, b_amax};
    hipStream_t st = cim::as_stream(stream);
    int rc;
    if (!a_mcontig && !b_kcontig) rc = launch<A_KCONTIG, B_NCONTIG>(g, splits, workspace, st, 2);
    else if (!a_mcontig && b_kcontig) rc = launch<A_KCONTIG, B_KCONTIG>(g, splits, workspace, st, 2);
    else if (a_mcontig && !b_kcontig) rc = launch<A_MCONTIG, B_NCONTIG>(g, splits, workspace, st, 2);
    else rc = launch<A_MCONTIG, B_KCONTIG>(g, splits, workspace, st, 2);
    if (rc) { cim::set_error("cim_gemm_f16x2: launch setup failed (%d)", rc); return rc; }
    CIM_CHECK_LAUNCH();
    return 0;
}

extern "C" int cim_gemm_f16x2_batched(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb,
                                      int ldc, int a_mcontig, int b_kcontig, int batch, long long a_bs, long long b_bs,
                                      long long c_bs, const uint32_t* a_amax, const uint32_t* b_amax, void* stream) {
    CIM_CHECK_ARG(A && B && C && a_amax && b_amax && M > 0 && N > 0 && K > 0 && batch > 0 && batch <= 65535);
    CIM_CHECK_ARG(N % 4 == 0 && ldc % 4 == 0 && ldc >= N && a_bs % 4 == 0 && b_bs % 4 == 0 && c_bs % 4 == 0);
    CIM_CHECK_ARG(a_mcontig ? (M % 4 == 0 && lda % 4 == 0 && lda >= M) : (K % 4 == 0 && lda % 4 == 0 && lda >= K));
    CIM_CHECK_ARG(b_kcontig ? (K % 4 == 0 && ldb % 4 == 0 && ldb >= K) : (ldb % 4 == 0 && ldb >= N));
    GemmArgs g{A, B, C, nullptr, M, N, K, lda, ldb, ldc, 0, 0, 0, 0, 0, batch, a_bs, b_bs, c_bs, a_amax, b_amax};
    hipStream_t st = cim::as_stream(stream);
    int rc;
    if (!a_mcontig && !b_kcontig) rc = launch<A_KCONTIG, B_NCONTIG>(g, 1, nullptr, st, 2);
    else if (!a_mcontig && b_kcontig) rc = launch<A_KCONTIG, B_KCONTIG>(g, 1, nullptr, st, 2);
    else if (a_mcontig && !b_kcontig) rc = launch<A_MCONTIG, B_NCONTIG>(g, 1, nullptr, st, 2);
    else rc = launch<A_MCONTIG, B_KCONTIG>(g, 1, nullptr, st, 2);
    if (rc) { cim::set_error("cim_gemm_f16x2_batched: launch setup failed (%d)", rc); return rc; }
    CIM_CHECK_LAUNCH();
    return 0;
}
