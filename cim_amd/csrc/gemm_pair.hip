// f16x2p engine: the MaskFuse contractions on operands that their PRODUCERS already split.
//
// Replaces the dense contractions of MaskFuse.forward / backward, /root/reference/lib/modeling/resnet50.py:104-110,135-136
// (Conv2d(2C, C, 3, pad = 1) in the Winograd domain, Linear(49C, 4096), Linear(4096, 4096), data and weight gradients).
//
// Arithmetic: the scaled two-term fp16 split of the f16x2 engine (gemm_f32.hip) - x * s = h + l, h = rne_f16(x s),
// l = rne_f16(x s - h), a * b evaluated as l*h + h*l + h*h on v_mfma_f32_32x32x16_f16 with fp32 accumulation - with two
// differences that take every non-MFMA instruction out of the main loop:
//   * ONE power-of-two scale per stored matrix (per batch entry), not per row / column: an operand is then the same
//     bytes for the product that contracts over its columns and for the one that contracts over its rows (V is the A
//     operand of the forward product AND of the weight gradient; a weight is the B operand of the forward product AND
//     of the data gradient), so the kernel that PRODUCES a tensor can write the split image once;
//   * the split image lives in HBM: "pair" layout [row][col / 8][h: 8 x f16 | l: 8 x f16] = 4 bytes per element, the
//     bytes of the fp32 tensor it replaces.  The GEMM brings 16-byte chunks straight into LDS with
//     global_load_lds_dwordx4 (no VGPR round trip, no conversion, no ds_write) and its loop is MFMA + ds_read only.
// Error: every element carries 22 significant bits relative to ITSELF while |x| >= 2^-13 max|X|, and an absolute error of
// 2^-39 max|X| below that (fp16 subnormals; MFMA does not flush them): the bound of an fp32 GEMM relative to
// max|A| max|B| instead of |a_row| |b_col| (tests/test_gpu_gemm.py::test_pair_engine_*).
//
// Tiling: workgroup 256 x 256 x 32, 8 waves as 2 (M) x 4 (N), wave tile 128 x 64 = 4 x 2 MFMA tiles (128 accumulator
// registers), 48 MFMAs per wave and slab.  Two LDS buffers of 64 KB; ONE barrier per slab, placed between the slab's two
// 16-k steps: the loads of slab t+1 were issued one slab earlier, the fragments of the second step are in registers, so
// behind the barrier the wave issues the loads of slab t+2 into the buffer it just finished and multiplies while the first
// fragments of slab t+1 arrive.
// Operand layouts (template parameters):
//   KC  rows of the stored matrix = tile rows, k contiguous   (A: activations [M,K];  B: nn.Linear weight [N,K])
//       LDS [256 rows][128 B], 16-byte chunks XOR-swizzled inside 256-byte double rows -> conflict-free ds_read_b128
//   MC  rows of the stored matrix = k, tile rows contiguous     (A: X^T for weight gradients;  B: [K,N])
//       LDS [32 k][1024 B] chunks XOR-swizzled by k -> ds_read_b64_tr_b16 (hardware 4 x 4 transpose) delivers 4 consecutive
//       k per lane: two reads = one MFMA operand
// The swizzles are applied on the SOURCE address of the LDS-DMA (its destination is lane-linear) and on the reads.
#include "common.h"
#include <type_traits>
#include "../../include/cim_hip.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
typedef short s16x8 __attribute__((__vector_size__(8 * sizeof(short))));

#ifndef CIM_RING_SCHED
#define CIM_RING_SCHED 1
#endif
#ifndef CIM_RING_PRODUCER
#define CIM_RING_PRODUCER 0     // 1: a fifth wave issues every LDS-DMA instruction of the workgroup, the four others hold MFMAs and fragment
#endif                          // reads only.  FASTER ALONE (0.46-0.47 against 0.42-0.43 of the f16 peak: the 256 x 256 kernel's rate) and
                                // SLOWER WHERE IT IS USED: beside the backbone's backward the last phase takes 3.69 ms with it, 3.60 without
                                // (the chains' kernels get less of the CU: profiles/r6/gemm_pair_ring_kernel.txt) - not the product's build
#ifndef CIM_RING_DBG
#define CIM_RING_DBG 0
#endif
#ifndef CIM_PAIR_EXP
#define CIM_PAIR_EXP 0          // ablation switches (tools/bench_gemm_pair.py); 0 = product
#endif

// Two kernels (wave tile 128 x 64 in both, four waves along N):
//   gemm_pair_kernel: 256 x 256 tiles, eight waves = two per SIMD at <= 256 registers, two 64 KB slabs of 32 k in LDS: the workgroup OWNS
//     its CU (every product of the forward / data-gradient chains; `form` = 0 of the entry points);
//   gemm_pair_ring_kernel: 128 x 256 tiles, four waves = ONE per SIMD, a ring of five 24 KB slabs of 16 k (120 KB): half of a CU's
//     registers and 40 KB of its LDS stay free, workgroups of other streams' kernels (the backbone's backward chains: 35 KB, 64 VGPRs)
//     run on the same CU beside it - the form of MaskFuse's late weight gradients (`form` = 1; ops/maskfuse_pair.py).
// (Round 6's other experiment - four waves of 128 x 128, 512 registers per wave - is recorded in profiles/r6/gemm_pair_four_wave_experiment.txt.)
constexpr int BM = 256, BN = 256, BK = 32;
constexpr int NW = 8, NT = 64 * NW;
constexpr int WNC = 4;                              // waves along N
constexpr int WM = 128, WN = BN / WNC, MI = 4, NI = WN / 32;
constexpr int IPW = 32 / NW;                        // LDS-DMA instructions per wave, operand and slab (1 KiB each)
constexpr int OPER = 256 * BK * 4;        // 32768 B per operand and slab
constexpr int SLAB = 2 * OPER;
constexpr int LDS_BYTES = 2 * SLAB;       // 131072
// the ring kernel
constexpr int RBM = 128, RBK = 16, RNT = 256 + 64 * CIM_RING_PRODUCER, RSTAGES = 5;
constexpr int RSTAGE_A = RBM * RBK * 4, RSTAGE_B = BN * RBK * 4, RSTAGE = RSTAGE_A + RSTAGE_B;      // 8 KB + 16 KB
constexpr int RIPA = RSTAGE_A / 1024 / 4, RIPB = RSTAGE_B / 1024 / 4;                                // LDS-DMA instructions per wave and slab: 2 + 4
constexpr int RLDS_BYTES = RSTAGES * RSTAGE;      // 122880

enum { L_KC = 0, L_MC = 1 };

struct PairArgs {
    const char* A;          // pair images
    const char* B;
    float* C;
    const float* bias;      // [N] or null
    int M, N, K;
    int lda, ldb, ldc;      // logical elements per stored row (multiples of 8 for A / B)
    int relu;
    int k_per_split;        // multiple of BK
    long long c_split_stride;
    int batch;
    long long a_bs, b_bs, c_bs;     // elements between batch entries
    const float* a_scale;   // [batch] power-of-two scales the images were written with
    const float* b_scale;
    unsigned* c_amax;       // optional: max |C| bit pattern (atomicMax; caller zeroes), final outputs only
    int tn, tm, tz;         // the product's tile grid (N tiles, M tiles, batch entries / k-splits); a launch covers the tiles
    int tile0;              // tile0 .. tile0 + gridDim.x - 1 of its linear order (all of them unless the launch is chunked)
};

typedef __attribute__((address_space(3))) char* lds_ptr_t;

// Four LDS-DMA loads (global_load_lds_dwordx4: 16 B per lane, destination = M0 + 16 * lane) of one wave into four consecutive
// 1 KiB pieces of LDS starting at byte address lds_dst (wave-uniform), sources sbase (SGPR pair) + 32-bit lane offsets.
// Inline asm on purpose: through __builtin_amdgcn_global_load_lds hipcc treats every later ds_read as a possible reader of
// the DMA's destination and puts s_waitcnt vmcnt(0) in front of it - the loads of slab t+2 would be waited for right
// after their issue.  As an asm statement the DMA is invisible to that bookkeeping; the kernel waits with its own
// s_waitcnt vmcnt(0) in front of the one barrier per slab.  M0 is saved and restored (compiler-reserved); SCC is declared clobbered
// (s_addk_i32 writes it: without the clobber the compiler carried the low / high halves of a 64-bit pointer increment ACROSS the
// statement - s_add_u32 ... asm ... s_addc_u32 - and an operand that crossed a 4 GiB boundary faulted, found in round 6).
__device__ __forceinline__ void glds16x4(const char* sbase, unsigned o0, unsigned o1, unsigned o2, unsigned o3, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %6\n\t"
        "s_nop 4\n\t"
        "global_load_lds_dwordx4 %1, %5\n\t"
        "s_addk_i32 m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, %5\n\t"
        "s_addk_i32 m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %3, %5\n\t"
        "s_addk_i32 m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %4, %5\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(sbase), "s"(lds_dst)
        : "memory", "scc");
}
__device__ __forceinline__ void glds16x2(const char* sbase, unsigned o0, unsigned o1, unsigned lds_dst) {     // (two pieces)
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %4\n\t"
        "s_nop 4\n\t"
        "global_load_lds_dwordx4 %1, %3\n\t"
        "s_addk_i32 m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, %3\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(o0), "v"(o1), "s"(sbase), "s"(lds_dst)
        : "memory", "scc");
}
__device__ __forceinline__ unsigned lds_addr(const char* p) { return (unsigned)(size_t)(lds_ptr_t)p; }

// ---- staging ---------------------------------------------------------------------------------------
// KC: instruction i (0..31) covers tile rows 8i .. 8i+7 = double rows 4i .. 4i+3; a wave issues i = 4w .. 4w+3.
// LDS position (double row d, slot s) holds global (row 2d + (s >> 3), chunk (s & 7) ^ (d & 7)); chunk p = 2 * kgroup + plane.
template <int W, int IPW, int KROWS = BK>
struct StageKC {
    unsigned off[IPW];    // byte offsets of this lane's source chunks relative to (base + k0 * 4)
    __device__ __forceinline__ void init(int row0, int rows, int ld, int wave, int lane) {
#pragma unroll
        for (int ii = 0; ii < IPW; ++ii) {
            const int d = 4 * (wave * IPW + ii) + (lane >> 4);
            const int s = lane & 15;
            const int r = min(row0 + 2 * d + (s >> 3), rows - 1);
            const int p = (s & 7) ^ (d & 7);
            off[ii] = (unsigned)r * (unsigned)ld * 4u + (unsigned)p * 16u;
        }
    }
    __device__ __forceinline__ void issue(const char* base_k, const char* lds_oper, int wave) const {
#pragma unroll
        for (int q = 0; q < IPW; q += 4)
            glds16x4(base_k, off[q], off[q + 1], off[q + 2], off[q + 3], lds_addr(lds_oper) + (wave * IPW + q) * 1024);
    }
    static __device__ __forceinline__ size_t k_step_bytes(int) { return (size_t)BK * 4; }
};
// MC, W = 256 tile columns: instruction i = k row i of the slab (1024 B); lane q loads chunk q ^ swz(k),
// swz(k) = (k & 1) | ((k >> 1) & 1) << 3 (k & 3 = ii for k = 4w + ii).  W = 128 (a k row = 512 B = 32 chunks): instruction i = k rows
// 2i and 2i + 1, lane q loads chunk (q & 31) ^ swz(k) of row k = 2i + (q >> 5) - the same LDS image at a row pitch of 512 B.
template <int W, int IPW, int KROWS = BK>
struct StageMC {
    unsigned off[IPW];
    __device__ __forceinline__ void init(int col0, int cols, int ld, int wave, int lane) {
        const int maxchunk = ((cols - col0) * 4 - 16) / 16;       // last whole chunk of the row that belongs to the matrix
#pragma unroll
        for (int ii = 0; ii < IPW; ++ii) {
            if constexpr (W == 256) {
                const int swz = (ii & 1) | (((ii >> 1) & 1) << 3);    // (k & 3 = ii & 3 for k = IPW w + ii)
                const int c = min(lane ^ swz, maxchunk);
                off[ii] = (unsigned)(wave * IPW + ii) * (unsigned)ld * 4u + (unsigned)col0 * 4u + (unsigned)c * 16u;
            } else {
                const int k = 2 * (wave * IPW + ii) + (lane >> 5);
                const int swz = (lane >> 5) | ((ii & 1) << 3);        // (k & 1, (k >> 1) & 1 = ii & 1: IPW is even)
                const int c = min((lane & 31) ^ swz, maxchunk);
                off[ii] = (unsigned)k * (unsigned)ld * 4u + (unsigned)col0 * 4u + (unsigned)c * 16u;
            }
        }
    }
    __device__ __forceinline__ void issue(const char* base_k, const char* lds_oper, int wave) const {
#if CIM_RING_DBG == 2 || CIM_RING_DBG == 4 || CIM_RING_DBG == 6
        if constexpr (W == 128) return;
#endif
#if CIM_RING_DBG == 3 || CIM_RING_DBG == 4 || CIM_RING_DBG == 6
        if constexpr (KROWS == 16 && W == 256) return;
#endif
        if constexpr (IPW == 2) {
            glds16x2(base_k, off[0], off[1], lds_addr(lds_oper) + (wave * IPW) * 1024);
        } else {
#pragma unroll
            for (int q = 0; q < IPW; q += 4)
                glds16x4(base_k, off[q], off[q + 1], off[q + 2], off[q + 3], lds_addr(lds_oper) + (wave * IPW + q) * 1024);
        }
    }
    // instructions 2h, 2h + 1 of this wave's IPW
    __device__ __forceinline__ void issue_pair(const char* base_k, const char* lds_oper, int wave, int h) const {
#if CIM_RING_DBG == 3 || CIM_RING_DBG == 4 || CIM_RING_DBG == 6
        if constexpr (KROWS == 16 && W == 256) return;
#endif
        glds16x2(base_k, off[2 * h], off[2 * h + 1], lds_addr(lds_oper) + (wave * IPW + 2 * h) * 1024);
    }
    static __device__ __forceinline__ size_t k_step_bytes(int ld) { return (size_t)KROWS * ld * 4; }
};

// ---- fragment reads ----------------------------------------------------------------------------------
// KC: tile row r = wbase + 32 i + (lane & 31), k group kg = 2 ks + (lane >> 5):
//   byte = (r >> 1) * 256 + (((r & 1) << 3 | (2 kg + plane)) ^ ((r >> 1) & 7)) * 16
template <int CNT>
struct FragKC {
    unsigned o[2][2];       // [ks][plane], tile 0
    __device__ __forceinline__ void init(int wbase, int lane) {
        const int l31 = lane & 31, lk = lane >> 5;
        const int x = (l31 >> 1) & 7;
        const unsigned base = (unsigned)((wbase >> 1) + (l31 >> 1)) * 256u + (unsigned)((l31 & 1) << 7);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) o[ks][pl] = base + (unsigned)(((4 * ks + 2 * lk + pl) ^ x) * 16);
    }
    __device__ __forceinline__ f16x8 read(const char* oper, int i, int ks, int pl) const {
        return *reinterpret_cast<const f16x8*>(oper + o[ks][pl] + i * 4096);
    }
};
// MC: lane supplies k row kb + (i16 >> 2) and 4 consecutive tile columns wbase + 32 i + 16 mg + 4 (i16 & 3); two reads
// (k rows +0, +4) make the 8 k values of column wbase + 32 i + (lane & 31).
template <int CNT, int W = 256>
struct FragMC {
    static constexpr int ROW = W * 4;      // bytes per k row of the LDS image
    unsigned o[2][2];       // [i & 1][plane]
    __device__ __forceinline__ void init(int wbase, int lane) {
        const int i16 = lane & 15, mg = (lane >> 4) & 1, lk = lane >> 5;
        const int s = ((i16 >> 2) & 1) | (((i16 >> 3) & 1) << 3);
        const unsigned base = (unsigned)(8 * lk + (i16 >> 2)) * (unsigned)ROW + (unsigned)wbase * 4u + (unsigned)(i16 & 1) * 8u;
#pragma unroll
        for (int ip = 0; ip < 2; ++ip)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
                o[ip][pl] = base + (unsigned)(((8 * ip + 4 * mg + 2 * ((i16 >> 1) & 1) + pl) ^ s) * 16);
    }
    __device__ __forceinline__ f16x8 read(const char* oper, int i, int ks, int pl) const {
        const char* p = oper + o[i & 1][pl] + ks * (16 * ROW) + (i >> 1) * 256;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * ROW));
        const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(f16x8, v);
    }
};

template <int L, int W, int IPW> struct StageSel { using type = StageKC<W, IPW>; };
template <int W, int IPW> struct StageSel<L_MC, W, IPW> { using type = StageMC<W, IPW>; };
template <int L, int CNT, int W> struct FragSel { using type = FragKC<CNT>; };
template <int CNT, int W> struct FragSel<L_MC, CNT, W> { using type = FragMC<CNT, W>; };

// XCD-aware work order (same policy as gemm_f32.hip: xcd_tile_map): >= 8 z slices -> whole slices per XCD; otherwise a
// contiguous run of tiles per XCD; inside a run the index walking the smaller operand's panels runs fastest.
__device__ __forceinline__ void pair_tile_map(int M, int N, int tn, int tm, int Z, int vb, int& tile_m, int& tile_n, int& z) {
    const int T = tn * tm;
    int b = vb % T;                               // (blockIdx.y * tn + blockIdx.x and blockIdx.z of a (tn, tm, Z) grid)
    z = vb / T;
    const int bz = z;
    bool remap_in_slice = true;
    const bool m_fast = N > M;
    if (Z >= 8) {
        const long long L = (long long)bz * T + b;
        const int zfull = Z & ~7;
        if (L < (long long)zfull * T) {
            const int xcd = (int)(L & 7);
            const long long s = L >> 3;
            z = (int)(s / T) * 8 + xcd;
            b = (int)(s % T);
            remap_in_slice = false;
        }
    }
    int t = b;
    if (remap_in_slice) {
        const int q = T >> 3, r = T & 7, xcd = b & 7, i = b >> 3;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
    }
    if (m_fast) {
        tile_n = t / tm;
        tile_m = t - tile_n * tm;
    } else {
        tile_m = t / tn;
        tile_n = t - tile_m * tn;
    }
}

__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, o));
    return v;
}

template <int TBM>
__device__ __forceinline__ void pair_epilogue(const PairArgs& g, f32x16 (&acc)[MI][NI], float (&bvj)[NI], float* Cb, int zsplit, int zb,
                                              int m0, int n0, int wm, int wn, int lane) {
    // epilogue: undo the two scales (powers of two: exact), bias, ReLU.  Nothing may be in flight on the vector-memory counter
    // when the stores start: stores count on vmcnt as well, and a load whose completion the compiler cannot prove at a
    // control-flow join draws an s_waitcnt vmcnt(0) in front of EVERY guarded store (128 per lane, each then waiting for
    // the previous store's acknowledgement).  So the bias is loaded above the loop and consumed here once, and full tiles
    // take a path without per-row guards.  (Tried: each wave passes its 32 x 64 blocks through its slice of the idle LDS and
    // stores rows as 16 bytes per lane - 32 store instructions per wave instead of 128: SLOWER, 1.43 vs 1.26 ms on the
    // Winograd-forward launch; a wave's 4-byte stores already cover two whole 128-byte row segments per instruction.  Also
    // tried: MFMA operands swapped so that the accumulator tile is the transpose and a lane stores four consecutive columns of
    // ITS row as 16 bytes - 32 stores per wave, no LDS: 1.33 vs 1.26 ms (32-byte pieces of 32 different rows per
    // instruction).  The stores themselves cost 4-8 % of a launch (ablation without them: 1.246 vs 1.303 ms).)
    const float inv = 1.0f / (g.a_scale[zb] * g.b_scale[zb]);
#pragma unroll
    for (int j = 0; j < NI; ++j) asm volatile("" : "+v"(bvj[j]));
    float* C = Cb + (size_t)zsplit * g.c_split_stride;
    const int lk = lane >> 5, l31 = lane & 31;
    unsigned amax = 0;
    const bool relu = g.relu != 0;
    if (m0 + TBM <= g.M && n0 + BN <= g.N) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            float* cj = C + (size_t)(m0 + wm * WM + 4 * lk) * g.ldc + n0 + wn * WN + j * 32 + l31;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][j][r] * inv + bvj[j];
                    if (relu) v = fmaxf(v, 0.0f);
                    amax = max(amax, __float_as_uint(v) & 0x7fffffffu);
                    cj[(size_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * g.ldc] = v;
                }
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int n = n0 + wn * WN + j * 32 + l31;
            if (n >= g.N) continue;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                    if (m >= g.M) continue;
                    float v = acc[i][j][r] * inv + bvj[j];
                    if (relu) v = fmaxf(v, 0.0f);
                    amax = max(amax, __float_as_uint(v) & 0x7fffffffu);
                    C[(size_t)m * g.ldc + n] = v;
                }
            }
        }
    }
    if (g.c_amax != nullptr && g.c_split_stride == 0) {
        amax = wave_max_u32(amax);
        if (lane == 0) cim::amax_publish(g.c_amax, amax);
    }
}

// ONEP: the h * h product ALONE (one MFMA product per multiply-add instead of three): operands carry 11 significant bits - fp16
// inputs with fp32 accumulation, the arithmetic class of TF32 (10 bits), which is what the reference's conv / matmul run in on
// its own hardware (torch 1.10 defaults, tools/train.py:153-154 sets only cudnn.deterministic / benchmark).  An explicit
// argument of the entry points (products = 1); the step's default is the fp32-class three-product evaluation.  The l halves of
// the images are still moved (they are interleaved with the h halves in memory) and ignored.
template <int AL, int BL, bool ONEP>
__global__ __launch_bounds__(NT, NW / 4) void gemm_pair_kernel(const PairArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wm = wave / WNC, wn = wave % WNC;
    int tile_m, tile_n, zidx;
    pair_tile_map(g.M, g.N, g.tn, g.tm, g.tz, (int)blockIdx.x + g.tile0, tile_m, tile_n, zidx);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const char* Ab = g.A;
    const char* Bb = g.B;
    float* Cb = g.C;
    int zsplit = zidx, zb = 0;
    if (g.batch > 1) {
        Ab += (size_t)zidx * g.a_bs * 4;
        Bb += (size_t)zidx * g.b_bs * 4;
        Cb += (size_t)zidx * g.c_bs;
        zsplit = 0;
        zb = zidx;
    }
    const int kbeg = zsplit * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);
    const int nslab = (kend - kbeg) / BK;

    using StageA = typename StageSel<AL, BM, IPW>::type;
    using StageB = typename StageSel<BL, BN, IPW>::type;
    StageA sa;
    StageB sb;
    sa.init(m0, g.M, g.lda, wave, lane);
    sb.init(n0, g.N, g.ldb, wave, lane);
    const char* ak = Ab + (AL == L_KC ? (size_t)kbeg * 4 : (size_t)kbeg * g.lda * 4);
    const char* bk = Bb + (BL == L_KC ? (size_t)kbeg * 4 : (size_t)kbeg * g.ldb * 4);
    const size_t a_adv = StageA::k_step_bytes(g.lda), b_adv = StageB::k_step_bytes(g.ldb);

    typename FragSel<AL, MI, BM>::type fa;
    typename FragSel<BL, NI, BN>::type fb;
    fa.init(wm * WM, lane);
    fb.init(wn * WN, lane);

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    float bvj[NI];          // bias of this lane's output columns (consumed in the epilogue)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int n = n0 + wn * WN + j * 32 + (lane & 31);
        bvj[j] = (g.bias != nullptr && n < g.N) ? g.bias[n] : 0.0f;
    }
    // Ragged last M-tile (800 ... 1200 proposals against 256-row tiles): when the tile's second 128 rows lie past M, the
    // four waves that own them (wm = 1) only stage their share of the operands and keep the barrier sequence - the
    // tile then costs its four working waves' MFMA time, about half a tile.  The working waves in turn leave out their 32-row
    // sub-tiles past M (main_loop<MIV> below).
#if CIM_PAIR_EXP != 6
    if (wm == 1 && m0 + WM >= g.M) {
        if constexpr (AL == L_MC) sa.issue(ak, smem, wave);
        sb.issue(bk, smem + OPER, wave);
        ak += a_adv;
        bk += b_adv;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (nslab > 1) {
            if constexpr (AL == L_MC) sa.issue(ak, smem + SLAB, wave);
            sb.issue(bk, smem + SLAB + OPER, wave);
            ak += a_adv;
            bk += b_adv;
        }
        for (int t = 0; t < nslab; ++t) {
            const char* cur = smem + (t & 1) * SLAB;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t + 2 < nslab) {
                if constexpr (AL == L_MC) sa.issue(ak, cur, wave);
                sb.issue(bk, cur + OPER, wave);
                ak += a_adv;
                bk += b_adv;
            }
        }
        return;
    }
#endif
    // prologue: slab 0 -> buffer 0 (waited for), slab 1 -> buffer 1 (in flight), first fragments of slab 0
    sa.issue(ak, smem, wave);
    sb.issue(bk, smem + OPER, wave);
    ak += a_adv;
    bk += b_adv;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (nslab > 1) {
        sa.issue(ak, smem + SLAB, wave);
        sb.issue(bk, smem + SLAB + OPER, wave);
        ak += a_adv;
        bk += b_adv;
    }

    // The main loop, instantiated for MIV = 1 .. 4 valid 32-row sub-tiles of this wave's 128 rows: a working wave of the ragged last
    // M-tile whose rows reach past M leaves out the sub-tiles that lie entirely outside (their fragment reads and MFMAs) - as a
    // compile-time bound, selected once per workgroup.  Run-time `if (i < valid)` guards inside the one loop measured SLOWER (they
    // break the MFMA / LDS-read interleaving of every tile: Winograd forward at 857 rows 1.343 vs 1.221 ms, step 15.05 vs 14.66 ms);
    // the four instantiations: step 14.54 / 14.71 / 14.37 vs 14.63 / 14.76 / 14.50 ms (same box, interleaved), roofline.frac of the
    // mix 0.470 -> 0.480; alone the products gain 2-4 % at 1086 rows, the KC x MC one loses 3.6 % at 857 (tools/bench_gemm_pair.py).
    auto main_loop = [&](auto miv_c) {
    constexpr int MIV = decltype(miv_c)::value;
    f16x8 ah0[MIV], al0[MIV], bh0[NI], bl0[NI];
    f16x8 ah1[MIV], al1[MIV], bh1[NI], bl1[NI];
#define PAIR_READ(AH, AL_, BH, BL_, BUF, KS)                                                   \
    _Pragma("unroll") for (int j = 0; j < NI; ++j) {                                           \
        BH[j] = fb.read((BUF) + OPER, j, KS, 0);                                               \
        if constexpr (!ONEP) BL_[j] = fb.read((BUF) + OPER, j, KS, 1);                         \
    }                                                                                          \
    _Pragma("unroll") for (int i = 0; i < MIV; ++i) {                                          \
        if constexpr (!ONEP) AL_[i] = fa.read((BUF), i, KS, 1);                                \
        AH[i] = fa.read((BUF), i, KS, 0);                                                      \
    }
#define PAIR_MMA(AF, BF)                                                                       \
    _Pragma("unroll") for (int i = 0; i < MIV; ++i) _Pragma("unroll") for (int j = 0; j < NI; ++j) \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AF[i], BF[j], acc[i][j], 0, 0, 0)

#if CIM_PAIR_EXP == 1
#define PAIR_PRIO(x) __builtin_amdgcn_s_setprio(x)
#else
#define PAIR_PRIO(x)
#endif
// scheduling fences around the MFMA clusters: they help the all-K-contiguous instantiation (+1 %) and cost the ones with a
// transposed-read operand 1.5-6 % (tools/bench_gemm_pair.py, CIM_PAIR_EXP=2 removes them everywhere)
#if CIM_PAIR_EXP == 2
#define PAIR_FENCE()
#else
#define PAIR_FENCE() if constexpr (AL == L_KC && BL == L_KC) __builtin_amdgcn_sched_barrier(0)
#endif
// Experiment switch (round 6): the half-step's fragment reads interleaved one to one with its MFMAs, as in the ring kernel, for the
// instantiations with a K-contiguous A.  With two waves per SIMD the other wave already fills those issue slots: two interleaved runs of
// tools/bench_gemm_pair.py gave +2.3 / 0 / +1.6 % and +0.4 / -0.2 / +0.9 % (Winograd forward / data gradient / fc1 forward), the
// M-contiguous instantiations lose 1.7 % - within noise, not the product's build.
#ifndef CIM_PAIR_SCHED
#define CIM_PAIR_SCHED 0
#endif
#define PAIR_SCHED_ON (CIM_PAIR_SCHED && AL == L_KC)
#if CIM_PAIR_SCHED
#define PAIR_NREADS (((AL == L_KC ? 1 : 2) * MIV + (BL == L_KC ? 1 : 2) * NI) * (ONEP ? 1 : 2))
#define PAIR_SCHED()                                                                           \
    if constexpr (PAIR_SCHED_ON) {                                                             \
        _Pragma("unroll") for (int q = 0; q < (ONEP ? 1 : 3) * MIV * NI; ++q) {                 \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                 \
            if (q < PAIR_NREADS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);            \
        }                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                     \
    }
#else
#define PAIR_SCHED()
#endif
#if CIM_PAIR_EXP == 3 || CIM_PAIR_EXP == 5       /* ablation: no LDS-DMA in the loop */
#define PAIR_ISSUE(A, B)
#else
#define PAIR_ISSUE(A, B) { sa.issue(ak, A, wave); sb.issue(bk, B, wave); ak += a_adv; bk += b_adv; }
#endif
#if CIM_PAIR_EXP == 4 || CIM_PAIR_EXP == 5       /* ablation: fragments stay in registers (MFMA only) */
#define PAIR_READ_L(AH, AL_, BH, BL_, BUF, KS) asm volatile("" : "+v"(AH[0]), "+v"(AL_[0]), "+v"(BH[0]), "+v"(BL_[0]));
#else
#define PAIR_READ_L(AH, AL_, BH, BL_, BUF, KS) PAIR_READ(AH, AL_, BH, BL_, BUF, KS)
#endif
    PAIR_READ(ah0, al0, bh0, bl0, smem, 0)
#if CIM_PAIR_EXP == 4 || CIM_PAIR_EXP == 5
    PAIR_READ(ah1, al1, bh1, bl1, smem, 1)
#endif
    for (int t = 0; t < nslab; ++t) {
        const char* cur = smem + (t & 1) * SLAB;
        char* nxt = smem + ((t + 1) & 1) * SLAB;
        // first 16-k step of slab t; its second step's fragments arrive meanwhile
        PAIR_READ_L(ah1, al1, bh1, bl1, cur, 1)
        PAIR_FENCE();
        PAIR_PRIO(1);
        if constexpr (!ONEP) {
            PAIR_MMA(al0, bh0);
            PAIR_MMA(ah0, bl0);
        }
        PAIR_MMA(ah0, bh0);
        PAIR_SCHED();
        PAIR_PRIO(0);
        PAIR_FENCE();
        // every wave holds its fragments of slab t and its share of slab t+1 has landed: slab t+1 is complete and
        // buffer t & 1 is free behind this barrier
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 2 < nslab) PAIR_ISSUE(cur, cur + OPER)
        if (PAIR_SCHED_ON || t + 1 < nslab) {       /* (unconditional with the interleave - reads and MFMAs in ONE basic block; past the
                                                        last slab it reads a buffer nobody writes any more) */
            PAIR_READ_L(ah0, al0, bh0, bl0, nxt, 0)
        }
        PAIR_FENCE();
        PAIR_PRIO(1);
        if constexpr (!ONEP) {
            PAIR_MMA(al1, bh1);
            PAIR_MMA(ah1, bl1);
        }
        PAIR_MMA(ah1, bh1);
        PAIR_SCHED();
        PAIR_PRIO(0);
        PAIR_FENCE();
    }
#undef PAIR_READ_L
#undef PAIR_SCHED
#undef PAIR_ISSUE
#undef PAIR_FENCE
#undef PAIR_PRIO
#undef PAIR_READ
#undef PAIR_MMA
    };
#if CIM_PAIR_EXP == 7
    main_loop(std::integral_constant<int, MI>{});
#else
    switch (__builtin_amdgcn_readfirstlane(min(MI, (g.M - m0 - wm * WM + 31) / 32))) {
        case 1: main_loop(std::integral_constant<int, 1>{}); break;
        case 2: main_loop(std::integral_constant<int, 2>{}); break;
        case 3: main_loop(std::integral_constant<int, 3>{}); break;
        default: main_loop(std::integral_constant<int, MI>{}); break;
    }
#endif


    pair_epilogue<BM>(g, acc, bvj, Cb, zsplit, zb, m0, n0, wm, wn, lane);
}

// The co-resident form (`form` = 1): C tile 128 x 256, four waves (wave tile 128 x 64, one wave per SIMD), both operands K-major (the
// weight gradients' layout).  With one wave per SIMD and half the MFMA work per slab of the kernel above, two 32-k slabs do not cover
// the LDS-DMA's latency (measured with them: 0.33 of the f16 peak against 0.47; 0.48 with the DMA taken out of the loop) - so the
// operands go through a RING of five 16-k slabs (24 KB each): the DMA of slab t + 4 is issued at step t and is waited for at step
// t + 3 (s_waitcnt vmcnt(12): the two younger slabs stay in flight), one barrier per step of four waves.  Per output element the same
// MFMA products in the same order as the kernel above: same bits.
template <bool ONEP>
__global__ __launch_bounds__(RNT, 2) void gemm_pair_ring_kernel(const PairArgs g) {       // (2 waves per SIMD's worth of registers at most: <= 256)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wn = wave;
    int tile_m, tile_n, zidx;
    pair_tile_map(g.M, g.N, g.tn, g.tm, g.tz, (int)blockIdx.x + g.tile0, tile_m, tile_n, zidx);
    const int m0 = tile_m * RBM, n0 = tile_n * BN;
    const char* Ab = g.A;
    const char* Bb = g.B;
    float* Cb = g.C;
    int zsplit = zidx, zb = 0;
    if (g.batch > 1) {
        Ab += (size_t)zidx * g.a_bs * 4;
        Bb += (size_t)zidx * g.b_bs * 4;
        Cb += (size_t)zidx * g.c_bs;
        zsplit = 0;
        zb = zidx;
    }
    const int kbeg = zsplit * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);
    const int nslab = (kend - kbeg) / RBK;          // even (K and k_per_split are multiples of 32)

    const char* ak = Ab + (size_t)kbeg * g.lda * 4;
    const char* bk = Bb + (size_t)kbeg * g.ldb * 4;
    const size_t a_adv = (size_t)RBK * g.lda * 4, b_adv = (size_t)RBK * g.ldb * 4;
#if CIM_RING_PRODUCER
    // The producer wave: all 24 DMA instructions of a slab (8 of A, 16 of B), one slab per step behind the step's barrier; in front of
    // the barrier of step t it waits for slab t + 1 (s_waitcnt vmcnt(48): slabs t + 2 and t + 3 stay in flight; the counter's 63 are
    // exceeded right after an issue - the hardware holds the issue back, which only this wave waits for).
    if (wave == 4) {
        StageMC<RBM, 4 * RIPA, RBK> pa;
        StageMC<BN, 4 * RIPB, RBK> pb;
        pa.init(m0, g.M, g.lda, 0, lane);
        pb.init(n0, g.N, g.ldb, 0, lane);
        int left = nslab - 1;
        for (int q = 0; q < RSTAGES - 1; ++q) {
            pa.issue(ak, smem + q * RSTAGE, 0);
            pb.issue(bk, smem + q * RSTAGE + RSTAGE_A, 0);
            const bool more = left > 0;
            ak += more ? a_adv : 0;
            bk += more ? b_adv : 0;
            left -= more ? 1 : 0;
        }
        asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int wb = RSTAGES - 1;
        for (int t = 0; t < nslab; ++t) {
            asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            pa.issue(ak, smem + wb * RSTAGE, 0);
            pb.issue(bk, smem + wb * RSTAGE + RSTAGE_A, 0);
            const bool more = left > 0;
            ak += more ? a_adv : 0;
            bk += more ? b_adv : 0;
            left -= more ? 1 : 0;
            wb = wb == RSTAGES - 1 ? 0 : wb + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
#else
    StageMC<RBM, RIPA, RBK> sa;
    StageMC<BN, RIPB, RBK> sb;
    sa.init(m0, g.M, g.lda, wave, lane);
    sb.init(n0, g.N, g.ldb, wave, lane);
#endif
    FragMC<MI, RBM> fa;
    FragMC<NI, BN> fb;
    fa.init(0, lane);
    fb.init(wn * WN, lane);

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    float bvj[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int n = n0 + wn * WN + j * 32 + (lane & 31);
        bvj[j] = (g.bias != nullptr && n < g.N) ? g.bias[n] : 0.0f;
    }

    // Every step issues exactly one slab's DMA (six instructions per wave), so the wait in front of a step's barrier is always
    // vmcnt(12); past the last slab the source stays on the last one (re-read into buffers nobody consumes: 4 of K / 16 slabs) - no
    // branch around the issue, the step is one basic block.  `left` = slabs the source pointers can still advance by.
#if !CIM_RING_PRODUCER
    int left = nslab - 1;
    auto advance = [&]() {
#if CIM_RING_DBG == 7     /* ablation: the DMA re-reads slab 0 for ever (its mechanics with constant data) */
        const bool more = false;
#else
        const bool more = left > 0;
#endif
        ak += more ? a_adv : 0;
        bk += more ? b_adv : 0;
        left -= more ? 1 : 0;
    };
    // prologue: slabs 0 .. 3 -> buffers 0 .. 3; slab 0 waited for
#pragma unroll
    for (int q = 0; q < RSTAGES - 1; ++q) {
        sa.issue(ak, smem + q * RSTAGE, wave);
        sb.issue(bk, smem + q * RSTAGE + RSTAGE_A, wave);
        advance();
    }
    asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    __syncthreads();
#define RING_WAIT() asm volatile("s_waitcnt vmcnt(12)" ::: "memory")
#define RING_ISSUE_A() sa.issue(ak, wbuf, wave)
#define RING_ISSUE_B(H) sb.issue_pair(bk, wbuf + RSTAGE_A, wave, H)
#define RING_ADVANCE() advance()
#else
    __builtin_amdgcn_s_barrier();       // (slab 0 has landed: the producer waited for it)
#define RING_WAIT()
#define RING_ISSUE_A()
#define RING_ISSUE_B(H)
#define RING_ADVANCE()
#endif

    auto main_loop = [&](auto miv_c) {
    constexpr int MIV = decltype(miv_c)::value;
    f16x8 ah0[MIV], al0[MIV], bh0[NI], bl0[NI];
    f16x8 ah1[MIV], al1[MIV], bh1[NI], bl1[NI];
#define RING_READ_B(BH, BL_, BUF)                                                              \
    _Pragma("unroll") for (int j = 0; j < NI; ++j) {                                           \
        BH[j] = fb.read((BUF) + RSTAGE_A, j, 0, 0);                                            \
        if constexpr (!ONEP) BL_[j] = fb.read((BUF) + RSTAGE_A, j, 0, 1);                      \
    }
#define RING_READ_A(AF, BUF, PL)                                                               \
    _Pragma("unroll") for (int i = 0; i < MIV; ++i) AF[i] = fa.read((BUF), i, 0, PL);
#if CIM_RING_DBG == 5 || CIM_RING_DBG == 6      /* ablation: the loop's fragment reads are left out (registers keep the first slab's) */
#define RING_LREAD_B(BH, BL_, BUF) asm volatile("" : "+v"(BH[0]), "+v"(BL_[0]));
#define RING_LREAD_A(AF, BUF, PL) asm volatile("" : "+v"(AF[0]));
#else
#define RING_LREAD_B(BH, BL_, BUF) RING_READ_B(BH, BL_, BUF)
#define RING_LREAD_A(AF, BUF, PL) RING_READ_A(AF, BUF, PL)
#endif
#define RING_MMA(AF, BF)                                                                       \
    _Pragma("unroll") for (int i = 0; i < MIV; ++i) _Pragma("unroll") for (int j = 0; j < NI; ++j) \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AF[i], BF[j], acc[i][j], 0, 0, 0)
// With ONE wave per SIMD nothing else fills the matrix pipe while this wave issues LDS reads or DMA: a step is three groups of
// MIV * NI MFMAs, each interleaved one to one with a third of the NEXT step's fragment reads (no dependence), and the step's six DMA
// instructions go out in three pieces BETWEEN the groups (a piece issues while the group's last MFMA executes).  Fused ahead of the
// MFMAs as a block (reads, then DMA) the same kernel ran at 0.36 of the f16 peak, interleaved 0.39+ (fc1's weight gradient).
#if CIM_RING_SCHED == 0
#define RING_SCHED(NREAD)
#else
#define RING_SCHED(NREAD)                                                                      \
    _Pragma("unroll") for (int q = 0; q < MIV * NI; ++q) {                                     \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                     \
        if (q < (NREAD)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                    \
    }                                                                                          \
    if ((NREAD) > MIV * NI) __builtin_amdgcn_sched_group_barrier(0x100, (NREAD) - MIV * NI, 0);\
    __builtin_amdgcn_sched_barrier(0);
#endif
// The step's barrier orders (i) every wave's part of slab T + 1 having landed (each wave waits for its own DMA in front of it) and (ii)
// the DMA of this step against the fragment reads of the buffer it overwrites - those were issued two steps ago and consumed by the last
// step's MFMAs.  The fragment reads still in flight (slab T's last ones) touch a buffer nobody writes before the NEXT barrier, by when this
// step's MFMAs have consumed them: the bare s_barrier is enough, __syncthreads()'s s_waitcnt lgkmcnt(0) in front of it would expose
// those reads' latency on every step (one wave per SIMD: nothing else runs meanwhile).
#if CIM_RING_SCHED == 2
#define RING_BARRIER() __syncthreads()
#else
#define RING_BARRIER() __builtin_amdgcn_s_barrier()
#endif
// one 16-k step: slab T + 1 has landed behind the barrier and buffer (T - 1) % 5 is free (its fragments went into registers at
// step T - 2): issue slab T + 4 into it, fetch the fragments of slab T + 1, multiply those of slab T
#define RING_STEP(AH, AL_, BH, BL_, NAH, NAL, NBH, NBL)                                        \
    {                                                                                          \
        RING_WAIT();                                                                           \
        RING_BARRIER();                                                                        \
        const char* nxt = smem + rb * RSTAGE;                                                  \
        const char* wbuf = smem + wb * RSTAGE;                                                 \
        if constexpr (ONEP) {                                                                  \
            RING_ISSUE_A();                                                                    \
            RING_ISSUE_B(0);                                                                   \
            RING_ISSUE_B(1);                                                                   \
            RING_LREAD_B(NBH, NBL, nxt)                                                         \
            RING_LREAD_A(NAH, nxt, 0)                                                           \
            RING_MMA(AH, BH);                                                                  \
            RING_SCHED(2 * NI + 2 * MIV)                                                       \
        } else {                                                                               \
            RING_LREAD_B(NBH, NBL, nxt)                                                         \
            RING_MMA(AL_, BH);                                                                 \
            RING_SCHED(4 * NI)                                                                 \
            RING_ISSUE_A();                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                 \
            RING_LREAD_A(NAL, nxt, 1)                                                           \
            RING_MMA(AH, BL_);                                                                 \
            RING_SCHED(2 * MIV)                                                                \
            RING_ISSUE_B(0);                                                                   \
            __builtin_amdgcn_sched_barrier(0);                                                 \
            RING_LREAD_A(NAH, nxt, 0)                                                           \
            RING_MMA(AH, BH);                                                                  \
            RING_SCHED(2 * MIV)                                                                \
            RING_ISSUE_B(1);                                                                   \
        }                                                                                      \
        RING_ADVANCE();                                                                        \
        (void)wbuf;                                                                            \
        wb = wb == RSTAGES - 1 ? 0 : wb + 1;                                                   \
        rb = rb == RSTAGES - 1 ? 0 : rb + 1;                                                   \
    }
    RING_READ_B(bh0, bl0, smem)
    if constexpr (!ONEP) { RING_READ_A(al0, smem, 1) }
    RING_READ_A(ah0, smem, 0)
#if CIM_RING_DBG == 5 || CIM_RING_DBG == 6
    RING_READ_B(bh1, bl1, smem)
    if constexpr (!ONEP) { RING_READ_A(al1, smem, 1) }
    RING_READ_A(ah1, smem, 0)
#endif
    int rb = 1, wb = RSTAGES - 1;           // buffers of slab t + 1 (read) and slab t + 4 (written)
    for (int t = 0; t < nslab; t += 2) {
        RING_STEP(ah0, al0, bh0, bl0, ah1, al1, bh1, bl1)
        RING_STEP(ah1, al1, bh1, bl1, ah0, al0, bh0, bl0)
    }
#undef RING_STEP
#undef RING_WAIT
#undef RING_ISSUE_A
#undef RING_ISSUE_B
#undef RING_ADVANCE
#undef RING_BARRIER
#undef RING_SCHED
#undef RING_MMA
#undef RING_READ_A
#undef RING_LREAD_A
#undef RING_LREAD_B
#undef RING_READ_B
    };
    switch (__builtin_amdgcn_readfirstlane(min(MI, (g.M - m0 + 31) / 32))) {
        case 1: main_loop(std::integral_constant<int, 1>{}); break;
        case 2: main_loop(std::integral_constant<int, 2>{}); break;
        case 3: main_loop(std::integral_constant<int, 3>{}); break;
        default: main_loop(std::integral_constant<int, MI>{}); break;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the re-read slabs of the last steps: nothing may be in flight when the stores start)
#if CIM_RING_DBG != 1
    pair_epilogue<RBM>(g, acc, bvj, Cb, zsplit, zb, m0, n0, 0, wn, lane);
#endif
}

// split-K reduce: fixed order, bias, ReLU, optional max |C|.  Round 6: a launch of at most 1024 workgroups that WALK over the result,
// and ONE max |C| atomic per workgroup (through LDS) instead of one per wave of a launch with a workgroup per 1024 elements - the
// fc1 forward's reduce (4 M elements: 16 k waves) took 54-77 us in the step for 80 MB: its first ~2000 resident waves all see a
// zero word and fire their atomicMax on ONE address (~12 ns each in L2); the same kernel without max |C| takes 13 us.
__global__ __launch_bounds__(256) void pair_splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ C,
                                                                 const float* __restrict__ bias, int M, int N, int ldc,
                                                                 int splits, long long stride, int relu,
                                                                 unsigned* __restrict__ c_amax) {
    __shared__ unsigned s_am[4];
    const long long total = (long long)M * N;
    unsigned am = 0;
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < total; i += (long long)gridDim.x * 1024) {
        const int m = (int)(i / N), n = (int)(i % N);
        float4 s = *reinterpret_cast<const float4*>(ws + i);
        for (int k0 = 1; k0 < splits; k0 += 4) {          // four partial tiles in flight, added in split order
            float4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                v[j] = k0 + j < splits ? *reinterpret_cast<const float4*>(ws + (long long)(k0 + j) * stride + i) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (k0 + j < splits) { s.x += v[j].x; s.y += v[j].y; s.z += v[j].z; s.w += v[j].w; }
        }
        if (bias) {
            const float4 b = *reinterpret_cast<const float4*>(bias + n);
            s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
        }
        if (relu) {
            s.x = fmaxf(s.x, 0.f); s.y = fmaxf(s.y, 0.f); s.z = fmaxf(s.z, 0.f); s.w = fmaxf(s.w, 0.f);
        }
        *reinterpret_cast<float4*>(C + (size_t)m * ldc + n) = s;
        am = max(am, max(max(__float_as_uint(s.x) & 0x7fffffffu, __float_as_uint(s.y) & 0x7fffffffu),
                         max(__float_as_uint(s.z) & 0x7fffffffu, __float_as_uint(s.w) & 0x7fffffffu)));
    }
    if (c_amax != nullptr) {
        am = wave_max_u32(am);
        if ((threadIdx.x & 63) == 0) s_am[threadIdx.x >> 6] = am;
        __syncthreads();
        if (threadIdx.x == 0) cim::amax_publish(c_amax, max(max(s_am[0], s_am[1]), max(s_am[2], s_am[3])));
    }
}

// ---- producers of pair images ---------------------------------------------------------------------------
using cim::pair_split2;
using cim::pair_scale_of;

// scale[i] = pair_scale_of(amax[min(i, n_amax - 1)] * factor[i])   (factor may be null = 1; amax entries are bit patterns)
// ReLU backward of a fully connected layer in front of its split: with dz = y > 0 ? dy : 0 (never stored)
//   amax = max |dz| (atomicMax of the bit pattern, caller-zeroed word) and, when `part` is given, the bias gradient's partial
//   sums part[rows_chunk][col] = sum of dz over the chunk's 64 rows, added up in row order (the caller sums the chunks).
// A block owns 64 columns x 64 rows: thread = (4 consecutive columns, row lane of 16), 4 rows each, fixed-order sum in LDS.
__global__ __launch_bounds__(256) void pair_masked_stats_kernel(const float* __restrict__ dy, const float* __restrict__ y, int rows,
                                                                int cols, float* __restrict__ part, unsigned* __restrict__ amax) {
    __shared__ float4 red[16][16];
    const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
    const int col = blockIdx.x * 64 + cx * 4, r0 = blockIdx.y * 64;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned m = 0;
    if (col < cols) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + ry * 4 + i;                      // a thread's 4 rows are consecutive: row order inside the chunk
            if (r < rows) {
                const float4 g = *reinterpret_cast<const float4*>(dy + (size_t)r * cols + col);
                const float4 v = *reinterpret_cast<const float4*>(y + (size_t)r * cols + col);
                const float4 z = make_float4(v.x > 0.f ? g.x : 0.f, v.y > 0.f ? g.y : 0.f, v.z > 0.f ? g.z : 0.f, v.w > 0.f ? g.w : 0.f);
                s.x += z.x; s.y += z.y; s.z += z.z; s.w += z.w;
                m = max(m, max(max(__float_as_uint(z.x) & 0x7fffffffu, __float_as_uint(z.y) & 0x7fffffffu),
                               max(__float_as_uint(z.z) & 0x7fffffffu, __float_as_uint(z.w) & 0x7fffffffu)));
            }
        }
    }
    // ONE atomicMax per workgroup (round 6; was one per wave: the ~2000 waves that are resident when the launch starts all see a
    // zero word and fire on one address at ~12 ns each - 65 / 35 us for the two 32 MB launches of the step)
    __shared__ unsigned s_m[4];
    m = wave_max_u32(m);
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    if (part != nullptr) red[ry][cx] = s;
    __syncthreads();
    if (threadIdx.x == 0) cim::amax_publish(amax, max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3])));
    if (part == nullptr) return;
    if (ry == 0 && col < cols) {
        float4 t = red[0][cx];
#pragma unroll
        for (int k = 1; k < 16; ++k) { const float4 u = red[k][cx]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
        *reinterpret_cast<float4*>(part + (size_t)blockIdx.y * cols + col) = t;
    }
}
__global__ void pair_scales_kernel(const unsigned* __restrict__ amax, int n_amax, const float* __restrict__ factor,
                                   float* __restrict__ scale, int n, int reduce_all) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned bits = 0;
    if (reduce_all) {                       // one scale source: the maximum of all words (a wave-strided pass + wave maximum)
        for (int t = threadIdx.x & 63; t < n_amax; t += 64) bits = max(bits, amax[t] & 0x7fffffffu);
        bits = wave_max_u32(bits);
    }
    if (i >= n) return;
    float a = __uint_as_float(reduce_all ? bits : amax[min(i, n_amax - 1)]);
    if (factor != nullptr) a *= factor[i];
    scale[i] = pair_scale_of(__float_as_uint(a));
}

// X [batch][rows][ld] fp32 (cols used, cols % 8 == 0) -> pair image [batch][rows_pad][ldp]; rows >= rows are written as zeros.
// One lane = one 8-element chunk (32 B in, 32 B out).
__global__ __launch_bounds__(256) void pair_split_kernel(const float* __restrict__ X, char* __restrict__ P, int rows,
                                                         int rows_pad, int cols, int ld, int ldp, long long x_bs,
                                                         long long p_bs, const float* __restrict__ scale,
                                                         const float* __restrict__ relu_y) {
    const int chunks = cols >> 3;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)rows_pad * chunks) return;
    const int r = (int)(idx / chunks), c = (int)(idx % chunks);
    const int z = blockIdx.y;
    uint4 h = make_uint4(0, 0, 0, 0), l = make_uint4(0, 0, 0, 0);
    if (r < rows) {
        const float s = scale[z];
        const float* src = X + (size_t)z * x_bs + (size_t)r * ld + c * 8;
        float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
        if (relu_y != nullptr) {        // x * (y > 0): the ReLU mask of the layer that produced y, fused into the split
            const float* ys = relu_y + (size_t)z * x_bs + (size_t)r * ld + c * 8;
            const float4 ya = *reinterpret_cast<const float4*>(ys), yb = *reinterpret_cast<const float4*>(ys + 4);
            a = make_float4(ya.x > 0.f ? a.x : 0.f, ya.y > 0.f ? a.y : 0.f, ya.z > 0.f ? a.z : 0.f, ya.w > 0.f ? a.w : 0.f);
            b = make_float4(yb.x > 0.f ? b.x : 0.f, yb.y > 0.f ? b.y : 0.f, yb.z > 0.f ? b.z : 0.f, yb.w > 0.f ? b.w : 0.f);
        }
        pair_split2(a.x * s, a.y * s, h.x, l.x);
        pair_split2(a.z * s, a.w * s, h.y, l.y);
        pair_split2(b.x * s, b.y * s, h.z, l.z);
        pair_split2(b.z * s, b.w * s, h.w, l.w);
    }
    char* dst = P + ((size_t)z * p_bs + (size_t)r * ldp + c * 8) * 4;
    *reinterpret_cast<uint4*>(dst) = h;
    *reinterpret_cast<uint4*>(dst + 16) = l;
}

// max |x| of a dense fp32 array as a bit pattern (atomicMax into a caller-zeroed word)
__global__ __launch_bounds__(256) void pair_amax_kernel(const float* __restrict__ X, long long n4, int tail, unsigned* __restrict__ out) {
    unsigned m = 0;
    if (blockIdx.x == 0 && (int)threadIdx.x < tail) m = __float_as_uint(X[n4 * 4 + threadIdx.x]) & 0x7fffffffu;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const uint4 v = *reinterpret_cast<const uint4*>(X + i * 4);
        m = max(m, max(max(v.x & 0x7fffffffu, v.y & 0x7fffffffu), max(v.z & 0x7fffffffu, v.w & 0x7fffffffu)));
    }
    __shared__ unsigned s_m[4];
    m = wave_max_u32(m);
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) cim::amax_publish(out, max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3])));      // one atomic per workgroup
}

template <int AL, int BL, bool RING>
int launch_pair(PairArgs g, int splits, float* workspace, hipStream_t st, int max_workgroups, int products) {
    static_assert(!RING || (AL == L_MC && BL == L_MC), "the ring kernel is the weight gradients' layout only");
    constexpr int TBM = RING ? RBM : BM, LDSB = RING ? RLDS_BYTES : LDS_BYTES, THREADS = RING ? RNT : NT;
    const int tm = (g.M + TBM - 1) / TBM, tn = (g.N + BN - 1) / BN;
    void (*kern)(const PairArgs);
    if constexpr (RING) kern = products == 1 ? gemm_pair_ring_kernel<true> : gemm_pair_ring_kernel<false>;
    else kern = products == 1 ? gemm_pair_kernel<AL, BL, true> : gemm_pair_kernel<AL, BL, false>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);
    if (e != hipSuccess) return (int)e;
    const int slabs = g.K / BK;
    if (splits < 1) splits = 1;
    if (splits > slabs) splits = slabs;
    g.k_per_split = ((slabs + splits - 1) / splits) * BK;
    splits = (g.K + g.k_per_split - 1) / g.k_per_split;
    float* out = g.C;
    const float* bias = g.bias;
    const int relu = g.relu, ldc = g.ldc;
    if (splits > 1) {
        if (g.batch > 1) return -1;
        g.C = workspace;
        g.ldc = g.N;
        g.c_split_stride = (long long)g.M * g.N;
        g.bias = nullptr;
        g.relu = 0;
    } else {
        g.c_split_stride = 0;
    }
    g.tn = tn; g.tm = tm; g.tz = g.batch > 1 ? g.batch : splits;
    const long long total = (long long)tn * tm * g.tz;
    if (total >= (1ll << 31)) return -2;
    // max_workgroups > 0: the product goes out as consecutive launches of at most that many workgroups.  A workgroup owns its CU
    // (128 KB of LDS) and the launches of a stream run one after the other, so the product never holds more CUs than that.
    const long long chunk = max_workgroups > 0 ? max_workgroups : total;
    for (long long t0 = 0; t0 < total; t0 += chunk) {
        g.tile0 = (int)t0;
        hipLaunchKernelGGL(kern, dim3((unsigned)(total - t0 < chunk ? total - t0 : chunk)), dim3(THREADS), LDSB, st, g);
    }
    if (splits > 1) {
        const long long quads = ((long long)g.M * g.N + 3) / 4;
        const long long rwgs = (quads + 255) / 256;
        hipLaunchKernelGGL(pair_splitk_reduce_kernel, dim3((unsigned)(rwgs < 1024 ? rwgs : 1024)), dim3(256), 0, st, workspace,
                           out, bias, g.M, g.N, ldc, splits, g.c_split_stride, relu, g.c_amax);
    }
    return 0;
}

int dispatch_pair(const PairArgs& g, int a_mcontig, int b_kcontig, int splits, float* workspace, hipStream_t st, int max_workgroups,
                  int products, int form) {
    if (form == 1) {        // the co-resident form exists for the weight gradients' layout (both operands K-major)
        if (a_mcontig && !b_kcontig) return launch_pair<L_MC, L_MC, true>(g, splits, workspace, st, max_workgroups, products);
        return -3;
    }
    if (!a_mcontig && !b_kcontig) return launch_pair<L_KC, L_MC, false>(g, splits, workspace, st, max_workgroups, products);
    if (!a_mcontig && b_kcontig) return launch_pair<L_KC, L_KC, false>(g, splits, workspace, st, max_workgroups, products);
    if (a_mcontig && !b_kcontig) return launch_pair<L_MC, L_MC, false>(g, splits, workspace, st, max_workgroups, products);
    return launch_pair<L_MC, L_KC, false>(g, splits, workspace, st, max_workgroups, products);
}

}  // namespace

static bool pair_dims_ok(int M, int N, int K, int lda, int ldb, int ldc, int a_mcontig, int b_kcontig) {
    if (M <= 0 || N <= 0 || K <= 0 || K % 32 != 0 || N % 4 != 0 || ldc % 4 != 0 || ldc < N) return false;
    if (lda % 8 != 0 || ldb % 8 != 0) return false;
    if (a_mcontig ? (M % 8 != 0 || lda < M) : (lda < K)) return false;
    if (b_kcontig ? (ldb < K) : (N % 8 != 0 || ldb < N)) return false;
    // the stagers address a lane's chunks with 32-bit byte offsets from the operand's base (StageKC / StageMC: r * ld * 4): the
    // addressed extent of each operand (per batch entry) must stay below 4 GiB - fc1's 822 MB weight is the largest today
    const long long ext_a = (long long)(a_mcontig ? K : M) * lda * 4, ext_b = (long long)(b_kcontig ? N : K) * ldb * 4;
    if (ext_a >= (1ll << 32) || ext_b >= (1ll << 32)) return false;
    return true;
}

extern "C" int cim_gemm_pair_splits(int M, int N, int K) {
    // same model as pick_splits() of gemm_f32.hip at this engine's rate and slab depth
    const double CUS = 256.0;
    const double tiles = (double)((M + 255) / 256) * ((N + BN - 1) / BN);
    const int slabs = K / BK;
    const double flops = 2.0 * M * (double)N * K;
    int best = 1;
    double best_t = 1e30;
    for (int s = 1; s <= 16; ++s) {
        if (s > 1 && slabs / s < 8) break;
        const double units = tiles * s;
        const double rounds = (double)(long long)((units + CUS - 1) / CUS);
        const double eff = units / (rounds * CUS);
        const double t = flops / (420e12 * eff) + (s > 1 ? (2.0 * s + 1.0) * M * (double)N * 4.0 / 4e12 : 0.0);
        if (t < best_t) { best_t = t; best = s; }
    }
    return best;
}

extern "C" int cim_gemm_pair(const void* A, const void* B, float* C, const float* bias, int M, int N, int K, int lda,
                             int ldb, int ldc, int a_mcontig, int b_kcontig, int relu, int splits, float* workspace,
                             const float* a_scale, const float* b_scale, uint32_t* c_amax, int max_workgroups, int products,
                             int form, void* stream) {
    CIM_CHECK_ARG(A && B && C && a_scale && b_scale && max_workgroups >= 0 && (products == 3 || products == 1));
    CIM_CHECK_ARG(form == 0 || (form == 1 && a_mcontig && !b_kcontig));
    CIM_CHECK_ARG(pair_dims_ok(M, N, K, lda, ldb, ldc, a_mcontig, b_kcontig));
    CIM_CHECK_ARG(splits <= 1 || workspace != nullptr);
    PairArgs g{(const char*)A, (const char*)B, C, bias, M, N, K, lda, ldb, ldc, relu, 0, 0, 1, 0, 0, 0, a_scale, b_scale, c_amax, 0, 0, 0, 0};
    int rc = dispatch_pair(g, a_mcontig, b_kcontig, splits, workspace, cim::as_stream(stream), max_workgroups, products, form);
    if (rc) { cim::set_error("cim_gemm_pair: launch setup failed (%d)", rc); return rc; }
    CIM_CHECK_LAUNCH();
    return 0;
}

extern "C" int cim_gemm_pair_batched(const void* A, const void* B, float* C, int M, int N, int K, int lda, int ldb,
                                     int ldc, int a_mcontig, int b_kcontig, int batch, long long a_bs, long long b_bs,
                                     long long c_bs, const float* a_scale, const float* b_scale, int max_workgroups, int products,
                                     int form, void* stream) {
    CIM_CHECK_ARG(A && B && C && a_scale && b_scale && batch > 0 && batch <= 65535 && max_workgroups >= 0 && (products == 3 || products == 1));
    CIM_CHECK_ARG(form == 0 || (form == 1 && a_mcontig && !b_kcontig));
    CIM_CHECK_ARG(pair_dims_ok(M, N, K, lda, ldb, ldc, a_mcontig, b_kcontig));
    CIM_CHECK_ARG(a_bs % 8 == 0 && b_bs % 8 == 0 && c_bs % 4 == 0);
    PairArgs g{(const char*)A, (const char*)B, C, nullptr, M, N, K, lda, ldb, ldc, 0, 0, 0, batch, a_bs, b_bs, c_bs, a_scale, b_scale, nullptr, 0, 0, 0, 0};
    int rc = dispatch_pair(g, a_mcontig, b_kcontig, 1, nullptr, cim::as_stream(stream), max_workgroups, products, form);
    if (rc) { cim::set_error("cim_gemm_pair_batched: launch setup failed (%d)", rc); return rc; }
    CIM_CHECK_LAUNCH();
    return 0;
}

extern "C" int cim_pair_scales(const uint32_t* amax, int n_amax, const float* factor, float* scale, int n, int reduce_all, void* stream) {
    CIM_CHECK_ARG(amax && scale && n > 0 && n_amax > 0);
    hipLaunchKernelGGL(pair_scales_kernel, dim3((n + 255) / 256), dim3(256), 0, cim::as_stream(stream), amax, n_amax, factor, scale, n,
                       reduce_all);
    CIM_CHECK_LAUNCH();
    return 0;
}

extern "C" int cim_pair_split(const float* X, void* P, int rows, int rows_pad, int cols, int ld, int ldp, int batch,
                              long long x_bs, long long p_bs, const float* scale, const float* relu_y, void* stream) {
    CIM_CHECK_ARG(X && P && scale && rows > 0 && rows_pad >= rows && cols > 0 && batch > 0 && batch <= 65535);
    CIM_CHECK_ARG(cols % 8 == 0 && ld % 4 == 0 && ld >= cols && ldp % 8 == 0 && ldp >= cols && x_bs % 4 == 0 && p_bs % 8 == 0);
    const long long chunks = (long long)rows_pad * (cols / 8);
    hipLaunchKernelGGL(pair_split_kernel, dim3((unsigned)((chunks + 255) / 256), batch), dim3(256), 0, cim::as_stream(stream), X,
                       (char*)P, rows, rows_pad, cols, ld, ldp, x_bs, p_bs, scale, relu_y);
    CIM_CHECK_LAUNCH();
    return 0;
}

extern "C" int cim_pair_masked_stats(const float* dy, const float* y, int rows, int cols, float* part, uint32_t* amax, void* stream) {
    CIM_CHECK_ARG(dy && y && amax && rows > 0 && cols > 0 && cols % 4 == 0);
    hipLaunchKernelGGL(pair_masked_stats_kernel, dim3((unsigned)((cols + 63) / 64), (unsigned)((rows + 63) / 64)), dim3(256), 0,
                       cim::as_stream(stream), dy, y, rows, cols, part, amax);
    CIM_CHECK_LAUNCH();
    return 0;
}

extern "C" int cim_pair_amax(const float* X, long long n, uint32_t* amax, void* stream) {
    CIM_CHECK_ARG(X && amax && n > 0 && ((size_t)X & 15) == 0);
    const long long n4 = n / 4;
    const unsigned blocks = (unsigned)((n4 + 256 * 8 - 1) / (256 * 8) < 2048 ? (n4 + 256 * 8 - 1) / (256 * 8) : 2048);
    hipLaunchKernelGGL(pair_amax_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, cim::as_stream(stream), X, n4, (int)(n - n4 * 4), amax);
    CIM_CHECK_LAUNCH();
    return 0;
}
