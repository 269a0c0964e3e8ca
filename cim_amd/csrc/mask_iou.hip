// Mask-IoU and containment maps from bit-packed proposal masks, for gfx950.
//
// Replaces /root/reference/lib/utils/mask_utils.py:6-32 as driven one column at a time by
// tools/pre/create_cob_iou.py:43-49 and create_cob_asy_iou.py:43-53 (N cupy launches per
// image, offline) and the per-step pickle.load + H2D of model_builder.py:147-159.
//
//   iou[i,j] = |m_i & m_j| / |m_i | m_j|        asy[i,j] = |m_i & m_j| / |m_j|
//   rounding chain of the reference: int64 counts -> f64 divide -> f32 -> f16 (bit-exact here).
//
// Data layout in HBM: packed masks are WORD-MAJOR, packed[w*N + n] = 64 pixels of mask n, so a
// 64-mask tile of one word column is one contiguous 512 B segment.  The pair kernel is an
// integer "popcount GEMM": 64x64 output tile per workgroup, 4x4 per lane, word chunks staged
// through LDS as [word][mask] so both operand reads are conflict-free ds_read_b128.
// Integer VALU-bound (v_and + v_bcnt), no MFMA: this is bit work, not a contraction MFMA can do.
#include "common.h"
#include "../../include/cim_hip.h"

namespace {

constexpr int TILE = 64;   // masks per tile side
constexpr int WK = 32;     // 64-bit words per LDS stage

// wave per (mask n, word w): 64 lanes read 64 consecutive pixels, ballot packs them.
__global__ __launch_bounds__(256) void mask_pack_kernel(const uint8_t* __restrict__ m, unsigned long long* __restrict__ packed,
                                                        int N, int HW, int words) {
    const long long gw = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (gw >= (long long)N * words) return;
    const int n = (int)(gw / words), w = (int)(gw % words);
    const int px = w * 64 + lane;
    const bool bit = (px < HW) && (m[(size_t)n * HW + px] != 0);
    const unsigned long long word = __ballot(bit);
    if (lane == 0) packed[(size_t)w * N + n] = word;
}

__global__ __launch_bounds__(256) void mask_area_kernel(const unsigned long long* __restrict__ packed, int N, int words,
                                                        int32_t* __restrict__ area) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    int a = 0;
    for (int w = 0; w < words; ++w) a += __popcll(packed[(size_t)w * N + n]);
    area[n] = a;
}

__device__ __forceinline__ uint16_t ratio_f16(int num, int den) {
    const double q = (double)num / (double)den;       // int64 / int64 -> float64 (numpy true_divide)
    float f = (float)q;                               // stored into a float32 array (mask_utils.py:12)
    // Keep the two roundings separate: without this barrier LLVM folds f64->f32->f16 into one
    // direct f64->f16 truncation, which differs from the reference in ~4e-5 of the entries
    // (values whose f32 rounding lands on an f16 tie).
    asm volatile("" : "+v"(f));
    return __half_as_ushort(__float2half_rn(f));      // .astype(float16) (create_cob_iou.py:48)
}

// grid = (ceil(N/64), ceil(N/64)); block = 256 = 16 x 16 lanes, 4 x 4 outputs per lane.
__global__ __launch_bounds__(256) void mask_iou_pair_kernel(const unsigned long long* __restrict__ packed, int N, int words,
                                                            const int32_t* __restrict__ area,
                                                            uint16_t* __restrict__ iou, uint16_t* __restrict__ asy) {
    __shared__ __attribute__((aligned(16))) unsigned long long sa[WK][TILE];
    __shared__ __attribute__((aligned(16))) unsigned long long sb[WK][TILE];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int i0 = blockIdx.y * TILE, j0 = blockIdx.x * TILE;
    int acc[4][4] = {};
    for (int w0 = 0; w0 < words; w0 += WK) {
        // stage: 2 * WK * TILE words, 256 lanes -> 8 + 8 loads per lane, 512 B contiguous per wave-row
        for (int e = tid; e < WK * TILE; e += 256) {
            const int wl = e / TILE, r = e % TILE;
            const int w = w0 + wl;
            unsigned long long va = 0ull, vb = 0ull;
            if (w < words) {
                if (i0 + r < N) va = packed[(size_t)w * N + i0 + r];
                if (j0 + r < N) vb = packed[(size_t)w * N + j0 + r];
            }
            sa[wl][r] = va;
            sb[wl][r] = vb;
        }
        __syncthreads();
#pragma unroll 4
        for (int wl = 0; wl < WK; ++wl) {
            unsigned long long a[4], b[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                a[q] = sa[wl][ty * 4 + q];
                b[q] = sb[wl][tx * 4 + q];
            }
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[p][q] += __popcll(a[p] & b[q]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int i = i0 + ty * 4 + p;
        if (i >= N) continue;
        const int ai = area[i];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = j0 + tx * 4 + q;
            if (j >= N) continue;
            const int aj = area[j];
            const int inter = acc[p][q];
            iou[(size_t)i * N + j] = ratio_f16(inter, ai + aj - inter);
            asy[(size_t)i * N + j] = ratio_f16(inter, aj);
        }
    }
}

}  // namespace

extern "C" int cim_mask_pack(const uint8_t* masks_u8, uint64_t* packed, int N, int HW, void* stream) {
    CIM_CHECK_ARG(N >= 0 && HW > 0);
    if (N == 0) return 0;
    CIM_CHECK_ARG(masks_u8 && packed);
    const int words = (HW + 63) / 64;
    const long long waves = (long long)N * words;
    hipLaunchKernelGGL(mask_pack_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, cim::as_stream(stream),
                       masks_u8, reinterpret_cast<unsigned long long*>(packed), N, HW, words);
    CIM_CHECK_LAUNCH();
    return 0;
}

extern "C" int cim_mask_iou_pair(const uint64_t* packed, int N, int words, int32_t* area, uint16_t* iou_f16,
                                 uint16_t* asy_f16, void* stream) {
    CIM_CHECK_ARG(N >= 0 && words > 0);
    if (N == 0) return 0;
    CIM_CHECK_ARG(packed && area && iou_f16 && asy_f16);
    CIM_CHECK_ARG((long long)words * 64 < (1ll << 31));
    const unsigned long long* p = reinterpret_cast<const unsigned long long*>(packed);
    hipLaunchKernelGGL(mask_area_kernel, dim3((N + 255) / 256), dim3(256), 0, cim::as_stream(stream), p, N, words, area);
    const int T = (N + TILE - 1) / TILE;
    hipLaunchKernelGGL(mask_iou_pair_kernel, dim3(T, T), dim3(256), 0, cim::as_stream(stream), p, N, words, area,
                       iou_f16, asy_f16);
    CIM_CHECK_LAUNCH();
    return 0;
}
