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

// 16 pixels (bytes) per lane: a wave packs 1024 pixels = 16 words per load instead of 64 pixels (64 B per wave-load ran at
// 0.5 TB/s: 0.36 ms for the 187 MB of cfg2's masks).  Lane l holds bits [16 l', 16 l' + 16) of word l / 4 (l' = l % 4); the
// four lanes of a word are merged by two xor-shuffles.  Needs 4-byte aligned rows (HW % 4 == 0) for the 16-byte loads.
typedef unsigned int u32x4u __attribute__((ext_vector_type(4), aligned(4)));
__global__ __launch_bounds__(256) void mask_pack16_kernel(const uint8_t* __restrict__ m, unsigned long long* __restrict__ packed,
                                                          int N, int HW, int words, int chunks) {
    const long long gw = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (gw >= (long long)N * chunks) return;
    const int n = (int)(gw / chunks), c = (int)(gw % chunks);
    const int px0 = c * 1024 + lane * 16;
    const uint8_t* row = m + (size_t)n * HW;
    unsigned int bits = 0;
    if (px0 + 15 < HW) {
        const u32x4u v = *reinterpret_cast<const u32x4u*>(row + px0);
        const unsigned int d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int b = 0; b < 4; ++b) bits |= (((d[q] >> (8 * b)) & 0xffu) != 0u ? 1u : 0u) << (4 * q + b);
    } else {
        for (int b = 0; b < 16; ++b)
            if (px0 + b < HW && row[px0 + b] != 0) bits |= 1u << b;
    }
    unsigned long long v = (unsigned long long)bits << (16 * (lane & 3));
    v |= __shfl_xor(v, 1);
    v |= __shfl_xor(v, 2);
    const int w = c * 16 + (lane >> 2);
    if ((lane & 3) == 0 && w < words) packed[(size_t)w * N + n] = v;
}

// area[n] = popcount of mask n.  grid = (ceil(N/64), chunks): 64 masks x 4 word phases per workgroup over one chunk of the
// words (coalesced 512 B rows of the word-major layout); the chunks meet through integer atomicAdd into the zeroed array
// (chunks x N atomics: a few thousand).  The first version walked all words of a mask in ONE lane: 4 workgroups on the
// whole chip, 0.73 ms for 23 MB (32 GB/s).
__global__ __launch_bounds__(256) void mask_area_kernel(const unsigned long long* __restrict__ packed, int N, int words,
                                                        int wpc, int32_t* __restrict__ area) {
    __shared__ int part[4][64];
    const int m = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + m;
    const int w0 = blockIdx.y * wpc, w1 = min(words, w0 + wpc);
    int a = 0;
    if (n < N)
        for (int w = w0 + ph; w < w1; w += 4) a += __popcll(packed[(size_t)w * N + n]);
    part[ph][m] = a;
    __syncthreads();
    if (ph == 0 && n < N) atomicAdd(&area[n], part[0][m] + part[1][m] + part[2][m] + part[3][m]);
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

// grid = T (T + 1) / 2 tiles of the UPPER triangle (T = ceil(N/64)); block = 256 = 16 x 16 lanes, 4 x 4 outputs per lane.
// The intersection counts are symmetric: tile (ti, tj >= ti) also writes the mirrored entries (iou[j,i] = iou[i,j],
// asy[j,i] = inter / area[i]) - half the and + popcount work of the full grid.
__global__ __launch_bounds__(256) void mask_iou_pair_kernel(const unsigned long long* __restrict__ packed, int N, int words,
                                                            const int32_t* __restrict__ area,
                                                            uint16_t* __restrict__ iou, uint16_t* __restrict__ asy, int T) {
    __shared__ __attribute__((aligned(16))) unsigned long long sa[WK][TILE];
    __shared__ __attribute__((aligned(16))) unsigned long long sb[WK][TILE];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    // linear index -> (ti <= tj): row ti holds T - ti tiles
    int ti = 0, rem = blockIdx.x;
    while (rem >= T - ti) { rem -= T - ti; ++ti; }
    const int tj = ti + rem;
    const int i0 = ti * TILE, j0 = tj * TILE;
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
            const uint16_t u = ratio_f16(inter, ai + aj - inter);
            iou[(size_t)i * N + j] = u;
            asy[(size_t)i * N + j] = ratio_f16(inter, aj);
            if (ti != tj) {                               // mirrored tile
                iou[(size_t)j * N + i] = u;
                asy[(size_t)j * N + i] = ratio_f16(inter, ai);
            }
        }
    }
}

}  // namespace

extern "C" int cim_mask_pack(const uint8_t* masks_u8, uint64_t* packed, int N, int HW, void* stream) {
    CIM_CHECK_ARG(N >= 0 && HW > 0);
    if (N == 0) return 0;
    CIM_CHECK_ARG(masks_u8 && packed);
    const int words = (HW + 63) / 64;
    if (HW % 4 == 0 && (reinterpret_cast<uintptr_t>(masks_u8) & 3) == 0) {
        const int chunks = (words + 15) / 16;
        const long long waves16 = (long long)N * chunks;
        hipLaunchKernelGGL(mask_pack16_kernel, dim3((unsigned)((waves16 + 3) / 4)), dim3(256), 0, cim::as_stream(stream), masks_u8,
                           reinterpret_cast<unsigned long long*>(packed), N, HW, words, chunks);
        CIM_CHECK_LAUNCH();
        return 0;
    }
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
    CIM_CHECK_HIP(hipMemsetAsync(area, 0, sizeof(int32_t) * (size_t)N, cim::as_stream(stream)));
    const int groups = (N + 63) / 64;
    int chunks = (1024 + groups - 1) / groups;                       // ~1024 workgroups, at least 32 words each
    if (chunks > (words + 31) / 32) chunks = (words + 31) / 32;
    const int wpc = (words + chunks - 1) / chunks;
    hipLaunchKernelGGL(mask_area_kernel, dim3(groups, (words + wpc - 1) / wpc), dim3(256), 0, cim::as_stream(stream), p, N, words, wpc, area);
    const int T = (N + TILE - 1) / TILE;
    hipLaunchKernelGGL(mask_iou_pair_kernel, dim3((unsigned)((long long)T * (T + 1) / 2)), dim3(256), 0, cim::as_stream(stream), p, N, words,
                       area, iou_f16, asy_f16, T);
    CIM_CHECK_LAUNCH();
    return 0;
}
