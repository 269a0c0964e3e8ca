// Complete-Instances-Mining kernels for gfx950: containment flag, per-class seed selection
// (stable top-K + greedy mask-IoU NMS), containment argmax, cross-class arbitration,
// IoU-based pseudo-label assignment.
//
// Replaces CIM_layer.{instance_nms, MIST_label, CIM_label, forward} of
// /root/reference/lib/modeling/heads.py:237-503.  All results are integer indices or exact
// copies of inputs, so they are bit-identical to the reference CPU path; the tie / compare
// rules are the ones listed in SURVEY.md App. B.  Everything here is HBM/latency-bound
// integer and fp16-compare work: no MFMA, LDS for the sort / NMS bit-matrix / scans.
#include "common.h"
#include "../../include/cim_hip.h"
#include <limits.h>

using cim::h2f;

namespace {

// ---------------------------------------------------------------- heads.py:338
// One wave per row of the N x N fp16 containment map; 16 B/lane coalesced reads.
__global__ __launch_bounds__(256) void asy_flag_kernel(const uint16_t* __restrict__ asy, int N, float thr,
                                                       double limit, int vec_ok, uint8_t* __restrict__ flag) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= N) return;
    const uint16_t* __restrict__ r = asy + (size_t)row * N;
    int cnt = 0;
    if (vec_ok) {
        const uint4* __restrict__ r4 = reinterpret_cast<const uint4*>(r);
        for (int j = lane; j < N / 8; j += 64) {
            const uint4 v = r4[j];
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                cnt += h2f((uint16_t)(w[t] & 0xffffu)) > thr;
                cnt += h2f((uint16_t)(w[t] >> 16)) > thr;
            }
        }
    } else {
        for (int j = lane; j < N; j += 64) cnt += h2f(r[j]) > thr;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (lane == 0) flag[row] = ((double)cnt < limit) ? 1 : 0;
}

// ---------------------------------------------------------------- heads.py:354-380
__device__ __forceinline__ uint32_t orderable(float f) {
    if (f == 0.0f) f = 0.0f;  // -0 == +0 for the comparison sort
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// Bitonic sort of NP = R * 1024 keys held R per lane (element e = r * 1024 + tid): partners at distance < 64 are in the
// same wave (__shfl_xor, no barrier), at distance >= 1024 in the same lane (registers); only the distances 64 ... 512 go
// through LDS - 10 exchanges (two barriers each) instead of the 55 barrier stages of the all-LDS network.  Keys are a
// total order, so any correct sort gives the stable descending argsort of heads.py:354.
template <int R>      // R = 1 or 2
__device__ __forceinline__ void seed_sort_regs(unsigned long long (&k)[R], unsigned long long* __restrict__ keys, int tid) {
    constexpr int NP = R * 1024;
#pragma unroll 1
    for (int size = 2; size <= NP; size <<= 1) {
#pragma unroll 1
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (stride >= 1024) {
                if constexpr (R == 2) {              // stride == 1024: both elements live in this lane
                    const bool up = (tid & size) == 0;
                    const unsigned long long a = k[0], b = k[R - 1];
                    if ((a > b) == up) { k[0] = b; k[R - 1] = a; }
                }
            } else if (stride >= 64) {
#pragma unroll
                for (int r = 0; r < R; ++r) keys[r * 1024 + tid] = k[r];
                __syncthreads();
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int e = r * 1024 + tid;
                    const unsigned long long o = keys[e ^ stride];
                    const bool lower = (e & stride) == 0, up = (e & size) == 0;
                    const bool take_min = lower == up;
                    k[r] = take_min ? (o < k[r] ? o : k[r]) : (o > k[r] ? o : k[r]);
                }
                __syncthreads();
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int e = r * 1024 + tid;
                    const unsigned long long o = __shfl_xor(k[r], stride);
                    const bool lower = (e & stride) == 0, up = (e & size) == 0;
                    const bool take_min = lower == up;
                    k[r] = take_min ? (o < k[r] ? o : k[r]) : (o > k[r] ? o : k[r]);
                }
            }
        }
    }
}

template <int R>
__device__ __forceinline__ void seed_topk_regs(const float* __restrict__ score, int score_ld, int col, int N,
                                               unsigned long long* __restrict__ keys, int32_t* __restrict__ kidx,
                                               int32_t* __restrict__ topk_out, int K, int tid) {
    unsigned long long k[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = r * 1024 + tid;
        k[r] = ~0ull;
        if (i < N) k[r] = ((unsigned long long)(~orderable(score[(size_t)i * score_ld + col])) << 32) | (unsigned)i;
    }
    seed_sort_regs<R>(k, keys, tid);
    if (tid < K) {                                   // K <= 1024: the K smallest keys sit in k[0] of lanes 0 .. K-1
        const int32_t id = (int32_t)(k[0] & 0xffffffffu);
        kidx[tid] = id;
        topk_out[tid] = id;
    }
    __syncthreads();
}

// grid = n_cls, block = 1024.  Dynamic LDS: [max(NP*8, K*KW*8)] bytes + K*4 bytes.
__global__ __launch_bounds__(1024) void seed_select_kernel(const float* __restrict__ score, int score_ld, int score_off,
                                                           const uint16_t* __restrict__ iou, int N, int NP,
                                                           const int32_t* __restrict__ classes, int K, int KW,
                                                           float nms_thr, size_t idx_off,
                                                           int32_t* __restrict__ topk_idx, int32_t* __restrict__ seeds,
                                                           int32_t* __restrict__ n_seeds) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);
    int32_t* kidx = reinterpret_cast<int32_t*>(smem + idx_off);
    const int ci = blockIdx.x;
    const int c = classes[ci];
    const int tid = threadIdx.x;

    // key = (descending score, ascending index): a total order -> the sort is the stable
    // descending argsort of heads.py:354 (App. B item 4).
    if (NP == 1024) {
        seed_topk_regs<1>(score, score_ld, score_off + c, N, keys, kidx, topk_idx + (size_t)ci * K, K, tid);
    } else if (NP == 2048) {
        seed_topk_regs<2>(score, score_ld, score_off + c, N, keys, kidx, topk_idx + (size_t)ci * K, K, tid);
    } else {
        for (int i = tid; i < NP; i += 1024) {
            unsigned long long key = ~0ull;
            if (i < N) {
                const float s = score[(size_t)i * score_ld + score_off + c];
                key = ((unsigned long long)(~orderable(s)) << 32) | (unsigned)i;
            }
            keys[i] = key;
        }
        __syncthreads();
        for (int size = 2; size <= NP; size <<= 1) {
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                for (int t = tid; t < NP / 2; t += 1024) {
                    const int lo = (t / stride) * (2 * stride) + (t % stride);
                    const int hi = lo + stride;
                    const bool up = ((lo & size) == 0);
                    const unsigned long long a = keys[lo], b = keys[hi];
                    if ((a > b) == up) {
                        keys[lo] = b;
                        keys[hi] = a;
                    }
                }
                __syncthreads();
            }
        }
        for (int r = tid; r < K; r += 1024) {
            const int32_t id = (int32_t)(keys[r] & 0xffffffffu);
            kidx[r] = id;
            topk_idx[(size_t)ci * K + r] = id;
        }
        __syncthreads();
    }

    // Suppression bit-matrix over the K x K gathered sub-block of the mask-IoU map:
    // bit (i, j) set <=> NOT (iou[idx_i, idx_j] < nms_thr)   (heads.py:250-254, fp16 compare).
    unsigned long long* sup = keys;  // keys are dead from here on
    const int wave = tid >> 6, lane = tid & 63;
    for (int t = wave; t < K * KW; t += 16) {
        const int i = t / KW, w = t % KW;
        const int j = w * 64 + lane;
        bool bit = false;
        if (j < K) bit = !(h2f(iou[(size_t)kidx[i] * N + kidx[j]]) < nms_thr);
        const unsigned long long word = __ballot(bit);
        if (lane == 0) sup[(size_t)i * KW + w] = word;
    }
    __syncthreads();

    // Greedy scan in score order by one wave; lane w owns word w of the "removed" set.
    if (wave == 0) {
        int cnt = 0;
        unsigned long long removed = 0ull;  // lane l holds word l (l < KW <= 64)
        for (int i = 0; i < K; ++i) {
            const unsigned long long rw = __shfl(removed, i >> 6);
            if (!((rw >> (i & 63)) & 1ull)) {
                if (lane == 0) seeds[(size_t)ci * K + cnt] = kidx[i];
                ++cnt;
                if (lane < KW) removed |= sup[(size_t)i * KW + lane];
            }
        }
        for (int r = cnt + lane; r < K; r += 64) seeds[(size_t)ci * K + r] = -1;
        if (lane == 0) n_seeds[ci] = cnt;
    }
}

// ---------------------------------------------------------------- heads.py:386-395
// grid = (K, n_cls), block = 256: one workgroup per seed column of the containment map.
__global__ __launch_bounds__(256) void contain_argmax_kernel(const uint16_t* __restrict__ asy,
                                                             const uint8_t* __restrict__ flag,
                                                             const float* __restrict__ det, int det_ld, int det_off,
                                                             int det_cstride, int N, const int32_t* __restrict__ classes,
                                                             int K, float thr, const int32_t* __restrict__ seeds,
                                                             const int32_t* __restrict__ n_seeds,
                                                             int32_t* __restrict__ res_idx) {
    const int s = blockIdx.x, ci = blockIdx.y;
    if (s >= n_seeds[ci]) {
        if (threadIdx.x == 0) res_idx[(size_t)ci * K + s] = -1;
        return;
    }
    const int c = classes[ci];
    const int seed = seeds[(size_t)ci * K + s];
    float best = -INFINITY;
    int besti = INT_MAX;
    int any = 0;
    for (int i = threadIdx.x; i < N; i += 256) {
        const bool cond = (h2f(asy[(size_t)i * N + seed]) > thr) && flag[i];
        any |= cond;
        const float v = cond ? det[(size_t)i * det_ld + det_off + c * det_cstride] : 0.0f;   // heads.py:393
        if (v > best) {  // ascending i per thread: strict '>' keeps the first maximum
            best = v;
            besti = i;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o);
        const int oi = __shfl_xor(besti, o);
        if (ov > best || (ov == best && oi < besti)) {
            best = ov;
            besti = oi;
        }
    }
    __shared__ float sv[4];
    __shared__ int si[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
        sv[wave] = best;
        si[wave] = besti;
    }
    any = __syncthreads_or(any);
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < besti)) {
                best = sv[w];
                besti = si[w];
            }
        res_idx[(size_t)ci * K + s] = any ? besti : -1;
    }
}

// ---------------------------------------------------------------- heads.py:397-405 / 306-314
// One workgroup; classes applied sequentially (App. B item 7), then an ordered compaction.
__global__ __launch_bounds__(1024) void arbitrate_kernel(const int32_t* __restrict__ cand,
                                                         const int32_t* __restrict__ classes, int n_cls, int K, int N,
                                                         const float* __restrict__ wa, int wa_ld, int wa_off,
                                                         const float* __restrict__ wb, int wb_ld, int wb_off,
                                                         int wb_cstride, int32_t* __restrict__ gt_class,
                                                         float* __restrict__ gt_weight, int32_t* __restrict__ gt_pack) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint8_t* mark = smem;                                              // [N]
    int32_t* part = reinterpret_cast<int32_t*>(smem + ((N + 15) & ~15));  // [1024]
    const int tid = threadIdx.x;
    for (int i = tid; i < N; i += 1024) {
        gt_class[i] = 0;
        gt_weight[i] = -1.0f;                                          // heads.py:336
    }
    for (int ci = 0; ci < n_cls; ++ci) {
        const int c = classes[ci];
        for (int i = tid; i < N; i += 1024) mark[i] = 0;
        __syncthreads();
        for (int r = tid; r < K; r += 1024) {
            const int32_t p = cand[(size_t)ci * K + r];
            if (p >= 0) mark[p] = 1;                                   // torch.unique: set semantics
        }
        __syncthreads();
        for (int i = tid; i < N; i += 1024) {
            if (!mark[i]) continue;
            float w = wa[(size_t)i * wa_ld + wa_off + c];
            if (wb) w = w * wb[(size_t)i * wb_ld + wb_off + c * wb_cstride];   // preds = cls * det, heads.py:330
            if (w > gt_weight[i]) {                                    // strict '>' (heads.py:397)
                gt_class[i] = c + 1;
                gt_weight[i] = w;
            }
        }
        __syncthreads();
    }
    // ordered compaction of {i : gt_class[i] > 0} (ascending i, App. B item 8)
    const int chunk = (N + 1023) / 1024;
    const int lo = tid * chunk, hi = min(N, lo + chunk);
    int cnt = 0;
    for (int i = lo; i < hi; ++i) cnt += gt_class[i] > 0;
    part[tid] = cnt;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = (tid >= o) ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    // gt_pack = [G | idx[N] | class[N] | weight bits[N]] so the host needs ONE D2H copy
    int pos = part[tid] - cnt;
    for (int i = lo; i < hi; ++i)
        if (gt_class[i] > 0) {
            gt_pack[1 + pos] = i;
            gt_pack[1 + N + pos] = gt_class[i];
            gt_pack[1 + 2 * N + pos] = __float_as_int(gt_weight[i]);
            ++pos;
        }
    if (tid == 1023) gt_pack[0] = part[1023];
}

// ---------------------------------------------------------------- heads.py:435,477-501
// One wave per proposal row: gather the G pseudo-GT columns of the fp16 mask-IoU map,
// first-index arg-max, then the ignore / background / IoU-label rules.
__global__ __launch_bounds__(256) void assign_kernel(const uint16_t* __restrict__ iou, int N,
                                                     const int32_t* __restrict__ gt_idx,
                                                     const int32_t* __restrict__ gt_cls, const float* __restrict__ gt_w,
                                                     int G, int C1, float cls_thr, float iou_thr,
                                                     float* __restrict__ pseudo_labels,
                                                     uint16_t* __restrict__ pseudo_iou, float* __restrict__ loss_w,
                                                     int32_t* __restrict__ max_idx) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= N) return;
    const uint16_t* __restrict__ r = iou + (size_t)row * N;
    float best = -INFINITY;
    int bestj = INT_MAX;
    uint16_t bestbits = 0;
    for (int j = lane; j < G; j += 64) {
        const uint16_t bits = r[gt_idx[j]];
        const float v = h2f(bits);
        if (v > best) {
            best = v;
            bestj = j;
            bestbits = bits;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o);
        const int oj = __shfl_xor(bestj, o);
        const int ob = __shfl_xor((int)bestbits, o);
        if (ov > best || (ov == best && oj < bestj)) {
            best = ov;
            bestj = oj;
            bestbits = (uint16_t)ob;
        }
    }
    if (bestj == INT_MAX) {  // every gathered value was NaN: keep memory-safe, pick column 0
        bestj = 0;
        bestbits = r[gt_idx[0]];
        best = h2f(bestbits);
    }
    const bool ignore = (best == 0.0f);                                 // heads.py:484
    const bool bg = (best < cls_thr) && !ignore;                        // heads.py:489
    const int hot = ignore ? -1 : (bg ? 0 : gt_cls[bestj]);
    for (int col = lane; col < C1; col += 64) pseudo_labels[(size_t)row * C1 + col] = (col == hot) ? 1.0f : 0.0f;
    if (lane == 0) {
        loss_w[row] = ignore ? 0.0f : gt_w[bestj];                      // heads.py:480,486
        max_idx[row] = bestj;
        // heads.py:500-501, literally: (> thr) -> 1, then (<= thr) -> 0
        uint16_t bits = bestbits;
        float v = best;
        if (v > iou_thr) { bits = 0x3C00; v = 1.0f; }
        if (v <= iou_thr) bits = 0;
        pseudo_iou[row] = bits;
    }
}

}  // namespace

extern "C" int cim_asy_flag(const uint16_t* asy_f16, int N, float con_thr, uint8_t* flag, void* stream) {
    CIM_CHECK_ARG(N >= 0);
    if (N == 0) return 0;
    CIM_CHECK_ARG(asy_f16 && flag);
    const int vec_ok = (N % 8 == 0) && ((reinterpret_cast<uintptr_t>(asy_f16) & 15) == 0);
    const double limit = 0.9 * (double)N;                               // heads.py:338 (Python float arithmetic)
    hipLaunchKernelGGL(asy_flag_kernel, dim3((N + 3) / 4), dim3(256), 0, cim::as_stream(stream), asy_f16, N,
                       cim::round_to_f16(con_thr), limit, vec_ok, flag);
    CIM_CHECK_LAUNCH();
    return 0;
}

extern "C" int cim_seed_select(const float* seed_score, int score_ld, int score_off, const uint16_t* iou_f16, int N,
                               const int32_t* classes, int n_cls, int K, float nms_thr, int32_t* topk_idx,
                               int32_t* seeds, int32_t* n_seeds, void* stream) {
    CIM_CHECK_ARG(N > 0 && N <= 8192 && K > 0 && K <= 1024 && K <= N && n_cls >= 0);
    if (n_cls == 0) return 0;
    CIM_CHECK_ARG(seed_score && iou_f16 && classes && topk_idx && seeds && n_seeds);
    int NP = 2;
    while (NP < N) NP <<= 1;
    const int KW = (K + 63) / 64;
    size_t big = (size_t)NP * 8;
    if ((size_t)K * KW * 8 > big) big = (size_t)K * KW * 8;
    big = (big + 15) & ~(size_t)15;
    const size_t lds = big + (size_t)K * 4;
    CIM_CHECK_ARG(lds <= 160 * 1024);
    if (lds > 64 * 1024)
        CIM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(seed_select_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(seed_select_kernel, dim3(n_cls), dim3(1024), lds, cim::as_stream(stream), seed_score, score_ld,
                       score_off, iou_f16, N, NP, classes, K, KW, cim::round_to_f16(nms_thr), big, topk_idx, seeds,
                       n_seeds);
    CIM_CHECK_LAUNCH();
    return 0;
}

extern "C" int cim_contain_argmax(const uint16_t* asy_f16, const uint8_t* flag, const float* det, int det_ld,
                                  int det_off, int det_cstride, int N, const int32_t* classes, int n_cls, int K,
                                  float con_thr, const int32_t* seeds, const int32_t* n_seeds, int32_t* res_idx,
                                  void* stream) {
    CIM_CHECK_ARG(N > 0 && K > 0 && n_cls >= 0 && n_cls <= 65535);
    if (n_cls == 0) return 0;
    CIM_CHECK_ARG(asy_f16 && flag && det && classes && seeds && n_seeds && res_idx);
    hipLaunchKernelGGL(contain_argmax_kernel, dim3(K, n_cls), dim3(256), 0, cim::as_stream(stream), asy_f16, flag, det,
                       det_ld, det_off, det_cstride, N, classes, K, cim::round_to_f16(con_thr), seeds, n_seeds,
                       res_idx);
    CIM_CHECK_LAUNCH();
    return 0;
}

extern "C" int cim_arbitrate(const int32_t* cand, const int32_t* classes, int n_cls, int K, int N, const float* wa,
                             int wa_ld, int wa_off, const float* wb, int wb_ld, int wb_off, int wb_cstride,
                             int32_t* gt_class, float* gt_weight, int32_t* gt_pack, void* stream) {
    CIM_CHECK_ARG(N > 0 && N <= 65536 && K > 0 && n_cls >= 0);
    CIM_CHECK_ARG(gt_class && gt_weight && gt_pack && (n_cls == 0 || (cand && classes && wa)));
    const size_t lds = ((N + 15) & ~15) + 1024 * sizeof(int32_t);
    hipLaunchKernelGGL(arbitrate_kernel, dim3(1), dim3(1024), lds, cim::as_stream(stream), cand, classes, n_cls, K, N,
                       wa, wa_ld, wa_off, wb, wb_ld, wb_off, wb_cstride, gt_class, gt_weight, gt_pack);
    CIM_CHECK_LAUNCH();
    return 0;
}

extern "C" int cim_assign(const uint16_t* iou_f16, int N, const int32_t* gt_idx, const int32_t* gt_cls,
                          const float* gt_w, int G, int C1, float cls_thr, float iou_thr, float* pseudo_labels,
                          uint16_t* pseudo_iou_f16, float* loss_weights, int32_t* max_idx, void* stream) {
    CIM_CHECK_ARG(N > 0 && G > 0 && C1 > 0);
    CIM_CHECK_ARG(iou_f16 && gt_idx && gt_cls && gt_w && pseudo_labels && pseudo_iou_f16 && loss_weights && max_idx);
    hipLaunchKernelGGL(assign_kernel, dim3((N + 3) / 4), dim3(256), 0, cim::as_stream(stream), iou_f16, N, gt_idx,
                       gt_cls, gt_w, G, C1, cim::round_to_f16(cls_thr), cim::round_to_f16(iou_thr), pseudo_labels,
                       pseudo_iou_f16, loss_weights, max_idx);
    CIM_CHECK_LAUNCH();
    return 0;
}
