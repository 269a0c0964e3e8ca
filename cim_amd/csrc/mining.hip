// Complete-Instances-Mining kernels for gfx950: containment flag, per-class seed selection
// (stable top-K + greedy mask-IoU NMS), containment argmax, cross-class arbitration,
// anti-noise sampling (NumPy's np.random.choice restated on pre-drawn MT19937 uniforms),
// IoU-based pseudo-label assignment - for ALL CIM layers of a training step in 2 launches on the step's stream
// (+ the input-only preparation of the containment map - flags and a transposed copy - which the host runs ahead of the
// backbone on a side stream), with no host round trip.
//
// Replaces CIM_layer.{instance_nms, MIST_label, CIM_label, forward} of
// /root/reference/lib/modeling/heads.py:237-503.  All results are integer indices or exact
// copies of inputs, so they are bit-identical to the reference CPU path; the tie / compare
// rules are the ones listed in SURVEY.md App. B.  Everything here is HBM/latency-bound
// integer and fp16-compare work: no MFMA, LDS for the select / NMS bit-matrix / scans.
//
// The image's class list is never known to the host: every kernel reads `labels` [C] on the device
// (class c is active iff labels[c] != 0, heads.py:340) and the per-class outputs are indexed by the
// class id itself ([C][K] arrays), inactive classes exit at once.
//
// No library state: the words through which the workgroups of the mining launch meet (arrival counters, list lengths) live in a
// CALLER-owned scratch (cim_mining_sync_bytes(), zero-filled once at allocation) that every call leaves zeroed again.
#include "common.h"
#include "../../include/cim_hip.h"
#include <limits.h>

using cim::h2f;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

#ifndef CIM_MINING_CLOCKS
#define CIM_MINING_CLOCKS 0          // 1: phase stamps (100 MHz wall clock) of the mining launch -> cim_debug_mining_clocks (tools only)
#endif
#if CIM_MINING_CLOCKS
__device__ unsigned long long g_mining_clk[2][8][16];                    // [phase: seed / arbitrate][workgroup (first 8)][stamp]
#define MCLK(KERNEL, WG, I) do { if (threadIdx.x == 0 && (WG) < 8) g_mining_clk[KERNEL][WG][I] = wall_clock64(); } while (0)
extern "C" int cim_debug_mining_clocks(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_mining_clk), sizeof(g_mining_clk)) == hipSuccess ? 0 : 1;
}
#else
#define MCLK(KERNEL, WG, I) do { } while (0)
#endif

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

// The containment map transposed (asy_t[j][i] = asy[i][j], rows padded to ldt = 8 ceil(N / 8) entries: 16-byte aligned rows): the
// mining reads COLUMNS of the map (every proposal against one seed, heads.py:386) - in the transposed copy such a column is
// one contiguous row.  64 x 64 tiles through LDS, both global sides in 128-byte runs; the pad entries are written as zero.
__global__ __launch_bounds__(256) void transpose_f16_kernel(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst, int N, int ldt) {
    __shared__ uint16_t tile[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll 4
    for (int k = ty; k < 64; k += 4) {
        const int r = r0 + k, c = c0 + tx;
        tile[k][tx] = (r < N && c < N) ? src[(size_t)r * N + c] : (uint16_t)0;
    }
    __syncthreads();
#pragma unroll 4
    for (int k = ty; k < 64; k += 4) {
        const int r = c0 + k, c = r0 + tx;          // row of dst = column of src
        if (r < N && c < ldt) dst[(size_t)r * ldt + c] = c < N ? tile[tx][k] : (uint16_t)0;
    }
}

// ---------------------------------------------------------------- heads.py:354-380
__device__ __forceinline__ uint32_t orderable(float f) {
    if (f == 0.0f) f = 0.0f;  // -0 == +0 for the comparison sort
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ int block_exclusive_scan(int v, int* part, int* total) {      // 1024 lanes; returns the exclusive prefix
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;       // wave scans in registers, 16 wave totals through LDS:
    int incl = v;                                                        // two barriers instead of the 20 of an all-LDS scan
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int x = __shfl_up(incl, o);
        if (lane >= o) incl += x;
    }
    __syncthreads();                                                     // (part[] of the previous scan has been read)
    if (lane == 63) part[wave] = incl;
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const int p = part[w];
        tot += p;
        woff += (w < wave) ? p : 0;
    }
    *total = tot;
    return woff + incl - v;
}

// The words the workgroups of the mining launch meet through (caller-owned, zero between calls).
struct MiningSync {
    unsigned arrive[CIM_MAX_LAYERS];                 // (class, layer) workgroups of a layer that finished their seed phase
    unsigned done;                                   // arbitrators that finished
    unsigned pad_[3];
    unsigned long long gword[CIM_MAX_LAYERS];        // bit 63: published | error bits << 32 | list length of a sampling layer
};

// LDS layout of the seed phase (byte offsets; the arbitration phase re-uses the array from offset 0)
struct SeedLayout {
    int off_cand, off_sup, off_kidx, off_seed, off_det, off_flag, off_row;
    int det_lds;                                     // the layer's detector column and the flags staged in LDS (else read from memory)
    int row_lds, row_halves;                         // the NMS matrix from whole map rows streamed through 16 wave-private LDS rows
    int total;
};

// ================================================================== the mining launch, phase 1: seeds of one (class, layer)
// Top-K by RADIX SELECT (the reference takes argsort(descending)[:K], heads.py:354: only K = ceil(0.1 N) of the N keys are
// wanted in order): four 8-bit histogram passes over the orderable score bits find the K-th largest value T and how many
// keys equal to T belong to the top K (ties: the lowest proposal indices - a stable descending sort, App. B item 4); the K
// survivors are compacted by wave ballots and ranked among themselves by counting (K^2 / 1024 compares per lane).  One
// barrier per pass: the histograms of all four passes are cleared up front and every wave scans a pass's 256 bins for
// itself.  Round 4 sorted all N keys (bitonic, 55 exchange stages): 12.5 of the kernel's 37 us.
template <int MAXR>
__device__ __forceinline__ void seed_topk_select(const float (&sc)[MAXR], int N, int K,
                                                 unsigned* __restrict__ hist, unsigned long long* __restrict__ cand,
                                                 int32_t* __restrict__ kidx, int32_t* __restrict__ topk_out, int* s_misc,
                                                 int* s_part, int tid) {
    const int lane = tid & 63;
    const int Rk = (N + 1023) >> 10;                 // keys per lane: lane t owns the proposals [t Rk, t Rk + Rk) (scores in sc[])
    unsigned key[MAXR];
    unsigned valid = 0;
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
        key[r] = 0;
        const int i = tid * Rk + r;
        if (r < Rk && i < N) {
            key[r] = orderable(sc[r]);
            valid |= 1u << r;
        }
    }
    hist[tid] = 0;                                   // the four passes' histograms (4 x 256 words), cleared once
    if (tid == 0) s_misc[2] = 0;                     // (the compaction counter below)
    __syncthreads();
    unsigned prefix = 0, pmask = 0;
    int need = K, tot_eq = 0;
#pragma unroll 1
    for (int pass = 0; pass < 4; ++pass) {           // ONE barrier per pass: every wave scans the histogram for itself
        const int shift = 24 - 8 * pass;
        unsigned* __restrict__ h = hist + 256 * pass;
#pragma unroll
        for (int r = 0; r < MAXR; ++r)
            if (((valid >> r) & 1u) && (key[r] & pmask) == prefix) atomicAdd(&h[(key[r] >> shift) & 255u], 1u);
        __syncthreads();
        // lane holds the bins 4 lane .. 4 lane + 3; suffix sums from the top bin down
        const uint4 hv = reinterpret_cast<const uint4*>(h)[lane];
        const int cnt[4] = {(int)hv.x, (int)hv.y, (int)hv.z, (int)hv.w};
        const int s4 = cnt[0] + cnt[1] + cnt[2] + cnt[3];
        int suf = s4;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int x = __shfl_down(suf, o);
            if (lane + o < 64) suf += x;
        }
        int above = suf - s4;                        // keys in the bins of higher lanes
        int found = -1, left = 0, inbin = 0;
#pragma unroll
        for (int b = 3; b >= 0; --b) {
            if (above < need && need <= above + cnt[b]) {
                found = 4 * lane + b;                // the bin that holds the need-th largest key
                left = need - above;
                inbin = cnt[b];
            }
            above += cnt[b];
        }
        const int src = __builtin_ctzll(__ballot(found >= 0));              // exactly one lane found it
        prefix |= (unsigned)__shfl(found, src) << shift;
        pmask |= 255u << shift;
        need = __shfl(left, src);
        tot_eq = __shfl(inbin, src);
    }
    // T = prefix: `need` of the tot_eq keys equal to T are in - the lowest proposal indices (only when the boundary value is
    // duplicated does the order among the ties matter)
    int tie_before = 0;
    if (tot_eq > need) {                             // (uniform)
        int tie = 0;
#pragma unroll
        for (int r = 0; r < MAXR; ++r) tie += (((valid >> r) & 1u) && key[r] == prefix) ? 1 : 0;
        int tie_total;
        tie_before = block_exclusive_scan(tie, s_part, &tie_total);
    }
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
        if (r >= Rk) break;                          // (uniform)
        bool take = false;
        if ((valid >> r) & 1u) {
            take = key[r] > prefix;
            if (key[r] == prefix) take = tie_before++ < need;
        }
        const unsigned long long m = __ballot(take);                         // one LDS atomic per wave
        int base = 0;
        if (lane == 0 && m) base = atomicAdd(&s_misc[2], __popcll(m));
        base = __shfl(base, 0);
        if (take) cand[base + __popcll(m & ((1ull << lane) - 1ull))] =
            ((unsigned long long)(~key[r]) << 32) | (unsigned)(tid * Rk + r);     // (descending score, ascending index)
    }
    __syncthreads();
    // rank of each survivor among the K: G lanes per survivor count a share of the list each
    int Kp = 1;
    while (Kp < K) Kp <<= 1;
    int G = 1024 / Kp;
    if (G > 64) G = 64;
    const int j = tid / G, sub = tid - j * G;
    unsigned long long me = 0ull;
    int cnt = 0;
    if (j < K) {
        me = cand[j];
        for (int m = sub; m < K; m += G) cnt += cand[m] < me ? 1 : 0;
    }
    for (int o = G >> 1; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (j < K && sub == 0) {
        const int32_t id = (int32_t)(me & 0xffffffffu);
        kidx[cnt] = id;
        topk_out[cnt] = id;
    }
    __syncthreads();
}

// Seeds of (class c, layer l): top-K select, K x K suppression bit-matrix, greedy scan, and - CIM layers - the containment
// arg-max of every seed (heads.py:386-395; round 4: a launch of its own, one workgroup per seed column): one wave per seed
// walks the seed's column of the containment map - a contiguous row of the transposed copy - against the layer's detector
// column, first-index arg-max.
__device__ __forceinline__ void seed_phase(const cim_mining_args& a, const SeedLayout& lay, unsigned char* smem, int* s_misc,
                                           int* s_part, int c, int l) {
    const cim_mining_layer& L = a.layer[l];
    const int tid = threadIdx.x, N = a.N, K = a.K;
    const int wave = tid >> 6, lane = tid & 63;
    const int KW = (K + 63) >> 6;
    unsigned* hist = reinterpret_cast<unsigned*>(smem);
    unsigned long long* cand = reinterpret_cast<unsigned long long*>(smem + lay.off_cand);
    unsigned long long* sup = reinterpret_cast<unsigned long long*>(smem + lay.off_sup);
    int32_t* kidx = reinterpret_cast<int32_t*>(smem + lay.off_kidx);
    int32_t* sseed = reinterpret_cast<int32_t*>(smem + lay.off_seed);
    float* sdet = reinterpret_cast<float*>(smem + lay.off_det);
    uint8_t* sflag = smem + lay.off_flag;
    int32_t* __restrict__ topk_out = L.topk + (size_t)c * K;
    int32_t* __restrict__ seeds = L.seeds + (size_t)c * K;
    [[maybe_unused]] const int wg_act = (int)(l * 2 + (c & 1));          // (debug stamps only)
    MCLK(0, wg_act, 0);
    const uint8_t* __restrict__ flag = L.using_cim ? a.flags + (size_t)L.flag_slot * N : nullptr;
    // ONE round trip for everything this phase reads per proposal: lane t owns the proposals [t Rk, t Rk + Rk) - their seed
    // scores stay in registers (the select's keys), their detector scores and flags go to LDS for the containment step
    constexpr int MAXR = 8;
    const int Rk = (N + 1023) >> 10;
    float sc[MAXR];
    {
        const float* __restrict__ score = L.seed_score;
        const int score_ld = L.seed_ld, col = L.seed_off + c;
        const bool stage = L.using_cim && lay.det_lds;
        float dv[MAXR];
        uint8_t fv[MAXR];
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {
            const int i = tid * Rk + r;
            const bool in = r < Rk && i < N;
            sc[r] = in ? score[(size_t)i * score_ld + col] : 0.0f;
            dv[r] = (in && stage) ? L.det[(size_t)i * L.det_ld + L.det_off + c * L.det_cs] : 0.0f;
            fv[r] = (in && stage) ? flag[i] : (uint8_t)0;
        }
        if (stage) {
#pragma unroll
            for (int r = 0; r < MAXR; ++r) {
                const int i = tid * Rk + r;
                if (r < Rk && i < N) {
                    sdet[i] = dv[r];
                    sflag[i] = fv[r];
                }
            }
        }
    }
    // key = (descending score, ascending index): a total order -> the stable descending argsort of heads.py:354 (App. B item 4)
    seed_topk_select<MAXR>(sc, N, K, hist, cand, kidx, topk_out, s_misc, s_part, tid);
    MCLK(0, wg_act, 1);
    // Suppression bit-matrix over the K x K gathered sub-block of the mask-IoU map:
    // bit (i, j) set <=> NOT (iou[idx_i, idx_j] < nms_thr)   (heads.py:250-254, fp16 compare).
    const float nms_thr = L.nms_thr;
    if (lay.row_lds) {
        // ROW STREAMING: one wave per candidate row reads the candidate's WHOLE row of the map in 16-byte pieces (coalesced; the K
        // wanted columns - a tenth of the row, scattered - touch nearly every 64-byte line of it anyway: gathering them one by one
        // moved 3x the bytes through this CU's L1, 24 of the phase's 52 us at 2000 proposals), parks it in a wave-private LDS
        // row and picks its K columns from there (one 2-byte LDS read per candidate and 64-candidate word, a ballot per word).
        // Rows start at any 2-byte offset: the pieces are loaded from the 16-byte boundary below through a buffer resource over
        // the map (reads behind its end return 0), the columns are shifted accordingly.  The next row's pieces are in flight
        // while this one is picked apart.
        const rsrc_t R = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.iou), 0, (int)(2u * (unsigned)N * (unsigned)N), 0x00020000);
        uint16_t* __restrict__ myrow = reinterpret_cast<uint16_t*>(smem + lay.off_row) + (size_t)wave * lay.row_halves;
        const int nch = lay.row_halves >> 3;                             // 16-byte pieces per row (<= 4 per lane)
        int cj[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int j = w * 64 + lane;
            cj[w] = (w < KW && j < K) ? kidx[j] : -1;
        }
        u32x4 cur[4], nxt[4];
        int sh_c = 0, sh_n = 0;
        auto issue = [&](int i, u32x4 (&buf)[4], int& shift) {
            const unsigned b = 2u * (unsigned)kidx[i] * (unsigned)N;
            shift = (int)((b & 15u) >> 1);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ch = lane + 64 * k;
                buf[k] = ch < nch ? __builtin_amdgcn_raw_buffer_load_b128(R, (b & ~15u) + 16u * (unsigned)ch, 0, 0) : u32x4{0, 0, 0, 0};
            }
        };
        int i = wave;
        if (i < K) issue(i, cur, sh_c);
        for (; i < K; i += 16) {
            if (i + 16 < K) issue(i + 16, nxt, sh_n);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ch = lane + 64 * k;
                if (ch < nch) *reinterpret_cast<u32x4*>(myrow + ch * 8) = cur[k];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // (LDS operations of a wave execute in order: no wait needed,
            __builtin_amdgcn_wave_barrier();                             // only the compiler must keep the writes above the reads)
            const int w_lo = i >> 6;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                if (w >= KW) break;
                unsigned long long word = 0ull;
                if (w >= w_lo) {                                         // (the greedy scan only looks at candidates behind row i)
                    const uint16_t v = cj[w] >= 0 ? myrow[cj[w] + sh_c] : (uint16_t)0;
                    word = __ballot(cj[w] >= 0 && !(h2f(v) < nms_thr));
                }
                if (lane == 0) sup[(size_t)i * KW + w] = word;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 4; ++k) cur[k] = nxt[k];
            sh_c = sh_n;
        }
    } else {
        constexpr int GU = 13;                                               // gathers in flight per wave (latency-bound: the map is
        for (int t0 = wave; t0 < K * KW; t0 += 16 * GU) {                    // L2-resident, each gather a round trip; K = 100: one round)
            uint16_t v[GU];
    #pragma unroll
            for (int u = 0; u < GU; ++u) {
                const int t = t0 + 16 * u;
                const int i = t / KW, w = t % KW, j = w * 64 + lane;
                // (the greedy scan only looks at candidates behind row i: the words below its own 64-block stay zero)
                v[u] = (t < K * KW && j < K && w >= (i >> 6)) ? a.iou[(size_t)kidx[i] * N + kidx[j]] : (uint16_t)0;
            }
    #pragma unroll
            for (int u = 0; u < GU; ++u) {
                const int t = t0 + 16 * u;
                if (t >= K * KW) break;
                const int i = t / KW, w = t % KW, j = w * 64 + lane;
                const unsigned long long word = __ballot(j < K && w >= (i >> 6) && !(h2f(v[u]) < nms_thr));
                if (lane == 0) sup[t] = word;
            }
        }
    }
    __syncthreads();
    MCLK(0, wg_act, 2);

    // Greedy scan in score order by one wave; lane w owns word w of the "removed" set.  The scan is sequential in the
    // KEPT candidates, 64 candidates at a time: the block's own 64 x 64 suppression bits sit one row per lane in
    // registers, the next kept candidate is a count-trailing-zeros on the scalar "removed" word and its row a readlane (no LDS
    // on the serial chain); the rows of the kept candidates are then OR-ed into the later words with independent LDS reads.
    if (wave == 0) {
        int cnt = 0;
        unsigned long long removed = 0ull;  // lane l holds word l (l < KW <= 64)
        for (int i0 = 0; i0 < K; i0 += 64) {
            const int nb = min(64, K - i0), wb = i0 >> 6;
            const unsigned long long rowblk = lane < nb ? sup[(size_t)(i0 + lane) * KW + wb] : 0ull;
            const unsigned rlo = (unsigned)rowblk, rhi = (unsigned)(rowblk >> 32);
            // rem / kept live in SGPRs (readfirstlane): the 64 steps are scalar shifts and selects, the row reads
            // (v_readlane, independent of rem) run ahead of them
            const unsigned long long r0 = __shfl(removed, wb);
            unsigned long long rem = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(r0 >> 32)) << 32) |
                                     (unsigned)__builtin_amdgcn_readfirstlane((int)r0);
            unsigned long long kept = 0ull;
            // walk the ALIVE candidates only: the next one is the lowest clear bit of `rem` (a kept candidate's row always has
            // its own bit set - iou(i, i) is 1 or NaN, neither is < nms_thr - and is cleared explicitly as well)
            const unsigned long long inblk = nb == 64 ? ~0ull : ((1ull << nb) - 1ull);
            unsigned long long alive = ~rem & inblk;
            while (alive) {
                const int t = __builtin_amdgcn_readfirstlane(__builtin_ctzll(alive));
                const unsigned long long row = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(rhi, t) << 32) |
                                               (unsigned)__builtin_amdgcn_readlane(rlo, t);          // (readlane returns int)
                kept |= 1ull << t;
                rem |= row | (1ull << t);
                alive = ~rem & inblk & (t == 63 ? 0ull : (~0ull << (t + 1)));
            }
            if ((kept >> lane) & 1ull) {
                const int at = cnt + __popcll(kept & ((1ull << lane) - 1ull));
                const int32_t id = kidx[i0 + lane];
                seeds[at] = id;
                sseed[at] = id;
            }
            cnt += __popcll(kept);
            if (lane < KW)
                for (unsigned long long k = kept; k;) {                  // four independent LDS reads per round (a repeated row is harmless)
                    const int t0 = __builtin_ctzll(k);
                    k &= k - 1ull;
                    const int t1 = k ? __builtin_ctzll(k) : t0;
                    k &= k - 1ull;
                    const int t2 = k ? __builtin_ctzll(k) : t0;
                    k &= k - 1ull;
                    const int t3 = k ? __builtin_ctzll(k) : t0;
                    k &= k - 1ull;
                    removed |= (sup[(size_t)(i0 + t0) * KW + lane] | sup[(size_t)(i0 + t1) * KW + lane]) |
                               (sup[(size_t)(i0 + t2) * KW + lane] | sup[(size_t)(i0 + t3) * KW + lane]);
                }
        }
        for (int r = cnt + lane; r < K; r += 64) seeds[r] = -1;
        if (lane == 0) {
            L.n_seeds[c] = cnt;
            s_misc[3] = cnt;
        }
    }
    __syncthreads();
    MCLK(0, wg_act, 3);
    if (!L.using_cim) return;                                            // MIST_label: the seeds are the pseudo ground truths

    // ---- containment arg-max (heads.py:386-395): res[s] = the best-det proposal among {i : asy[i, seed_s] > con_thr and not huge(i)}
    const int n_seeds = s_misc[3];
    const float thr = L.con_thr;
    int32_t* __restrict__ res = L.res + (size_t)c * K;
    if (a.asy_t != nullptr) {
        // Two seeds per wave and round, a lane takes 8 consecutive proposals per 512-proposal chunk: one 16-byte load per seed and
        // chunk from the seed's row of the transposed map (rows padded to 8 entries: 16-byte aligned), all of a round's loads in
        // flight before the first compare.
        const int ldt = (N + 7) & ~7;
        for (int s0 = wave * 2; s0 < K; s0 += 32) {
            const bool on0 = s0 < n_seeds, on1 = s0 + 1 < n_seeds;       // (uniform per wave)
            if (!on0) {
                if (lane == 0) res[s0] = -1;
                if (lane == 1 && s0 + 1 < K) res[s0 + 1] = -1;
                continue;
            }
            const uint4* __restrict__ row0 = reinterpret_cast<const uint4*>(a.asy_t + (size_t)sseed[s0] * ldt);
            const uint4* __restrict__ row1 = reinterpret_cast<const uint4*>(a.asy_t + (size_t)sseed[on1 ? s0 + 1 : s0] * ldt);
            float best0 = -INFINITY, best1 = -INFINITY;
            int bi0 = INT_MAX, bi1 = INT_MAX, any0 = 0, any1 = 0;
            for (int i0 = lane * 8; i0 < ldt; i0 += 512) {
                const uint4 v0 = row0[i0 >> 3], v1 = row1[i0 >> 3];
                float d[8];
                unsigned fl = 0;
                if (lay.det_lds) {
                    const float4 da = *reinterpret_cast<const float4*>(sdet + i0), db = *reinterpret_cast<const float4*>(sdet + i0 + 4);
                    const uint2 fb = *reinterpret_cast<const uint2*>(sflag + i0);
                    d[0] = da.x; d[1] = da.y; d[2] = da.z; d[3] = da.w; d[4] = db.x; d[5] = db.y; d[6] = db.z; d[7] = db.w;
#pragma unroll
                    for (int e = 0; e < 8; ++e) fl |= ((((e < 4 ? fb.x : fb.y) >> (8 * (e & 3))) & 0xffu) ? 1u : 0u) << e;
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int i = i0 + e;
                        d[e] = i < N ? L.det[(size_t)i * L.det_ld + L.det_off + c * L.det_cs] : 0.0f;
                        fl |= (i < N && flag[i] ? 1u : 0u) << e;
                    }
                }
                const unsigned w0[4] = {v0.x, v0.y, v0.z, v0.w}, w1[4] = {v1.x, v1.y, v1.z, v1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int i = i0 + e;
                    const bool in = i < N && ((fl >> e) & 1u);
                    const bool c0 = in && h2f((uint16_t)(w0[e >> 1] >> (16 * (e & 1)))) > thr;
                    const bool c1 = in && h2f((uint16_t)(w1[e >> 1] >> (16 * (e & 1)))) > thr;
                    any0 |= c0;
                    any1 |= c1;
                    const float x0 = c0 ? d[e] : 0.0f, x1 = c1 ? d[e] : 0.0f;        // heads.py:393
                    if (i < N && x0 > best0) { best0 = x0; bi0 = i; }                 // ascending i per lane: strict '>' keeps the first maximum
                    if (i < N && x1 > best1) { best1 = x1; bi1 = i; }
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov0 = __shfl_xor(best0, o), ov1 = __shfl_xor(best1, o);
                const int oi0 = __shfl_xor(bi0, o), oi1 = __shfl_xor(bi1, o);
                if (ov0 > best0 || (ov0 == best0 && oi0 < bi0)) { best0 = ov0; bi0 = oi0; }
                if (ov1 > best1 || (ov1 == best1 && oi1 < bi1)) { best1 = ov1; bi1 = oi1; }
            }
            any0 = __any(any0);
            any1 = __any(any1);
            if (lane == 0) res[s0] = any0 ? bi0 : -1;
            if (lane == 1 && s0 + 1 < K) res[s0 + 1] = (on1 && any1) ? bi1 : -1;
        }
    } else {
        for (int s = wave; s < K; s += 16) {                             // no transposed copy: strided reads of the map's column
            if (s >= n_seeds) {
                if (lane == 0) res[s] = -1;
                continue;
            }
            const int seed = sseed[s];
            float best = -INFINITY;
            int besti = INT_MAX;
            int any = 0;
#pragma unroll 4
            for (int i = lane; i < N; i += 64) {
                const bool cond = (h2f(a.asy[(size_t)i * N + seed]) > thr) && flag[i];
                any |= cond;
                const float v = cond ? L.det[(size_t)i * L.det_ld + L.det_off + c * L.det_cs] : 0.0f;
                if (v > best) {
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
            any = __any(any);
            if (lane == 0) res[s] = any ? besti : -1;
        }
    }
    MCLK(0, wg_act, 4);
}

// ================================================================== the mining launch, phase 2 - the NumPy arithmetic it restates
// Per layer: cross-class arbitration (heads.py:397-405 / 306-314), ordered compaction (App. B item 8), anti-noise
// sampling (heads.py:447-473) on the uniforms the reference's sequential CIM_layer calls would draw, compaction of the
// survivors.
//
// Sampling = np.random.choice(class_idx, size=n, replace=True, p=prob / prob.sum()) + np.unique, restated
// (numpy/random/mtrand.pyx, RandomState.choice):
//     prob  = gt_weights[class_idx]                       f32
//     total = prob.sum()                                  f32, NumPy's pairwise summation (np_pairwise_sum below)
//     p     = prob / total                                f32 division (evaluated in f64 and rounded once: exact)
//     cdf   = cumsum(f64(p)); cdf /= cdf[-1]              sequential f64 accumulation
//     idx   = searchsorted(cdf, u, side='right')          u = the next n doubles of the global MT19937 stream
// The host draws the uniforms (np.random.random_sample) BEFORE the launch, at most `max_uniforms` of them, and
// rewinds its generator to `used` afterwards (cim_amd/modeling/heads.py: RngLedger): same values, same stream position.

// numpy/core/src/umath/loops_utils.h.src, @TYPE@_pairwise_sum (PW_BLOCKSIZE = 128, 8 accumulators).  NumPy's version
// recurses on halves above 128 elements; here the recursion is unrolled by a template depth (D = 4: n <= 2048).
__device__ __forceinline__ float np_pairwise_leaf(const float* a, int n) {      // n <= 128
    if (n < 8) {
        float res = 0.0f;
        for (int i = 0; i < n; ++i) res += a[i];
        return res;
    }
    float r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
        r0 += a[i + 0]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3];
        r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7];
    }
    float res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; ++i) res += a[i];
    return res;
}

template <int D>
__device__ float np_pairwise_sum(const float* a, int n) {
    if (n <= 128) return np_pairwise_leaf(a, n);
    if constexpr (D == 0) {
        return np_pairwise_leaf(a, 128);           // unreachable for n <= 128 << depth (checked by the caller)
    } else {
        int n2 = n / 2;
        n2 -= n2 % 8;
        const float lo = np_pairwise_sum<D - 1>(a, n2);
        const float hi = np_pairwise_sum<D - 1>(a + n2, n - n2);
        return lo + hi;
    }
}

// Arbitration + sampling of layer l by ONE workgroup - the layer's LAST ARRIVING seed workgroup (round 4: a launch of its
// own, one workgroup per layer).  Arbitration and the ordered compaction of a layer depend on nothing but that layer's
// scores, so the layers run side by side; only the anti-noise sampling is ordered across layers, through the position in the
// pre-drawn uniform stream: layer l consumes uniforms [base_l, base_l + G_l) with base_l = the sum of the earlier sampling
// layers' list lengths (np.random.choice draws one double per class member, every member of the list belongs to exactly one
// class).  Each arbitrator publishes its G_l in sync->gword[l] and reads the earlier layers' words: at most R <= 4 workgroups
// ever wait, on workgroups that are running or will be dispatched whatever the waiters hold, and layer 0 waits for nobody.
__device__ void arbitrate_phase(const cim_mining_args& a, MiningSync* __restrict__ sync, unsigned char* smem, int l, int n_act,
                                const int16_t* __restrict__ s_act) {
#pragma clang fp contract(off)                      // the sums below restate NumPy's: no fused multiply-adds
    const int N = a.N, K = a.K, C = a.C, tid = threadIdx.x;
    // LDS: cdf f64 [K] | part i32 [1024] | prob f32 [K] | pos i32 [K] | gclass i32 [N] | gweight f32 [N] | mark u8 [N]
    double* cdf = reinterpret_cast<double*>(smem);
    int32_t* part = reinterpret_cast<int32_t*>(cdf + K);
    float* prob = reinterpret_cast<float*>(part + 1024);
    int32_t* pos = reinterpret_cast<int32_t*>(prob + K);
    int32_t* gclass = pos + K;
    float* gweight = reinterpret_cast<float*>(gclass + N);
    uint8_t* mark = reinterpret_cast<uint8_t*>(gweight + N);
    __shared__ int s_used;
    __shared__ float s_total;
    __shared__ int s_err;
    const int chunk = (N + 1023) / 1024;
    const int lo = min(N, tid * chunk), hi = min(N, lo + chunk);
    const cim_mining_layer& L = a.layer[l];

    MCLK(1, l, 0);
    if (tid == 0) s_err = 0;
    // (s_act[0 .. n_act): the image's classes, ascending - listed once at the start of the launch; the class loops below walk it)
    (void)C;
    // ---- arbitration (heads.py:397-405 / 306-314): the reference applies the image's classes one after the other in ascending
    // order, `w > gt_weight` (strict) deciding - per proposal that is: the largest weight among the classes that list it, the
    // LOWEST class on equal weights, and nothing unless w > -1 (heads.py:336).  All (class, candidate) pairs go through one
    // LDS atomic max on key = (orderable(w) << 32) | ~(c + 1) (a proposal listed twice by a class - torch.unique's set
    // semantics - just repeats its key); the initial key (orderable(-1), ~0) loses to any w > -1 and to nothing else.  NaN
    // weights never win a strict '>': skipped.  The key array lies over gclass | gweight and is decoded in place through
    // registers (N <= 8192: at most 8 keys per thread).
    unsigned long long* key = reinterpret_cast<unsigned long long*>(gclass);
    const unsigned long long key0 = ((unsigned long long)orderable(-1.0f) << 32) | 0xffffffffull;
    for (int i = tid; i < N; i += 1024) key[i] = key0;
    __syncthreads();
    MCLK(1, l, 1);
    const int32_t* __restrict__ cand = L.using_cim ? L.res : L.seeds;
    for (int e = tid; e < n_act * K; e += 1024) {
        const int c = s_act[e / K];
        const int32_t p = cand[(size_t)c * K + (e % K)];
        if (p < 0) continue;
        float w = L.wa[(size_t)p * L.wa_ld + L.wa_off + c];
        if (L.wb) w = w * L.wb[(size_t)p * L.wb_ld + L.wb_off + c * L.wb_cs];       // preds = cls * det, heads.py:330
        if (w == w) atomicMax(key + p, ((unsigned long long)orderable(w) << 32) | (unsigned)~(c + 1));
    }
    __syncthreads();
    MCLK(1, l, 2);
    {
        unsigned long long kreg[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = tid + u * 1024;
            kreg[u] = i < N ? key[i] : key0;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = tid + u * 1024;
            if (i < N) {
                const int gc = (int)~(unsigned)kreg[u];
                const unsigned o = (unsigned)(kreg[u] >> 32);
                const float gw = gc > 0 ? __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o) : -1.0f;   // orderable^-1
                gclass[i] = gc;
                gweight[i] = gw;
                L.gt_class[i] = gc;
                L.gt_weight[i] = gw;
            }
        }
    }
    __syncthreads();
    MCLK(1, l, 3);
    // ---- ordered compaction of {i : gclass[i] > 0} -> pre-sampling list (ascending proposal index).  Thread t owns the
    // proposals [lo, hi) and therefore the list positions [p0, p0 + cnt): every later pass walks the list through gclass /
    // gweight in LDS and these two numbers, never through the list in global memory.
    int cnt = 0;
    for (int i = lo; i < hi; ++i) cnt += gclass[i] > 0;
    int G;
    const int p0 = block_exclusive_scan(cnt, part, &G);
    // ---- this layer's share of the uniform stream, and where it starts
    const bool sampling = L.anti_noise && G > 0;
    MCLK(1, l, 4);
    if (tid == 0) {
        __hip_atomic_store(sync->gword + l, (1ull << 63) | (unsigned long long)(sampling ? G : 0), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int base = 0;
        for (int e = 0; e < l; ++e) {
            unsigned long long v;
            do {
                v = __hip_atomic_load(sync->gword + e, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                if (!(v >> 63)) __builtin_amdgcn_s_sleep(2);
            } while (!(v >> 63));
            base += (int)(v & 0xfffffull);
        }
        s_used = base;
    }
    MCLK(1, l, 5);
    uint8_t* __restrict__ keep = mark;                                   // keep[j], j = list position (mark[] is dead after the arbitration)
    for (int i = lo, j = p0; i < hi; ++i)
        if (gclass[i] > 0) {
            L.pre_idx[j] = i;
            keep[j] = 1;
            ++j;
        }
    __syncthreads();                                                     // keep[], s_used
    MCLK(1, l, 6);
    // ---- anti-noise sampling, class by class in ascending order (heads.py:451-466)
    if (sampling) {
        for (int ac = 0; ac < n_act; ++ac) {
            const int c = s_act[ac];
            // members of class c in list order
            int m = 0;
            for (int i = lo; i < hi; ++i) m += gclass[i] == c + 1;
            int Gc;
            int q0 = block_exclusive_scan(m, part, &Gc);
            if (Gc == 0) continue;                                       // heads.py:454-455 (uniform)
            if (Gc > K) {                                                // cannot happen (<= K candidates per class); stay memory-safe:
                if (tid == 0) {                                          // the step fails with status bit 1, the uniforms count as drawn
                    s_err |= 1;
                    s_used += Gc;
                }
                __syncthreads();
                continue;
            }
            // this class's uniforms: the load flies under the sums below
            const int ubase = s_used;
            const bool u_ok = ubase + Gc <= a.max_uniforms;
            const double u_first = (u_ok && tid < Gc) ? a.uniforms[ubase + tid] : 0.0;
            for (int i = lo, j = p0; i < hi; ++i) {
                const int gc = gclass[i];
                if (gc == c + 1) {
                    pos[q0] = j;
                    prob[q0] = gweight[i];
                    keep[j] = 0;                                         // inds[class_idx] = 0
                    ++q0;
                }
                j += gc > 0;
            }
            __syncthreads();
            if (tid == 0) s_total = np_pairwise_sum<4>(prob, Gc);
            __syncthreads();
            const float total = s_total;
            // p = prob / total in f32 (== the f64 quotient rounded once), then np.cumsum in f64: a strictly sequential sum
            // (out[0] = p[0]).  The divisions are done by everybody; the chain that is left for one lane is load / add / store,
            // eight loads ahead of the additions.
            for (int j = tid; j < Gc; j += 1024) cdf[j] = (double)(float)((double)prob[j] / (double)total);
            __syncthreads();
            if (tid == 0) {
                double acc = cdf[0];
                int j = 1;
                for (; j + 8 <= Gc; j += 8) {
                    double x[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) x[u] = cdf[j + u];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        acc += x[u];
                        cdf[j + u] = acc;
                    }
                }
                for (; j < Gc; ++j) {
                    acc += cdf[j];
                    cdf[j] = acc;
                }
            }
            __syncthreads();
            const double last = cdf[Gc - 1];
            const int base = ubase;
            if (!u_ok) {                                                 // the host drew too few (cannot happen: bound = R * min(N, classes * K))
                if (tid == 0) s_err |= 2;
            } else {
                for (int j = tid; j < Gc; j += 1024) {
                    const double u = j == tid ? u_first : a.uniforms[base + j];
                    int b = 0, e = Gc;                                   // searchsorted(cdf / cdf[-1], u, side='right'): first index with cdf > u
                    while (b < e) {
                        const int mid = (b + e) >> 1;
                        if (cdf[mid] / last <= u) b = mid + 1; else e = mid;
                    }
                    if (b >= Gc) b = Gc - 1;                             // u < 1 == cdf[-1]: unreachable; memory safety
                    keep[pos[b]] = 1;                                    // inds[np.unique(sampled)] = 1
                }
            }
            __syncthreads();
            if (tid == 0) s_used = base + Gc;
        }
    }
    MCLK(1, l, 7);
    // ---- survivors, in list order -> post-sampling list
    {
        int m = 0;
        for (int j = p0; j < p0 + cnt; ++j) m += keep[j];
        int Gk;
        int q0 = block_exclusive_scan(m, part, &Gk);
        for (int i = lo, j = p0; i < hi; ++i) {
            const int gc = gclass[i];
            if (gc > 0) {
                const uint8_t kp = keep[j];
                L.pre_keep[j] = kp;
                if (kp) {
                    L.gt_idx[q0] = i;
                    L.gt_cls[q0] = gc;
                    L.gt_w[q0] = gweight[i];
                    ++q0;
                }
                ++j;
            }
        }
        if (tid == 0) {
            L.counts[0] = G;
            L.counts[1] = Gk;
            a.layer_valid[l] = G > 0 ? 1 : 0;                            // heads.py:429-430: no pseudo GT -> layer skipped
            if (l == a.R - 1) a.used[0] = s_used;                        // (the last layer ends where the whole step's stream ends)
            // This arbitrator is done.  Its error bits travel in its word; the LAST of the R arbitrators to finish writes the
            // step's status word (the losses launch ORs its own bits into it later) and puts every sync word back to zero -
            // all R have read what they needed by then, and every seed workgroup has arrived (each layer's arbitrator was
            // elected by its last arrival): the scratch is as the caller handed it over.
            if (s_err) __hip_atomic_fetch_or(sync->gword + l, (unsigned long long)s_err << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned fin = __hip_atomic_fetch_add(&sync->done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (fin == (unsigned)a.R - 1u) {
                int bits = 0;
                for (int e = 0; e < a.R; ++e) {
                    bits |= (int)((__hip_atomic_load(sync->gword + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) & 0xffu);
                    __hip_atomic_store(sync->gword + e, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(sync->arrive + e, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __hip_atomic_store(&sync->done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *a.status = bits;
            }
        }
    }
    MCLK(1, l, 8);
}

// ================================================================== the mining launch
// grid = (C, R), block = 1024: workgroup (c, l) = class c of CIM layer l; inactive classes leave at once.  The last of a layer's
// active workgroups to finish its seeds goes on as the layer's arbitrator (with no class at all, workgroup (0, l) does).
__global__ __launch_bounds__(1024) void step_mine_kernel(const cim_mining_args a, MiningSync* __restrict__ sync, const SeedLayout lay) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int s_misc[8];
    __shared__ int s_part[16];
    __shared__ int16_t s_act[1024];                                      // the image's classes, ascending
    const int c = blockIdx.x, l = blockIdx.y, tid = threadIdx.x;
    const cim_mining_layer& L = a.layer[l];
    int n_act;
    {
        const int on = (tid < a.C && a.labels[tid] != 0.0f) ? 1 : 0;
        const int at = block_exclusive_scan(on, s_part, &n_act);
        if (on) s_act[at] = (int16_t)tid;
    }
    if (a.labels[c] == 0.0f) {                                           // heads.py:340: only the image's classes
        if (tid == 0) L.n_seeds[c] = 0;
        if (n_act != 0 || c != 0) return;
    } else {
        seed_phase(a, lay, smem, s_misc, s_part, c, l);
        // arrival: this workgroup's seeds / res are in memory (release), then one ticket on the layer's counter
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(sync->arrive + l, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            s_misc[4] = old == (unsigned)n_act - 1u;
        }
        __syncthreads();
        if (!s_misc[4]) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");              // the other classes' lists
    }
    arbitrate_phase(a, sync, smem, l, n_act, s_act);
}

// ================================================================== the assignment launch (heads.py:435,477-501)
// grid = (ceil(N/4), R): one wave per proposal row and layer: gather the G' surviving pseudo-GT columns of the fp16
// mask-IoU map, first-index arg-max, then the ignore / background / IoU-label rules.
__device__ __forceinline__ void assign_row(const uint16_t* __restrict__ r, int row, int lane, const int32_t* __restrict__ gt_idx,
                                           const int32_t* __restrict__ gt_cls, const float* __restrict__ gt_w, int G, int C1,
                                           float cls_thr, float iou_thr, float* __restrict__ pseudo_labels,
                                           uint16_t* __restrict__ pseudo_iou, float* __restrict__ loss_w,
                                           int32_t* __restrict__ max_idx) {
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

__global__ __launch_bounds__(256) void step_assign_kernel(const cim_mining_args a) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const cim_mining_layer& L = a.layer[blockIdx.y];
    if (row >= a.N) return;
    const int G = L.counts[1];
    if (G <= 0) return;                                                  // layer skipped: its outputs are never read
    assign_row(a.iou + (size_t)row * a.N, row, lane, L.gt_idx, L.gt_cls, L.gt_w, G, a.C + 1, L.cls_thr, L.iou_thr,
               L.pseudo_labels, L.pseudo_iou, L.loss_weights, L.max_idx);
}

long long arbitrate_lds_bytes(int N, int K) {
    return (long long)K * 8 + 1024 * 4 + (long long)K * 8 + (long long)N * 9 + 16;
}

SeedLayout seed_layout(int N, int K) {
    const long long KW = (K + 63) / 64;
    auto up16 = [](long long v) { return (v + 15) & ~15ll; };
    SeedLayout s{};
    long long o = 4096;                                                  // the four passes' 256-bin histograms
    s.off_cand = (int)o; o = up16(o + (long long)K * 8);
    s.off_sup = (int)o;  o = up16(o + (long long)K * KW * 8);
    s.off_kidx = (int)o; o = up16(o + (long long)K * 4);
    s.off_seed = (int)o; o = up16(o + (long long)K * 4);
    const long long det = up16((long long)N * 4) + up16(N);
    s.det_lds = (o + det <= 160 * 1024) ? 1 : 0;
    s.off_det = (int)o;
    s.off_flag = (int)(o + up16((long long)N * 4));
    if (s.det_lds) o += det;
    // whole-row streaming of the NMS matrix: <= 4 sixteen-byte pieces per lane (N <= 2040), <= 4 words per row (K <= 256)
    const long long pieces = (2ll * N + 14 + 15) / 16;
    s.row_halves = (int)(pieces * 8);
    s.off_row = (int)o;
    s.row_lds = (pieces <= 256 && KW <= 4 && o + 16 * pieces * 16 <= 156 * 1024) ? 1 : 0;
    if (s.row_lds) o += 16 * pieces * 16;
    const long long arb = arbitrate_lds_bytes(N, K);
    s.total = (int)(o > arb ? o : arb);
    return s;
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

extern "C" int cim_asy_prep(const uint16_t* asy_f16, int N, const float* con_thr_host, int n_slots, uint8_t* flags,
                            uint16_t* asy_t, void* stream) {
    CIM_CHECK_ARG(N >= 0 && n_slots >= 0 && n_slots <= CIM_MAX_LAYERS);
    if (N == 0) return 0;
    CIM_CHECK_ARG(asy_f16 && (n_slots == 0 || (con_thr_host && flags)));
    for (int s = 0; s < n_slots; ++s) {
        const int rc = cim_asy_flag(asy_f16, N, con_thr_host[s], flags + (size_t)s * N, stream);
        if (rc) return rc;
    }
    if (asy_t != nullptr) {
        const unsigned t = (unsigned)((N + 63) / 64);
        hipLaunchKernelGGL(transpose_f16_kernel, dim3(t, t), dim3(256), 0, cim::as_stream(stream), asy_f16, asy_t, N, (N + 7) & ~7);
    }
    CIM_CHECK_LAUNCH();
    return 0;
}

extern "C" long long cim_mining_lds_bytes(int N, int K) {
    if (N <= 0 || K <= 0) return 0;
    return seed_layout(N, K).total;
}

extern "C" long long cim_mining_sync_bytes(void) { return (long long)sizeof(MiningSync); }

extern "C" int cim_mining_step(const cim_mining_args* args, void* sync, void* stream) {
    CIM_CHECK_ARG(args != nullptr && sync != nullptr && (reinterpret_cast<uintptr_t>(sync) & 7) == 0);
    cim_mining_args a = *args;
    const int N = a.N, K = a.K, C = a.C, R = a.R;
    CIM_CHECK_ARG(N > 0 && N <= 8192 && K > 0 && K <= 1024 && K <= N && C > 0 && C <= 1024 && R >= 1 && R <= CIM_MAX_LAYERS);
    CIM_CHECK_ARG(a.labels && a.iou && a.uniforms && a.max_uniforms >= 0 && a.used && a.status && a.layer_valid);
    bool any_cim = false;
    for (int l = 0; l < R; ++l) {
        cim_mining_layer& L = a.layer[l];
        CIM_CHECK_ARG(L.seed_score && L.wa && L.topk && L.seeds && L.n_seeds && L.res && L.gt_class && L.gt_weight);
        CIM_CHECK_ARG(L.pre_idx && L.pre_keep && L.gt_idx && L.gt_cls && L.gt_w && L.counts);
        CIM_CHECK_ARG(L.pseudo_labels && L.pseudo_iou && L.loss_weights && L.max_idx);
        CIM_CHECK_ARG(!L.using_cim || (L.det && L.flag_slot >= 0 && L.flag_slot < CIM_MAX_LAYERS));
        any_cim |= L.using_cim != 0;
        // Python-float thresholds -> the binary16 value the reference's fp16 tensors are compared against
        L.nms_thr = cim::round_to_f16(L.nms_thr);
        L.cls_thr = cim::round_to_f16(L.cls_thr);
        L.iou_thr = cim::round_to_f16(L.iou_thr);
        L.con_thr = cim::round_to_f16(L.con_thr);
    }
    CIM_CHECK_ARG(!any_cim || ((a.asy || a.asy_t) && a.flags));          // MIST-only steps never touch the containment map
    hipStream_t st = cim::as_stream(stream);
    const SeedLayout lay = seed_layout(N, K);
    CIM_CHECK_ARG(lay.total <= 156 * 1024);               // (+ ~2.5 KB of static LDS)
    if (lay.total > 64 * 1024)
        CIM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(step_mine_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lay.total));
    // The layer arbitrators (at most R workgroups) wait for each other's list lengths inside the launch: no dispatch ORDER is
    // assumed - a waiting workgroup sleeps on the word while the others are placed wherever a CU is free.  Nothing per call is
    // baked into the arguments (the sync words are left zeroed), so the launches may be captured into a HIP graph and replayed.
    hipLaunchKernelGGL(step_mine_kernel, dim3(C, R), dim3(1024), (size_t)lay.total, st, a, static_cast<MiningSync*>(sync), lay);
    hipLaunchKernelGGL(step_assign_kernel, dim3((N + 3) / 4, R), dim3(256), 0, st, a);
    CIM_CHECK_LAUNCH();
    return 0;
}
