// Post-process on device: box recovery (reference dataset/base_sample.py:98-139) and multi-label class-aware NMS
// (reference tools.py:540-566 + torchvision.ops.boxes.batched_nms, restated -- see oracle/postprocess.py).
//
// NMS pipeline, one image, no host synchronisation:
//   1. mark    : every (box, class) score > thr becomes a candidate; key = score (non-candidates -inf), value = the
//                row-major element index; also count candidates and take the max coordinate over candidate boxes.
//   2. sort    : stable descending radix sort by score (rocPRIM device primitive) -> rank order = torchvision's
//                processing order (ties keep ascending candidate index).
//   3. bucket  : stable sort of ranks by class -> per-class lists in rank order (+ class histogram / offsets).
//   4. greedy  : one 256-thread workgroup per class walks its list 64 candidates at a time: all four waves test the
//                chunk against the boxes kept so far (kept list striped over waves), wave 0 resolves the 64x64
//                intra-chunk dependencies with ballot masks; arithmetic is torchvision's, bit for bit:
//                boxes + cls*(max+1), area = (x2-x1)*(y2-y1), suppress iff inter/(a_i+a_j-inter) > thr
//                (compiled with -ffp-contract=off so no FMA contraction changes a comparison).
//   5. compact : kept flags are compacted in rank order -> dets rows (x1,y1,x2,y2,score,class), count.
#include <cstring>
#include <string.h>
#include <rocprim/rocprim.hpp>
#include "common.h"

namespace pq {

__global__ __launch_bounds__(256) void recover_kernel(pq_recover_args a) {
    const int nv = 5 + a.C, no = 4 + a.C;
    const long rows = (long)a.B * a.N;
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < rows; r += (long)gridDim.x * 256) {
        const int b = (int)(r / a.N);
        const int bi = a.per_image ? b : 0;
        float* src = a.pred + (size_t)r * nv;
        float* dst = a.out + (size_t)r * no;
        const float dyv = a.delta[bi * 2 + 0], dxv = a.delta[bi * 2 + 1];
        const float ratio = a.ratio[bi];
        const float mh = a.orig[bi * 2 + 0] - 1.f, mw = a.orig[bi * 2 + 1] - 1.f;
        float x1 = (src[0] - dxv) / ratio, y1 = (src[1] - dyv) / ratio;
        float x2 = (src[2] - dxv) / ratio, y2 = (src[3] - dyv) / ratio;
        x1 = fmaxf(x1, 0.f); y1 = fmaxf(y1, 0.f);
        x2 = fminf(x2, mw); y2 = fminf(y2, mh);
        const float conf = src[4];
        dst[0] = x1; dst[1] = y1; dst[2] = x2; dst[3] = y2;
        if (a.mutate) { src[0] = x1; src[1] = y1; src[2] = x2; src[3] = y2; }
        for (int k = 0; k < a.C; ++k) {
            const float v = src[5 + k] * conf;
            dst[4 + k] = v;
            if (a.mutate) src[5 + k] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------ label assignment
// TrainDataset.create_label (dataset/train_dataset.py:109-150) for a whole batch.  One wave per image walks its ground
// truth boxes IN ORDER (later boxes overwrite earlier ones in a cell, and the per-scale box lists keep insertion order and
// duplicates), the 64 lanes cooperate on each (6+C)-float label row.  Arithmetic follows NumPy's promotions in the
// reference expression by expression: box corners / centre / size in float32, everything that meets the int64 strides or
// the anchor table (cell index, cell centre, anchor IoU, smoothed one-hot) in float64.
__global__ __launch_bounds__(256) void label_init_kernel(pq_create_label_args a) {
    const int nv = 6 + a.C;
    for (int s = 0; s < 3; ++s) {
        const long n = (long)a.B * a.H[s] * a.W[s] * a.gt_per_grid * nv;
        float* dst = a.label[s];
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = (i % nv == nv - 1) ? 1.f : 0.f;
        const long m = (long)a.B * a.cap * 4;
        float* bl = a.boxes[s];
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < m; i += (long)gridDim.x * 256) bl[i] = 0.f;
    }
}

__global__ __launch_bounds__(64) void label_assign_kernel(pq_create_label_args a) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int nv = 6 + a.C;
    const int n = min(a.counts ? a.counts[b] : a.G, a.G);
    int filled[3] = {0, 0, 0};
    for (int gi = 0; gi < n; ++gi) {
        const float* box = a.gt + ((size_t)b * a.G + gi) * 6;
        const float x1 = box[0], y1 = box[1], x2 = box[2], y2 = box[3];
        const int cls = (int)box[4];
        const float mix = box[5];
        const float cx = (x2 + x1) * 0.5f, cy = (y2 + y1) * 0.5f, bw = x2 - x1, bh = y2 - y1;     // float32, as in NumPy
        // lanes 0..8: IoU of the box with anchor k centred on its cell of scale k/3 (float64)
        double iou = -1.0;
        int cellx[3], celly[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            cellx[s] = (int)floor((double)cx / (double)a.stride[s]);
            celly[s] = (int)floor((double)cy / (double)a.stride[s]);
        }
        if (lane < 9) {
            const int s = lane / 3;
            const double acx = (double)((float)cellx[s] + 0.5f) * (double)a.stride[s];
            const double acy = (double)((float)celly[s] + 0.5f) * (double)a.stride[s];
            const double aw = (double)a.anchors[lane * 2 + 0], ah = (double)a.anchors[lane * 2 + 1];
            const float area1 = bw * bh;                                                               // float32 product
            const double area2 = aw * ah;
            const float l1 = cx - bw * 0.5f, t1 = cy - bh * 0.5f, r1 = cx + bw * 0.5f, b1 = cy + bh * 0.5f;
            const double l2 = acx - aw * 0.5, t2 = acy - ah * 0.5, r2 = acx + aw * 0.5, b2 = acy + ah * 0.5;
            const double iw = fmax(fmin((double)r1, r2) - fmax((double)l1, l2), 0.0);
            const double ih = fmax(fmin((double)b1, b2) - fmax((double)t1, t2), 0.0);
            const double inter = iw * ih;
            iou = inter / ((double)area1 + area2 - inter);
        }
        unsigned long long hit = __ballot(lane < 9 && iou > (double)a.iou_threshold) & 0x1ffull;
        if (hit == 0) {                                           // no anchor passes: the best one is taken (first maximum)
            double best = iou;
            int arg = lane;
#pragma unroll
            for (int off = 8; off >= 1; off >>= 1) {
                const double ob = __shfl_down(best, off);
                const int oa = __shfl_down(arg, off);
                const bool take = (lane + off < 9) && (lane < 9) && (ob > best || (ob == best && oa < arg));
                if (take) { best = ob; arg = oa; }
            }
            hit = 1ull << __shfl(arg, 0);
        }
        for (int k = 0; k < 9; ++k) {
            if (!((hit >> k) & 1ull)) continue;
            const int s = k / 3, slot = k % 3;
            const int gx = cellx[s], gy = celly[s];
            if (gx >= 0 && gx < a.W[s] && gy >= 0 && gy < a.H[s]) {
                float* row = a.label[s] + ((((size_t)b * a.H[s] + gy) * a.W[s] + gx) * a.gt_per_grid + slot) * nv;
                for (int e = lane; e < nv; e += 64) {
                    float v;
                    if (e < 4) v = box[e];
                    else if (e == 4) v = 1.f;
                    else if (e == nv - 1) v = mix;
                    else {
                        const double onehot = (e - 5 == cls) ? 1.0 : 0.0;
                        v = (float)(onehot * (1.0 - 0.01) + 0.01 * (1.0 / (double)a.C));
                    }
                    row[e] = v;
                }
            }
            if (filled[s] < a.cap && lane < 4) a.boxes[s][((size_t)b * a.cap + filled[s]) * 4 + lane] = box[lane];
            filled[s] += 1;
        }
    }
    if (lane < 3 && a.box_counts) a.box_counts[b * 3 + lane] = min(filled[lane], a.cap);
}

// ------------------------------------------------------------------------------------------------ NMS
struct NmsMeta { unsigned count; unsigned maxbits; unsigned kept; unsigned pad; unsigned greedy; unsigned r0, r1, r2; };

__device__ __forceinline__ unsigned f2ord(float f) { unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ord2f(unsigned o) { unsigned u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o; return __uint_as_float(u); }

// maxb[img] = largest coordinate of image img's candidates (the class offsets of batched_nms are per call, i.e. per image);
// nimg = boxes per image (== N for a single image, where the wave-reduced path below serves maxb[0])
__global__ __launch_bounds__(256) void nms_mark_kernel(const float* bboxes, int N, int C, float thr, float* keys, unsigned* vals,
                                                        NmsMeta* meta, unsigned* maxb, int nimg) {
    const long E = (long)N * C;
    const bool batch = nimg < N;
    unsigned cnt = 0, mx = 0;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < E; e += (long)gridDim.x * 256) {
        const int i = (int)(e / C), c = (int)(e % C);
        const float* row = bboxes + (size_t)i * (4 + C);
        const float s = row[4 + c];
        const bool hit = s > thr;
        keys[e] = hit ? s : -INFINITY;
        vals[e] = (unsigned)e;
        if (hit) {
            ++cnt;
            const float m = fmaxf(fmaxf(row[0], row[1]), fmaxf(row[2], row[3]));
            const unsigned o = f2ord(m);
            if (batch) atomicMax(&maxb[i / nimg], o);          // candidates are sparse
            else mx = o > mx ? o : mx;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        cnt += __shfl_xor(cnt, off);
        const unsigned o = __shfl_xor(mx, off);
        mx = o > mx ? o : mx;
    }
    if ((threadIdx.x & 63) == 0) {
        if (cnt) atomicAdd(&meta->count, cnt);
        if (mx) atomicMax(&maxb[0], mx);
    }
}

// ---- candidate compaction (cand_cap > 0): only the (box, class) pairs above the threshold enter the sorts.  Ordered: block b
// owns entries [b*4096, (b+1)*4096), thread t 16 consecutive ones, so the compacted list is in ascending pair index like the
// reference's nonzero() and the stable score sort breaks ties the same way as on the full list.
constexpr int NMS_CHUNK = 4096;

__global__ __launch_bounds__(256) void nms_fill_kernel(float* keys, unsigned* vals, int S) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < S; i += gridDim.x * 256) { keys[i] = -INFINITY; vals[i] = 0u; }
}

__global__ __launch_bounds__(256) void nms_count_kernel(const float* bboxes, int N, int C, float thr, unsigned* bcnt, NmsMeta* meta,
                                                         unsigned* maxb, int nimg) {
    __shared__ unsigned sc[4], sm[4];
    const long E = (long)N * C;
    const bool batch = nimg < N;
    const long base = (long)blockIdx.x * NMS_CHUNK + (long)threadIdx.x * 16;
    unsigned cnt = 0, mx = 0;
    for (int j = 0; j < 16; ++j) {
        const long e = base + j;
        if (e >= E) break;
        const int i = (int)(e / C), c = (int)(e % C);
        const float* row = bboxes + (size_t)i * (4 + C);
        if (row[4 + c] > thr) {
            ++cnt;
            const unsigned o = f2ord(fmaxf(fmaxf(row[0], row[1]), fmaxf(row[2], row[3])));
            if (batch) atomicMax(&maxb[i / nimg], o);
            else mx = o > mx ? o : mx;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        cnt += __shfl_xor(cnt, off);
        const unsigned o = __shfl_xor(mx, off);
        mx = o > mx ? o : mx;
    }
    if ((threadIdx.x & 63) == 0) { sc[threadIdx.x >> 6] = cnt; sm[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        bcnt[blockIdx.x] = sc[0] + sc[1] + sc[2] + sc[3];
        const unsigned m = max(max(sm[0], sm[1]), max(sm[2], sm[3]));
        if (m) atomicMax(&maxb[0], m);
    }
    (void)meta;
}

__global__ void nms_scan_kernel(unsigned* bcnt, int G, NmsMeta* meta, unsigned S) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        unsigned s = 0;
        for (int b = 0; b < G; ++b) { const unsigned o = bcnt[b]; bcnt[b] = s; s += o; }
        meta->pad = s > S ? 1u : 0u;              // more candidates than the caller's capacity: the result is discarded
        meta->count = s > S ? S : s;
    }
}

__global__ __launch_bounds__(256) void nms_write_kernel(const float* bboxes, int N, int C, float thr, const unsigned* boff, float* keys,
                                                         unsigned* vals, unsigned S) {
    __shared__ unsigned pre[256];
    const long E = (long)N * C;
    const long base = (long)blockIdx.x * NMS_CHUNK + (long)threadIdx.x * 16;
    unsigned cnt = 0;
    for (int j = 0; j < 16; ++j) {
        const long e = base + j;
        if (e >= E) break;
        if (bboxes[(size_t)(e / C) * (4 + C) + 4 + (e % C)] > thr) ++cnt;
    }
    pre[threadIdx.x] = cnt;
    __syncthreads();
    unsigned pos = boff[blockIdx.x];
    for (int q = 0; q < (int)threadIdx.x; ++q) pos += pre[q];          // 256 short adds: this kernel is not on any hot path
    if (cnt == 0) return;
    for (int j = 0; j < 16; ++j) {
        const long e = base + j;
        if (e >= E) break;
        const float sc = bboxes[(size_t)(e / C) * (4 + C) + 4 + (e % C)];
        if (sc > thr) {
            if (pos < S) { keys[pos] = sc; vals[pos] = (unsigned)e; }
            ++pos;
        }
    }
}

// class key per rank (0xFFFF for non-candidates) + histogram
__global__ __launch_bounds__(256) void nms_classkey_kernel(const unsigned* sorted_vals, long E, int C, const NmsMeta* meta,
                                                            unsigned* ckeys, unsigned* ranks, unsigned* hist, int nimg) {
    const unsigned K = meta->count;
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < E; r += (long)gridDim.x * 256) {
        ranks[r] = (unsigned)r;
        if (r < K) {
            const unsigned e = sorted_vals[r];
            const unsigned c = e % (unsigned)C + (unsigned)C * ((e / (unsigned)C) / (unsigned)nimg);     // class id, per image
            ckeys[r] = c;
            atomicAdd(&hist[c], 1u);
        } else {
            ckeys[r] = 0xFFFFu;
        }
    }
}

__global__ void nms_offsets_kernel(const unsigned* hist, int C, unsigned* offs) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        unsigned s = 0;
        for (int c = 0; c < C; ++c) { offs[c] = s; s += hist[c]; }
        offs[C] = s;
    }
}

// ---- both sorts of a call with a capacity of up to NMS_SMALL_MAX entries in ONE workgroup (no vendor library on this path): bitonic sort of
// 64-bit keys in LDS.  The radix sorts' results depend on their stability (ties in score keep ascending pair index, classes keep
// score order); here both orders are made TOTAL by putting the tie-breaker into the key -- (descending score, ascending pair
// index) and (class, score rank) --, so any correct sort gives the same, bit-identical permutation.  The per-class segment
// offsets are lower bounds in the class-sorted keys.  Replaces ~12 dependent launches (rocPRIM radix sort x 2, class keys,
// histogram scan) of a call that is launch-bound at these sizes.
constexpr int NMS_SMALL_MAX = 8192;      // = the default candidate capacity of ops.nms; measured against the library sorts (one
                                        // image, 16128 x 20 pairs): 100 / 1000 / 2000 candidates 0.063 / 0.092 / 0.123 ms against 0.103 / 0.120 /
                                        // 0.140, equal at ~4000, slower beyond (7000: 0.284 against 0.213): larger capacities keep the library

__device__ __forceinline__ void bitonic_sort_lds(unsigned long long* a, int P, int t, int nt) {
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            for (int i = t; i < (P >> 1); i += nt) {
                const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1)), hi = lo | j;     // the pair (lo, lo + j) of compare distance j
                const bool up = (lo & k) == 0;
                const unsigned long long x = a[lo], y = a[hi];
                if ((x > y) == up) { a[lo] = y; a[hi] = x; }
            }
        }
    __syncthreads();
}

__global__ __launch_bounds__(1024) void nms_sort_small_kernel(const float* keys0, const unsigned* vals0, int S, int P, int compact, int C, int NC,
                                                               int nimg, const NmsMeta* meta, float* keys1, unsigned* vals1, unsigned* rk1,
                                                               unsigned* offs) {
    extern __shared__ unsigned long long sk[];
    const int t = threadIdx.x, nt = blockDim.x;
    const unsigned K = meta->count;
    // after the ordered compaction the candidates are the first K entries (the rest is fill): only the next power of two takes
    // part in the sorts -- a trained detector leaves 10^2..10^3 candidates in a capacity of 8192
    int Q = P;
    if (compact) { Q = 64; while (Q < (int)K) Q <<= 1; if (Q > P) Q = P; }
    const int SQ = S < Q ? S : Q;
    // (descending score, ascending pair index): the ordered image of the score is inverted, entries beyond S sort last
    for (int i = t; i < Q; i += nt)
        sk[i] = i < S ? ((unsigned long long)(~f2ord(keys0[i])) << 32) | (unsigned long long)vals0[i] : ~0ull;
    bitonic_sort_lds(sk, Q, t, nt);
    for (int i = t; i < Q; i += nt) {
        const unsigned long long v = sk[i];
        unsigned long long k2 = ~0ull;
        if (i < S) {
            const unsigned e = (unsigned)v;
            keys1[i] = ord2f(~(unsigned)(v >> 32));
            vals1[i] = e;
            const unsigned c = (unsigned)i < K ? e % (unsigned)C + (unsigned)C * ((e / (unsigned)C) / (unsigned)nimg) : 0xFFFFu;
            k2 = ((unsigned long long)c << 32) | (unsigned long long)(unsigned)i;       // (class per image, score rank)
        }
        sk[i] = k2;               // (element i is read and rewritten by the same thread)
    }
    for (int i = Q + t; i < S; i += nt) { keys1[i] = keys0[i]; vals1[i] = vals0[i]; rk1[i] = (unsigned)i; }   // fill, already in place
    bitonic_sort_lds(sk, Q, t, nt);
    for (int i = t; i < SQ; i += nt) rk1[i] = (unsigned)sk[i];
    // offs[c] = first position whose class is >= c (c = 0 .. NC): lower bound over the sorted (class, rank) keys; beyond the
    // sorted part every entry is a non-candidate (class 0xFFFF)
    for (int c = t; c <= NC; c += nt) {
        const unsigned long long want = (unsigned long long)(unsigned)c << 32;
        int lo = 0, hi = SQ;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (sk[mid] < want) lo = mid + 1; else hi = mid; }
        offs[c] = (unsigned)lo;
    }
}

// The same for a BATCHED call whose images have at most NMS_SMALL_MAX candidates each: one workgroup per image sorts that image's
// segment of the (image-major) candidate list in its own LDS -- the suppression is per image anyway.  Positions stay global
// (rk1 holds indices into keys1 / vals1 of the whole call), so every later stage is unchanged; the kept rows then come grouped by
// image, in descending score order inside an image.  An image with more candidates than that poisons the call (count = -1, the
// caller's overflow path), its segment is passed through unsorted so that every index stays in range.
__global__ __launch_bounds__(1024) void nms_sort_seg_kernel(const float* keys0, const unsigned* vals0, int S, int C, int NC, int nimg,
                                                             NmsMeta* meta, float* keys1, unsigned* vals1, unsigned* rk1, unsigned* offs) {
    extern __shared__ unsigned long long sk[];
    __shared__ int s_lo, s_hi;
    const int t = threadIdx.x, nt = blockDim.x, b = blockIdx.x, B = gridDim.x;
    const int K = (int)meta->count;
    if (t < 2) {                                     // this image's candidates: pair indices in [b, b + 1) * nimg * C (vals0 ascending)
        const unsigned long long want = (unsigned long long)(b + t) * (unsigned long long)nimg * (unsigned long long)C;
        int lo = 0, hi = K;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if ((unsigned long long)vals0[mid] < want) lo = mid + 1; else hi = mid; }
        if (t == 0) s_lo = lo; else s_hi = lo;
    }
    __syncthreads();
    const int lo = s_lo, hi = s_hi, n = hi - lo;
    if (b == 0)
        for (int i = K + t; i < S; i += nt) { keys1[i] = keys0[i]; vals1[i] = vals0[i]; rk1[i] = (unsigned)i; }     // fill, already in place
    if (n > NMS_SMALL_MAX) {                         // uniform
        if (t == 0) atomicOr(&meta->pad, 1u);
        for (int i = lo + t; i < hi; i += nt) { keys1[i] = keys0[i]; vals1[i] = vals0[i]; rk1[i] = (unsigned)i; }
        for (int c = t; c < C; c += nt) offs[b * C + c] = (unsigned)lo;
        if (b == B - 1 && t == 0) offs[NC] = (unsigned)K;
        return;
    }
    int Q = 64;
    while (Q < n) Q <<= 1;
    for (int i = t; i < Q; i += nt)
        sk[i] = i < n ? ((unsigned long long)(~f2ord(keys0[lo + i])) << 32) | (unsigned long long)vals0[lo + i] : ~0ull;
    bitonic_sort_lds(sk, Q, t, nt);
    for (int i = t; i < Q; i += nt) {
        const unsigned long long v = sk[i];
        unsigned long long k2 = ~0ull;
        if (i < n) {
            const unsigned e = (unsigned)v;
            keys1[lo + i] = ord2f(~(unsigned)(v >> 32));
            vals1[lo + i] = e;
            k2 = ((unsigned long long)(e % (unsigned)C) << 32) | (unsigned long long)(unsigned)i;       // (class, score rank in the image)
        }
        sk[i] = k2;
    }
    bitonic_sort_lds(sk, Q, t, nt);
    for (int i = t; i < n; i += nt) rk1[lo + i] = (unsigned)lo + (unsigned)sk[i];
    for (int c = t; c < C; c += nt) {                // first position of class c of this image (global)
        const unsigned long long want = (unsigned long long)(unsigned)c << 32;
        int l2 = 0, h2 = n;
        while (l2 < h2) { const int mid = (l2 + h2) >> 1; if (sk[mid] < want) l2 = mid + 1; else h2 = mid; }
        offs[b * C + c] = (unsigned)(lo + l2);
    }
    if (b == B - 1 && t == 0) offs[NC] = (unsigned)K;
}

__device__ __forceinline__ bool nms_overlap(float ax1, float ay1, float ax2, float ay2, float aarea, float bx1, float by1, float bx2,
                                            float by2, float barea, float thr) {
    const float w = fmaxf(0.f, fminf(ax2, bx2) - fmaxf(ax1, bx1));
    const float h = fmaxf(0.f, fminf(ay2, by2) - fmaxf(ay1, by1));
    const float inter = w * h;
    const float ovr = inter / (aarea + barea - inter);
    return ovr > thr;
}

__global__ __launch_bounds__(256) void nms_greedy_kernel(const float* bboxes, int C, const unsigned* sorted_vals,
                                                          const unsigned* class_ranks, const unsigned* offs, const unsigned* maxb,
                                                          float thr, float* keptbuf, unsigned* keepflag, const NmsMeta* meta) {
    __shared__ unsigned sdead[4][64];
    __shared__ unsigned s_nk;
    const int c = blockIdx.x;
    const unsigned beg = offs[c], end = offs[c + 1];
    if (beg == end || meta->greedy != 1u) return;                           // 0: the bit-mask path below handles this call; 2: nothing does
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float off = (float)(c % C) * (ord2f(maxb[c / C]) + 1.f);          // blockIdx = image * C + class
    float* kept = keptbuf + (size_t)beg * 5;
    if (threadIdx.x == 0) s_nk = 0;
    __syncthreads();
    for (unsigned base = beg; base < end; base += 64) {
        const unsigned j = base + lane;
        const bool valid = j < end;
        unsigned rank = 0;
        float x1 = 0.f, y1 = 0.f, x2 = 0.f, y2 = 0.f, area = 0.f;
        if (valid) {
            rank = class_ranks[j];
            const unsigned e = sorted_vals[rank];
            const float* row = bboxes + (size_t)(e / (unsigned)C) * (4 + C);
            x1 = row[0] + off; y1 = row[1] + off; x2 = row[2] + off; y2 = row[3] + off;
            area = (x2 - x1) * (y2 - y1);
        }
        const unsigned nk = s_nk;
        unsigned dead = 0;
        for (unsigned k = w; k < nk; k += 4) {
            const float kx1 = __hip_atomic_load(&kept[k * 5 + 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float ky1 = __hip_atomic_load(&kept[k * 5 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float kx2 = __hip_atomic_load(&kept[k * 5 + 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float ky2 = __hip_atomic_load(&kept[k * 5 + 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float ka = __hip_atomic_load(&kept[k * 5 + 4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (nms_overlap(kx1, ky1, kx2, ky2, ka, x1, y1, x2, y2, area, thr)) dead = 1;
        }
        sdead[w][lane] = dead;
        __syncthreads();
        if (w == 0) {
            bool alive = valid && !(sdead[0][lane] | sdead[1][lane] | sdead[2][lane] | sdead[3][lane]);
            // which earlier lanes of this chunk would suppress me
            unsigned long long supp = 0ull;
            for (int i = 0; i < 64; ++i) {
                const float ix1 = __shfl(x1, i), iy1 = __shfl(y1, i), ix2 = __shfl(x2, i), iy2 = __shfl(y2, i), ia = __shfl(area, i);
                if (i < lane && nms_overlap(ix1, iy1, ix2, iy2, ia, x1, y1, x2, y2, area, thr)) supp |= (1ull << i);
            }
            unsigned long long am = __ballot(alive);
            for (int i = 0; i < 63; ++i) {
                if ((am >> i) & 1ull) {
                    if ((supp >> i) & 1ull) alive = false;
                    am = __ballot(alive);
                }
            }
            const unsigned long long below = (lane == 0) ? 0ull : (am & ((1ull << lane) - 1ull));
            if (alive) {
                const unsigned pos = nk + (unsigned)__popcll(below);
                __hip_atomic_store(&kept[pos * 5 + 0], x1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&kept[pos * 5 + 1], y1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&kept[pos * 5 + 2], x2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&kept[pos * 5 + 3], y2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&kept[pos * 5 + 4], area, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                keepflag[rank] = 1u;
            }
            if (lane == 0) s_nk = nk + (unsigned)__popcll(am);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
    }
}

// ---- greedy suppression through a suppression bit matrix (the long lists) ----------------------------------------------
// nms_greedy_kernel walks a class segment 64 candidates at a time against everything kept so far: one workgroup per class, so
// 100 k candidates in 20 classes keep 20 CUs busy for 5.7 ms (88 % of pq_nms, rocprofv3).  Here the pair tests are taken out
// of the sequential part: every (row block, column block >= row block) of a class computes a 64 x 64 tile of the relation
// "i (earlier, higher score) suppresses j" in parallel over the whole chip, one 64-bit word per row and tile; the sequential
// part that remains is bit arithmetic: walk the chunks in order, resolve a chunk from its diagonal word, OR the rows of the kept
// candidates into a `removed` bit vector in LDS.  Same order, same IoU expression (nms_overlap), same strict `>`: the selection
// is bit-identical to nms_greedy_kernel and to the CPU oracle.  Work list and matrix offsets are computed on the device (no
// host synchronisation); if the matrix would not fit the scratch the plan sets meta->greedy and the kernel above runs instead.
struct NmsMaskPlan { unsigned* tp; unsigned long long* wp; unsigned long long* mask; unsigned long long budget_words; };

__global__ void nms_mask_plan_kernel(const unsigned* offs, int NC, NmsMaskPlan pl, NmsMeta* meta) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (meta->pad) {                                        // candidate capacity exceeded: the result is discarded (count = -1), and the
        for (int c = 0; c <= NC; ++c) { pl.tp[c] = 0u; pl.wp[c] = 0ull; }      // segments may be unsorted pass-through: no suppression
        meta->greedy = 2u;                                  // stage touches them
        return;
    }
    unsigned long long words = 0;
    unsigned tiles = 0;
    bool fits = true;
    for (int c = 0; c < NC; ++c) {
        const unsigned n = offs[c + 1] - offs[c], nb = (n + 63u) >> 6;
        pl.tp[c] = tiles;
        pl.wp[c] = words;
        if (nb > 2048u) fits = false;                       // `removed` lives in 16 KiB of LDS
        tiles += nb * (nb + 1u) / 2u;
        words += (unsigned long long)n * nb;
        if (words > pl.budget_words || tiles > 0x7fffffffu) { fits = false; break; }
    }
    pl.tp[NC] = tiles;
    pl.wp[NC] = words;
    meta->greedy = fits ? 0u : 1u;
}

__device__ __forceinline__ void nms_load_box(const float* bboxes, int C, const unsigned* sorted_vals, const unsigned* class_ranks, unsigned j,
                                             float off, float& x1, float& y1, float& x2, float& y2, float& area) {
    const unsigned e = sorted_vals[class_ranks[j]];
    const float* row = bboxes + (size_t)(e / (unsigned)C) * (4 + C);
    x1 = row[0] + off; y1 = row[1] + off; x2 = row[2] + off; y2 = row[3] + off;
    area = (x2 - x1) * (y2 - y1);
}

__global__ __launch_bounds__(64) void nms_mask_kernel(const float* bboxes, int C, int NC, const unsigned* sorted_vals, const unsigned* class_ranks,
                                                      const unsigned* offs, const unsigned* maxb, float thr, NmsMaskPlan pl, const NmsMeta* meta) {
    if (meta->greedy) return;
    __shared__ float cb_box[5][64];
    const int lane = threadIdx.x;
    const unsigned T = pl.tp[NC];
    for (unsigned t = blockIdx.x; t < T; t += gridDim.x) {
        int lo = 0, hi = NC - 1;                            // class of tile t: last c with tp[c] <= t
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (pl.tp[mid] <= t) lo = mid; else hi = mid - 1; }
        const int c = lo;
        const unsigned beg = offs[c], n = offs[c + 1] - beg, nb = (n + 63u) >> 6;
        unsigned rem = t - pl.tp[c], rb = 0;
        while (rem >= nb - rb) { rem -= nb - rb; ++rb; }    // row block rb owns (nb - rb) tiles; nb <= 2048
        const unsigned cb = rb + rem;
        const float off = (float)(c % C) * (ord2f(maxb[c / C]) + 1.f);
        const unsigned i = rb * 64u + lane, j = cb * 64u + lane;
        float x1 = 0.f, y1 = 0.f, x2 = 0.f, y2 = 0.f, area = 0.f, jx1 = 0.f, jy1 = 0.f, jx2 = 0.f, jy2 = 0.f, ja = 0.f;
        if (i < n) nms_load_box(bboxes, C, sorted_vals, class_ranks, beg + i, off, x1, y1, x2, y2, area);
        if (j < n) nms_load_box(bboxes, C, sorted_vals, class_ranks, beg + j, off, jx1, jy1, jx2, jy2, ja);
        __syncthreads();                                    // previous tile's readers are done
        cb_box[0][lane] = jx1; cb_box[1][lane] = jy1; cb_box[2][lane] = jx2; cb_box[3][lane] = jy2; cb_box[4][lane] = ja;
        __syncthreads();
        unsigned long long word = 0ull;
        const unsigned jmax = min(64u, n - cb * 64u);
        for (unsigned q = 0; q < jmax; ++q) {
            if (cb * 64u + q > i && nms_overlap(x1, y1, x2, y2, area, cb_box[0][q], cb_box[1][q], cb_box[2][q], cb_box[3][q], cb_box[4][q], thr))
                word |= 1ull << q;
        }
        if (i < n) pl.mask[pl.wp[c] + (unsigned long long)i * nb + cb] = word;
    }
}

__global__ __launch_bounds__(256) void nms_mask_scan_kernel(const unsigned* class_ranks, const unsigned* offs, NmsMaskPlan pl, const NmsMeta* meta,
                                                            unsigned* keepflag) {
    if (meta->greedy) return;
    __shared__ unsigned long long removed[2048];
    __shared__ unsigned long long s_keep;
    const int c = blockIdx.x, t = threadIdx.x;
    const unsigned beg = offs[c], n = offs[c + 1] - beg, nb = (n + 63u) >> 6;
    if (n == 0) return;
    const unsigned long long* M = pl.mask + pl.wp[c];
    for (unsigned w = t; w < nb; w += 256) removed[w] = 0ull;
    __syncthreads();
    for (unsigned cb = 0; cb < nb; ++cb) {
        if (t < 64) {
            const unsigned i = cb * 64u + t;
            const bool valid = i < n;
            bool alive = valid && !((removed[cb] >> t) & 1ull);
            const unsigned long long diag = valid ? M[(unsigned long long)i * nb + cb] : 0ull;      // later candidates of this chunk I suppress
            unsigned long long am = __ballot(alive);
            unsigned long long todo = am;
            while (todo) {                                  // in order: a kept candidate removes the later ones it overlaps
                const int k = __ffsll((long long)todo) - 1;
                const unsigned long long dk = __shfl(diag, k);
                am &= ~dk;
                todo &= am;
                todo &= ~((2ull << k) - 1ull);
            }
            if (valid && ((am >> t) & 1ull)) keepflag[class_ranks[beg + i]] = 1u;
            if (t == 0) s_keep = am;
        }
        __syncthreads();
        const unsigned long long am = s_keep;
        for (unsigned w = cb + 1u + t; w < nb; w += 256) {
            unsigned long long acc = 0ull, todo = am;
            while (todo) {
                const int k = __ffsll((long long)todo) - 1;
                todo &= todo - 1ull;
                acc |= M[(unsigned long long)(cb * 64u + k) * nb + w];
            }
            removed[w] |= acc;
        }
        __syncthreads();
    }
}

// single workgroup: ordered compaction of keepflag[0..K) -> dets
__global__ __launch_bounds__(1024) void nms_compact_kernel(const float* bboxes, int C, const float* sorted_keys, const unsigned* sorted_vals,
                                                            const unsigned* keepflag, NmsMeta* meta, float* dets, int cap, int32_t* count,
                                                            int32_t* img, int nimg) {
    __shared__ unsigned wsum[16];
    __shared__ unsigned running;
    const unsigned K = meta->count;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) running = 0;
    __syncthreads();
    for (unsigned base = 0; base < K; base += 1024) {
        const unsigned r = base + threadIdx.x;
        const bool keep = r < K && keepflag[r] != 0u;
        const unsigned long long bal = __ballot(keep);
        const unsigned inwave = (unsigned)__popcll(bal & ((lane == 0) ? 0ull : ((1ull << lane) - 1ull)));
        if (lane == 0) wsum[w] = (unsigned)__popcll(bal);
        __syncthreads();
        unsigned before = running;
        for (int q = 0; q < w; ++q) before += wsum[q];
        if (keep) {
            const unsigned pos = before + inwave;
            if ((int)pos < cap) {
                const unsigned e = sorted_vals[r];
                const float* row = bboxes + (size_t)(e / (unsigned)C) * (4 + C);
                float* d = dets + (size_t)pos * 6;
                d[0] = row[0]; d[1] = row[1]; d[2] = row[2]; d[3] = row[3];
                d[4] = sorted_keys[r];
                d[5] = (float)(e % (unsigned)C);
                if (img) img[pos] = (int32_t)((e / (unsigned)C) / (unsigned)nimg);
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned tot = 0;
            for (int q = 0; q < 16; ++q) tot += wsum[q];
            running += tot;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        meta->kept = running;
        *count = meta->pad ? -1 : (int32_t)running;          // -1: cand_cap was too small, call again with a larger one / 0
    }
}

}  // namespace pq

using namespace pq;

extern "C" int pq_recover(const pq_recover_args* a, void* stream) {
    PQ_CHECK_ARG(a && a->pred && a->delta && a->ratio && a->orig && a->out, "null pointer");
    PQ_CHECK_ARG(a->B > 0 && a->N > 0 && a->C > 0, "bad shape");
    long rows = (long)a->B * a->N;
    long nb = (rows + 255) / 256; if (nb > 4096) nb = 4096;
    PQ_LAUNCH(recover_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, *a);
    PQ_CHECK_LAUNCH();
    return PQ_OK;
}

extern "C" int pq_create_label(const pq_create_label_args* a, void* stream) {
    PQ_CHECK_ARG(a && a->gt && a->anchors, "null pointer");
    PQ_CHECK_ARG(a->B > 0 && a->G > 0 && a->C > 0 && a->gt_per_grid == 3 && a->cap > 0, "bad shape (3 anchors per scale)");
    for (int s = 0; s < 3; ++s) PQ_CHECK_ARG(a->label[s] && a->boxes[s] && a->H[s] > 0 && a->W[s] > 0 && a->stride[s] > 0, "bad scale description");
    hipStream_t st = (hipStream_t)stream;
    PQ_LAUNCH(label_init_kernel, dim3(1024), dim3(256), 0, st, *a);
    PQ_CHECK_LAUNCH();
    PQ_LAUNCH(label_assign_kernel, dim3(a->B), dim3(64), 0, st, *a);
    PQ_CHECK_LAUNCH();
    return PQ_OK;
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct NmsLayout {
    size_t keys0, keys1, vals0, vals1, ckeys0, ckeys1, rk0, rk1, keep, kept, hist, offs, bcnt, meta, maxb, tp, wp, mask, mask_bytes, tmp, total;
};
// C = number of class segments (classes x images for a batched call), B = images
static NmsLayout nms_layout(long E, int C, int B = 1) {
    NmsLayout L;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += align256(bytes); return at; };
    L.meta = take(sizeof(NmsMeta));
    L.maxb = take((size_t)(B + 1) * 4);
    L.hist = take((size_t)(C + 1) * 4);
    L.offs = take((size_t)(C + 2) * 4);
    L.bcnt = take((size_t)((E + NMS_CHUNK - 1) / NMS_CHUNK + 2) * 4);
    L.keep = take((size_t)E * 4);
    L.keys0 = take((size_t)E * 4); L.keys1 = take((size_t)E * 4);
    L.vals0 = take((size_t)E * 4); L.vals1 = take((size_t)E * 4);
    L.ckeys0 = take((size_t)E * 4); L.ckeys1 = take((size_t)E * 4);
    L.rk0 = take((size_t)E * 4); L.rk1 = take((size_t)E * 4);
    L.kept = take((size_t)E * 5 * 4);
    L.tp = take((size_t)(C + 2) * 4);
    L.wp = take((size_t)(C + 2) * 8);
    // suppression bit matrix: n_c * ceil(n_c / 64) words per class; worst case one class holds everything; capped at 96 MiB
    // (beyond that the plan kernel falls back to nms_greedy_kernel)
    const size_t worst = (size_t)E * (size_t)((E + 63) / 64) * 8;
    L.mask_bytes = worst < ((size_t)96 << 20) ? worst : ((size_t)96 << 20);
    L.mask = take(L.mask_bytes);
    L.tmp = o;
    L.total = o;
    return L;
}

extern "C" int64_t pq_nms_scratch_bytes(int32_t N, int32_t C) {
    const long E = (long)N * C;
    NmsLayout L = nms_layout(E, C);
    // rocPRIM radix-sort temporary storage: double buffers for keys+values plus histograms; bounded generously
    return (int64_t)(L.total + (size_t)E * 24 + (8u << 20));
}

extern "C" int64_t pq_nms_batch_scratch_bytes(int32_t B, int32_t N, int32_t C) {
    const long E = (long)B * N * C;
    NmsLayout L = nms_layout(E, B * C, B);
    return (int64_t)(L.total + (size_t)E * 24 + (8u << 20));
}

static thread_local int g_nms_path = -1;
// which sort the calling thread's last pq_nms took: 0 = one workgroup for the whole call, 1 = one workgroup per image (batch),
// 2 = the library radix sorts (tests assert the path they mean to cover)
extern "C" int32_t pq_nms_last_path(void) { return g_nms_path; }

extern "C" int pq_nms(const pq_nms_args* a, void* stream) {
    PQ_CHECK_ARG(a && a->bboxes && a->dets && a->count && a->scratch, "null pointer");
    PQ_CHECK_ARG(a->N > 0 && a->C > 0 && a->C < 65535 && a->cap > 0, "bad shape");
    const int B = a->B > 1 ? a->B : 1;                 // images; bboxes is [B*N][4+C], classes are separated per image
    const int NC = B * a->C;                           // class segments
    PQ_CHECK_ARG(NC < 65535, "too many (image, class) segments");
    PQ_CHECK_ARG(B == 1 || a->img, "a batched call needs the img output");
    const int Ntot = B * a->N;
    const long E = (long)Ntot * a->C;
    PQ_CHECK_ARG(E < (1l << 31), "too many (box, class) pairs");
    hipStream_t st = (hipStream_t)stream;
    NmsLayout L = nms_layout(E, NC, B);
    PQ_CHECK_ARG((int64_t)L.total < a->scratch_bytes, "scratch too small");
    char* base = (char*)a->scratch;
    NmsMeta* meta = (NmsMeta*)(base + L.meta);
    unsigned* maxb = (unsigned*)(base + L.maxb);
    unsigned* hist = (unsigned*)(base + L.hist);
    unsigned* offs = (unsigned*)(base + L.offs);
    unsigned* keep = (unsigned*)(base + L.keep);
    float* keys0 = (float*)(base + L.keys0); float* keys1 = (float*)(base + L.keys1);
    unsigned* vals0 = (unsigned*)(base + L.vals0); unsigned* vals1 = (unsigned*)(base + L.vals1);
    unsigned* ck0 = (unsigned*)(base + L.ckeys0); unsigned* ck1 = (unsigned*)(base + L.ckeys1);
    unsigned* rk0 = (unsigned*)(base + L.rk0); unsigned* rk1 = (unsigned*)(base + L.rk1);
    float* kept = (float*)(base + L.kept);
    void* tmp = base + L.tmp;
    const size_t tmp_avail = (size_t)a->scratch_bytes - L.tmp;

    // zero meta + hist + offs + keep (contiguous prefix of the layout)
    if (hipMemsetAsync(base, 0, L.keys0, st) != hipSuccess) { pq_set_error("pq_nms: memset failed"); return PQ_ELAUNCH; }
    // S = how many entries go through the two sorts: everything, or the caller's candidate capacity after compaction
    const long S = (a->cand_cap > 0 && a->cand_cap < E) ? a->cand_cap : E;
    long nb = (S + 255) / 256; if (nb > 2048) nb = 2048;
    if (S == E) {
        long nbe = (E + 255) / 256; if (nbe > 2048) nbe = 2048;
        PQ_LAUNCH(nms_mark_kernel, dim3((unsigned)nbe), dim3(256), 0, st, a->bboxes, Ntot, a->C, a->score_thr, keys0, vals0, meta, maxb, a->N);
        PQ_CHECK_LAUNCH();
    } else {
        unsigned* bcnt = (unsigned*)(base + L.bcnt);
        const int G = (int)((E + NMS_CHUNK - 1) / NMS_CHUNK);
        PQ_LAUNCH(nms_fill_kernel, dim3((unsigned)nb), dim3(256), 0, st, keys0, vals0, (int)S);
        PQ_LAUNCH(nms_count_kernel, dim3(G), dim3(256), 0, st, a->bboxes, Ntot, a->C, a->score_thr, bcnt, meta, maxb, a->N);
        PQ_LAUNCH(nms_scan_kernel, dim3(1), dim3(64), 0, st, bcnt, G, meta, (unsigned)S);
        PQ_LAUNCH(nms_write_kernel, dim3(G), dim3(256), 0, st, a->bboxes, Ntot, a->C, a->score_thr, (const unsigned*)bcnt, keys0, vals0,
                           (unsigned)S);
        PQ_CHECK_LAUNCH();
    }
    static const char* lib_env = getenv("PQ_NMS_LIBSORT");            // "1": the library radix sorts at every size (A/B timing, tests)
    const bool libsort = lib_env && lib_env[0] == '1';
    const bool small = S <= NMS_SMALL_MAX && !libsort;
    // a batch whose candidate capacity is NMS_SMALL_MAX per image: one workgroup per image sorts its own segment
    const bool seg = !small && !libsort && B > 1 && S < E && S <= (long)B * NMS_SMALL_MAX;
    g_nms_path = small ? 0 : (seg ? 1 : 2);
    if (seg) {
        static bool raised2 = hipFuncSetAttribute((const void*)nms_sort_seg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                  NMS_SMALL_MAX * 8) == hipSuccess;
        if (!raised2) { pq_set_error("pq_nms: cannot raise the dynamic LDS limit"); return PQ_ELAUNCH; }
        PQ_LAUNCH(nms_sort_seg_kernel, dim3(B), dim3(1024), (size_t)NMS_SMALL_MAX * 8, st, (const float*)keys0, (const unsigned*)vals0, (int)S,
                           a->C, NC, a->N, meta, keys1, vals1, rk1, offs);
        PQ_CHECK_LAUNCH();
    } else if (small) {
        int P = 1024;                                                    // power of two >= S, >= the block size (uniform loops)
        while (P < S) P <<= 1;
        static bool raised = hipFuncSetAttribute((const void*)nms_sort_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 NMS_SMALL_MAX * 8) == hipSuccess;
        if (!raised) { pq_set_error("pq_nms: cannot raise the dynamic LDS limit"); return PQ_ELAUNCH; }
        PQ_LAUNCH(nms_sort_small_kernel, dim3(1), dim3(1024), (size_t)P * 8, st, (const float*)keys0, (const unsigned*)vals0, (int)S, P,
                           S == E ? 0 : 1, a->C, NC, a->N, (const NmsMeta*)meta, keys1, vals1, rk1, offs);
        PQ_CHECK_LAUNCH();
    } else {
    size_t need = 0;
    hipError_t e = rocprim::radix_sort_pairs_desc(nullptr, need, keys0, keys1, vals0, vals1, (size_t)S, 0, 32, st);
    if (e != hipSuccess || need > tmp_avail) { pq_set_error("pq_nms: sort scratch (%zu needed, %zu available)", need, tmp_avail); return PQ_EINVAL; }
    e = rocprim::radix_sort_pairs_desc(tmp, need, keys0, keys1, vals0, vals1, (size_t)S, 0, 32, st);
    if (e != hipSuccess) { pq_set_error("pq_nms: radix sort failed: %s", hipGetErrorString(e)); return PQ_ELAUNCH; }
    PQ_LAUNCH(nms_classkey_kernel, dim3((unsigned)nb), dim3(256), 0, st, (const unsigned*)vals1, S, a->C, (const NmsMeta*)meta, ck0, rk0, hist, a->N);
    PQ_CHECK_LAUNCH();
    PQ_LAUNCH(nms_offsets_kernel, dim3(1), dim3(64), 0, st, (const unsigned*)hist, NC, offs);
    PQ_CHECK_LAUNCH();
    size_t need2 = 0;
    e = rocprim::radix_sort_pairs(nullptr, need2, ck0, ck1, rk0, rk1, (size_t)S, 0, 16, st);
    if (e != hipSuccess || need2 > tmp_avail) { pq_set_error("pq_nms: class-sort scratch (%zu needed, %zu available)", need2, tmp_avail); return PQ_EINVAL; }
    e = rocprim::radix_sort_pairs(tmp, need2, ck0, ck1, rk0, rk1, (size_t)S, 0, 16, st);
    if (e != hipSuccess) { pq_set_error("pq_nms: class sort failed: %s", hipGetErrorString(e)); return PQ_ELAUNCH; }
    }
    NmsMaskPlan mp{(unsigned*)(base + L.tp), (unsigned long long*)(base + L.wp), (unsigned long long*)(base + L.mask),
                   (unsigned long long)(L.mask_bytes / 8)};
    static const char* greedy_env = getenv("PQ_NMS_GREEDY");          // "1": always the chunk-walking kernel (A/B timing, tests)
    if (greedy_env && greedy_env[0] == '1') mp.budget_words = 0;
    PQ_LAUNCH(nms_mask_plan_kernel, dim3(1), dim3(64), 0, st, (const unsigned*)offs, NC, mp, meta);
    PQ_LAUNCH(nms_mask_kernel, dim3(8192), dim3(64), 0, st, a->bboxes, a->C, NC, (const unsigned*)vals1, (const unsigned*)rk1,
                       (const unsigned*)offs, (const unsigned*)maxb, a->iou_thr, mp, (const NmsMeta*)meta);
    PQ_LAUNCH(nms_mask_scan_kernel, dim3(NC), dim3(256), 0, st, (const unsigned*)rk1, (const unsigned*)offs, mp, (const NmsMeta*)meta, keep);
    PQ_CHECK_LAUNCH();
    PQ_LAUNCH(nms_greedy_kernel, dim3(NC), dim3(256), 0, st, a->bboxes, a->C, (const unsigned*)vals1, (const unsigned*)rk1,
                       (const unsigned*)offs, (const unsigned*)maxb, a->iou_thr, kept, keep, (const NmsMeta*)meta);
    PQ_CHECK_LAUNCH();
    PQ_LAUNCH(nms_compact_kernel, dim3(1), dim3(1024), 0, st, a->bboxes, a->C, (const float*)keys1, (const unsigned*)vals1,
                       (const unsigned*)keep, meta, a->dets, a->cap, a->count, a->img, a->N);
    PQ_CHECK_LAUNCH();
    return PQ_OK;
}
