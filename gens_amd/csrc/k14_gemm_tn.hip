// K14: C (M x N) = A^T B for tall operands A (K x M), B (K x N), K >> M, N -- the weight-gradient GEMM of the training step
// (dW = dY^T X with K = number of points: 61 835 x {128, 101} x {188, 27}; 247 340 x {64, 33, 32, 23, 16, 8} x {69, 64, 37, 32, 16}).
// hipBLASLt runs these without splitting K: 160 us for 128 x 61 835 x 188 (18 TFLOP/s) and 0.5 ms for 32 x 247 340 x 32
// (1 TFLOP/s, a 64 MB read) -- 8 ms of a 44 ms training step.
//
// Mapping.  K is cut into slabs; a unit of work is one (slab, 32 x 32 tile of C) pair and belongs to ONE wave (four units per
// workgroup), the slab length chosen so that the launch has ~8 000 units whatever the shape.  For v_mfma_f32_32x32x2_f32 (exact float32)
// lane l supplies A[k + l / 32][m0 + l % 32] and B[k + l / 32][n0 + l % 32]: both operands are read ALONG the rows of A and B, 128
// contiguous bytes per half wave, straight from global memory (the re-reads of a row by the other tiles hit L1 / L2) -- no
// transposition, no LDS.  Every unit writes its partial tile into a workspace; a second kernel adds the partials in slab order, so the
// result does not depend on scheduling (no atomics).
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void gemm_tn_partial_k(const float* __restrict__ a, const float* __restrict__ b, int64_t kk, int m, int n,
                                                         int64_t slab, int64_t n_units, float* __restrict__ ws) {
    const int lane = threadIdx.x & 63;
    const int i = lane & 31, h = lane >> 5;
    const int mt = (m + 31) >> 5, nt = (n + 31) >> 5;
    const int64_t unit = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (unit >= n_units) return;
    const int64_t s = unit / (mt * nt);
    const int t = (int)(unit % (mt * nt));
    const int64_t k_begin = min(kk, s * slab), k_end = min(kk, k_begin + slab);
    const int m0 = (t / nt) << 5, n0 = (t % nt) << 5;
    const bool am = m0 + i < m, bn = n0 + i < n;
    const float* ap = a + (k_begin + h) * m + (am ? m0 + i : 0);
    const float* bp = b + (k_begin + h) * n + (bn ? n0 + i : 0);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    int64_t k = k_begin;
    for (; k + 16 <= k_end; k += 16) {                        // eight MFMAs (16 rows of K) per trip, their sixteen loads issued together
        float av[8], bv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            av[q] = ap[(int64_t)(2 * q) * m];
            bv[q] = bp[(int64_t)(2 * q) * n];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(am ? av[q] : 0.0f, bn ? bv[q] : 0.0f, acc, 0, 0, 0);
        ap += 16 * (int64_t)m;
        bp += 16 * (int64_t)n;
    }
    for (; k < k_end; k += 2) {                               // tail: the second row of a pair may lie beyond the slab
        const bool row = k + h < k_end;
        const float av = (am && row) ? ap[0] : 0.0f, bv = (bn && row) ? bp[0] : 0.0f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        ap += 2 * (int64_t)m;
        bp += 2 * (int64_t)n;
    }
    if (bn) {
        float* w = ws + s * (int64_t)m * n;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row < m) w[(int64_t)row * n + n0 + i] = acc[r];
        }
    }
}

// partial sums in fixed order: out[g][e] = sum of in[q][e] over the g-th group of `per` consecutive slabs (coalesced over e)
__global__ __launch_bounds__(256) void gemm_tn_reduce_k(const float* __restrict__ in, int n_in, int per, int64_t mn, float* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= mn) return;
    const int q0 = blockIdx.y * per, q1 = min(n_in, q0 + per);
    float s = 0.0f;
    for (int q = q0; q < q1; ++q) s += in[(int64_t)q * mn + e];
    out[(int64_t)blockIdx.y * mn + e] = s;
}

#define GEMM_TN_GROUPS 64     // second-stage groups: the workspace holds (slabs + GEMM_TN_GROUPS) partial results

static int64_t gemm_tn_slab(int64_t k, int m, int n) {      // rows of K per unit: ~8 000 units per launch, multiples of 16 rows, at least 64
    const int64_t tiles = (int64_t)((m + 31) / 32) * ((n + 31) / 32);
    int64_t slab = (k * tiles + 8191) / 8192;
    slab = (slab + 15) / 16 * 16;
    return slab < 64 ? 64 : slab;
}

extern "C" int gens_gemm_tn_slabs(int64_t k, int m, int n) {
    if (k <= 0 || m <= 0 || n <= 0) return 0;
    const int64_t slab = gemm_tn_slab(k, m, n);
    return (int)((k + slab - 1) / slab) + GEMM_TN_GROUPS;     // slabs + room for the second reduction stage
}

extern "C" int gens_gemm_tn(const float* a, const float* b, int64_t k, int m, int n, float* workspace, float* c, void* stream) {
    GENS_CHECK_ARG(a && b && workspace && c, GENS_EINVAL, "gens_gemm_tn: null pointer");
    GENS_CHECK_ARG(k > 0 && m > 0 && n > 0 && m <= 1024 && n <= 1024, GENS_ELIMIT, "gens_gemm_tn: k=%lld m=%d n=%d (1 <= m, n <= 1024)", (long long)k, m, n);
    const int64_t slab = gemm_tn_slab(k, m, n);
    const int n_slabs = (int)((k + slab - 1) / slab);
    const int64_t units = (int64_t)n_slabs * ((m + 31) / 32) * ((n + 31) / 32);
    hipStream_t s = (hipStream_t)stream;
    gemm_tn_partial_k<<<gens_blocks(units, 4), 256, 0, s>>>(a, b, k, m, n, slab, units, workspace);
    const int64_t mn = (int64_t)m * n;
    if (n_slabs <= GEMM_TN_GROUPS) {
        gemm_tn_reduce_k<<<dim3(gens_blocks(mn, 256), 1), 256, 0, s>>>(workspace, n_slabs, n_slabs, mn, c);
    } else {                                                                       // two stages, both in slab order
        const int per = (n_slabs + GEMM_TN_GROUPS - 1) / GEMM_TN_GROUPS, groups = (n_slabs + per - 1) / per;
        float* stage = workspace + (int64_t)n_slabs * mn;
        gemm_tn_reduce_k<<<dim3(gens_blocks(mn, 256), groups), 256, 0, s>>>(workspace, n_slabs, per, mn, stage);
        gemm_tn_reduce_k<<<dim3(gens_blocks(mn, 256), 1), 256, 0, s>>>(stage, groups, groups, mn, c);
    }
    return gens_launch_status("gens_gemm_tn");
}

// ---------------------------------------------------------------------------------------------------------------------------
// Batched, strided form: up to GEMM_TN_MAX_BATCH (12) products C_p (m_p x n_p) = A_p^T B_p over the SAME K rows in ONE launch, operands
// given with leading dimensions (column blocks of wider row-major buffers: no copies).  The weight-gradient products of the fused
// training-mode SDF network (K17: seven products over 4 x points rows).  A unit = (slab, tile of one product); consecutive units
// are the tiles of one slab, so the rows a slab's tiles share are served by L2.  Results: the C_p concatenated in one vector.
// ---------------------------------------------------------------------------------------------------------------------------
#define GEMM_TN_MAX_BATCH 12
struct GemmTnBatch {
    const float* a[GEMM_TN_MAX_BATCH];
    const float* b[GEMM_TN_MAX_BATCH];
    int lda[GEMM_TN_MAX_BATCH], ldb[GEMM_TN_MAX_BATCH], m[GEMM_TN_MAX_BATCH], n[GEMM_TN_MAX_BATCH];
    int tile0[GEMM_TN_MAX_BATCH + 1];      // first tile of product p in the per-slab tile list
    int64_t off[GEMM_TN_MAX_BATCH];        // offset of C_p in the concatenated result
    int count;
    int64_t csz;
    // rows that exist: min(k, k_rows * ceil(*k_live / k_div)) when k_live != NULL -- a DEVICE count of the producer's work items (points of a
    // masked evaluation), k_div of which share a workgroup that wrote k_rows operand rows; slabs past them add zero tiles
    const int32_t* k_live;
    int k_div, k_rows;
};
__device__ __forceinline__ int64_t gemm_tn_live_rows(const GemmTnBatch& B, int64_t kk) {
    if (!B.k_live) return kk;
    const int64_t groups = ((int64_t)B.k_live[0] + B.k_div - 1) / B.k_div;
    return min(kk, max((int64_t)0, groups) * B.k_rows);
}

// Workgroup -> (slab, four tiles of it), XCD-aware.  The tiles of ONE slab read the same K rows of the same operands (the seven products of the SDF
// network's weight gradients share `lop`: every row is read by up to six tiles), and workgroup b runs on XCD b % 8, each with its own L2: numbered
// slab after slab, a slab's eight workgroups sat on eight different XCDs and every L2 fetched the rows for itself -- 3.9 GB over the fabric for
// 1.6 GB of operands, the launch's whole 0.94 ms.  Here the workgroups of a slab are eight apart: one XCD, one fetch.
// n_wg workgroups per slab; slabs in octets (one slab per XCD); -> false when this workgroup / wave has no unit.
__device__ __forceinline__ bool gemm_tn_batch_unit(int n_wg, int64_t n_slabs, int tiles, int64_t& s, int& t) {
    const int64_t per_octet = 8 * (int64_t)n_wg, octet = blockIdx.x / per_octet;
    const int in_octet = (int)(blockIdx.x - octet * per_octet);
    s = 8 * octet + (in_octet & 7);
    t = 4 * (in_octet >> 3) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // (the wave's number as a SCALAR: everything derived from the unit -- product, strides, descriptors -- stays in SGPRs)
    return s < n_slabs && t < tiles;
}

__global__ __launch_bounds__(256) void gemm_tn_batch_partial_k(GemmTnBatch B, int64_t kk, int64_t slab, int64_t n_slabs, int n_wg, float* __restrict__ ws) {
    kk = gemm_tn_live_rows(B, kk);
    const int lane = threadIdx.x & 63;
    const int i = lane & 31, h = lane >> 5;
    const int tiles = B.tile0[B.count];
    int64_t s;
    int t;
    if (!gemm_tn_batch_unit(n_wg, n_slabs, tiles, s, t)) return;
    int p = 0;
    while (p + 1 < B.count && t >= B.tile0[p + 1]) ++p;
    t -= B.tile0[p];
    const int m = B.m[p], n = B.n[p], lda = B.lda[p], ldb = B.ldb[p];
    const int nt = (n + 31) >> 5;
    const int64_t k_begin = min(kk, s * slab), k_end = min(kk, k_begin + slab);
    const int m0 = (t / nt) << 5, n0 = (t % nt) << 5;
    const bool am = m0 + i < m, bn = n0 + i < n;
    const float* ap = B.a[p] + (k_begin + h) * lda + (am ? m0 + i : 0);
    const float* bp = B.b[p] + (k_begin + h) * ldb + (bn ? n0 + i : 0);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    int64_t k = k_begin;
    for (; k + 16 <= k_end; k += 16) {
        float av[8], bv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            av[q] = ap[(int64_t)(2 * q) * lda];
            bv[q] = bp[(int64_t)(2 * q) * ldb];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(am ? av[q] : 0.0f, bn ? bv[q] : 0.0f, acc, 0, 0, 0);
        ap += 16 * (int64_t)lda;
        bp += 16 * (int64_t)ldb;
    }
    for (; k < k_end; k += 2) {
        const bool row = k + h < k_end;
        const float av = (am && row) ? ap[0] : 0.0f, bv = (bn && row) ? bp[0] : 0.0f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        ap += 2 * (int64_t)lda;
        bp += 2 * (int64_t)ldb;
    }
    if (bn) {
        float* w = ws + s * B.csz + B.off[p];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row < m) w[(int64_t)row * n + n0 + i] = acc[r];
        }
    }
}

// Register-blocked form of the batched kernel for EVEN m, n, lda, ldb (8-byte aligned operands): a unit is a 64 x 64 block of C = 2 x 2
// MFMA tiles whose rows / columns INTERLEAVE (tile a holds rows m0 + 2 i + a), so one 8-byte load per operand and lane feeds four
// v_mfma_f32_32x32x2_f32 -- 0.5 load instructions per MFMA instead of 2 (the 32 x 32 kernel is bound by load issue: 1.7 TB/s).
__global__ __launch_bounds__(256) void gemm_tn_batch2_partial_k(GemmTnBatch B, int64_t kk, int64_t slab, int64_t n_slabs, int n_wg, float* __restrict__ ws) {
    kk = gemm_tn_live_rows(B, kk);
    const int lane = threadIdx.x & 63;
    const int i = lane & 31, h = lane >> 5;
    const int tiles = B.tile0[B.count];
    int64_t s;
    int t;
    if (!gemm_tn_batch_unit(n_wg, n_slabs, tiles, s, t)) return;
    int p = 0;
    while (p + 1 < B.count && t >= B.tile0[p + 1]) ++p;
    t -= B.tile0[p];
    const int m = B.m[p], n = B.n[p], lda = B.lda[p], ldb = B.ldb[p];
    const int nt = (n + 63) >> 6;
    const int64_t k_begin = min(kk, s * slab), k_end = min(kk, k_begin + slab);
    const int m0 = (t / nt) << 6, n0 = (t % nt) << 6;
    const bool am = m0 + 2 * i < m, bn = n0 + 2 * i < n;               // (m, n even: a pair is inside or outside as a whole)
    const float2* ap = (const float2*)(B.a[p] + (k_begin + h) * lda + (am ? m0 + 2 * i : 0));
    const float2* bp = (const float2*)(B.b[p] + (k_begin + h) * ldb + (bn ? n0 + 2 * i : 0));
    const int64_t sa = lda, sb = ldb;                                   // row strides in floats = 2 x (stride in float2) per 2 rows
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    int64_t k = k_begin;
    // four row pairs per trip, their eight loads issued together -- and ONE TRIP AHEAD of the products that use them (two register sets): a trip's
    // sixteen MFMAs are 1 024 cycles of work, a load under this traffic takes longer, and the waves of a SIMD could not cover the difference
    // (matrix pipe 57 % busy, waves waiting 76 % of their time with the loads issued at the top of their own trip).
    // Round 5: the loads go through buffer descriptors -- a lane's byte offset is ONE register for the whole slab, the row of a load is a scalar
    // offset -- and the operands of rows / columns outside the matrix are no longer zeroed per product (such a lane reads column 0 and its outer-product
    // rows / columns of the accumulators are never stored): 179 -> <= 128 registers, four waves per SIMD instead of two.
    typedef float k14_f2 __attribute__((ext_vector_type(2)));
    const uint32_t slab_rows = (uint32_t)(k_end - k_begin);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(B.a[p] + k_begin * lda), 0, (int)(slab_rows * (uint32_t)lda * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(B.b[p] + k_begin * ldb), 0, (int)(slab_rows * (uint32_t)ldb * 4u), 0x00020000);
    const uint32_t va = ((uint32_t)h * (uint32_t)lda + (uint32_t)(am ? m0 + 2 * i : 0)) * 4u, vb = ((uint32_t)h * (uint32_t)ldb + (uint32_t)(bn ? n0 + 2 * i : 0)) * 4u;
    const uint32_t pa = 8u * (uint32_t)lda, pb = 8u * (uint32_t)ldb;           // bytes from a row pair to the next
    uint32_t oa = 0u, ob = 0u;                                                  // (scalar) byte offset of the next row pair to load
#define K14_LOAD(AV, BV)                                                                                       \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                            \
        AV[q] = __builtin_bit_cast(k14_f2, __builtin_amdgcn_raw_buffer_load_b64(ra, va, oa + (uint32_t)q * pa, 0));   \
        BV[q] = __builtin_bit_cast(k14_f2, __builtin_amdgcn_raw_buffer_load_b64(rb, vb, ob + (uint32_t)q * pb, 0));   \
    }                                                                                                          \
    oa += 4u * pa;                                                                                             \
    ob += 4u * pb;
#define K14_MULT(AV, BV)                                                                                                                 \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                                                      \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV[q].x, BV[q].x, acc[0][0], 0, 0, 0);                                          \
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV[q].x, BV[q].y, acc[0][1], 0, 0, 0);                                          \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV[q].y, BV[q].x, acc[1][0], 0, 0, 0);                                          \
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV[q].y, BV[q].y, acc[1][1], 0, 0, 0);                                          \
    }
    // Three register sets, a set loaded TWO products (32 MFMAs, 2 048 cycles) before it is multiplied; every LOAD / MULT pinned in place (the
    // scheduler otherwise pulls the later loads to the top of the trip and the counter wait of the second product then covers them too -- an
    // effective distance of one product, 1 024 cycles, less than an L2 round trip under this traffic).
#define K14_STEP(LD, MU)                      \
    K14_LOAD(a##LD, b##LD)                    \
    __builtin_amdgcn_sched_barrier(0);        \
    K14_MULT(a##MU, b##MU)                    \
    __builtin_amdgcn_sched_barrier(0);
    {
        const int n8 = (int)((k_end - k_begin) >> 3);                  // whole 8-row steps of the slab
        k14_f2 a0[4], b0[4], a1[4], b1[4], a2[4], b2[4];
        if (n8 >= 1) { K14_LOAD(a0, b0) }
        if (n8 >= 2) { K14_LOAD(a1, b1) }
        int step = 0;
        for (; step + 5 <= n8; step += 3) {                            // sets 0 and 1 hold steps `step` and `step + 1`
            K14_STEP(2, 0)
            K14_STEP(0, 1)
            K14_STEP(1, 2)
        }
        const int r = n8 - step;                                       // 0 .. 4 steps left, the first two of them loaded
        if (r >= 3) { K14_LOAD(a2, b2) }
        __builtin_amdgcn_sched_barrier(0);
        if (r >= 1) { K14_MULT(a0, b0) }
        if (r == 4) { K14_LOAD(a0, b0) }
        __builtin_amdgcn_sched_barrier(0);
        if (r >= 2) { K14_MULT(a1, b1) }
        if (r >= 3) { K14_MULT(a2, b2) }
        if (r == 4) { K14_MULT(a0, b0) }
        k = k_begin + 8 * (int64_t)n8;
    }
#undef K14_STEP
#undef K14_LOAD
#undef K14_MULT
    ap += (k - k_begin) / 2 * sa;                                       // (the tail walks pointers: row pairs, sa float2 = 2 rows)
    bp += (k - k_begin) / 2 * sb;
    for (; k < k_end; k += 2) {
        const bool row = k + h < k_end;
        float2 av = make_float2(0.f, 0.f), bv = make_float2(0.f, 0.f);
        if (am && row) av = ap[0];
        if (bn && row) bv = bp[0];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.y, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.x, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[1][1], 0, 0, 0);
        ap += sa;
        bp += sb;
    }
    if (bn) {
        float* w = ws + s * B.csz + B.off[p];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * h) + a;
                if (row < m) *(float2*)(w + (int64_t)row * n + n0 + 2 * i) = make_float2(acc[a][0][r], acc[a][1][r]);
            }
    }
}

static int64_t gemm_tn_batch_slab(int64_t k, int64_t tiles) {
    int64_t slab = (k * tiles + 8191) / 8192;
    slab = (slab + 15) / 16 * 16;
    return slab < 64 ? 64 : slab;
}

static int gemm_tn_batch_layout(int count, const int* m, const int* n, int64_t* tiles, int64_t* csz, int tile = 32) {
    *tiles = 0;
    *csz = 0;
    if (count <= 0 || count > GEMM_TN_MAX_BATCH || !m || !n) return -1;
    for (int p = 0; p < count; ++p) {
        if (m[p] <= 0 || n[p] <= 0 || m[p] > 1024 || n[p] > 1024) return -1;
        *tiles += (int64_t)((m[p] + tile - 1) / tile) * ((n[p] + tile - 1) / tile);
        *csz += (int64_t)m[p] * n[p];
    }
    return 0;
}

// floats of workspace a gens_gemm_tn_batch call needs
extern "C" int64_t gens_gemm_tn_batch_workspace(int count, const int* m, const int* n, int64_t k) {
    int64_t tiles, tiles64, csz;
    if (k <= 0 || gemm_tn_batch_layout(count, m, n, &tiles, &csz) || gemm_tn_batch_layout(count, m, n, &tiles64, &csz, 64)) return 0;
    const int64_t slab = gemm_tn_batch_slab(k, tiles), slab64 = gemm_tn_batch_slab(k, tiles64);      // either kernel may run: room for both
    const int64_t slabs = (k + slab - 1) / slab, slabs64 = (k + slab64 - 1) / slab64;
    return ((slabs > slabs64 ? slabs : slabs64) + GEMM_TN_GROUPS) * csz;
}

static int gemm_tn_batch_run(int count, const float* const* a, const int* lda, const float* const* b, const int* ldb, const int* m,
                             const int* n, int64_t k, const int32_t* k_live, int k_div, int k_rows, float* workspace, float* c, void* stream);

extern "C" int gens_gemm_tn_batch(int count, const float* const* a, const int* lda, const float* const* b, const int* ldb, const int* m,
                                  const int* n, int64_t k, float* workspace, float* c, void* stream) {
    return gemm_tn_batch_run(count, a, lda, b, ldb, m, n, k, nullptr, 1, 1, workspace, c, stream);
}

extern "C" int gens_gemm_tn_batch_live(int count, const float* const* a, const int* lda, const float* const* b, const int* ldb, const int* m,
                                       const int* n, int64_t k, const int32_t* k_live, int k_div, int k_rows, float* workspace, float* c,
                                       void* stream) {
    GENS_CHECK_ARG(k_live && k_div >= 1 && k_rows >= 1, GENS_EINVAL, "gens_gemm_tn_batch_live: null count / bad group size");
    return gemm_tn_batch_run(count, a, lda, b, ldb, m, n, k, k_live, k_div, k_rows, workspace, c, stream);
}

static int gemm_tn_batch_run(int count, const float* const* a, const int* lda, const float* const* b, const int* ldb, const int* m,
                             const int* n, int64_t k, const int32_t* k_live, int k_div, int k_rows, float* workspace, float* c, void* stream) {
    int64_t tiles, csz;
    GENS_CHECK_ARG(a && lda && b && ldb && workspace && c, GENS_EINVAL, "gens_gemm_tn_batch: null pointer");
    GENS_CHECK_ARG(k > 0 && gemm_tn_batch_layout(count, m, n, &tiles, &csz) == 0, GENS_ELIMIT,
                   "gens_gemm_tn_batch: 1 <= count <= %d products with 1 <= m, n <= 1024 and k > 0", GEMM_TN_MAX_BATCH);
    GemmTnBatch B;
    B.count = count;
    B.csz = csz;
    B.k_live = k_live;
    B.k_div = k_div;
    B.k_rows = k_rows;
    int t0 = 0;
    int64_t off = 0;
    for (int p = 0; p < GEMM_TN_MAX_BATCH; ++p) {
        const bool live = p < count;
        GENS_CHECK_ARG(!live || (a[p] && b[p] && lda[p] >= m[p] && ldb[p] >= n[p]), GENS_EINVAL, "gens_gemm_tn_batch: product %d: null operand or leading dimension too small", p);
        B.a[p] = live ? a[p] : nullptr;
        B.b[p] = live ? b[p] : nullptr;
        B.lda[p] = live ? lda[p] : 0;
        B.ldb[p] = live ? ldb[p] : 0;
        B.m[p] = live ? m[p] : 0;
        B.n[p] = live ? n[p] : 0;
        B.tile0[p] = t0;
        B.off[p] = off;
        if (live) {
            t0 += ((m[p] + 31) / 32) * ((n[p] + 31) / 32);
            off += (int64_t)m[p] * n[p];
        }
    }
    B.tile0[GEMM_TN_MAX_BATCH] = t0;
    for (int p = count; p <= GEMM_TN_MAX_BATCH; ++p) B.tile0[p] = t0;
    // even sizes and 8-byte aligned operands: the register-blocked kernel (64 x 64 blocks)
    bool even = !getenv("GENS_GEMM_TN_32");
    for (int p = 0; p < count; ++p)
        even = even && !((m[p] | n[p] | lda[p] | ldb[p]) & 1) && !(((uintptr_t)a[p] | (uintptr_t)b[p]) & 7);
    even = even && !((uintptr_t)workspace & 7);
    for (int p = 0; p < count && even; ++p) even = !(B.off[p] & 1);
    hipStream_t s = (hipStream_t)stream;
    int n_slabs;
    if (even) {
        int64_t tiles64, csz64;
        gemm_tn_batch_layout(count, m, n, &tiles64, &csz64, 64);
        int t64 = 0;
        for (int p = 0; p <= GEMM_TN_MAX_BATCH; ++p) {
            B.tile0[p] = t64;
            if (p < count) t64 += ((m[p] + 63) / 64) * ((n[p] + 63) / 64);
        }
        const int64_t slab = gemm_tn_batch_slab(k, tiles64);
        n_slabs = (int)((k + slab - 1) / slab);
        const int n_wg = (int)((tiles64 + 3) / 4);
        gemm_tn_batch2_partial_k<<<(unsigned)(((n_slabs + 7) / 8) * 8 * n_wg), 256, 0, s>>>(B, k, slab, n_slabs, n_wg, workspace);
    } else {
        const int64_t slab = gemm_tn_batch_slab(k, tiles);
        n_slabs = (int)((k + slab - 1) / slab);
        const int n_wg = (int)((tiles + 3) / 4);
        gemm_tn_batch_partial_k<<<(unsigned)(((n_slabs + 7) / 8) * 8 * n_wg), 256, 0, s>>>(B, k, slab, n_slabs, n_wg, workspace);
    }
    if (n_slabs <= GEMM_TN_GROUPS) {
        gemm_tn_reduce_k<<<dim3(gens_blocks(csz, 256), 1), 256, 0, s>>>(workspace, n_slabs, n_slabs, csz, c);
    } else {
        const int per = (n_slabs + GEMM_TN_GROUPS - 1) / GEMM_TN_GROUPS, groups = (n_slabs + per - 1) / per;
        float* stage = workspace + (int64_t)n_slabs * csz;
        gemm_tn_reduce_k<<<dim3(gens_blocks(csz, 256), groups), 256, 0, s>>>(workspace, n_slabs, per, csz, stage);
        gemm_tn_reduce_k<<<dim3(gens_blocks(csz, 256), 1), 256, 0, s>>>(stage, groups, groups, csz, c);
    }
    return gens_launch_status("gens_gemm_tn_batch");
}
