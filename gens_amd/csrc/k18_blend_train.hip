// K18: the colour branch of a TRAINING step -- source-view feature look-up (K4) + the IBRNet-style BlendingNetwork -- forward in one
// launch, backward in one launch (+ one batched K14 launch for the eleven weight gradients), instead of ~250 PyTorch launches.
//
// Replaces, in train / fine-tune mode, lookup_feature + compute_angle (/root/reference/models/modules/projector.py:278-349) followed by
// BlendingNetwork.forward (models/modules/blending_network.py:69-118) as called from implicit_surface.py:196-199, and their autograd
// backward under loss.backward().  First order only (the colour branch is never differentiated twice).
//
// A training step has ~62 000 valid samples x 4 source views: the arithmetic is 10 GFLOP, nothing; what the PyTorch path pays is
// launches and host time (4.4 ms of a 15 ms fine-tune step).  So this kernel is written for clarity, not for the last TFLOP/s: one
// workgroup of four waves = 32 (point, view) rows, EVERY activation of the rows kept in LDS (79 KB, two workgroups per CU; the waves split
// the element-wise passes, the operand stores and the output tiles of a layer), every layer on the fp32
// matrix cores straight from the RAW row-major weights (no packed streams: the weights change every step, and 43 KB of them live in L1 / L2).
// The backward launch recomputes the forward for its rows (cheaper than a stash), walks the layers in reverse, and leaves for each
// layer the operand rows of its weight-gradient product: L = cotangent of the pre-activation, R = [input | 1] (the 1 yields the bias
// gradient), which gens_gemm_tn_batch multiplies over all rows.  The cotangent of the looked-up features goes to K4's own backward
// (gens_lookup_feature_bwd) when the feature maps / images ask for a gradient.
#include "k4_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BT_NLAYER 11
#define BT_WAVES 4
#define BT_THREADS (64 * BT_WAVES)
// LDS row strides (odd: conflict-free column walks) of the per-row arrays
#define BT_S_RD 9
#define BT_S_D1 17
#define BT_S_DFE 25
#define BT_S_H0 73
#define BT_S_TB 65
#define BT_S_H 33
#define BT_S_TV 33
#define BT_S_HV 37
#define BT_S_HH 41
#define BT_S_TU 33
#define BT_S_T1 17
#define BT_S_T2 9
#define BT_S_A 73
#define BT_S_SC 12

struct BlendRaw {   // raw nn.Linear parameters, row major (out, in)
    const float *rd1, *rd1b, *rd2, *rd2b;     // ray_dir_fc: 4 -> 16 -> F
    const float *b1, *b1b, *b2, *b2b;         // base_fc:    3F -> 64 -> 32
    const float *v1, *v1b, *v2, *v2b;         // vis_fc:     32 -> 32 -> 33
    const float *u1, *u1b, *u2, *u2b;         // vis_fc2:    32 -> 32 -> 1
    const float *r1, *r1b, *r2, *r2b, *r3, *r3b;   // rgb_fc: 37 -> 16 -> 8 -> 1
    const float* s;                           // anti-alias temperature (1)
};

struct BlendTrainIO {
    const float4* imgs;
    const float *w2c, *intr, *c2w;
    int nv;
    const float* pts;
    const int64_t* index;  // point i of the launch is pts[index[i]]; rgb / vis / g_rgb live at row index[i] (NULL: i)
    const int32_t* n_dev;  // only min(n, *n_dev) points exist (NULL: n)
    int64_t n;
    float* rgb_out;        // forward: (N, 3)
    uint8_t* vis_out;      // forward: (N, S) or NULL
    // backward
    const float* g_rgb;    // (N, 3) cotangent of rgb_out
    float* R[BT_NLAYER];   // (rows_pad, in_l + 1 rounded up to even): [input | 1 | 0]
    float* L[BT_NLAYER];   // (rows_pad, out_l rounded up to even)
    float* g_feat;         // (n, S, F) cotangent of the looked-up [rgb | features] rows, or NULL
    float* s_part;         // (waves) partial sums of d loss / d |s|
    float* acc_parts;      // ACC launches: (workgroups, csz) the workgroups' sums of L^T [R | 1] over their row tiles, layout of gens_gemm_tn_batch's result
    int n_tiles;           // ACC launches: row tiles of the launch (a workgroup walks tiles blockIdx.x, blockIdx.x + gridDim.x, ...)
};

// The weight gradients INSIDE the backward launch (ACC): the eleven [dW | db] blocks are 11 K floats -- 65 (NLEV = 5) 16 x 16 tiles that fit the
// registers of a workgroup's four waves (tile j of a layer belongs to wave j % 4: 21 accumulators of four registers per lane).  The workgroups
// are persistent; where the other form stores a layer's cotangent rows L and input rows [R | 1] for gens_gemm_tn_batch (619 MB written per launch
// and read again: ~0.1 ms of this kernel and 0.45 ms of that one), this form multiplies them out of LDS -- 8 MFMAs per tile and row tile -- and
// leaves ONE block of sums per workgroup.
__host__ __device__ constexpr int bt_ev(int x) { return (x + 1) & ~1; }
__host__ __device__ constexpr int bt_in(int l, int F) { return l == 0 ? 4 : l == 1 ? 16 : l == 2 ? 3 * F : l == 3 ? 64 : l == 8 ? 37 : l == 9 ? 16 : l == 10 ? 8 : 32; }
__host__ __device__ constexpr int bt_out(int l, int F) { return l == 0 ? 16 : l == 1 ? F : l == 2 ? 64 : l == 5 ? 33 : l == 7 ? 1 : l == 8 ? 16 : l == 9 ? 8 : l == 10 ? 1 : 32; }
__host__ __device__ constexpr int bt_ms(int l, int F) { return bt_ev(bt_out(l, F)); }            // rows of block l (gens_gemm_tn_batch's m)
__host__ __device__ constexpr int bt_ns(int l, int F) { return bt_ev(bt_in(l, F) + 1); }         // columns (n): [input | 1 | 0]
__host__ __device__ constexpr int bt_tiles(int l, int F) { return ((bt_ms(l, F) + 15) / 16) * ((bt_ns(l, F) + 15) / 16); }
__host__ __device__ constexpr int bt_acc_base(int l, int F) { return l == 0 ? 0 : bt_acc_base(l - 1, F) + (bt_tiles(l - 1, F) + BT_WAVES - 1) / BT_WAVES; }
__host__ __device__ constexpr int bt_cc_off(int l, int F) { return l == 0 ? 0 : bt_cc_off(l - 1, F) + bt_ms(l - 1, F) * bt_ns(l - 1, F); }
#define BT_NACC(F) bt_acc_base(BT_NLAYER, F)

__device__ __forceinline__ float bt_elu(float x) { return x > 0.0f ? x : hw_exp(x) - 1.0f; }
__device__ __forceinline__ float bt_elu_d(float out) { return out > 0.0f ? 1.0f : out + 1.0f; }    // d elu / d a from the OUTPUT
__device__ __forceinline__ int bt_crow(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// One 32 x 32 tile on the matrix cores from an LDS A tile (rows of stride rs, reduction length K) and a RAW weight matrix W (w_out, w_in):
//   TRANS = false: y[row][n0 + j] = sum_k A[row][k] W[n0 + j][k]      (forward, K = w_in)
//   TRANS = true : x[row][n0 + j] = sum_k A[row][k] W[k][n0 + j]      (reverse, K = w_out)
template <bool TRANS, int K>
__device__ __forceinline__ f32x16 bt_gemm(const float* __restrict__ A, int rs, const float* __restrict__ W, int w_out, int w_in, int n0, int lane) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int i = lane & 31, half = lane >> 5, n = n0 + i;
    const int N = TRANS ? w_in : w_out;
    constexpr int KP = (K + 1) / 2;
    // every weight of the tile first (the loads overlap each other's L2 latency: two waves per CU cannot hide a load -> MFMA chain), then
    // the matrix instructions
    float bv[KP];
#pragma unroll
    for (int j = 0; j < KP; ++j) {
        const int k = 2 * j + half;
        bv[j] = (k < K && n < N) ? (TRANS ? W[(size_t)k * w_in + n] : W[(size_t)n * w_in + k]) : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < KP; ++j) {
        const int k = 2 * j + half;
        const float a = k < K ? A[i * rs + k] : 0.0f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[j], acc, 0, 0, 0);
    }
    return acc;
}

// The same product as 16 x 16 tiles of v_mfma_f32_16x16x4_f32 (round 4): a layer's 32 x n_out outputs are 2 x ceil(n_out / 16) tiles dealt to
// the four waves, each a chain of ceil(K / 4) dependent 32-cycle MFMAs -- a quarter of the 32 x 32 x 2 chain's cycles, and every wave works in the
// 16- and 32-column layers that one wave used to serve alone (this kernel is latency-bound per workgroup: 22 dependent products with a barrier each).
//   tile rows [r0, r0 + 16), columns [n0, n0 + 16);  lane l: A[row l & 15][k = l >> 4], B[k = l >> 4][column l & 15], D rows 4 (l >> 4) + r
typedef float f32x4t __attribute__((ext_vector_type(4)));
typedef float bt_f2 __attribute__((ext_vector_type(2)));
// The weights of one tile (columns [n0, n0 + 16) of the product) into registers, and the product of rows [r0, r0 + 16) of the LDS tile A with them.
// Operand slot j of lane (i = lane & 15, kq = lane >> 4) is reduction index k(j, kq):
//   reverse products (TRANS: W[k][n], consecutive lanes on consecutive n -- 64-byte runs):   k = 4 j + kq;
//   forward products (W[n][k], a lane walks along ITS row):  k = 16 (j / 4) + 4 kq + j % 4 -- four consecutive floats per lane, ONE 16-byte load
//   instead of four 4-byte loads that each touch 16 different 128-byte lines: the vector L1 looks up one line per clock (gather_rate_probe), and
//   after every barrier all eight waves of a CU queue their weight loads there -- that queue, not the memory latency (90 ns when the pipe is
//   quiet), is what a layer waits for.  The reduction is over ceil(K / 16) * 16 slots then (the products are sums: any order of k is the same sum
//   up to float32 rounding; slots with k >= K carry zeros).
template <bool TRANS, int K>
struct BtSlots {
    static constexpr int n = TRANS ? (K + 3) / 4 : 4 * ((K + 15) / 16);
    static __device__ __forceinline__ int k(int j, int kq) { return TRANS ? 4 * j + kq : 16 * (j >> 2) + 4 * kq + (j & 3); }
};
struct __attribute__((packed, aligned(4))) bt_f4u { float x, y, z, w; };      // four floats at a 4-byte aligned address (rows of 69 / 37 floats)

template <bool TRANS, int K>
__device__ __forceinline__ void bt_load16(const float* __restrict__ W, int w_out, int w_in, int n0, int lane, float (&bv)[BtSlots<TRANS, K>::n]) {
    const int i = lane & 15, kq = lane >> 4, n = n0 + i;
    const int N = TRANS ? w_in : w_out;
    if constexpr (TRANS) {
#pragma unroll
        for (int j = 0; j < BtSlots<TRANS, K>::n; ++j) {
            const int k = 4 * j + kq;
            bv[j] = (k < K && n < N) ? W[(size_t)k * w_in + n] : 0.0f;
        }
    } else {
        const float* row = W + (size_t)(n < N ? n : 0) * w_in;
#pragma unroll
        for (int m = 0; m < (K + 15) / 16; ++m) {
            const int k0 = 16 * m + 4 * kq;
            if (k0 + 3 < K) {
                const bt_f4u v = *(const bt_f4u*)(row + k0);
                bv[4 * m] = v.x; bv[4 * m + 1] = v.y; bv[4 * m + 2] = v.z; bv[4 * m + 3] = v.w;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) bv[4 * m + r] = k0 + r < K ? row[k0 + r] : 0.0f;
            }
            if (!(n < N)) { bv[4 * m] = 0.0f; bv[4 * m + 1] = 0.0f; bv[4 * m + 2] = 0.0f; bv[4 * m + 3] = 0.0f; }
        }
    }
}
template <bool TRANS, int K>
__device__ __forceinline__ f32x4t bt_mma16(const float* __restrict__ A, int rs, int r0, int lane, const float (&bv)[BtSlots<TRANS, K>::n]) {
    f32x4t acc = {0.0f, 0.0f, 0.0f, 0.0f};
    const int i = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int j = 0; j < BtSlots<TRANS, K>::n; ++j) {
        const int k = BtSlots<TRANS, K>::k(j, kq);
        const float a = k < K ? A[(r0 + i) * rs + k] : 0.0f;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[j], acc, 0, 0, 0);
    }
    return acc;
}

// The kernel's arguments as the tile function sees them: IN the kernarg segment (constant address space, scalar loads on demand).  A persistent
// launch that takes them as by-value parameters keeps ~240 scalar registers of pointers and map geometry live across its tile loop (their loads are
// loop-invariant): 135 of them spilled to vector lanes, 65 v_readlane per layer and the wait states behind each (3 % of the kernel).
// The loop re-derives the pointer per tile (an opaque scalar), so a tile loads what it needs where it needs it.
struct BlendTrainArgs {
    BlendRaw W;
    MapSet fs;
    BlendTrainIO io;
};
typedef const __attribute__((address_space(4))) BlendTrainArgs* BlendTrainArgsPtr;

    // The weights (and the bias) of a product are loaded ONE PRODUCT AHEAD into registers (BT_PRE_F / BT_PRE_R, placed in front of the previous
    // product): a global load takes 0.7 - 1.9 us here even when it hits L1 / L2 (s_memtime stamps inside a layer: 3 500 cycles for a 32 x 32
    // weight tile, the same when repeated; 1 700 for a bias) -- with a load -> product -> barrier chain per layer that latency WAS the kernel
    // (22 products x ~4 000 cycles of a 100 000-cycle tile).  Loads and stores complete in issue order (one vmcnt counter on gfx9): the
    // operand-row stores of a layer (STORE_) are issued after the loads of the NEXT product, so nobody waits for them.
#define BT_PRE_F(TAG, k_in, Wm, Bv, n_out)                                                                   \
    constexpr int nt_##TAG = 2 * (((n_out) + 15) / 16), tpw_##TAG = (nt_##TAG + BT_WAVES - 1) / BT_WAVES;    \
    float pw_##TAG[tpw_##TAG][BtSlots<false, k_in>::n], pb_##TAG[tpw_##TAG];                                        \
    _Pragma("unroll") for (int ti_ = 0; ti_ < tpw_##TAG; ++ti_) {                                            \
        const int t_ = wave + BT_WAVES * ti_;                                                                \
        const int cb_ = 16 * (t_ >> 1) + (lane & 15);                                                        \
        pb_##TAG[ti_] = (t_ < nt_##TAG && cb_ < (n_out)) ? (Bv)[cb_] : 0.0f;                                 \
        if (t_ < nt_##TAG) bt_load16<false, k_in>(Wm, n_out, k_in, 16 * (t_ >> 1), lane, pw_##TAG[ti_]);     \
    }
#define BT_PRE_R(TAG, k_out, Wm, n_in)                                                                       \
    constexpr int nt_##TAG = 2 * (((n_in) + 15) / 16), tpw_##TAG = (nt_##TAG + BT_WAVES - 1) / BT_WAVES;     \
    float pw_##TAG[tpw_##TAG][BtSlots<true, k_out>::n];                                                            \
    if (BWD) {                                                                                               \
        _Pragma("unroll") for (int ti_ = 0; ti_ < tpw_##TAG; ++ti_) {                                        \
            const int t_ = wave + BT_WAVES * ti_;                                                            \
            if (t_ < nt_##TAG) bt_load16<true, k_out>(Wm, k_out, n_in, 16 * (t_ >> 1), lane, pw_##TAG[ti_]); \
        }                                                                                                    \
    }
template <int NLEV, bool BWD, bool ACC>
__device__ __forceinline__ void blend_train_tile(const __attribute__((address_space(4))) BlendRaw& W, const __attribute__((address_space(4))) MapSet& fs,
                                                 const __attribute__((address_space(4))) BlendTrainIO& io, const unsigned tile,
                                                 f32x4t (&wacc)[BT_NACC(3 + 4 * NLEV)]) {
    constexpr int F = 3 + 4 * NLEV, F3 = 3 * F;
    static_assert(F3 + 1 <= BT_S_A && F3 <= BT_S_H0, "tile too narrow");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* RD = smem;                      // [32][9]   ray difference (4)
    float* D1 = RD + 32 * BT_S_RD;         // [32][17]  ray_dir_fc hidden (16)
    float* DFE = D1 + 32 * BT_S_D1;        // [32][25]  direction feature (F)
    float* H0 = DFE + 32 * BT_S_DFE;       // [32][73]  [mean | var | x]
    float* TB = H0 + 32 * BT_S_H0;         // [32][65]  base_fc hidden (64)
    float* H = TB + 32 * BT_S_TB;          // [32][33]  base_fc output (32)
    float* TV = H + 32 * BT_S_H;           // [32][33]  vis_fc hidden
    float* HV = TV + 32 * BT_S_TV;         // [32][37]  vis_fc output (33)
    float* HH = HV + 32 * BT_S_HV;         // [32][41]  [h + res | vis2 | ray difference] (37)
    float* TU = HH + 32 * BT_S_HH;         // [32][33]  vis_fc2 hidden
    float* T1 = TU + 32 * BT_S_TU;         // [32][17]  rgb_fc hidden 1
    float* T2 = T1 + 32 * BT_S_T1;         // [32][9]   rgb_fc hidden 2
    float* A0 = T2 + 32 * BT_S_T2;         // [32][73]  scratch operand / cotangent tiles
    float* A1 = A0 + 32 * BT_S_A;          // [32][73]
    float* SC = A1 + 32 * BT_S_A;          // [32][12]  per row: 0 mask, 1 e, 2 w, 3 vis, 4 vis2, 5 score, 6 p, 7 w_bar, 8 vis_bar, 9 e_bar, 10 dot
    float* C = SC + 32 * BT_S_SC;          // [32][3]   rgb_in
    float* GX = C + 32 * 3;                // [32][25]  cotangent of the looked-up [rgb | features] row
    float* GH = GX + 32 * BT_S_DFE;        // [32][33]  cotangent of h / h + res
    float* PP = GH + 32 * BT_S_H;          // [32][4]   per point: 0 sum of raw weights, 1 arg-min view, 2 spare, 3 spare

    int tid_ = threadIdx.x;
    if (ACC) {                                   // the thread index OPAQUE per tile: hoisted out of the persistent loop, the LDS address arithmetic of a
        asm volatile("" : "+v"(tid_));           // tile costs ~300 registers (402 instead of 182; at most 256 fit two workgroups per CU)
        __builtin_assume((unsigned)tid_ < (unsigned)BT_THREADS);
    }
    const int tid = tid_, lane = tid & 63, wave = ACC ? __builtin_amdgcn_readfirstlane(tid >> 6) : tid >> 6, row = tid & 31;
    const int lane_w = lane;
    const int S = io.nv - 1, PPW = 32 / S;
    const int64_t first = (int64_t)tile * PPW;
    const int64_t n = io.n_dev ? min(io.n, (int64_t)io.n_dev[0]) : io.n;
    if (first >= n) {                                                // a workgroup past the device-side count: nothing to do
        if (BWD && threadIdx.x == 0) io.s_part[tile] = 0.0f;   // (its operand rows are never read: gens_gemm_tn_batch_live)
        return;
    }
    const int pl = row / S, sv = row % S + 1;
    const bool live = pl < PPW && first + pl < n;
    const int64_t src = live ? (io.index ? io.index[first + pl] : first + pl) : 0;      // dense row of this (point, view) row's point
    const int64_t grow0 = (int64_t)tile * 32;                  // first operand row of this workgroup
    const bool owner = tid < 32;                                     // one thread per row for the per-row scalars

    BT_PRE_F(rd1, 4, W.rd1, W.rd1b, 16)
    // ---------------------------------------------------------------- look-up (K4): [rgb | features] of the row into H0[:, 2F..3F)
    {   // thread (row, part): part l < NLEV reads feature level l (level 0 also the image); the in-frustum flags meet in GH (free here)
        const int part = tid >> 5;
        float x = 0.f, y = 0.f, z = 0.f;
        if (live) { x = io.pts[3 * src]; y = io.pts[3 * src + 1]; z = io.pts[3 * src + 2]; }
        float* xr = H0 + row * BT_S_H0 + 2 * F;
        if (part < NLEV) {
            const int l = part;
            const int h = fs.h[l], w = fs.w[l];
            SrcProj p = project_src(io.w2c + 16 * sv, io.intr + 16 * sv, exp2f(-(float)l), h, w, fs.cw[l], fs.ch[l], fs.rcw[l], fs.rch[l], x, y, z);
            GH[row * BT_S_H + l] = p.inside ? 1.0f : 0.0f;
            float4 f = f4_zero(), c = f4_zero();
            if (live) {
                Taps2 t = bilinear_taps(p.ix, p.iy, h, w);
                f = sample_texel(fs.data[l] + (int64_t)sv * h * w, h, w, 1, 0, t);
                if (l == 0) c = sample_texel(io.imgs + (int64_t)sv * h * w, h, w, 1, 0, t);
            }
            xr[3 + 4 * l] = f.x; xr[4 + 4 * l] = f.y; xr[5 + 4 * l] = f.z; xr[6 + 4 * l] = f.w;
            if (l == 0) {
                xr[0] = c.x; xr[1] = c.y; xr[2] = c.z;
                C[row * 3] = c.x; C[row * 3 + 1] = c.y; C[row * 3 + 2] = c.z;
            }
        }
        __syncthreads();
        if (owner) {
            bool inside = true;
            for (int l = 0; l < NLEV; ++l) inside = inside && GH[row * BT_S_H + l] != 0.0f;
            SC[row * BT_S_SC] = (live && inside) ? 1.0f : 0.0f;
            if (live && io.vis_out && !BWD) io.vis_out[src * S + (sv - 1)] = inside ? 1 : 0;
            // compute_angle (projector.py:278-291), IEEE square roots / divisions as the PyTorch path takes them
            float rx = io.c2w[3] - x, ry = io.c2w[7] - y, rz = io.c2w[11] - z;
            const float rn = sqrtf(rx * rx + ry * ry + rz * rz) + 1e-6f;
            rx /= rn; ry /= rn; rz /= rn;
            const float* cs = io.c2w + 16 * sv;
            float sx = cs[3] - x, sy = cs[7] - y, sz = cs[11] - z;
            const float sn = sqrtf(sx * sx + sy * sy + sz * sz) + 1e-6f;
            sx /= sn; sy /= sn; sz /= sn;
            const float dx = rx - sx, dy = ry - sy, dz = rz - sz;
            const float dn = fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-6f);
            float* rd = RD + row * BT_S_RD;
            rd[0] = live ? dx / dn : 0.0f;
            rd[1] = live ? dy / dn : 0.0f;
            rd[2] = live ? dz / dn : 0.0f;
            rd[3] = live ? rx * sx + ry * sy + rz * sz : 0.0f;
        }
    }
    __syncthreads();

#define BT_FOR(i, count) for (int i = tid; i < (count); i += BT_THREADS)
    // write the operand rows [input | 1] of layer l (width `in`) from an LDS array
#define BT_STORE_R(l, SRC, stride, in)                                                                       \
    if (BWD && !ACC) {                                                                                               \
        constexpr int rw_ = ((in) + 2) & ~1;          /* [input | 1 | 0]: even width (8-byte loads of the batched product) */ \
        constexpr int hw_ = rw_ / 2;                  /* ... and 8-byte stores here: the operand rows are 45 % of this kernel's time as 4-byte stores */ \
        BT_FOR(i_, 32 * hw_) {                                                                               \
            const int r_ = i_ / hw_, c_ = 2 * (i_ % hw_);                                                    \
            float2 v_;                                                                                       \
            v_.x = c_ < (in) ? (SRC)[r_ * (stride) + c_] : (c_ == (in) ? 1.0f : 0.0f);                       \
            v_.y = c_ + 1 < (in) ? (SRC)[r_ * (stride) + c_ + 1] : (c_ + 1 == (in) ? 1.0f : 0.0f);           \
            __builtin_nontemporal_store((bt_f2){v_.x, v_.y}, (bt_f2*)(io.R[l] + (grow0 + r_) * rw_ + c_));                                              \
        }                                                                                                    \
    }
    // one forward layer: OUT[row][c] = elu(bias + IN W^T) for c < n_out, from the registers BT_PRE_F(TAG) filled
#define BT_LAYER(TAG, IN, s_in, k_in, n_out, OUT, s_out, ACT, STORE_)                                        \
    {                                                                                                        \
        STORE_                                                                                               \
        _Pragma("unroll") for (int ti_ = 0; ti_ < tpw_##TAG; ++ti_) {                                        \
            const int t_ = wave + BT_WAVES * ti_;                                                            \
            if (t_ < nt_##TAG) {                                                                             \
                const int r0_ = 16 * (t_ & 1), n0_ = 16 * (t_ >> 1);                                         \
                const f32x4t acc_ = bt_mma16<false, k_in>(IN, s_in, r0_, lane, pw_##TAG[ti_]);                      \
                const int c_ = n0_ + (lane & 15);                                                            \
                if (c_ < (n_out)) {                                                                          \
                    const float bias_ = pb_##TAG[ti_];                                                       \
                    _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) {                                       \
                        const float v_ = acc_[r_] + bias_;                                                   \
                        (OUT)[(r0_ + 4 * (lane >> 4) + r_) * (s_out) + c_] = ACT ? bt_elu(v_) : v_;          \
                    }                                                                                        \
                }                                                                                            \
            }                                                                                                \
        }                                                                                                    \
    }

    // ---------------------------------------------------------------- ray_dir_fc, x = rgb_feat + direction feature (:87-89)
    BT_PRE_F(rd2, 16, W.rd2, W.rd2b, F)
    BT_LAYER(rd1, RD, BT_S_RD, 4, 16, D1, BT_S_D1, true, BT_STORE_R(0, RD, BT_S_RD, 4))
    __syncthreads();
    BT_PRE_F(b1, F3, W.b1, W.b1b, 64)
    BT_LAYER(rd2, D1, BT_S_D1, 16, F, DFE, BT_S_DFE, true, BT_STORE_R(1, D1, BT_S_D1, 16))
    __syncthreads();
    BT_FOR(i, 32 * F) {
        const int r = i / F, c = i % F;
        H0[r * BT_S_H0 + 2 * F + c] += DFE[r * BT_S_DFE + c];
    }
    const float s_abs = fabsf(W.s[0]);
    if (owner) SC[row * BT_S_SC + 1] = expf(s_abs * (RD[row * BT_S_RD + 3] - 1.0f));           // exp(|s| (dot - 1))  (:93)
    __syncthreads();
    // ---------------------------------------------------------------- view weights, weighted mean / variance (:94-101)
    if (owner && pl < PPW && sv == 1) {                       // one lane per point
        const int base = pl * S;
        float mn = 3.4e38f;
        int arg = 0;
        for (int v = 0; v < S; ++v) {
            const float e = SC[(base + v) * BT_S_SC + 1];
            if (e < mn) { mn = e; arg = v; }
        }
        float sum = 0.0f;
        for (int v = 0; v < S; ++v) sum += (SC[(base + v) * BT_S_SC + 1] - mn) * SC[(base + v) * BT_S_SC];
        for (int v = 0; v < S; ++v) SC[(base + v) * BT_S_SC + 2] = (SC[(base + v) * BT_S_SC + 1] - mn) * SC[(base + v) * BT_S_SC] / (sum + 1e-8f);
        PP[pl * 4] = sum;
        PP[pl * 4 + 1] = (float)arg;
    }
    if (owner && pl >= PPW) SC[row * BT_S_SC + 2] = 0.0f;
    __syncthreads();
    BT_FOR(it, PPW * F) {
        const int p = it / F, c = it % F, base = p * S;
        float mean = 0.0f, var = 0.0f;
        for (int v = 0; v < S; ++v) mean += H0[(base + v) * BT_S_H0 + 2 * F + c] * SC[(base + v) * BT_S_SC + 2];
        for (int v = 0; v < S; ++v) {
            const float d = H0[(base + v) * BT_S_H0 + 2 * F + c] - mean;
            var += SC[(base + v) * BT_S_SC + 2] * (d * d);
        }
        for (int v = 0; v < S; ++v) {
            H0[(base + v) * BT_S_H0 + c] = mean;
            H0[(base + v) * BT_S_H0 + F + c] = var;
        }
    }
    BT_FOR(it, (32 - PPW * S) * 2 * F) {                          // unused rows (32 % S != 0): keep them finite
        const int rr = PPW * S + it / (2 * F);
        H0[rr * BT_S_H0 + it % (2 * F)] = 0.0f;
    }
    __syncthreads();
    // ---------------------------------------------------------------- base_fc (:103-104)
    BT_PRE_F(b2, 64, W.b2, W.b2b, 32)
    BT_LAYER(b1, H0, BT_S_H0, F3, 64, TB, BT_S_TB, true, BT_STORE_R(2, H0, BT_S_H0, F3))
    __syncthreads();
    BT_PRE_F(v1, 32, W.v1, W.v1b, 32)
    BT_LAYER(b2, TB, BT_S_TB, 64, 32, H, BT_S_H, true, BT_STORE_R(3, TB, BT_S_TB, 64))
    __syncthreads();
    // ---------------------------------------------------------------- vis_fc on h * w (:106-109)
    BT_FOR(i, 32 * 32) {
        const int r = i >> 5, c = i & 31;
        A0[r * BT_S_A + c] = H[r * BT_S_H + c] * SC[r * BT_S_SC + 2];
    }
    __syncthreads();
    BT_PRE_F(v2, 32, W.v2, W.v2b, 33)
    BT_LAYER(v1, A0, BT_S_A, 32, 32, TV, BT_S_TV, true, BT_STORE_R(4, A0, BT_S_A, 32))
    __syncthreads();
    BT_PRE_F(u1, 32, W.u1, W.u1b, 32)
    BT_LAYER(v2, TV, BT_S_TV, 32, 33, HV, BT_S_HV, true, BT_STORE_R(5, TV, BT_S_TV, 32))
    __syncthreads();
    if (owner) SC[row * BT_S_SC + 3] = (1.0f / (1.0f + expf(-HV[row * BT_S_HV + 32]))) * SC[row * BT_S_SC];      // vis
    BT_FOR(i, 32 * 32) {
        const int r = i >> 5, c = i & 31;
        HH[r * BT_S_HH + c] = H[r * BT_S_H + c] + HV[r * BT_S_HV + c];                                              // x = x + x_res
    }
    __syncthreads();
    // ---------------------------------------------------------------- vis_fc2 on x * vis (:110)
    BT_FOR(i, 32 * 32) {
        const int r = i >> 5, c = i & 31;
        A0[r * BT_S_A + c] = HH[r * BT_S_HH + c] * SC[r * BT_S_SC + 3];
    }
    __syncthreads();
    BT_PRE_F(r1, 37, W.r1, W.r1b, 16)
    BT_LAYER(u1, A0, BT_S_A, 32, 32, TU, BT_S_TU, true, BT_STORE_R(6, A0, BT_S_A, 32))
    __syncthreads();
    BT_STORE_R(7, TU, BT_S_TU, 32)
    if (owner) {
        float q = W.u2b[0];
        for (int k = 0; k < 32; ++k) q += TU[row * BT_S_TU + k] * W.u2[k];
        const float v2 = (1.0f / (1.0f + expf(-q))) * SC[row * BT_S_SC];
        SC[row * BT_S_SC + 4] = v2;
        float* hh = HH + row * BT_S_HH;
        hh[32] = v2;
        hh[33] = RD[row * BT_S_RD]; hh[34] = RD[row * BT_S_RD + 1]; hh[35] = RD[row * BT_S_RD + 2]; hh[36] = RD[row * BT_S_RD + 3];
    }
    __syncthreads();
    // ---------------------------------------------------------------- rgb_fc on cat([x, vis, ray_diff]) (:113-114)
    BT_PRE_F(r2, 16, W.r2, W.r2b, 8)
    BT_LAYER(r1, HH, BT_S_HH, 37, 16, T1, BT_S_T1, true, BT_STORE_R(8, HH, BT_S_HH, 37))
    __syncthreads();
    BT_PRE_R(xr2, 8, W.r2, 16)
    BT_LAYER(r2, T1, BT_S_T1, 16, 8, T2, BT_S_T2, true, BT_STORE_R(9, T1, BT_S_T1, 16))
    __syncthreads();
    BT_STORE_R(10, T2, BT_S_T2, 8)
    if (owner) {
        float sc = W.r3b[0];
        for (int k = 0; k < 8; ++k) sc += T2[row * BT_S_T2 + k] * W.r3[k];
        SC[row * BT_S_SC + 5] = (SC[row * BT_S_SC] == 0.0f) ? -1e9f : sc;                  // masked_fill(mask == 0, -1e9)  (:115)
    }
    __syncthreads();
    // ---------------------------------------------------------------- softmax over views, colour (:116-117)
    if (tid < PPW && first + tid < n) {
        const int base = tid * S;
        float mx = -3.4e38f;
        for (int v = 0; v < S; ++v) mx = fmaxf(mx, SC[(base + v) * BT_S_SC + 5]);
        float den = 0.0f;
        for (int v = 0; v < S; ++v) den += expf(SC[(base + v) * BT_S_SC + 5] - mx);
        float cr = 0.0f, cg = 0.0f, cb = 0.0f;
        for (int v = 0; v < S; ++v) {
            const float p = expf(SC[(base + v) * BT_S_SC + 5] - mx) / den;
            SC[(base + v) * BT_S_SC + 6] = p;
            cr += C[(base + v) * 3] * p;
            cg += C[(base + v) * 3 + 1] * p;
            cb += C[(base + v) * 3 + 2] * p;
        }
        if (!BWD) {
            const int64_t dst = io.index ? io.index[first + tid] : first + tid;
            io.rgb_out[3 * dst] = cr;
            io.rgb_out[3 * dst + 1] = cg;
            io.rgb_out[3 * dst + 2] = cb;
        }
    }
    if constexpr (!BWD) return;
    __syncthreads();

    // ================================================================ reverse
    // store the cotangent rows of layer l's pre-activation (width out) from an LDS tile
    // ACC: block l += L^T [R | 1] of this row tile.  L = (LS)[row][c < out], R = (RS)[row][c < in] (times column SCOL of SC when SCOL >= 0: the two
    // layers whose input is h * weight / x * visibility).  D[m = out index][n = in index]: lane supplies A[m = lane & 15][k = row 4 q + lane / 16] and
    // B[k][n = lane & 15]
#define BT_WACC(l, LS, ls, RS, rs, SCOL)                                                                     \
    if (ACC) {                                                                                               \
        constexpr int out_ = bt_out(l, F), in_ = bt_in(l, F), tn_ = (bt_ns(l, F) + 15) / 16, t_all_ = bt_tiles(l, F), base_ = bt_acc_base(l, F); \
        _Pragma("unroll") for (int j_ = 0; j_ < t_all_; ++j_) {                                              \
            if (wave == (j_ & (BT_WAVES - 1))) {                                                             \
                const int oc_ = 16 * (j_ / tn_) + (lane_w & 15), ic_ = 16 * (j_ % tn_) + (lane_w & 15);          \
                f32x4t a_ = wacc[base_ + j_ / BT_WAVES];                                                     \
                _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) {                                           \
                    const int r_ = 4 * q_ + (lane_w >> 4);                                                     \
                    const float av_ = oc_ < out_ ? (LS)[r_ * (ls) + oc_] : 0.0f;                             \
                    float bv_ = ic_ < in_ ? (RS)[r_ * (rs) + ic_] : (ic_ == in_ ? 1.0f : 0.0f);              \
                    if ((SCOL) >= 0 && ic_ < in_) bv_ *= SC[r_ * BT_S_SC + ((SCOL) >= 0 ? (SCOL) : 0)];     \
                    a_ = __builtin_amdgcn_mfma_f32_16x16x4f32(av_, bv_, a_, 0, 0, 0);                        \
                }                                                                                            \
                wacc[base_ + j_ / BT_WAVES] = a_;                                                            \
                __builtin_amdgcn_sched_barrier(0);        /* (one tile's 16 LDS reads in flight, not every tile's) */ \
            }                                                                                                \
        }                                                                                                    \
    }
#define BT_STORE_L(l, SRC, stride, out)                                                                      \
    if (!ACC) {                                                                                                        \
        constexpr int lw_ = ((out) + 1) & ~1, hl_ = lw_ / 2;                                                 \
        BT_FOR(i_, 32 * hl_) {                                                                               \
            const int r_ = i_ / hl_, c_ = 2 * (i_ % hl_);                                                    \
            float2 v_;                                                                                       \
            v_.x = c_ < (out) ? (SRC)[r_ * (stride) + c_] : 0.0f;                                            \
            v_.y = c_ + 1 < (out) ? (SRC)[r_ * (stride) + c_ + 1] : 0.0f;                                    \
            __builtin_nontemporal_store((bt_f2){v_.x, v_.y}, (bt_f2*)(io.L[l] + (grow0 + r_) * lw_ + c_));                                              \
        }                                                                                                    \
    }
    // X_bar tile(s) = A W (reverse product, weights from BT_PRE_R(TAG)), then DST[row][c] (=|+=) X_bar * elu'(OUT_ACT) for c < n_in
#define BT_REVERSE(TAG, IN, s_in, k_out, n_in, DST, s_dst, BODY, STORE_)                                     \
    {                                                                                                        \
        STORE_                                                                                               \
        _Pragma("unroll") for (int ti_ = 0; ti_ < tpw_##TAG; ++ti_) {                                        \
            const int t_ = wave + BT_WAVES * ti_;                                                            \
            if (t_ < nt_##TAG) {                                                                             \
                const int r0_ = 16 * (t_ & 1), n0_ = 16 * (t_ >> 1);                                         \
                const f32x4t acc_ = bt_mma16<true, k_out>(IN, s_in, r0_, lane, pw_##TAG[ti_]);                     \
                const int c_ = n0_ + (lane & 15);                                                            \
                if (c_ < (n_in)) {                                                                           \
                    _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) {                                       \
                        const int rr_ = r0_ + 4 * (lane >> 4) + r_;                                          \
                        const float xb_ = acc_[r_];                                                          \
                        BODY                                                                                 \
                    }                                                                                        \
                }                                                                                            \
            }                                                                                                \
        }                                                                                                    \
    }

    // colour = sum_v rgb_in p_v: score_bar = p (p_bar - sum_u p_u p_bar_u), rgb_in_bar = g p
    if (owner) {
        float sb = 0.0f;
        float gx0 = 0.0f, gx1 = 0.0f, gx2 = 0.0f;
        if (live) {
            const float g0 = io.g_rgb[3 * src], g1 = io.g_rgb[3 * src + 1], g2 = io.g_rgb[3 * src + 2];
            const int base = pl * S;
            float dot = 0.0f;
            for (int v = 0; v < S; ++v)
                dot += SC[(base + v) * BT_S_SC + 6] * (g0 * C[(base + v) * 3] + g1 * C[(base + v) * 3 + 1] + g2 * C[(base + v) * 3 + 2]);
            const float p = SC[row * BT_S_SC + 6];
            sb = p * ((g0 * C[row * 3] + g1 * C[row * 3 + 1] + g2 * C[row * 3 + 2]) - dot);
            // masked_fill(mask == 0, -1e9) (blending_network.py:115) passes no gradient to the score it replaced.  With a visible view beside it p is
            // 0 there anyway; a point NO source view sees has p = 1 / S on every (masked) view -- a mid-point inside the mask volumes but outside
            // both source images, an image-corner ray of a three-view set -- and its softmax gradient must not reach the network
            if (SC[row * BT_S_SC] == 0.0f) sb = 0.0f;
            gx0 = g0 * p; gx1 = g1 * p; gx2 = g2 * p;
        }
        A1[row * BT_S_A] = sb;                                   // cotangent of the score = L of rgb_fc.4
        GX[row * BT_S_DFE] = gx0; GX[row * BT_S_DFE + 1] = gx1; GX[row * BT_S_DFE + 2] = gx2;
        for (int c = 3; c < F; ++c) GX[row * BT_S_DFE + c] = 0.0f;
        SC[row * BT_S_SC + 7] = 0.0f;                            // w_bar
    }
    __syncthreads();
    BT_STORE_L(10, A1, BT_S_A, 1)
    BT_WACC(10, A1, BT_S_A, T2, BT_S_T2, -1)
    // rgb_fc.4 -> rgb_fc.2 pre-activation
    BT_FOR(i, 32 * 8) {
        const int r = i >> 3, c = i & 7;
        A0[r * BT_S_A + c] = A1[r * BT_S_A] * W.r3[c] * bt_elu_d(T2[r * BT_S_T2 + c]);
    }
    __syncthreads();
    BT_PRE_R(xr1, 16, W.r1, 37)
    BT_REVERSE(xr2, A0, BT_S_A, 8, 16, A1, BT_S_A, A1[rr_ * BT_S_A + c_] = xb_ * bt_elu_d(T1[rr_ * BT_S_T1 + c_]);, BT_STORE_L(9, A0, BT_S_A, 8) BT_WACC(9, A0, BT_S_A, T1, BT_S_T1, -1))
    __syncthreads();
    // rgb_fc.0 input = [h2 (32) | vis2 | ray difference]: cotangent of h2 -> GH, of vis2 -> SC[8]
    BT_PRE_R(xu1, 32, W.u1, 32)
    BT_REVERSE(xr1, A1, BT_S_A, 16, 37, GH, BT_S_H,
               if (c_ < 32) GH[rr_ * BT_S_H + c_] = xb_; else if (c_ == 32) SC[rr_ * BT_S_SC + 8] = xb_;, BT_STORE_L(8, A1, BT_S_A, 16) BT_WACC(8, A1, BT_S_A, HH, BT_S_HH, -1))
    __syncthreads();
    // vis2 = sigmoid(q) mask ; q = u2 . tu + b
    if (owner) {
        const float v2 = SC[row * BT_S_SC + 4];
        A1[row * BT_S_A] = SC[row * BT_S_SC + 8] * SC[row * BT_S_SC] * v2 * (1.0f - v2);       // q_bar (mask is 0 or 1: vis2 = sigmoid there)
    }
    __syncthreads();
    BT_STORE_L(7, A1, BT_S_A, 1)
    BT_WACC(7, A1, BT_S_A, TU, BT_S_TU, -1)
    BT_FOR(i, 32 * 32) {
        const int r = i >> 5, c = i & 31;
        A0[r * BT_S_A + c] = A1[r * BT_S_A] * W.u2[c] * bt_elu_d(TU[r * BT_S_TU + c]);
    }
    __syncthreads();
    // vis_fc2.0 input = h2 * vis: h2_bar += m vis ; vis_bar = sum_k m_k h2_k
    BT_PRE_R(xv2, 33, W.v2, 32)
    BT_REVERSE(xu1, A0, BT_S_A, 32, 32, A1, BT_S_A, A1[rr_ * BT_S_A + c_] = xb_;, BT_STORE_L(6, A0, BT_S_A, 32) BT_WACC(6, A0, BT_S_A, HH, BT_S_HH, 3))
    __syncthreads();
    if (owner) {
        float vb = 0.0f;
        for (int k = 0; k < 32; ++k) vb += A1[row * BT_S_A + k] * HH[row * BT_S_HH + k];
        const float vis = SC[row * BT_S_SC + 3];
        SC[row * BT_S_SC + 8] = vb * SC[row * BT_S_SC] * vis * (1.0f - vis);                    // cotangent of hv[32] before its ELU
    }
    BT_FOR(i, 32 * 32) {
        const int r = i >> 5, c = i & 31;
        GH[r * BT_S_H + c] += A1[r * BT_S_A + c] * SC[r * BT_S_SC + 3];
    }
    __syncthreads();
    // h2 = h + hv[:32]: hv_bar[:32] = h_bar = GH ; vis = sigmoid(hv[32]) mask
    BT_FOR(i, 32 * 33) {
        const int r = i / 33, c = i % 33;
        const float g = c < 32 ? GH[r * BT_S_H + c] : SC[r * BT_S_SC + 8];
        A0[r * BT_S_A + c] = g * bt_elu_d(HV[r * BT_S_HV + c]);
    }
    __syncthreads();
    BT_PRE_R(xv1, 32, W.v1, 32)
    BT_REVERSE(xv2, A0, BT_S_A, 33, 32, A1, BT_S_A, A1[rr_ * BT_S_A + c_] = xb_ * bt_elu_d(TV[rr_ * BT_S_TV + c_]);, BT_STORE_L(5, A0, BT_S_A, 33) BT_WACC(5, A0, BT_S_A, TV, BT_S_TV, -1))
    __syncthreads();
    // vis_fc.0 input = h * w: h_bar += m w ; w_bar += sum_k m_k h_k
    BT_PRE_R(xb2, 32, W.b2, 64)
    BT_REVERSE(xv1, A1, BT_S_A, 32, 32, A0, BT_S_A, A0[rr_ * BT_S_A + c_] = xb_;, BT_STORE_L(4, A1, BT_S_A, 32) BT_WACC(4, A1, BT_S_A, H, BT_S_H, 2))
    __syncthreads();
    if (owner) {
        float wb = 0.0f;
        for (int k = 0; k < 32; ++k) wb += A0[row * BT_S_A + k] * H[row * BT_S_H + k];
        SC[row * BT_S_SC + 7] += wb;
    }
    BT_FOR(i, 32 * 32) {
        const int r = i >> 5, c = i & 31;
        const float hb = GH[r * BT_S_H + c] + A0[r * BT_S_A + c] * SC[r * BT_S_SC + 2];
        A1[r * BT_S_A + c] = hb * bt_elu_d(H[r * BT_S_H + c]);                                   // base_fc.2 pre-activation
    }
    __syncthreads();
    BT_PRE_R(xb1, 64, W.b1, F3)
    BT_REVERSE(xb2, A1, BT_S_A, 32, 64, A0, BT_S_A, A0[rr_ * BT_S_A + c_] = xb_ * bt_elu_d(TB[rr_ * BT_S_TB + c_]);, BT_STORE_L(3, A1, BT_S_A, 32) BT_WACC(3, A1, BT_S_A, TB, BT_S_TB, -1))
    __syncthreads();
    BT_PRE_R(xrd2, F, W.rd2, 16)
    BT_REVERSE(xb1, A0, BT_S_A, 64, F3, A1, BT_S_A, A1[rr_ * BT_S_A + c_] = xb_;, BT_STORE_L(2, A0, BT_S_A, 64) BT_WACC(2, A0, BT_S_A, H0, BT_S_H0, -1))                // cotangent of [mean | var | x]
    __syncthreads();
    // mean = sum_v w x, var = sum_v w (x - mean)^2 (shared by the views of a point)
    BT_FOR(it, PPW * F) {
        const int p = it / F, c = it % F, base = p * S;
        float mb = 0.0f, vb = 0.0f, cross = 0.0f;
        const float mean = H0[base * BT_S_H0 + c];
        for (int v = 0; v < S; ++v) {
            mb += A1[(base + v) * BT_S_A + c];
            vb += A1[(base + v) * BT_S_A + F + c];
            cross += SC[(base + v) * BT_S_SC + 2] * (H0[(base + v) * BT_S_H0 + 2 * F + c] - mean);
        }
        mb -= 2.0f * vb * cross;                                                                 // var depends on mean too
        for (int v = 0; v < S; ++v) {
            const float xv = H0[(base + v) * BT_S_H0 + 2 * F + c], w = SC[(base + v) * BT_S_SC + 2], d = xv - mean;
            const float xb = A1[(base + v) * BT_S_A + 2 * F + c] + w * mb + 2.0f * w * d * vb;
            GX[(base + v) * BT_S_DFE + c] += xb;
            A0[(base + v) * BT_S_A + c] = xb * bt_elu_d(DFE[(base + v) * BT_S_DFE + c]);         // ray_dir_fc.2 pre-activation
            atomicAdd(&SC[(base + v) * BT_S_SC + 7], mb * xv + vb * d * d);                      // w_bar (LDS)
        }
    }
    BT_FOR(it, (32 - PPW * S) * F) {
        const int rr = PPW * S + it / F;
        A0[rr * BT_S_A + it % F] = 0.0f;
    }
    __syncthreads();
    // w = wr / (sum wr + 1e-8), wr = (e - min e) mask, e = exp(|s| (dot - 1))
    float s_bar = 0.0f;
    if (owner && pl < PPW && sv == 1 && first + pl < n) {
        const int base = pl * S;
        const float sum = PP[pl * 4] + 1e-8f;
        const int arg = (int)PP[pl * 4 + 1];
        float ww = 0.0f;
        for (int v = 0; v < S; ++v) ww += SC[(base + v) * BT_S_SC + 7] * SC[(base + v) * BT_S_SC + 2];
        float tot = 0.0f;
        for (int v = 0; v < S; ++v) {
            const float wrb = (SC[(base + v) * BT_S_SC + 7] - ww) / sum * SC[(base + v) * BT_S_SC];     // wr_bar mask
            SC[(base + v) * BT_S_SC + 9] = wrb;
            tot += wrb;
        }
        SC[(base + arg) * BT_S_SC + 9] -= tot;                                                   // the minimum's share
        for (int v = 0; v < S; ++v)
            s_bar += SC[(base + v) * BT_S_SC + 9] * SC[(base + v) * BT_S_SC + 1] * (RD[(base + v) * BT_S_RD + 3] - 1.0f);
    }
    if (wave == 0) {                                               // (the owners all sit in wave 0)
        s_bar = wave_sum(s_bar);
        if (lane == 0) io.s_part[tile] = s_bar;
    }
    BT_REVERSE(xrd2, A0, BT_S_A, F, 16, A1, BT_S_A, A1[rr_ * BT_S_A + c_] = xb_ * bt_elu_d(D1[rr_ * BT_S_D1 + c_]);, BT_STORE_L(1, A0, BT_S_A, F) BT_WACC(1, A0, BT_S_A, D1, BT_S_D1, -1))
    __syncthreads();
    BT_STORE_L(0, A1, BT_S_A, 16)
    BT_WACC(0, A1, BT_S_A, RD, BT_S_RD, -1)
    if (io.g_feat) {
        BT_FOR(i, PPW * S * F) {
            const int r = i / F, c = i % F;
            const int64_t pt = first + r / S;
            if (pt < n) io.g_feat[(pt * S + r % S) * F + c] = GX[r * BT_S_DFE + c];
        }
    }
#undef BT_FOR
#undef BT_STORE_R
#undef BT_STORE_L
#undef BT_WACC
#undef BT_PRE_F
#undef BT_PRE_R
#undef BT_LAYER
#undef BT_REVERSE
}

template <int NLEV, bool BWD, bool ACC>
__global__ __launch_bounds__(BT_THREADS, 2) void blend_train_k(BlendRaw W, MapSet fs, BlendTrainIO io) {      // (two workgroups per CU: at most 256 registers)
    constexpr int F = 3 + 4 * NLEV, NACC = BT_NACC(F), CSZ = bt_cc_off(BT_NLAYER, F);   // (constexpr VARIABLES: a constexpr function in a loop bound is a run-time call)
    f32x4t wacc[NACC];
    BlendTrainArgsPtr kp = (BlendTrainArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();       // (W, fs, io) in declaration order
    if constexpr (!ACC) {
        blend_train_tile<NLEV, BWD, false>(kp->W, kp->fs, kp->io, blockIdx.x, wacc);
    } else {
#pragma unroll
        for (int k = 0; k < NACC; ++k) wacc[k] = (f32x4t){0.0f, 0.0f, 0.0f, 0.0f};
        for (unsigned tile = blockIdx.x; tile < (unsigned)io.n_tiles; tile += gridDim.x) {
            asm volatile("" : "+s"(kp));
            blend_train_tile<NLEV, true, true>(kp->W, kp->fs, kp->io, tile, wacc);
            __syncthreads();                                   // the next tile overwrites the LDS arrays this one's last products read
        }
        // this workgroup's block of sums: accumulator register r of lane l = D[m = 4 (l / 16) + r][n = l & 15] of its tile
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        float* out = io.acc_parts + (size_t)blockIdx.x * CSZ;
#define BT_FLUSH(l)                                                                                          \
        {                                                                                                    \
            constexpr int ms_ = bt_ms(l, F), ns_ = bt_ns(l, F), tn_ = (ns_ + 15) / 16, nt_ = bt_tiles(l, F), off_ = bt_cc_off(l, F), ab_ = bt_acc_base(l, F); \
            _Pragma("unroll") for (int j_ = 0; j_ < nt_; ++j_) {                                             \
                if (wave == (j_ & (BT_WAVES - 1))) {                                                         \
                    const int n_ = 16 * (j_ % tn_) + (lane & 15);                                            \
                    _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) {                                       \
                        const int m_ = 16 * (j_ / tn_) + 4 * (lane >> 4) + r_;                               \
                        if (m_ < ms_ && n_ < ns_) out[off_ + m_ * ns_ + n_] = wacc[ab_ + j_ / BT_WAVES][r_];     \
                    }                                                                                        \
                }                                                                                            \
            }                                                                                                \
        }
        BT_FLUSH(0) BT_FLUSH(1) BT_FLUSH(2) BT_FLUSH(3) BT_FLUSH(4) BT_FLUSH(5) BT_FLUSH(6) BT_FLUSH(7) BT_FLUSH(8) BT_FLUSH(9) BT_FLUSH(10)
#undef BT_FLUSH
    }
}

// out[e] = sum over the parts in a FIXED order (the result does not depend on scheduling): a workgroup = 32 elements x 8 groups of parts, a thread
// adds the parts p = g, g + 8, ... of its element, the eight sums are added in group order
__global__ __launch_bounds__(256) void blend_train_reduce_k(const float* __restrict__ parts, int n_parts, int csz, float* __restrict__ out) {
    __shared__ float red[8][32];
    const int el = threadIdx.x & 31, g = threadIdx.x >> 5, e = blockIdx.x * 32 + el;
    float s = 0.0f;
    if (e < csz)
        for (int p = g; p < n_parts; p += 8) s += parts[(size_t)p * csz + e];
    red[g][el] = s;
    __syncthreads();
    if (g == 0 && e < csz) out[e] = ((red[0][el] + red[1][el]) + (red[2][el] + red[3][el])) + ((red[4][el] + red[5][el]) + (red[6][el] + red[7][el]));
}

// ====================================================================================================================
int gens_fill_maps(const char* who, MapSet* ms, const float* const* feats, const int* hw, int n_levels);

static constexpr size_t bt_lds_bytes() {
    return sizeof(float) * 32 * (BT_S_RD + BT_S_D1 + BT_S_DFE + BT_S_H0 + BT_S_TB + BT_S_H + BT_S_TV + BT_S_HV + BT_S_HH + BT_S_TU + BT_S_T1 + BT_S_T2 +
                                 2 * BT_S_A + BT_S_SC + 3 + BT_S_DFE + BT_S_H + 4);
}

static int bt_fill(const char* who, BlendRaw* W, const float* const* w) {
    GENS_CHECK_ARG(w, GENS_EINVAL, "%s: null weight table", who);
    for (int k = 0; k < 23; ++k) GENS_CHECK_ARG(w[k], GENS_EINVAL, "%s: weight %d is null", who, k);
    W->rd1 = w[0]; W->rd1b = w[1]; W->rd2 = w[2]; W->rd2b = w[3];
    W->b1 = w[4]; W->b1b = w[5]; W->b2 = w[6]; W->b2b = w[7];
    W->v1 = w[8]; W->v1b = w[9]; W->v2 = w[10]; W->v2b = w[11];
    W->u1 = w[12]; W->u1b = w[13]; W->u2 = w[14]; W->u2b = w[15];
    W->r1 = w[16]; W->r1b = w[17]; W->r2 = w[18]; W->r2b = w[19]; W->r3 = w[20]; W->r3b = w[21];
    W->s = w[22];
    return 0;
}

extern "C" int gens_blend_train_acc_parts(int64_t n, int nv);
template <int MODE>      // 0 forward, 1 backward leaving operand rows, 2 backward with the weight-gradient sums inside (persistent workgroups)
static int bt_launch(const char* who, const float* const* feats, const int* hw, int n_levels, const float* imgs, const float* w2c, const float* intr,
                     const float* c2w, int nv, const float* const* weights, BlendTrainIO io, void* stream) {
    MapSet fs;
    GENS_CHECK_ARG(feats, GENS_EINVAL, "%s: null table", who);
    if (int e = gens_fill_maps(who, &fs, feats, hw, n_levels)) return e;
    GENS_CHECK_ARG(n_levels >= 1 && n_levels <= 5, GENS_ELIMIT, "%s: 1..5 feature levels (d_feature <= 20), got %d", who, n_levels);
    GENS_CHECK_ARG(nv >= 2 && nv <= GENS_MAX_VIEWS, GENS_ELIMIT, "%s: nv=%d not in 2..%d", who, nv, GENS_MAX_VIEWS);
    GENS_CHECK_ARG(imgs && w2c && intr && c2w, GENS_EINVAL, "%s: null camera / image pointer", who);
    BlendRaw W;
    if (int e = bt_fill(who, &W, weights)) return e;
    io.imgs = (const float4*)imgs;
    io.w2c = w2c; io.intr = intr; io.c2w = c2w; io.nv = nv;
    const int ppw = 32 / (nv - 1);
    unsigned grid = gens_blocks(io.n, ppw);
    if (MODE == 2) {
        io.n_tiles = (int)grid;
        grid = (unsigned)gens_blend_train_acc_parts(io.n, nv);
    }
    hipStream_t st = (hipStream_t)stream;
    static GensLdsOptIn once[6][3];
#define BT_LAUNCH(NL)                                                                                                              \
    {                                                                                                                              \
        if (int e_ = gens_lds_opt_in(once[NL][MODE], (const void*)blend_train_k<NL, MODE != 0, MODE == 2>, (int)bt_lds_bytes(), "gens_blend_train")) return e_; \
        blend_train_k<NL, MODE != 0, MODE == 2><<<grid, BT_THREADS, bt_lds_bytes(), st>>>(W, fs, io);                              \
    }
    switch (n_levels) {
        case 1: BT_LAUNCH(1) break;
        case 2: BT_LAUNCH(2) break;
        case 3: BT_LAUNCH(3) break;
        case 4: BT_LAUNCH(4) break;
        default: BT_LAUNCH(5) break;
    }
#undef BT_LAUNCH
    return gens_launch_status(who);
}

extern "C" int64_t gens_blend_train_rows(int64_t n, int nv) {
    if (n <= 0 || nv < 2) return 0;
    return (int64_t)gens_blocks(n, 32 / (nv - 1)) * 32;
}

extern "C" int gens_blend_train_fwd(const float* const* feats, const int* hw, int n_levels, const float* imgs, const float* w2c, const float* intr,
                                    const float* c2w, int nv, const float* const* weights, const float* pts, const int64_t* index, int64_t n,
                                    const int32_t* n_device, float* rgb_out, uint8_t* vis_out, void* stream) {
    GENS_CHECK_ARG(n >= 0 && (n == 0 || (pts && rgb_out)), GENS_EINVAL, "gens_blend_train_fwd: null pts / output");
    if (n == 0) return 0;
    BlendTrainIO io = {};
    io.pts = pts; io.index = index; io.n_dev = n_device; io.n = n; io.rgb_out = rgb_out; io.vis_out = vis_out;
    return bt_launch<0>("gens_blend_train_fwd", feats, hw, n_levels, imgs, w2c, intr, c2w, nv, weights, io, stream);
}

extern "C" int gens_blend_train_bwd(const float* const* feats, const int* hw, int n_levels, const float* imgs, const float* w2c, const float* intr,
                                    const float* c2w, int nv, const float* const* weights, const float* pts, const int64_t* index, int64_t n,
                                    const int32_t* n_device, const float* g_rgb, float* const* r_ops, float* const* l_ops, float* g_feat,
                                    float* s_part, void* stream) {
    GENS_CHECK_ARG(n >= 0 && (n == 0 || (pts && g_rgb && r_ops && l_ops && s_part)), GENS_EINVAL, "gens_blend_train_bwd: null pointer");
    if (n == 0) return 0;
    BlendTrainIO io = {};
    io.pts = pts; io.index = index; io.n_dev = n_device; io.n = n; io.g_rgb = g_rgb; io.g_feat = g_feat; io.s_part = s_part;
    for (int l = 0; l < BT_NLAYER; ++l) {
        GENS_CHECK_ARG(r_ops[l] && l_ops[l], GENS_EINVAL, "gens_blend_train_bwd: operand buffer %d is null", l);
        io.R[l] = r_ops[l];
        io.L[l] = l_ops[l];
    }
    return bt_launch<1>("gens_blend_train_bwd", feats, hw, n_levels, imgs, w2c, intr, c2w, nv, weights, io, stream);
}

// The backward launch with the weight-gradient sums inside (no operand rows): workgroups = gens_blend_train_acc_parts(n, nv), each leaves one
// block of gens_blend_train_acc_floats(n_levels) floats in `parts`; `cc` = their sum = what gens_gemm_tn_batch would have returned for the
// eleven products (the input of gens_blend_train_wgrad).
extern "C" int gens_blend_train_acc_parts(int64_t n, int nv) {
    if (n <= 0 || nv < 2) return 0;
    const int64_t tiles = gens_blocks(n, 32 / (nv - 1));
    int64_t cap = getenv("GENS_K18_PARTS") ? atoi(getenv("GENS_K18_PARTS")) : 512;   // two workgroups per CU (78 KB of LDS each); the switch: occupancy probes
    if (cap < 1) cap = 1;                                                            // ("0" or a non-number would be a grid of no workgroups: a failed launch)
    return (int)(tiles < cap ? tiles : cap);
}
extern "C" int gens_blend_train_acc_floats(int n_levels) {
    switch (n_levels) {
        case 1: return bt_cc_off(BT_NLAYER, 7);
        case 2: return bt_cc_off(BT_NLAYER, 11);
        case 3: return bt_cc_off(BT_NLAYER, 15);
        case 4: return bt_cc_off(BT_NLAYER, 19);
        case 5: return bt_cc_off(BT_NLAYER, 23);
        default: return 0;
    }
}
extern "C" int gens_blend_train_bwd_acc(const float* const* feats, const int* hw, int n_levels, const float* imgs, const float* w2c, const float* intr,
                                        const float* c2w, int nv, const float* const* weights, const float* pts, const int64_t* index, int64_t n,
                                        const int32_t* n_device, const float* g_rgb, float* g_feat, float* s_part, float* parts, float* cc, void* stream) {
    GENS_CHECK_ARG(n >= 0 && (n == 0 || (pts && g_rgb && s_part && parts && cc)), GENS_EINVAL, "gens_blend_train_bwd_acc: null pointer");
    if (n == 0) return 0;
    BlendTrainIO io = {};
    io.pts = pts; io.index = index; io.n_dev = n_device; io.n = n; io.g_rgb = g_rgb; io.g_feat = g_feat; io.s_part = s_part; io.acc_parts = parts;
    if (int e = bt_launch<2>("gens_blend_train_bwd_acc", feats, hw, n_levels, imgs, w2c, intr, c2w, nv, weights, io, stream)) return e;
    const int csz = gens_blend_train_acc_floats(n_levels);
    blend_train_reduce_k<<<gens_blocks(csz, 32), 256, 0, (hipStream_t)stream>>>(parts, gens_blend_train_acc_parts(n, nv), csz, cc);
    return gens_launch_status("gens_blend_train_bwd_acc");
}
