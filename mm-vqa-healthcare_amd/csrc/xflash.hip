// Image queries over text keys (dir 1 of the fused cross-attention sub-block) as ONE launch per layer for gfx950.
// Reference: m3ae/modules/language_encoders/bert_model.py:253-350 (cross branch :275-278), :353-364 (BertSelfOutput up to the
// LayerNorm), called at :480-488.  Algebra (xattn.hip): with K' = (k_h / sqrt(dh)) W_q,h and V' = v_h W_o[:, h]^T, both
// [B, R = H*T, D],
//     P = softmax over each head's T columns of (x K'^T + c),        s = drop(drop(P) V' + b_o) + x.
//
// A workgroup owns BM rows of x of one sample.  The scores S[BM x R] accumulate over D in registers (8 compute waves as
// 2 x 4, MFMA 16x16x32), the T-column softmax and the attention dropout run in the accumulator layout, the (dropped)
// probabilities go to LDS as bf16 -- stored directly in the image format of a K-contiguous A operand, [R/32 chunks][BM rows][64 B]
// swizzled -- and the second product streams V' in 256-column passes against that resident operand.  Scores and
// probabilities never touch HBM in a forward-only call; a training call copies P (and drop(P)) out of the LDS image for the
// backward (m3ae_xattn_bwd reads them).
//
// LDS (160 KiB):   product 1:  ring of 5 slots x (BM + R) x 64 B = 160 KiB      (32-deep chunks of x and K')
//                  product 2:  [P image: BM x R x 2 B = 96 KiB][ring: 4 x 16 KiB chunks of V' (32 k-rows x 256 cols)]; the
//                              epilogue's eight 2-KiB slabs live in the ring slot of a pass's last chunk
// Waves: 8 compute waves in two barrier-staggered groups (one group's MFMA cluster covers the other's fragment reads) + 4
// loader waves that only issue the LDS-DMA pieces -- the structure of xg_kernel (xattn.hip):
//   compute phase c:  [fragment reads of chunk c; lgkmcnt(0)]  barrier  [MFMAs]  barrier
//   loader  phase c:  [part of chunk c + DEPTH into the slot of chunk c - 1]  barrier  [the rest; counted vmcnt: chunk c + 1 landed]  barrier
// Measured (profiles/r03_xflash_*.txt): 337-363 us at B = 256 for 174 GFLOP; 256 us of it with no operand staged at
// all (the barrier-coupled structure), a tile's 66 us = prologue 5 + product 1 24 + softmax / image 5 + product 2 24 + epilogues 6.
#include "xattn_common.h"
#include <type_traits>

#ifndef XF_ST_AUX
#define XF_ST_AUX 2   // cache policy of the output rows (aux bits of buffer_store): 2 = nt, streaming (-0.8 % on the call, profiles/r04_xflash_nt_store.log)
#endif

namespace {

constexpr int XF_LDS = 160 * 1024;

#ifdef M3AE_XF_TRACE   // diagnostic build only (tools/xf_trace.py): per-workgroup phase stamps, never in the product build
__device__ uint64_t g_xf_trace[4096 * 3 * 16];   // [block][early wave 0 / late wave 4 / loader wave 8][16 x 100-MHz ticks]
#define XF_STAMP(k) do { if (lane == 0 && (wave & 3) == 0 && blockIdx.x < 4096) \
    g_xf_trace[((size_t)blockIdx.x * 3 + (wave >> 2)) * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
// in-loop cycle accumulators (shader clocks): slots 12..15 of the wave kind's row hold 4 sums for product 1, the row of
// g_xf_trace2 4 sums for product 2.  compute waves: [fragment reads + lgkmcnt wait | first barrier | MFMA cluster | second barrier];
// loader waves: [issue | first barrier | wait for the DMA | second barrier]
__device__ uint64_t g_xf_trace2[4096 * 3 * 4];
#define XF_CLK() ({ __builtin_amdgcn_sched_barrier(0); uint64_t t_ = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); t_; })
#define XF_ACC_DECL uint64_t xa_[4] = {0, 0, 0, 0}, xt_ = 0
#define XF_ACC_START() do { xt_ = XF_CLK(); } while (0)
#define XF_ACC(i) do { const uint64_t n_ = XF_CLK(); xa_[i] += n_ - xt_; xt_ = n_; } while (0)
#define XF_ACC_FLUSH(prod) do { if (lane == 0 && (wave & 3) == 0 && blockIdx.x < 4096) { for (int q_ = 0; q_ < 4; ++q_) { \
    if ((prod) == 1) g_xf_trace[((size_t)blockIdx.x * 3 + (wave >> 2)) * 16 + 12 + q_] = xa_[q_]; \
    else g_xf_trace2[((size_t)blockIdx.x * 3 + (wave >> 2)) * 4 + q_] = xa_[q_]; xa_[q_] = 0; } } } while (0)
#else
#define XF_STAMP(k) do { } while (0)
#define XF_ACC_DECL
#define XF_ACC_START() do { } while (0)
#define XF_ACC(i) do { } while (0)
#define XF_ACC_FLUSH(prod) do { } while (0)
#endif

template <int BM, int R, int T>
__global__ __launch_bounds__(768, 3) void xf1_kernel(XfArgs a) {
    if (a.has_drop) { drop_resolve(a.drop_a); drop_resolve(a.drop_h); }
    constexpr int WM = BM / 2, WN1 = R / 4, MI = WM / 16, NJ1 = WN1 / 16, TG = T / 16;
    constexpr int A1_BYTES = BM * 64, SLOT1 = (BM + R) * 64;
    constexpr int NSLOT1 = XF_LDS / SLOT1 >= 5 ? 5 : 3, DEPTH1 = NSLOT1 - 1;
    constexpr int GA1 = BM / 16 / 4, GB1 = R / 16 / 4, G1 = GA1 + GB1;
    constexpr int P_BYTES = BM * R * 2, PCH = BM * 64, KC2 = R / 32;
    constexpr int NJ2 = 4, SLOT2 = 32 * 256 * 2, NSLOT2 = 4, DEPTH2 = NSLOT2 - 1, G2 = 4;
    constexpr int RING2 = P_BYTES, SLAB_BYTES = 2048;   // the epilogue's 8 slabs live in the ring slot of a pass's last chunk
    static_assert(GA1 >= 1 && NJ1 % TG == 0 && MI >= 1, "tile shape");
    static_assert(NSLOT1 * SLOT1 <= XF_LDS && RING2 + NSLOT2 * SLOT2 <= XF_LDS && 8 * SLAB_BYTES <= SLOT2, "LDS budget");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const unsigned wg = xcd_remap(blockIdx.x, gridDim.x);   // an XCD walks a contiguous range: a sample's tiles share K' / V' in its L2
    const unsigned bi = wg / (unsigned)a.tiles_m;
    const int m0 = (int)(wg - bi * (unsigned)a.tiles_m) * BM;
    const int D = a.D, I = a.I;
    const int nc1 = D >> 5;                     // 32-deep chunks of product 1
    const int npass = D >> 8, nc2 = npass * KC2; // product 2: 256-column passes x R / 32 chunks
    const bf16_t* X = a.X + (int64_t)bi * I * D;
    const bf16_t* Kp = a.Kp + (int64_t)bi * R * D;
    const bf16_t* Vp = a.Vp + (int64_t)bi * R * D;
    const bool save_pd = a.P != nullptr && a.Pd != nullptr;   // a training call under dropout saves P and drop(P): two image copies

    XF_STAMP(0);
    if (wave >= 8) {
        // ------------------------------------------------------------------------------------------------ loader waves
        const int lw = wave - 8;
        // one chunk's pieces in two parts (the loader issues one part in each of a phase's two barrier intervals: a 1-KiB piece
        // costs the issuing wave ~130 clocks, and a loader that issues a whole chunk in one interval is that interval's critical path)
        auto stage_rows = [&](const bf16_t* G, int64_t row0, int64_t nrows, int c, char* tile, int q0, int q1) {
            for (int q = q0; q < q1; ++q) {   // piece q of this loader: 16 rows x 64 B (nt_stage_m's image)
                const int seg = q * 4 + lw;
                const int row = seg * 16 + (lane >> 2);
                const int chunk = (lane & 3) ^ nt_swz<32>(row);
                int64_t grow = row0 + row;
                grow = grow < nrows ? grow : nrows - 1;
                glds16(G + grow * D + (int64_t)c * 32 + chunk * 8, tile + seg * 1024);
            }
        };
        auto stage1 = [&](int c, char* slot, int part) {   // part 0 / 1: halves; part 2: whole
            const int kb0 = part == 1 ? GB1 / 2 : 0, kb1 = part == 0 ? GB1 / 2 : GB1;
            const int ka0 = part == 1 ? (GA1 + 1) / 2 : 0, ka1 = part == 0 ? (GA1 + 1) / 2 : GA1;
#ifndef XF_EXP_NO_KP   // XF_EXP_*: timing experiments only (operands left unstaged: wrong results), never in the product build
            stage_rows(Kp, 0, R, c, slot + A1_BYTES, kb0, kb1);
#endif
#ifndef XF_EXP_NO_X
            stage_rows(X, m0, I, c, slot, ka0, ka1);
#endif
        };
        auto stage2 = [&](int cc, char* slot, int part) {
            const int pass = cc / KC2, c = cc - pass * KC2;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                if (part != 2 && p != part) continue;
#ifndef XF_EXP_NO_VP
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    t_stage128(Vp, D, c * 32, R, pass * 256 + p * 128, D, slot + p * 8192, lw + 4 * h, lane);
#endif
            }
        };
        // Ring protocol (both products): at the start of phase k the loader tops the ring up to chunk k + DEPTH (the slot of chunk
        // k - 1: every compute wave retired its reads of it before the barrier that ends phase k - 1), and BETWEEN the phase's two
        // barriers it waits until chunk k + 1 has landed (the early group reads it after the second barrier): the DMA of a chunk
        // has DEPTH - 1/2 phases to land.
        auto wait_landed = [&](auto g, int outstanding) {   // all but the youngest `outstanding` chunks of this wave have landed
            constexpr int G = decltype(g)::value;
            if (outstanding >= 3) wait_vm<3 * G>();
            else if (outstanding == 2) wait_vm<2 * G>();
            else if (outstanding == 1) wait_vm<G>();
            else wait_vm<0>();
        };
        int issued = 0;
        for (; issued < DEPTH1 && issued < nc1; ++issued) stage1(issued, smem + issued * SLOT1, 2);
        wait_landed(std::integral_constant<int, G1>{}, issued - 1);
        PP_FENCE();
        __builtin_amdgcn_s_barrier();   // chunk 0 is in LDS
        PP_FENCE();
        int nxt = issued == NSLOT1 ? 0 : issued;
        XF_ACC_DECL;
        XF_ACC_START();
        for (int k = 0; k < nc1; ++k) {
            const bool more = issued < nc1;
            if (more) stage1(issued, smem + nxt * SLOT1, 0);
            XF_ACC(0);
            PP_FENCE();
            __builtin_amdgcn_s_barrier();
            PP_FENCE();
            XF_ACC(1);
            if (more) {
                stage1(issued, smem + nxt * SLOT1, 1);
                ++issued;
                nxt = nxt + 1 == NSLOT1 ? 0 : nxt + 1;
            }
            if (k + 1 < nc1) wait_landed(std::integral_constant<int, G1>{}, issued - (k + 2));
            XF_ACC(2);
            PP_FENCE();
            __builtin_amdgcn_s_barrier();
            PP_FENCE();
            XF_ACC(3);
        }
        XF_ACC_FLUSH(1);
        __builtin_amdgcn_s_barrier();   // A: the early group's re-alignment barrier; every read of product 1 is retired
        PP_FENCE();
        XF_STAMP(2);
        // product 2: the first chunks land while the compute waves run the softmax
        issued = 0;
        for (; issued < DEPTH2 && issued < nc2; ++issued) stage2(issued, smem + RING2 + issued * SLOT2, 2);
        wait_landed(std::integral_constant<int, G2>{}, issued - 1);
        PP_FENCE();
        if (a.P != nullptr) {           // the compute waves' image copies (training)
            __builtin_amdgcn_s_barrier();
            if (save_pd) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); }
        }
        XF_STAMP(3);                    // the loaders' first chunk of V' has landed
        __builtin_amdgcn_s_barrier();   // B: the P image is complete, chunk 0 of V' is in LDS
        PP_FENCE();
        XF_STAMP(4);
        nxt = issued == NSLOT2 ? 0 : issued;
        int kp = 0;                     // k modulo KC2
        XF_ACC_START();
        for (int k = 0; k < nc2; ++k) {
            // the first phase of a pass (but the first) leaves the slot of the previous pass's last chunk alone: both wave groups
            // run that pass's epilogue slabs in it until this phase's second barrier; the next phase issues two chunks
            int target = k + 1 + DEPTH2 - ((kp == 0 && k > 0) ? 1 : 0);
            target = target < nc2 ? target : nc2;
            const int n_new = target - issued;      // 0 (the phase after a pass's last chunk), 1, or 2 (the phase after that)
            if (n_new == 2) {                       // one whole chunk in each barrier interval
                stage2(issued, smem + RING2 + nxt * SLOT2, 2);
                ++issued;
                nxt = nxt + 1 == NSLOT2 ? 0 : nxt + 1;
            } else if (n_new == 1) {
                stage2(issued, smem + RING2 + nxt * SLOT2, 0);
            }
            XF_ACC(0);
            PP_FENCE();
            __builtin_amdgcn_s_barrier();
            PP_FENCE();
            XF_ACC(1);
            if (n_new >= 1) {
                stage2(issued, smem + RING2 + nxt * SLOT2, n_new == 2 ? 2 : 1);
                ++issued;
                nxt = nxt + 1 == NSLOT2 ? 0 : nxt + 1;
            }
            if (k + 1 < nc2) wait_landed(std::integral_constant<int, G2>{}, issued - (k + 2));
            XF_ACC(2);
            PP_FENCE();
            __builtin_amdgcn_s_barrier();
            PP_FENCE();
            XF_ACC(3);
            kp = kp + 1 == KC2 ? 0 : kp + 1;
        }
        XF_ACC_FLUSH(2);
        __builtin_amdgcn_s_barrier();   // the early group's re-alignment barrier
        XF_STAMP(11);
        return;
    }

    // --------------------------------------------------------------------------------------------------- compute waves
    const int wr = wave >> 2, wc = wave & 3;
    const bool late = wave >= 4;
    // fragment byte offsets: ONE per-lane register per operand, the 16-row steps as immediates.  nt_frag_off<32>'s swizzle depends on
    // (row >> 2) & 3 only, i.e. on lane & 15 (WM, WN1 and the 16-row steps are multiples of 16): fragment i sits 16 rows = 1024 B
    // behind fragment 0.  As MI + NJ1 separately computed offsets hipcc kept 10 registers live through both products and, at the 168
    // registers three waves per SIMD leave, spilled them: the product-2 loop reloaded three from scratch in EVERY chunk, each behind
    // s_waitcnt vmcnt(0) in the middle of its fragment reads (round-4 .s).
    static_assert(WM % 16 == 0 && WN1 % 16 == 0, "16-row fragment steps");
    const int aoff0 = nt_frag_off<32>(wr * WM + (lane & 15), lane >> 4);
    const int boff10 = A1_BYTES + nt_frag_off<32>(wc * WN1 + (lane & 15), lane >> 4);

    f32x4 acc[MI][NJ1];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ1; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    PP_FENCE();
    __builtin_amdgcn_s_barrier();   // chunk 0 is in LDS
    PP_FENCE();
    if (late) { __builtin_amdgcn_s_barrier(); PP_FENCE(); }
    XF_STAMP(1);

    // ---- product 1: S = x K'^T
    int cur = 0;
    XF_ACC_DECL;
    XF_ACC_START();
    for (int c = 0; c < nc1; ++c) {
        const char* At = smem + cur * SLOT1;
        s16x8 bfr[NJ1], af[MI];
#pragma unroll
        for (int j = 0; j < NJ1; ++j) bfr[j] = nt_frag_at(At, boff10 + j * 1024);
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = nt_frag_at(At, aoff0 + i * 1024);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // fragments in registers, the slot's reads retired, before the barrier
        XF_ACC(0);
        PP_FENCE();
        __builtin_amdgcn_s_barrier();
        PP_FENCE();
        XF_ACC(1);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ1; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8_t, bfr[j]), __builtin_bit_cast(bf16x8_t, af[i]), acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        XF_ACC(2);
        PP_FENCE();
        __builtin_amdgcn_s_barrier();
        PP_FENCE();
        XF_ACC(3);
        cur = cur + 1 == NSLOT1 ? 0 : cur + 1;
    }
    XF_ACC_FLUSH(1);
    if (!late) { __builtin_amdgcn_s_barrier(); PP_FENCE(); }   // A: every wave has executed the same number of barriers
    XF_STAMP(2);
    // every fragment read of product 1 is retired and none of its DMA is outstanding: the LDS below RING2 is free

    // ---- softmax over each head's T columns, in the accumulator layout:
    //      acc[i][j][r] = S[m = wr WM + 16 i + (lane & 15)][n = wc WN1 + 16 j + 4 (lane >> 4) + r]
    const int mrow = wr * WM + (lane & 15);          // + 16 i: tile-local row of this lane
    const int ncol = wc * WN1 + 4 * (lane >> 4);     // + 16 j: first of this lane's 4 columns
    {
        const float* cb = a.colbias + (int64_t)bi * R;
#pragma unroll
        for (int g = 0; g < NJ1 / TG; ++g) {
            f32x4 c4[TG];
#pragma unroll
            for (int t = 0; t < TG; ++t) c4[t] = *(const f32x4*)(cb + ncol + 16 * (g * TG + t));
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                float v[4 * TG];
#pragma unroll
                for (int t = 0; t < TG; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[4 * t + r] = (acc[i][g * TG + t][r] + c4[t][r]) * LOG2E;
                float mx = v[0];
#pragma unroll
                for (int r = 1; r < 4 * TG; ++r) mx = fmaxf(mx, v[r]);
                mx = quad16_max(mx);
                float sum = 0.f;
#pragma unroll
                for (int r = 0; r < 4 * TG; ++r) { v[r] = __builtin_amdgcn_exp2f(v[r] - mx); sum += v[r]; }
                sum = quad16_sum(sum);
                const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
                for (int t = 0; t < TG; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][g * TG + t][r] = v[4 * t + r] * inv;
            }
        }
    }
    XF_STAMP(3);
    // the image of a K-contiguous A operand: chunk n / 32 -> [BM rows][64 B], 16-B pieces swizzled as nt_frag_off<32>
    auto image_put = [&](bool dropped) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int row = mrow + 16 * i;
#pragma unroll
            for (int j = 0; j < NJ1; ++j) {
                const int n = ncol + 16 * j;
                float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                if (dropped) {   // attention-probability dropout (bert_model.py:334): index ((b H + h) Lq + q) ld + k
                    const uint64_t base = ((uint64_t)((int64_t)bi * a.H + n / T) * I + (m0 + row)) * a.drop_ld;
                    drop_apply4(a.drop_a, base + (n % T), v);
                }
                const int kk = n & 31;
                *(u32x2*)(smem + (n >> 5) * PCH + nt_frag_off<32>(row, kk >> 3) + ((kk & 4) << 1)) =
                    (u32x2){pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    // image -> global [I][R] rows (training saves): 16-B pieces, consecutive lanes along a row
    auto image_copy = [&](bf16_t* dst) {
        constexpr int PR = R / 8;
        bf16_t* base = dst + ((int64_t)bi * I + m0) * R;
        for (int idx = tid; idx < BM * PR; idx += 512) {
            const int row = idx / PR, p = idx - row * PR;
            const u32x4 v = *(const u32x4*)(smem + (p >> 2) * PCH + nt_frag_off<32>(row, p & 3));
            if (m0 + row < I) *(u32x4*)(base + (int64_t)row * R + p * 8) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    if (a.P != nullptr) {
        image_put(false);
        PP_FENCE(); __builtin_amdgcn_s_barrier(); PP_FENCE();
        image_copy(a.P);
        if (save_pd) {
            PP_FENCE(); __builtin_amdgcn_s_barrier(); PP_FENCE();   // every copy read is retired: the image may be overwritten
            image_put(true);
            PP_FENCE(); __builtin_amdgcn_s_barrier(); PP_FENCE();
            image_copy(a.Pd);
        }
    } else {
        image_put(a.has_drop != 0);
    }
    PP_FENCE();
    __builtin_amdgcn_s_barrier();   // B: the image is complete (every wave's stores retired), chunk 0 of V' is in LDS
    PP_FENCE();
    XF_STAMP(4);
    if (late) { __builtin_amdgcn_s_barrier(); PP_FENCE(); }

    // ---- product 2: out = drop(P) V' in 256-column passes; wave tile WM x 64
    f32x4 acc2[MI][NJ2];
    cur = 0;
    int cc = 0;
    XF_ACC_START();
    const int r8 = lane >> 3, cq = lane & 7;      // row-contiguous side of the epilogue: row r8 of a slab, columns 8 cq .. 8 cq + 7
    const int rows_here = I - m0 < BM ? I - m0 : BM;
    const int64_t tile_off = ((int64_t)bi * I + m0) * D;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.X + tile_off), 0, rows_here * D * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t s_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.S + tile_off), 0, rows_here * D * 2, 0x00020000);
    const int ep_voff = ((wr * WM + r8) * D + wc * 64 + 8 * cq) * 2;   // bytes from the tile's first row
    const int row8_bytes = 8 * D * 2;
    for (int pass = 0; pass < npass; ++pass) {
        // The epilogue's global operands (residual pieces, bias) are requested AHEAD of their use -- a dependent load in the epilogue
        // costs its whole latency, nothing else runs on the CU then: the first two row blocks' pieces go out before the pass's last
        // chunk, piece k + 2 at the start of step k (a window of three pieces: registers are 168 per lane with the loader waves).
        const int nw0 = pass * 256 + wc * 64;
        // buffer addressing: ONE per-lane byte offset (ep_voff) + a scalar offset per pass and step; rows beyond the sample's
        // last (the descriptor's range) read as zero and are not stored
        auto res_load = [&](int step) -> u32x4 {      // step = 2 i + half: rows wr WM + 8 step + r8
            return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, ep_voff, pass * 512 + step * row8_bytes, 0));
        };
        u32x4 rwin[3];
        f32x4 bo0, bo1;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ2; ++j) acc2[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // the V' fragment offsets are recomputed per pass from an opaque copy of the lane id: hoisted out of the pass loop they
        // stayed live through every pass epilogue (the kernel's register peak at 168) and were spilled -- five scratch reloads at
        // the top of EVERY chunk, each behind a vmcnt wait (round-4 .s)
        int boff2[NJ2][2];
        {
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));
#pragma unroll
            for (int j = 0; j < NJ2; ++j) {
                const int r = wc * 64 + j * 16;
                tn_frag_offs<128>(r & 127, lane_o, boff2[j][0], boff2[j][1]);
                boff2[j][0] += (r >> 7) * 8192;
                boff2[j][1] += (r >> 7) * 8192;
            }
        }
        for (int c = 0; c < KC2; ++c, ++cc) {
            const char* Bt = smem + RING2 + cur * SLOT2;
            const char* Pc = smem + c * PCH;
            s16x8 bfr[NJ2], af[MI];
            if (c == KC2 - 1) {
                rwin[0] = res_load(0);
                rwin[1] = res_load(1);
                bo0 = *(const f32x4*)(a.bo + nw0 + 8 * cq);
                bo1 = *(const f32x4*)(a.bo + nw0 + 8 * cq + 4);
            }
#pragma unroll
            for (int j = 0; j < NJ2; ++j) bfr[j] = tn_frag_at(Bt, boff2[j][0], boff2[j][1]);
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = nt_frag_at(Pc, aoff0 + i * 1024);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            XF_ACC(0);
            PP_FENCE();
            __builtin_amdgcn_s_barrier();
            PP_FENCE();
            XF_ACC(1);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ2; ++j)
                    acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8_t, bfr[j]), __builtin_bit_cast(bf16x8_t, af[i]), acc2[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            XF_ACC(2);
            PP_FENCE();
            __builtin_amdgcn_s_barrier();
            PP_FENCE();
            XF_ACC(3);
            cur = cur + 1 == NSLOT2 ? 0 : cur + 1;
        }
        XF_STAMP(5 + pass);
        // ---- pass epilogue (BertSelfOutput up to the LayerNorm, bert_model.py:360-363): s = dropout(acc + b_o) + x.
        // A wave-private [8 rows][64 cols] fp32 slab turns each 16 x 64 accumulator block into whole 128-B row segments (two
        // halves of 8 rows = 2 MI steps); bias, hidden dropout (index = (row of the [B*I, D] matrix) * D + n), residual add and
        // the 16-B store run on the row-contiguous side.  The slabs live in the ring slot of the pass's last chunk (every read of
        // it is retired; the loaders leave it alone until both wave groups are through, see their loop).  Step s + 1's slab
        // stores go out as soon as step s's reads are back, under step s's arithmetic.
        {
            char* slab = smem + RING2 + (cur == 0 ? NSLOT2 - 1 : cur - 1) * SLOT2 + wave * SLAB_BYTES;
            auto slab_put = [&](int step) {
                if (((lane >> 3) & 1) == (step & 1)) {   // lanes whose accumulator row (lane & 15) is in this half of the row block
                    const int row8 = lane & 7;
#pragma unroll
                    for (int j = 0; j < NJ2; ++j)
                        *(f32x4*)(slab + row8 * 256 + (((4 * j + (lane >> 4)) ^ row8) << 4)) = acc2[step >> 1][j];
                }
            };
            slab_put(0);
#pragma unroll
            for (int step = 0; step < 2 * MI; ++step) {
#pragma clang fp contract(off)     // x * 1/(1-p) + residual stays two roundings (as the GEMM epilogue's slab path: bit-identical)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const f32x4 v0 = *(const f32x4*)(slab + r8 * 256 + (((2 * cq) ^ r8) << 4));
                const f32x4 v1 = *(const f32x4*)(slab + r8 * 256 + (((2 * cq + 1) ^ r8) << 4));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slab may be overwritten
                if (step + 1 < 2 * MI) slab_put(step + 1);
                if (step + 2 < 2 * MI) rwin[(step + 2) % 3] = res_load(step + 2);
                const int rl = wr * WM + 8 * step + r8;              // tile-local row this lane finishes
                float x[8] = {v0[0] + bo0[0], v0[1] + bo0[1], v0[2] + bo0[2], v0[3] + bo0[3],
                              v1[0] + bo1[0], v1[1] + bo1[1], v1[2] + bo1[2], v1[3] + bo1[3]};
                if (a.has_drop) {
                    const uint64_t idx = (uint64_t)(((int64_t)bi * I + m0 + rl) * D + nw0 + 8 * cq);
                    drop_apply4(a.drop_h, idx, x);
                    drop_apply4(a.drop_h, idx + 4, x + 4);
                }
                const u32x4 rr = rwin[step % 3];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    x[2 * q] += __uint_as_float(rr[q] << 16);
                    x[2 * q + 1] += __uint_as_float(rr[q] & 0xffff0000u);
                }
                const u32x4 packed = {pack2bf(x[0], x[1]), pack2bf(x[2], x[3]), pack2bf(x[4], x[5]), pack2bf(x[6], x[7])};
                __builtin_amdgcn_raw_buffer_store_b128(packed, s_rsrc, ep_voff, pass * 512 + step * row8_bytes, XF_ST_AUX);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        XF_STAMP(8 + pass);
        XF_ACC_START();
    }
    XF_ACC_FLUSH(2);
    if (!late) { __builtin_amdgcn_s_barrier(); PP_FENCE(); }
    XF_STAMP(11);
}

template <int BM, int R, int T>
int launch_xf1(XfArgs a, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&xf1_kernel<BM, R, T>), hipFuncAttributeMaxDynamicSharedMemorySize, XF_LDS);
        attr_set = true;
    }
    a.tiles_m = (a.I + BM - 1) / BM;
    const unsigned grid = (unsigned)(a.B * a.tiles_m);
    hipLaunchKernelGGL((xf1_kernel<BM, R, T>), dim3(grid), dim3(768), XF_LDS, s, a);
    return hip_launch_status();
}

}  // namespace

#ifdef M3AE_XF_TRACE
extern "C" int m3ae_xf_trace_dump(uint64_t* host_out) {   // diagnostic build only: [4096 blocks][3 wave kinds][16 stamps], then [4096][3][4]
    const int rc = (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_xf_trace), sizeof(uint64_t) * 4096 * 3 * 16);
    if (rc) return rc;
    return (int)hipMemcpyFromSymbol(host_out + 4096 * 3 * 16, HIP_SYMBOL(g_xf_trace2), sizeof(uint64_t) * 4096 * 3 * 4);
}
#endif

int m3ae_xflash_dir1(const XfArgs& a, int T, hipStream_t s) {
    if (a.D % 256 != 0 || a.B <= 0 || a.I <= 0) return M3AE_ERR_UNSUPPORTED;
    if (T == 32 && a.H * T == 384) return launch_xf1<128, 384, 32>(a, s);
    if (T == 64 && a.H * T == 768) return launch_xf1<64, 768, 64>(a, s);
    return M3AE_ERR_UNSUPPORTED;
}
