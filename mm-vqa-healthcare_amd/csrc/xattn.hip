// Fused cross-attention sub-block of BertCrossLayer for gfx950 (north_star's kernel; reference:
// m3ae/modules/language_encoders/bert_model.py:253-350 (cross branch :275-278), :353-364, :480-488).
//
// Both directions of the co-attention have ONE long side (image, I = 577 tokens) and one short side (text, T = 32
// tokens), and H * T = 384 < D = 768.  The projection of the long side is therefore absorbed into the short side:
//
//   text queries  (dir 0, Lq = T, Lk = I):   S_h = Q_h K_h^T = (Q_h W_k,h) y^T + [row constant: drops out of softmax]
//        Q' = (q_h / sqrt(dh)) W_k,h   [B, T*H, D]           (rows r = t*H + h)
//        P  = softmax_rows(Q' y^T + mask)                    [B, T*H, I]     one GEMM per sample, K = D
//        Z  = drop(P) y                                      [B, T*H, D]     one GEMM per sample, K = I
//        ctx_h = Z_h W_v,h^T + rowsum(drop(P)) b_v,h         [B*T, D]
//   image queries (dir 1, Lq = I, Lk = T):   S_h = x (K_h W_q,h)^T + b_q,h K_h^T
//        K' = (k_h / sqrt(dh)) W_q,h   [B, H*T, D],  V' = v_h W_o[:, h]^T   [B, H*T, D]   (rows n = h*T + j)
//        P  = softmax_32-column-groups(x K'^T + c),  c[n] = b_q,h . k_h[j] / sqrt(dh) + mask[j]      [B, I, H*T]
//        out = drop(P) V' + b_o                              [B*I, D]        (the output dense is absorbed as well)
//
// so the 577-token K/V (dir 0) and Q/output (dir 1) projections -- 2 x 0.68 GFLOP per sample and direction -- become
// two [384 x 577 x 768] products (2 x 0.34 GFLOP), and the softmax runs in the epilogue of the score GEMM.  b_k drops
// out of the softmax exactly (it adds the same number to every key of a row).
//
// All big products run on ONE kernel template (xg_kernel): per-sample batched bf16 MFMA GEMM in the ping-pong
// structure of gemm_nt_pp_kernel (8 waves in two barrier-staggered groups, 32-deep chunks DMA-staged into an LDS ring,
// counted vmcnt), with the tile shape, the operand forms (K-contiguous rows or reduction-strided [K][cols] read with
// ds_read_b64_tr_b16) and the epilogue as template parameters.
#include "xattn_common.h"

namespace {

enum { FORM_K = 0, FORM_T = 1 };
enum { XE_STORE = 0, XE_DENSE = 1, XE_SOFTMAX32 = 2, XE_SOFTMAXROW = 3, XE_ATOMIC = 4, XE_DSOFT = 5 };

struct XgArgs {
    const bf16_t* A; int64_t lda, a_sb;    // FORM_K: [M][K] (lda = row stride); FORM_T: [K][M] (lda = stride of a k row)
    const bf16_t* B; int64_t ldb, b_sb;    // FORM_K: [N][K];                    FORM_T: [K][N]
    int M, N, K;                           // per batch item; FORM_K operands are readable (zero / finite) up to ceil32(K)
    int tiles_m, tiles_n;
    bf16_t* C; int64_t ldc, c_sb;
    bf16_t* C2;                            // softmax epilogues: the dropped probabilities (same layout), or nullptr
    int rdiv, rmul;                        // XE_STORE row map: crow = (m / rdiv) * rmul + m % rdiv   (rdiv = 0: identity)
    float alpha;
    const float* bias; int64_t bias_sb;    // [N] (+ batch * bias_sb)
    const float* rowscale; int64_t rs_sm, rs_sb;   // XE_STORE: bias is multiplied by rowscale[map(m) * rs_sm + batch * rs_sb]
    int rs_div, rs_mul;                    //           map(m) = (m / rs_div) * rs_mul + m % rs_div   (rs_div = 0: m)
    int b_rows;                            // rows of a K-contiguous B that exist (0: N); rows beyond are clamped duplicates
    const bf16_t* residual;                // XE_DENSE: [batch * M + m][ldc]
    const float* colbias; int64_t cb_sb;   // softmax epilogues: additive [batch][N] (nullptr: none)
    float* rowsum_out;                     // XE_SOFTMAXROW with dropout: [batch][M] row sums of the dropped probabilities
    int n_valid;                           // XE_SOFTMAXROW: columns >= n_valid are masked out (P = 0)
    DropState drop; int has_drop;
    int H, Lq, drop_ld;                    // dropout index convention of m3ae_attn_desc: ((b*H + h)*Lq + q) * ld + k
    // memory-row maps of the operands, row -> (row / div) * mul + row % div (div = 0: identity): the rows of a K-contiguous
    // operand, the REDUCTION rows of a reduction-strided one (per-head views of the [B][H*T][D] intermediates)
    int a_div, a_mul, b_div, b_mul;
    int ksplit, kchunks;                   // split of the reduction over workgroups (XE_ATOMIC): `kchunks` 32-deep chunks each
    const bf16_t* A2; const bf16_t* B2;    // second operand pair (same strides): the reduction continues over it from chunk
    int k_switch;                          //   `k_switch` on (0: none): C = A B + A2 B2 in one pass (K counts both; K2 = K - 32 k_switch)
    float* Cf;                             // XE_ATOMIC: fp32 [batch][M][ldc], += alpha * acc
    int accumulate;                        // XE_STORE: C += (bf16 read-modify-write)
    const bf16_t* P;                       // XE_DSOFT: the probabilities, laid out like C
    const float* delta; const float* radd; // XE_DSOFT: per-row [batch][M] softmax-backward term / addend to dP (nullptr: 32-column groups / 0)
    int drop_mode;                         // XE_DSOFT dropout index: 0: row = t*H + h, key = n;  1: head = n / tkeys, query = m, key = n % tkeys
    int tkeys;                             // XE_DSOFT without `delta`: columns per softmax group (32 or 64 text keys; 0 = 32)
    int trace_slot;                        // diagnostic builds (M3AE_XG_TRACE) only
};


#ifdef M3AE_XG_TRACE   // diagnostic build only (tools/xg_trace.py): per-workgroup timeline stamps, never in the product build
__device__ uint64_t g_xg_trace[8 * 16 * 4096];   // [launch slot][block][4 x 100-MHz ticks, 4 x shader clocks, 8 in-loop clocks]
#define XG_STAMP(k) do { if (tid == 0 && blockIdx.x < 4096) { uint64_t* t_ = g_xg_trace + ((size_t)a.trace_slot * 4096 + blockIdx.x) * 16; \
                                                             t_[(k)] = __builtin_amdgcn_s_memrealtime(); t_[4 + (k)] = __builtin_readcyclecounter(); } } while (0)
// in-loop stamps of chunk 8 (shader clocks), wave 0 of the workgroup
#define XG_LSTAMP(k) do { if (c == 8 && tid == 0 && blockIdx.x < 4096) g_xg_trace[((size_t)a.trace_slot * 4096 + blockIdx.x) * 16 + 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define XG_STAMP(k) do { } while (0)
#define XG_LSTAMP(k) do { } while (0)
#endif

// ---- epilogue plumbing: the wave's accumulators reach memory through a wave-private LDS slab, 32 rows at a time, so that
// every global access is 16 B per lane along a row (row-scattered 8-byte accesses straight from the MFMA layout run at
// ~7 B/clk/CU: measured 2-3x the main loop on these shapes).  A slab pass is [32 rows][WN cols] with 16 B of row padding;
// "piece" = 8 consecutive columns of one row; lane l handles pieces l, l + 64, ...
template <int WN, int ES> struct Slab {   // ES = element bytes (2: bf16, 4: fp32)
    static constexpr int RS = WN * ES + 16, BYTES = 32 * RS, PR = WN / 8, PPL = 32 * PR / 64;
    static_assert((32 * PR) % 64 == 0, "pieces per lane");
};
// lane's 4 values of row 16 ii + (lane & 15), columns 16 j + 4 (lane >> 4) .. + 3
template <int WN> DEVINL void slab_put_bf16(char* slab, int lane, int ii, int j, const float* v) {
    *(u32x2*)(slab + (16 * ii + (lane & 15)) * Slab<WN, 2>::RS + (16 * j + 4 * (lane >> 4)) * 2) =
        (u32x2){pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
}
template <int WN> DEVINL void slab_put_f32(char* slab, int lane, int ii, int j, f32x4 v) {
    *(f32x4*)(slab + (16 * ii + (lane & 15)) * Slab<WN, 4>::RS + (16 * j + 4 * (lane >> 4)) * 4) = v;
}
// store one bf16 slab pass: rowptr(r) = global address of column n_w of slab row r (nullptr: row out of range)
template <int WN, class RowPtr> DEVINL void slab_store_bf16(const char* slab, int lane, int ncols_ok, RowPtr rowptr) {
    using S = Slab<WN, 2>;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int t = 0; t < S::PPL; ++t) {
        const int piece = lane + 64 * t, row = piece / S::PR, c8 = piece - row * S::PR;
        const u32x4 v = *(const u32x4*)(slab + row * S::RS + c8 * 16);
        bf16_t* p = rowptr(row);
        if (p != nullptr && c8 * 8 < ncols_ok) *(u32x4*)(p + c8 * 8) = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slab may be overwritten by the next pass
}

// BM x BN output tile of batch item `bi`.  12 waves: 8 COMPUTE waves as WAVES_M x (8 / WAVES_M), each (BM / WAVES_M) x
// (BN / WAVES_N), in two groups (waves 0-3 / 4-7: one wave of each per SIMD) that run one barrier apart, so one group's
// MFMA cluster covers the other's fragment reads; and 4 LOADER waves (one per SIMD) that do nothing but request the
// 32-deep operand chunks (LDS-DMA, 1-KiB pieces) into an NSLOT-slot ring, NSLOT - 1 chunks ahead.
// Why loaders: the LDS-DMA path of a CU moves ~30 B/clk; when the compute waves issue the pieces themselves, every piece
// stalls its wave for ~135 clk (measured: 540 clk per wave and chunk at 4 pieces, in-kernel stamps) in the read section
// that the partner group's MFMAs are supposed to hide -- the section grows to twice the MFMA cluster and the chunk period
// with it (2200 clk against 2 x 384 of MFMA).  A loader's stalls cost nothing.
//   compute phase c:  [fragment reads of chunk c; lgkmcnt(0)]  barrier  [MI x NJ MFMAs]  barrier
//   loader  phase c:  [request chunk c + NSLOT - 1 into the slot of chunk c - 1; counted vmcnt: chunk c + 1 landed]  barrier  barrier
// Ring safety: the loaders run in step with the early group.  RAW: a loader's wait for chunk c + 1 precedes its first
// barrier of phase c; the early group reads chunk c + 1 after its second barrier of phase c, the late group later still.
// WAR: chunk c + NSLOT - 1 overwrites chunk c - 1, whose reads every compute wave retired (lgkmcnt(0)) BEFORE the first
// barrier of its phase c - 1; for the late group that barrier is the loaders' second barrier of phase c - 1.
// Lane layout of the accumulators (D'[n][m] orientation, as gemm_nt_pp_kernel): acc[i][j][r] =
// C[m = .. + 16 i + (lane & 15)][n = .. + 16 j + 4 (lane >> 4) + r].
//   NLOAD == 0 (the 128 x 640 whole-row score tile: 160 accumulators per lane leave no room for a third wave per SIMD): the 8
//   compute waves stage their own operands in gemm_nt_pp_kernel's two-phase schedule (B part of chunk c + NSLOT - 1 requested
//   in phase 2c, A part in phase 2c + 1, chunk consumed in two halves of the row blocks).
//   GEN = 0 (most launches): no memory-row maps and no second operand pair -- their address arithmetic (an integer division per
//   staged piece) and branches compile out of the staging code, which sits in every chunk's read phase.
template <int BM, int BN, int WAVES_M, int NSLOT, int NLOAD, int AFORM, int BFORM, int EPI, int GEN>
__global__ __launch_bounds__(NLOAD ? 768 : 512, NLOAD ? 3 : 2) void xg_kernel(XgArgs a) {
    if (a.has_drop) drop_resolve(a.drop);
    constexpr int WAVES_N = 8 / WAVES_M, WM = BM / WAVES_M, WN = BN / WAVES_N, MI = WM / 16, NJ = WN / 16;
    constexpr int NST = NLOAD ? NLOAD : 8;   // waves that issue the LDS-DMA pieces
    // pieces per staging wave and chunk; an operand with fewer pieces than staging waves (BM = 64 with 8 of them) is staged by
    // the first waves only and left out of the counted vmcnt (those waves then wait for a little more than they must)
    constexpr int GA = BM / 16 / NST, GB = BN / 16 / NST, G = GA + GB, DEPTH = NSLOT - 1;
    static_assert(GB >= 1 && (BM / 16) % (GA ? NST : 1) == 0 && (BN / 16) % NST == 0, "pieces per staging wave");
    constexpr int A_BYTES = BM * 64, SLOT = (BM + BN) * 64;
    static_assert(BM % 64 == 0 && BN % 64 == 0 && WM % 16 == 0 && WN % 16 == 0 && DEPTH >= 2 && DEPTH <= 3 && (NLOAD == 0 || NLOAD == 4), "tile shape");
    static_assert((AFORM == FORM_K || BM % 128 == 0) && (BFORM == FORM_K || BN % 128 == 0), "T-form operands come in 128-column panels");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const unsigned per_t = (unsigned)(a.tiles_m * a.tiles_n), per_b = per_t * (unsigned)(a.ksplit > 1 ? a.ksplit : 1);
    const unsigned wg = xcd_remap(blockIdx.x, gridDim.x);   // an XCD walks a contiguous range: a sample's tiles share its L2
    const unsigned bi = wg / per_b, t2 = wg - bi * per_b;
    const unsigned ks = t2 / per_t, tt = t2 - ks * per_t;
    const int m0 = (int)(tt / a.tiles_n) * BM, n0 = (int)(tt % a.tiles_n) * BN;
    const int nc_all = (a.K + 31) >> 5;
    const int c_first = a.ksplit > 1 ? (int)ks * a.kchunks : 0;   // this workgroup's chunks of the reduction
    const int nc = a.ksplit > 1 ? (nc_all - c_first < a.kchunks ? nc_all - c_first : a.kchunks) : nc_all;
    XG_STAMP(0);

    const bf16_t* A = a.A + (int64_t)bi * a.a_sb;
    const bf16_t* B = a.B + (int64_t)bi * a.b_sb;
    const int lw = NLOAD ? wave - 8 : wave;   // index among the staging waves
    const bf16_t* A2 = (GEN && a.A2) ? a.A2 + (int64_t)bi * a.a_sb : nullptr;
    const bf16_t* B2 = (GEN && a.B2) ? a.B2 + (int64_t)bi * a.b_sb : nullptr;
    const int k_switch = GEN ? a.k_switch : 0;
    const int a_div = GEN ? a.a_div : 0, a_mul = GEN ? a.a_mul : 0, b_div = GEN ? a.b_div : 0, b_mul = GEN ? a.b_mul : 0;
    const int k1_rows = k_switch ? k_switch * 32 : a.K;     // reduction rows of the first pair (T-form bound)
    auto stage_b = [&](int c, char* slot) {
        int cc = c + c_first;
        const bool second = k_switch && cc >= k_switch;
        const bf16_t* Bp = second ? B2 : B;
        const int kend = second ? a.K - k_switch * 32 : k1_rows;
        if (second) cc -= k_switch;
        if constexpr (BFORM == FORM_K) nt_stage_m<GB, NST>(Bp, a.ldb, n0, a.b_rows ? a.b_rows : a.N, (int64_t)cc * 32, slot + A_BYTES, lw, lane, b_div, b_mul);
        else {
#pragma unroll
            for (int p = 0; p < BN / 128; ++p)
#pragma unroll
                for (int h = 0; h < 8 / NST; ++h)
                    t_stage128(Bp, a.ldb, cc * 32, kend, n0 + p * 128, a.N, slot + A_BYTES + p * 8192, lw + NST * h, lane, b_div, b_mul);
        }
    };
    auto stage_a = [&](int c, char* slot) {
        int cc = c + c_first;
        const bool second = k_switch && cc >= k_switch;
        const bf16_t* Ap = second ? A2 : A;
        const int kend = second ? a.K - k_switch * 32 : k1_rows;
        if (second) cc -= k_switch;
        if constexpr (AFORM == FORM_K) {
            if constexpr (GA >= 1) nt_stage_m<GA, NST>(Ap, a.lda, m0, a.M, (int64_t)cc * 32, slot, lw, lane, a_div, a_mul);
            else if (lw < BM / 16) nt_stage_m<1, NST>(Ap, a.lda, m0, a.M, (int64_t)cc * 32, slot, lw, lane, a_div, a_mul);
        } else {
#pragma unroll
            for (int p = 0; p < BM / 128; ++p)
#pragma unroll
                for (int h = 0; h < 8 / NST; ++h)
                    t_stage128(Ap, a.lda, cc * 32, kend, m0 + p * 128, a.M, slot + p * 8192, lw + NST * h, lane, a_div, a_mul);
        }
    };
    auto stage = [&](int c, char* slot) { stage_b(c, slot); stage_a(c, slot); };

    if (NLOAD > 0 && wave >= 8) {
        // ------------------------------------------------------------------------------------------------ loader waves
#pragma unroll
        for (int c = 0; c < DEPTH; ++c)
            if (c < nc) stage(c, smem + c * SLOT);
        wait_vm_chunks<G>((nc < DEPTH ? nc : DEPTH) - 1);
        PP_FENCE();
        __builtin_amdgcn_s_barrier();   // chunk 0 is in LDS
        PP_FENCE();
        int nxt = DEPTH;
        for (int c = 0; c < nc; ++c) {
            if (c + DEPTH < nc) stage(c + DEPTH, smem + nxt * SLOT);
            const int rem = nc - 1 - c;
            wait_vm_chunks<G>(rem >= 1 ? (rem - 1 < DEPTH - 1 ? rem - 1 : DEPTH - 1) : 0);
            PP_FENCE();
            __builtin_amdgcn_s_barrier();
            PP_FENCE();
            __builtin_amdgcn_s_barrier();
            PP_FENCE();
            nxt = nxt + 1 == NSLOT ? 0 : nxt + 1;
        }
        __builtin_amdgcn_s_barrier();   // the early group's re-alignment barrier
        return;                         // ended waves are not counted by the epilogue's barriers
    }

    // --------------------------------------------------------------------------------------------------- compute waves
    const int wr = wave / WAVES_N, wc = wave % WAVES_N;
    const bool late = wave >= 4;
    // per-lane fragment offsets inside a slot (loop-invariant)
    int aoff[MI][2], boff[NJ][2];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int r = wr * WM + i * 16;
        if constexpr (AFORM == FORM_K) { aoff[i][0] = nt_frag_off<32>(r + (lane & 15), lane >> 4); aoff[i][1] = 0; }
        else { tn_frag_offs<128>(r & 127, lane, aoff[i][0], aoff[i][1]); aoff[i][0] += (r >> 7) * 8192; aoff[i][1] += (r >> 7) * 8192; }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int r = wc * WN + j * 16;
        if constexpr (BFORM == FORM_K) { boff[j][0] = A_BYTES + nt_frag_off<32>(r + (lane & 15), lane >> 4); boff[j][1] = 0; }
        else { tn_frag_offs<128>(r & 127, lane, boff[j][0], boff[j][1]); boff[j][0] += A_BYTES + (r >> 7) * 8192; boff[j][1] += A_BYTES + (r >> 7) * 8192; }
    }
    auto frag_a = [&](const char* slot, int i) -> s16x8 {
        if constexpr (AFORM == FORM_K) return nt_frag_at(slot, aoff[i][0]);
        else return tn_frag_at(slot, aoff[i][0], aoff[i][1]);
    };
    auto frag_b = [&](const char* slot, int j) -> s16x8 {
        if constexpr (BFORM == FORM_K) return nt_frag_at(slot, boff[j][0]);
        else return tn_frag_at(slot, boff[j][0], boff[j][1]);
    };

    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if constexpr (NLOAD == 0) {
#pragma unroll
        for (int c = 0; c < DEPTH; ++c)
            if (c < nc) stage(c, smem + c * SLOT);
        wait_vm_chunks<G>((nc < DEPTH ? nc : DEPTH) - 1);
    }
    PP_FENCE();
    __builtin_amdgcn_s_barrier();   // chunk 0 is in LDS
    PP_FENCE();
    if (late) { __builtin_amdgcn_s_barrier(); PP_FENCE(); }
    XG_STAMP(1);

    int cur = 0;
    if constexpr (NLOAD == 0) {
        constexpr int HI = MI / 2;
        static_assert(MI % 2 == 0, "two phases per chunk");
        int nxt = DEPTH;
        for (int c = 0; c < nc; ++c) {
            const char* At = smem + cur * SLOT;
            char* nx = smem + nxt * SLOT;
            const bool more = c + DEPTH < nc;
            s16x8 bfr[NJ], af[HI];
            // ---------------- phase 2c: first half of the row blocks; B part of chunk c + DEPTH (slot of chunk c - 1: last read in phase 2c - 2)
#pragma unroll
            for (int j = 0; j < NJ; ++j) bfr[j] = frag_b(At, j);
#pragma unroll
            for (int i = 0; i < HI; ++i) af[i] = frag_a(At, i);
            if (more) stage_b(c + DEPTH, nx);
            PP_FENCE();
            __builtin_amdgcn_s_barrier();
            PP_FENCE();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < HI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8_t, bfr[j]), __builtin_bit_cast(bf16x8_t, af[i]), acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            PP_FENCE();
            __builtin_amdgcn_s_barrier();
            PP_FENCE();
            // ---------------- phase 2c + 1: second half; A part of chunk c + DEPTH (last read in phase 2c - 1); chunk c + 1 landed
#pragma unroll
            for (int i = 0; i < HI; ++i) af[i] = frag_a(At, HI + i);
            if (more) stage_a(c + DEPTH, nx);
            {
                const int rem = nc - 1 - c;
                wait_vm_chunks<G>(rem >= 1 ? (rem - 1 < DEPTH - 1 ? rem - 1 : DEPTH - 1) : 0);
            }
            PP_FENCE();
            __builtin_amdgcn_s_barrier();
            PP_FENCE();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < HI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[HI + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8_t, bfr[j]), __builtin_bit_cast(bf16x8_t, af[i]), acc[HI + i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            PP_FENCE();
            __builtin_amdgcn_s_barrier();
            PP_FENCE();
            cur = cur + 1 == NSLOT ? 0 : cur + 1;
            nxt = nxt + 1 == NSLOT ? 0 : nxt + 1;
        }
    } else
    for (int c = 0; c < nc; ++c) {
        const char* At = smem + cur * SLOT;
        s16x8 bfr[NJ], af[MI];
        XG_LSTAMP(0);
#pragma unroll
        for (int j = 0; j < NJ; ++j) bfr[j] = frag_b(At, j);
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = frag_a(At, i);
        XG_LSTAMP(1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // fragments in registers, the slot's reads retired, before the barrier
        XG_LSTAMP(2);
        PP_FENCE();
        __builtin_amdgcn_s_barrier();
        PP_FENCE();
        XG_LSTAMP(3);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8_t, bfr[j]), __builtin_bit_cast(bf16x8_t, af[i]), acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        XG_LSTAMP(4);
        PP_FENCE();
        __builtin_amdgcn_s_barrier();
        PP_FENCE();
        XG_LSTAMP(5);
        cur = cur + 1 == NSLOT ? 0 : cur + 1;
    }
    if (!late) { __builtin_amdgcn_s_barrier(); PP_FENCE(); }   // re-align: every wave has executed the same number of barriers
    // every fragment read is retired and no DMA is outstanding: the ring is free
    XG_STAMP(2);

    const int mw0 = m0 + wr * WM;                         // first row of the wave's sub-tile
    const int nw0 = n0 + wc * WN;                         // first column
    const int mw = mw0 + (lane & 15);                     // + 16 i: this lane's row in the accumulator layout
    const int nw = nw0 + 4 * (lane >> 4);                 // + 16 j
    const int ncols_ok = a.N - nw0;                       // columns of the sub-tile inside the matrix (multiple of 8)
    constexpr int SLAB_STRIDE = Slab<WN, (EPI == XE_DENSE || EPI == XE_ATOMIC) ? 4 : 2>::BYTES;   // per-wave slab region (XE_DSOFT: 16-row fp32 passes, half of the fp32 size = the bf16 size)
    char* slab = smem + wave * SLAB_STRIDE;
    static_assert(8 * SLAB_STRIDE + 4 * WAVES_N * BM <= NSLOT * SLOT, "slabs + reduction scratch fit the ring");
    float* red = (float*)(smem + 8 * SLAB_STRIDE);        // [WAVES_N][BM] cross-wave reduction scratch (XE_SOFTMAXROW)

    if constexpr (EPI == XE_STORE) {
        bf16_t* C = a.C + (int64_t)bi * a.c_sb;
        const float* bias = a.bias ? a.bias + (int64_t)bi * a.bias_sb : nullptr;
#pragma unroll
        for (int ps = 0; ps < MI / 2; ++ps) {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int i = 2 * ps + ii, m = mw + 16 * i;
                const float rs = (a.rowscale && m < a.M) ? a.rowscale[map_row(m, a.rs_div, a.rs_mul) * a.rs_sm + (int64_t)bi * a.rs_sb] : 1.0f;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int n = nw + 16 * j;
                    float v[4] = {acc[i][j][0] * a.alpha, acc[i][j][1] * a.alpha, acc[i][j][2] * a.alpha, acc[i][j][3] * a.alpha};
                    if (bias && n < a.N) {
                        const f32x4 b4 = *(const f32x4*)(bias + n);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = fmaf(b4[r], rs, v[r]);
                    }
                    slab_put_bf16<WN>(slab, lane, ii, j, v);
                }
            }
            auto rowptr = [&](int row) -> bf16_t* {
                const int m = mw0 + 32 * ps + row;
                if (m >= a.M) return nullptr;
                const int64_t crow = a.rdiv ? (int64_t)(m / a.rdiv) * a.rmul + (m % a.rdiv) : (int64_t)m;
                return C + crow * a.ldc + nw0;
            };
            if (!a.accumulate) slab_store_bf16<WN>(slab, lane, ncols_ok, rowptr);
            else {   // C += : bf16 read-modify-write along the rows (second product into the same output)
                using S2 = Slab<WN, 2>;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int t = 0; t < S2::PPL; ++t) {
                    const int piece = lane + 64 * t, row = piece / S2::PR, c8 = piece - row * S2::PR;
                    const u32x4 v = *(const u32x4*)(slab + row * S2::RS + c8 * 16);
                    bf16_t* p = rowptr(row);
                    if (p != nullptr && c8 * 8 < ncols_ok) {
                        const u32x4 o = *(const u32x4*)(p + c8 * 8);
                        u32x4 r;
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            r[q] = pack2bf(__uint_as_float(v[q] << 16) + __uint_as_float(o[q] << 16),
                                           __uint_as_float(v[q] & 0xffff0000u) + __uint_as_float(o[q] & 0xffff0000u));
                        *(u32x4*)(p + c8 * 8) = r;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
    } else if constexpr (EPI == XE_DENSE) {
        // BertSelfOutput (bert_model.py:360-364) up to the LayerNorm: dropout(dense) + residual; dropout index = the GEMM
        // epilogue's (row of the [B * M, N] matrix) * N + n.  bias and dropout in the accumulator layout, the residual add
        // on the row-contiguous side of the (fp32) slab; the residual pieces are requested before the slab passes
        using S = Slab<WN, 4>;
#pragma unroll
        for (int ps = 0; ps < MI / 2; ++ps) {
            u32x4 res[S::PPL];   // this pass's residual pieces, requested before the slab round trip
#pragma unroll
            for (int t = 0; t < S::PPL; ++t) {
                const int piece = lane + 64 * t, row = piece / S::PR, c8 = piece - row * S::PR;
                const int m = mw0 + 32 * ps + row;
                res[t] = (u32x4){0u, 0u, 0u, 0u};
                if (a.residual && m < a.M && c8 * 8 < ncols_ok)
                    res[t] = *(const u32x4*)(a.residual + ((int64_t)bi * a.M + m) * a.ldc + nw0 + c8 * 8);
            }
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int i = 2 * ps + ii;
                const int64_t gm = (int64_t)bi * a.M + mw + 16 * i;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int n = nw + 16 * j;
                    float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                    if (a.bias && n < a.N) {
                        const f32x4 b4 = *(const f32x4*)(a.bias + n);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += b4[r];
                    }
                    if (a.has_drop) drop_apply4(a.drop, (uint64_t)(gm * a.N + n), v);
                    slab_put_f32<WN>(slab, lane, ii, j, (f32x4){v[0], v[1], v[2], v[3]});
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int t = 0; t < S::PPL; ++t) {
                const int piece = lane + 64 * t, row = piece / S::PR, c8 = piece - row * S::PR;
                const f32x4 v0 = *(const f32x4*)(slab + row * S::RS + c8 * 32);
                const f32x4 v1 = *(const f32x4*)(slab + row * S::RS + c8 * 32 + 16);
                const u32x4 rr = res[t];
                float x[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    x[2 * q] += __uint_as_float(rr[q] << 16);
                    x[2 * q + 1] += __uint_as_float(rr[q] & 0xffff0000u);
                }
                const int m = mw0 + 32 * ps + row;
                if (m < a.M && c8 * 8 < ncols_ok)
                    *(u32x4*)(a.C + ((int64_t)bi * a.M + m) * a.ldc + nw0 + c8 * 8) =
                        (u32x4){pack2bf(x[0], x[1]), pack2bf(x[2], x[3]), pack2bf(x[4], x[5]), pack2bf(x[6], x[7])};
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    } else if constexpr (EPI == XE_ATOMIC) {
        // split-K weight gradients: fp32 += alpha * acc, through the slab so that a wave-instruction adds two 128-B row
        // segments (the full-rate shape of global_atomic_add_f32; 64 scattered dwords run ~17x slower)
        using S = Slab<WN, 4>;
        float* Cf = a.Cf + (int64_t)bi * a.c_sb;
#pragma unroll
        for (int ps = 0; ps < MI / 2; ++ps) {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const f32x4 v = acc[2 * ps + ii][j];
                    slab_put_f32<WN>(slab, lane, ii, j, (f32x4){v[0] * a.alpha, v[1] * a.alpha, v[2] * a.alpha, v[3] * a.alpha});
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            static_assert(WN % 32 == 0, "32-float row segments");
#pragma unroll 4
            for (int t = 0; t < 32 * WN / 64; ++t) {
                const int e = lane + 64 * t, seg = e >> 5, row = seg / (WN / 32), col = (seg % (WN / 32)) * 32 + (e & 31);
                const int m = mw0 + 32 * ps + row, n = nw0 + col;
                const float v = *(const float*)(slab + row * S::RS + col * 4);
                if (m < a.M && n < a.N) atomicAdd(Cf + (int64_t)m * a.ldc + n, v);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    } else if constexpr (EPI == XE_DSOFT) {
        // softmax backward in the row-contiguous layout: acc = dL/d(dropped probabilities); dS = P (dP - delta) with
        // dP = keep / (1 - p) (acc + radd[row]) and delta = sum_k P dP over the softmax's extent -- given per row (whole-row
        // softmax: delta = dZ . Z + radd * rowsum, the flash-attention identity, computed by xattn_rowdot_kernel) or summed
        // here over each group of `tkeys` columns (4 or 8 adjacent lanes of the row layout; WN is a multiple of tkeys: the caller
        // picks the tile).  16-row fp32 slab passes.
        const int tk = a.tkeys ? a.tkeys : 32;
        constexpr int RS = WN * 4 + 16, PR = WN / 8, PPL = 16 * PR / 64;
        static_assert((16 * PR) % 64 == 0 && 16 * RS <= SLAB_STRIDE * 2 && PR % 4 == 0, "16-row fp32 slab pass");
        char* slab16 = smem + wave * (16 * RS);
        static_assert(8 * 16 * RS <= NSLOT * SLOT, "slabs fit the ring");
        bf16_t* dS = a.C + (int64_t)bi * a.c_sb;
        const bf16_t* Pp = a.P + (int64_t)bi * a.c_sb;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                *(f32x4*)(slab16 + (lane & 15) * RS + (16 * j + 4 * (lane >> 4)) * 4) = acc[i][j];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int t = 0; t < PPL; ++t) {
                const int piece = lane + 64 * t, row = piece / PR, c8 = piece - row * PR;
                const int m = mw0 + 16 * i + row, n = nw0 + c8 * 8;
                const bool ok = m < a.M && n < a.N;
                const f32x4 v0 = *(const f32x4*)(slab16 + row * RS + c8 * 32);
                const f32x4 v1 = *(const f32x4*)(slab16 + row * RS + c8 * 32 + 16);
                float dp[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                u32x4 pr = {0u, 0u, 0u, 0u};
                if (ok) pr = *(const u32x4*)(Pp + (int64_t)m * a.ldc + n);
                float p[8];
#pragma unroll
                for (int q = 0; q < 4; ++q) { p[2 * q] = __uint_as_float(pr[q] << 16); p[2 * q + 1] = __uint_as_float(pr[q] & 0xffff0000u); }
                const int64_t gr = (int64_t)bi * a.M + m;
                const float radd = (a.radd && ok) ? a.radd[gr] : 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) dp[e] += radd;
                if (a.has_drop) {
                    uint64_t base;
                    int k0;
                    if (a.drop_mode == 0) { base = ((uint64_t)((int64_t)bi * a.H + (m % a.H)) * a.Lq + (m / a.H)) * a.drop_ld; k0 = n; }
                    else { base = ((uint64_t)((int64_t)bi * a.H + n / tk) * a.Lq + m) * a.drop_ld; k0 = n % tk; }
                    drop_apply4(a.drop, base + k0, dp);
                    drop_apply4(a.drop, base + k0 + 4, dp + 4);
                }
                float dl;
                if (a.delta) dl = ok ? a.delta[gr] : 0.f;
                else {
                    float part = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) part = fmaf(p[e], dp[e], part);
                    part += __shfl_xor(part, 1, 64);
                    part += __shfl_xor(part, 2, 64);
                    if (tk == 64) part += __shfl_xor(part, 4, 64);
                    dl = part;
                }
                if (ok) {
                    float ds_[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) ds_[e] = p[e] * (dp[e] - dl);
                    *(u32x4*)(dS + (int64_t)m * a.ldc + n) =
                        (u32x4){pack2bf(ds_[0], ds_[1]), pack2bf(ds_[2], ds_[3]), pack2bf(ds_[4], ds_[5]), pack2bf(ds_[6], ds_[7])};
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    } else if constexpr (EPI == XE_SOFTMAX32) {
        // image queries: column n = h * 32 + j; one softmax per (row, head) = one pair of 16-column blocks of this wave
        static_assert(NJ % 2 == 0, "heads are pairs of 16-column blocks");
        bf16_t* P = a.C + (int64_t)bi * a.c_sb;
        bf16_t* Pd = a.C2 ? a.C2 + (int64_t)bi * a.c_sb : nullptr;
        const float* cb = a.colbias ? a.colbias + (int64_t)bi * a.cb_sb : nullptr;
#pragma unroll
        for (int g = 0; g < NJ / 2; ++g) {
            const int n = nw + 32 * g;        // this lane's columns: n .. n + 3 and n + 16 .. n + 19
            f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
            if (cb && n < a.N) { c0 = *(const f32x4*)(cb + n); c1 = *(const f32x4*)(cb + n + 16); }   // N % 32 == 0
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                float v[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = (acc[i][2 * g][r] * a.alpha + c0[r]) * LOG2E;
                    v[4 + r] = (acc[i][2 * g + 1][r] * a.alpha + c1[r]) * LOG2E;
                }
                float mx = v[0];
#pragma unroll
                for (int r = 1; r < 8; ++r) mx = fmaxf(mx, v[r]);
                mx = quad16_max(mx);
                float sum = 0.f;
#pragma unroll
                for (int r = 0; r < 8; ++r) { v[r] = __builtin_amdgcn_exp2f(v[r] - mx); sum += v[r]; }
                sum = quad16_sum(sum);
                const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
                for (int r = 0; r < 4; ++r) { acc[i][2 * g][r] = v[r] * inv; acc[i][2 * g + 1][r] = v[4 + r] * inv; }
            }
        }
#pragma unroll
        for (int out = 0; out < 2; ++out) {
            if (out == 1 && Pd == nullptr) break;
            bf16_t* dst = out == 0 ? P : Pd;
#pragma unroll
            for (int ps = 0; ps < MI / 2; ++ps) {
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) {
                    const int i = 2 * ps + ii, m = mw + 16 * i;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                        if (out == 1) {   // attention-probability dropout (bert_model.py:334): index ((b H + h) Lq + q) ld + k
                            const int n = nw + 16 * j;
                            const uint64_t base = ((uint64_t)((int64_t)bi * a.H + (n >> 5)) * a.Lq + m) * a.drop_ld;
                            drop_apply4(a.drop, base + (n & 31), v);
                        }
                        slab_put_bf16<WN>(slab, lane, ii, j, v);
                    }
                }
                slab_store_bf16<WN>(slab, lane, ncols_ok, [&](int row) -> bf16_t* {
                    const int m = mw0 + 32 * ps + row;
                    return m < a.M ? dst + (int64_t)m * a.ldc + nw0 : nullptr;
                });
            }
        }
    } else if constexpr (EPI == XE_SOFTMAXROW) {
        // text queries: the tile holds whole rows (n0 == 0, BN >= n_valid); a row is spread over the WAVES_N waves of a
        // wave row: max and sum go through LDS
        bf16_t* P = a.C + (int64_t)bi * a.c_sb;
        bf16_t* Pd = a.C2 ? a.C2 + (int64_t)bi * a.c_sb : nullptr;
        const float* cb = a.colbias ? a.colbias + (int64_t)bi * a.cb_sb : nullptr;
        const int rl = wr * WM + (lane & 15);          // row inside the tile, + 16 i
        float mx[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) mx[i] = -INFINITY;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = nw + 16 * j;
            float c4[4] = {0.f, 0.f, 0.f, 0.f};
            if (cb) {   // additive key mask [batch][n_valid] (n_valid is odd in general: scalar loads)
#pragma unroll
                for (int r = 0; r < 4; ++r) if (n + r < a.n_valid) c4[r] = cb[n + r];
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = (n + r < a.n_valid) ? (acc[i][j][r] * a.alpha + c4[r]) * LOG2E : -INFINITY;
                    acc[i][j][r] = v;
                    mx[i] = fmaxf(mx[i], v);
                }
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            mx[i] = quad16_max(mx[i]);
            if (lane < 16) red[wc * BM + rl + 16 * i] = mx[i];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            float m_ = red[rl + 16 * i];
#pragma unroll
            for (int w = 1; w < WAVES_N; ++w) m_ = fmaxf(m_, red[w * BM + rl + 16 * i]);
            mx[i] = m_;
        }
        __syncthreads();
        float sm[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            float s_ = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __builtin_amdgcn_exp2f(acc[i][j][r] - mx[i]);
                    acc[i][j][r] = e;
                    s_ += e;
                }
            s_ = quad16_sum(s_);
            if (lane < 16) red[wc * BM + rl + 16 * i] = s_;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            float s_ = red[rl + 16 * i];
#pragma unroll
            for (int w = 1; w < WAVES_N; ++w) s_ += red[w * BM + rl + 16 * i];
            sm[i] = __builtin_amdgcn_rcpf(s_);
        }
        float dsum[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) dsum[i] = 0.f;
#pragma unroll
        for (int out = 0; out < 2; ++out) {
            if (out == 1 && Pd == nullptr) break;
            bf16_t* dst = out == 0 ? P : Pd;
#pragma unroll
            for (int ps = 0; ps < MI / 2; ++ps) {
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) {
                    const int i = 2 * ps + ii, m = mw + 16 * i;   // row r = t * H + h of the sample
                    const uint64_t base = ((uint64_t)((int64_t)bi * a.H + (m % a.H)) * a.Lq + (m / a.H)) * a.drop_ld;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        float v[4];
                        if (out == 0) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) { v[r] = acc[i][j][r] * sm[i]; acc[i][j][r] = v[r]; }
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r];
                            drop_apply4(a.drop, base + (nw + 16 * j), v);
                            dsum[i] += (v[0] + v[1]) + (v[2] + v[3]);
                        }
                        slab_put_bf16<WN>(slab, lane, ii, j, v);
                    }
                }
                slab_store_bf16<WN>(slab, lane, WN, [&](int row) -> bf16_t* {   // every column up to BN: the zero padding is read as K
                    const int m = mw0 + 32 * ps + row;
                    return m < a.M ? dst + (int64_t)m * a.ldc + nw0 : nullptr;
                });
            }
        }
        if (Pd && a.rowsum_out) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const float s_ = quad16_sum(dsum[i]);
                if (lane < 16) red[wc * BM + rl + 16 * i] = s_;
            }
            __syncthreads();
            if (wc == 0 && lane < 16) {
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    float s_ = red[rl + 16 * i];
#pragma unroll
                    for (int w = 1; w < WAVES_N; ++w) s_ += red[w * BM + rl + 16 * i];
                    const int m = mw + 16 * i;
                    if (m < a.M) a.rowsum_out[(int64_t)bi * a.M + m] = s_;
                }
            }
        }
    }
#ifdef M3AE_XG_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    XG_STAMP(3);
#endif
}

#ifdef M3AE_XG_TRACE
int g_trace_next = 0;
#endif
template <int BM, int BN, int WAVES_M, int NSLOT, int NLOAD, int AFORM, int BFORM, int EPI, int GEN = 0>
int launch_xg(XgArgs a, int nbatch, hipStream_t s) {
    if (!GEN && (a.a_div || a.b_div || a.k_switch || a.A2 || a.B2)) return M3AE_ERR_ARG;   // needs the GEN = 1 instantiation
    constexpr int lds = NSLOT * (BM + BN) * 64;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&xg_kernel<BM, BN, WAVES_M, NSLOT, NLOAD, AFORM, BFORM, EPI, GEN>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.N + BN - 1) / BN;
#ifdef M3AE_XG_TRACE
    a.trace_slot = g_trace_next++ & 7;
#endif
    const unsigned grid = (unsigned)(nbatch * a.tiles_m * a.tiles_n * (a.ksplit > 1 ? a.ksplit : 1));
    hipLaunchKernelGGL((xg_kernel<BM, BN, WAVES_M, NSLOT, NLOAD, AFORM, BFORM, EPI, GEN>), dim3(grid), dim3(NLOAD ? 768 : 512), lds, s, a);
    return hip_launch_status();
}

// ---- the per-head "absorbed operand" builds (K', V', Q', dZ): C[map(m)][n] = alpha * sum_{d < 64} A[m][h dh + d] B[n][h dh + d], one
// batch item per head, M = B*T rows.  The reduction is TWO 32-deep chunks, so xg_kernel's ring never reaches steady state on
// it: a 128 x 384 tile there is a DMA round trip, 2 MFMA clusters and 96 KiB of stores in strict sequence on a CU that holds
// one workgroup (128 KiB ring) -- 51-54 us for 151 MB of output.  This kernel stages the whole reduction of a BM x BN tile at
// once ((BM + BN) * 128 B of LDS), 4 waves, and is sized so that OCC workgroups share a CU: one's stores overlap the others'
// operand round trips.  Products and their order are xg_kernel's (chunk 0 then chunk 1 into the same accumulator): bit-identical.
// PAIR: the launch builds TWO operands from two column ranges of the same projection buffer -- batch items [0, H) are the heads
// of (A, B, C, alpha), items [H, 2H) those of (A2, B2, C2, alpha2): K' and V' of the image-query direction -- and the workgroups
// of the first column tile of set 0 also write the query-bias vector c[b][h T + j] = scale (b_q,h . k_h[j]) + mask[b][j] from
// the key tile they hold in LDS (the separate launch read the projection a second time for it).
struct XbPair {
    const bf16_t* A2; const bf16_t* B2; bf16_t* C2; float alpha2;
    const float* bq; const float* mask; float* cb;   // cb == nullptr: no bias vector
    int T, H; float scale;
};
template <int BM, int BN, int WAVES_M, int OCC, int PAIR>
__global__ __launch_bounds__(256, OCC) void xbuild_kernel(XgArgs a, XbPair pr) {
    constexpr int WAVES_N = 4 / WAVES_M, WM = BM / WAVES_M, WN = BN / WAVES_N, MI = WM / 16, NJ = WN / 16;
    constexpr int GA = BM / 64, GB = BN / 64, G = GA + GB;   // 1-KiB pieces per wave and chunk
    constexpr int A_BYTES = BM * 64, SLOT = (BM + BN) * 64;
    static_assert(BM % 64 == 0 && BN % 64 == 0 && MI % 2 == 0 && WN % 16 == 0, "tile shape");
    static_assert(4 * Slab<WN, 2>::BYTES <= 2 * SLOT, "slabs fit the operand tiles");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned per_t = (unsigned)(a.tiles_m * a.tiles_n);
    const unsigned wg = xcd_remap(blockIdx.x, gridDim.x);   // an XCD walks a contiguous range: a head's weight slice stays in its L2
    const unsigned bi = wg / per_t, tt = wg - bi * per_t;
    const int m0 = (int)(tt / a.tiles_n) * BM, n0 = (int)(tt % a.tiles_n) * BN;
    const bool second = PAIR && bi >= (unsigned)pr.H;
    const unsigned head = second ? bi - (unsigned)pr.H : bi;
    const bf16_t* A = (second ? pr.A2 : a.A) + (int64_t)head * a.a_sb;
    const bf16_t* B = (second ? pr.B2 : a.B) + (int64_t)head * a.b_sb;
    const float alpha = second ? pr.alpha2 : a.alpha;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        nt_stage_m<GB, 4>(B, a.ldb, n0, a.N, (int64_t)c * 32, smem + c * SLOT + A_BYTES, wave, lane, 0, 0);
        nt_stage_m<GA, 4>(A, a.lda, m0, a.M, (int64_t)c * 32, smem + c * SLOT, wave, lane, 0, 0);
    }
    const int wr = wave / WAVES_N, wc = wave % WAVES_N;
    int aoff[MI], boff[NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i) aoff[i] = nt_frag_off<32>(wr * WM + i * 16 + (lane & 15), lane >> 4);
#pragma unroll
    for (int j = 0; j < NJ; ++j) boff[j] = A_BYTES + nt_frag_off<32>(wc * WN + j * 16 + (lane & 15), lane >> 4);
    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        if (c == 0) wait_vm<G>(); else wait_vm<0>();
        PP_FENCE();
        __builtin_amdgcn_s_barrier();   // every wave's pieces of chunk c are in LDS
        PP_FENCE();
        const char* At = smem + c * SLOT;
        s16x8 bfr[NJ], af[MI];
#pragma unroll
        for (int j = 0; j < NJ; ++j) bfr[j] = nt_frag_at(At, boff[j]);
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = nt_frag_at(At, aoff[i]);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8_t, bfr[j]), __builtin_bit_cast(bf16x8_t, af[i]), acc[i][j], 0, 0, 0);
    }
    if (PAIR && !second && n0 == 0 && pr.cb != nullptr) {
        // both chunks of the key tile are in LDS (second barrier above): two lanes per row, 32 channels each
        static_assert(!PAIR || BM == 128, "two lanes per tile row");
        const int row = tid >> 1, half = tid & 1, m = m0 + row;
        const float* bq = pr.bq + head * 64 + half * 32;
        float s_ = 0.f;
#pragma unroll 1
        for (int g = 0; g < 4; ++g) {
            const s16x8 kv = nt_frag_at(smem + half * SLOT, nt_frag_off<32>(row, g));
#pragma unroll
            for (int e = 0; e < 8; ++e) s_ = fmaf(__uint_as_float((uint32_t)(uint16_t)kv[e] << 16), bq[g * 8 + e], s_);
        }
        s_ += __shfl_xor(s_, 1, 64);
        if (half == 0 && m < a.M) {
            const int b = m / pr.T, j = m - b * pr.T;
            pr.cb[(int64_t)b * pr.H * pr.T + head * pr.T + j] = s_ * pr.scale + (pr.mask ? pr.mask[m] : 0.f);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    PP_FENCE();
    __builtin_amdgcn_s_barrier();       // every fragment read is retired: the tiles become the waves' slabs
    PP_FENCE();
    const int mw0 = m0 + wr * WM, nw0 = n0 + wc * WN;
    const int ncols_ok = a.N - nw0;
    char* slab = smem + wave * Slab<WN, 2>::BYTES;
    bf16_t* C = (second ? pr.C2 : a.C) + (int64_t)head * a.c_sb;
#pragma unroll
    for (int ps = 0; ps < MI / 2; ++ps) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int i = 2 * ps + ii;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                float v[4] = {acc[i][j][0] * alpha, acc[i][j][1] * alpha, acc[i][j][2] * alpha, acc[i][j][3] * alpha};
                slab_put_bf16<WN>(slab, lane, ii, j, v);
            }
        }
        slab_store_bf16<WN>(slab, lane, ncols_ok, [&](int row) -> bf16_t* {
            const int m = mw0 + 32 * ps + row;
            if (m >= a.M) return nullptr;
            const int64_t crow = a.rdiv ? (int64_t)(m / a.rdiv) * a.rmul + (m % a.rdiv) : (int64_t)m;
            return C + crow * a.ldc + nw0;
        });
    }
}

#ifndef M3AE_XBUILD_SHAPE
#define M3AE_XBUILD_SHAPE 128, 192, 2, 3
#endif
template <int BM, int BN, int WAVES_M, int OCC, int PAIR>
int launch_xbuild_t(XgArgs a, const XbPair& pr, int nbatch, hipStream_t s) {
    if (a.K != 64 || a.bias || a.rowscale || a.accumulate || a.a_div || a.b_div || a.k_switch) return M3AE_ERR_ARG;
    constexpr int lds = 2 * (BM + BN) * 64;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&xbuild_kernel<BM, BN, WAVES_M, OCC, PAIR>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.N + BN - 1) / BN;
    hipLaunchKernelGGL((xbuild_kernel<BM, BN, WAVES_M, OCC, PAIR>), dim3((unsigned)(nbatch * a.tiles_m * a.tiles_n)), dim3(256), lds, s, a, pr);
    return hip_launch_status();
}
// K = dh = 64 is the only head size of the one-launch image-query path (m3ae_xattn_supported); other head sizes keep the general template
int launch_xbuild(const XgArgs& a, int nbatch, hipStream_t s) {
    if (a.K == 64) return launch_xbuild_t<M3AE_XBUILD_SHAPE, 0>(a, XbPair{}, nbatch, s);
    return launch_xg<128, 384, 2, 4, 4, FORM_K, FORM_K, XE_STORE>(a, nbatch, s);
}
// two operands of one projection buffer (+ the query-bias vector) in ONE launch; K == 64 only
int launch_xbuild_pair(const XgArgs& a, const XbPair& pr, hipStream_t s) {
    return launch_xbuild_t<M3AE_XBUILD_SHAPE, 1>(a, pr, 2 * pr.H, s);
}

// c[b][h * T + j] = scale * sum_d bq[h dh + d] * k[b T + j][h dh + d] + mask[b][j]      (dir 1: the query bias against the keys)
// One wave per text row (b, j): a lane multiplies 4 consecutive channels of each 256-channel third of the row, 64 / 4 = 16
// adjacent lanes hold one head (dh = 64): four xor-shuffles, lane 16 g of third c writes head 4 c + g.  (Round 2 ran one THREAD
// per (row, head) with strided 128-B reads: 10 us at B = 256; this form: the row is read once, 16 B per lane.)
__global__ __launch_bounds__(256) void xattn_colbias_kernel(const bf16_t* k, int64_t ldk, const float* bq, const float* mask,
                                                           float* cb, int rows, int T, int H, int dh, float scale) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int b = row / T, j = row - b * T, D = H * dh;
    const bf16_t* kr = k + (int64_t)row * ldk;
    const float mk = mask ? mask[row] : 0.f;
    if (dh == 64 && D % 256 == 0) {
        for (int c0 = 0; c0 < D; c0 += 256) {
            const int e = c0 + 4 * lane;
            float x[4];
            ld_bf4(kr + e, x);
            const f32x4 q = *(const f32x4*)(bq + e);
            float s_ = x[0] * q[0] + x[1] * q[1] + x[2] * q[2] + x[3] * q[3];
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) s_ += __shfl_xor(s_, o, 64);
            if ((lane & 15) == 0) cb[(int64_t)b * H * T + (e / dh) * T + j] = s_ * scale + mk;
        }
        return;
    }
    for (int h = lane; h < H; h += 64) {   // any other head width: one lane per head
        float s_ = 0.f;
        for (int d = 0; d < dh; ++d) s_ = fmaf(bf2f(kr[h * dh + d]), bq[h * dh + d], s_);
        cb[(int64_t)b * H * T + h * T + j] = s_ * scale + mk;
    }
}


// dcb[b][n] = scale * sum_i dS[b][i][n]   (dir 1: gradient of the per-column score bias c).  grid (R / 128, B); a workgroup is
// 16 column groups of 8 (16-B loads) x 16 row lanes, four rows in flight per lane.
__global__ __launch_bounds__(256) void xattn_colsum_kernel(const bf16_t* dS, float* out, int I, int R, float scale) {
    __shared__ float part[16][16][9];
    const int b = blockIdx.y, cg = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int n0 = blockIdx.x * 128 + cg * 8;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (n0 < R) {
        const bf16_t* p = dS + (int64_t)b * I * R + n0;
        auto add = [&](const u32x4 v) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { acc[2 * q] += __uint_as_float(v[q] << 16); acc[2 * q + 1] += __uint_as_float(v[q] & 0xffff0000u); }
        };
        int i = rl;
        for (; i + 48 < I; i += 64) {
            u32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *(const u32x4*)(p + (int64_t)(i + 16 * u) * R);
#pragma unroll
            for (int u = 0; u < 4; ++u) add(v[u]);
        }
        for (; i < I; i += 16) add(*(const u32x4*)(p + (int64_t)i * R));
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) part[rl][cg][t] = acc[t];
    __syncthreads();
    if (rl == 0 && n0 < R) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            float s_ = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) s_ += part[r][cg][t];
            out[(int64_t)b * R + n0 + t] = scale * s_;
        }
    }
}

// out[h dh + d] += sum_m w[(m / T) R + (m % T) w_t + h w_h] * X[m][h dh + d]      (bias gradients of the absorbed projections;
// w == nullptr: weights 1).  grid (H, row blocks of 256), block 256 = dh columns x (256 / dh) row lanes.
__global__ __launch_bounds__(256) void xattn_headvec_kernel(const float* w, const bf16_t* X, int64_t ldx, float* out, int Mrows,
                                                            int T, int R, int w_t, int w_h, int dh) {
    __shared__ float part[256];
    const int h = blockIdx.x, d = threadIdx.x % dh, sub = threadIdx.x / dh, nsub = 256 / dh;
    const int r0 = blockIdx.y * 256, r1 = r0 + 256 < Mrows ? r0 + 256 : Mrows;
    float acc = 0.f;
    if (sub < nsub)
        for (int m = r0 + sub; m < r1; m += nsub) {
            const float wt = w ? w[(int64_t)(m / T) * R + (m % T) * w_t + h * w_h] : 1.0f;
            acc = fmaf(wt, bf2f(X[(int64_t)m * ldx + h * dh + d]), acc);
        }
    part[threadIdx.x] = acc;
    __syncthreads();
    if (sub == 0) {
        for (int q = 1; q < nsub; ++q) acc += part[q * dh + d];
        atomicAdd(out + h * dh + d, acc);
    }
}

// dir 0, one wave per row r = t H + h of sample b:  radd[b][r] = dctx[b T + t][h dh ..] . bv[h dh ..]  (gradient wrt the row sum of the
// dropped probabilities; only with dropout), delta[b][r] = dZ[b][r][:] . Z[b][r][:] + radd * rowsum
__global__ __launch_bounds__(256) void xattn_rowdot_kernel(const bf16_t* dZ, const bf16_t* Z, const bf16_t* dctx, const float* bv,
                                                           const float* rowsum, float* delta, float* radd, int64_t rows, int D,
                                                           int T, int H, int dh) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const bf16_t* a = dZ + row * D;
    const bf16_t* z = Z + row * D;
    float acc = 0.f;
    for (int c = lane * 8; c < D; c += 512) {
        const u32x4 x = *(const u32x4*)(a + c), y = *(const u32x4*)(z + c);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc = fmaf(__uint_as_float(x[q] << 16), __uint_as_float(y[q] << 16), acc);
            acc = fmaf(__uint_as_float(x[q] & 0xffff0000u), __uint_as_float(y[q] & 0xffff0000u), acc);
        }
    }
    acc = wave_sum(acc);
    float ra = 0.f;
    if (radd) {
        const int R = T * H, r = (int)(row % R), b = (int)(row / R), t = r / H, h = r % H;
        float s_ = 0.f;
        for (int d = lane; d < dh; d += 64) s_ = fmaf(bf2f(dctx[((int64_t)b * T + t) * D + h * dh + d]), bv[h * dh + d], s_);
        ra = wave_sum(s_);
    }
    if (lane == 0) {
        if (radd) radd[row] = ra;
        delta[row] = acc + (radd ? ra * rowsum[row] : 0.f);
    }
}

}  // namespace

#ifdef M3AE_XG_TRACE
extern "C" int m3ae_xg_trace_dump(uint64_t* host_out) {   // diagnostic build only: [8 launch slots][4096 blocks][8]; restarts the slots
    g_trace_next = 0;
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_xg_trace), sizeof(uint64_t) * 8 * 16 * 4096);
}
#endif

extern "C" int m3ae_xattn_supported(const m3ae_xattn_desc* d) {
    if (!d) return 0;
    const int64_t T = d->dir == 0 ? d->Lq : d->Lk, I = d->dir == 0 ? d->Lk : d->Lq;
    if (d->H <= 0 || d->D % d->H != 0) return 0;
    const int64_t dh = d->D / d->H;
    if (I < 1 || dh % 32 != 0 || d->D % 128 != 0) return 0;
    if (d->dir == 0) return ((T == 32 || T == 64) && I <= 640) ? 1 : 0;   // text queries: the whole-row score tile covers 640 keys
    // image queries (xflash.hip): 32 text keys (fine-tuning) or 64 (pre-training), any number of image tokens
    if (d->launch_flags & M3AE_XATTN_LEGACY_CHAIN) return (T == 32 && I <= 640) ? 1 : 0;
    return ((T == 32 || T == 64) && d->H * T == (T == 32 ? 384 : 768) && d->D % 256 == 0) ? 1 : 0;
}

extern "C" int m3ae_xattn_bwd_supported(const m3ae_xattn_desc* d) {
    if (!m3ae_xattn_supported(d)) return 0;
    return 1;   // round 3: 32 or 64 text tokens in both directions (the softmax-backward epilogue sums 32- or 64-column groups)
}

extern "C" int64_t m3ae_xattn_probs_ld(const m3ae_xattn_desc* d) {   // row stride (elements) of `probs` / `probs_drop`
    return d->dir == 0 ? 640 : d->H * d->Lk;
}

#define XCHK(e) do { const int rc_ = (e); if (rc_ != 0) return rc_; } while (0)

extern "C" int m3ae_xattn_fwd(const m3ae_xattn_desc* dp, void* stream) {
    if (!dp || !m3ae_xattn_supported(dp)) return M3AE_ERR_UNSUPPORTED;
    const m3ae_xattn_desc& d = *dp;
    hipStream_t s = (hipStream_t)stream;
    const int B = (int)d.B, H = (int)d.H, D = (int)d.D, dh = D / H;
    const int Lq = (int)d.Lq, Lk = (int)d.Lk;
    const float scale = 1.0f / sqrtf((float)dh);
    const bool drop = d.dropout_p > 0.f;
    if (!d.x || !d.y || !d.proj || !d.prime || !d.s || !d.out) return M3AE_ERR_ARG;
    if (d.dir == 0 && (!d.probs || (drop && !d.probs_drop))) return M3AE_ERR_ARG;

    m3ae_gemm_desc g{};
    g.batch1 = g.batch2 = 1;
    g.dtype_a = g.dtype_b = g.dtype_c = M3AE_BF16;
    g.alpha = 1.0f;
    g.a_sk = g.b_sk = g.c_sn = 1;
    g.launch_flags = (d.launch_flags & M3AE_XATTN_NO_PERSISTENT) ? M3AE_GEMM_NO_PERSISTENT : 0;

    if (d.dir == 0) {
        const int T = Lq, I = Lk, R = T * H;
        if (!d.zctx || !d.ctx || (drop && !d.rowsum)) return M3AE_ERR_ARG;
        // q = x Wq^T + bq                                                         (bert_model.py:263)
        g.M = (int64_t)B * T; g.N = D; g.K = D;
        g.A = d.x; g.a_sm = D; g.B = d.wq; g.b_sn = D; g.C = d.proj; g.c_sm = D; g.bias = d.bq;
        XCHK(m3ae_gemm(&g, stream));
        {   // Q'[b, t*H + h, :] = scale * q_h Wk_h: one batch item per head, M = B*T rows
            XgArgs a{};
            a.A = (const bf16_t*)d.proj; a.lda = D; a.a_sb = dh;
            a.B = (const bf16_t*)d.wkv_t; a.ldb = 2 * D; a.b_sb = dh;      // Wk^T [c][h dh + d]
            a.M = B * T; a.N = D; a.K = dh;
            a.C = (bf16_t*)d.prime; a.ldc = (int64_t)H * D; a.c_sb = D;
            a.alpha = scale;
            XCHK(launch_xbuild(a, H, s));
        }
        {   // P = softmax(Q' y^T + mask) per sample, whole rows per tile
            XgArgs a{};
            a.A = (const bf16_t*)d.prime; a.lda = D; a.a_sb = (int64_t)R * D;
            a.B = (const bf16_t*)d.y; a.ldb = D; a.b_sb = (int64_t)I * D;
            a.M = R; a.N = I; a.K = D;
            a.C = (bf16_t*)d.probs; a.ldc = 640; a.c_sb = (int64_t)R * 640;
            a.C2 = drop ? (bf16_t*)d.probs_drop : nullptr;
            a.alpha = 1.0f;
            a.colbias = d.key_mask; a.cb_sb = I;
            a.rowsum_out = drop ? d.rowsum : nullptr;
            a.n_valid = I;
            a.has_drop = drop; a.drop = make_drop(d.dropout_p, d.seed_attn, d.dropout_salt);
            a.H = H; a.Lq = T; a.drop_ld = (int)drop_ld(I);
            XCHK((launch_xg<128, 640, 2, 3, 0, FORM_K, FORM_K, XE_SOFTMAXROW>(a, B, s)));
        }
        {   // Z = drop(P) y
            XgArgs a{};
            a.A = (const bf16_t*)(drop ? d.probs_drop : d.probs); a.lda = 640; a.a_sb = (int64_t)R * 640;
            a.B = (const bf16_t*)d.y; a.ldb = D; a.b_sb = (int64_t)I * D;
            a.M = R; a.N = D; a.K = I;
            a.C = (bf16_t*)d.zctx; a.ldc = D; a.c_sb = (int64_t)R * D;
            a.alpha = 1.0f;
            XCHK((launch_xg<128, 384, 2, 4, 4, FORM_K, FORM_T, XE_STORE>(a, B, s)));
        }
        {   // ctx[b*T + t, h dh + d] = Z[b, t*H + h, :] . Wv[h dh + d, :] + rowsum * bv
            XgArgs a{};
            a.A = (const bf16_t*)d.zctx; a.lda = (int64_t)H * D; a.a_sb = D;
            a.B = (const bf16_t*)d.wkv + (int64_t)D * D; a.ldb = D; a.b_sb = (int64_t)dh * D;
            a.M = B * T; a.N = dh; a.K = D;
            a.C = (bf16_t*)d.ctx; a.ldc = D; a.c_sb = dh;
            a.alpha = 1.0f;
            a.bias = d.bkv + D; a.bias_sb = dh;
            if (drop) { a.rowscale = d.rowsum; a.rs_sm = H; a.rs_sb = 1; }
            XCHK((launch_xg<384, 128, 4, 4, 4, FORM_K, FORM_K, XE_STORE>(a, H, s)));
        }
        // s = dropout(ctx Wo^T + bo) + x                                          (bert_model.py:361-363)
        m3ae_gemm_desc o = g;
        o.M = (int64_t)B * T; o.N = D; o.K = D;
        o.A = d.ctx; o.a_sm = D; o.B = d.wo; o.b_sn = D; o.C = d.s; o.c_sm = D; o.bias = d.bo; o.residual = d.x;
        o.dropout_p = d.dropout_p; o.dropout_seed = d.seed_hidden; o.dropout_salt = d.dropout_salt;
        XCHK(m3ae_gemm(&o, stream));
    } else {
        const int I = Lq, T = Lk, R = H * T;
        if (!d.colbias) return M3AE_ERR_ARG;
        // k | v = y Wkv^T + bkv                                                   (bert_model.py:276-277)
        g.M = (int64_t)B * T; g.N = 2 * D; g.K = D;
        g.A = d.y; g.a_sm = D; g.B = d.wkv; g.b_sn = D; g.C = d.proj; g.c_sm = 2 * D; g.bias = d.bkv;
        XCHK(m3ae_gemm(&g, stream));
        bf16_t* Kp = (bf16_t*)d.prime;
        bf16_t* Vp = Kp + (int64_t)B * R * D;
        bool fused_build = false;
        {   // K'[b, h*T + j, :] = scale * k_h Wq_h
            XgArgs a{};
            a.A = (const bf16_t*)d.proj; a.lda = 2 * D; a.a_sb = dh;
            a.B = (const bf16_t*)d.wq_t; a.ldb = D; a.b_sb = dh;             // Wq^T [c][h dh + d]
            a.M = B * T; a.N = D; a.K = dh;
            a.C = Kp; a.ldc = D; a.c_sb = (int64_t)T * D;
            a.rdiv = T; a.rmul = R;
            a.alpha = scale;
            fused_build = dh == 64;
            if (fused_build) {   // K', V' (= v_h Wo[:, h]^T) and the query-bias vector in one launch
                XbPair pr{};
                pr.A2 = (const bf16_t*)d.proj + D; pr.B2 = (const bf16_t*)d.wo; pr.C2 = Vp; pr.alpha2 = 1.0f;
                pr.bq = d.bq; pr.mask = d.key_mask; pr.cb = d.colbias; pr.T = T; pr.H = H; pr.scale = scale;
                XCHK(launch_xbuild_pair(a, pr, s));
            } else {
                XCHK(launch_xbuild(a, H, s));
                // V'[b, h*T + j, :] = v_h Wo[:, h]^T
                a.A = (const bf16_t*)d.proj + D;
                a.B = (const bf16_t*)d.wo; a.ldb = D; a.b_sb = dh;               // Wo [n][h dh + d]
                a.C = Vp;
                a.alpha = 1.0f;
                XCHK(launch_xbuild(a, H, s));
            }
        }
        if (!fused_build) {
            const int rows = B * T;
            hipLaunchKernelGGL(xattn_colbias_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, (const bf16_t*)d.proj,
                               (int64_t)2 * D, d.bq, d.key_mask, d.colbias, rows, T, H, dh, scale);
            XCHK(hip_launch_status());
        }
        if (!(d.launch_flags & M3AE_XATTN_LEGACY_CHAIN)) {
            // ONE launch (csrc/xflash.hip): S = x K'^T, softmax + attention dropout in registers, drop(P) resident in LDS,
            // s = dropout(drop(P) V' + bo) + x.  probs / probs_drop are written only when the caller passes them (training).
            if (drop && d.probs && !d.probs_drop) return M3AE_ERR_ARG;
            XfArgs f{};
            f.X = (const bf16_t*)d.x; f.Kp = Kp; f.Vp = Vp; f.colbias = d.colbias; f.bo = d.bo;
            f.S = (bf16_t*)d.s; f.P = (bf16_t*)d.probs; f.Pd = drop ? (bf16_t*)d.probs_drop : nullptr;
            f.B = B; f.I = I; f.D = D; f.H = H;
            f.has_drop = drop; f.drop_a = make_drop(d.dropout_p, d.seed_attn, d.dropout_salt); f.drop_h = make_drop(d.dropout_p, d.seed_hidden, d.dropout_salt);
            f.drop_ld = (int)drop_ld(T);
            // (LayerNorm inside this launch -- single-pass statistics in the pass epilogues, a cross-wave exchange, a normalise
            // pass over the tile's own stores -- was built and measured in round 3: kernel 340 -> 487 us against 89 us for the
            // separate pass below: the re-reads wait for the tile's stores to drain and nothing else runs on the CU meanwhile)
            XCHK(m3ae_xflash_dir1(f, T, s));
        } else {
            if (T != 32 || !d.probs || (drop && !d.probs_drop)) return M3AE_ERR_UNSUPPORTED;
            {   // P = softmax over each head's 32 keys of (x K'^T + c)
                XgArgs a{};
                a.A = (const bf16_t*)d.x; a.lda = D; a.a_sb = (int64_t)I * D;
                a.B = Kp; a.ldb = D; a.b_sb = (int64_t)R * D;
                a.M = I; a.N = R; a.K = D;
                a.C = (bf16_t*)d.probs; a.ldc = R; a.c_sb = (int64_t)I * R;
                a.C2 = drop ? (bf16_t*)d.probs_drop : nullptr;
                a.alpha = 1.0f;
                a.colbias = d.colbias; a.cb_sb = R;
                a.has_drop = drop; a.drop = make_drop(d.dropout_p, d.seed_attn, d.dropout_salt);
                a.H = H; a.Lq = I; a.drop_ld = (int)drop_ld(T);
                XCHK((launch_xg<128, 384, 2, 4, 4, FORM_K, FORM_K, XE_SOFTMAX32>(a, B, s)));
            }
            {   // s = dropout(drop(P) V' + bo) + x
                XgArgs a{};
                a.A = (const bf16_t*)(drop ? d.probs_drop : d.probs); a.lda = R; a.a_sb = (int64_t)I * R;
                a.B = Vp; a.ldb = D; a.b_sb = (int64_t)R * D;
                a.M = I; a.N = D; a.K = R;
                a.C = (bf16_t*)d.s; a.ldc = D;
                a.bias = d.bo;
                a.residual = (const bf16_t*)d.x;
                a.has_drop = drop; a.drop = make_drop(d.dropout_p, d.seed_hidden, d.dropout_salt);
                XCHK((launch_xg<128, 384, 2, 4, 4, FORM_K, FORM_T, XE_DENSE>(a, B, s)));
            }
        }
    }
    // out = LayerNorm(s)                                                          (bert_model.py:363)
    return m3ae_layernorm_fwd(d.s, d.ln_g, d.ln_b, d.out, d.mean, d.rstd, (int64_t)B * Lq, D, d.ln_eps, M3AE_BF16,
                              M3AE_ACT_NONE, 0, stream);
}

namespace {
// split of a reduction of `nchunks` 32-deep chunks so that the launch has ~256 workgroups
void pick_split(XgArgs& a, int nchunks, int tiles) {
    int ks = 256 / (tiles > 0 ? tiles : 1);
    if (ks > nchunks / 8) ks = nchunks / 8;     // >= 8 chunks (256 reduction rows) per workgroup
    if (ks < 1) ks = 1;
    a.kchunks = (nchunks + ks - 1) / ks;
    a.ksplit = (nchunks + a.kchunks - 1) / a.kchunks;
}
}  // namespace

extern "C" int m3ae_xattn_bwd(const m3ae_xattn_desc* dp, void* stream) {
    if (!dp || !m3ae_xattn_bwd_supported(dp)) return M3AE_ERR_UNSUPPORTED;
    const m3ae_xattn_desc& d = *dp;
    hipStream_t s = (hipStream_t)stream;
    const int B = (int)d.B, H = (int)d.H, D = (int)d.D, dh = D / H;
    const int Lq = (int)d.Lq, Lk = (int)d.Lk;
    const float scale = 1.0f / sqrtf((float)dh);
    const bool drop = d.dropout_p > 0.f;
    if (!d.d_out || !d.dx || !d.x || !d.y || !d.proj || !d.prime || !d.probs || !d.s || !d.mean || !d.rstd || !d.ws_ds ||
        !d.ws_dscores || !d.ws_dprime || !d.ws_dproj || !d.ws_vec || !d.ws_ln || (drop && (!d.probs_drop || !d.ws_dsd)) ||
        !d.g_wq || !d.g_wkv || !d.g_wo || !d.g_bq || !d.g_bkv || !d.g_bo)
        return M3AE_ERR_ARG;
    const int64_t Mq = (int64_t)B * Lq;
    // LayerNorm backward: ds (gradient of the pre-LayerNorm sum = residual branch) and dsd (through the hidden dropout)
    const void* dsd = d.ws_ds;
    if (drop) {
        XCHK(m3ae_layernorm_bwd_drop(d.d_out, d.s, d.ln_g, d.ln_b, d.mean, d.rstd, d.ws_ds, d.ws_dsd, d.dropout_p, d.seed_hidden,
                                     d.dropout_salt, d.g_ln_g, d.g_ln_b, d.ws_ln, Mq, D, M3AE_BF16, stream));
        dsd = d.ws_dsd;
    } else {
        XCHK(m3ae_layernorm_bwd(d.d_out, d.s, d.ln_g, d.ln_b, d.mean, d.rstd, d.ws_ds, nullptr, d.g_ln_g, d.g_ln_b, d.ws_ln, Mq, D,
                                M3AE_BF16, M3AE_ACT_NONE, 0, stream));
    }
    const bf16_t* Pd = (const bf16_t*)(drop ? d.probs_drop : d.probs);
    m3ae_gemm_desc g{};
    g.batch1 = g.batch2 = 1;
    g.dtype_a = g.dtype_b = M3AE_BF16;
    g.alpha = 1.0f;
    g.launch_flags = (d.launch_flags & M3AE_XATTN_NO_PERSISTENT) ? M3AE_GEMM_NO_PERSISTENT : 0;

    if (d.dir == 1) {
        const int I = Lq, T = Lk, R = H * T;
        if (!d.colbias) return M3AE_ERR_ARG;
        const bf16_t* Kp = (const bf16_t*)d.prime;
        const bf16_t* Vp = Kp + (int64_t)B * R * D;
        bf16_t* dKp = (bf16_t*)d.ws_dprime;
        bf16_t* dVp = dKp + (int64_t)B * R * D;
        bf16_t* dS = (bf16_t*)d.ws_dscores;
        bf16_t* dkv = (bf16_t*)d.ws_dproj;
        float* dcb = d.ws_vec;
        XCHK(m3ae_colsum(dsd, d.g_bo, Mq, D, D, M3AE_BF16, 1, stream));
        {   // dS = softmax'( (dsd V'^T) ) per 32-key group, attention dropout regenerated
            XgArgs a{};
            a.A = (const bf16_t*)dsd; a.lda = D; a.a_sb = (int64_t)I * D;
            a.B = Vp; a.ldb = D; a.b_sb = (int64_t)R * D;
            a.M = I; a.N = R; a.K = D;
            a.C = dS; a.ldc = R; a.c_sb = (int64_t)I * R;
            a.P = (const bf16_t*)d.probs;
            a.has_drop = drop; a.drop = make_drop(d.dropout_p, d.seed_attn, d.dropout_salt);
            a.drop_mode = 1; a.H = H; a.Lq = I; a.drop_ld = (int)drop_ld(T); a.tkeys = T;
            // a wave's columns must hold whole heads: 96 = 3 x 32 (384-wide tile) or 64 (256-wide tile)
            if (T == 32) XCHK((launch_xg<128, 384, 2, 4, 4, FORM_K, FORM_K, XE_DSOFT>(a, B, s)));
            else XCHK((launch_xg<128, 256, 2, 4, 4, FORM_K, FORM_K, XE_DSOFT>(a, B, s)));
        }
        {   // dV' = drop(P)^T dsd ; dK' = dS^T x      (reductions over the image tokens)
            XgArgs a{};
            a.A = Pd; a.lda = R; a.a_sb = (int64_t)I * R;
            a.B = (const bf16_t*)dsd; a.ldb = D; a.b_sb = (int64_t)I * D;
            a.M = R; a.N = D; a.K = I;
            a.C = dVp; a.ldc = D; a.c_sb = (int64_t)R * D; a.alpha = 1.0f;
            XCHK((launch_xg<128, 384, 2, 4, 4, FORM_T, FORM_T, XE_STORE>(a, B, s)));
            a.A = dS; a.B = (const bf16_t*)d.x; a.C = dKp;
            XCHK((launch_xg<128, 384, 2, 4, 4, FORM_T, FORM_T, XE_STORE>(a, B, s)));
        }
        {   // dx = dS K' + ds
            XgArgs a{};
            a.A = dS; a.lda = R; a.a_sb = (int64_t)I * R;
            a.B = Kp; a.ldb = D; a.b_sb = (int64_t)R * D;
            a.M = I; a.N = D; a.K = R;
            a.C = (bf16_t*)d.dx; a.ldc = D;
            a.residual = (const bf16_t*)d.ws_ds;
            XCHK((launch_xg<128, 384, 2, 4, 4, FORM_K, FORM_T, XE_DENSE>(a, B, s)));
        }
        hipLaunchKernelGGL(xattn_colsum_kernel, dim3((R + 127) / 128, B), dim3(256), 0, s, dS, dcb, I, R, scale);
        XCHK(hip_launch_status());
        {   // per head: dk_h = scale dK'_h Wq_h^T + dcb bq_h ; dv_h = dV'_h Wo[:, h]
            XgArgs a{};
            a.A = dKp; a.lda = D; a.a_sb = (int64_t)T * D; a.a_div = T; a.a_mul = R;
            a.B = (const bf16_t*)d.wq; a.ldb = D; a.b_sb = (int64_t)dh * D;
            a.M = B * T; a.N = dh; a.K = D;
            a.C = dkv; a.ldc = 2 * D; a.c_sb = dh; a.alpha = scale;
            a.bias = d.bq; a.bias_sb = dh;
            a.rowscale = dcb; a.rs_sm = 1; a.rs_sb = T; a.rs_div = T; a.rs_mul = R;
            XCHK((launch_xg<384, 128, 4, 4, 4, FORM_K, FORM_K, XE_STORE, 1>(a, H, s)));
            XgArgs v{};
            v.A = dVp; v.lda = D; v.a_sb = (int64_t)T * D; v.a_div = T; v.a_mul = R;
            v.B = (const bf16_t*)d.wo_t; v.ldb = D; v.b_sb = (int64_t)dh * D;
            v.M = B * T; v.N = dh; v.K = D;
            v.C = dkv + D; v.ldc = 2 * D; v.c_sb = dh; v.alpha = 1.0f;
            XCHK((launch_xg<384, 128, 4, 4, 4, FORM_K, FORM_K, XE_STORE, 1>(v, H, s)));
        }
        {   // dWq[h] += scale k_h^T dK'_h ; dWo[:, h] += dV'_h^T v_h      (reductions over the B*T text rows: split-K atomics)
            XgArgs a{};
            a.A = (const bf16_t*)d.proj; a.lda = 2 * D; a.a_sb = dh;
            a.B = dKp; a.ldb = D; a.b_sb = (int64_t)T * D; a.b_div = T; a.b_mul = R;
            a.M = dh; a.N = D; a.K = B * T;
            a.Cf = d.g_wq; a.ldc = D; a.c_sb = (int64_t)dh * D; a.alpha = scale;
            pick_split(a, (B * T + 31) / 32, H * ((D + 383) / 384));
            XCHK((launch_xg<128, 384, 2, 4, 4, FORM_T, FORM_T, XE_ATOMIC, 1>(a, H, s)));
            XgArgs o{};
            o.A = dVp; o.lda = D; o.a_sb = (int64_t)T * D; o.a_div = T; o.a_mul = R;
            o.B = (const bf16_t*)d.proj + D; o.ldb = 2 * D; o.b_sb = dh;
            o.M = D; o.N = dh; o.K = B * T;
            o.Cf = d.g_wo; o.ldc = D; o.c_sb = dh; o.alpha = 1.0f;
            pick_split(o, (B * T + 31) / 32, H * ((D + 383) / 384));
            XCHK((launch_xg<384, 128, 4, 4, 4, FORM_T, FORM_T, XE_ATOMIC, 1>(o, H, s)));
        }
        hipLaunchKernelGGL(xattn_headvec_kernel, dim3(H, (B * T + 255) / 256), dim3(256), 0, s, (const float*)dcb,
                           (const bf16_t*)d.proj, (int64_t)2 * D, d.g_bq, B * T, T, R, 1, T, dh);
        XCHK(hip_launch_status());
        // dWkv += dkv^T y (+ bias gradient), dy = dkv Wkv                                     (bert_model.py:276-277 backward)
        g.M = 2 * D; g.N = D; g.K = (int64_t)B * T;
        g.A = dkv; g.a_sm = 1; g.a_sk = 2 * D; g.B = d.y; g.b_sk = D; g.b_sn = 1;
        g.C = d.g_wkv; g.c_sm = D; g.c_sn = 1; g.dtype_c = M3AE_F32; g.accumulate = 1; g.a_rowsum = d.g_bkv;
        XCHK(m3ae_gemm(&g, stream));
        if (d.dy) {
            m3ae_gemm_desc y = g;
            y.M = (int64_t)B * T; y.N = D; y.K = 2 * D;
            y.A = dkv; y.a_sm = 2 * D; y.a_sk = 1; y.B = d.wkv_t; y.b_sk = 1; y.b_sn = 2 * D;
            y.C = d.dy; y.c_sm = D; y.dtype_c = M3AE_BF16; y.accumulate = 0; y.a_rowsum = nullptr;
            XCHK(m3ae_gemm(&y, stream));
        }
        return 0;
    }

    // ---------------------------------------------------------------------------------------------------- dir 0
    const int T = Lq, I = Lk, R = T * H;
    if (!d.zctx || !d.ctx || !d.ws_dz || !d.ws_dctx || (drop && !d.rowsum)) return M3AE_ERR_ARG;
    bf16_t* dctx = (bf16_t*)d.ws_dctx;
    bf16_t* dZ = (bf16_t*)d.ws_dz;
    bf16_t* dS = (bf16_t*)d.ws_dscores;
    bf16_t* dQp = (bf16_t*)d.ws_dprime;
    bf16_t* dq = (bf16_t*)d.ws_dproj;
    float* delta = d.ws_vec;
    float* radd = drop ? d.ws_vec + (int64_t)B * R : nullptr;
    // output dense: dWo += dsd^T ctx (+ dbo), dctx = dsd Wo                                         (bert_model.py:361 backward)
    g.M = D; g.N = D; g.K = Mq;
    g.A = dsd; g.a_sm = 1; g.a_sk = D; g.B = d.ctx; g.b_sk = D; g.b_sn = 1;
    g.C = d.g_wo; g.c_sm = D; g.c_sn = 1; g.dtype_c = M3AE_F32; g.accumulate = 1; g.a_rowsum = d.g_bo;
    XCHK(m3ae_gemm(&g, stream));
    {
        m3ae_gemm_desc y = g;
        y.M = Mq; y.N = D; y.K = D;
        y.A = dsd; y.a_sm = D; y.a_sk = 1; y.B = d.wo_t; y.b_sk = 1; y.b_sn = D;
        y.C = dctx; y.c_sm = D; y.dtype_c = M3AE_BF16; y.accumulate = 0; y.a_rowsum = nullptr;
        XCHK(m3ae_gemm(&y, stream));
    }
    {   // dZ[b, t*H + h, :] = dctx_h Wv_h ; dWv[h] += dctx_h^T Z_h ; dbv
        XgArgs a{};
        a.A = dctx; a.lda = D; a.a_sb = dh;
        a.B = (const bf16_t*)d.wkv_t + D; a.ldb = 2 * D; a.b_sb = dh;
        a.M = B * T; a.N = D; a.K = dh;
        a.C = dZ; a.ldc = (int64_t)H * D; a.c_sb = D; a.alpha = 1.0f;
        XCHK(launch_xbuild(a, H, s));
        XgArgs w{};
        w.A = dctx; w.lda = D; w.a_sb = dh;
        w.B = (const bf16_t*)d.zctx; w.ldb = (int64_t)H * D; w.b_sb = D;
        w.M = dh; w.N = D; w.K = B * T;
        w.Cf = d.g_wkv + (int64_t)D * D; w.ldc = D; w.c_sb = (int64_t)dh * D; w.alpha = 1.0f;
        pick_split(w, (B * T + 31) / 32, H * ((D + 383) / 384));
        XCHK((launch_xg<128, 384, 2, 4, 4, FORM_T, FORM_T, XE_ATOMIC>(w, H, s)));
        hipLaunchKernelGGL(xattn_headvec_kernel, dim3(H, (B * T + 255) / 256), dim3(256), 0, s,
                           (const float*)(drop ? d.rowsum : nullptr), (const bf16_t*)dctx, (int64_t)D, d.g_bkv + D, B * T, T, R, H, 1, dh);
        XCHK(hip_launch_status());
    }
    hipLaunchKernelGGL(xattn_rowdot_kernel, dim3((unsigned)(((int64_t)B * R + 3) / 4)), dim3(256), 0, s, (const bf16_t*)dZ,
                       (const bf16_t*)d.zctx, (const bf16_t*)dctx, d.bkv + D, (const float*)d.rowsum, delta, radd, (int64_t)B * R, D, T, H, dh);
    XCHK(hip_launch_status());
    {   // dS = P ((dZ y^T + radd) dropped - delta)
        XgArgs a{};
        a.A = dZ; a.lda = D; a.a_sb = (int64_t)R * D;
        a.B = (const bf16_t*)d.y; a.ldb = D; a.b_sb = (int64_t)I * D; a.b_rows = I;
        a.M = R; a.N = 640; a.K = D;
        a.C = dS; a.ldc = 640; a.c_sb = (int64_t)R * 640;
        a.P = (const bf16_t*)d.probs; a.delta = delta; a.radd = radd;
        a.has_drop = drop; a.drop = make_drop(d.dropout_p, d.seed_attn, d.dropout_salt);
        a.drop_mode = 0; a.H = H; a.Lq = T; a.drop_ld = (int)drop_ld(I);
        XCHK((launch_xg<128, 640, 2, 3, 0, FORM_K, FORM_K, XE_DSOFT>(a, B, s)));
    }
    {   // dQ' = dS y
        XgArgs a{};
        a.A = dS; a.lda = 640; a.a_sb = (int64_t)R * 640;
        a.B = (const bf16_t*)d.y; a.ldb = D; a.b_sb = (int64_t)I * D;
        a.M = R; a.N = D; a.K = I;
        a.C = dQp; a.ldc = D; a.c_sb = (int64_t)R * D; a.alpha = 1.0f;
        XCHK((launch_xg<128, 384, 2, 4, 4, FORM_K, FORM_T, XE_STORE>(a, B, s)));
    }
    if (d.dy) {   // dy = drop(P)^T dZ + dS^T Q'      (reductions over the 384 text-query rows)
        XgArgs a{};
        a.A = Pd; a.lda = 640; a.a_sb = (int64_t)R * 640;
        a.B = dZ; a.ldb = D; a.b_sb = (int64_t)R * D;
        a.M = I; a.N = D; a.K = 2 * R;          // one pass over both reductions: chunks 0 .. R/32 - 1 from (drop(P), dZ), then (dS, Q')
        a.A2 = dS; a.B2 = (const bf16_t*)d.prime; a.k_switch = R / 32;
        a.C = (bf16_t*)d.dy; a.ldc = D; a.c_sb = (int64_t)I * D; a.alpha = 1.0f;
        XCHK((launch_xg<128, 384, 2, 4, 4, FORM_T, FORM_T, XE_STORE, 1>(a, B, s)));
    }
    {   // per head: dq_h = scale dQ'_h Wk_h^T ; dWk[h] += scale q_h^T dQ'_h
        XgArgs a{};
        a.A = dQp; a.lda = (int64_t)H * D; a.a_sb = D;
        a.B = (const bf16_t*)d.wkv; a.ldb = D; a.b_sb = (int64_t)dh * D;
        a.M = B * T; a.N = dh; a.K = D;
        a.C = dq; a.ldc = D; a.c_sb = dh; a.alpha = scale;
        XCHK((launch_xg<384, 128, 4, 4, 4, FORM_K, FORM_K, XE_STORE>(a, H, s)));
        XgArgs w{};
        w.A = (const bf16_t*)d.proj; w.lda = D; w.a_sb = dh;
        w.B = dQp; w.ldb = (int64_t)H * D; w.b_sb = D;
        w.M = dh; w.N = D; w.K = B * T;
        w.Cf = d.g_wkv; w.ldc = D; w.c_sb = (int64_t)dh * D; w.alpha = scale;
        pick_split(w, (B * T + 31) / 32, H * ((D + 383) / 384));
        XCHK((launch_xg<128, 384, 2, 4, 4, FORM_T, FORM_T, XE_ATOMIC>(w, H, s)));
    }
    // query projection: dWq += dq^T x (+ dbq), dx = dq Wq + ds                                   (bert_model.py:263 backward)
    g.M = D; g.N = D; g.K = Mq;
    g.A = dq; g.a_sm = 1; g.a_sk = D; g.B = d.x; g.b_sk = D; g.b_sn = 1;
    g.C = d.g_wq; g.c_sm = D; g.c_sn = 1; g.dtype_c = M3AE_F32; g.accumulate = 1; g.a_rowsum = d.g_bq;
    XCHK(m3ae_gemm(&g, stream));
    {
        m3ae_gemm_desc y = g;
        y.M = Mq; y.N = D; y.K = D;
        y.A = dq; y.a_sm = D; y.a_sk = 1; y.B = d.wq_t; y.b_sk = 1; y.b_sn = D;
        y.C = d.dx; y.c_sm = D; y.dtype_c = M3AE_BF16; y.accumulate = 0; y.a_rowsum = nullptr; y.residual = d.ws_ds;
        XCHK(m3ae_gemm(&y, stream));
    }
    return 0;
}
