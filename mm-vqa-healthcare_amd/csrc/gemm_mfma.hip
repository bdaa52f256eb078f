// bf16 MFMA GEMM kernels for gfx950 (CDNA4): the 93 % of the hot path's FLOPs.
//
//   NT : C[M,N] = epi(alpha * A[M,K] . B[N,K]^T)  -- forward Linear (x . W^T) and dgrad (dY . (W^T)^T with the
//        transposed bf16 weight shadow), both operands K-contiguous.
//   TN : C[N1,N2] += alpha * A[Mr,N1]^T . B[Mr,N2] -- wgrad (dY^T . X): the reduction runs over the ROWS of both
//        operands, so MFMA fragments are fetched with the hardware transposing LDS read ds_read_b64_tr_b16;
//        split over the reduction with fp32 atomics into the (fp32) gradient buffer.
//
// Structure (both): 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave = 4x4 MFMA
// 16x16x32 tiles, 64 fp32 accumulators/lane), 64-deep reduction step, operands staged global->LDS with
// global_load_lds_dwordx4 (no VGPR round trip) into a 2-stage ring (64 KiB -> 2 workgroups/CU), one barrier per
// step with the next stage's DMA in flight under the MFMAs.  LDS images are XOR-swizzled on the SOURCE address
// (the LDS-DMA destination is lane-linear) with the same involution on the read, so ds_read_b128 / tr reads are
// bank-conflict free.  Workgroup ids are remapped so that each XCD (own L2) walks a contiguous range of tiles.
#include "gemm_nt_common.h"
#include <stdlib.h>

namespace {

constexpr int BM = 128, BN = 128, BK = 64;         // TN kernel tile
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;    // 32 KiB
constexpr int LDS_BYTES = 2 * STAGE_BYTES;         // 64 KiB

using namespace m3g;

// BM_ x BN_ output tile, one WM x 64 sub-tile per wave ((BM_/WM) x (BN_/64) waves), BKT-deep reduction steps, NST-stage
// LDS ring with the DMA of stage t + NST - 1 issued before the MFMAs of stage t and retired by a COUNTED s_waitcnt
// (loads stay in flight across the raw s_barrier).
template <int BM_, int BN_, int BKT, int NST, int WM, int EPI>
__global__ __launch_bounds__((BM_ / WM) * (BN_ / 64) * 64, 2) void gemm_nt_bf16_kernel(MfmaArgs a) {
    if (a.has_drop) drop_resolve(a.drop);
    constexpr int WAVES_N = BN_ / 64, NWAVES = (BM_ / WM) * WAVES_N, MI = WM / 16;
    constexpr int A_BYTES = BM_ * BKT * 2, ST_BYTES = (BM_ + BN_) * BKT * 2;
    constexpr int A_SEGS = A_BYTES / 1024 / NWAVES, B_SEGS = BN_ * BKT * 2 / 1024 / NWAVES;  // DMA pieces per wave
    constexpr int G = A_SEGS + B_SEGS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WAVES_N, wc = wave % WAVES_N;

    const unsigned tiles_n = (unsigned)((a.N + BN_ - 1) / BN_);
    unsigned tm, tn;
    nt_tile_coords(xcd_remap(blockIdx.x, gridDim.x), gridDim.x / tiles_n, tiles_n, tm, tn);
    const int64_t m0 = (int64_t)tm * BM_;
    const int64_t n0 = (int64_t)tn * BN_;

    f32x4 acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nt = (int)(a.K / BKT);
#pragma unroll
    for (int s = 0; s < NST - 1; ++s) {
        if (s < nt) {
            nt_stage<BKT, A_SEGS, NWAVES>(a.A, a.lda, m0, a.M, (int64_t)s * BKT, smem + s * ST_BYTES, wave, lane);
            nt_stage<BKT, B_SEGS, NWAVES>(a.B, a.ldb, n0, a.N, (int64_t)s * BKT, smem + s * ST_BYTES + A_BYTES, wave, lane);
        }
    }

    const int frow = lane & 15, fchunk = lane >> 4;
    int cur_s = 0, nxt_s = NST - 1;
    for (int t = 0; t < nt; ++t) {
        // stage t has landed once at most the NST - 2 younger stages of this wave are still in flight
        if (t + NST - 2 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * G) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // every wave's pieces of stage t landed; everyone left stage t - 1's buffer
        asm volatile("" ::: "memory");
        if (t + NST - 1 < nt) {
            char* nxt = smem + nxt_s * ST_BYTES;
            nt_stage<BKT, A_SEGS, NWAVES>(a.A, a.lda, m0, a.M, (int64_t)(t + NST - 1) * BKT, nxt, wave, lane);
            nt_stage<BKT, B_SEGS, NWAVES>(a.B, a.ldb, n0, a.N, (int64_t)(t + NST - 1) * BKT, nxt + A_BYTES, wave, lane);
        }
        const char* At = smem + cur_s * ST_BYTES;
        const char* Bt = At + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < BKT / 32; ++kk) {
            s16x8 af[MI], bfr[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = nt_frag<BKT>(Bt, wc * 64 + j * 16 + frow, kk * 4 + fchunk);
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = nt_frag<BKT>(At, wr * WM + i * 16 + frow, kk * 4 + fchunk);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    // D'[n][m] = sum_k B[n][k] * A[m][k]: each lane ends with 4 consecutive n of one m
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8_t, bfr[j]), __builtin_bit_cast(bf16x8_t, af[i]), acc[i][j], 0, 0, 0);
        }
        asm volatile("" ::: "memory");
        cur_s = cur_s + 1 == NST ? 0 : cur_s + 1;
        nxt_s = nxt_s + 1 == NST ? 0 : nxt_s + 1;
    }

    if (a.rows_epi) {
        __builtin_amdgcn_s_barrier();  // every wave has consumed its last stage: the ring is free for the C slabs
        asm volatile("" ::: "memory");
        if (a.c_f32) epilogue_rows<float, EPI, MI>(a, smem, wave, lane, m0 + wr * WM, n0 + wc * 64, acc);
        else epilogue_rows<bf16_t, EPI, MI>(a, smem, wave, lane, m0 + wr * WM, n0 + wc * 64, acc);
        return;
    }
    // direct epilogue (N % 8 != 0): lane holds C[m = .. + (lane & 15)][n = .. + 4 * (lane >> 4) + 0..3]
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int64_t m = m0 + wr * WM + i * 16 + (lane & 15);
        if (m >= a.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t n = n0 + wc * 64 + j * 16 + 4 * (lane >> 4);
            if (n >= a.N) continue;  // N % 4 == 0 is a precondition, so n < N implies n + 3 < N
            if (a.c_f32) epilogue4<float, EPI>(a, m, n, acc[i][j]);
            else epilogue4<bf16_t, EPI>(a, m, n, acc[i][j]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// TN kernel (wgrad)
// ---------------------------------------------------------------------------------------------------------
// C[n1][n2] (+)= alpha * sum_r A[r][n1] * B[r][n2];  a.M = N1, a.N = N2, a.K = reduction rows.
// BM_ x BN_ output tile, one WM x 64 sub-tile per wave, 64 reduction rows per step, 2-stage ring.
template <int BM_, int BN_, int WM, int BKR, int NST>
__global__ __launch_bounds__((BM_ / WM) * (BN_ / 64) * 64, 2) void gemm_tn_bf16_kernel(MfmaArgs a) {
    constexpr int WAVES_N = BN_ / 64, NWAVES = (BM_ / WM) * WAVES_N, MI = WM / 16;
    constexpr int A_BYTES = BM_ * BKR * 2, ST_BYTES = (BM_ + BN_) * BKR * 2;
    constexpr int A_SEGS = A_BYTES / 1024 / NWAVES, B_SEGS = BN_ * BKR * 2 / 1024 / NWAVES;
    constexpr int G = A_SEGS + B_SEGS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WAVES_N, wc = wave % WAVES_N;

    const unsigned tiles_n = (unsigned)(a.N / BN_);
    const unsigned tiles = (unsigned)(a.M / BM_) * tiles_n;
    const unsigned wg = xcd_remap(blockIdx.x, gridDim.x);
    const unsigned tile_id = wg % tiles, split = wg / tiles;
    const int64_t m0 = (int64_t)(tile_id / tiles_n) * BM_;
    const int64_t n0 = (int64_t)(tile_id % tiles_n) * BN_;
    const int64_t r_begin = (int64_t)split * a.k_chunk;
    int64_t r_end = r_begin + a.k_chunk;
    if (r_end > a.K) r_end = a.K;
    if (r_begin >= r_end) return;  // whole workgroup: uniform
    const int nt = (int)((r_end - r_begin + BKR - 1) / BKR);

    f32x4 acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // bias gradient rides along: the waves of the first output-column tile also multiply their A^T fragments by a
    // ones fragment (every column of that product is the row sum of A^T = the column sum of dY)
    const bool do_rowsum = a.a_rowsum != nullptr && n0 == 0 && wc == 0;
    f32x4 rsum[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) rsum[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const s16x8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};

#pragma unroll
    for (int s = 0; s < NST - 1; ++s) {
        if (s < nt) {
            const int64_t r0 = r_begin + (int64_t)s * BKR;
            tn_stage<BM_, A_SEGS, NWAVES>(a.A, a.lda, r0, r_end, m0, smem + s * ST_BYTES, wave, lane);
            tn_stage<BN_, B_SEGS, NWAVES>(a.B, a.ldb, r0, r_end, n0, smem + s * ST_BYTES + A_BYTES, wave, lane);
        }
    }
    int cur_s = 0, nxt_s = NST - 1;
    for (int t = 0; t < nt; ++t) {
        if (t + NST - 2 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * G) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + NST - 1 < nt) {
            char* nxt = smem + nxt_s * ST_BYTES;
            const int64_t r0 = r_begin + (int64_t)(t + NST - 1) * BKR;
            tn_stage<BM_, A_SEGS, NWAVES>(a.A, a.lda, r0, r_end, m0, nxt, wave, lane);
            tn_stage<BN_, B_SEGS, NWAVES>(a.B, a.ldb, r0, r_end, n0, nxt + A_BYTES, wave, lane);
        }
        const char* At = smem + cur_s * ST_BYTES;
        const char* Bt = At + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < BKR / 32; ++kk) {
            s16x8 af[MI], bfr[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = tn_frag<BN_>(Bt, kk * 32, wc * 64 + j * 16, lane);
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = tn_frag<BM_>(At, kk * 32, wr * WM + i * 16, lane);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    // D[n1][n2] = sum_r A^T[n1][r] * B[r][n2]
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8_t, af[i]), __builtin_bit_cast(bf16x8_t, bfr[j]), acc[i][j], 0, 0, 0);
            if (do_rowsum) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    rsum[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8_t, af[i]), __builtin_bit_cast(bf16x8_t, ones), rsum[i], 0, 0, 0);
            }
        }
        asm volatile("" ::: "memory");
        cur_s = cur_s + 1 == NST ? 0 : cur_s + 1;
        nxt_s = nxt_s + 1 == NST ? 0 : nxt_s + 1;
    }
    if (do_rowsum && (lane & 15) == 0) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) atomicAdd(a.a_rowsum + m0 + wr * WM + i * 16 + 4 * (lane >> 4) + r, rsum[i][r]);
    }

    // lane holds D[n1 = .. + 4 * (lane >> 4) + reg][n2 = .. + (lane & 15)]
    float* C = (float*)a.C;
    const bool atomic = a.splits > 1;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t n1 = m0 + wr * WM + i * 16 + 4 * (lane >> 4) + r;
                const int64_t n2 = n0 + wc * 64 + j * 16 + (lane & 15);
                float* p = C + n1 * a.ldc + n2;
                const float x = acc[i][j][r] * a.alpha;
                if (atomic) atomicAdd(p, x);
                else if (a.accumulate) *p += x;
                else *p = x;
            }
}

}  // namespace

static thread_local const char* g_last_path = "none";
const char* m3ae_last_gemm_path(void) { return g_last_path; }

int m3ae_gemm_generic(const m3ae_gemm_desc& d, hipStream_t s);

static bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// Kernel selection is by shape (the "auto" rules below).  A caller may pin a variant PER CALL through the selector fields of
// m3ae_gemm_desc.launch_flags (M3AE_GEMM_NT_VARIANT / _TN_VARIANT / _COL_GROUP: tests compare the variants bit for bit, tools
// time them against each other); the library keeps no tuning state and reads no environment variable.
// The persistent form of the ping-pong kernel (static tile lists, one workgroup per CU) is never taken under
// M3AE_GEMM_NO_PERSISTENT (data-parallel runs): when RCCL's kernels hold some CUs the persistent workgroups that found no CU only
// start after others have walked their whole tile list (the kernel's time doubles); the one-tile-per-workgroup launch just runs
// on the CUs that are free.

template <int BM_, int BN_, int BKT, int NST, int WM, int EPI>
static int launch_nt_t(const MfmaArgs& a, hipStream_t s) {
    constexpr int nwaves = (BM_ / WM) * (BN_ / 64);
    constexpr int ring = NST * (BM_ + BN_) * BKT * 2, slabs = nwaves * 32 * 68 * 4;
    constexpr int lds = ring > slabs ? ring : slabs;
    constexpr int threads = nwaves * 64;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)gemm_nt_bf16_kernel<BM_, BN_, BKT, NST, WM, EPI>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    const int64_t tiles = cdiv(a.M, BM_) * cdiv(a.N, BN_);
    hipLaunchKernelGGL((gemm_nt_bf16_kernel<BM_, BN_, BKT, NST, WM, EPI>), dim3((unsigned)tiles), dim3(threads), lds, s,
                       a);
    return hip_launch_status();
}


// ---------------------------------------------------------------------------------------------------------
// NT "ping-pong" kernel: 256 x 256 tile, 8 waves (2 x 4, 128 x 64 each), 32-deep reduction chunks in a 4-slot LDS
// ring (128 KiB).  Each chunk is consumed in TWO phases (rows 0-63 / 64-127 of the wave's sub-tile: 16 MFMAs each);
// a phase is  [LDS fragment reads + 2 LDS-DMA issues] s_barrier [16 MFMAs] s_barrier.  The two wave rows (wr = 0 / 1:
// one wave of each per SIMD) run staggered by one barrier, so on every SIMD one wave is in its MFMA cluster while the
// other fetches fragments and issues DMA -- the MFMA pipe never waits for LDS.
//
// Ring schedule (phase p = 2c / 2c+1 works on chunk c in slot c & 3; loads are issued in the order B(0) A(0) B(1) A(1) ...):
//   phase 2c   issues B(c+3) into slot (c-1)&3 -- B(c-1) was last read in phase 2c-2 (two phases earlier: every wave's
//              reads were retired by an lgkmcnt(0) at least two barriers ago, stagger included);
//   phase 2c+1 issues A(c+3)                   -- A(c-1) was last read in phase 2c-1;
//   phase 2c+1 waits (counted vmcnt: the 8 loads of chunks c+2, c+3 stay in flight) for chunk c+1 BEFORE its first
//              barrier; chunk c+1 is first read in phase 2c+2, i.e. after a barrier every wave passed post-wait.
// ---------------------------------------------------------------------------------------------------------
// M3AE_EXP_NT_*: timing-only experiments on the ping-pong kernels (tools/nt_exp.sh; wrong results, never in the product build):
// operands left unstaged / staged from contiguous 1-KiB source pieces (as if stored reduction-chunk-major) / stores dropped
#if defined(M3AE_EXP_NT_NODMA)
#define PP_STAGE(G, ld, r0, nr, k0, tile) do { } while (0)
#elif defined(M3AE_EXP_NT_CONTIG)
template <int BKT, int SEGS_PER_WAVE, int NWAVES>
DEVINL void nt_stage_contig(const bf16_t* G, int64_t ld, int64_t row0, int64_t nrows, int64_t k0, char* tile, int wave, int lane) {
    constexpr int CPR = BKT / 8, RPS = 64 / CPR;
#pragma unroll
    for (int q = 0; q < SEGS_PER_WAVE; ++q) {
        const int seg = q * NWAVES + wave;
        const int row = seg * RPS + lane / CPR;
        int64_t grow = row0 + row;
        grow = grow < nrows ? grow : nrows - 1;
        glds16(G + (k0 / BKT) * nrows * BKT + grow * BKT + (lane % CPR) * 8, tile + seg * 1024);
    }
}
#ifdef M3AE_EXP_NT_L2HOT
#define PP_STAGE(G, ld, r0, nr, k0, tile) nt_stage_contig<CK, 2, NW>(G, ld, 0, nr, k0, tile, wave, lane)
#else
#define PP_STAGE(G, ld, r0, nr, k0, tile) nt_stage_contig<CK, 2, NW>(G, ld, r0, nr, k0, tile, wave, lane)
#endif
#elif defined(M3AE_EXP_NT_L2HOT)   // every tile stages rows 0..255 of both operands: the sources stay L2-resident (true hits)
#define PP_STAGE(G, ld, r0, nr, k0, tile) nt_stage<CK, 2, NW>(G, ld, 0, nr, k0, tile, wave, lane)
#else
#define PP_STAGE(G, ld, r0, nr, k0, tile) nt_stage<CK, 2, NW>(G, ld, r0, nr, k0, tile, wave, lane)
#endif

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt_pp_kernel(MfmaArgs a) {
    if (a.has_drop) drop_resolve(a.drop);
    constexpr int CK = 32, NW = 8, A_BYTES = 256 * CK * 2, SLOT = 2 * A_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    const unsigned tiles_n = (unsigned)((a.N + 255) / 256);
    unsigned tm, tn;
    nt_tile_coords(xcd_remap(blockIdx.x, gridDim.x), gridDim.x / tiles_n, tiles_n, tm, tn, (unsigned)a.col_group);
    const int64_t m0 = (int64_t)tm * 256;
    const int64_t n0 = (int64_t)tn * 256;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nc = (int)(a.K / CK);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (c < nc) {
            PP_STAGE(a.B, a.ldb, n0, a.N, (int64_t)c * CK, smem + c * SLOT + A_BYTES);
            PP_STAGE(a.A, a.lda, m0, a.M, (int64_t)c * CK, smem + c * SLOT);
        }
    }
    if (nc >= 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (nc == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PP_FENCE();
    __builtin_amdgcn_s_barrier();  // chunk 0 is in LDS for every wave
    PP_FENCE();
    if (wr == 1) { __builtin_amdgcn_s_barrier(); PP_FENCE(); }  // stagger the second wave row by one barrier

    const int frow = lane & 15, fchunk = lane >> 4;
    for (int c = 0; c < nc; ++c) {
        const char* At = smem + (c & 3) * SLOT;
        const char* Bt = At + A_BYTES;
        char* nxt = smem + ((c + 3) & 3) * SLOT;
        const bool more = c + 3 < nc;
        s16x8 bfr[4], af[4];
        // ---------------- phase 2c: rows 0..63 of the wave's sub-tile
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = nt_frag<CK>(Bt, wc * 64 + j * 16 + frow, fchunk);
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = nt_frag<CK>(At, wr * 128 + i * 16 + frow, fchunk);
        if (more) PP_STAGE(a.B, a.ldb, n0, a.N, (int64_t)(c + 3) * CK, nxt + A_BYTES);
        PP_FENCE();
        __builtin_amdgcn_s_barrier();
        PP_FENCE();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8_t, bfr[j]), __builtin_bit_cast(bf16x8_t, af[i]), acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        PP_FENCE();
        __builtin_amdgcn_s_barrier();
        PP_FENCE();
        // ---------------- phase 2c + 1: rows 64..127 (the B fragments stay in registers)
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = nt_frag<CK>(At, wr * 128 + 64 + i * 16 + frow, fchunk);
        if (more) PP_STAGE(a.A, a.lda, m0, a.M, (int64_t)(c + 3) * CK, nxt);
        {
            const int rem = nc - 1 - c;  // chunks after this one; chunk c + 1 must have landed before the next phase
            if (rem >= 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (rem == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        PP_FENCE();
        __builtin_amdgcn_s_barrier();
        PP_FENCE();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8_t, bfr[j]), __builtin_bit_cast(bf16x8_t, af[i]), acc[4 + i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        PP_FENCE();
        __builtin_amdgcn_s_barrier();
        PP_FENCE();
    }
    if (wr == 0) { __builtin_amdgcn_s_barrier(); PP_FENCE(); }  // re-align: every wave has executed 4 nc + 2 barriers
    // all fragment reads are retired and no DMA is outstanding (vmcnt(0) in the last odd phase): the ring is free

    if (a.rows_epi) {
        if (a.c_f32) epilogue_rows<float, EPI, 8>(a, smem, wave, lane, m0 + wr * 128, n0 + wc * 64, acc);
        else epilogue_rows<bf16_t, EPI, 8>(a, smem, wave, lane, m0 + wr * 128, n0 + wc * 64, acc);
        return;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int64_t m = m0 + wr * 128 + i * 16 + (lane & 15);
        if (m >= a.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t n = n0 + wc * 64 + j * 16 + 4 * (lane >> 4);
            if (n >= a.N) continue;
            if (a.c_f32) epilogue4<float, EPI>(a, m, n, acc[i][j]);
            else epilogue4<bf16_t, EPI>(a, m, n, acc[i][j]);
        }
    }
}

template <int EPI>
static int launch_nt_pp(const MfmaArgs& a, hipStream_t s) {
    constexpr int lds = 4 * (256 + 256) * 32 * 2;  // 128 KiB ring; the epilogue slabs (8 x 8704 B) reuse it
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_pp_kernel<EPI>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    const int64_t tiles = cdiv(a.M, 256) * cdiv(a.N, 256);
    g_last_path = "mfma_nt_pp";
    hipLaunchKernelGGL((gemm_nt_pp_kernel<EPI>), dim3((unsigned)tiles), dim3(512), lds, s, a);
    return hip_launch_status();
}


// ---------------------------------------------------------------------------------------------------------
// Persistent form of gemm_nt_pp_kernel: one workgroup per CU walks tiles v = block, block + grid, ... (grid a multiple of
// 8, so every tile of a workgroup maps to the same XCD range as in the one-tile-per-workgroup launch).  A K = 768 tile
// spends ~12 % of its time waiting for its first chunks and ~16-30 % in the epilogue: here the NEXT tile's chunks 0 and 1
// are requested before the epilogue of the current one (ring slots 2, 3; the epilogue's 16-row slabs live in slots
// 0, 1), chunk 2 right after it, so the main loop of the next tile starts on landed data.
// Ring slot of chunk c is (c + 2) & 3; everything else is the schedule of gemm_nt_pp_kernel.
// ---------------------------------------------------------------------------------------------------------
#ifdef M3AE_NT_TRACE   // diagnostic build only (tools/nt_trace.py): shader clocks per section of the main loop, summed per workgroup
__device__ uint64_t g_nt_trace[1024 * 2 * 8];   // [block][wave row][fragment reads, DMA issue, barrier 1, lgkmcnt wait, MFMA issue, barrier 2, chunks, -]
#define NT_CLK() ({ __builtin_amdgcn_sched_barrier(0); uint64_t t_ = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); t_; })
#define NT_ACC(i) do { const uint64_t n_ = NT_CLK(); nt_acc[i] += n_ - nt_t; nt_t = n_; } while (0)
#else
#define NT_ACC(i) do { } while (0)
#endif

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt_pp_persistent_kernel(MfmaArgs a) {
    if (a.has_drop) drop_resolve(a.drop);
    constexpr int CK = 32, NW = 8, A_BYTES = 256 * CK * 2, SLOT = 2 * A_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const unsigned tiles_n = (unsigned)((a.N + 255) / 256);
    const unsigned tiles_m = (unsigned)((a.M + 255) / 256);
    const unsigned total = tiles_m * tiles_n;
    const int nc = (int)(a.K / CK);   // >= 3 (host check)
    const int frow = lane & 15, fchunk = lane >> 4;

    unsigned v = blockIdx.x;
    unsigned tm, tn;
    nt_tile_coords(xcd_remap(v, total), tiles_m, tiles_n, tm, tn, (unsigned)a.col_group);
    int64_t m0 = (int64_t)tm * 256, n0 = (int64_t)tn * 256;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        char* sl = smem + ((c + 2) & 3) * SLOT;
        PP_STAGE(a.B, a.ldb, n0, a.N, (int64_t)c * CK, sl + A_BYTES);
        PP_STAGE(a.A, a.lda, m0, a.M, (int64_t)c * CK, sl);
    }
    int top_wait = 0;
#ifdef M3AE_NT_TRACE
    uint64_t nt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, nt_t = 0;
#endif
    // stores per wave of an interior tile's epilogue: 16 row groups x (C [+ pre-activation / derivative]); bf16 only
    const int interior_wait = a.c_f32 ? 1 : (a.preact ? 3 : 2);
    for (;;) {
        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // chunks 0 (and 1) of this tile have landed.  vmcnt retires in order, so the wait names how many YOUNGER operations
        // may stay in flight: chunks 1, 2 on the first tile; afterwards chunk 2 plus -- when the previous tile was an
        // interior one, whose epilogue issued a known number of stores -- those stores (no store drain before the main loop)
        if (top_wait == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (top_wait == 2) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
        else if (top_wait == 3) asm volatile("s_waitcnt vmcnt(36)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        PP_FENCE();
        __builtin_amdgcn_s_barrier();
        PP_FENCE();
        if (wr == 1) { __builtin_amdgcn_s_barrier(); PP_FENCE(); }
#ifdef M3AE_NT_TRACE
        nt_t = NT_CLK();
#endif
        for (int c = 0; c < nc; ++c) {
            const char* At = smem + ((c + 2) & 3) * SLOT;
            const char* Bt = At + A_BYTES;
            char* nxt = smem + ((c + 5) & 3) * SLOT;
            const bool more = c + 3 < nc;
            s16x8 bfr[4], af[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = nt_frag<CK>(Bt, wc * 64 + j * 16 + frow, fchunk);
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = nt_frag<CK>(At, wr * 128 + i * 16 + frow, fchunk);
            NT_ACC(0);
            if (more) PP_STAGE(a.B, a.ldb, n0, a.N, (int64_t)(c + 3) * CK, nxt + A_BYTES);
            NT_ACC(1);
            PP_FENCE();
            __builtin_amdgcn_s_barrier();
            PP_FENCE();
            NT_ACC(2);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            NT_ACC(3);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8_t, bfr[j]), __builtin_bit_cast(bf16x8_t, af[i]), acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            NT_ACC(4);
            PP_FENCE();
            __builtin_amdgcn_s_barrier();
            PP_FENCE();
            NT_ACC(5);
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = nt_frag<CK>(At, wr * 128 + 64 + i * 16 + frow, fchunk);
            NT_ACC(0);
            if (more) PP_STAGE(a.A, a.lda, m0, a.M, (int64_t)(c + 3) * CK, nxt);
            NT_ACC(1);
            {
                const int rem = nc - 1 - c;
                if (rem >= 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if (rem == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            NT_ACC(6);
            PP_FENCE();
            __builtin_amdgcn_s_barrier();
            PP_FENCE();
            NT_ACC(2);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            NT_ACC(3);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8_t, bfr[j]), __builtin_bit_cast(bf16x8_t, af[i]), acc[4 + i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            NT_ACC(4);
            PP_FENCE();
            __builtin_amdgcn_s_barrier();
            PP_FENCE();
            NT_ACC(5);
        }
#ifdef M3AE_NT_TRACE
        nt_acc[7] += (uint64_t)nc;
#endif
        if (wr == 0) { __builtin_amdgcn_s_barrier(); PP_FENCE(); }
        // every fragment read of this tile is retired, no DMA outstanding: request the next tile's chunks 0, 1 (slots 2, 3)
        const unsigned vn = v + gridDim.x;
        const bool again = vn < total;
        const int64_t m_cur = m0, n_cur = n0;
        if (again) {
            nt_tile_coords(xcd_remap(vn, total), tiles_m, tiles_n, tm, tn, (unsigned)a.col_group);
            m0 = (int64_t)tm * 256; n0 = (int64_t)tn * 256;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                char* sl = smem + ((c + 2) & 3) * SLOT;
                PP_STAGE(a.B, a.ldb, n0, a.N, (int64_t)c * CK, sl + A_BYTES);
                PP_STAGE(a.A, a.lda, m0, a.M, (int64_t)c * CK, sl);
            }
        }
        if (a.c_f32) epilogue_rows<float, EPI, 8, 1>(a, smem, wave, lane, m_cur + wr * 128, n_cur + wc * 64, acc);
        else epilogue_rows<bf16_t, EPI, 8, 1>(a, smem, wave, lane, m_cur + wr * 128, n_cur + wc * 64, acc);
#ifdef M3AE_NT_TRACE
        if (!again && lane == 0 && (wave & 3) == 0 && blockIdx.x < 1024)
            for (int q_ = 0; q_ < 8; ++q_) g_nt_trace[((size_t)blockIdx.x * 2 + wr) * 8 + q_] = nt_acc[q_];
#endif
        if (!again) break;
        top_wait = (m_cur + 256 <= a.M && n_cur + 256 <= a.N) ? interior_wait : 1;
        PP_FENCE();
        __builtin_amdgcn_s_barrier();   // every wave is done with its slab (slots 0, 1): chunk 2 may land in slot 0
        PP_FENCE();
        {
            char* sl = smem + ((2 + 2) & 3) * SLOT;
            PP_STAGE(a.B, a.ldb, n0, a.N, (int64_t)2 * CK, sl + A_BYTES);
            PP_STAGE(a.A, a.lda, m0, a.M, (int64_t)2 * CK, sl);
        }
        v = vn;
    }
}

template <int EPI>
static int launch_nt_pp_persistent(const MfmaArgs& a, hipStream_t s) {
    constexpr int lds = 4 * (256 + 256) * 32 * 2;
    static bool attr_set = false;
    static int cus = 256;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_pp_persistent_kernel<EPI>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            cus = prop.multiProcessorCount / 8 * 8;
        attr_set = true;
    }
    g_last_path = "mfma_nt_pp";
    hipLaunchKernelGGL((gemm_nt_pp_persistent_kernel<EPI>), dim3((unsigned)cus), dim3(512), lds, s, a);
    return hip_launch_status();
}

// variants (launch_flags: M3AE_GEMM_NT_VARIANT(v)):
//   0: 128x128 tile, BK 64, 2 stages, 4 waves x (64x64)   -- 64 KiB LDS, 2 workgroups / CU (small / few-tile shapes)
//   4: 256x256 tile, BK 64, 2 stages, 8 waves x (128x64)  -- 128 KiB LDS, 1 workgroup / CU (the ping-pong kernel's
//      bit-exact reference in tools/gemm_race.py)
//   7: 256x256 tile, 32-deep chunks in a 4-slot ring, 8 waves in two staggered rows (ping-pong, gemm_nt_pp_kernel)
//   8: the persistent form of 7 (static tile lists, one workgroup per CU; >= 512 tiles)
// Tilings tried and dropped (measured slower on every shape of the path, r01 logs): a 128x256 "dual" kernel (4 waves, two
// workgroups per CU so that one's main loop runs under the other's epilogue; bit-identical, 3-9 % slower, removed in r02);
// 256x128 with BK 64 / 3 stages,
// 256x128 with BK 32 / 2 and 3 stages, 256x256 with BK 32 / 4 stages (single barrier), 128x128 with BK 32;
// 256x256 with FOUR waves of 128x128 (256 accumulators pinned to AGPRs through asm constraints, one wave per SIMD, fragments
// software-pipelined in registers, 33 % fewer LDS fragment bytes per MFMA): bit-identical, 1143 vs 1281 TF/s at 8192^3
// and 12-25 % slower on the path's shapes (profiles/r01_nt_w4_probe.log) -- one wave per SIMD does not keep the MFMA
// pipe as busy as the two staggered wave rows do.
template <int EPI>
static int launch_nt_v(const MfmaArgs& a, hipStream_t s) {
    const int g_nt_variant = a.nt_variant;
    if (g_nt_variant < 0) {  // auto (default): measured on MI355X, profiles/r01_gemm_shapes.log
        // 256 x 256 ping-pong kernel (one workgroup per CU, 256 slots) or 128 x 128 kernel (two per CU, 512 slots)?  What
        // decides is how full the last round of tiles is: efficiency = tiles / (rounds * slots).  The ping-pong kernel is
        // ~12 % faster per FLOP on full rounds, so it is taken when eff256 >= 0.88 eff128.  Fits every measured pair
        // (MI355X, K = 768 / 3072): 8192 x 3072 (0.75 vs 1.00: 66 vs 47 us -> 128), 8192 x 768 (0.38 vs 0.75 -> 128),
        // 36928 x 768 (0.85 vs 0.85: 54 vs 57 us, 164 vs 181 us -> 256), 3072 x 3072 (0.56 vs 0.56: 25.7 vs 26.0 us),
        // 2048 x 3072 (0.38 vs 0.75: 23.5 vs 17.1 us -> 128); profiles/r01_gemm_shapes.log, r01_nt_tile_rule.log.
        const int64_t t256 = cdiv(a.M, 256) * cdiv(a.N, 256), t128 = cdiv(a.M, 128) * cdiv(a.N, 128);
        const double eff256 = (double)t256 / (double)(cdiv(t256, 256) * 256);
        const double eff128 = (double)t128 / (double)(cdiv(t128, 512) * 512);
        // (round 4, pp2 against the 128 x 128 kernel at 18464 rows: N = 3072 (0.855 vs 0.971) is 5-13 % faster on the 128 x 128 kernel,
        // N = 768 / 2304 (0.855 vs 0.85) 4-19 % faster on pp2: the threshold moved from 0.88 to 0.93, profiles/r04_nt_kernel_choice_small_batch.log)
        const bool big = a.M > 128 && a.N > 128 && eff256 >= 0.93 * eff128;
        const bool persist_ok = t256 >= 512;   // the persistent form pays from two full rounds on (+1..3 %)
        // second-generation ping-pong kernel (gemm_nt_pp2.hip): -2.2 % against the persistent kernel below over the step's eleven
        // shape / epilogue classes at per-GPU batch 256, -3.6 % against the one-tile-per-workgroup form data-parallel runs take
        // (profiles/r04_nt_pp2_second_ab.log).  Its persistent launch (grid = CUs) is 0.9 % faster than one workgroup per tile on
        // the kernels alone and 0.8 % on the whole step (1303 vs 1293 pairs/s, profiles/r04_nt_pp2_stagger_and_persistent_ab.log);
        // data-parallel runs (M3AE_GEMM_NO_PERSISTENT) take the per-tile launch: static tile lists start late beside RCCL's kernels.
        if (big && a.rows_epi && a.K >= 256) {
            g_last_path = "mfma_nt_pp2";
            return launch_nt_pp2(a, EPI, persist_ok && !a.no_persist, s);
        }
        if (big && persist_ok && !a.no_persist && a.rows_epi && a.K >= 96) return launch_nt_pp_persistent<EPI>(a, s);  // +1..3 % (next tile's
        if (big) return launch_nt_pp<EPI>(a, s);                                                  // chunks under the epilogue)
        return launch_nt_t<128, 128, 64, 2, 64, EPI>(a, s);
    }
    if ((g_nt_variant == 9 || g_nt_variant == 10) && a.M > 128 && a.N > 128 && a.rows_epi && a.K >= 256) {   // gemm_nt_pp2.hip
        g_last_path = "mfma_nt_pp2";   // 9: grid = tiles; 10: persistent (grid = CUs)
        return launch_nt_pp2(a, EPI, g_nt_variant == 10, s);
    }
    if (g_nt_variant == 7 && a.M > 128 && a.N > 128) return launch_nt_pp<EPI>(a, s);  // ping-pong 8-phase
    if (g_nt_variant == 8 && a.rows_epi && a.K >= 96 && cdiv(a.M, 256) * cdiv(a.N, 256) >= 512)
        return launch_nt_pp_persistent<EPI>(a, s);                                    // persistent ping-pong
    if (g_nt_variant == 8 && a.M > 128 && a.N > 128) return launch_nt_pp<EPI>(a, s);
    if (g_nt_variant == 4 && a.M > 128 && a.N > 128) return launch_nt_t<256, 256, 64, 2, 128, EPI>(a, s);
    return launch_nt_t<128, 128, 64, 2, 64, EPI>(a, s);
}

static int launch_nt(const m3ae_gemm_desc& d, hipStream_t s) {
    MfmaArgs a{};
    a.A = (const bf16_t*)d.A; a.lda = d.a_sm;
    a.B = (const bf16_t*)d.B; a.ldb = d.b_sn;
    a.C = d.C; a.ldc = d.c_sm;
    a.M = d.M; a.N = d.N; a.K = d.K;
    a.c_f32 = d.dtype_c == M3AE_F32;
    a.alpha = d.alpha; a.accumulate = d.accumulate; a.bias = d.bias; a.act = d.act; a.preact = d.preact;
    a.preact_grad = d.preact_grad;
    a.residual = d.residual; a.dact_aux = d.dact_aux; a.dact = d.dact;
    a.rows_epi = (d.N % 8 == 0 && d.c_sm % 8 == 0) ? 1 : 0;
    {   // column tiles per group of the ping-pong kernels' tile order: all of them up to 9 (N <= 2304: B <= 3.4 MiB at
        // K = 768), else 6 -- measured against 3 / 4 / 12 on the path's shapes (profiles/r01_nt_col_group.log: -3..5 %)
        const int tiles_n = (int)((d.N + 255) / 256);
        const int g_nt_col_group = (d.launch_flags >> 16) & 0xf;
        a.col_group = g_nt_col_group > 0 ? g_nt_col_group : (tiles_n <= 9 ? tiles_n : 6);
    }
    a.has_drop = d.dropout_p > 0.f;
    a.drop = make_drop(d.dropout_p, d.dropout_seed, d.dropout_salt);
    a.no_persist = (d.launch_flags & M3AE_GEMM_NO_PERSISTENT) ? 1 : 0;
    a.nt_variant = ((d.launch_flags >> 8) & 0xf) - 1;
    a.st_policy = (d.launch_flags >> 20) & 0x3;
    // by shape: outputs of 32 MB and more (the image-side GEMMs: 113-900 MB at per-GPU batch 256) are written -- and their residual /
    // derivative operands read -- with the streaming policy: with the default one they evict the weight panel and the activation rows
    // the XCD's other tiles are about to read (-5.2 % over the step's shapes, profiles/r04_nt_store_cache_policy_ab.log); small
    // outputs (the text stream) keep the default: the next kernel finds them in L2 / Infinity Cache
    if (a.st_policy == 0) a.st_policy = (d.M * d.N >= (int64_t)16 << 20) ? 3 : 1;
    const bool has_act = d.act != M3AE_ACT_NONE, has_dact = d.dact_aux != nullptr;
    if (!has_act && !has_dact && !d.preact) return launch_nt_v<EPI_PLAIN>(a, s);
    if (!has_dact && d.act == M3AE_ACT_RELU) return launch_nt_v<EPI_RELU>(a, s);                       // dropout allowed
    if (!has_act && has_dact && d.dact == M3AE_ACT_MULAUX) return launch_nt_v<EPI_DMUL>(a, s);          // dropout allowed
    if (a.has_drop) return launch_nt_v<EPI_ANY>(a, s);
    if (!has_dact && d.act == M3AE_ACT_GELU) return launch_nt_v<EPI_GELU>(a, s);
    if (!has_dact && d.act == M3AE_ACT_QUICKGELU) return launch_nt_v<EPI_QGELU>(a, s);
    if (!has_act && d.dact == M3AE_ACT_GELU) return launch_nt_v<EPI_DGELU>(a, s);
    if (!has_act && d.dact == M3AE_ACT_QUICKGELU) return launch_nt_v<EPI_DQGELU>(a, s);
    return launch_nt_v<EPI_ANY>(a, s);
}


// ---------------------------------------------------------------------------------------------------------
// TN "ping-pong" kernel (wgrad): the structure of gemm_nt_pp_kernel with transposing fragment reads.  256 x 256 output
// tile, 8 waves (2 x 4, 128 x 64 each), 32 reduction rows per chunk ([32][256] bf16 image of each operand, 16 KiB) in a
// 4-slot ring; two phases per chunk (output rows 0-63 / 64-127 of the wave), the two wave rows staggered by one
// barrier; B(c+3) / A(c+3) issued in phases 2c / 2c+1, counted vmcnt(8) in the odd phase.  Split over the reduction
// with fp32 atomics; the bias gradient rides on a ones-fragment MFMA in the first column tile's waves.
// Ring depth: a 5-slot ring (160 KiB, three chunks in flight behind the one awaited) measured the same as 4 slots on every
// wgrad shape of the step (r02, profiles/r02_tn_ring_depth.log): the kernel is not waiting on load latency.  An L2 prefetch
// of chunk c + 4 / 8 / 16 (one 4-byte LDS-DMA per 64-B segment, one instruction per wave and chunk) made every shape 9 %
// SLOWER (profiles/r02_tn_l2_prefetch.log): one more LDS-DMA instruction per wave and chunk (5 instead of 4) costs ~240
// clocks of the chunk -- what bounds the loop is the issue of the DMA instructions themselves (in-loop stamps,
// tools/tn_trace.py: a wave row's "fragment reads + 2 DMA issues" phase takes 650-1000 clocks against 420 for the other
// row's 16 MFMAs).  Moving the two DMA issues of a phase INTO the wave's own MFMA cluster (after its first 4 MFMAs) is 10 %
// slower again (profiles/r02_tn_dma_in_mfma.log): a 1-KiB piece costs its wave ~250 clocks wherever it is issued while
// eight waves stage 32 KiB per chunk -- ~30 B / clk / CU of LDS-DMA issue, the same ceiling the fused cross-attention
// kernels run into.  Starting the tiles that share an operand panel apart in time (so that the later ones
// find the panel's lines in L2) changes nothing either: up to 11 us of skew is caught up within the launch, then the tiles
// run in step again (profiles/r02_tn_stagger.log).  Nor does the length of the contiguous global runs of a piece: rotating a
// row by tn_swz(row) slots instead of XOR-permuting its 16-B chunks (same bank-conflict freedom, 288-480-B runs instead of
// 32-B pairs) measured 825 / 868 / 854 / 726 TFLOP/s against 826-834 / 868-882 / 860-863 / 735 on the four wgrad shapes.
// And staging through REGISTERS instead of LDS-DMA (global_load_dwordx4 in iteration c, ds_write_b128 in c + 1, 16 staging
// registers per wave -- all the kernel has left) is 4 % slower: 784 / 825-834 / 808-811 / 707-714 against 818 / 857-863 /
// 849-853 / 732-736 (bit-identical results).  For calibration: torch.matmul (hipBLASLt) runs these four wgrad shapes at
// 699 / 721 / 550 / 347 TFLOP/s on the same box (tools/vendor_gemm_ref.py, profiles/r02_vendor_gemm_reference.log).
// ---------------------------------------------------------------------------------------------------------
#ifdef M3AE_TN_TRACE   // diagnostic build only: in-loop stamps of one chunk of the TN ping-pong kernel
__device__ uint64_t g_tn_trace[512 * 2 * 16];
#define TN_STAMP(k) do { if (tn_tr) g_tn_trace[((size_t)blockIdx.x * 2 + wr) * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
extern "C" int m3ae_tn_trace_dump(uint64_t* out) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tn_trace), sizeof(uint64_t) * 512 * 2 * 16);
    static uint64_t zeros[512 * 2 * 16];
    hipMemcpyToSymbol(HIP_SYMBOL(g_tn_trace), zeros, sizeof(zeros));
    return 0;
}
#else
#define TN_STAMP(k) do { } while (0)
#endif
__global__ __launch_bounds__(512, 2) void gemm_tn_pp_kernel(MfmaArgs a) {
    constexpr int CK = 32, NW = 8, A_BYTES = 256 * CK * 2, SLOT = 2 * A_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    const unsigned tiles_n = (unsigned)(a.N / 256);
    const unsigned tiles = (unsigned)(a.M / 256) * tiles_n;
    const unsigned wg = xcd_remap(blockIdx.x, gridDim.x);
    const unsigned tile_id = wg % tiles, split = wg / tiles;
    const int64_t m0 = (int64_t)(tile_id / tiles_n) * 256;
    const int64_t n0 = (int64_t)(tile_id % tiles_n) * 256;
    const int64_t r_begin = (int64_t)split * a.k_chunk;
    int64_t r_end = r_begin + a.k_chunk;
    if (r_end > a.K) r_end = a.K;
    if (r_begin >= r_end) return;  // whole workgroup: uniform
    const int nc = (int)((r_end - r_begin + CK - 1) / CK);

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool do_rowsum = a.a_rowsum != nullptr && n0 == 0 && wc == 0;
    f32x4 rsum[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) rsum[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const s16x8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};

#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (c < nc) {
            const int64_t r0 = r_begin + (int64_t)c * CK;
            tn_stage<256, 2, NW>(a.B, a.ldb, r0, r_end, n0, smem + c * SLOT + A_BYTES, wave, lane);
            tn_stage<256, 2, NW>(a.A, a.lda, r0, r_end, m0, smem + c * SLOT, wave, lane);
        }
    }
    if (nc >= 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (nc == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PP_FENCE();
    __builtin_amdgcn_s_barrier();
    PP_FENCE();
    if (wr == 1) { __builtin_amdgcn_s_barrier(); PP_FENCE(); }

    for (int c = 0; c < nc; ++c) {
#ifdef M3AE_TN_TRACE
        const bool tn_tr = (c == 64 || c == 65) && lane == 0 && wc == 0 && blockIdx.x < 512;
        if (tn_tr && c == 65) g_tn_trace[((size_t)blockIdx.x * 2 + wr) * 16 + 8] = __builtin_readcyclecounter();
#undef TN_STAMP
#define TN_STAMP(k) do { if (tn_tr && c == 64) g_tn_trace[((size_t)blockIdx.x * 2 + wr) * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
#endif
        TN_STAMP(0);
        const char* At = smem + (c & 3) * SLOT;
        const char* Bt = At + A_BYTES;
        char* nxt = smem + ((c + 3) & 3) * SLOT;
        const bool more = c + 3 < nc;
        const int64_t r3 = r_begin + (int64_t)(c + 3) * CK;
        s16x8 bfr[4], af[4];
        // ---------------- phase 2c: output rows 0..63 of the wave's sub-tile
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = tn_frag<256>(Bt, 0, wc * 64 + j * 16, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = tn_frag<256>(At, 0, wr * 128 + i * 16, lane);
        if (more) tn_stage<256, 2, NW>(a.B, a.ldb, r3, r_end, n0, nxt + A_BYTES, wave, lane);
        PP_FENCE();
        TN_STAMP(1);
        __builtin_amdgcn_s_barrier();
        PP_FENCE();
        TN_STAMP(2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8_t, af[i]), __builtin_bit_cast(bf16x8_t, bfr[j]), acc[i][j], 0, 0, 0);
        }
        if (do_rowsum) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                rsum[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8_t, af[i]), __builtin_bit_cast(bf16x8_t, ones), rsum[i], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        PP_FENCE();
        TN_STAMP(3);
        __builtin_amdgcn_s_barrier();
        PP_FENCE();
        TN_STAMP(4);
        // ---------------- phase 2c + 1: output rows 64..127 (the B fragments stay in registers)
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = tn_frag<256>(At, 0, wr * 128 + 64 + i * 16, lane);
        if (more) tn_stage<256, 2, NW>(a.A, a.lda, r3, r_end, m0, nxt, wave, lane);
        {
            const int rem = nc - 1 - c;
            if (rem >= 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (rem == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        PP_FENCE();
        TN_STAMP(5);
        __builtin_amdgcn_s_barrier();
        PP_FENCE();
        TN_STAMP(6);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8_t, af[i]), __builtin_bit_cast(bf16x8_t, bfr[j]), acc[4 + i][j], 0, 0, 0);
        }
        if (do_rowsum) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                rsum[4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8_t, af[i]), __builtin_bit_cast(bf16x8_t, ones), rsum[4 + i], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        PP_FENCE();
        __builtin_amdgcn_s_barrier();
        PP_FENCE();
    }
    if (wr == 0) { __builtin_amdgcn_s_barrier(); PP_FENCE(); }

    if (do_rowsum && (lane & 15) == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) atomicAdd(a.a_rowsum + m0 + wr * 128 + i * 16 + 4 * (lane >> 4) + r, rsum[i][r]);
    }
    // lane holds D[n1 = .. + 4 * (lane >> 4) + reg][n2 = .. + (lane & 15)]
    float* C = (float*)a.C;
    const bool atomic = a.splits > 1;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t n1 = m0 + wr * 128 + i * 16 + 4 * (lane >> 4) + r;
                const int64_t n2 = n0 + wc * 64 + j * 16 + (lane & 15);
                float* p = C + n1 * a.ldc + n2;
                const float x = acc[i][j][r] * a.alpha;
#ifdef M3AE_EXP_TN_NOATOMIC   // timing experiment: the split-K epilogue's share of the kernel
                asm volatile("" ::"v"(x), "v"(p));
#else
                if (atomic) atomicAdd(p, x);
                else if (a.accumulate) *p += x;
                else *p = x;
#endif
            }
}

static int launch_tn_pp(MfmaArgs a, const m3ae_gemm_desc& d, hipStream_t s) {
    constexpr int lds = 4 * (256 + 256) * 32 * 2;
    const int64_t tiles = (d.M / 256) * (d.N / 256);
    const int64_t ksteps = cdiv(d.K, 64);
    int64_t splits = 256 / tiles;  // one workgroup per CU, one round
    if (splits > ksteps / 8) splits = ksteps / 8;
    if (splits < 1) splits = 1;
    if (!d.accumulate) splits = 1;
    const int64_t steps_per = cdiv(ksteps, splits);
    a.k_chunk = steps_per * 64;
    splits = cdiv(ksteps, steps_per);
    a.splits = (int)splits;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_pp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    hipLaunchKernelGGL(gemm_tn_pp_kernel, dim3((unsigned)(tiles * splits)), dim3(512), lds, s, a);
    return hip_launch_status();
}

template <int BM_, int BN_, int WM, int BKR, int NST>
static int launch_tn_t(MfmaArgs a, const m3ae_gemm_desc& d, hipStream_t s) {
    constexpr int lds = NST * (BM_ + BN_) * BKR * 2;
    constexpr int threads = (BM_ / WM) * (BN_ / 64) * 64;
    const int64_t tiles = (d.M / BM_) * (d.N / BN_);
    const int64_t ksteps = cdiv(d.K, 64);
#ifdef M3AE_EXP_TN_TARGET   // timing experiment: split-K fan-out of the 128 x 128 kernel (workgroups per launch)
    const int64_t target = M3AE_EXP_TN_TARGET;
#else
    // workgroups in flight: 1 or ~3 per CU; a 768 x 768 output (36 tiles) is bound by its splits' fp32 atomic tiles, not by
    // parallelism: 10-14 splits instead of 17-21 measured -10..17 % at 18464 / 36928 rows (profiles/r04_tn_split_fanout_small_batch.log)
    // (long reductions keep the full fan-out: their atomics are a small share and 3 workgroups per CU balance better)
    const int64_t target = lds > 65536 ? 256 : (tiles <= 48 && ksteps <= 1024 ? 448 : 768);
#endif
    // >= 1024 reduction rows per split (K = 8192, 768 x 768: 42 -> 31.5 us).  Shorter splits on the short reductions of the text
    // stream at small per-GPU batches (1024-4096 rows) were measured in round 3 and LOSE: 256 rows per split took 37-43 us against
    // 24-32 us (profiles/r03_tn_small_batch.log): the extra workgroups' atomic tiles cost more than the parallelism buys
    constexpr int64_t min_steps = 16;
    int64_t splits = target / tiles;
    if (splits > ksteps / min_steps) splits = ksteps / min_steps;
    if (splits < 1) splits = 1;
    if (!d.accumulate) splits = 1;
    const int64_t steps_per = cdiv(ksteps, splits);
    a.k_chunk = steps_per * 64;
    splits = cdiv(ksteps, steps_per);
    a.splits = (int)splits;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)gemm_tn_bf16_kernel<BM_, BN_, WM, BKR, NST>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_tn_bf16_kernel<BM_, BN_, WM, BKR, NST>), dim3((unsigned)(tiles * splits)), dim3(threads), lds,
                       s, a);
    return hip_launch_status();
}

static int launch_tn(const m3ae_gemm_desc& d, hipStream_t s) {
    // C[M=N1][N=N2] += alpha * sum_k A[m][k] B[k][n] with a_sm == 1 (A stored [K][N1]) and b_sn == 1
    MfmaArgs a{};
    a.A = (const bf16_t*)d.A; a.lda = d.a_sk;
    a.B = (const bf16_t*)d.B; a.ldb = d.b_sk;
    a.C = d.C; a.ldc = d.c_sm;
    a.M = d.M; a.N = d.N; a.K = d.K;
    a.c_f32 = 1; a.alpha = d.alpha; a.accumulate = d.accumulate; a.a_rowsum = d.a_rowsum;
    // variant 1: 256x256 tile, 8 waves x (128x64): half the operand re-read traffic of the 128x128 tile
    const bool pp_ok = d.M % 256 == 0 && d.N % 256 == 0 && d.K >= 4096;
    const int g_tn_variant = ((d.launch_flags >> 12) & 0xf) - 1;
    if (g_tn_variant == 5 && pp_ok) return launch_tn_pp(a, d, s);
    if (g_tn_variant < 0 || g_tn_variant == 6) {
        // auto (default).  Re-measured in round 4 after both kernels' LDS-DMA moved to inline asm (mfma_tiles.h: the 128 x 128 kernel
        // gained as much as the ping-pong one): the 256 x 256 ping-pong kernel wins on the 768 x 3072 / 3072 x 768 outputs from 65536
        // reduction rows (+3..4 % at 110784-147712, equal at 73856) and nowhere else -- 2304 x 768 and 768 x 768 are equal at
        // 147712 and 5-15 % SLOWER on it at 36928-110784, the large outputs 8 % slower at 36928
        // (profiles/r04_tn_kernel_choice_by_batch.log).  Variant 6 = the rule of rounds 1-3, kept for A/B runs.
        const bool r3_rule = pp_ok && (d.K >= 65536 || (d.K >= 32768 && d.M * d.N >= 768 * 3072));
        const bool r4_rule = pp_ok && d.K >= 65536 && d.M * d.N >= 768 * 3072;
        if (g_tn_variant == 6 ? r3_rule : r4_rule) return launch_tn_pp(a, d, s);
        return launch_tn_t<128, 128, 64, 32, 2>(a, d, s);
    }
    if (g_tn_variant == 2) return launch_tn_t<128, 128, 64, 32, 2>(a, d, s);  // 32 KiB LDS: 3 workgroups / CU
    return launch_tn_t<128, 128, 64, 64, 2>(a, d, s);
}

#ifdef M3AE_NT_TRACE
extern "C" int m3ae_nt_trace_dump(uint64_t* host_out) {   // diagnostic build only
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_nt_trace), sizeof(uint64_t) * 1024 * 2 * 8);
}
#endif

extern "C" int m3ae_gemm(const m3ae_gemm_desc* dp, void* stream) {
    if (!dp || !dp->A || !dp->B || !dp->C) return M3AE_ERR_ARG;
    const m3ae_gemm_desc& d = *dp;
    if (d.M <= 0 || d.N <= 0 || d.K <= 0 || d.batch1 <= 0 || d.batch2 <= 0) return M3AE_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const bool bf = d.dtype_a == M3AE_BF16 && d.dtype_b == M3AE_BF16;
    const bool single = d.batch1 == 1 && d.batch2 == 1;
    if (bf && single && !d.force_generic && d.c_sn == 1) {
        const bool ptr_ok = aligned16(d.A) && aligned16(d.B) && aligned16(d.C) &&
                            (!d.preact || aligned16(d.preact)) && (!d.residual || aligned16(d.residual)) &&
                            (!d.dact_aux || aligned16(d.dact_aux)) && (!d.bias || aligned16(d.bias));
        // NT: both operands K-contiguous
        if (ptr_ok && !d.a_rowsum && d.a_sk == 1 && d.b_sk == 1 && d.K % BK == 0 && d.N % 4 == 0 && d.a_sm % 8 == 0 &&
            d.b_sn % 8 == 0 && d.c_sm % 4 == 0 && d.M >= 1) {
            g_last_path = "mfma_nt";
            return launch_nt(d, s);
        }
        // TN: both operands reduction-strided (wgrad), fp32 output
        if (ptr_ok && d.a_sm == 1 && d.b_sn == 1 && d.dtype_c == M3AE_F32 && d.M % BM == 0 && d.N % BN == 0 &&
            d.a_sk % 8 == 0 && d.b_sk % 8 == 0 && !d.bias && d.act == M3AE_ACT_NONE && !d.preact && !d.residual &&
            !d.dact_aux) {
            g_last_path = "mfma_tn";
            return launch_tn(d, s);
        }
    }
    g_last_path = "generic";
    return m3ae_gemm_generic(d, s);
}
