// NT "ping-pong" GEMM, second generation (round 4): C[M,N] = epi(alpha * A[M,K] . B[N,K]^T), both operands K-contiguous.
// Reference call sites: clip_model.py:44-49 (packed QKV / out_proj / c_fc / c_proj), bert_model.py:419-427, 434-441 (FFN), their
// dgrads against the transposed weight shadows.
//
// Same tile as gemm_nt_pp_kernel (256 x 256 per 512-thread workgroup, 8 waves as 2 x 4 of 128 x 64, 32-deep chunks in a 4-slot
// 128-KiB LDS ring, the two wave rows staggered by one barrier so that one wave per SIMD computes while the other fetches) and
// the same accumulation order (bit-identical results); what changed, and why (DESIGN.md 6e has the measurements):
//
//  * (Measured and NOT kept: one 32-MFMA phase per chunk instead of two 16-MFMA phases -- the in-loop stamps of round 3 suggested
//    the partner's fetch segment outlasts a 256-clock cluster; with a 512-clock cluster the kernel ran 1 % SLOWER on the step's
//    shapes, profiles/r04_nt_pp2_first_ab.log v10 / v9.  The wait for a phase's fragment reads in front of the barrier or behind
//    it: the same, profiles/r04_nt_pp2_second_ab.log v11 / v12.)
//  * LDS-DMA from inline asm.  With the builtin, hipcc tracks the DMA as an LDS store and puts s_waitcnt vmcnt(0) in front of
//    the first ds_write that follows it: the epilogue's slab writes waited for the next tile's prefetch to LAND before starting.
//  * The epilogue slabs live in the 32 KiB of LDS beside the ring (4 KiB per wave, XOR-swizzled instead of padded), so all FOUR
//    ring slots are prefetched for the next tile before the epilogue runs, and the epilogue's own loads (bias, residual /
//    derivative operand) are issued ahead of that prefetch: vmcnt retires in order, and a chunk requested BEHIND the epilogue's
//    stores cannot land before they have drained (~128 KiB per tile and CU) -- the old schedule requested chunk 2 there and needed
//    it two chunks into the next main loop; now the first chunk behind the stores is chunk 4.
//  * One kernel for both launch forms: grid = CUs (persistent, tiles v = block, block + grid, ...) or grid = tiles.
#include "gemm_nt_common.h"

using namespace m3g;

#ifdef M3AE_EXP_PP2_CLOCK   // diagnostic build only (tools/nt_clock.py): the clock the chip holds inside this kernel (MI355X_MICROARCH.md,
                            // 'DVFS give-back' item 6): shader cycles (s_memtime) over 100-MHz ticks (s_memrealtime) across the tile loop
__device__ uint64_t g_pp2_clock[2 * 256];
extern "C" int m3ae_diag_pp2_clock(uint64_t* out, int n) {   // n <= 512 values: [workgroup][cycles, ticks]
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pp2_clock), (size_t)n * sizeof(uint64_t)) == hipSuccess ? 0 : 1;
}
#endif

namespace {

// wait until at most base + s (s = 0 / 16 / 32: the stores of the previous tile's epilogue, wave-uniform) vector-memory operations
// of this wave are outstanding
template <int BASE> DEVINL void wait_vm_s(int s) {
    if (s == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BASE) : "memory");
    else if (s == 16) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BASE + 16) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BASE + 32) : "memory");
}

// one 1-KiB LDS-DMA piece (64 lanes x 16 B, lane-linear at the wave-uniform LDS byte address dst); M0 is written and restored in the
// same statement (hipcc reserves it).  Invisible to hipcc's vmcnt bookkeeping: every wait for these pieces is explicit (wait_vm).
// Source address = wave-uniform base (SGPR pair) + this lane's 32-bit byte offset: one VGPR per piece stays live across the main loop.
DEVINL void glds16_asm(const void* base, unsigned voff, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(dst) : "memory");
}

// Epilogue of one 128 x 64 wave sub-tile through the wave's private 4-KiB slab [16 rows][64 fp32], 16-B chunk c of row r stored at
// chunk c ^ r: the accumulator-layout writes (lane = row, 4 columns) and the row-contiguous reads (8 lanes = one row, 8 columns
// each) are both bank-conflict free without padding.  LDS operations of one wave execute in order: no waits between the slab's
// writes and reads beyond the data dependences hipcc tracks itself.
// The epilogue's own loads (bias, residual / derivative operand) are issued from inline asm as well, and waited for by ONE counted
// s_waitcnt of the kernel's own.  As compiler-visible loads they sat under run-time conditions (operand present, row in range), and so
// did their consumers; hipcc merges those paths conservatively and (the .s of the first pp2 build shows it) put
//   - s_waitcnt vmcnt(0) in front of the bias add and the residual add of EVERY one of the 16 row blocks of the epilogue: each block
//     waited for the previous block's output stores to be acknowledged (and for the next tile's prefetch to land), and
//   - s_waitcnt vmcnt(0) in front of the first fragment read of the next tile that reused one of those registers: the three
//     chunks prefetched ahead and the epilogue's stores were drained at every tile start, behind the counted wait meant to keep
//     them in flight.
// Now: loads (unconditional, addresses clamped into the operand: rows / columns past the edge are loaded but never used) ->
// the next tile's 16 prefetch pieces -> first slab writes -> s_waitcnt vmcnt(16) (vmcnt retires in order: the loads are the
// older operations) -> 16 row blocks that only compute and store.
template <typename TC, int EPI>
struct EpiLoads {
    u32x4 bias[2];
    u32x4 pre[8][2];
    bool has_bias, has_pre;
};

template <typename TC, int EPI>
DEVINL void epi_issue_loads(const MfmaArgs& a, int lane, int64_t m_base, int64_t n_base, EpiLoads<TC, EPI>& L) {
    constexpr bool BF = sizeof(TC) == 2;
    constexpr bool PRE_IS_AUX = (EPI == EPI_DGELU || EPI == EPI_DQGELU || EPI == EPI_DMUL);
    int64_t ncol = n_base + (lane & 7) * 8;
    ncol = ncol < a.N ? ncol : 0;   // N % 8 == 0: a lane's 8 columns are all inside or all outside
    L.bias[0] = L.bias[1] = (u32x4){0u, 0u, 0u, 0u};
    L.has_bias = a.bias != nullptr;
    const TC* src = (const TC*)(PRE_IS_AUX ? a.dact_aux : a.residual);
    L.has_pre = BF && src != nullptr;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int p = 0; p < 2; ++p) L.pre[i][p] = (u32x4){0u, 0u, 0u, 0u};
    const int64_t m_last = a.M - 1;
    if (EPI == EPI_ANY) {   // the catch-all instantiation spills registers: compiler-visible loads (see gemm_nt_common.h, epilogue_rows)
        if (L.has_bias) {
            L.bias[0] = *(const u32x4*)(a.bias + ncol);
            L.bias[1] = *(const u32x4*)(a.bias + ncol + 4);
        }
        if (L.has_pre) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int64_t m = m_base + 16 * i + p * 8 + (lane >> 3);
                    L.pre[i][p] = *(const u32x4*)(src + (m < m_last ? m : m_last) * a.ldc + ncol);
                }
        }
        return;
    }
    if (L.has_bias) {
        gload16_asm(L.bias[0], a.bias + ncol);
        gload16_asm(L.bias[1], a.bias + ncol + 4);
    }
    if (L.has_pre) {
        if (a.st_policy >= 3) {   // the residual / derivative operand is read once: streaming policy
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int64_t m = m_base + 16 * i + p * 8 + (lane >> 3);
                    gload16_asm_nt(L.pre[i][p], src + (m < m_last ? m : m_last) * a.ldc + ncol);
                }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int64_t m = m_base + 16 * i + p * 8 + (lane >> 3);
                    gload16_asm(L.pre[i][p], src + (m < m_last ? m : m_last) * a.ldc + ncol);
                }
        }
    }
}

// younger = the prefetch pieces issued behind the loads (16, or none behind the last tile)
template <typename TC, int EPI>
DEVINL void epi_wait_loads(EpiLoads<TC, EPI>& L, bool prefetched) {
    if (EPI == EPI_ANY) return;   // compiler-visible loads: hipcc places the waits
    if (prefetched) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // the loaded registers are "redefined" behind the wait: nothing that reads them can be scheduled in front of it
    asm volatile("" : "+v"(L.bias[0]), "+v"(L.bias[1]));
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(L.pre[i][0]), "+v"(L.pre[i][1]));
}

template <typename TC, int EPI>
DEVINL void epi_finish(const MfmaArgs& a, float* slab, int lane, int64_t m_base, int64_t n_base, f32x4 (&acc)[8][4],
                       EpiLoads<TC, EPI>& L, bool prefetched) {
    const int wrow = lane & 15, wq = lane >> 4;
    const int rq = lane & 7;
    float bias8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) *(f32x4*)(slab + wrow * 64 + (((4 * j + wq) ^ wrow) << 2)) = acc[i][j];
        if (i == 0) {
            epi_wait_loads(L, prefetched);
#pragma unroll
            for (int t = 0; t < 8; ++t) {   // (through a copy: __builtin_bit_cast of a vector-element lvalue reads element 0, hipcc 7.2)
                const u32x4 v = L.bias[t >> 2];
                bias8[t] = __uint_as_float(v[t & 3]);
            }
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int row = p * 8 + (lane >> 3);
            const f32x4 v0 = *(const f32x4*)(slab + row * 64 + (((2 * rq) ^ row) << 2));
            const f32x4 v1 = *(const f32x4*)(slab + row * 64 + (((2 * rq + 1) ^ row) << 2));
            float x[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            const int64_t m = m_base + 16 * i + row, n = n_base + rq * 8;
            if (m < a.M && n < a.N) epilogue8<TC, EPI>(a, m, n, x, bias8, L.has_pre, L.pre[i][p]);
        }
    }
}

// Built, measured, NOT kept (round 4, second half; both bit-identical to this loop, in the tree at commit 39eb520 as variants 11-14):
//  * a software-pipelined main loop -- every wave requests the fragments of its NEXT 16-MFMA group before issuing the current one
//    (double-buffered B / A fragments, 253 VGPRs, no spills), ONE barrier per chunk, the two wave rows issuing their DMA on opposite
//    sides of the second group: 9 % SLOWER over the step's eleven classes (profiles/r04_nt_pp2_pipelined_loop_ab.log);
//  * this ping-pong schedule (four barriers, rows staggered) WITH that fragment prefetch, so that a wave's segment between two of its
//    MFMA clusters is DMA issue + waits only: the SAME time (6554 vs 6583 us, profiles/r04_nt_pp2_pingpong_prefetch_ab.log);
//  * a third fewer LDS fragment reads (timing-only build, wrong results): -2 % (profiles/r04_nt_pp2_half_lds_reads_timing.log).
// Neither the fragment-read latency nor the barrier count bounds this loop.  What the chip does under it (tools/nt_clock.py,
// profiles/r04_nt_in_kernel_clock.log): it holds 1.81-1.94 GHz on random data (2.38 GHz on zero-filled operands, +20 % TF/s at the same
// cycle count), i.e. the kernel delivers 0.52 (K = 768) to 0.62 (K = 3072) of the bf16 MFMA rate AT THE CLOCK IT RUNS AT, and cycles
// saved in the loop come back partly as a lower clock (MI355X_MICROARCH.md, 'DVFS give-back').
template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt_pp2_kernel(MfmaArgs a) {
    if (a.has_drop) drop_resolve(a.drop);
    constexpr int CK = 32, A_BYTES = 256 * CK * 2, SLOT = 2 * A_BYTES, RING = 4 * SLOT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const unsigned tiles_n = (unsigned)((a.N + 255) / 256);
    const unsigned tiles_m = (unsigned)((a.M + 255) / 256);
    const unsigned total = tiles_m * tiles_n;
    const int nc = (int)(a.K / CK);
    constexpr int npre = 4;   // nc >= 8 (host check)
    const int frow = lane & 15, fchunk = lane >> 4;
    const unsigned lds_wave = (unsigned)(uintptr_t)(lds_void*)smem + (unsigned)wave * 1024u;
    unsigned v = blockIdx.x;
    if (v >= total) return;
#ifdef M3AE_EXP_PP2_STAGGER   // timing experiment (tools/nt_exp.sh): de-phase the workgroups' tile boundaries by quarter tiles
    {
        const unsigned k = (blockIdx.x >> 3) & 3u;   // same XCD (blockIdx & 7), neighbouring workgroups differ
        const uint64_t t0 = __builtin_readcyclecounter();
        while (__builtin_readcyclecounter() - t0 < (uint64_t)k * M3AE_EXP_PP2_STAGGER) __builtin_amdgcn_s_sleep(32);
    }
#endif
    unsigned tm, tn;
    nt_tile_coords(xcd_remap(v, total), tiles_m, tiles_n, tm, tn, (unsigned)a.col_group);
    int64_t m0 = (int64_t)tm * 256, n0 = (int64_t)tn * 256;

    // the wave's four pieces of a chunk: rows (q * 8 + wave) * 16 + lane / 4 (q = 0, 1) of either operand; source = the tile's
    // first row (scalar base, advanced by the chunk) + this lane's byte offset inside the tile (rows past the edge are clamped:
    // duplicated rows are computed but never stored)
    const bf16_t *abase, *bbase;
    unsigned va0, va1, vb0, vb1;
    auto set_ptrs = [&]() {
        int lane_p = lane;   // opaque copy: the lane-dependent parts are recomputed per tile, not carried through the main loop
        asm volatile("" : "+v"(lane_p));
        const int r0 = wave * 16 + (lane_p >> 2), r1 = r0 + 128;
        const unsigned c0 = (unsigned)(((lane_p & 3) ^ nt_swz<CK>(r0)) * 16), c1 = (unsigned)(((lane_p & 3) ^ nt_swz<CK>(r1)) * 16);
        const int ma = (int)(a.M - 1 - m0 < 255 ? a.M - 1 - m0 : 255), mb = (int)(a.N - 1 - n0 < 255 ? a.N - 1 - n0 : 255);
        abase = a.A + m0 * a.lda;
        bbase = a.B + n0 * a.ldb;
        va0 = (unsigned)(r0 < ma ? r0 : ma) * (unsigned)a.lda * 2u + c0;
        va1 = (unsigned)(r1 < ma ? r1 : ma) * (unsigned)a.lda * 2u + c1;
        vb0 = (unsigned)(r0 < mb ? r0 : mb) * (unsigned)a.ldb * 2u + c0;
        vb1 = (unsigned)(r1 < mb ? r1 : mb) * (unsigned)a.ldb * 2u + c1;
    };
    // (nt on these loads was measured too: -30 % -- the tiles of an XCD share both operands through its L2,
    // profiles/r04_nt_dma_nt_policy_measured.log)
    auto issue_b = [&](int c) {
        const unsigned dst = lds_wave + (unsigned)(c & 3) * SLOT + A_BYTES;
        glds16_asm(bbase + (int64_t)c * CK, vb0, dst);
        glds16_asm(bbase + (int64_t)c * CK, vb1, dst + 8192u);
    };
    auto issue_a = [&](int c) {
        const unsigned dst = lds_wave + (unsigned)(c & 3) * SLOT;
        glds16_asm(abase + (int64_t)c * CK, va0, dst);
        glds16_asm(abase + (int64_t)c * CK, va1, dst + 8192u);
    };
    set_ptrs();
    for (int c = 0; c < npre; ++c) { issue_b(c); issue_a(c); }

#ifdef M3AE_EXP_PP2_CLOCK
    const uint64_t clk_c0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    int s_prev = 0;   // stores the previous tile's epilogue issued behind this tile's prefetch (counted only when known exactly)
    const int s_interior = a.c_f32 || a.accumulate ? 0 : (a.preact ? 32 : 16);
    for (;;) {
        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // chunk 0 has landed: younger operations of this wave = the other prefetched chunks + the previous epilogue's stores
        wait_vm_s<12>(s_prev);
        PP_FENCE();
        __builtin_amdgcn_s_barrier();
        PP_FENCE();
        if (wr == 1) { __builtin_amdgcn_s_barrier(); PP_FENCE(); }   // stagger the second wave row by one barrier

        for (int c = 0; c < nc; ++c) {
            // the slot's byte offset is kept opaque: hipcc peels the first three chunks (their waits differ) and, knowing the slot there,
            // materialises one address VGPR per fragment for the slots beyond the 64-KiB reach of the ds_read offset field (~12 VGPRs
            // held across the whole kernel; with the epilogue's 200+ live registers that meant spill reloads inside the main loop)
            unsigned slot_off = (unsigned)(c & 3) * SLOT;
            asm volatile("" : "+s"(slot_off));
            const char* At = smem + slot_off;
            const char* Bt = At + A_BYTES;
            const bool issue = c >= 1 && c + 3 < nc;   // chunk c + 3 -> the slot of chunk c - 1 (read in the previous phase; every
                                                       // wave's reads were retired by its lgkmcnt(0) BEFORE a barrier this wave passed)
            // chunk c + 1 must have landed before the next phase reads it.  vmcnt retires in order: the wait names how many YOUNGER
            // operations may stay in flight = chunks c + 2, c + 3 (4 pieces each, as far as issued) + -- while chunk c + 1 is one of
            // the four prefetched ahead of the previous epilogue -- that epilogue's stores
            auto wait_next = [&]() {
                if (c >= 3 && c + 3 < nc) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if (c < 3) wait_vm_s<8>(s_prev);
                else if (c == nc - 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else if (c == nc - 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            };
            s16x8 bfr[4];
            {
                s16x8 af[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) bfr[j] = nt_frag<CK>(Bt, wc * 64 + j * 16 + frow, fchunk);
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = nt_frag<CK>(At, wr * 128 + i * 16 + frow, fchunk);
                if (issue) issue_b(c + 3);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // before the barrier: measured the same as behind it, and
                PP_FENCE();                                          // every wave's reads are retired when a partner passes
                __builtin_amdgcn_s_barrier();
                PP_FENCE();
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
#ifndef M3AE_EXP_PP2_HALFREADS   // timing experiment (wrong results): the second phase reuses the first phase's A fragments = the LDS
                                 // read traffic of a 128 x 128-per-wave layout (8 instead of 12 KiB per wave and chunk)
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = nt_frag<CK>(At, wr * 128 + 64 + i * 16 + frow, fchunk);
#endif
                if (issue) issue_a(c + 3);
                wait_next();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // before the barrier: measured the same as behind it, and
                PP_FENCE();                                          // every wave's reads are retired when a partner passes
                __builtin_amdgcn_s_barrier();
                PP_FENCE();
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
        }
        if (wr == 0) { __builtin_amdgcn_s_barrier(); PP_FENCE(); }   // re-align: every fragment read of the tile is retired, no DMA
                                                                     // of this tile is outstanding: all four slots are free
        const unsigned vn = v + gridDim.x;
        const bool again = vn < total;
        const int64_t m_cur = m0 + wr * 128, n_cur = n0 + wc * 64;
        const bool interior = m0 + 256 <= a.M && n0 + 256 <= a.N;
        auto prefetch_next = [&]() {
            if (again) {
                nt_tile_coords(xcd_remap(vn, total), tiles_m, tiles_n, tm, tn, (unsigned)a.col_group);
                m0 = (int64_t)tm * 256; n0 = (int64_t)tn * 256;
                set_ptrs();
                for (int c = 0; c < npre; ++c) { issue_b(c); issue_a(c); }
            }
        };
        // the epilogue's lane-dependent addresses are recomputed per tile from an opaque copy of the lane id: hipcc would otherwise
        // hoist them out of the tile loop and carry them (or their spills) through the main loop
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        float* slab = (float*)(smem + RING) + wave * 1024;
        if (a.c_f32) {
            EpiLoads<float, EPI> L;
            epi_issue_loads<float, EPI>(a, lane_e, m_cur, n_cur, L);
            prefetch_next();
            epi_finish<float, EPI>(a, slab, lane_e, m_cur, n_cur, acc, L, again);
        } else {
            EpiLoads<bf16_t, EPI> L;
            epi_issue_loads<bf16_t, EPI>(a, lane_e, m_cur, n_cur, L);
            prefetch_next();
            epi_finish<bf16_t, EPI>(a, slab, lane_e, m_cur, n_cur, acc, L, again);
        }
#ifdef M3AE_EXP_PP2_CLOCK
        if (!again && tid == 0 && blockIdx.x < 256) {
            g_pp2_clock[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - clk_c0;
            g_pp2_clock[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - clk_r0;
        }
#endif
        if (!again) break;
        s_prev = interior ? s_interior : 0;
        v = vn;
    }
}

template <int EPI>
int launch_pp2_t(const MfmaArgs& a, bool persistent, hipStream_t s) {
    constexpr int lds = 4 * (256 + 256) * 32 * 2 + 8 * 4096;   // 128-KiB ring + 8 slabs = all 160 KiB
    static bool attr_set = false;
    static int cus = 256;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_pp2_kernel<EPI>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            cus = prop.multiProcessorCount / 8 * 8;
        attr_set = true;
    }
    const int64_t tiles = cdiv(a.M, 256) * cdiv(a.N, 256);
    const unsigned grid = (unsigned)(persistent && tiles > cus ? cus : tiles);
    hipLaunchKernelGGL((gemm_nt_pp2_kernel<EPI>), dim3(grid), dim3(512), lds, s, a);
    return hip_launch_status();
}

int launch_pp2_e(const MfmaArgs& a, int epi, bool persistent, hipStream_t s) {
    switch (epi) {
        case EPI_PLAIN: return launch_pp2_t<EPI_PLAIN>(a, persistent, s);
        case EPI_GELU: return launch_pp2_t<EPI_GELU>(a, persistent, s);
        case EPI_QGELU: return launch_pp2_t<EPI_QGELU>(a, persistent, s);
        case EPI_DGELU: return launch_pp2_t<EPI_DGELU>(a, persistent, s);
        case EPI_DQGELU: return launch_pp2_t<EPI_DQGELU>(a, persistent, s);
        case EPI_DMUL: return launch_pp2_t<EPI_DMUL>(a, persistent, s);
        case EPI_RELU: return launch_pp2_t<EPI_RELU>(a, persistent, s);
        default: return launch_pp2_t<EPI_ANY>(a, persistent, s);
    }
}

}  // namespace

// preconditions (checked by the caller, gemm_mfma.hip::launch_nt_v): rows_epi (N % 8 == 0, ldc % 8 == 0), K % 32 == 0, K >= 256, M, N > 128
int m3g::launch_nt_pp2(const MfmaArgs& a, int epi, bool persistent, hipStream_t s) { return launch_pp2_e(a, epi, persistent, s); }
