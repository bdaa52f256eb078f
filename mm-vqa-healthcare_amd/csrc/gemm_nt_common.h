// Shared by the NT GEMM translation units (gemm_mfma.hip, gemm_nt_pp2.hip): argument block, epilogue classes, tile order,
// the fused epilogues on the MFMA accumulator layout.
#pragma once
#include "mfma_tiles.h"

namespace m3g {

// epilogue classes (template parameter: keeps erf/exp code out of the kernels that do not need it)
// EPI_RELU: the T5 feed-forward (ReLU, derivative saved, dropout after the activation); with EPI_DMUL (+ dropout) it keeps the
// T5 head off EPI_ANY, whose run-time activation switch spills (784 B of scratch per lane: 308 vs ~900 TFLOP/s)
enum { EPI_PLAIN = 0, EPI_GELU = 1, EPI_QGELU = 2, EPI_DGELU = 3, EPI_DQGELU = 4, EPI_ANY = 5, EPI_DMUL = 6, EPI_RELU = 7 };

struct MfmaArgs {
    const bf16_t* A; int64_t lda;
    const bf16_t* B; int64_t ldb;
    void* C; int64_t ldc;
    int64_t M, N, K;
    int c_f32;
    float alpha;
    int accumulate;
    const float* bias;
    int act;
    void* preact;
    int preact_grad;  // store act'(pre) instead of pre
    const void* residual;
    const void* dact_aux;
    int dact;
    float* a_rowsum;  // TN only: fp32 [N1] += column sums of A (bias gradient)
    DropState drop;   // NT only: dropout on (acc + bias), before the residual add
    int has_drop;
    int rows_epi;     // NT only: LDS-transposed row-contiguous epilogue (N % 8 == 0)
    int splits;       // TN only
    int64_t k_chunk;  // TN only: reduction rows per split (multiple of BK)
    int col_group;      // ping-pong NT kernels: column tiles per group of the tile order (nt_tile_coords)
    int no_persist;     // desc.launch_flags & M3AE_GEMM_NO_PERSISTENT
    int nt_variant;     // desc.launch_flags selector (-1: by shape)
    int st_policy;      // cache policy of the epilogue's streams: 1 plain; 2 output stores nt; 3 stores nt + residual / aux loads nt
                        // (launch_nt resolves the desc.launch_flags selector 0 = by shape)
};

// Tile order inside the (XCD-contiguous) id range: column tiles in groups of GC, row-major inside a group.  An XCD then
// works against <= GC column tiles of B (resident in its 4-MiB L2) while A streams, instead of cycling through all of B
// for every row block (N = 3072: B = 4.7 MiB thrashed the L2 -- 7x the algorithmic reads, PMC); A is re-read once per
// group.  GC: 4 for the generic kernels, by shape for the ping-pong kernels (launch_nt).
DEVINL void nt_tile_coords(unsigned wg, unsigned tiles_m, unsigned tiles_n, unsigned& tm, unsigned& tn, unsigned GC = 4) {
    const unsigned per_group = tiles_m * GC;
    const unsigned g = wg / per_group;
    const unsigned first = g * GC;
    const unsigned gc = tiles_n - first < GC ? tiles_n - first : GC;
    const unsigned local = wg - g * per_group;
    tm = local / gc;
    tn = first + local % gc;
}

// ---------------------------------------------------------------------------------------------------------
// NT kernel
// ---------------------------------------------------------------------------------------------------------
template <typename TC> struct Vec4;
template <> struct Vec4<float> {
    static DEVINL void ld(const float* p, float* x) {
        const f32x4 v = *(const f32x4*)p;
        x[0] = v[0]; x[1] = v[1]; x[2] = v[2]; x[3] = v[3];
    }
    static DEVINL void st(float* p, const float* x) { *(f32x4*)p = (f32x4){x[0], x[1], x[2], x[3]}; }
};
template <> struct Vec4<bf16_t> {
    static DEVINL void ld(const bf16_t* p, float* x) {
        const u32x2 v = *(const u32x2*)p;
        x[0] = __uint_as_float(v[0] << 16); x[1] = __uint_as_float(v[0] & 0xffff0000u);
        x[2] = __uint_as_float(v[1] << 16); x[3] = __uint_as_float(v[1] & 0xffff0000u);
    }
    static DEVINL void st(bf16_t* p, const float* x) { *(u32x2*)p = (u32x2){pack2bf(x[0], x[1]), pack2bf(x[2], x[3])}; }
};

// Fused epilogue on 4 consecutive n of row m (8-byte bf16 / 16-byte fp32 accesses).
template <typename TC, int EPI>
DEVINL void epilogue4(const MfmaArgs& a, int64_t m, int64_t n, f32x4 v) {
    const int64_t off = m * a.ldc + n;
    float x[4] = {v[0] * a.alpha, v[1] * a.alpha, v[2] * a.alpha, v[3] * a.alpha};
    float y[4];
    if (a.bias) {
        Vec4<float>::ld(a.bias + n, y);
#pragma unroll
        for (int t = 0; t < 4; ++t) x[t] += y[t];
    }
    if (EPI == EPI_GELU || EPI == EPI_QGELU || EPI == EPI_RELU || EPI == EPI_ANY) {
        const int act = EPI == EPI_GELU ? M3AE_ACT_GELU : (EPI == EPI_QGELU ? M3AE_ACT_QUICKGELU : (EPI == EPI_RELU ? M3AE_ACT_RELU : a.act));
        if (a.preact && a.preact_grad) {
            float dd[4];
            act_fwd_grad_fast_n<4>(x, dd, act);
            Vec4<TC>::st((TC*)a.preact + off, dd);
        } else {
            if (a.preact) Vec4<TC>::st((TC*)a.preact + off, x);
            act_fwd_fast_n<4>(x, act);
        }
    }
    if ((EPI == EPI_PLAIN || EPI == EPI_ANY || EPI == EPI_RELU || EPI == EPI_DMUL) && a.has_drop) {  // dropout follows a plain dense layer on this path
        drop_apply4(a.drop, (uint64_t)(m * a.N + n), x);  // N % 4 == 0 on this path: ld = N
    }
    if (a.residual) {
        Vec4<TC>::ld((const TC*)a.residual + off, y);
#pragma unroll
        for (int t = 0; t < 4; ++t) x[t] += y[t];
    }
    if (EPI == EPI_DMUL) {
        Vec4<TC>::ld((const TC*)a.dact_aux + off, y);
#pragma unroll
        for (int t = 0; t < 4; ++t) x[t] *= y[t];
    }
    if (EPI == EPI_DGELU || EPI == EPI_DQGELU || EPI == EPI_ANY) {
        if (a.dact_aux) {
            const int dact = EPI == EPI_DGELU ? M3AE_ACT_GELU : (EPI == EPI_DQGELU ? M3AE_ACT_QUICKGELU : a.dact);
            Vec4<TC>::ld((const TC*)a.dact_aux + off, y);
            act_bwd_mul_fast_n<4>(x, y, dact);
        }
    }
    if (a.accumulate) {
        Vec4<TC>::ld((const TC*)a.C + off, y);
#pragma unroll
        for (int t = 0; t < 4; ++t) x[t] += y[t];
    }
    Vec4<TC>::st((TC*)a.C + off, x);
}

template <typename TC> struct Vec8;
template <> struct Vec8<float> {
    static DEVINL void unpack(const u32x4&, float*) {}  // fp32 C tiles are never prefetched
    static DEVINL void ld(const float* p, float* x) { Vec4<float>::ld(p, x); Vec4<float>::ld(p + 4, x + 4); }
    static DEVINL void st(float* p, const float* x) { Vec4<float>::st(p, x); Vec4<float>::st(p + 4, x + 4); }
    static DEVINL void st_policy(float* p, const float* x, int) { st(p, x); }
};
template <> struct Vec8<bf16_t> {
    static DEVINL void unpack(const u32x4& v, float* x) {
#pragma unroll
        for (int t = 0; t < 4; ++t) { x[2 * t] = __uint_as_float(v[t] << 16); x[2 * t + 1] = __uint_as_float(v[t] & 0xffff0000u); }
    }
    static DEVINL void ld(const bf16_t* p, float* x) { unpack(*(const u32x4*)p, x); }
    static DEVINL void st(bf16_t* p, const float* x) {
        *(u32x4*)p = (u32x4){pack2bf(x[0], x[1]), pack2bf(x[2], x[3]), pack2bf(x[4], x[5]), pack2bf(x[6], x[7])};
    }
    // Output stores with a cache policy (wave-uniform).  >= 2: nt (streaming: the line is the first to leave the XCD's L2).  The
    // output of a large GEMM (113-900 MB at the step's shapes) is never re-read by the kernel itself, but written with the default
    // policy it evicts the weight panel and the activation rows the XCD's other tiles are about to read: -11 % on 147712 x 2304
    // x 768, -4.5 % over the step's eleven shape / epilogue classes, -5.2 % with the residual / derivative operand loaded nt as
    // well (profiles/r04_nt_store_cache_policy_ab.log; sc1 = write-through stores measured the same as plain ones).
    static DEVINL void st_policy(bf16_t* p, const float* x, int policy) {
        const u32x4 v = (u32x4){pack2bf(x[0], x[1]), pack2bf(x[2], x[3]), pack2bf(x[4], x[5]), pack2bf(x[6], x[7])};
        // inline asm: hipcc sinks the two arms' stores of a __builtin_nontemporal_store / plain pair into ONE plain store (the
        // nontemporal flag is dropped when the instructions are merged); "s_nop 1": the data registers stay valid until read
        if (policy >= 2) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
        else *(u32x4*)p = v;
    }
};

// Fused epilogue on 8 consecutive n of row m: 16-byte bf16 (2 x 16-byte fp32) accesses, 8 lanes = one 128-B line.
// bias8: the 8 bias values of columns n .. n + 7 (preloaded once per wave); pre: the 16 bytes of the residual (or,
// for the dact classes, of dact_aux) at (m, n .. n + 7), fetched before the LDS transposes so that the epilogue pays
// the global-load latency once per wave instead of once per 32-row pass (bf16 C only; nullptr = load here).
// 16-B global loads from inline asm: invisible to hipcc's vmcnt bookkeeping, so the kernel places its own (counted) wait and fences
// the destination registers behind it with an empty asm that "redefines" them.  The destination is tied ("+v"): a load skipped on a
// wave-uniform condition leaves the caller's initial value.
DEVINL void gload16_asm(u32x4& d, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(d) : "v"(p) : "memory"); }
DEVINL void gload16_asm_nt(u32x4& d, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off nt" : "+v"(d) : "v"(p) : "memory"); }

template <typename TC, int EPI>
DEVINL void epilogue8(const MfmaArgs& a, int64_t m, int64_t n, float* x, const float* bias8, bool has_pre,
                      const u32x4 pre) {
    constexpr bool PRE_IS_AUX = (EPI == EPI_DGELU || EPI == EPI_DQGELU || EPI == EPI_DMUL);
    const int64_t off = m * a.ldc + n;
    float y[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) x[t] *= a.alpha;
    if (a.bias) {
#pragma unroll
        for (int t = 0; t < 8; ++t) x[t] += bias8[t];
    }
    if (EPI == EPI_GELU || EPI == EPI_QGELU || EPI == EPI_RELU || EPI == EPI_ANY) {
        const int act = EPI == EPI_GELU ? M3AE_ACT_GELU : (EPI == EPI_QGELU ? M3AE_ACT_QUICKGELU : (EPI == EPI_RELU ? M3AE_ACT_RELU : a.act));
        if (a.preact && a.preact_grad) {
            float dd[8];
            act_fwd_grad_fast_n<8>(x, dd, act);
#ifdef M3AE_EXP_NT_NOSTORE
#pragma unroll
            for (int t = 0; t < 8; ++t) asm volatile("" ::"v"(dd[t]));
#else
            Vec8<TC>::st_policy((TC*)a.preact + off, dd, a.st_policy);
#endif
        } else {
            if (a.preact) Vec8<TC>::st_policy((TC*)a.preact + off, x, a.st_policy);
            act_fwd_fast_n<8>(x, act);
        }
    }
    if ((EPI == EPI_PLAIN || EPI == EPI_ANY || EPI == EPI_RELU || EPI == EPI_DMUL) && a.has_drop) {  // dropout follows a plain dense layer on this path
        drop_apply4(a.drop, (uint64_t)(m * a.N + n), x);
        drop_apply4(a.drop, (uint64_t)(m * a.N + n + 4), x + 4);
    }
    if (a.residual) {
        if (has_pre && !PRE_IS_AUX) Vec8<TC>::unpack(pre, y);
        else Vec8<TC>::ld((const TC*)a.residual + off, y);
#pragma unroll
        for (int t = 0; t < 8; ++t) x[t] += y[t];
    }
    if (EPI == EPI_DMUL) {  // the saved tensor already holds act'(pre): one multiply
        if (has_pre) Vec8<TC>::unpack(pre, y);
        else Vec8<TC>::ld((const TC*)a.dact_aux + off, y);
#pragma unroll
        for (int t = 0; t < 8; ++t) x[t] *= y[t];
    }
    if (EPI == EPI_DGELU || EPI == EPI_DQGELU || EPI == EPI_ANY) {
        if (a.dact_aux) {
            const int dact = EPI == EPI_DGELU ? M3AE_ACT_GELU : (EPI == EPI_DQGELU ? M3AE_ACT_QUICKGELU : a.dact);
            if (has_pre && PRE_IS_AUX) Vec8<TC>::unpack(pre, y);
            else Vec8<TC>::ld((const TC*)a.dact_aux + off, y);
            act_bwd_mul_fast_n<8>(x, y, dact);
        }
    }
    if (a.accumulate) {
        Vec8<TC>::ld((const TC*)a.C + off, y);
#pragma unroll
        for (int t = 0; t < 8; ++t) x[t] += y[t];
    }
#ifdef M3AE_EXP_NT_NOSTORE
#pragma unroll
    for (int t = 0; t < 8; ++t) asm volatile("" ::"v"(x[t]));
#else
    Vec8<TC>::st_policy((TC*)a.C + off, x, a.st_policy);
#endif
}

// Row-contiguous epilogue: the wave's WM x 64 fp32 accumulator tile goes through its private LDS slab (32-row passes,
// 68-float rows: conflict-free ds_write_b128 / ds_read_b128) so that every global access of the epilogue is
// 8 lanes x 16 B = one whole 128-B line per row (the direct fragment layout touches 16 lines per instruction, 32 B
// each, and made the N = 3072 GELU GEMMs store-issue bound).
template <typename TC, int EPI, int MI, int RT = 2>   // RT: 16-row tiles per slab pass (slab = 16 RT rows x 68 floats)
DEVINL void epilogue_rows(const MfmaArgs& a, char* smem, int wave, int lane, int64_t m_base, int64_t n_base,
                          f32x4 (&acc)[MI][4]) {
    constexpr int LDW = 68;
    constexpr bool BF = sizeof(TC) == 2;
    constexpr bool PRE_IS_AUX = (EPI == EPI_DGELU || EPI == EPI_DQGELU || EPI == EPI_DMUL);
    float* t = (float*)smem + wave * (16 * RT) * LDW;
    int64_t ncol = n_base + (lane & 7) * 8;
    ncol = ncol < a.N ? ncol : 0;   // N % 8 == 0 on this path: a lane's 8 columns are all inside or all outside
    // Everything the epilogue reads from global memory is requested up front, from inline asm, and waited for ONCE (gload16_asm):
    // as compiler-visible loads under run-time conditions they made hipcc put s_waitcnt vmcnt(0) in front of the bias add and the
    // residual add of every row pass, i.e. every pass waited for the previous pass's output stores to be acknowledged.
    // Addresses are clamped into the operand: rows / columns past the edge are loaded but never used.
    // (EPI_ANY, the catch-all instantiation, spills registers: a spill between an asm load and its wait would save a register whose
    // load is still in flight -- it keeps compiler-visible loads; m3ae_amd/build.py refuses a build in which any OTHER instantiation
    // of a kernel with asm loads spills.)
    constexpr bool ASM_LOADS = EPI != EPI_ANY;
    u32x4 bias_v[2] = {(u32x4){0u, 0u, 0u, 0u}, (u32x4){0u, 0u, 0u, 0u}};
    const TC* src = (const TC*)(PRE_IS_AUX ? a.dact_aux : a.residual);
    const bool has_pre = BF && src != nullptr;
    u32x4 pre[MI / RT][2 * RT];
#pragma unroll
    for (int half = 0; half < MI / RT; ++half)
#pragma unroll
        for (int pass = 0; pass < 2 * RT; ++pass) pre[half][pass] = (u32x4){0u, 0u, 0u, 0u};
    if (ASM_LOADS) {
        if (a.bias) {
            gload16_asm(bias_v[0], a.bias + ncol);
            gload16_asm(bias_v[1], a.bias + ncol + 4);
        }
        if (has_pre) {
            const int64_t m_last = a.M - 1;
#pragma unroll
            for (int half = 0; half < MI / RT; ++half)
#pragma unroll
                for (int pass = 0; pass < 2 * RT; ++pass) {
                    const int64_t m = m_base + 16 * RT * half + pass * 8 + (lane >> 3);
                    gload16_asm(pre[half][pass], src + (m < m_last ? m : m_last) * a.ldc + ncol);
                }
        }
    } else {
        if (a.bias) {
            bias_v[0] = *(const u32x4*)(a.bias + ncol);
            bias_v[1] = *(const u32x4*)(a.bias + ncol + 4);
        }
        if (has_pre) {
#pragma unroll
            for (int half = 0; half < MI / RT; ++half)
#pragma unroll
                for (int pass = 0; pass < 2 * RT; ++pass) {
                    const int64_t m = m_base + 16 * RT * half + pass * 8 + (lane >> 3);
                    if (m < a.M) pre[half][pass] = *(const u32x4*)(src + m * a.ldc + ncol);
                }
        }
    }
    float bias8[8];
#pragma unroll
    for (int half = 0; half < MI / RT; ++half) {
#pragma unroll
        for (int ii = 0; ii < RT; ++ii)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *(f32x4*)(t + (16 * ii + (lane & 15)) * LDW + 16 * j + 4 * (lane >> 4)) = acc[RT * half + ii][j];
        if (half == 0) {   // the loads' latency runs under the first slab writes
            if (ASM_LOADS) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("" : "+v"(bias_v[0]), "+v"(bias_v[1]));   // "redefined" behind the wait: no reader can be scheduled in front of it
#pragma unroll
                for (int hh = 0; hh < MI / RT; ++hh)
#pragma unroll
                    for (int pass = 0; pass < 2 * RT; ++pass) asm volatile("" : "+v"(pre[hh][pass]));
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {   // (through a copy: __builtin_bit_cast of a vector-element lvalue reads element 0, hipcc 7.2)
                const u32x4 v = bias_v[q >> 2];
                bias8[q] = __uint_as_float(v[q & 3]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int pass = 0; pass < 2 * RT; ++pass) {
            const int row = pass * 8 + (lane >> 3), col = (lane & 7) * 8;
            const f32x4 v0 = *(const f32x4*)(t + row * LDW + col);
            const f32x4 v1 = *(const f32x4*)(t + row * LDW + col + 4);
            float x[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            const int64_t m = m_base + 16 * RT * half + row, n = n_base + col;
            if (m < a.M && n < a.N) epilogue8<TC, EPI>(a, m, n, x, bias8, has_pre, pre[half][pass]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// gemm_nt_pp2.hip: the second-generation ping-pong kernel (preconditions: rows_epi, K % 32 == 0, M, N > 128)
int launch_nt_pp2(const MfmaArgs& a, int epi, bool persistent, hipStream_t s);

}  // namespace m3g
