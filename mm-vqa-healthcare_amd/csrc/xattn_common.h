// Helpers shared by the fused cross-attention kernels (xattn.hip: the per-sample batched GEMM template; xflash.hip: the
// two-GEMM image-query kernel): memory-row maps, LDS-DMA staging with row maps, counted waits, lane-quad reductions.
#pragma once
#include "mfma_tiles.h"

// ---- xflash.hip: image queries over text keys (dir 1) as ONE launch: S = x K'^T, the per-head softmax in registers, the
// (dropped) probabilities kept in LDS as the A operand of the second product, s = drop(drop(P) V' + b_o) + x.
struct XfArgs {
    const bf16_t* X;                   // [B][I][D] image-side hidden states (queries and residual), row stride D
    const bf16_t* Kp; const bf16_t* Vp;   // [B][R][D] each: K' rows n = h*T + j (reduction-contiguous), V' rows = reduction index
    const float* colbias;              // [B][R]: b_q,h . k_h[j] / sqrt(dh) + mask[j]
    const float* bo;                   // [D]
    bf16_t* S;                         // [B*I][D] out: the pre-LayerNorm sum
    bf16_t* P; bf16_t* Pd;             // [B][I][R] saves for the backward (nullptr: forward only; Pd only with dropout)
    int B, I, D, H, tiles_m;
    DropState drop_a, drop_h; int has_drop;
    int drop_ld;                       // drop_ld(T): attention-dropout index ((b*H + h)*I + q) * drop_ld + k
};
// T = 32 (R = 384) or 64 (R = 768); D % 256 == 0.  0 = launched, else hipError_t / M3AE_ERR_*
int m3ae_xflash_dir1(const XfArgs& a, int T, hipStream_t s);

namespace {

DEVINL int64_t map_row(int64_t r, int div, int mul) { return div ? (r / div) * (int64_t)mul + r % div : r; }

// nt_stage (mfma_tiles.h) with a memory-row map
template <int SEGS_PER_WAVE, int NWAVES>
DEVINL void nt_stage_m(const bf16_t* G, int64_t ld, int64_t row0, int64_t nrows, int64_t k0, char* tile, int wave, int lane,
                       int div, int mul) {
#pragma unroll
    for (int q = 0; q < SEGS_PER_WAVE; ++q) {
        const int seg = q * NWAVES + wave;
        const int row = seg * 16 + (lane >> 2);
        const int chunk = (lane & 3) ^ nt_swz<32>(row);
        int64_t grow = row0 + row;
        grow = grow < nrows ? grow : nrows - 1;
        glds16(G + map_row(grow, div, mul) * ld + k0 + chunk * 8, tile + seg * 1024);
    }
}

// one 1-KiB piece (4 rows) of a [32 k-rows][128 cols] panel of a reduction-strided operand; piece = 0..7
DEVINL void t_stage128(const bf16_t* G, int64_t ld, int r0, int r_end, int col0, int ncols, char* panel, int piece,
                       int lane, int div = 0, int mul = 0) {
    const int wave = piece;
    const int row = wave * 4 + (lane >> 4);
    const int chunk = (lane & 15) ^ tn_swz(row);
    const int grow = r0 + row, col = col0 + chunk * 8;
    const int64_t mrow = div ? ((int64_t)(grow / div) * mul + grow % div) : (int64_t)grow;
    const void* src = (grow < r_end && col < ncols) ? (const void*)(G + mrow * ld + col)
                                                    : (const void*)((const char*)g_m3ae_zero_page + (lane & 15) * 16);
    glds16(src, panel + wave * 1024);
}

template <int N> DEVINL void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int G> DEVINL void wait_vm_chunks(int chunks) {   // chunks in {0, 1, 2}
    if (chunks >= 2) wait_vm<2 * G>();
    else if (chunks == 1) wait_vm<G>();
    else wait_vm<0>();
}

DEVINL float quad16_max(float v) {   // over the 4 lanes l, l^16, l^32, l^48
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
DEVINL float quad16_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}
DEVINL void st_bf4(bf16_t* p, const float* x) { *(u32x2*)p = (u32x2){pack2bf(x[0], x[1]), pack2bf(x[2], x[3])}; }
DEVINL void ld_bf4(const bf16_t* p, float* x) {
    const u32x2 v = *(const u32x2*)p;
    x[0] = __uint_as_float(v[0] << 16); x[1] = __uint_as_float(v[0] & 0xffff0000u);
    x[2] = __uint_as_float(v[1] << 16); x[3] = __uint_as_float(v[1] & 0xffff0000u);
}

constexpr float LOG2E = 1.4426950408889634f;

}  // namespace
