// MFMA tile plumbing shared by the GEMM kernels (gemm_mfma.hip) and the fused cross-attention kernels (xattn.hip):
// LDS-DMA staging of operand tiles, swizzled LDS images, fragment reads, XCD-aware workgroup order.
#pragma once
#include "common.h"

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_cvoid;

// 256 B of zeros for out-of-range reduction rows of the transposing (T-form) stages; one copy per translation unit
static __device__ __attribute__((aligned(256))) uint32_t g_m3ae_zero_page[64];

#define PP_FENCE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

// Bijective XCD remap (workgroups are dealt round-robin over the 8 XCDs; give each XCD a contiguous tile range).
DEVINL unsigned xcd_remap(unsigned bid, unsigned nwg) {
    const unsigned q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const unsigned base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

DEVINL void glds16(const void* src, char* lds_dst_uniform) {
    __builtin_amdgcn_global_load_lds((gbl_cvoid*)src, (lds_void*)lds_dst_uniform, 16, 0, 0);
}
// The same piece from inline asm (M0 saved / restored in the statement).  hipcc tracks the BUILTIN as a pending LDS store and puts
// `s_waitcnt vmcnt(0)` in front of every later LDS access it cannot prove disjoint -- in front of the first ds_write of an epilogue
// (gemm_nt_pp2.hip) and, found in round 4, in front of EVERY ds_read_b64_tr_b16 fragment read of the wgrad (TN) kernels: each phase of
// their main loops drained the whole ring (the .s of round 3 shows it), i.e. the counted-vmcnt prefetch never overlapped anything.
// The asm form is invisible to that bookkeeping: every wait for these pieces is the kernel's own counted s_waitcnt + barrier.
DEVINL void glds16_raw(const void* src, char* lds_dst_uniform) {
    unsigned keep;
    const unsigned dst = (unsigned)(uintptr_t)(lds_void*)lds_dst_uniform;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}

// LDS image of an operand tile: [rows][BKT k] bf16, 16-B chunk c of row r stored at chunk position c ^ swz(r), chosen
// so that every ds_read_b128 lane group ({0-3,12-15,20-27}, ...: 8 rows x chunk c + 8 rows x chunk c^1) covers all 16
// slots of the 256-B bank row:
//   BKT = 64 (128-B rows, 2 rows per bank row): swz = (r >> 1) & 7
//   BKT = 32 ( 64-B rows, 4 rows per bank row): swz = (4 - ((r >> 2) & 3)) & 3
template <int BKT> DEVINL int nt_swz(int row) {
    return BKT == 64 ? ((row >> 1) & 7) : ((4 - ((row >> 2) & 3)) & 3);
}

// one 1-KiB LDS-DMA piece per wave-instruction = 1024 / (2 * BKT) rows
template <int BKT, int SEGS_PER_WAVE, int NWAVES>
DEVINL void nt_stage(const bf16_t* G, int64_t ld, int64_t row0, int64_t nrows, int64_t k0, char* tile, int wave,
                     int lane) {
    constexpr int CPR = BKT / 8;         // 16-B chunks per row
    constexpr int RPS = 64 / CPR;        // rows per piece
#pragma unroll
    for (int q = 0; q < SEGS_PER_WAVE; ++q) {
        const int seg = q * NWAVES + wave;
        const int row = seg * RPS + lane / CPR;
        const int chunk = (lane % CPR) ^ nt_swz<BKT>(row);
        int64_t grow = row0 + row;
        grow = grow < nrows ? grow : nrows - 1;  // clamp: duplicated rows are computed but never stored
        glds16(G + grow * ld + k0 + chunk * 8, tile + seg * 1024);
    }
}

template <int BKT> DEVINL s16x8 nt_frag(const char* tile, int row, int chunk) {
    return *(const s16x8*)(tile + row * (BKT * 2) + ((chunk ^ nt_swz<BKT>(row)) << 4));
}

// LDS image of an operand tile: [64 reduction rows][128 n] bf16, 256-B rows, 16-B chunk c of row r stored at
// chunk position c ^ tn_swz(r).  A ds_read_b64_tr_b16 32-lane half reads 8 rows (r & 3 = 0..3, (r >> 3) & 1 =
// 0,1) x 32 B; the swizzle sends those 8 rows to 8 different 32-B column pairs -> all 64 banks, conflict free.
DEVINL int tn_swz(int row) { return (((row & 3) << 1) | ((row >> 3) & 1)) << 1; }

// stage a [64 reduction rows][COLS] bf16 operand tile: 1-KiB DMA pieces = 1024 / (2 * COLS) rows each
template <int COLS, int SEGS_PER_WAVE, int NWAVES>
DEVINL void tn_stage(const bf16_t* G, int64_t ld, int64_t r0, int64_t r_end, int64_t n0, char* tile, int wave,
                     int lane) {
    constexpr int CPR = COLS / 8;   // 16-B chunks per row (16 or 32)
    constexpr int RPS = 64 / CPR;   // rows per piece (4 or 2)
#pragma unroll
    for (int q = 0; q < SEGS_PER_WAVE; ++q) {
        const int seg = q * NWAVES + wave;
        const int row = seg * RPS + lane / CPR;
        const int chunk = (lane % CPR) ^ tn_swz(row);  // XOR < 16: stays inside the row's aligned 256-B half
        const int64_t grow = r0 + row;
        const void* src = (grow < r_end) ? (const void*)(G + grow * ld + n0 + chunk * 8)
                                         : (const void*)((const char*)g_m3ae_zero_page + (lane & 15) * 16);
        glds16_raw(src, tile + seg * 1024);
    }
}

// MFMA 16x16x32 operand whose k index runs over tile ROWS kbase..kbase+31 and whose row/col index is tile
// column ncol0 + (lane & 15): two transposing reads of 4 rows x 16 columns each.
template <int COLS>
DEVINL s16x8 tn_frag(const char* tile, int kbase, int ncol0, int lane) {
    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
    const int col = ncol0 + 4 * p;
    const int chunk = col >> 3, within = (col & 7) * 2;
    const int r1 = kbase + 8 * g + q, r2 = r1 + 4;
    const char* a1 = tile + r1 * (COLS * 2) + ((chunk ^ tn_swz(r1)) << 4) + within;
    const char* a2 = tile + r2 * (COLS * 2) + ((chunk ^ tn_swz(r2)) << 4) + within;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a1);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a2);
    return (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}


// The same fragments by precomputed per-lane byte offsets (loop-invariant; hoisted by hand for the kernels whose slot base
// changes every iteration): K-contiguous image ...
template <int BKT> DEVINL int nt_frag_off(int row, int chunk) { return row * (BKT * 2) + ((chunk ^ nt_swz<BKT>(row)) << 4); }
DEVINL s16x8 nt_frag_at(const char* tile, int off) { return *(const s16x8*)(tile + off); }
// ... and the reduction-strided image ([rows][COLS], tn_frag with kbase = 0): two offsets per fragment
template <int COLS> DEVINL void tn_frag_offs(int ncol0, int lane, int& o1, int& o2) {
    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
    const int col = ncol0 + 4 * p;
    const int chunk = col >> 3, within = (col & 7) * 2;
    const int r1 = 8 * g + q, r2 = r1 + 4;
    o1 = r1 * (COLS * 2) + ((chunk ^ tn_swz(r1)) << 4) + within;
    o2 = r2 * (COLS * 2) + ((chunk ^ tn_swz(r2)) << 4) + within;
}
DEVINL s16x8 tn_frag_at(const char* tile, int o1, int o2) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + o1));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + o2));
    return (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
