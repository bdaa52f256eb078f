// Scaled-dot-product attention for the M3AE hot path (self 577x577, text 32x32, cross 32x577 / 577x32; Dh = 64).
//
// bf16 ("perf mode"): flash-style, nothing of size Lq x Lk ever reaches HBM.
//   One wave owns one 32-row tile and streams 32-row tiles of the other side.  Products use
//   v_mfma_f32_32x32x16_bf16 with the streamed index on the accumulator ROWS ("swapped" S^T = K.Q^T), so that
//     * the softmax reduction over keys is in-lane (16 registers) + one exchange with lane^32,
//     * the bf16-converted accumulator is directly the B operand of the next product (O^T = V^T.P^T): no LDS
//       round trip for P (cdna_hip_programming.md 3 "An accumulator tile as the next MFMA's operand"),
//     * the only operand that needs a transpose (V^T, K^T, Q^T, dO^T fragments: 4 consecutive rows per lane) is
//       read from a small per-wave LDS tile with the transposing read ds_read_b64_tr_b16.
//   Row-major fragments (8 consecutive d per lane) come straight from global/L2 as 16-byte loads.
//   Waves never synchronise with each other (per-wave LDS regions, no barriers).
//   Backward is two kernels in the same mould (dQ: wave = 32 queries; dK/dV: wave = 32 keys), recomputing P from
//   the saved log-sum-exp: deterministic, no atomics.
// fp32 ("parity mode"): the reference's own algorithm (bert_model.py:301-340): S = QK^T/sqrt(dh) + mask
//   materialised in a caller workspace, row softmax, PV, through the generic GEMM kernel.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

int m3ae_gemm_generic(const m3ae_gemm_desc& d, hipStream_t s);

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr int VRS = 192;  // LDS row stride (bytes) of a [32][64] bf16 tile: 4 consecutive rows hit 4 disjoint bank quarters
constexpr int TILE_LDS = 32 * VRS;

struct AttnArgs {
    const bf16_t* q; int64_t q_sb, q_sl;
    const bf16_t* k; int64_t k_sb, k_sl;
    const bf16_t* v; int64_t v_sb, v_sl;
    bf16_t* o; int64_t o_sb, o_sl;
    const float* key_mask;
    const float* pos_bias;
    float scale, scale_log2;
    int causal;
    float* lse; int64_t lse_stride;
    int64_t B, H, Lq, Lk;
    const bf16_t* d_o;
    bf16_t* dq; bf16_t* dk; bf16_t* dv;
    float* delta;
    float* d_pos_bias;
    DropState drop;  // attention-probability dropout (bert_model.py:334); mask index ((b H + h) Lq + q) * ld(Lk) + k
    int has_drop;
};
DEVINL int64_t drop_ldk(int64_t Lk) { return (Lk + 3) & ~(int64_t)3; }

DEVINL f32x16 mfma32(s16x8 a, s16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c,
                                                   0, 0, 0);
}

// accumulator register `reg` of lane half `h` holds ROW crow(reg, h) of a 32x32 tile (column = lane & 31)
DEVINL int crow(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// value of lane J of the caller's quad (lanes 4 m .. 4 m + 3), by DPP: one VALU move, no LDS
template <int J> DEVINL uint32_t quad_bcast(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, J | (J << 2) | (J << 4) | (J << 6), 0xf, 0xf, true);
}

// bf16 B-operand fragment for k-step s from accumulator registers 8s..8s+7 (k order = crow order)
DEVINL s16x8 pack_acc(const f32x16& p, int s) {
    u32x4 w;
    w[0] = pack2bf(p[8 * s + 0], p[8 * s + 1]);
    w[1] = pack2bf(p[8 * s + 2], p[8 * s + 3]);
    w[2] = pack2bf(p[8 * s + 4], p[8 * s + 5]);
    w[3] = pack2bf(p[8 * s + 6], p[8 * s + 7]);
    return __builtin_bit_cast(s16x8, w);
}

// 16-byte row fragment: row-major [row][64] bf16 operand, element j <-> d = 16 * ks + 8 * h + j
DEVINL s16x8 row_frag(const bf16_t* rowp, int ks, int h) { return *(const s16x8*)(rowp + 16 * ks + 8 * h); }

// Stage a [32 rows][64 d] tile (rows row0.., clamped to nrows-1) into this wave's LDS tile: 4 x 16 B per lane.
struct Stage4 { s16x8 v[4]; };
DEVINL Stage4 tile_load(const bf16_t* base, int64_t sl, int64_t row0, int64_t nrows, int lane) {
    Stage4 s;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = i * 64 + lane;
        int64_t row = row0 + (c >> 3);
        row = row < nrows ? row : nrows - 1;
        s.v[i] = *(const s16x8*)(base + row * sl + (c & 7) * 8);
    }
    return s;
}
// Fast forms for FULL tiles: wave-uniform tile base (SGPR pair) + per-lane 32-bit element offsets computed once,
// so the streaming loads cost no per-iteration vector address arithmetic.
struct TileOffs { int t[4]; int rf; };
DEVINL TileOffs make_offs(int lane, int64_t sl) {
    TileOffs o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = i * 64 + lane;
        o.t[i] = (c >> 3) * (int)sl + (c & 7) * 8;
    }
    o.rf = (lane & 31) * (int)sl + 8 * (lane >> 5);
    return o;
}
DEVINL Stage4 tile_load_u(const bf16_t* tbase, const TileOffs& o) {
    Stage4 s;
#pragma unroll
    for (int i = 0; i < 4; ++i) s.v[i] = *(const s16x8*)(tbase + o.t[i]);
    return s;
}
DEVINL s16x8 row_frag_u(const bf16_t* tbase, const TileOffs& o, int ks) { return *(const s16x8*)(tbase + o.rf + 16 * ks); }

DEVINL void tile_store(char* tile, const Stage4& s, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = i * 64 + lane;
        *(s16x8*)(tile + (c >> 3) * VRS + (c & 7) * 16) = s.v[i];
    }
}

// Row-major A/B-operand fragment of tile row r from the same LDS image (16-byte read)
DEVINL s16x8 lds_row_frag(const char* tile, int r, int ks, int h) {
    return *(const s16x8*)(tile + r * VRS + 32 * ks + 16 * h);
}

// Transposed A-operand fragment from a [32 rows][64 d] LDS tile: MFMA row index = d = 32 * dt + (lane & 31),
// element j <-> tile row 16 * s + 8 * (j >> 2) + 4 * h + (j & 3)   (= crow order of pack_acc).
DEVINL s16x8 tr_frag(const char* tile, int dt, int s, int lane) {
    const int h = lane >> 5, gg = (lane >> 4) & 1, i16 = lane & 15, qq = i16 >> 2, p = i16 & 3;
    const int row = 16 * s + 4 * h + qq;
    const int dcol = 32 * dt + 16 * gg + 4 * p;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + row * VRS + dcol * 2));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + (row + 8) * VRS + dcol * 2));
    return (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// Store a transposed accumulator pair (O^T / dQ^T / dK^T / dV^T: rows = d, column = lane's token) as [token][64].
DEVINL void store_rows(bf16_t* rowp, const f32x16& t0, const f32x16& t1, float mul, int h) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        const f32x16& t = dt ? t1 : t0;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            u32x2 w;
            w[0] = pack2bf(t[4 * g4 + 0] * mul, t[4 * g4 + 1] * mul);
            w[1] = pack2bf(t[4 * g4 + 2] * mul, t[4 * g4 + 3] * mul);
            *(u32x2*)(rowp + 32 * dt + 8 * g4 + 4 * h) = w;
        }
    }
}

// raw v_exp_f32 (2^x): arguments here are <= 0 up to rounding, denormal results may flush -- no range fix-up code
DEVINL float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

DEVINL f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// ---------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------
// Flags are template parameters so that the hot instance (image keys: no mask, no bias, not causal) has a
// branch-free loop body; only the LAST key tile pays for bounds handling.
struct Flags { bool mask, bias, causal; };

template <bool MASK, bool BIAS, bool CAUSAL, bool LAST, bool DROP = false>
DEVINL void score_to_prob(f32x16& s, float& m, float& l, f32x16& o0, f32x16& o1, const AttnArgs& a, const float* mrow,
                          const float* brow, int64_t key0, int64_t qi, int h, uint64_t drop_row = 0) {
    // log2-domain scores, additive mask / bias, bounds; online softmax update of (m, l, o)
    constexpr bool PLAIN = !MASK && !BIAS && !CAUSAL && !LAST;  // hot path: scale folded into the exp argument
    float tmax = -1e30f;
    if (PLAIN) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) tmax = fmaxf(tmax, s[reg]);
        tmax *= a.scale_log2;
    } else {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int64_t key = key0 + crow(reg, h);
            float x = s[reg] * a.scale_log2;
            if (MASK || BIAS) {
                const int64_t kc = (LAST && key >= a.Lk) ? a.Lk - 1 : key;
                if (MASK) x = fmaf(mrow[kc], LOG2E, x);
                if (BIAS) x = fmaf(brow[kc], LOG2E, x);
            }
            if (LAST) x = key < a.Lk ? x : -INFINITY;
            if (CAUSAL) x = key > qi ? -INFINITY : x;
            s[reg] = x;
            tmax = fmaxf(tmax, x);
        }
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    if (__any(tmax > m)) {  // wave-uniform: the running max rarely moves after the first tiles
        const float mnew = fmaxf(m, tmax);
        const float alpha = fast_exp2(m - mnew);
        m = mnew;
        l *= alpha;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) { o0[reg] *= alpha; o1[reg] *= alpha; }
    }
    float lsum = 0.f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const float p = PLAIN ? fast_exp2(fmaf(s[reg], a.scale_log2, -m)) : fast_exp2(s[reg] - m);
        s[reg] = p;
        lsum += p;
    }
    l += lsum;
    if (DROP) {  // the normaliser keeps every key; only the P that multiplies V is dropped (and rescaled)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {  // registers 4 r4 .. 4 r4 + 3 = keys key0 + 8 r4 + 4 h + (0..3): one hash
            float x4[4] = {s[4 * r4], s[4 * r4 + 1], s[4 * r4 + 2], s[4 * r4 + 3]};
            drop_apply4(a.drop, drop_row + (uint64_t)(key0 + 8 * r4 + 4 * h), x4);
#pragma unroll
            for (int t = 0; t < 4; ++t) s[4 * r4 + t] = x4[t];
        }
    }
}

// =========================================================================================================
// Workgroup-cooperative variants ("coop"): the 4 waves of a workgroup own consecutive row tiles of ONE (batch, head)
// and share every streamed tile through LDS: one coalesced 16-B-per-lane global load per tile for the whole
// workgroup (8 lanes = one 128-B row), instead of each wave fetching its own copy with 32-B-per-line fragment
// loads.  L2 -> CU traffic drops 4x and is fully coalesced (the per-wave kernels ran at the ~8 TB/s L2 fabric
// ceiling).  Two LDS images per streamed operand where both kinds of MFMA fragment are needed:
//   row image  [32][128 B], chunk ^= (row >> 1) & 7  -> conflict-free ds_read_b128 row fragments
//   tr  image  [32][192 B]                           -> conflict-free ds_read_b64_tr_b16 transposed fragments
// Double-buffered LDS images, ONE raw s_barrier per streamed tile; the global fetch runs TWO tiles ahead in registers
// (tile t + 2 is requested before tile t is computed and written to LDS at the end of tile t + 1), so a tile's
// L2 / HBM latency has a whole tile of MFMA + softmax work to hide behind.  __syncthreads() is not used in the loops:
// its implicit vmcnt(0) would drain that prefetch at every tile.
// =========================================================================================================
constexpr int RIMG = 32 * 128;  // row image bytes

DEVINL int rswz(int row) { return (row >> 1) & 7; }
// cooperative tile fetch: thread t of 256 -> row t >> 3, 16-B chunk t & 7 of a [32][64] bf16 tile
DEVINL s16x8 coop_load(const bf16_t* base, int64_t sl, int64_t row0, int64_t nrows, int t) {
    int64_t row = row0 + (t >> 3);
    row = row < nrows ? row : nrows - 1;
    return *(const s16x8*)(base + row * sl + (t & 7) * 8);
}
DEVINL void coop_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this lane's LDS image writes (and fragment reads) retired
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
DEVINL void put_row_img(char* img, s16x8 v, int t) {
    const int row = t >> 3, ch = t & 7;
    *(s16x8*)(img + row * 128 + ((ch ^ rswz(row)) << 4)) = v;
}
DEVINL void put_tr_img(char* img, s16x8 v, int t) { *(s16x8*)(img + (t >> 3) * VRS + (t & 7) * 16) = v; }
DEVINL s16x8 get_row_frag(const char* img, int r, int ks, int h) {
    return *(const s16x8*)(img + r * 128 + (((2 * ks + h) ^ rswz(r)) << 4));
}

// Workgroups are dealt round-robin over the 8 XCDs (each with its own L2).  Remap the linear block id so that every XCD
// walks a contiguous range of (b, head, block) triples: all the query (key) blocks of one (b, head) then run on ONE XCD
// close in time and share its K / V (Q / dO) in that L2, instead of re-fetching them from the fabric once per block
// (PMC, profiles/r01_pmc_traffic.txt: 2.3x the algorithmic read bytes before this remap).
struct AttnBlock { int bx, head; int64_t b; };
DEVINL AttnBlock attn_block() {
    const unsigned gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
    const unsigned lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned q = total >> 3, r = total & 7, xcd = lin & 7;
    const unsigned id = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
    AttnBlock o;
    o.bx = (int)(id % gx);
    o.head = (int)((id / gx) % gy);
    o.b = (int64_t)(id / (gx * gy));
    return o;
}

// DROP (attention-probability dropout) is a template parameter: a run-time test per score element would split the
// unrolled softmax into 16 basic blocks and fence the MFMA / VALU interleave for the eval-mode instances too.
// (the dropout instance at four waves per SIMD -- 128 registers, 7 spilled -- measured 711 vs 648 us at 577 x 577, B = 256: it stays at
// the 150 registers / three waves hipcc gives it, round 4)
template <int NQ, bool MASK, bool BIAS, bool CAUSAL, bool DROP>
__global__ __launch_bounds__(256, (NQ == 1 && !MASK && !BIAS && !CAUSAL && !DROP) ? 4 : 2) void attn_fwd_coop_kernel(AttnArgs a) {
    if (DROP) drop_resolve(a.drop);
    constexpr int BUF = RIMG + TILE_LDS;  // K row image + V tr image
    __shared__ __attribute__((aligned(16))) char lds[2 * BUF];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = lane & 31, h = lane >> 5;
    const AttnBlock blk = attn_block();
    const int head = blk.head;
    const int64_t b = blk.b;
    const int64_t q0 = ((int64_t)blk.bx * 4 + wave) * (32 * NQ);
    const bool active = q0 < a.Lq;  // wave-uniform; inactive waves still help with the cooperative loads

    int64_t qi[NQ];
    s16x8 qf[NQ][4];
    const float* brow[NQ];
    f32x16 o[NQ][2];
    float m[NQ], l[NQ];
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
        qi[n] = q0 + 32 * n + r;
        const int64_t qrow = qi[n] < a.Lq ? qi[n] : a.Lq - 1;
        const bf16_t* qp = a.q + b * a.q_sb + qrow * a.q_sl + head * 64;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[n][ks] = row_frag(qp, ks, h);
        brow[n] = BIAS ? a.pos_bias + ((int64_t)head * a.Lq + qrow) * a.Lk : nullptr;
        o[n][0] = zero16(); o[n][1] = zero16();
        m[n] = -1e30f; l[n] = 0.f;
    }
    const bf16_t* kbase = a.k + b * a.k_sb + head * 64;
    const bf16_t* vbase = a.v + b * a.v_sb + head * 64;
    const float* mrow = MASK ? a.key_mask + b * a.Lk : nullptr;
    const int nkt = (int)((a.Lk + 31) / 32);

    s16x8 kreg = coop_load(kbase, a.k_sl, 0, a.Lk, t);
    s16x8 vreg = coop_load(vbase, a.v_sl, 0, a.Lk, t);
    put_row_img(lds, kreg, t);
    put_tr_img(lds + RIMG, vreg, t);
    if (nkt > 1) {
        kreg = coop_load(kbase, a.k_sl, 32, a.Lk, t);
        vreg = coop_load(vbase, a.v_sl, 32, a.Lk, t);
    }
    coop_barrier();
    for (int kt = 0; kt < nkt; ++kt) {
        const bool last = kt + 1 == nkt;
        s16x8 kreg2 = kreg, vreg2 = vreg;
        if (kt + 2 < nkt) {
            kreg2 = coop_load(kbase, a.k_sl, (int64_t)(kt + 2) * 32, a.Lk, t);
            vreg2 = coop_load(vbase, a.v_sl, (int64_t)(kt + 2) * 32, a.Lk, t);
        }
        const char* kimg = lds + (kt & 1) * BUF;
        const char* vimg = kimg + RIMG;
        if (active) {
            s16x8 kf[4], vf[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) { kf[ks] = get_row_frag(kimg, r, ks, h); vf[ks] = get_row_frag(vimg, r, ks, h); }
            f32x16 s[NQ];
#pragma unroll
            for (int n = 0; n < NQ; ++n) {
                s[n] = zero16();
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) s[n] = mfma32(kf[ks], qf[n][ks], s[n]);  // S^T[key][q] = K . Q^T
            }
            const int64_t key0 = (int64_t)kt * 32;
#pragma unroll
            for (int n = 0; n < NQ; ++n) {
                const uint64_t drow = (uint64_t)(((b * a.H + head) * a.Lq + qi[n]) * drop_ldk(a.Lk));
                if (last) score_to_prob<MASK, BIAS, CAUSAL, true, DROP>(s[n], m[n], l[n], o[n][0], o[n][1], a, mrow, brow[n], key0, qi[n], h, drow);
                else score_to_prob<MASK, BIAS, CAUSAL, false, DROP>(s[n], m[n], l[n], o[n][0], o[n][1], a, mrow, brow[n], key0, qi[n], h, drow);
            }
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                const s16x8 v0 = tr_frag(vimg, 0, ss, lane), v1 = tr_frag(vimg, 1, ss, lane);
#pragma unroll
                for (int n = 0; n < NQ; ++n) {
                    const s16x8 pb = pack_acc(s[n], ss);
                    o[n][0] = mfma32(v0, pb, o[n][0]);  // O^T[d][q] += V^T[d][key] . P^T[key][q]
                    o[n][1] = mfma32(v1, pb, o[n][1]);
                }
            }
        }
        if (!last) {
            char* nb = lds + ((kt + 1) & 1) * BUF;
            put_row_img(nb, kreg, t);
            put_tr_img(nb + RIMG, vreg, t);
        }
        kreg = kreg2; vreg = vreg2;
        coop_barrier();
    }
    if (!active) return;
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
        const float ltot = l[n] + __shfl_xor(l[n], 32, 64);
        if (qi[n] < a.Lq) {
            store_rows(a.o + b * a.o_sb + qi[n] * a.o_sl + head * 64, o[n][0], o[n][1], 1.0f / ltot, h);
            if (h == 0) a.lse[(b * a.H + head) * a.lse_stride + qi[n]] = m[n] + log2f(ltot);
        } else if (h == 0 && qi[n] < a.lse_stride) {
            // padding rows of the log-sum-exp table (Lq .. lse_stride - 1, inside the last active wave's 32 rows): the
            // backward tiles read them (p = exp2(x - lse) of a clamped row, masked to 0 afterwards) -- they must be finite,
            // and the caller no longer zero-fills the table
            a.lse[(b * a.H + head) * a.lse_stride + qi[n]] = 0.f;
        }
    }
}

template <bool MASK, bool BIAS, bool CAUSAL, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_coop_kernel(AttnArgs a) {
    if (DROP) drop_resolve(a.drop);
    constexpr int BUF = RIMG + TILE_LDS + RIMG;  // K row image, K tr image, V row image
    __shared__ __attribute__((aligned(16))) char lds[2 * BUF];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = lane & 31, h = lane >> 5;
    const AttnBlock blk = attn_block();
    const int head = blk.head;
    const int64_t b = blk.b;
    const int64_t q0 = ((int64_t)blk.bx * 4 + wave) * 32;
    const bool active = q0 < a.Lq;

    const int64_t qi = q0 + r;
    const int64_t qrow = qi < a.Lq ? qi : a.Lq - 1;
    const bf16_t* qp = a.q + b * a.q_sb + qrow * a.q_sl + head * 64;
    const bf16_t* dop = a.d_o + b * a.o_sb + qrow * a.o_sl + head * 64;
    s16x8 qf[4], dof[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { qf[ks] = row_frag(qp, ks, h); dof[ks] = row_frag(dop, ks, h); }
    const float lse = a.lse[(b * a.H + head) * a.lse_stride + qrow];
    // delta[q] = sum_d dO[q][d] O[q][d]: this lane already holds half of its row of dO (the MFMA fragments); the matching
    // half of O is fetched once, the two halves meet through a lane swap, and the value is published for the dK/dV kernel
    // (which runs after this one) -- the stand-alone delta pass re-read dO and O from HBM for every attention call
    float dlt = 0.f;
    {
        const bf16_t* op = a.o + b * a.o_sb + qrow * a.o_sl + head * 64;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const s16x8 of = row_frag(op, ks, h);
#pragma unroll
            for (int j = 0; j < 8; ++j) dlt = fmaf(bf2f((bf16_t)of[j]), bf2f((bf16_t)dof[ks][j]), dlt);
        }
        dlt += __shfl_xor(dlt, 32, 64);
        if (h == 0 && active && qi < a.lse_stride)   // padding rows (>= Lq): 0, the caller does not zero-fill the table
            a.delta[(b * a.H + head) * a.lse_stride + qi] = qi < a.Lq ? dlt : 0.f;
    }

    const float s_init = -lse / a.scale_log2;                              // accumulator start of S': exp2(c S') = exp2(c S - lse)
    const float keep_p = DROP ? 1.0f / a.drop.inv_keep : 1.0f;             // 1 - p
    const float dp_init = -dlt * keep_p, dl_drop = -dlt * keep_p;          // (dropped element: dl_drop * inv_keep = -delta)
    const bf16_t* kbase = a.k + b * a.k_sb + head * 64;
    const bf16_t* vbase = a.v + b * a.v_sb + head * 64;
    const float* mrow = MASK ? a.key_mask + b * a.Lk : nullptr;
    const float* brow = BIAS ? a.pos_bias + ((int64_t)head * a.Lq + qrow) * a.Lk : nullptr;
    float* dbrow = (BIAS && a.d_pos_bias) ? a.d_pos_bias + ((int64_t)head * a.Lq + qrow) * a.Lk : nullptr;

    f32x16 g0 = zero16(), g1 = zero16();
    const int nkt = (int)((a.Lk + 31) / 32);
    const uint64_t drow = (uint64_t)(((b * a.H + head) * a.Lq + qi) * drop_ldk(a.Lk));
    s16x8 kreg = coop_load(kbase, a.k_sl, 0, a.Lk, t);
    s16x8 vreg = coop_load(vbase, a.v_sl, 0, a.Lk, t);
    put_row_img(lds, kreg, t);
    put_tr_img(lds + RIMG, kreg, t);
    put_row_img(lds + RIMG + TILE_LDS, vreg, t);
    // global prefetch distance: two tiles.  (Round 4 measured ONE tile at three waves per SIMD -- 168 registers with one prefetch set
    // less and 4-8 spilled registers: dQ 583 -> 730-745 us, profiles/r04_attn_bwd_experiments.log; PF2 = false is that form.)
    constexpr bool PF2 = true;
    if (PF2 && nkt > 1) {
        kreg = coop_load(kbase, a.k_sl, 32, a.Lk, t);
        vreg = coop_load(vbase, a.v_sl, 32, a.Lk, t);
    }
    coop_barrier();
    for (int kt = 0; kt < nkt; ++kt) {
        const bool last = kt + 1 == nkt;
        s16x8 kreg2 = kreg, vreg2 = vreg;
        if (PF2) {
            if (kt + 2 < nkt) {
                kreg2 = coop_load(kbase, a.k_sl, (int64_t)(kt + 2) * 32, a.Lk, t);
                vreg2 = coop_load(vbase, a.v_sl, (int64_t)(kt + 2) * 32, a.Lk, t);
            }
        } else if (!last) {   // tile kt + 1: requested here, written to LDS behind this tile's MFMAs
            kreg = coop_load(kbase, a.k_sl, (int64_t)(kt + 1) * 32, a.Lk, t);
            vreg = coop_load(vbase, a.v_sl, (int64_t)(kt + 1) * 32, a.Lk, t);
        }
        const char* kimg = lds + (kt & 1) * BUF;
        const char* ktr = kimg + RIMG;
        const char* vimg = ktr + TILE_LDS;
        if (active) {
            s16x8 kf[4], vf[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) { kf[ks] = get_row_frag(kimg, r, ks, h); vf[ks] = get_row_frag(vimg, r, ks, h); }
            // Row constants as the initial accumulators (round 4): S' = S - lse / c and dP' = dP - delta leave the MFMA chains, so
            // p = exp2(c S') and dS = p dP' cost a multiply, an exponential and a multiply per element -- the subtractions, the
            // bounds select (last key tile only) and the scale FMA of the round-3 loop are gone (the loop was VALU-bound: 16
            // elements per lane and tile against 12 MFMAs).  Under dropout dP' = dP - delta (1 - p), so that keep: dP' / (1 - p) =
            // dP / (1 - p) - delta, dropped: -delta = dl_drop.
            f32x16 s, dp;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) { s[reg] = s_init; dp[reg] = dp_init; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) s = mfma32(kf[ks], qf[ks], s);      // S'^T[key][q]
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) dp = mfma32(vf[ks], dof[ks], dp);   // dP'^T[key][q] = V . dO^T - delta
            uint32_t dh4[4] = {0u, 0u, 0u, 0u};
            if (DROP) {  // registers 4 r4 .. 4 r4 + 3 hold 4 consecutive keys: one hash per group
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) dh4[r4] = drop_hash(a.drop, (drow + (uint64_t)(kt * 32 + 8 * r4 + 4 * h)) >> 2);
            }
            auto elements = [&](auto last_c) {   // the bounds select only exists in the instantiation of the last key tile
                constexpr bool LAST = decltype(last_c)::value;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int64_t key = (int64_t)kt * 32 + crow(reg, h);
                    float x = s[reg] * a.scale_log2;
                    bool valid = true;
                    if (MASK || BIAS || LAST) {
                        const int64_t kc = (LAST && key >= a.Lk) ? a.Lk - 1 : key;
                        if (MASK) x = fmaf(mrow[kc], LOG2E, x);
                        if (BIAS) x = fmaf(brow[kc], LOG2E, x);
                        if (LAST) valid = key < a.Lk;
                    }
                    if (CAUSAL) valid = valid && key <= qi;
                    float p = fast_exp2(x);
                    if (CAUSAL || LAST) p = valid ? p : 0.f;
                    float t = dp[reg];
                    if (DROP) t = (rotr32(dh4[reg >> 2], 8u * (reg & 3)) >= a.drop.thr ? t : dl_drop) * a.drop.inv_keep;
                    const float ds = p * t;
                    if (BIAS) { if (dbrow && valid && qi < a.Lq) atomicAdd(dbrow + key, ds); }
                    s[reg] = ds;
                }
            };
            if (last) elements(std::true_type{}); else elements(std::false_type{});
            const s16x8 d0 = pack_acc(s, 0), d1 = pack_acc(s, 1);
            // dQ^T[d][q] += K^T[d][key] . dS^T[key][q]
            g0 = mfma32(tr_frag(ktr, 0, 0, lane), d0, g0);
            g1 = mfma32(tr_frag(ktr, 1, 0, lane), d0, g1);
            g0 = mfma32(tr_frag(ktr, 0, 1, lane), d1, g0);
            g1 = mfma32(tr_frag(ktr, 1, 1, lane), d1, g1);
        }
        if (!last) {
            char* nb = lds + ((kt + 1) & 1) * BUF;
            put_row_img(nb, kreg, t);
            put_tr_img(nb + RIMG, kreg, t);
            put_row_img(nb + RIMG + TILE_LDS, vreg, t);
        }
        if (PF2) { kreg = kreg2; vreg = vreg2; }
        coop_barrier();
    }
    if (active && qi < a.Lq) store_rows(a.dq + b * a.q_sb + qi * a.q_sl + head * 64, g0, g1, a.scale, h);
}

template <bool MASK, bool BIAS, bool CAUSAL, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkdv_coop_kernel(AttnArgs a) {
    if (DROP) drop_resolve(a.drop);
    // Q row, Q tr, dO row, dO tr images + the tile's 32 log-sum-exp and 32 delta values (staged with the tile by
    // threads 0..15: as per-lane global loads inside the tile they cost 13 % of the kernel -- timing ablation)
    constexpr int IMG = 2 * (RIMG + TILE_LDS), BUF = IMG + 256;
    __shared__ __attribute__((aligned(16))) char lds[2 * BUF];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = lane & 31, h = lane >> 5;
    const AttnBlock blk = attn_block();
    const int head = blk.head;
    const int64_t b = blk.b;
    const int64_t k0 = ((int64_t)blk.bx * 4 + wave) * 32;
    const bool active = k0 < a.Lk;

    const int64_t ki = k0 + r;
    const int64_t krow = ki < a.Lk ? ki : a.Lk - 1;
    const bf16_t* kp = a.k + b * a.k_sb + krow * a.k_sl + head * 64;
    const bf16_t* vp = a.v + b * a.v_sb + krow * a.v_sl + head * 64;
    s16x8 kfb[4], vfb[4];  // B operands: K^T[d][key], V^T[d][key]
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { kfb[ks] = row_frag(kp, ks, h); vfb[ks] = row_frag(vp, ks, h); }
    const float mk = MASK ? a.key_mask[b * a.Lk + krow] * LOG2E : 0.f;
    const bool key_ok = ki < a.Lk;

    const bf16_t* qbase = a.q + b * a.q_sb + head * 64;
    const bf16_t* dobase = a.d_o + b * a.o_sb + head * 64;
    const float* lrow = a.lse + (b * a.H + head) * a.lse_stride;
    const float* drow = a.delta + (b * a.H + head) * a.lse_stride;
    const float* bcol = BIAS ? a.pos_bias + (int64_t)head * a.Lq * a.Lk + krow : nullptr;

    f32x16 dk0 = zero16(), dk1 = zero16(), dv0 = zero16(), dv1 = zero16();
    const int nqt = (int)((a.Lq + 31) / 32);
    s16x8 qreg = coop_load(qbase, a.q_sl, 0, a.Lq, t);
    s16x8 doreg = coop_load(dobase, a.o_sl, 0, a.Lq, t);
    // threads 0..7: lse[4 t .. 4 t + 3], threads 8..15: delta[...] of the tile (rows padded to lse_stride % 32 == 0)
    const float* ldsrc = (t < 8 ? lrow : drow) + 4 * (t & 7);
    // the table is staged as the accumulator start values: -lse / c (threads 0..7) and -delta (1 - p) (threads 8..15)
    const float ldmul = t < 8 ? -1.0f / a.scale_log2 : (DROP ? -1.0f / a.drop.inv_keep : -1.0f);
    f32x4 ldreg = t < 16 ? *(const f32x4*)ldsrc * ldmul : (f32x4){0.f, 0.f, 0.f, 0.f};
    put_row_img(lds, qreg, t);
    put_tr_img(lds + RIMG, qreg, t);
    put_row_img(lds + RIMG + TILE_LDS, doreg, t);
    put_tr_img(lds + 2 * RIMG + TILE_LDS, doreg, t);
    if (t < 16) *(f32x4*)(lds + IMG + 16 * t) = ldreg;
    if (nqt > 1) {
        qreg = coop_load(qbase, a.q_sl, 32, a.Lq, t);
        doreg = coop_load(dobase, a.o_sl, 32, a.Lq, t);
        if (t < 16) ldreg = *(const f32x4*)(ldsrc + 32) * ldmul;
    }
    coop_barrier();
    for (int qt = 0; qt < nqt; ++qt) {
        const bool last = qt + 1 == nqt;
        s16x8 qreg2 = qreg, doreg2 = doreg;
        f32x4 ldreg2 = ldreg;
        if (qt + 2 < nqt) {
            qreg2 = coop_load(qbase, a.q_sl, (int64_t)(qt + 2) * 32, a.Lq, t);
            doreg2 = coop_load(dobase, a.o_sl, (int64_t)(qt + 2) * 32, a.Lq, t);
            if (t < 16) ldreg2 = *(const f32x4*)(ldsrc + (int64_t)(qt + 2) * 32) * ldmul;
        }
        const char* qimg = lds + (qt & 1) * BUF;
        const char* qtr = qimg + RIMG;
        const char* doimg = qtr + TILE_LDS;
        const char* dotr = doimg + RIMG;
        if (active) {
            // Row constants as the initial accumulators (round 4): the staged table holds -lse / c and -delta (1 - p) per query row
            // (written that way by the staging threads), read straight into the accumulator tuples: S' = S - lse / c, dP' = dP -
            // delta (1 - p); p = exp2(c S' + mask), dS = p dP' (dropout: keep dP' / (1 - p), dropped dl' / (1 - p) = -delta).
            f32x16 s, dp;
            f32x4 dl4[4];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {  // broadcast LDS reads: rows 8 g4 + 4 h .. + 3 of the tile
                const f32x4 l4 = *(const f32x4*)(qimg + IMG + 4 * (8 * g4 + 4 * h));
                dl4[g4] = *(const f32x4*)(qimg + IMG + 128 + 4 * (8 * g4 + 4 * h));
#pragma unroll
                for (int j = 0; j < 4; ++j) { s[4 * g4 + j] = l4[j]; dp[4 * g4 + j] = dl4[g4][j]; }
            }
            s16x8 qfa[4], dofa[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) { qfa[ks] = get_row_frag(qimg, r, ks, h); dofa[ks] = get_row_frag(doimg, r, ks, h); }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) s = mfma32(qfa[ks], kfb[ks], s);      // S'[q][key]
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) dp = mfma32(dofa[ks], vfb[ks], dp);   // dP'[q][key] = dO . V^T - delta (1 - p)
            f32x16 p;
            // Dropout mask of (query row, key): hash of the GROUP (row * ld + key) >> 2 = row * (ld / 4) + (key >> 2), byte
            // key & 3.  A lane owns ONE key and 16 rows, so per lane every element has its own group -- but the four lanes
            // of a quad own the four keys of one group: each lane hashes the 4 rows (reg & 3) == (lane & 3) and the quad
            // shares them by DPP (4 hashes + 16 moves per lane and tile instead of 16 hashes with 64-bit index products:
            // the dropout instances ran 1.9x the time of the plain ones, VALU-bound).  Rows past Lq / keys past Lk carry
            // p = 0 (rows) or feed columns that are never stored (keys): their mask value never matters, so no clamping.
            uint32_t hq[4] = {0u, 0u, 0u, 0u};
            if (DROP) {
                const uint64_t ldq = (uint64_t)(drop_ldk(a.Lk) >> 2);
                const uint64_t g0 = (uint64_t)((b * a.H + head) * a.Lq + (int64_t)qt * 32 + 4 * h + (lane & 3)) * ldq +
                                    (uint64_t)((k0 + (r & ~3)) >> 2);
#pragma unroll
                for (int g = 0; g < 4; ++g) hq[g] = drop_hash(a.drop, g0 + (uint64_t)(8 * g) * ldq);
            }
            const uint32_t rot = 8u * ((uint32_t)r & 3u);
            auto elements = [&](auto last_c) {   // the row-bounds select only exists in the instantiation of the last query tile
                constexpr bool LAST = decltype(last_c)::value;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int64_t qq = (int64_t)qt * 32 + crow(reg, h);
                    float x = MASK ? fmaf(s[reg], a.scale_log2, mk) : s[reg] * a.scale_log2;
                    if (BIAS) x = fmaf(bcol[(qq < a.Lq ? qq : a.Lq - 1) * a.Lk], LOG2E, x);
                    float pv = fast_exp2(x);
                    if (LAST || CAUSAL) {
                        bool valid = true;
                        if (LAST) valid = qq < a.Lq;
                        if (CAUSAL) valid = valid && ki <= qq;
                        pv = valid ? pv : 0.f;
                    }
                    if (DROP) {
                        const uint32_t hv = (reg & 3) == 0 ? quad_bcast<0>(hq[reg >> 2])
                                          : (reg & 3) == 1 ? quad_bcast<1>(hq[reg >> 2])
                                          : (reg & 3) == 2 ? quad_bcast<2>(hq[reg >> 2]) : quad_bcast<3>(hq[reg >> 2]);
                        const bool keep = rotr32(hv, rot) >= a.drop.thr;
                        const float pik = pv * a.drop.inv_keep;
                        p[reg] = keep ? pik : 0.f;                                   // the P that multiplied V in the forward pass
                        s[reg] = pik * (keep ? dp[reg] : dl4[reg >> 2][reg & 3]);   // p (dP / (1 - p) - delta) | p (-delta)
                    } else {
                        p[reg] = pv;
                        s[reg] = pv * dp[reg];
                    }
                }
            };
            if (last) elements(std::true_type{}); else elements(std::false_type{});
            const s16x8 p0 = pack_acc(p, 0), p1 = pack_acc(p, 1);
            const s16x8 d0 = pack_acc(s, 0), d1 = pack_acc(s, 1);
            // dV^T[d][key] += dO^T[d][q] . P[q][key] ;  dK^T[d][key] += Q^T[d][q] . dS[q][key]
            dv0 = mfma32(tr_frag(dotr, 0, 0, lane), p0, dv0);
            dv1 = mfma32(tr_frag(dotr, 1, 0, lane), p0, dv1);
            dv0 = mfma32(tr_frag(dotr, 0, 1, lane), p1, dv0);
            dv1 = mfma32(tr_frag(dotr, 1, 1, lane), p1, dv1);
            dk0 = mfma32(tr_frag(qtr, 0, 0, lane), d0, dk0);
            dk1 = mfma32(tr_frag(qtr, 1, 0, lane), d0, dk1);
            dk0 = mfma32(tr_frag(qtr, 0, 1, lane), d1, dk0);
            dk1 = mfma32(tr_frag(qtr, 1, 1, lane), d1, dk1);
        }
        if (!last) {
            char* nb = lds + ((qt + 1) & 1) * BUF;
            put_row_img(nb, qreg, t);
            put_tr_img(nb + RIMG, qreg, t);
            put_row_img(nb + RIMG + TILE_LDS, doreg, t);
            put_tr_img(nb + 2 * RIMG + TILE_LDS, doreg, t);
            if (t < 16) *(f32x4*)(nb + IMG + 16 * t) = ldreg;
        }
        qreg = qreg2; doreg = doreg2; ldreg = ldreg2;
        coop_barrier();
    }
    if (active && key_ok) {
        store_rows(a.dk + b * a.k_sb + ki * a.k_sl + head * 64, dk0, dk1, a.scale, h);
        store_rows(a.dv + b * a.v_sb + ki * a.v_sl + head * 64, dv0, dv1, 1.0f, h);
    }
}

// =========================================================================================================
// Round 4: ONE LDS image per streamed operand, read by rows (ds_read_b128) AND transposed (ds_read_b64_tr_b16), and TWO streamed
// tiles per barrier.  Counters of the round-3 kernels (profiles/r04_attn_pmc_probe.txt): waves parked at s_waitcnt / s_barrier
// 48 % of their cycles, MFMA pipe 26 %, VALU 30 %, LDS 28 % busy -- nothing saturated, two waves per SIMD that wait; and 40 KiB of
// ds_write_b128 per tile step and CU (five images of 4-6 KiB per workgroup: the store path moves ~79 B / clk) beside the reads.
// Image: [32 rows][64 bf16] in 8-row x 32-column subtiles of 512 B (cdna_hip_programming.md T10 "one image for row reads AND
// transposed reads", form (a) cut to 128-B rows): off(row, ch) = 1024 (row >> 3) + 512 (ch >> 2) + 64 (row & 7)
// + 16 ((ch & 3) ^ ((row >> 2) & 3)), ch = 16-B chunk 0..7.  Row fragments and transposed fragments are both bank-conflict free
// (16 lanes of a ds_read_b128 group cover 4 rows x 4 row groups = 16 different 16-B slots; the 32 lanes of a tr read cover 4
// rows x 4 chunks x 2 halves); the staging store of a whole row by 8 lanes is 2-way (chunks c and c + 4 share a slot), 16 cycles
// instead of 13.  4 KiB per operand tile instead of 10: the double buffer of a TWO-tile step is smaller than the old one-tile one.
// =========================================================================================================
#ifndef TPS_PLAIN
#define TPS_PLAIN 4   // streamed tiles per barrier of the instances without dropout: 4 (-3 % against 2, profiles/r04_attn_tiles_per_barrier.log); the dropout instances keep 2 (registers)
#endif
DEVINL int dual_off(int row, int ch) {
    return 1024 * (row >> 3) + 512 * (ch >> 2) + 64 * (row & 7) + 16 * ((ch & 3) ^ ((row >> 2) & 3));
}
// staging map of the dual images: thread t -> row (t & 127) >> 2, chunk (t & 3) + 4 (t >> 7).  The 8 lanes of a ds_write_b128
// group then cover 2 rows x 4 chunks = 8 different 16-B slots (a whole row by 8 lanes puts chunks c and c + 4 on one slot: 2-way,
// SQ_LDS_BANK_CONFLICT = 13 % of the LDS cycles of the first version); the global loads become 64-B segments of 16 rows.
DEVINL s16x8 dual_load(const bf16_t* base, int64_t sl, int64_t row0, int64_t nrows, int t) {
    int64_t row = row0 + ((t & 127) >> 2);
    row = row < nrows ? row : nrows - 1;
    return *(const s16x8*)(base + row * sl + ((t & 3) + 4 * (t >> 7)) * 8);
}
DEVINL void put_dual_img(char* img, s16x8 v, int t) { *(s16x8*)(img + dual_off((t & 127) >> 2, (t & 3) + 4 * (t >> 7))) = v; }
DEVINL s16x8 dual_row_frag(const char* img, int r, int ks, int h) { return *(const s16x8*)(img + dual_off(r, 2 * ks + h)); }
DEVINL s16x8 dual_tr_frag(const char* img, int dt, int s, int lane) {
    const int h = lane >> 5, gg = (lane >> 4) & 1, i16 = lane & 15, qq = i16 >> 2, p = i16 & 3;
    const int row = 16 * s + 4 * h + qq, ch = 4 * dt + 2 * gg + (p >> 1), byte = 8 * (p & 1);
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + dual_off(row, ch) + byte));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + dual_off(row + 8, ch) + byte));
    return (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// dK / dV of the plain / key-masked / dropout instances (BERT, ViT): the math of attn_bwd_dkdv_coop_kernel on the dual images,
// two query tiles per barrier (one global prefetch step = two tiles ahead of the tile being computed).
template <bool MASK, bool DROP, int TPS>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkdv2_kernel(AttnArgs a) {
    if (DROP) drop_resolve(a.drop);
    constexpr int TIMG = 4096, SUB = 2 * TIMG + 256, BUF = TPS * SUB;   // per sub-tile: Q image, dO image, 32 x (-lse / c), 32 x (-delta')
    constexpr int UNR = DROP ? 1 : 2;   // sub-tiles interleaved by the compiler (registers)
    __shared__ __attribute__((aligned(16))) char lds[2 * BUF];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = lane & 31, h = lane >> 5;
    const AttnBlock blk = attn_block();
    const int head = blk.head;
    const int64_t b = blk.b;
    const int64_t k0 = ((int64_t)blk.bx * 4 + wave) * 32;
    const bool active = k0 < a.Lk;

    const int64_t ki = k0 + r;
    const int64_t krow = ki < a.Lk ? ki : a.Lk - 1;
    const bf16_t* kp = a.k + b * a.k_sb + krow * a.k_sl + head * 64;
    const bf16_t* vp = a.v + b * a.v_sb + krow * a.v_sl + head * 64;
    s16x8 kfb[4], vfb[4];  // B operands: K^T[d][key], V^T[d][key]
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { kfb[ks] = row_frag(kp, ks, h); vfb[ks] = row_frag(vp, ks, h); }
    const float mk = MASK ? a.key_mask[b * a.Lk + krow] * LOG2E : 0.f;
    const bool key_ok = ki < a.Lk;

    const bf16_t* qbase = a.q + b * a.q_sb + head * 64;
    const bf16_t* dobase = a.d_o + b * a.o_sb + head * 64;
    const float* lrow = a.lse + (b * a.H + head) * a.lse_stride;
    const float* drow = a.delta + (b * a.H + head) * a.lse_stride;

    f32x16 dk0 = zero16(), dk1 = zero16(), dv0 = zero16(), dv1 = zero16();
    const int nqt = (int)((a.Lq + 31) / 32);
    const int nst = (nqt + TPS - 1) / TPS;
    // threads 0..15 stage sub-tile 0's table, 16..31 sub-tile 1's: -lse / c (first 8 of each) and -delta (1 - p) (last 8)
    const int tsub = t >> 4, tq = t & 15;   // (threads 0 .. 16 TPS - 1 stage the tables)
    const float* ldsrc = (tq < 8 ? lrow : drow) + 4 * (tq & 7);
    const float ldmul = tq < 8 ? -1.0f / a.scale_log2 : (DROP ? -1.0f / a.drop.inv_keep : -1.0f);
    const int64_t ld_rows = a.lse_stride;   // a multiple of 32: the last step's second sub-tile may lie wholly past it
    s16x8 qreg[TPS], doreg[TPS];
    f32x4 ldreg = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto fetch = [&](int st) {   // the two tiles of step st (rows clamped; tiles past the end are never computed)
#pragma unroll
        for (int u = 0; u < TPS; ++u) {
            qreg[u] = dual_load(qbase, a.q_sl, (int64_t)(TPS * st + u) * 32, a.Lq, t);
            doreg[u] = dual_load(dobase, a.o_sl, (int64_t)(TPS * st + u) * 32, a.Lq, t);
        }
        if (t < 16 * TPS) {
            const int64_t row0 = (int64_t)(TPS * st + tsub) * 32;
            ldreg = row0 < ld_rows ? *(const f32x4*)(ldsrc + row0) * ldmul : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    auto stage = [&](char* buf) {
#pragma unroll
        for (int u = 0; u < TPS; ++u) {
            put_dual_img(buf + u * SUB, qreg[u], t);
            put_dual_img(buf + u * SUB + TIMG, doreg[u], t);
        }
        if (t < 16 * TPS) *(f32x4*)(buf + tsub * SUB + 2 * TIMG + 16 * tq) = ldreg;
    };
    fetch(0);
    stage(lds);
    coop_barrier();
    for (int st = 0; st < nst; ++st) {
        const bool more = st + 1 < nst;
        if (more) fetch(st + 1);   // lands under this step's two tiles; written to the other buffer behind them
        const char* buf = lds + (st & 1) * BUF;
        if (active) {
#pragma unroll UNR   // (the dropout instances' two sub-tiles interleaved need more than 256 registers: one after the other)
            for (int u = 0; u < TPS; ++u) {
                const int qt = TPS * st + u;
                if (qt >= nqt) break;
                const bool last = qt + 1 == nqt;
                const char* qimg = buf + u * SUB;
                const char* doimg = qimg + TIMG;
                const char* tbl = doimg + TIMG;
                f32x16 s, dp;
                f32x4 dl4[4];
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {  // broadcast LDS reads: rows 8 g4 + 4 h .. + 3 of the tile -> accumulator start values
                    const f32x4 l4 = *(const f32x4*)(tbl + 4 * (8 * g4 + 4 * h));
                    dl4[g4] = *(const f32x4*)(tbl + 128 + 4 * (8 * g4 + 4 * h));
#pragma unroll
                    for (int j = 0; j < 4; ++j) { s[4 * g4 + j] = l4[j]; dp[4 * g4 + j] = dl4[g4][j]; }
                }
                s16x8 qfa[4], dofa[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) { qfa[ks] = dual_row_frag(qimg, r, ks, h); dofa[ks] = dual_row_frag(doimg, r, ks, h); }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) s = mfma32(qfa[ks], kfb[ks], s);      // S'[q][key] = Q K^T - lse / c
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) dp = mfma32(dofa[ks], vfb[ks], dp);   // dP'[q][key] = dO V^T - delta (1 - p)
                f32x16 p;
                uint32_t hq[4] = {0u, 0u, 0u, 0u};
                if (DROP) {   // (the quad-shared hashes of attn_bwd_dkdv_coop_kernel)
                    const uint64_t ldq = (uint64_t)(drop_ldk(a.Lk) >> 2);
                    const uint64_t g0 = (uint64_t)((b * a.H + head) * a.Lq + (int64_t)qt * 32 + 4 * h + (lane & 3)) * ldq +
                                        (uint64_t)((k0 + (r & ~3)) >> 2);
#pragma unroll
                    for (int g = 0; g < 4; ++g) hq[g] = drop_hash(a.drop, g0 + (uint64_t)(8 * g) * ldq);
                }
                const uint32_t rot = 8u * ((uint32_t)r & 3u);
                auto elements = [&](auto last_c) {
                    constexpr bool LAST = decltype(last_c)::value;
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const float x = MASK ? fmaf(s[reg], a.scale_log2, mk) : s[reg] * a.scale_log2;
                        float pv = fast_exp2(x);
                        if (LAST) pv = (int64_t)qt * 32 + crow(reg, h) < a.Lq ? pv : 0.f;
                        if (DROP) {
                            const uint32_t hv = (reg & 3) == 0 ? quad_bcast<0>(hq[reg >> 2])
                                              : (reg & 3) == 1 ? quad_bcast<1>(hq[reg >> 2])
                                              : (reg & 3) == 2 ? quad_bcast<2>(hq[reg >> 2]) : quad_bcast<3>(hq[reg >> 2]);
                            const bool keep = rotr32(hv, rot) >= a.drop.thr;
                            const float pik = pv * a.drop.inv_keep;
                            p[reg] = keep ? pik : 0.f;
                            s[reg] = pik * (keep ? dp[reg] : dl4[reg >> 2][reg & 3]);
                        } else {
                            p[reg] = pv;
                            s[reg] = pv * dp[reg];
                        }
                    }
                };
                if (last) elements(std::true_type{}); else elements(std::false_type{});
                const s16x8 p0 = pack_acc(p, 0), p1 = pack_acc(p, 1);
                const s16x8 d0 = pack_acc(s, 0), d1 = pack_acc(s, 1);
                // dV^T[d][key] += dO^T[d][q] . P[q][key] ;  dK^T[d][key] += Q^T[d][q] . dS[q][key]
                dv0 = mfma32(dual_tr_frag(doimg, 0, 0, lane), p0, dv0);
                dv1 = mfma32(dual_tr_frag(doimg, 1, 0, lane), p0, dv1);
                dv0 = mfma32(dual_tr_frag(doimg, 0, 1, lane), p1, dv0);
                dv1 = mfma32(dual_tr_frag(doimg, 1, 1, lane), p1, dv1);
                dk0 = mfma32(dual_tr_frag(qimg, 0, 0, lane), d0, dk0);
                dk1 = mfma32(dual_tr_frag(qimg, 1, 0, lane), d0, dk1);
                dk0 = mfma32(dual_tr_frag(qimg, 0, 1, lane), d1, dk0);
                dk1 = mfma32(dual_tr_frag(qimg, 1, 1, lane), d1, dk1);
            }
        }
        if (more) stage(lds + ((st + 1) & 1) * BUF);
        coop_barrier();
    }
    if (active && key_ok) {
        store_rows(a.dk + b * a.k_sb + ki * a.k_sl + head * 64, dk0, dk1, a.scale, h);
        store_rows(a.dv + b * a.v_sb + ki * a.v_sl + head * 64, dv0, dv1, 1.0f, h);
    }
}

// dQ (and delta) of the plain / key-masked / dropout instances: the math of attn_bwd_dq_coop_kernel on the dual images (K: row
// fragments for S, transposed fragments for dQ; V: row fragments for dP), two key tiles per barrier.
template <bool MASK, bool DROP, int TPS>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq2_kernel(AttnArgs a) {
    if (DROP) drop_resolve(a.drop);
    constexpr int TIMG = 4096, SUB = 2 * TIMG, BUF = TPS * SUB;   // per sub-tile: K image, V image
    constexpr int UNR = DROP ? 1 : 2;   // sub-tiles interleaved by the compiler (registers)
    __shared__ __attribute__((aligned(16))) char lds[2 * BUF];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = lane & 31, h = lane >> 5;
    const AttnBlock blk = attn_block();
    const int head = blk.head;
    const int64_t b = blk.b;
    const int64_t q0 = ((int64_t)blk.bx * 4 + wave) * 32;
    const bool active = q0 < a.Lq;

    const int64_t qi = q0 + r;
    const int64_t qrow = qi < a.Lq ? qi : a.Lq - 1;
    const bf16_t* qp = a.q + b * a.q_sb + qrow * a.q_sl + head * 64;
    const bf16_t* dop = a.d_o + b * a.o_sb + qrow * a.o_sl + head * 64;
    s16x8 qf[4], dof[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { qf[ks] = row_frag(qp, ks, h); dof[ks] = row_frag(dop, ks, h); }
    const float lse = a.lse[(b * a.H + head) * a.lse_stride + qrow];
    float dlt = 0.f;   // delta[q] = sum_d dO[q][d] O[q][d] (published for the dK/dV kernel, as attn_bwd_dq_coop_kernel does)
    {
        const bf16_t* op = a.o + b * a.o_sb + qrow * a.o_sl + head * 64;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const s16x8 of = row_frag(op, ks, h);
#pragma unroll
            for (int j = 0; j < 8; ++j) dlt = fmaf(bf2f((bf16_t)of[j]), bf2f((bf16_t)dof[ks][j]), dlt);
        }
        dlt += __shfl_xor(dlt, 32, 64);
        if (h == 0 && active && qi < a.lse_stride)
            a.delta[(b * a.H + head) * a.lse_stride + qi] = qi < a.Lq ? dlt : 0.f;
    }
    const float s_init = -lse / a.scale_log2;
    const float keep_p = DROP ? 1.0f / a.drop.inv_keep : 1.0f;
    const float dp_init = -dlt * keep_p, dl_drop = -dlt * keep_p;
    const bf16_t* kbase = a.k + b * a.k_sb + head * 64;
    const bf16_t* vbase = a.v + b * a.v_sb + head * 64;
    const float* mrow = MASK ? a.key_mask + b * a.Lk : nullptr;

    f32x16 g0 = zero16(), g1 = zero16();
    const int nkt = (int)((a.Lk + 31) / 32);
    const int nst = (nkt + TPS - 1) / TPS;
    const uint64_t drow = (uint64_t)(((b * a.H + head) * a.Lq + qi) * drop_ldk(a.Lk));
    s16x8 kreg[TPS], vreg[TPS];
    auto fetch = [&](int st) {
#pragma unroll
        for (int u = 0; u < TPS; ++u) {
            kreg[u] = dual_load(kbase, a.k_sl, (int64_t)(TPS * st + u) * 32, a.Lk, t);
            vreg[u] = dual_load(vbase, a.v_sl, (int64_t)(TPS * st + u) * 32, a.Lk, t);
        }
    };
    auto stage = [&](char* buf) {
#pragma unroll
        for (int u = 0; u < TPS; ++u) {
            put_dual_img(buf + u * SUB, kreg[u], t);
            put_dual_img(buf + u * SUB + TIMG, vreg[u], t);
        }
    };
    fetch(0);
    stage(lds);
    coop_barrier();
    for (int st = 0; st < nst; ++st) {
        const bool more = st + 1 < nst;
        if (more) fetch(st + 1);
        const char* buf = lds + (st & 1) * BUF;
        if (active) {
#pragma unroll UNR
            for (int u = 0; u < TPS; ++u) {
                const int kt = TPS * st + u;
                if (kt >= nkt) break;
                const bool last = kt + 1 == nkt;
                const char* kimg = buf + u * SUB;
                const char* vimg = kimg + TIMG;
                s16x8 kf[4], vf[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) { kf[ks] = dual_row_frag(kimg, r, ks, h); vf[ks] = dual_row_frag(vimg, r, ks, h); }
                f32x16 s, dp;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) { s[reg] = s_init; dp[reg] = dp_init; }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) s = mfma32(kf[ks], qf[ks], s);      // S'^T[key][q]
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) dp = mfma32(vf[ks], dof[ks], dp);   // dP'^T[key][q]
                uint32_t dh4[4] = {0u, 0u, 0u, 0u};
                if (DROP) {
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) dh4[r4] = drop_hash(a.drop, (drow + (uint64_t)(kt * 32 + 8 * r4 + 4 * h)) >> 2);
                }
                auto elements = [&](auto last_c) {
                    constexpr bool LAST = decltype(last_c)::value;
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int64_t key = (int64_t)kt * 32 + crow(reg, h);
                        float x = s[reg] * a.scale_log2;
                        if (MASK) x = fmaf(mrow[(LAST && key >= a.Lk) ? a.Lk - 1 : key], LOG2E, x);
                        float p = fast_exp2(x);
                        if (LAST) p = key < a.Lk ? p : 0.f;
                        float tt = dp[reg];
                        if (DROP) tt = (rotr32(dh4[reg >> 2], 8u * (reg & 3)) >= a.drop.thr ? tt : dl_drop) * a.drop.inv_keep;
                        s[reg] = p * tt;
                    }
                };
                if (last) elements(std::true_type{}); else elements(std::false_type{});
                const s16x8 d0 = pack_acc(s, 0), d1 = pack_acc(s, 1);
                // dQ^T[d][q] += K^T[d][key] . dS^T[key][q]
                g0 = mfma32(dual_tr_frag(kimg, 0, 0, lane), d0, g0);
                g1 = mfma32(dual_tr_frag(kimg, 1, 0, lane), d0, g1);
                g0 = mfma32(dual_tr_frag(kimg, 0, 1, lane), d1, g0);
                g1 = mfma32(dual_tr_frag(kimg, 1, 1, lane), d1, g1);
            }
        }
        if (more) stage(lds + ((st + 1) & 1) * BUF);
        coop_barrier();
    }
    if (active && qi < a.Lq) store_rows(a.dq + b * a.q_sb + qi * a.q_sl + head * 64, g0, g1, a.scale, h);
}

// (A forward kernel in the same mould -- K by rows, V transposed from dual images, two key tiles per barrier -- was built and measured
// SLOWER than attn_fwd_coop_kernel: 543 vs 452 us plain, 686 vs 620 with dropout, 784 vs 571 with a key mask at 577 x 577, B = 256,
// profiles/r04_attn_generation_check.log: the forward reads each image one way only, so the dual layout saves no staging stores
// there, and at 32 KiB of LDS and 167 registers it loses the fourth wave per SIMD the round-3 kernel runs with.  Not kept.)

// coop kernels: mask / bias / causal as below plus the dropout instances: BERT layers (optional key mask) and T5 layers
// (relative-position bias, optionally causal)
#define ATTN_DISPATCH_COOP(rc, KERNEL, grid, s, a, ...)                                                                  \
    do {                                                                                                                 \
        const int f = (a.key_mask ? 1 : 0) | (a.pos_bias ? 2 : 0) | (a.causal ? 4 : 0);                                  \
        if (a.has_drop) {                                                                                                \
            if (f == 0) hipLaunchKernelGGL((KERNEL<__VA_ARGS__ false, false, false, true>), grid, dim3(256), 0, s, a);   \
            else if (f == 1) hipLaunchKernelGGL((KERNEL<__VA_ARGS__ true, false, false, true>), grid, dim3(256), 0, s, a); \
            else if (f == 2) hipLaunchKernelGGL((KERNEL<__VA_ARGS__ false, true, false, true>), grid, dim3(256), 0, s, a); \
            else if (f == 6) hipLaunchKernelGGL((KERNEL<__VA_ARGS__ false, true, true, true>), grid, dim3(256), 0, s, a);  \
            else rc = M3AE_ERR_UNSUPPORTED;                                                                              \
            break;                                                                                                       \
        }                                                                                                                \
        switch (f) {                                                                                                     \
            case 0: hipLaunchKernelGGL((KERNEL<__VA_ARGS__ false, false, false, false>), grid, dim3(256), 0, s, a); break; \
            case 1: hipLaunchKernelGGL((KERNEL<__VA_ARGS__ true, false, false, false>), grid, dim3(256), 0, s, a); break;  \
            case 2: hipLaunchKernelGGL((KERNEL<__VA_ARGS__ false, true, false, false>), grid, dim3(256), 0, s, a); break;  \
            case 3: hipLaunchKernelGGL((KERNEL<__VA_ARGS__ true, true, false, false>), grid, dim3(256), 0, s, a); break;   \
            case 4: hipLaunchKernelGGL((KERNEL<__VA_ARGS__ false, false, true, false>), grid, dim3(256), 0, s, a); break;  \
            case 5: hipLaunchKernelGGL((KERNEL<__VA_ARGS__ true, false, true, false>), grid, dim3(256), 0, s, a); break;   \
            case 6: hipLaunchKernelGGL((KERNEL<__VA_ARGS__ false, true, true, false>), grid, dim3(256), 0, s, a); break;   \
            default: hipLaunchKernelGGL((KERNEL<__VA_ARGS__ true, true, true, false>), grid, dim3(256), 0, s, a); break;   \
        }                                                                                                                \
    } while (0)

// flag dispatch (mask / bias / causal are wave-uniform launch properties)
// ---------------------------------------------------------------------------------------------------------
// fp32 reference-shaped path: row softmax kernels over the materialised score matrix
// ---------------------------------------------------------------------------------------------------------
// S[b][h][q][:] <- softmax(S + key_mask[b][:] + pos_bias[h][q][:])   (S already scaled by the GEMM's alpha)
__global__ void softmax_rows_kernel(float* S, const float* key_mask, const float* pos_bias, int64_t B, int64_t H,
                                    int64_t Lq, int64_t Lk, int causal) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= B * H * Lq) return;
    const int64_t q = row % Lq, hh = (row / Lq) % H, b = row / (Lq * H);
    float* s = S + row * Lk;
    const float* mk = key_mask ? key_mask + b * Lk : nullptr;
    const float* pb = pos_bias ? pos_bias + (hh * Lq + q) * Lk : nullptr;
    float mx = -INFINITY;
    for (int64_t j = lane; j < Lk; j += 64) {
        float x = s[j];
        if (mk) x += mk[j];
        if (pb) x += pb[j];
        if (causal && j > q) x = -INFINITY;
        s[j] = x;
        mx = fmaxf(mx, x);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int64_t j = lane; j < Lk; j += 64) {
        const float e = expf(s[j] - mx);
        s[j] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int64_t j = lane; j < Lk; j += 64) s[j] *= inv;
}

// dS <- P * (dP - sum_j P dP)   (in place on dP)
__global__ void softmax_bwd_rows_kernel(const float* P, float* dP, int64_t rows, int64_t Lk) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* p = P + row * Lk;
    float* d = dP + row * Lk;
    float dot = 0.f;
    for (int64_t j = lane; j < Lk; j += 64) dot += p[j] * d[j];
    dot = wave_sum(dot);
    for (int64_t j = lane; j < Lk; j += 64) d[j] = p[j] * (d[j] - dot);
}

// d_pos_bias[h][q][k] += sum_b dS[b][h][q][k]
__global__ void pos_bias_grad_kernel(const float* dS, float* dpb, int64_t B, int64_t HQK) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HQK) return;
    float acc = 0.f;
    for (int64_t b = 0; b < B; ++b) acc += dS[b * HQK + i];
    dpb[i] += acc;
}

m3ae_gemm_desc bgemm(const m3ae_attn_desc& d) {
    m3ae_gemm_desc g{};
    g.batch1 = d.B; g.batch2 = d.H;
    g.dtype_a = g.dtype_b = g.dtype_c = M3AE_F32;
    g.alpha = 1.0f;
    return g;
}

int attn_f32_scores(const m3ae_attn_desc& d, float* S, hipStream_t s) {
    const int64_t QK = d.Lq * d.Lk;
    m3ae_gemm_desc g = bgemm(d);
    g.M = d.Lq; g.N = d.Lk; g.K = d.Dh; g.alpha = d.scale;
    g.A = d.q; g.a_sm = d.q_sl; g.a_sk = 1; g.a_sb1 = d.q_sb; g.a_sb2 = d.Dh;
    g.B = d.k; g.b_sk = 1; g.b_sn = d.k_sl; g.b_sb1 = d.k_sb; g.b_sb2 = d.Dh;
    g.C = S; g.c_sm = d.Lk; g.c_sn = 1; g.c_sb1 = d.H * QK; g.c_sb2 = QK;
    int rc = m3ae_gemm_generic(g, s);
    if (rc) return rc;
    const int64_t rows = d.B * d.H * d.Lq;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, s, S, d.key_mask, d.pos_bias,
                       d.B, d.H, d.Lq, d.Lk, d.causal);
    return hip_launch_status();
}

AttnArgs to_args(const m3ae_attn_desc& d) {
    AttnArgs a{};
    a.q = (const bf16_t*)d.q; a.q_sb = d.q_sb; a.q_sl = d.q_sl;
    a.k = (const bf16_t*)d.k; a.k_sb = d.k_sb; a.k_sl = d.k_sl;
    a.v = (const bf16_t*)d.v; a.v_sb = d.v_sb; a.v_sl = d.v_sl;
    a.o = (bf16_t*)d.o; a.o_sb = d.o_sb; a.o_sl = d.o_sl;
    a.key_mask = d.key_mask; a.pos_bias = d.pos_bias;
    a.scale = d.scale; a.scale_log2 = d.scale * LOG2E; a.causal = d.causal;
    a.lse = d.lse; a.lse_stride = d.lse_stride;
    a.B = d.B; a.H = d.H; a.Lq = d.Lq; a.Lk = d.Lk;
    a.d_o = (const bf16_t*)d.d_o; a.dq = (bf16_t*)d.dq; a.dk = (bf16_t*)d.dk; a.dv = (bf16_t*)d.dv;
    a.delta = d.delta; a.d_pos_bias = d.d_pos_bias;
    a.has_drop = d.dropout_p > 0.f;
    a.drop = make_drop(d.dropout_p, d.dropout_seed, d.dropout_salt);
    return a;
}

bool bf16_layout_ok(const m3ae_attn_desc& d, bool bwd) {
    auto al = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
    auto st = [](int64_t x) { return x % 8 == 0; };
    bool ok = d.Dh == 64 && al(d.q) && al(d.k) && al(d.v) && al(d.o) && st(d.q_sl) && st(d.k_sl) && st(d.v_sl) &&
              st(d.o_sl) && st(d.q_sb) && st(d.k_sb) && st(d.v_sb) && st(d.o_sb) && d.lse && d.lse_stride >= d.Lq &&
              d.lse_stride % 32 == 0;
    if (bwd) ok = ok && al(d.d_o) && al(d.dq) && al(d.dk) && al(d.dv) && d.delta;
    return ok;
}

}  // namespace


extern "C" int64_t m3ae_attn_workspace_bytes(const m3ae_attn_desc* d, int backward) {
    if (!d) return 0;
    if (d->dtype == M3AE_BF16) return 0;
    const int64_t one = d->B * d->H * d->Lq * d->Lk * (int64_t)sizeof(float);
    return backward ? 2 * one : one;
}

extern "C" int m3ae_attn_fwd(const m3ae_attn_desc* dp, void* stream) {
    if (!dp || !dp->q || !dp->k || !dp->v || !dp->o) return M3AE_ERR_ARG;
    const m3ae_attn_desc& d = *dp;
    if (d.B <= 0 || d.H <= 0 || d.Lq <= 0 || d.Lk <= 0 || d.Dh <= 0) return M3AE_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (d.dtype == M3AE_BF16) {
        if (!bf16_layout_ok(d, false)) return M3AE_ERR_UNSUPPORTED;
        if (d.H > 65535 || d.B > 65535) return M3AE_ERR_UNSUPPORTED;
        AttnArgs a = to_args(d);
        // 32 query rows per wave everywhere (577 = 18 x 32 + 1: finer blocks waste less, occupancy 4: 188 -> 135 us at B = 64
        // against 64 rows per wave on the long sequences)
        int rc = 0;
        dim3 grid((unsigned)cdiv(d.Lq, 128), (unsigned)d.H, (unsigned)d.B);
        ATTN_DISPATCH_COOP(rc, attn_fwd_coop_kernel, grid, s, a, 1, );
        if (rc) return rc;
        return hip_launch_status();
    }
    if (d.dtype != M3AE_F32) return M3AE_ERR_UNSUPPORTED;
    if (!d.workspace || d.workspace_bytes < m3ae_attn_workspace_bytes(dp, 0)) return M3AE_ERR_WORKSPACE;
    float* S = (float*)d.workspace;
    int rc = attn_f32_scores(d, S, s);
    if (rc) return rc;
    if (d.dropout_p > 0.f &&  // P is [B H Lq][Lk]: rows x cols of the same mask index the bf16 kernels use
        (rc = m3ae_dropout(S, S, nullptr, d.B * d.H * d.Lq, d.Lk, d.dropout_p, d.dropout_seed, d.dropout_salt, M3AE_F32, stream)))
        return rc;
    const int64_t QK = d.Lq * d.Lk;
    m3ae_gemm_desc g = bgemm(d);
    g.M = d.Lq; g.N = d.Dh; g.K = d.Lk;
    g.A = S; g.a_sm = d.Lk; g.a_sk = 1; g.a_sb1 = d.H * QK; g.a_sb2 = QK;
    g.B = d.v; g.b_sk = d.v_sl; g.b_sn = 1; g.b_sb1 = d.v_sb; g.b_sb2 = d.Dh;
    g.C = d.o; g.c_sm = d.o_sl; g.c_sn = 1; g.c_sb1 = d.o_sb; g.c_sb2 = d.Dh;
    return m3ae_gemm_generic(g, s);
}

extern "C" int m3ae_attn_bwd(const m3ae_attn_desc* dp, void* stream) {
    if (!dp || !dp->q || !dp->k || !dp->v || !dp->o || !dp->d_o || !dp->dq || !dp->dk || !dp->dv)
        return M3AE_ERR_ARG;
    const m3ae_attn_desc& d = *dp;
    hipStream_t s = (hipStream_t)stream;
    if (d.dtype == M3AE_BF16) {
        if (!bf16_layout_ok(d, true)) return M3AE_ERR_UNSUPPORTED;
        AttnArgs a = to_args(d);
        dim3 gq((unsigned)cdiv(cdiv(d.Lq, 32), 4), (unsigned)d.H, (unsigned)d.B);
        dim3 gk((unsigned)cdiv(cdiv(d.Lk, 32), 4), (unsigned)d.H, (unsigned)d.B);
        int rc = 0;
        const bool gen4 = !a.pos_bias && !a.causal && !(d.launch_flags & M3AE_ATTN_LEGACY_KERNELS);
        if (gen4) {   // round-4 kernels: one LDS image per operand, two tiles per barrier; also publishes delta = rowsum(dO * O)
            if (a.key_mask) { if (a.has_drop) hipLaunchKernelGGL((attn_bwd_dq2_kernel<true, true, 2>), gq, dim3(256), 0, s, a);
                              else hipLaunchKernelGGL((attn_bwd_dq2_kernel<true, false, TPS_PLAIN>), gq, dim3(256), 0, s, a); }
            else { if (a.has_drop) hipLaunchKernelGGL((attn_bwd_dq2_kernel<false, true, 2>), gq, dim3(256), 0, s, a);
                   else hipLaunchKernelGGL((attn_bwd_dq2_kernel<false, false, TPS_PLAIN>), gq, dim3(256), 0, s, a); }
        } else {
            ATTN_DISPATCH_COOP(rc, attn_bwd_dq_coop_kernel, gq, s, a, );   // also computes (and publishes) delta = rowsum(dO * O)
        }
        if (gen4) {
            if (a.key_mask) { if (a.has_drop) hipLaunchKernelGGL((attn_bwd_dkdv2_kernel<true, true, 2>), gk, dim3(256), 0, s, a);
                              else hipLaunchKernelGGL((attn_bwd_dkdv2_kernel<true, false, TPS_PLAIN>), gk, dim3(256), 0, s, a); }
            else { if (a.has_drop) hipLaunchKernelGGL((attn_bwd_dkdv2_kernel<false, true, 2>), gk, dim3(256), 0, s, a);
                   else hipLaunchKernelGGL((attn_bwd_dkdv2_kernel<false, false, TPS_PLAIN>), gk, dim3(256), 0, s, a); }
        } else {
            ATTN_DISPATCH_COOP(rc, attn_bwd_dkdv_coop_kernel, gk, s, a, );
        }
        if (rc) return rc;
        return hip_launch_status();
    }
    if (d.dtype != M3AE_F32) return M3AE_ERR_UNSUPPORTED;
    if (!d.workspace || d.workspace_bytes < m3ae_attn_workspace_bytes(dp, 1)) return M3AE_ERR_WORKSPACE;
    const int64_t QK = d.Lq * d.Lk;
    const bool drop = d.dropout_p > 0.f;
    float* P = (float*)d.workspace;
    float* dS = P + d.B * d.H * QK;
    int rc = attn_f32_scores(d, P, s);
    if (rc) return rc;
    if (drop && (rc = m3ae_dropout(P, P, nullptr, d.B * d.H * d.Lq, d.Lk, d.dropout_p, d.dropout_seed, d.dropout_salt, M3AE_F32, stream)))
        return rc;  // dV below needs the dropped P that multiplied V in the forward pass
    // dP = dO . V^T
    m3ae_gemm_desc g = bgemm(d);
    g.M = d.Lq; g.N = d.Lk; g.K = d.Dh;
    g.A = d.d_o; g.a_sm = d.o_sl; g.a_sk = 1; g.a_sb1 = d.o_sb; g.a_sb2 = d.Dh;
    g.B = d.v; g.b_sk = 1; g.b_sn = d.v_sl; g.b_sb1 = d.v_sb; g.b_sb2 = d.Dh;
    g.C = dS; g.c_sm = d.Lk; g.c_sn = 1; g.c_sb1 = d.H * QK; g.c_sb2 = QK;
    if ((rc = m3ae_gemm_generic(g, s))) return rc;
    // dV = P^T . dO   (before dS overwrites nothing of P)
    g = bgemm(d);
    g.M = d.Lk; g.N = d.Dh; g.K = d.Lq;
    g.A = P; g.a_sm = 1; g.a_sk = d.Lk; g.a_sb1 = d.H * QK; g.a_sb2 = QK;
    g.B = d.d_o; g.b_sk = d.o_sl; g.b_sn = 1; g.b_sb1 = d.o_sb; g.b_sb2 = d.Dh;
    g.C = d.dv; g.c_sm = d.v_sl; g.c_sn = 1; g.c_sb1 = d.v_sb; g.c_sb2 = d.Dh;
    if ((rc = m3ae_gemm_generic(g, s))) return rc;
    const int64_t rows = d.B * d.H * d.Lq;
    if (drop) {  // the softmax backward needs the un-dropped P and dP wrt it
        if ((rc = attn_f32_scores(d, P, s))) return rc;
        if ((rc = m3ae_dropout(dS, dS, nullptr, d.B * d.H * d.Lq, d.Lk, d.dropout_p, d.dropout_seed, d.dropout_salt, M3AE_F32, stream)))
            return rc;
    }
    hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, s, P, dS, rows, d.Lk);
    if (d.d_pos_bias) {
        const int64_t HQK = d.H * QK;
        hipLaunchKernelGGL(pos_bias_grad_kernel, dim3((unsigned)cdiv(HQK, 256)), dim3(256), 0, s, dS, d.d_pos_bias,
                           d.B, HQK);
    }
    // dQ = scale * dS . K
    g = bgemm(d);
    g.M = d.Lq; g.N = d.Dh; g.K = d.Lk; g.alpha = d.scale;
    g.A = dS; g.a_sm = d.Lk; g.a_sk = 1; g.a_sb1 = d.H * QK; g.a_sb2 = QK;
    g.B = d.k; g.b_sk = d.k_sl; g.b_sn = 1; g.b_sb1 = d.k_sb; g.b_sb2 = d.Dh;
    g.C = d.dq; g.c_sm = d.q_sl; g.c_sn = 1; g.c_sb1 = d.q_sb; g.c_sb2 = d.Dh;
    if ((rc = m3ae_gemm_generic(g, s))) return rc;
    // dK = scale * dS^T . Q
    g = bgemm(d);
    g.M = d.Lk; g.N = d.Dh; g.K = d.Lq; g.alpha = d.scale;
    g.A = dS; g.a_sm = 1; g.a_sk = d.Lk; g.a_sb1 = d.H * QK; g.a_sb2 = QK;
    g.B = d.q; g.b_sk = d.q_sl; g.b_sn = 1; g.b_sb1 = d.q_sb; g.b_sb2 = d.Dh;
    g.C = d.dk; g.c_sm = d.k_sl; g.c_sn = 1; g.c_sb1 = d.k_sb; g.c_sb2 = d.Dh;
    if ((rc = m3ae_gemm_generic(g, s))) return rc;
    return hip_launch_status();
}
