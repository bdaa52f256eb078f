// LayerNorm / RMSNorm forward + backward (HBM-bound: one pass over x, fp32 statistics, wave64 reductions).
//
// One wave per row; the row lives in registers (4-element chunks, CPL chunks per lane) so x is read once.
// Statistics are two-pass in registers (mean, then variance of the centred values), matching nn.LayerNorm's
// fp32 numerics (clip_model.py:27-33 upcasts to fp32; bert_model.py:357 eps 1e-12 needs the centred form).
// Backward keeps per-lane partial dgamma/dbeta over the rows a wave walks, combines the 4 waves of a workgroup in
// LDS, writes one partial row per workgroup, and a second kernel folds the partials into the fp32 gradients.
#include "common.h"

namespace {

template <typename T> DEVINL void ld4(const T* p, float* x);
template <> DEVINL void ld4<float>(const float* p, float* x) {
    const f32x4 v = *(const f32x4*)p;
    x[0] = v[0]; x[1] = v[1]; x[2] = v[2]; x[3] = v[3];
}
template <> DEVINL void ld4<bf16_t>(const bf16_t* p, float* x) {
    const u32x2 v = *(const u32x2*)p;
    x[0] = __uint_as_float(v[0] << 16); x[1] = __uint_as_float(v[0] & 0xffff0000u);
    x[2] = __uint_as_float(v[1] << 16); x[3] = __uint_as_float(v[1] & 0xffff0000u);
}
// raw (unconverted) 4-element chunk: a load can be issued a row ahead and unpacked when it is used
template <typename T> struct Raw4;
template <> struct Raw4<float> {
    f32x4 v;
    DEVINL void ld(const float* p) { v = *(const f32x4*)p; }
    DEVINL void get(float* x) const { x[0] = v[0]; x[1] = v[1]; x[2] = v[2]; x[3] = v[3]; }
};
template <> struct Raw4<bf16_t> {
    u32x2 v;
    DEVINL void ld(const bf16_t* p) { v = *(const u32x2*)p; }
    DEVINL void get(float* x) const {
        x[0] = __uint_as_float(v[0] << 16); x[1] = __uint_as_float(v[0] & 0xffff0000u);
        x[2] = __uint_as_float(v[1] << 16); x[3] = __uint_as_float(v[1] & 0xffff0000u);
    }
};
template <typename T> DEVINL void st4(T* p, const float* x);
template <> DEVINL void st4<float>(float* p, const float* x) { *(f32x4*)p = (f32x4){x[0], x[1], x[2], x[3]}; }
template <> DEVINL void st4<bf16_t>(bf16_t* p, const float* x) {
    *(u32x2*)p = (u32x2){pack2bf(x[0], x[1]), pack2bf(x[2], x[3])};
}

template <typename T, int CPL>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* x, const float* gamma, const float* beta, T* y,
                                                     float* mean_o, float* rstd_o, int64_t M, int D, float eps,
                                                     int act, int rms) {
    const int lane = threadIdx.x & 63;
    const int nch = D >> 2;
    // grid-stride over rows (the grid is what fits on the chip at once): the next row is requested as soon as the current
    // one is unpacked, so its HBM latency runs under this row's two reductions, the output arithmetic and the stores
    const int64_t stride = (int64_t)gridDim.x * 4;
    int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    Raw4<T> raw[CPL];
    if (row < M) {
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int ch = lane + c * 64;
            if (ch < nch) raw[c].ld(x + row * D + ch * 4);
        }
    }
    for (; row < M; row += stride) {
        float v[CPL][4];
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int ch = lane + c * 64;
            if (ch < nch) {
                raw[c].get(v[c]);
                sum += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
            } else {
                v[c][0] = v[c][1] = v[c][2] = v[c][3] = 0.f;
            }
        }
        const int64_t nrow = row + stride;
        if (nrow < M) {
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const int ch = lane + c * 64;
                if (ch < nch) raw[c].ld(x + nrow * D + ch * 4);
            }
        }
        const float mean = rms ? 0.f : wave_sum(sum) / (float)D;
        float sq = 0.f;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int ch = lane + c * 64;
            if (ch < nch) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float dlt = v[c][t] - mean;
                    sq += dlt * dlt;
                }
            }
        }
        const float var = wave_sum(sq) / (float)D;
        const float rstd = 1.0f / sqrtf(var + eps);
        if (lane == 0) {
            if (mean_o) mean_o[row] = mean;
            if (rstd_o) rstd_o[row] = rstd;
        }
        T* yr = y + row * D;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int ch = lane + c * 64;
            if (ch < nch) {
                float g[4], b[4] = {0.f, 0.f, 0.f, 0.f}, o[4];
                ld4<float>(gamma + ch * 4, g);
                if (beta) ld4<float>(beta + ch * 4, b);
#pragma unroll
                for (int t = 0; t < 4; ++t) o[t] = act_fwd((v[c][t] - mean) * rstd * g[t] + b[t], act);
                st4<T>(yr + ch * 4, o);
            }
        }
    }
}

// Row-PAIR form of the forward for bf16 rows of D % 256 == 0 elements (plain LayerNorm: no fused activation, not RMS): a
// wave owns two consecutive rows = NP = D / 256 pieces of 16 B per lane (one contiguous stretch of 2 D elements), so every
// global access is 16 B per lane instead of 8 (the one-row form's 4-element chunks: 4.2 TB/s at 147712 x 768) and the four
// wave reductions of the pair cost what the one-row form spends on two rows.  Piece p = lane + 64 c of the pair: row p / (D / 8),
// elements 8 (p % (D / 8)) .. + 7.
template <int NP>
__global__ __launch_bounds__(256) void ln_fwd_pair_kernel(const bf16_t* x, const float* gamma, const float* beta, bf16_t* y,
                                                          float* mean_o, float* rstd_o, int64_t M, float eps) {
    constexpr int D = NP * 256, PPR = D / 8;
    const int lane = threadIdx.x & 63;
    const int64_t npair = (M + 1) >> 1;
    const int64_t stride = (int64_t)gridDim.x * 4;
    int64_t pr = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    int sel[NP], col[NP];   // row of the pair (0 / 1) and first element of each of the lane's pieces
#pragma unroll
    for (int c = 0; c < NP; ++c) {
        const int p = lane + 64 * c;
        sel[c] = p >= PPR;
        col[c] = (p - sel[c] * PPR) * 8;
    }
    u32x4 raw[NP];
    auto load = [&](int64_t pair) {
        const bf16_t* base = x + pair * 2 * D;
        const bool two = pair * 2 + 1 < M;
#pragma unroll
        for (int c = 0; c < NP; ++c)
            raw[c] = (two || !sel[c]) ? *(const u32x4*)(base + (lane + 64 * c) * 8) : (u32x4){0u, 0u, 0u, 0u};
    };
    if (pr < npair) load(pr);
    for (; pr < npair; pr += stride) {
        float v[NP][8];
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int c = 0; c < NP; ++c) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v[c][2 * q] = __uint_as_float(raw[c][q] << 16);
                v[c][2 * q + 1] = __uint_as_float(raw[c][q] & 0xffff0000u);
                t += v[c][2 * q] + v[c][2 * q + 1];
            }
            if (sel[c]) s1 += t; else s0 += t;
        }
        const int64_t cur = pr;
        if (pr + stride < npair) load(pr + stride);
        const float m0 = wave_sum(s0) / (float)D, m1 = wave_sum(s1) / (float)D;
        float q0 = 0.f, q1 = 0.f;
#pragma unroll
        for (int c = 0; c < NP; ++c) {
            const float mu = sel[c] ? m1 : m0;
            float t = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float dlt = v[c][e] - mu; t += dlt * dlt; }
            if (sel[c]) q1 += t; else q0 += t;
        }
        const float r0 = 1.0f / sqrtf(wave_sum(q0) / (float)D + eps), r1 = 1.0f / sqrtf(wave_sum(q1) / (float)D + eps);
        const bool two = cur * 2 + 1 < M;
        if (lane == 0) {
            if (mean_o) { mean_o[cur * 2] = m0; if (two) mean_o[cur * 2 + 1] = m1; }
            if (rstd_o) { rstd_o[cur * 2] = r0; if (two) rstd_o[cur * 2 + 1] = r1; }
        }
        bf16_t* yb = y + cur * 2 * D;
#pragma unroll
        for (int c = 0; c < NP; ++c) {
            if (!two && sel[c]) continue;
            const float mu = sel[c] ? m1 : m0, rs = sel[c] ? r1 : r0;
            const f32x4 g0 = *(const f32x4*)(gamma + col[c]), g1 = *(const f32x4*)(gamma + col[c] + 4);
            f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
            if (beta) { b0 = *(const f32x4*)(beta + col[c]); b1 = *(const f32x4*)(beta + col[c] + 4); }
            float o[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = (v[c][e] - mu) * rs * g0[e] + b0[e];
                o[4 + e] = (v[c][4 + e] - mu) * rs * g1[e] + b1[e];
            }
            *(u32x4*)(yb + (lane + 64 * c) * 8) = (u32x4){pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7])};
        }
    }
}

template <typename T, int CPL, bool ACT>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* dy, const T* x, const float* gamma, const float* beta,
                                                     const float* mean_i, const float* rstd_i, T* dx, const T* dx_add,
                                                     float* part, int64_t M, int D, int act, int rms, T* dx_drop,
                                                     DropState drop) {
    if (dx_drop) drop_resolve(drop);
    extern __shared__ float red[];  // [4 waves][2][D]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nch = D >> 2;
    // ACT (LayerNorm + activation fused in the forward: the vqa_head) is the only case that needs beta here; the plain
    // instantiation stays at 4 waves / SIMD with the next row's raw registers added
    float dg[CPL][4], db[CPL][4], g[CPL][4], bt[ACT ? CPL : 1][4];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        const int ch = lane + c * 64;
#pragma unroll
        for (int t = 0; t < 4; ++t) { dg[c][t] = 0.f; db[c][t] = 0.f; g[c][t] = 0.f; if (ACT) bt[c][t] = 0.f; }
        if (ch < nch) {
            ld4<float>(gamma + ch * 4, g[c]);
            if (ACT && beta) ld4<float>(beta + ch * 4, bt[c]);
        }
    }
    // A wave walks its rows one after the other.  The NEXT row's x / dy (raw, packed) and statistics are requested as soon
    // as the current row's registers have been unpacked into xh / dxh, so the two wave reductions, the output arithmetic
    // and the stores of a row run under the next row's HBM latency.
    const int64_t stride = (int64_t)gridDim.x * 4;
    int64_t row = (int64_t)blockIdx.x * 4 + wave;
    Raw4<T> xr[CPL], dr[CPL];
    float mean_n = 0.f, rstd_n = 0.f;
    if (row < M) {
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int ch = lane + c * 64;
            if (ch < nch) { xr[c].ld(x + row * D + ch * 4); dr[c].ld(dy + row * D + ch * 4); }
        }
        mean_n = rms ? 0.f : mean_i[row];
        rstd_n = rstd_i[row];
    }
    for (; row < M; row += stride) {
        const float mean = mean_n;
        const float rstd = rstd_n;
        float xh[CPL][4], dxh[CPL][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int ch = lane + c * 64;
            if (ch < nch) {
                float xv[4], dv[4];
                xr[c].get(xv);
                dr[c].get(dv);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float h = (xv[t] - mean) * rstd;
                    float d = dv[t];
                    if (ACT) d *= act_bwd(h * g[c][t] + bt[c][t], act);
                    dg[c][t] += d * h;
                    db[c][t] += d;
                    const float dh = d * g[c][t];
                    xh[c][t] = h;
                    dxh[c][t] = dh;
                    s1 += dh;
                    s2 += dh * h;
                }
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) { xh[c][t] = 0.f; dxh[c][t] = 0.f; }
            }
        }
        const int64_t nrow = row + stride;
        if (nrow < M) {
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const int ch = lane + c * 64;
                if (ch < nch) { xr[c].ld(x + nrow * D + ch * 4); dr[c].ld(dy + nrow * D + ch * 4); }
            }
            mean_n = rms ? 0.f : mean_i[nrow];
            rstd_n = rstd_i[nrow];
        }
        const float c1 = rms ? 0.f : wave_sum(s1) / (float)D;
        const float c2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int ch = lane + c * 64;
            if (ch < nch) {
                float o[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) o[t] = rstd * (dxh[c][t] - c1 - xh[c][t] * c2);
                if (dx_drop) {
                    float od[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) od[t] = o[t];
                    drop_apply4(drop, (uint64_t)(row * D + ch * 4), od);  // D % 4 == 0: ld = D
                    st4<T>(dx_drop + row * D + ch * 4, od);
                }
                if (dx_add) {
                    float e[4];
                    ld4<T>(dx_add + row * D + ch * 4, e);
#pragma unroll
                    for (int t = 0; t < 4; ++t) o[t] += e[t];
                }
                st4<T>(dx + row * D + ch * 4, o);
            }
        }
    }
    // combine the 4 waves, one partial row per workgroup
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        const int ch = lane + c * 64;
        if (ch < nch) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                red[(wave * 2 + 0) * D + ch * 4 + t] = dg[c][t];
                red[(wave * 2 + 1) * D + ch * 4 + t] = db[c][t];
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * D; i += 256) {
        const int which = i / D, col = i - which * D;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) s += red[(w * 2 + which) * D + col];
        part[((int64_t)blockIdx.x * 2 + which) * D + col] = s;
    }
}

// fold the per-workgroup partial rows into the fp32 gradients: grid (column blocks of 32, row chunks of 64 partials);
// 32 columns x 8 row-lanes per workgroup, one fp32 atomic per column per workgroup
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* part, float* dgamma, float* dbeta, int nblk,
                                                            int D) {
    __shared__ float red[8][33];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + cl;  // index into [2][D]
    const int b0 = blockIdx.y * 64;
    const int b1 = b0 + 64 < nblk ? b0 + 64 : nblk;
    float s = 0.f;
    if (i < 2 * D) {
        const int which = i / D, col = i - which * D;
        for (int b = b0 + rl; b < b1; b += 8) s += part[((int64_t)b * 2 + which) * D + col];
    }
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && i < 2 * D) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) t += red[r][cl];
        const int which = i / D, col = i - which * D;
        if (which == 0) atomicAdd(dgamma + col, t);
        else if (dbeta) atomicAdd(dbeta + col, t);
    }
}

// workgroups that can be resident at once (every wave walks its rows in a grid-stride loop: a workgroup that has to wait for
// a free slot would start a second round with a full share of the rows)
template <typename K> int resident_blocks(K kernel, size_t lds) {
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, lds) != hipSuccess || per_cu <= 0) return 1 << 30;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 1 << 30;
    return per_cu * prop.multiProcessorCount;
}
template <typename T, int CPL>
int launch_fwd(const void* x, const float* g, const float* b, void* y, float* mean, float* rstd, int64_t M, int D,
               float eps, int act, int rms, hipStream_t s) {
    static const int cap = resident_blocks(ln_fwd_kernel<T, CPL>, 0);
    const int64_t want = cdiv(M, 4);
    hipLaunchKernelGGL((ln_fwd_kernel<T, CPL>), dim3((unsigned)(want < cap ? want : cap)), dim3(256), 0, s, (const T*)x, g, b,
                       (T*)y, mean, rstd, M, D, eps, act, rms);
    return hip_launch_status();
}
template <typename T, int CPL>
int launch_bwd(const void* dy, const void* x, const float* g, const float* b, const float* mean, const float* rstd,
               void* dx, const void* dx_add, float* part, int* nblk_io, int64_t M, int D, int act, int rms, hipStream_t s,
               void* dx_drop = nullptr, DropState drop = DropState{}) {
    const size_t lds = (size_t)8 * D * sizeof(float);
    static int cap_act = 0, cap_plain = 0, cap_d = -1;   // per instantiation; the LDS size follows D
    if (cap_d != D) {
        cap_act = resident_blocks(ln_bwd_kernel<T, CPL, true>, lds);
        cap_plain = resident_blocks(ln_bwd_kernel<T, CPL, false>, lds);
        cap_d = D;
    }
    const int cap = act != M3AE_ACT_NONE ? cap_act : cap_plain;
    const int nblk = *nblk_io < cap ? *nblk_io : cap;
    *nblk_io = nblk;
    if (act != M3AE_ACT_NONE)
        hipLaunchKernelGGL((ln_bwd_kernel<T, CPL, true>), dim3((unsigned)nblk), dim3(256), lds, s,
                           (const T*)dy, (const T*)x, g, b, mean, rstd, (T*)dx, (const T*)dx_add, part, M, D, act, rms,
                           (T*)dx_drop, drop);
    else
        hipLaunchKernelGGL((ln_bwd_kernel<T, CPL, false>), dim3((unsigned)nblk), dim3(256), lds, s,
                           (const T*)dy, (const T*)x, g, b, mean, rstd, (T*)dx, (const T*)dx_add, part, M, D, act, rms,
                           (T*)dx_drop, drop);
    return hip_launch_status();
}

#define DISPATCH_CPL(FN, T, ...)                                                  \
    do {                                                                          \
        const int cpl = (int)cdiv(D / 4, 64);                                     \
        if (cpl <= 1) return FN<T, 1>(__VA_ARGS__);                               \
        if (cpl <= 2) return FN<T, 2>(__VA_ARGS__);                               \
        if (cpl <= 3) return FN<T, 3>(__VA_ARGS__);                               \
        if (cpl <= 4) return FN<T, 4>(__VA_ARGS__);                               \
        if (cpl <= 6) return FN<T, 6>(__VA_ARGS__);                               \
        if (cpl <= 8) return FN<T, 8>(__VA_ARGS__);                               \
        if (cpl <= 16) return FN<T, 16>(__VA_ARGS__);                             \
        return M3AE_ERR_UNSUPPORTED;                                              \
    } while (0)

}  // namespace

template <int NP>
int launch_fwd_pair(const void* x, const float* g, const float* b, void* y, float* mean, float* rstd, int64_t M, float eps,
                    hipStream_t s) {
    static const int cap = resident_blocks(ln_fwd_pair_kernel<NP>, 0);
    const int64_t want = cdiv((M + 1) / 2, 4);
    hipLaunchKernelGGL((ln_fwd_pair_kernel<NP>), dim3((unsigned)(want < cap ? want : cap)), dim3(256), 0, s, (const bf16_t*)x, g, b,
                       (bf16_t*)y, mean, rstd, M, eps);
    return hip_launch_status();
}

extern "C" int m3ae_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                                  float* rstd, int64_t M, int64_t D, float eps, int dtype, int act, int rms,
                                  void* stream) {
    if (!x || !gamma || !y || M <= 0 || D <= 0) return M3AE_ERR_ARG;
    if (D % 4 != 0 || D > 4096) return M3AE_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const bool pair_off = (rms & 2) != 0;   // rms bit 1: diagnostic A / B of the row-pair kernel (tools/ln_bench.py)
    rms &= 1;
    if (dtype == M3AE_BF16 && act == M3AE_ACT_NONE && !rms && M >= 1024 && ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0 &&
        ((((uintptr_t)gamma) | ((uintptr_t)beta)) & 15) == 0 && !pair_off) {
        if (D == 512) return launch_fwd_pair<2>(x, gamma, beta, y, mean, rstd, M, eps, s);
        if (D == 768) return launch_fwd_pair<3>(x, gamma, beta, y, mean, rstd, M, eps, s);
        if (D == 1024) return launch_fwd_pair<4>(x, gamma, beta, y, mean, rstd, M, eps, s);
    }
    if (dtype == M3AE_F32) DISPATCH_CPL(launch_fwd, float, x, gamma, beta, y, mean, rstd, M, (int)D, eps, act, rms, s);
    if (dtype == M3AE_BF16) DISPATCH_CPL(launch_fwd, bf16_t, x, gamma, beta, y, mean, rstd, M, (int)D, eps, act, rms, s);
    return M3AE_ERR_UNSUPPORTED;
}

extern "C" int64_t m3ae_layernorm_bwd_blocks(int64_t M) {
    // 1024 workgroups x 4 waves = 4 waves per SIMD on 256 CUs (the kernel's register footprint allows exactly that): a
    // wave works through its rows strictly one after the other, so resident waves are what hides the HBM latency
    int64_t n = cdiv(M, 4);
    return n < 1024 ? n : 1024;
}

extern "C" int m3ae_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* beta,
                                  const float* mean, const float* rstd, void* dx, const void* dx_add, float* dgamma,
                                  float* dbeta, float* workspace, int64_t M, int64_t D, int dtype, int act, int rms,
                                  void* stream) {
    if (!dy || !x || !gamma || !rstd || !dx || !workspace || M <= 0) return M3AE_ERR_ARG;
    if (D % 4 != 0 || D > 2048) return M3AE_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    int nblk = (int)m3ae_layernorm_bwd_blocks(M);   // upper bound (the workspace is sized for it); launch_bwd may lower it
    int rc = M3AE_ERR_UNSUPPORTED;
    auto run = [&]() -> int {
        if (dtype == M3AE_F32)
            DISPATCH_CPL(launch_bwd, float, dy, x, gamma, beta, mean, rstd, dx, dx_add, workspace, &nblk, M, (int)D, act, rms, s);
        if (dtype == M3AE_BF16)
            DISPATCH_CPL(launch_bwd, bf16_t, dy, x, gamma, beta, mean, rstd, dx, dx_add, workspace, &nblk, M, (int)D, act, rms, s);
        return M3AE_ERR_UNSUPPORTED;
    };
    rc = run();
    if (rc) return rc;
    if (!dgamma) return 0;
    hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((unsigned)cdiv(2 * D, 32), (unsigned)cdiv(nblk, 64)), dim3(256), 0, s, workspace, dgamma,
                       dbeta, nblk, (int)D);
    return hip_launch_status();
}

extern "C" int m3ae_layernorm_bwd_drop(const void* dy, const void* x, const float* gamma, const float* beta,
                                       const float* mean, const float* rstd, void* dx, void* dx_drop, float dropout_p,
                                       uint64_t dropout_seed, const void* dropout_salt, float* dgamma, float* dbeta,
                                       float* workspace, int64_t M, int64_t D, int dtype, void* stream) {
    if (!dy || !x || !gamma || !rstd || !dx || !dx_drop || !workspace || M <= 0) return M3AE_ERR_ARG;
    if (D % 4 != 0 || D > 2048) return M3AE_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    int nblk = (int)m3ae_layernorm_bwd_blocks(M);
    const DropState drop = make_drop(dropout_p, dropout_seed, dropout_salt);
    auto run = [&]() -> int {
        if (dtype == M3AE_F32)
            DISPATCH_CPL(launch_bwd, float, dy, x, gamma, beta, mean, rstd, dx, nullptr, workspace, &nblk, M, (int)D, 0, 0, s, dx_drop, drop);
        if (dtype == M3AE_BF16)
            DISPATCH_CPL(launch_bwd, bf16_t, dy, x, gamma, beta, mean, rstd, dx, nullptr, workspace, &nblk, M, (int)D, 0, 0, s, dx_drop, drop);
        return M3AE_ERR_UNSUPPORTED;
    };
    int rc = run();
    if (rc) return rc;
    if (!dgamma) return 0;
    hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((unsigned)cdiv(2 * D, 32), (unsigned)cdiv(nblk, 64)), dim3(256), 0, s,
                       workspace, dgamma, dbeta, nblk, (int)D);
    return hip_launch_status();
}
