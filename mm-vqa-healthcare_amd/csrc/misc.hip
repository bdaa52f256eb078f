// Memory-bound support kernels of the hot path: embedding gather / scatter, patch extraction, token assembly,
// column sums, losses, the fused AdamW update and dtype / layout conversions.  All are HBM-streaming kernels:
// coalesced 4-16 B per lane, grid-stride, fp32 arithmetic.
#include "common.h"

namespace {

template <typename T> DEVINL float ldf(const void* p, int64_t i) { return Elem<T>::ld((const T*)p + i); }
template <typename T> DEVINL void stf(void* p, int64_t i, float v) { Elem<T>::st((T*)p + i, v); }

#define DT_SWITCH(dtype, CALL)                        \
    do {                                              \
        if ((dtype) == M3AE_F32) { using T = float; CALL; }        \
        else if ((dtype) == M3AE_BF16) { using T = bf16_t; CALL; } \
        else return M3AE_ERR_UNSUPPORTED;             \
    } while (0)

constexpr int EW_BLOCK = 256;
inline unsigned ew_grid(int64_t n) {
    int64_t g = cdiv(n, EW_BLOCK);
    return (unsigned)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

// ---- column sum ------------------------------------------------------------------------------------------
// workgroup = 32 column-groups (8 columns = 16 B bf16 / 2 x 16 B fp32 per lane per row) x 8 row-lanes; grid
// (cdiv(N, 256), row_chunks); LDS-combine the row-lanes, one fp32 atomic per column per workgroup.
template <typename T> DEVINL void ld8(const T* p, float* x);
template <> DEVINL void ld8<float>(const float* p, float* x) {
    const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
    x[0] = a[0]; x[1] = a[1]; x[2] = a[2]; x[3] = a[3]; x[4] = b[0]; x[5] = b[1]; x[6] = b[2]; x[7] = b[3];
}
template <> DEVINL void ld8<bf16_t>(const bf16_t* p, float* x) {
    const u32x4 v = *(const u32x4*)p;
#pragma unroll
    for (int t = 0; t < 4; ++t) { x[2 * t] = __uint_as_float(v[t] << 16); x[2 * t + 1] = __uint_as_float(v[t] & 0xffff0000u); }
}
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void colsum_kernel(const T* x, float* out, int64_t M, int64_t N, int64_t ldx,
                                                     int64_t rows_per) {
    __shared__ float red[8][32][9];
    const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int64_t n0 = (int64_t)blockIdx.x * 256 + cg * 8;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per;
    int64_t r1 = r0 + rows_per;
    if (r1 > M) r1 = M;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (n0 < N) {
        int64_t r = r0 + rl;
        if (VEC) {   // four rows in flight per lane (the loop-carried sum serialised one 16-B load per iteration: 2.1 TB/s)
            for (; r + 24 < r1; r += 32) {
                float v[4][8];
#pragma unroll
                for (int u = 0; u < 4; ++u) ld8<T>(x + (r + 8 * u) * ldx + n0, v[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int t = 0; t < 8; ++t) acc[t] += v[u][t];
            }
        }
        for (; r < r1; r += 8) {
            if (VEC) {
                float v[8];
                ld8<T>(x + r * ldx + n0, v);
#pragma unroll
                for (int t = 0; t < 8; ++t) acc[t] += v[t];
            } else {
#pragma unroll
                for (int t = 0; t < 8; ++t)
                    if (n0 + t < N) acc[t] += Elem<T>::ld(x + r * ldx + n0 + t);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) red[rl][cg][t] = acc[t];
    __syncthreads();
    if (rl == 0 && n0 < N) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) s += red[r][cg][t];
            if (n0 + t < N) atomicAdd(out + n0 + t, s);
        }
    }
}

// Whole-row form for contiguous bf16 / fp32 matrices with N % 8 == 0 and N / 8 <= 512: a workgroup of P * R threads (P = N / 8
// 16-byte pieces per row, R rows per pass) reads R WHOLE rows per pass -- one contiguous stretch of memory -- and every
// thread keeps its column group; four passes in flight per thread.  (The 256-column form above reads 512 B of each row per
// workgroup and ran at 2.5 TB/s on 147712 x 768; profiles/r02_colsum.log.)
template <typename T>
__global__ __launch_bounds__(512) void colsum_rows_kernel(const T* x, float* out, int64_t M, int P, int R, int64_t rows_per) {
    extern __shared__ float cred[];   // [R][P * 8]
    const int t = threadIdx.x;
    const int pc = t % P, rl = t / P;
    const int64_t N = (int64_t)P * 8;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per;
    int64_t r1 = r0 + rows_per;
    if (r1 > M) r1 = M;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int64_t r = r0 + rl;
    const T* xp = x + pc * 8;
    for (; r + 3 * R < r1; r += 4 * R) {
        float v[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) ld8<T>(xp + (r + (int64_t)u * R) * N, v[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += v[u][k];
    }
    for (; r < r1; r += R) {
        float v[8];
        ld8<T>(xp + r * N, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += v[k];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) cred[(int64_t)rl * N + pc * 8 + k] = acc[k];
    __syncthreads();
    for (int c = t; c < N; c += blockDim.x) {
        float sum = 0.f;
        for (int q = 0; q < R; ++q) sum += cred[(int64_t)q * N + c];
        atomicAdd(out + c, sum);
    }
}

// ---- RoBERTa embeddings ------------------------------------------------------------------------------------
// one workgroup per sample: position ids by a serial scan over S (S <= 512), then D-wide gathers
template <typename T>
__global__ void roberta_embed_fwd_kernel(const int64_t* ids, const float* word, const float* pos, const float* type,
                                         T* out, int64_t S, int64_t D, int64_t pad_id) {
    extern __shared__ int pos_ids[];
    const int64_t b = blockIdx.x;
    if (threadIdx.x == 0) {
        int run = 0;
        for (int64_t s = 0; s < S; ++s) {
            const int ne = ids[b * S + s] != pad_id;
            run += ne;
            pos_ids[s] = run * ne + (int)pad_id;
        }
    }
    __syncthreads();
    for (int64_t i = threadIdx.x; i < S * D; i += blockDim.x) {
        const int64_t s = i / D, d = i - s * D;
        const int64_t id = ids[b * S + s];
        const float v = word[id * D + d] + type[d] + pos[(int64_t)pos_ids[s] * D + d];
        Elem<T>::st(out + (b * S + s) * D + d, v);
    }
}
template <typename T>
__global__ void roberta_embed_bwd_kernel(const int64_t* ids, const T* d_out, float* d_word, float* d_pos, int64_t S,
                                         int64_t D, int64_t pad_id) {
    extern __shared__ int pos_ids[];
    const int64_t b = blockIdx.x;
    if (threadIdx.x == 0) {
        int run = 0;
        for (int64_t s = 0; s < S; ++s) {
            const int ne = ids[b * S + s] != pad_id;
            run += ne;
            pos_ids[s] = run * ne + (int)pad_id;
        }
    }
    __syncthreads();
    for (int64_t i = threadIdx.x; i < S * D; i += blockDim.x) {
        const int64_t s = i / D, d = i - s * D;
        const int64_t id = ids[b * S + s];
        const float g = Elem<T>::ld(d_out + (b * S + s) * D + d);
        atomicAdd(d_word + id * D + d, g);
        atomicAdd(d_pos + (int64_t)pos_ids[s] * D + d, g);
    }
}

// ---- ViT patches -----------------------------------------------------------------------------------------
// out[(b * G + gy * g + gx)][c * P * P + py * P + px] = img[b][c][gy * P + py][gx * P + px]
template <typename T>
__global__ void patchify_kernel(const float* img, T* out, int64_t B, int64_t R, int64_t P) {
    const int64_t g = R / P, K = 3 * P * P, total = B * g * g * K;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t col = i % K, tok = i / K;
        const int64_t px = col % P, py = (col / P) % P, c = col / (P * P);
        const int64_t gx = tok % g, gy = (tok / g) % g, b = tok / (g * g);
        Elem<T>::st(out + i, img[((b * 3 + c) * R + gy * P + py) * R + gx * P + px]);
    }
}
// P % 8 == 0, bf16 output: one thread moves 8 consecutive px (32 B read, 16 B written); the index is split with 32-bit
// arithmetic (the scalar form above spends its time in seven 64-bit divisions per element: 1.3 TB/s)
__global__ void patchify8_bf16_kernel(const float* __restrict__ img, bf16_t* __restrict__ out, uint32_t n8, uint32_t R,
                                      uint32_t P, uint32_t g) {
    const uint32_t K8 = 3 * P * P / 8, P8 = P / 8;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += gridDim.x * blockDim.x) {
        const uint32_t col8 = i % K8, tok = i / K8;
        const uint32_t px8 = col8 % P8, py = (col8 / P8) % P, c = col8 / (P8 * P);
        const uint32_t gx = tok % g, gy = (tok / g) % g, b = tok / (g * g);
        const float* src = img + ((int64_t)(b * 3 + c) * R + gy * P + py) * R + gx * P + px8 * 8;
        const f32x4 v0 = *(const f32x4*)src, v1 = *(const f32x4*)(src + 4);
        *(u32x4*)(out + (int64_t)i * 8) = (u32x4){pack2bf(v0[0], v0[1]), pack2bf(v0[2], v0[3]), pack2bf(v1[0], v1[1]),
                                                  pack2bf(v1[2], v1[3])};
    }
}
// NHWC uint8 -> NCHW fp32, (u / 255 - mean) / std: one thread per output pixel-channel, reads coalesced over x
__global__ void image_normalize_u8_kernel(const uint8_t* in, float* out, int64_t B, int64_t H, int64_t W, float m0,
                                          float m1, float m2, float s0, float s1, float s2) {
    const int64_t total = B * 3 * H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t x = i % W, y = (i / W) % H, c = (i / (W * H)) % 3, b = i / (3 * W * H);
        const float u = (float)in[((b * H + y) * W + x) * 3 + c];
        const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
        out[i] = (u / 255.0f - mean) / sd;
    }
}
// out[b][0] = cls + pos[0];  out[b][1 + i] = patch[b][i] + pos[1 + i]
template <typename T>
__global__ void vit_tokens_fwd_kernel(const T* patch, const float* cls, const float* pos, T* out, int64_t B, int64_t G,
                                      int64_t D) {
    const int64_t L = G + 1, total = B * L * D;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t d = i % D, l = (i / D) % L, b = i / (D * L);
        const float base = l == 0 ? cls[d] : Elem<T>::ld(patch + (b * G + l - 1) * D + d);
        Elem<T>::st(out + i, base + pos[l * D + d]);
    }
}
// D % 8 == 0, bf16: 8 consecutive d per thread, 32-bit index split
__global__ void vit_tokens_fwd8_bf16_kernel(const bf16_t* __restrict__ patch, const float* __restrict__ cls,
                                            const float* __restrict__ pos, bf16_t* __restrict__ out, uint32_t n8, uint32_t G,
                                            uint32_t D8) {
    const uint32_t L = G + 1;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += gridDim.x * blockDim.x) {
        const uint32_t d8 = i % D8, l = (i / D8) % L, b = i / (D8 * L);
        float x[8];
        if (l == 0) {
            const f32x4 c0 = *(const f32x4*)(cls + d8 * 8), c1 = *(const f32x4*)(cls + d8 * 8 + 4);
            x[0] = c0[0]; x[1] = c0[1]; x[2] = c0[2]; x[3] = c0[3]; x[4] = c1[0]; x[5] = c1[1]; x[6] = c1[2]; x[7] = c1[3];
        } else {
            const u32x4 v = *(const u32x4*)(patch + ((int64_t)b * G + l - 1) * (D8 * 8) + d8 * 8);
#pragma unroll
            for (int t = 0; t < 4; ++t) { x[2 * t] = __uint_as_float(v[t] << 16); x[2 * t + 1] = __uint_as_float(v[t] & 0xffff0000u); }
        }
        const float* pr = pos + (int64_t)l * (D8 * 8) + d8 * 8;
        const f32x4 p0 = *(const f32x4*)pr, p1 = *(const f32x4*)(pr + 4);
        *(u32x4*)(out + (int64_t)i * 8) = (u32x4){pack2bf(x[0] + p0[0], x[1] + p0[1]), pack2bf(x[2] + p0[2], x[3] + p0[3]),
                                                  pack2bf(x[4] + p1[0], x[5] + p1[1]), pack2bf(x[6] + p1[2], x[7] + p1[3])};
    }
}
// d_patch = d_out[:, 1:];  d_pos[l] += sum_b d_out[b][l];  d_cls += sum_b d_out[b][0]
template <typename T>
__global__ void vit_tokens_bwd_kernel(const T* d_out, T* d_patch, float* d_cls, float* d_pos, int64_t B, int64_t G,
                                      int64_t D) {
    const int64_t L = G + 1, total = L * D;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t d = i % D, l = i / D;
        float acc = 0.f;
        for (int64_t b = 0; b < B; ++b) {
            const T raw = d_out[(b * L + l) * D + d];
            if (l > 0) d_patch[(b * G + l - 1) * D + d] = raw;
            acc += Elem<T>::ld(&raw);
        }
        d_pos[i] += acc;
        if (l == 0) d_cls[d] += acc;
    }
}

// ---- losses -----------------------------------------------------------------------------------------------
template <typename T>
__global__ void bce_kernel(const T* x, const float* z, float* loss, T* dx, int64_t n, float inv_n, float C,
                           float grad_scale) {
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float xi = Elem<T>::ld(x + i), zi = z[i];
        acc += fmaxf(xi, 0.f) - xi * zi + log1pf(expf(-fabsf(xi)));
        const float sig = 1.0f / (1.0f + expf(-xi));
        if (dx) Elem<T>::st(dx + i, (sig - zi) * inv_n * C * grad_scale);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) atomicAdd(loss, acc * inv_n * C);
}

// cross entropy: ws[0] = number of rows with label != -100 ; one workgroup per row
__global__ void xent_count_kernel(const int64_t* labels, float* ws, int64_t rows) {
    float c = 0.f;
    for (int64_t i = threadIdx.x; i < rows; i += blockDim.x) c += labels[i] != -100 ? 1.f : 0.f;
    c = wave_sum(c);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) ws[0] = part[0] + part[1] + part[2] + part[3];
}
template <typename T>
__global__ __launch_bounds__(256) void xent_kernel(const T* x, const int64_t* labels, float* loss, T* dx, const float* ws,
                                                   int64_t C, int64_t ld, float grad_scale) {
    __shared__ float red[4];
    const int64_t row = blockIdx.x;
    const int64_t lab = labels[row];
    const T* xr = x + row * ld;
    T* dr = dx ? dx + row * ld : nullptr;
    if (lab == -100) {
        if (dr) for (int64_t j = threadIdx.x; j < C; j += 256) Elem<T>::st(dr + j, 0.f);
        return;
    }
    float mx = -INFINITY;
    for (int64_t j = threadIdx.x; j < C; j += 256) mx = fmaxf(mx, Elem<T>::ld(xr + j));
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int64_t j = threadIdx.x; j < C; j += 256) sum += expf(Elem<T>::ld(xr + j) - mx);
    sum = wave_sum(sum);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    sum = red[0] + red[1] + red[2] + red[3];
    const float inv_valid = 1.0f / ws[0];
    const float lse = mx + logf(sum);
    if (threadIdx.x == 0) atomicAdd(loss, (lse - Elem<T>::ld(xr + lab)) * inv_valid);
    if (dr) {
        const float gs = grad_scale * inv_valid;
        for (int64_t j = threadIdx.x; j < C; j += 256) {
            const float p = expf(Elem<T>::ld(xr + j) - lse);
            Elem<T>::st(dr + j, (p - (j == lab ? 1.f : 0.f)) * gs);
        }
    }
}

// ---- AdamW (transformers 4.6.0 semantics) -----------------------------------------------------------------
__global__ void adamw_kernel(float* p, const float* g, float* m, float* v, bf16_t* shadow, int64_t n4, float lr,
                             float b1, float b2, float eps, float wd, float step_size, float grad_scale,
                             const float* hyper_dev) {
    if (hyper_dev) { lr = hyper_dev[0]; step_size = hyper_dev[1]; }   // a captured step: this step's values come from device memory
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 pp = ((f32x4*)p)[i], mm = ((f32x4*)m)[i], vv = ((f32x4*)v)[i];
        const f32x4 gg = ((const f32x4*)g)[i];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float gt = gg[t] * grad_scale;
            mm[t] = mm[t] * b1 + gt * (1.0f - b1);
            vv[t] = vv[t] * b2 + gt * gt * (1.0f - b2);
            const float denom = sqrtf(vv[t]) + eps;
            pp[t] = pp[t] - step_size * (mm[t] / denom);
            if (wd > 0.f) pp[t] = pp[t] - lr * wd * pp[t];
        }
        ((f32x4*)p)[i] = pp; ((f32x4*)m)[i] = mm; ((f32x4*)v)[i] = vv;
        if (shadow) ((u32x2*)shadow)[i] = (u32x2){pack2bf(pp[0], pp[1]), pack2bf(pp[2], pp[3])};
    }
}

// ---- cast / transpose -------------------------------------------------------------------------------------
__global__ void cast_transpose_kernel(const float* in, bf16_t* out, bf16_t* out_t, int64_t R, int64_t C) {
    __shared__ float tile[32][33];
    const int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int64_t r = r0 + j, c = c0 + tx;
        float v = 0.f;
        if (r < R && c < C) {
            v = in[r * C + c];
            if (out) out[r * C + c] = f2bf(v);
        }
        tile[j][tx] = v;
    }
    __syncthreads();
    if (out_t) {
        for (int j = ty; j < 32; j += 8) {
            const int64_t c = c0 + j, r = r0 + tx;
            if (r < R && c < C) out_t[c * R + r] = f2bf(tile[tx][j]);
        }
    }
}
// table-driven form: one launch transposes every GEMM weight unit (desc = {src, dst_t, R, C, first_tile}; tiles are
// 32 x 32, the unit of a workgroup is found by binary search over first_tile)
struct TransposeJob { const float* in; bf16_t* out_t; int64_t R, C; int64_t first_tile; };
__global__ void cast_transpose_batched_kernel(const TransposeJob* jobs, int njobs) {
    __shared__ float tile[32][33];
    const int64_t tid = blockIdx.x;
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_tile <= tid) lo = mid; else hi = mid - 1;
    }
    const TransposeJob j = jobs[lo];
    const int64_t local = tid - j.first_tile;
    const int64_t tiles_c = (j.C + 31) / 32;
    const int64_t c0 = (local % tiles_c) * 32, r0 = (local / tiles_c) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = ty; k < 32; k += 8) {
        const int64_t r = r0 + k, c = c0 + tx;
        tile[k][tx] = (r < j.R && c < j.C) ? j.in[r * j.C + c] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int64_t c = c0 + k, r = r0 + tx;
        if (r < j.R && c < j.C) j.out_t[c * j.R + r] = f2bf(tile[tx][k]);
    }
}

// the same from the bf16 shadows (what the optimizer kernel has just written): half the bytes read, 64 x 64 tiles, 16 B per lane
// on both sides for whole tiles of 8-aligned units (edge tiles / odd shapes: element-wise).  desc = {const bf16* in; bf16* out_t;
// R, C, first_tile} with first_tile counting 64 x 64 tiles.  (The 32 x 32 fp32-source form above: 586 us per step for the 300 M
// weight elements of M3AE-base, 1.8 GB moved at 3 TB/s; this form moves 1.2 GB.)
struct TransposeJob16 { const bf16_t* in; bf16_t* out_t; int64_t R, C; int64_t first_tile; };
__global__ __launch_bounds__(256) void transpose_bf16_batched_kernel(const TransposeJob16* jobs, int njobs) {
    __shared__ __attribute__((aligned(16))) bf16_t tile[64][66];   // 132-B pitch: 4-B aligned rows, column gathers spread over the banks
    const int64_t tid = blockIdx.x;
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_tile <= tid) lo = mid; else hi = mid - 1;
    }
    const TransposeJob16 j = jobs[lo];
    const int64_t local = tid - j.first_tile;
    const int64_t tiles_c = (j.C + 63) / 64;
    const int64_t c0 = (local % tiles_c) * 64, r0 = (local / tiles_c) * 64;
    const int t = threadIdx.x, q = t >> 3, ch = t & 7;
    const bool fast = ((j.R | j.C) & 7) == 0 && r0 + 64 <= j.R && c0 + 64 <= j.C &&
                      ((((uintptr_t)j.in) | ((uintptr_t)j.out_t)) & 15) == 0;
    if (fast) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int r = q + 32 * p;
            const u32x4 v = *(const u32x4*)(j.in + (r0 + r) * j.C + c0 + ch * 8);
            uint32_t* d = (uint32_t*)&tile[r][ch * 8];
            d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int c = q + 32 * p;
            u32x4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                v[i] = (uint32_t)tile[ch * 8 + 2 * i][c] | ((uint32_t)tile[ch * 8 + 2 * i + 1][c] << 16);
            *(u32x4*)(j.out_t + (c0 + c) * j.R + r0 + ch * 8) = v;
        }
        return;
    }
    for (int e = t; e < 64 * 64; e += 256) {
        const int r = e >> 6, c = e & 63;
        tile[r][c] = (r0 + r < j.R && c0 + c < j.C) ? j.in[(r0 + r) * j.C + c0 + c] : (bf16_t)0;
    }
    __syncthreads();
    for (int e = t; e < 64 * 64; e += 256) {
        const int c = e >> 6, r = e & 63;
        if (r0 + r < j.R && c0 + c < j.C) j.out_t[(c0 + c) * j.R + r0 + r] = tile[r][c];
    }
}

template <typename TI, typename TO>
__global__ void cast_kernel(const TI* in, TO* out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        Elem<TO>::st(out + i, Elem<TI>::ld(in + i));
}
template <typename T>
__global__ void add_kernel(const T* a, const T* b, T* out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        Elem<T>::st(out + i, Elem<T>::ld(a + i) + Elem<T>::ld(b + i));
}
// bf16, 8 elements (16 B) per thread and access: the gradient joins of the fusion layers (147712 x 768) run at the HBM rate
__global__ void add_bf16x8_kernel(const u32x4* a, const u32x4* b, u32x4* out, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const u32x4 x = a[i], y = b[i];
        u32x4 r;
#pragma unroll
        for (int t = 0; t < 4; ++t)
            r[t] = pack2bf(__uint_as_float(x[t] << 16) + __uint_as_float(y[t] << 16),
                           __uint_as_float(x[t] & 0xffff0000u) + __uint_as_float(y[t] & 0xffff0000u));
        out[i] = r;
    }
}
template <typename T>
__global__ void act_fwd_kernel(const T* x, T* y, int64_t n, int act) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        Elem<T>::st(y + i, act_fwd(Elem<T>::ld(x + i), act));
}
template <typename T>
__global__ void act_bwd_kernel(const T* dy, const T* x, T* dx, int64_t n, int act) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        Elem<T>::st(dx + i, Elem<T>::ld(dy + i) * act_bwd(Elem<T>::ld(x + i), act));
}
template <typename T>
__global__ void dropout_kernel(const T* x, T* out, uint8_t* keep, int64_t n, int64_t cols, int64_t ld, DropState ds) {
    drop_resolve(ds);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols;
        const bool k = drop_keep(ds, (uint64_t)(r * ld + (i - r * cols)));
        if (out) Elem<T>::st(out + i, k ? Elem<T>::ld(x + i) * ds.inv_keep : 0.f);
        if (keep) keep[i] = k ? 1 : 0;
    }
}

template <typename T>
__global__ void gather_rows_kernel(const T* in, const int64_t* idx, T* out, int64_t n_out, int64_t D) {
    const int64_t total = n_out * D;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / D, d = i - r * D;
        out[i] = in[idx[r] * D + d];
    }
}
template <typename T>
__global__ void scatter_add_rows_kernel(const T* d_out, const int64_t* idx, T* d_in, int64_t n_out, int64_t D) {
    // rows of idx are distinct per destination in the MIM use (a permutation slice), so plain read-modify-write
    const int64_t total = n_out * D;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / D, d = i - r * D;
        const int64_t o = idx[r] * D + d;
        Elem<T>::st(d_in + o, Elem<T>::ld(d_in + o) + Elem<T>::ld(d_out + i));
    }
}

}  // namespace

extern "C" int m3ae_abi_version(void) { return M3AE_ABI_VERSION; }
extern "C" void m3ae_desc_sizes(int64_t out3[3]) {
    out3[0] = (int64_t)sizeof(m3ae_gemm_desc); out3[1] = (int64_t)sizeof(m3ae_attn_desc); out3[2] = (int64_t)sizeof(m3ae_xattn_desc);
}

extern "C" int m3ae_colsum(const void* x, float* out, int64_t M, int64_t N, int64_t ldx, int dtype, int accumulate,
                           void* stream) {
    if (!x || !out || M <= 0 || N <= 0) return M3AE_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (!accumulate) {
        hipError_t e = hipMemsetAsync(out, 0, N * sizeof(float), s);
        if (e != hipSuccess) return (int)e;
    }
    if (N % 8 == 0 && ldx == N && N / 8 <= 512 && M >= 4096 && ((((uintptr_t)x) & 15) == 0) &&
        (dtype == M3AE_BF16 || dtype == M3AE_F32)) {
        const int P = (int)(N / 8), R = 512 / P;
        int64_t nwg = 2048;                      // 8 workgroups per CU
        int64_t rows_per = cdiv(cdiv(M, nwg), (int64_t)4 * R) * 4 * R;   // whole unrolled passes
        nwg = cdiv(M, rows_per);
        const size_t lds = (size_t)R * N * sizeof(float);
        DT_SWITCH(dtype, hipLaunchKernelGGL((colsum_rows_kernel<T>), dim3((unsigned)nwg), dim3((unsigned)(P * R)), lds, s,
                                            (const T*)x, out, M, P, R, rows_per));
        return hip_launch_status();
    }
    const int64_t col_blocks = cdiv(N, 256);
    int64_t chunks = cdiv(1024, col_blocks);  // ~1024 workgroups in flight
    if (chunks > cdiv(M, 64)) chunks = cdiv(M, 64);
    if (chunks < 1) chunks = 1;
    const int64_t rows_per = cdiv(M, chunks);
    chunks = cdiv(M, rows_per);
    dim3 grid((unsigned)col_blocks, (unsigned)chunks);
    const bool vec = (N % 8 == 0) && (ldx % 8 == 0) && ((((uintptr_t)x) & 15) == 0);
    if (vec) { DT_SWITCH(dtype, hipLaunchKernelGGL((colsum_kernel<T, true>), grid, dim3(256), 0, s, (const T*)x, out, M, N, ldx, rows_per)); }
    else { DT_SWITCH(dtype, hipLaunchKernelGGL((colsum_kernel<T, false>), grid, dim3(256), 0, s, (const T*)x, out, M, N, ldx, rows_per)); }
    return hip_launch_status();
}

extern "C" int m3ae_roberta_embed_fwd(const int64_t* ids, const float* word, const float* pos, const float* type,
                                      void* out, int64_t B, int64_t S, int64_t D, int64_t pad_id, int dtype,
                                      void* stream) {
    if (!ids || !word || !pos || !type || !out || B <= 0 || S <= 0 || S > 4096) return M3AE_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    DT_SWITCH(dtype, hipLaunchKernelGGL(roberta_embed_fwd_kernel<T>, dim3((unsigned)B), dim3(256), S * sizeof(int), s,
                                        ids, word, pos, type, (T*)out, S, D, pad_id));
    return hip_launch_status();
}
extern "C" int m3ae_roberta_embed_bwd(const int64_t* ids, const void* d_out, float* d_word, float* d_pos,
                                      float* d_type, int64_t B, int64_t S, int64_t D, int64_t pad_id, int dtype,
                                      void* stream) {
    if (!ids || !d_out || !d_word || !d_pos || B <= 0 || S <= 0 || S > 4096) return M3AE_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    DT_SWITCH(dtype, hipLaunchKernelGGL(roberta_embed_bwd_kernel<T>, dim3((unsigned)B), dim3(256), S * sizeof(int), s,
                                        ids, (const T*)d_out, d_word, d_pos, S, D, pad_id));
    int rc = hip_launch_status();
    if (rc) return rc;
    if (d_type) return m3ae_colsum(d_out, d_type, B * S, D, D, dtype, 1, stream);  // token_type 0 for every token
    return 0;
}

extern "C" int m3ae_patchify(const float* img, void* out, int64_t B, int64_t R, int64_t P, int dtype, void* stream) {
    if (!img || !out || B <= 0 || R <= 0 || P <= 0 || R % P) return M3AE_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = B * (R / P) * (R / P) * 3 * P * P;
    if (dtype == M3AE_BF16 && P % 8 == 0 && total / 8 < (int64_t)1 << 31) {
        hipLaunchKernelGGL(patchify8_bf16_kernel, dim3(ew_grid(total / 8)), dim3(EW_BLOCK), 0, s, img, (bf16_t*)out,
                           (uint32_t)(total / 8), (uint32_t)R, (uint32_t)P, (uint32_t)(R / P));
        return hip_launch_status();
    }
    DT_SWITCH(dtype, hipLaunchKernelGGL(patchify_kernel<T>, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, s, img, (T*)out, B, R, P));
    return hip_launch_status();
}
extern "C" int m3ae_image_normalize_u8(const uint8_t* in, float* out, int64_t B, int64_t H, int64_t W,
                                       const float* mean3, const float* std3, void* stream) {
    if (!in || !out || !mean3 || !std3 || B <= 0 || H <= 0 || W <= 0) return M3AE_ERR_ARG;  // mean3 / std3: HOST arrays
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(image_normalize_u8_kernel, dim3(ew_grid(B * 3 * H * W)), dim3(EW_BLOCK), 0, s, in, out, B, H, W,
                       mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
    return hip_launch_status();
}
extern "C" int m3ae_vit_tokens_fwd(const void* patch, const float* cls, const float* pos, void* out, int64_t B,
                                   int64_t G, int64_t D, int dtype, void* stream) {
    if (!patch || !cls || !pos || !out) return M3AE_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == M3AE_BF16 && D % 8 == 0 && B * (G + 1) * D / 8 < (int64_t)1 << 31) {
        const int64_t n8 = B * (G + 1) * D / 8;
        hipLaunchKernelGGL(vit_tokens_fwd8_bf16_kernel, dim3(ew_grid(n8)), dim3(EW_BLOCK), 0, s, (const bf16_t*)patch, cls,
                           pos, (bf16_t*)out, (uint32_t)n8, (uint32_t)G, (uint32_t)(D / 8));
        return hip_launch_status();
    }
    DT_SWITCH(dtype, hipLaunchKernelGGL(vit_tokens_fwd_kernel<T>, dim3(ew_grid(B * (G + 1) * D)), dim3(EW_BLOCK), 0, s,
                                        (const T*)patch, cls, pos, (T*)out, B, G, D));
    return hip_launch_status();
}
extern "C" int m3ae_vit_tokens_bwd(const void* d_out, void* d_patch, float* d_cls, float* d_pos, int64_t B, int64_t G,
                                   int64_t D, int dtype, void* stream) {
    if (!d_out || !d_patch || !d_cls || !d_pos) return M3AE_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    DT_SWITCH(dtype, hipLaunchKernelGGL(vit_tokens_bwd_kernel<T>, dim3(ew_grid((G + 1) * D)), dim3(EW_BLOCK), 0, s,
                                        (const T*)d_out, (T*)d_patch, d_cls, d_pos, B, G, D));
    return hip_launch_status();
}

extern "C" int m3ae_bce_logits(const void* logits, const float* targets, float* loss, void* d_logits, int64_t B,
                               int64_t C, float grad_scale, int dtype, void* stream) {
    if (!logits || !targets || !loss || B <= 0 || C <= 0) return M3AE_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(loss, 0, sizeof(float), s);
    if (e != hipSuccess) return (int)e;
    const int64_t n = B * C;
    unsigned grid = ew_grid(n);
    if (grid > 64) grid = 64;
    DT_SWITCH(dtype, hipLaunchKernelGGL(bce_kernel<T>, dim3(grid), dim3(EW_BLOCK), 0, s, (const T*)logits, targets, loss,
                                        (T*)d_logits, n, 1.0f / (float)n, (float)C, grad_scale));
    return hip_launch_status();
}

extern "C" int m3ae_xent(const void* logits, const int64_t* labels, float* loss, void* d_logits, float* workspace,
                         int64_t rows, int64_t C, int64_t ld, float grad_scale, int dtype, void* stream) {
    if (!logits || !labels || !loss || !workspace || rows <= 0 || C <= 0) return M3AE_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(loss, 0, sizeof(float), s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(xent_count_kernel, dim3(1), dim3(256), 0, s, labels, workspace, rows);
    DT_SWITCH(dtype, hipLaunchKernelGGL(xent_kernel<T>, dim3((unsigned)rows), dim3(256), 0, s, (const T*)logits, labels,
                                        loss, (T*)d_logits, workspace, C, ld, grad_scale));
    return hip_launch_status();
}

extern "C" int m3ae_adamw(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n, float lr,
                          float beta1, float beta2, float eps, float wd, int64_t step, float grad_scale,
                          const float* hyper_dev, void* stream) {
    if (!p || !g || !m || !v || n <= 0 || step <= 0) return M3AE_ERR_ARG;
    if (n % 4 != 0 || (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15)) return M3AE_ERR_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr * sqrt(bc2) / bc1);
    hipLaunchKernelGGL(adamw_kernel, dim3(ew_grid(n / 4)), dim3(EW_BLOCK), 0, s, p, g, m, v, (bf16_t*)shadow_bf16, n / 4,
                       lr, beta1, beta2, eps, wd, step_size, grad_scale, hyper_dev);
    return hip_launch_status();
}

extern "C" int m3ae_cast_transpose(const float* in, void* out, void* out_t, int64_t R, int64_t C, void* stream) {
    if (!in || R <= 0 || C <= 0) return M3AE_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)cdiv(C, 32), (unsigned)cdiv(R, 32));
    if (grid.y > 65535u) return M3AE_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(cast_transpose_kernel, grid, dim3(256), 0, s, in, (bf16_t*)out, (bf16_t*)out_t, R, C);
    return hip_launch_status();
}

extern "C" int m3ae_cast_transpose_batched(const void* jobs_dev, int njobs, int64_t total_tiles, void* stream) {
    if (!jobs_dev || njobs <= 0 || total_tiles <= 0) return M3AE_ERR_ARG;
    hipLaunchKernelGGL(cast_transpose_batched_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream,
                       (const TransposeJob*)jobs_dev, njobs);
    return hip_launch_status();
}

extern "C" int m3ae_transpose_bf16_batched(const void* jobs_dev, int njobs, int64_t total_tiles, void* stream) {
    if (!jobs_dev || njobs <= 0 || total_tiles <= 0) return M3AE_ERR_ARG;
    hipLaunchKernelGGL(transpose_bf16_batched_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream,
                       (const TransposeJob16*)jobs_dev, njobs);
    return hip_launch_status();
}

extern "C" int m3ae_cast(const void* in, void* out, int64_t n, int dtype_in, int dtype_out, void* stream) {
    if (!in || !out || n <= 0) return M3AE_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const dim3 g(ew_grid(n)), b(EW_BLOCK);
    if (dtype_in == M3AE_F32 && dtype_out == M3AE_BF16)
        hipLaunchKernelGGL((cast_kernel<float, bf16_t>), g, b, 0, s, (const float*)in, (bf16_t*)out, n);
    else if (dtype_in == M3AE_BF16 && dtype_out == M3AE_F32)
        hipLaunchKernelGGL((cast_kernel<bf16_t, float>), g, b, 0, s, (const bf16_t*)in, (float*)out, n);
    else if (dtype_in == M3AE_F32 && dtype_out == M3AE_F32)
        hipLaunchKernelGGL((cast_kernel<float, float>), g, b, 0, s, (const float*)in, (float*)out, n);
    else if (dtype_in == M3AE_BF16 && dtype_out == M3AE_BF16)
        hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), g, b, 0, s, (const bf16_t*)in, (bf16_t*)out, n);
    else return M3AE_ERR_UNSUPPORTED;
    return hip_launch_status();
}
extern "C" int m3ae_zero(void* p, int64_t bytes, void* stream) {   // stream-ordered zero fill (a memset node under capture)
    if (!p || bytes <= 0) return M3AE_ERR_ARG;
    const hipError_t e = hipMemsetAsync(p, 0, (size_t)bytes, (hipStream_t)stream);
    return e == hipSuccess ? 0 : (int)e;
}
extern "C" int m3ae_add(const void* a, const void* b, void* out, int64_t n, int dtype, void* stream) {
    if (!a || !b || !out || n <= 0) return M3AE_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == M3AE_BF16 && n % 8 == 0 && ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)out)) & 15) == 0) {
        hipLaunchKernelGGL(add_bf16x8_kernel, dim3(ew_grid(n / 8)), dim3(EW_BLOCK), 0, s, (const u32x4*)a, (const u32x4*)b, (u32x4*)out, n / 8);
        return hip_launch_status();
    }
    DT_SWITCH(dtype, hipLaunchKernelGGL(add_kernel<T>, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, s, (const T*)a, (const T*)b, (T*)out, n));
    return hip_launch_status();
}
extern "C" int m3ae_act_fwd(const void* x, void* y, int64_t n, int act, int dtype, void* stream) {
    if (!x || !y || n <= 0) return M3AE_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    DT_SWITCH(dtype, hipLaunchKernelGGL(act_fwd_kernel<T>, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, s, (const T*)x, (T*)y, n, act));
    return hip_launch_status();
}
extern "C" int m3ae_act_bwd(const void* dy, const void* x_pre, void* dx, int64_t n, int act, int dtype, void* stream) {
    if (!dy || !x_pre || !dx || n <= 0) return M3AE_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    DT_SWITCH(dtype, hipLaunchKernelGGL(act_bwd_kernel<T>, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, s, (const T*)dy, (const T*)x_pre, (T*)dx, n, act));
    return hip_launch_status();
}
extern "C" int m3ae_dropout(const void* x, void* out, uint8_t* keep_mask, int64_t rows, int64_t cols, float p,
                            uint64_t seed, const void* salt, int dtype, void* stream) {
    if (rows <= 0 || cols <= 0 || p < 0.f || p >= 1.f || (!out && !keep_mask) || (out && !x)) return M3AE_ERR_ARG;
    const int64_t n = rows * cols;
    hipStream_t s = (hipStream_t)stream;
    const DropState ds = make_drop(p, seed, salt);
    DT_SWITCH(dtype, hipLaunchKernelGGL(dropout_kernel<T>, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, s, (const T*)x, (T*)out,
                                        keep_mask, n, cols, drop_ld(cols), ds));
    return hip_launch_status();
}

extern "C" int m3ae_gather_rows(const void* in, const int64_t* idx, void* out, int64_t n_out, int64_t D, int dtype,
                                void* stream) {
    if (!in || !idx || !out || n_out <= 0 || D <= 0) return M3AE_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    DT_SWITCH(dtype, hipLaunchKernelGGL(gather_rows_kernel<T>, dim3(ew_grid(n_out * D)), dim3(EW_BLOCK), 0, s, (const T*)in, idx, (T*)out, n_out, D));
    return hip_launch_status();
}
extern "C" int m3ae_scatter_add_rows(const void* d_out, const int64_t* idx, void* d_in, int64_t n_out, int64_t D,
                                     int dtype, void* stream) {
    if (!d_out || !idx || !d_in || n_out <= 0 || D <= 0) return M3AE_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    DT_SWITCH(dtype, hipLaunchKernelGGL(scatter_add_rows_kernel<T>, dim3(ew_grid(n_out * D)), dim3(EW_BLOCK), 0, s, (const T*)d_out, idx, (T*)d_in, n_out, D));
    return hip_launch_status();
}

// ---------------------------------------------------------------------------------------------------------
// Masked-image-modelling bookkeeping (pre-training): random_masking's index work, the MIM target and
// the masked mean-squared error with its gradient.
// ---------------------------------------------------------------------------------------------------------
// m3ae_module.py:153-183: ids_shuffle = argsort(noise), ids_restore = argsort(ids_shuffle) = the RANK of every patch;
// kept patches = ranks < len_keep.  One workgroup per sample, the noise row in LDS, rank by counting
// (value, index) pairs in order -- L^2 comparisons (576^2 per sample) instead of a sort; ties break by index (a stable
// argsort).  keep_rows: flat row ids into [B * (L + 1)] token rows, the class row first (m3ae_module.py:176-180).
namespace {
__global__ void mask_ranks_kernel(const float* __restrict__ noise, int64_t* __restrict__ ids_restore,
                                  int64_t* __restrict__ keep_rows, float* __restrict__ mask, int L, int len_keep) {
    extern __shared__ float row[];
    const int64_t b = blockIdx.x;
    for (int i = threadIdx.x; i < L; i += blockDim.x) row[i] = noise[b * L + i];
    __syncthreads();
    if (threadIdx.x == 0) keep_rows[b * (len_keep + 1)] = b * (L + 1);
    for (int j = threadIdx.x; j < L; j += blockDim.x) {
        const float v = row[j];
        int rank = 0;
        for (int i = 0; i < L; ++i) {
            const float u = row[i];
            rank += (u < v || (u == v && i < j)) ? 1 : 0;
        }
        ids_restore[b * L + j] = rank;
        mask[b * L + j] = rank >= len_keep ? 1.0f : 0.0f;
        if (rank < len_keep) keep_rows[b * (len_keep + 1) + 1 + rank] = b * (L + 1) + 1 + j;
    }
}

// m3ae_module.py:185-192 (einsum nchpwq->nhwpqc) + objectives.py:52-56: one wave per patch; element (p, q, c) of patch
// (h, w) is img[n][c][h P + p][w P + q]; with norm_pix the patch is standardised with the UNBIASED variance + 1e-6.
__global__ void mim_targets_kernel(const float* __restrict__ img, float* __restrict__ out, int64_t n_patches, int C,
                                   int H, int W, int P, int norm_pix) {
    const int64_t patch = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (patch >= n_patches) return;
    const int lane = threadIdx.x & 63;
    const int gw = W / P, gh = H / P;
    const int64_t n = patch / (gh * gw);
    const int hw = (int)(patch % (gh * gw)), ph = hw / gw, pw = hw % gw;
    const int D = P * P * C;
    const float* base = img + n * (int64_t)C * H * W;
    // lanes walk the patch in IMAGE order (q fastest: 4 P-byte runs of the source rows, coalesced) and scatter into the
    // (p, q, c) order of the output row, which one wave fills completely
    const int PP = P * P;
    auto src = [&](int i) {  // i = (c * P + p) * P + q
        const int c = i / PP, pq = i % PP, p = pq / P, q = pq % P;
        return base[((int64_t)c * H + ph * P + p) * W + pw * P + q];
    };
    auto dst = [&](int i) { return (i % PP) * C + i / PP; };
    float sum = 0.f;
    for (int i = lane; i < D; i += 64) sum += src(i);
    const float mean = wave_sum(sum) / (float)D;
    float ss = 0.f;
    for (int i = lane; i < D; i += 64) { const float d = src(i) - mean; ss += d * d; }
    const float var = wave_sum(ss) / (float)(D - 1);
    const float rstd = 1.0f / sqrtf(var + 1.e-6f);
    float* o = out + patch * D;
    for (int i = lane; i < D; i += 64) o[dst(i)] = norm_pix ? (src(i) - mean) * rstd : src(i);
}

// objectives.py:58-62: loss = sum_n mask[n] mean_d (x[n][d] - t[n][d])^2 / sum_n mask[n].  x is the decoder output WITH its
// class row ([B, L + 1, D], row 0 of every sample skipped: prediction_heads.py:86), t / mask are [B, L, ...].
// acc[0] += masked per-patch errors, acc[1] += mask (fp32 atomics, one per wave).
template <typename T>
__global__ void mim_loss_fwd_kernel(const T* __restrict__ x, const float* __restrict__ t, const float* __restrict__ mask,
                                    float* __restrict__ acc, int64_t N, int L, int D) {
    // grid-stride over rows, sums kept per wave, one pair of atomics per WORKGROUP (one pair per row serialised 55 k
    // atomics on two addresses: 1.4 ms for 74 k rows)
    __shared__ float red[4][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float num = 0.f, den = 0.f;
    for (int64_t n = (int64_t)blockIdx.x * 4 + wave; n < N; n += (int64_t)gridDim.x * 4) {
        const float m = mask[n];
        if (m == 0.f) continue;
        const T* xr = x + ((n / L) * (L + 1) + 1 + n % L) * (int64_t)D;
        const float* tr = t + n * (int64_t)D;
        float ss = 0.f;
        for (int d = lane; d < D; d += 64) { const float e = Elem<T>::ld(xr + d) - tr[d]; ss += e * e; }
        num += m * wave_sum(ss) / (float)D;
        den += m;
    }
    if (lane == 0) { red[wave][0] = num; red[wave][1] = den; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(acc, (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]));
        atomicAdd(acc + 1, (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]));
    }
}
__global__ void mim_loss_finalize_kernel(const float* __restrict__ acc, float* __restrict__ loss) { loss[0] = acc[0] / acc[1]; }
// dx[b][0][:] = 0 ;  dx[b][1 + l][d] = gout * 2 (x - t) mask / (D * sum mask)
template <typename T>
__global__ void mim_loss_bwd_kernel(const T* __restrict__ x, const float* __restrict__ t, const float* __restrict__ mask,
                                    const float* __restrict__ acc, const float* __restrict__ gout, T* __restrict__ dx,
                                    int64_t rows, int L, int D) {
    const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);  // row of [B * (L + 1)]
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    const int64_t b = r / (L + 1);
    const int l = (int)(r % (L + 1)) - 1;
    T* dr = dx + r * (int64_t)D;
    const float m = l < 0 ? 0.f : mask[b * L + l];
    if (m == 0.f) {
        for (int d = lane; d < D; d += 64) Elem<T>::st(dr + d, 0.f);
        return;
    }
    const float k = gout[0] * 2.0f * m / ((float)D * acc[1]);
    const T* xr = x + r * (int64_t)D;
    const float* tr = t + (b * L + l) * (int64_t)D;
    for (int d = lane; d < D; d += 64) Elem<T>::st(dr + d, k * (Elem<T>::ld(xr + d) - tr[d]));
}
}  // namespace

extern "C" int m3ae_mask_ranks(const float* noise, int64_t* ids_restore, int64_t* keep_rows, float* mask, int64_t B,
                               int64_t L, int64_t len_keep, void* stream) {
    if (!noise || !ids_restore || !keep_rows || !mask || B <= 0 || L <= 0 || len_keep < 0 || len_keep > L) return M3AE_ERR_ARG;
    if (L > 16384) return M3AE_ERR_UNSUPPORTED;   // the noise row lives in LDS
    hipLaunchKernelGGL(mask_ranks_kernel, dim3((unsigned)B), dim3(256), (size_t)L * 4, (hipStream_t)stream, noise,
                       ids_restore, keep_rows, mask, (int)L, (int)len_keep);
    return hip_launch_status();
}
extern "C" int m3ae_mim_targets(const float* img, float* out, int64_t B, int64_t C, int64_t H, int64_t W, int64_t P,
                                int norm_pix, void* stream) {
    if (!img || !out || B <= 0 || C <= 0 || P <= 0 || H <= 0 || W <= 0 || H % P || W % P || P * P * C < 2) return M3AE_ERR_ARG;
    const int64_t n = B * (H / P) * (W / P);
    hipLaunchKernelGGL(mim_targets_kernel, dim3((unsigned)cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, img, out, n,
                       (int)C, (int)H, (int)W, (int)P, norm_pix);
    return hip_launch_status();
}
extern "C" int m3ae_mim_loss_fwd(const void* x, const float* target, const float* mask, float* acc, float* loss, int64_t B,
                                 int64_t L, int64_t D, int dtype, void* stream) {
    if (!x || !target || !mask || !acc || !loss || B <= 0 || L <= 0 || D <= 0) return M3AE_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    { const hipError_t e = hipMemsetAsync(acc, 0, 2 * sizeof(float), s); if (e != hipSuccess) return (int)e; }
    const int64_t fwd_blocks = cdiv(B * L, 4) < 1024 ? cdiv(B * L, 4) : 1024;
    DT_SWITCH(dtype, hipLaunchKernelGGL(mim_loss_fwd_kernel<T>, dim3((unsigned)fwd_blocks), dim3(256), 0, s, (const T*)x,
                                        target, mask, acc, B * L, (int)L, (int)D));
    hipLaunchKernelGGL(mim_loss_finalize_kernel, dim3(1), dim3(1), 0, s, acc, loss);
    return hip_launch_status();
}
extern "C" int m3ae_mim_loss_bwd(const void* x, const float* target, const float* mask, const float* acc, const float* gout,
                                 void* dx, int64_t B, int64_t L, int64_t D, int dtype, void* stream) {
    if (!x || !target || !mask || !acc || !gout || !dx || B <= 0 || L <= 0 || D <= 0) return M3AE_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    DT_SWITCH(dtype, hipLaunchKernelGGL(mim_loss_bwd_kernel<T>, dim3((unsigned)cdiv(B * (L + 1), 4)), dim3(256), 0, s,
                                        (const T*)x, target, mask, acc, gout, (T*)dx, B * (L + 1), (int)L, (int)D));
    return hip_launch_status();
}
