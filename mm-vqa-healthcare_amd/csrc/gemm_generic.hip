// Generic strided / batched GEMM with the fused epilogue of m3ae_gemm (include/m3ae_hip.h).
//
// Role: (1) every matmul of "parity mode" (fp32 operands, fp32 FMA accumulation in ascending k: deterministic and
// within ~1e-6 of the reference's fp32 ATen path); (2) the odd shapes of "perf mode" that the MFMA kernels reject
// (N = 498 answer logits, batched per-head products of the fp32 attention path, strided token-0 gathers of the
// poolers).  LDS-tiled 64x64x16, 256 threads, 4x4 micro-tile per thread, arbitrary element strides.
#include "common.h"

namespace {

constexpr int TM = 64, TN = 64, TK = 16, PAD = 4;

template <typename TA, typename TB, typename TC>
__global__ __launch_bounds__(256) void gemm_generic_kernel(m3ae_gemm_desc d) {
    __shared__ float As[TK][TM + PAD];
    __shared__ float Bs[TK][TN + PAD];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int64_t m0 = (int64_t)blockIdx.y * TM, n0 = (int64_t)blockIdx.x * TN;
    const int64_t b1 = blockIdx.z / d.batch2, b2 = blockIdx.z % d.batch2;
    const TA* A = (const TA*)d.A + b1 * d.a_sb1 + b2 * d.a_sb2;
    const TB* B = (const TB*)d.B + b1 * d.b_sb1 + b2 * d.b_sb2;
    const int64_t coff = b1 * d.c_sb1 + b2 * d.c_sb2;
    const bool a_kfast = (d.a_sk == 1), b_nfast = (d.b_sn == 1);

    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    const bool do_rowsum = d.a_rowsum != nullptr && blockIdx.x == 0 && tx == 0;
    float rs[4] = {0.f, 0.f, 0.f, 0.f};

    for (int64_t k0 = 0; k0 < d.K; k0 += TK) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int e = tid + i * 256;
            int mm, kk;
            if (a_kfast) { kk = e & 15; mm = e >> 4; } else { mm = e & 63; kk = e >> 6; }
            int64_t gm = m0 + mm, gk = k0 + kk;
            float v = 0.f;
            if (gm < d.M && gk < d.K) v = Elem<TA>::ld(A + gm * d.a_sm + gk * d.a_sk);
            As[kk][mm] = v;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int e = tid + i * 256;
            int nn, kk;
            if (b_nfast) { nn = e & 63; kk = e >> 6; } else { kk = e & 15; nn = e >> 4; }
            int64_t gn = n0 + nn, gk = k0 + kk;
            float v = 0.f;
            if (gn < d.N && gk < d.K) v = Elem<TB>::ld(B + gk * d.b_sk + gn * d.b_sn);
            Bs[kk][nn] = v;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < TK; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[kk][ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Bs[kk][tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
            if (do_rowsum) {
#pragma unroll
                for (int i = 0; i < 4; ++i) rs[i] += a[i];
            }
        }
        __syncthreads();
    }
    if (do_rowsum) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t gm = m0 + ty * 4 + i;
            if (gm < d.M) d.a_rowsum[gm] += rs[i];  // one writer per row (blockIdx.x == 0, tx == 0)
        }
    }

    DropState drop = make_drop_dev(d.dropout_p, d.dropout_seed, d.dropout_salt);
    drop_resolve(drop);
    TC* C = (TC*)d.C + coff;
    TC* P = d.preact ? (TC*)d.preact + coff : nullptr;
    const TC* R = d.residual ? (const TC*)d.residual + coff : nullptr;
    const TC* X = d.dact_aux ? (const TC*)d.dact_aux + coff : nullptr;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int64_t gm = m0 + ty * 4 + i;
        if (gm >= d.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int64_t gn = n0 + tx * 4 + j;
            if (gn >= d.N) continue;
            int64_t off = gm * d.c_sm + gn * d.c_sn;
            float x = acc[i][j] * d.alpha;
            if (d.bias) x += d.bias[gn];
            if (P) Elem<TC>::st(P + off, d.preact_grad ? act_bwd(x, d.act) : x);
            x = act_fwd(x, d.act);
            if (d.dropout_p > 0.f) x = drop_apply(drop, (uint64_t)(gm * drop_ld(d.N) + gn), x);
            if (R) x += Elem<TC>::ld(R + off);
            if (X) x *= act_bwd(Elem<TC>::ld(X + off), d.dact);
            if (d.accumulate) x += Elem<TC>::ld(C + off);
            Elem<TC>::st(C + off, x);
        }
    }
}

template <typename TA, typename TB, typename TC>
int launch(const m3ae_gemm_desc& d, hipStream_t s) {
    dim3 grid((unsigned)cdiv(d.N, TN), (unsigned)cdiv(d.M, TM), (unsigned)(d.batch1 * d.batch2));
    if (grid.y > 65535u || grid.z > 65535u) return M3AE_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((gemm_generic_kernel<TA, TB, TC>), grid, dim3(256), 0, s, d);
    return hip_launch_status();
}

}  // namespace

int m3ae_gemm_generic(const m3ae_gemm_desc& d, hipStream_t s) {
    if (d.dtype_a != d.dtype_b) return M3AE_ERR_UNSUPPORTED;
    if (d.dtype_a == M3AE_F32 && d.dtype_c == M3AE_F32) return launch<float, float, float>(d, s);
    if (d.dtype_a == M3AE_BF16 && d.dtype_c == M3AE_BF16) return launch<bf16_t, bf16_t, bf16_t>(d, s);
    if (d.dtype_a == M3AE_BF16 && d.dtype_c == M3AE_F32) return launch<bf16_t, bf16_t, float>(d, s);
    return M3AE_ERR_UNSUPPORTED;
}
