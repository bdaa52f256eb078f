// On-device self-test of the gfx950 idioms the MFMA kernels are built on.  Exact small-integer data, asymmetric
// operands (a swapped row/col map cannot pass).  out[0] = total mismatches, out[1..5] = per-test mismatches:
//   1: v_mfma_f32_16x16x32_bf16 A/B/C lane maps        2: v_mfma_f32_32x32x16_bf16 A/B/C lane maps
//   3: ds_read_b64_tr_b16 block/lane semantics         4: accumulator-as-B-operand k order (Y = A2 . X)
//   5: global_load_lds_dwordx4 destination = wave-uniform base + lane * 16
#include "common.h"

namespace {

DEVINL short ibf(int v) { return (short)f2bf((float)v); }  // small ints are exact in bf16

__global__ __launch_bounds__(64) void selftest_kernel(int32_t* out, const short* gsrc) {
    __shared__ __attribute__((aligned(16))) short lds[64 * 64];
    const int lane = threadIdx.x;
    int bad1 = 0, bad2 = 0, bad3 = 0, bad4 = 0, bad5 = 0;

    // ---- test 1: 16x16x32.  A[i][k] = (i + 2k) % 7 - 3, B[k][j] = (3k + j) % 5 - 2
    {
        s16x8 a, b;
        const int i = lane & 15, g = lane >> 4;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * g + j;
            a[j] = ibf((i + 2 * k) % 7 - 3);
            b[j] = ibf((3 * k + i) % 5 - 2);  // B[k][col = lane & 15]
        }
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c,
                                                    0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * g + r, col = i;
            int ref = 0;
            for (int k = 0; k < 32; ++k) ref += ((row + 2 * k) % 7 - 3) * ((3 * k + col) % 5 - 2);
            if ((int)c[r] != ref) ++bad1;
        }
    }
    // ---- test 2: 32x32x16
    f32x16 X;  // kept for test 4: X[row][col] = sum_k A[row][k] B[k][col]
    {
        s16x8 a, b;
        const int r = lane & 31, h = lane >> 5;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * h + j;
            a[j] = ibf((r + 2 * k) % 7 - 3);
            b[j] = ibf((3 * k + r) % 5 - 2);
        }
        f32x16 c;
#pragma unroll
        for (int t = 0; t < 16; ++t) c[t] = 0.f;
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c,
                                                    0, 0, 0);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int row = (t & 3) + 8 * (t >> 2) + 4 * h, col = r;
            int ref = 0;
            for (int k = 0; k < 16; ++k) ref += ((row + 2 * k) % 7 - 3) * ((3 * k + col) % 5 - 2);
            if ((int)c[t] != ref) ++bad2;
        }
        X = c;
    }
    // ---- test 3: transposing read.  tile[row][col] = row * 64 + col, 128-B rows
    for (int e = lane; e < 64 * 64; e += 64) lds[e] = (short)e;
    __syncthreads();
    {
        const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
        const int row0 = 4 * g + 8, col0 = 16 * (g & 1);  // an arbitrary 4 x 16 block per 16-lane group
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + (row0 + q) * 64 + col0 + 4 * p));
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if ((int)v[e] != (row0 + e) * 64 + col0 + i16) ++bad3;
    }
    __syncthreads();
    // ---- test 4: Y[i][col] = sum_{row} A2[i][row] * X[row][col], X taken from the accumulator as B operand
    {
        const int r = lane & 31, h = lane >> 5;
        f32x16 y;
#pragma unroll
        for (int t = 0; t < 16; ++t) y[t] = 0.f;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            u32x4 w;
            w[0] = pack2bf(X[8 * s + 0], X[8 * s + 1]);
            w[1] = pack2bf(X[8 * s + 2], X[8 * s + 3]);
            w[2] = pack2bf(X[8 * s + 4], X[8 * s + 5]);
            w[3] = pack2bf(X[8 * s + 6], X[8 * s + 7]);
            s16x8 a2;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int row = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);  // k order of the accumulator fragment
                a2[j] = ibf((r + row) % 3 - 1);                           // A2[i = r][row]
            }
            y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a2),
                                                        __builtin_bit_cast(bf16x8_t, __builtin_bit_cast(s16x8, w)), y, 0,
                                                        0, 0);
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int i = (t & 3) + 8 * (t >> 2) + 4 * h, col = r;
            int ref = 0;
            for (int row = 0; row < 32; ++row) {
                int x = 0;
                for (int k = 0; k < 16; ++k) x += ((row + 2 * k) % 7 - 3) * ((3 * k + col) % 5 - 2);
                ref += ((i + row) % 3 - 1) * x;  // |x| < 256: exact in bf16
            }
            if ((int)y[t] != ref) ++bad4;
        }
    }
    // ---- test 5: LDS-DMA destination order
    {
        typedef __attribute__((address_space(3))) void lds_void;
        typedef __attribute__((address_space(1))) const void gbl_cvoid;
        // lane l fetches source chunk (l ^ 5): LDS chunk l must then hold source chunk l ^ 5
        __builtin_amdgcn_global_load_lds((gbl_cvoid*)(gsrc + ((lane ^ 5) * 8)), (lds_void*)lds, 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (lds[lane * 8 + j] != gsrc[(lane ^ 5) * 8 + j]) ++bad5;
    }
    bad1 = (int)wave_sum((float)bad1); bad2 = (int)wave_sum((float)bad2); bad3 = (int)wave_sum((float)bad3);
    bad4 = (int)wave_sum((float)bad4); bad5 = (int)wave_sum((float)bad5);
    if (lane == 0) {
        out[0] = bad1 + bad2 + bad3 + bad4 + bad5;
        out[1] = bad1; out[2] = bad2; out[3] = bad3; out[4] = bad4; out[5] = bad5;
    }
}

__global__ void selftest_fill(short* g) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 512) g[i] = (short)(i * 7 + 3);
}

}  // namespace

// out: int32[8 + 256] device buffer (the tail is scratch for the LDS-DMA source)
extern "C" int m3ae_selftest(int32_t* out, void* stream) {
    if (!out) return M3AE_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    short* gsrc = (short*)(out + 8);
    hipLaunchKernelGGL(selftest_fill, dim3(2), dim3(256), 0, s, gsrc);
    hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, s, out, (const short*)gsrc);
    return hip_launch_status();
}
