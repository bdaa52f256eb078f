// EXPERIMENT (not in the product build): how fast can ONE CU take bytes in through LDS-DMA (global_load_lds_dwordx4) or plain
// 16-B register loads, as a function of the number of issuing waves, the pieces kept in flight per wave, the shape of a 1-KiB
// piece (16 rows x 64 B, 8 x 128 B, 4 x 256 B, 1 KiB contiguous) and where the bytes live (a small region every workgroup
// re-reads: L2 hits; a private region per workgroup read once: HBM).  One workgroup per CU, no compute.
#include "mfma_tiles.h"

template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// piece p of this wave: byte offset of lane's 16 B inside the wave's source region (row stride ld_bytes, rows of `seg` bytes)
template <int SEG>
__device__ __forceinline__ int64_t piece_off(int p, int lane, int64_t ld_bytes, int nk) {
    constexpr int LPR = SEG / 16, ROWS = 64 / LPR;       // lanes per row segment, rows per piece
    if (SEG == 1024) return (int64_t)p * 1024 + lane * 16;
    const int kc = p % nk, rb = p / nk;                  // walk the reduction first (as a GEMM operand stream), then the next rows
    return ((int64_t)rb * ROWS + lane / LPR) * ld_bytes + (int64_t)kc * SEG + (lane % LPR) * 16;
}

template <int SEG, int INFLIGHT, int REG>
__global__ __launch_bounds__(1024) void dma_probe_kernel(const char* src, int64_t wg_span, int64_t wave_span, int shared_src,
                                                         int64_t ld_bytes, int nk, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = src + (shared_src ? 0 : (int64_t)blockIdx.x * wg_span) + (int64_t)wave * wave_span;
    char* ring = smem + wave * (INFLIGHT * 1024);
    const int pieces_per_span = (int)(wave_span / 1024);
    u32x4 accv = {0u, 0u, 0u, 0u};
    int slot = 0, p = 0;
    for (int i = 0; i < iters; ++i) {
        const char* g = base + piece_off<SEG>(p, lane, ld_bytes, nk);
        if (REG) {
            const u32x4 v = *(const u32x4*)g;
            accv ^= v;
            if ((i % INFLIGHT) == INFLIGHT - 1) asm volatile("" ::"v"(accv));
        } else {
            glds16(g, ring + slot * 1024);
            vm_wait<INFLIGHT - 1>();
        }
        slot = slot + 1 == INFLIGHT ? 0 : slot + 1;
        p = p + 1 == pieces_per_span ? 0 : p + 1;
    }
    vm_wait<0>();
    if (REG && (accv[0] ^ accv[1] ^ accv[2] ^ accv[3]) == 0x12345u) sink[0] = 1;
}

template <int SEG, int INFLIGHT, int REG>
static int launch(const char* src, int64_t wg_span, int64_t wave_span, int shared_src, int64_t ld, int nk, int iters, int nwaves,
                  int grid, unsigned* sink, hipStream_t s) {
    const int lds = nwaves * INFLIGHT * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&dma_probe_kernel<SEG, INFLIGHT, REG>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((dma_probe_kernel<SEG, INFLIGHT, REG>), dim3(grid), dim3(nwaves * 64), lds, s, src, wg_span, wave_span, shared_src,
                       ld, nk, iters, sink);
    return (int)hipGetLastError();
}

#define CASE(SEG, INF, REG) if (seg == SEG && inflight == INF && reg == REG) return launch<SEG, INF, REG>(src, wg_span, wave_span, shared_src, ld, nk, iters, nwaves, grid, sink, (hipStream_t)stream);
extern "C" int dma_probe(const char* src, int64_t wg_span, int64_t wave_span, int shared_src, int64_t ld, int nk, int iters, int nwaves,
                         int grid, int seg, int inflight, int reg, unsigned* sink, void* stream) {
    CASE(64, 4, 0) CASE(64, 8, 0) CASE(64, 12, 0)
    CASE(128, 4, 0) CASE(128, 8, 0) CASE(128, 12, 0)
    CASE(256, 4, 0) CASE(256, 8, 0) CASE(256, 12, 0)
    CASE(1024, 2, 0) CASE(1024, 4, 0) CASE(1024, 8, 0) CASE(1024, 12, 0)
    CASE(64, 8, 1) CASE(128, 8, 1) CASE(1024, 8, 1)
    return -1;
}
