// How much MFMA-pipe time do co-issued instructions cost?  Each wave runs a loop of 8 x { v_mfma_f32_32x32x2_f32 ;
// V VALU ; S SALU ; D ds_read_b128 } and the achieved fp32-MFMA rate is reported for 1 and 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_mix.hip -o tools/bin/mfma_mix && tools/bin/mfma_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// KIND selects what the extra slot instructions are: 0 v_add_u32, 1 s_add_u32, 2 ds_read_b128, 3 s_waitcnt (no-op counts),
// 4 global_load_dwordx4 (L1/L2 hit), 5 ds_write_b128, 6 v_pk_fma_f32, 7 v_mov_b32 dpp, 8 ds_read_b32, 9 s_nop 0,
// 10 ds_read2st64_b32, 11 v_cndmask_b32
template <int KIND, int CNT>
__global__ __launch_bounds__(256) void kind_loop(float* out, const float* gin, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    f32x16 acc[2];
    for (int k = 0; k < 2; ++k)
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.f;
    int x0 = threadIdx.x, x1 = 3;
    int s0 = blockIdx.x;
    const unsigned addr = (threadIdx.x & 63) * 16;
    f32x4 d0 = {0, 0, 0, 0}, g0 = {0, 0, 0, 0};
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 p0 = {1.f, 2.f}, p1 = {0.5f, 0.25f};
    const float* gp = gin + (threadIdx.x & 63) * 4;
    lds[threadIdx.x] = a;
    __syncthreads();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u & 1], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < CNT; ++c) {
                if (KIND == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x0) : "v"(x1));
                if (KIND == 1) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s0) : : "scc");
                if (KIND == 2) asm volatile("ds_read_b128 %0, %1" : "=v"(d0) : "v"(addr));
                if (KIND == 3) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                if (KIND == 4) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(g0) : "v"(gp));
                if (KIND == 5) asm volatile("ds_write_b128 %0, %1" : : "v"(addr), "v"(d0) : "memory");
                if (KIND == 6) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p0) : "v"(p1));
                if (KIND == 7) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x0) : "v"(x1));
                if (KIND == 8) asm volatile("ds_read_b32 %0, %1" : "=v"(x1) : "v"(addr));
                if (KIND == 9) asm volatile("s_nop 0");
                if (KIND == 10) asm volatile("ds_read2st64_b32 %0, %1 offset0:0 offset1:1" : "=v"(p0) : "v"(addr));
                if (KIND == 11) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x0) : "v"(x1));
            }
        }
        // wait only for the PREVIOUS iteration's memory operations: latency stays hidden, only issue cost is measured
        if ((KIND == 2 || KIND == 5 || KIND == 8 || KIND == 10) && CNT == 1) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        if ((KIND == 2 || KIND == 5 || KIND == 8 || KIND == 10) && CNT == 2) asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");
        if (KIND == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    float s = d0[0] + g0[0] + x0 + x1 + s0 + p0[0];
    for (int k = 0; k < 2; ++k)
        for (int r = 0; r < 16; ++r) s += acc[k][r];
    if (s == 123.456f) out[0] = s;
}

template <int KIND, int CNT>
static void runk(const char* what, int blocks_per_cu) {
    float *out, *gin;
    (void)hipMalloc(&out, 4);
    (void)hipMalloc(&gin, 4096);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int nblk = 256 * blocks_per_cu;
    const int iters = 20000 / blocks_per_cu;
    kind_loop<KIND, CNT><<<nblk, 256>>>(out, gin, 4000);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    kind_loop<KIND, CNT><<<nblk, 256>>>(out, gin, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)nblk * 4 * iters * 8 * 4096.0;
    const double cyc = 64.0 * 157.3 / (flop / ms / 1e9);
    printf("%d x %-22s per MFMA | %d waves/SIMD: %7.1f TFLOP/s  %6.1f cyc/MFMA  -> %5.1f cyc per extra instruction\n", CNT, what,
           blocks_per_cu, flop / ms / 1e9, cyc, CNT ? (cyc - 65.0) / CNT : 0.0);
    (void)hipFree(out);
    (void)hipFree(gin);
}

template <int V, int S, int D>
__global__ __launch_bounds__(256) void mix_loop(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    f32x16 acc[2];
    for (int k = 0; k < 2; ++k)
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.f;
    int x0 = threadIdx.x, x1 = 3, x2 = 5, x3 = 7;
    int s0 = blockIdx.x;
    const unsigned addr = (threadIdx.x & 63) * 16;
    f32x4 d0 = {0, 0, 0, 0};
    lds[threadIdx.x] = a;
    __syncthreads();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u & 1], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < V; ++v) {
                if ((v & 3) == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x0) : "v"(x1));
                if ((v & 3) == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x1) : "v"(x2));
                if ((v & 3) == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x2) : "v"(x3));
                if ((v & 3) == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x3) : "v"(x0));
            }
#pragma unroll
            for (int s = 0; s < S; ++s) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s0));
#pragma unroll
            for (int d = 0; d < D; ++d) asm volatile("ds_read_b128 %0, %1" : "=v"(d0) : "v"(addr));
        }
        if (D > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = d0[0] + x0 + x1 + x2 + x3 + s0;
    for (int k = 0; k < 2; ++k)
        for (int r = 0; r < 16; ++r) s += acc[k][r];
    if (s == 123.456f) out[0] = s;
}

template <int V, int S, int D>
static void run(int blocks_per_cu) {
    float* out;
    (void)hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int nblk = 256 * blocks_per_cu;
    const int iters = 40000 / blocks_per_cu;
    mix_loop<V, S, D><<<nblk, 256>>>(out, 2000);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    mix_loop<V, S, D><<<nblk, 256>>>(out, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)nblk * 4 * iters * 8 * 4096.0;
    printf("per MFMA: %2d VALU %2d SALU %2d ds_read_b128 | %d waves/SIMD: %7.1f TFLOP/s (%5.1f cyc/MFMA at 2.4 GHz)\n", V, S, D,
           blocks_per_cu, flop / ms / 1e9, 64.0 * 157.3 / (flop / ms / 1e9));
    (void)hipFree(out);
}

int main() {
    // warm the clocks
    run<0, 0, 0>(1);
    run<0, 0, 0>(1);
#define BOTH(K, C, W) runk<K, C>(W, 1); runk<K, C>(W, 2);
    BOTH(0, 2, "v_add_u32")
    BOTH(2, 1, "ds_read_b128")
    BOTH(8, 1, "ds_read_b32")
    BOTH(8, 2, "ds_read_b32")
    BOTH(10, 1, "ds_read2st64_b32")
    BOTH(5, 1, "ds_write_b128")
    BOTH(4, 1, "global_load_dwordx4")
    BOTH(6, 2, "v_pk_fma_f32")
    BOTH(7, 2, "v_mov_b32_dpp")
    BOTH(11, 2, "v_cndmask_b32")
    return 0;
}
int main_old() {
    run<0, 0, 0>(1);
    run<0, 0, 0>(1);
    run<0, 0, 0>(2);
    run<1, 0, 0>(1);
    run<2, 0, 0>(1);
    run<4, 0, 0>(1);
    run<8, 0, 0>(1);
    run<12, 0, 0>(1);
    run<2, 0, 0>(2);
    run<4, 0, 0>(2);
    run<8, 0, 0>(2);
    run<0, 4, 0>(1);
    run<0, 8, 0>(1);
    run<0, 8, 0>(2);
    run<0, 0, 1>(1);
    run<0, 0, 2>(1);
    run<0, 0, 2>(2);
    run<2, 2, 1>(1);
    run<2, 2, 1>(2);
    run<4, 4, 1>(1);
    run<4, 4, 1>(2);
    return 0;
}
