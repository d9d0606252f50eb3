// Sustained bf16-MFMA rate (v_mfma_f32_32x32x16_bf16) with NO memory traffic, and with the instruction mix of the bf16
// convolution's inner loop added step by step: the ceiling the kernels of csrc/u3d_bf16.hip can be measured against at the clock
// the chip actually holds under dense matrix load (the 2.5 PFLOP/s datasheet figure assumes 2.4 GHz).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_bf16_peak.hip -o tools/bin/mfma_bf16_peak && tools/bin/mfma_bf16_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// MODE 0: MFMAs only; 1: + one ds_read_b128 per 2 MFMAs (A fragments); 2: + one 16-byte global load per 2 MFMAs (B fragments)
template <int NACC, int MODE>
__global__ __launch_bounds__(256, 2) void mfma_loop(float* out, const bf16x8* __restrict__ wbuf, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[32768];
    f32x16 acc[NACC];
    for (int k = 0; k < NACC; ++k)
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) a[e] = (__bf16)(threadIdx.x * 1e-3f), b[e] = (__bf16)(blockIdx.x * 1e-3f + 1.f);
    for (int i = threadIdx.x; i < 8192; i += 256) reinterpret_cast<float*>(lds)[i] = 0.001f * i;
    __syncthreads();
    const int lane_off = (threadIdx.x & 63) * 16;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            bf16x8 a2 = a, b2 = b;
            if (MODE >= 1) a2 = *reinterpret_cast<const bf16x8*>(lds + lane_off + ((i * 8 + u) & 15) * 1024);
            if (MODE >= 2) b2 = wbuf[(size_t)((i * 8 + u) & 255) * 64 + (threadIdx.x & 63)];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k & 1 ? a2 : a, k & 2 ? b2 : b, acc[k], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (MODE >= 1) a = a2;
            if (MODE >= 2) b = b2;
        }
    }
    float s = 0.f;
    for (int k = 0; k < NACC; ++k)
        for (int r = 0; r < 16; ++r) s += acc[k][r];
    if (s == 123.456f) out[0] = s;
}

template <int NACC, int MODE>
static void run(int blocks_per_cu, int iters, const char* tag) {
    float* out;
    bf16x8* wbuf;
    (void)hipMalloc(&out, 4);
    (void)hipMalloc(&wbuf, 256 * 64 * 16);
    (void)hipMemset(wbuf, 0, 256 * 64 * 16);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int nblk = 256 * blocks_per_cu;
    mfma_loop<NACC, MODE><<<nblk, 256>>>(out, wbuf, 16);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    mfma_loop<NACC, MODE><<<nblk, 256>>>(out, wbuf, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)nblk * 4 * iters * 8 * NACC * 32768.0;
    printf("%-44s %d waves/SIMD, %d acc, mode %d: %8.3f ms  %7.1f TFLOP/s  (= %.3f GHz x 1024 FLOP/clk x 1024 SIMD)\n", tag,
           blocks_per_cu, NACC, MODE, ms, flop / ms / 1e9, flop / ms / 1e9 / (1024.0 * 1024) * 1e3);
    (void)hipFree(out);
    (void)hipFree(wbuf);
}

int main() {
    run<4, 0>(1, 2000, "MFMA only, short (~0.1 ms)");
    run<4, 0>(1, 20000, "MFMA only, ~1 ms");
    run<4, 0>(2, 20000, "MFMA only, 2 waves/SIMD");
    run<4, 0>(2, 200000, "MFMA only, 2 waves/SIMD, ~20 ms");
    run<4, 1>(2, 20000, "+ ds_read_b128 per 2 MFMAs");
    run<4, 2>(2, 20000, "+ ds_read + 16 B global load per 2 MFMAs");
    run<8, 2>(2, 10000, "8 accumulators, loads per 4 MFMAs");
    run<2, 0>(2, 40000, "2 accumulators (chain distance 64 cycles)");
    return 0;
}
