// Sustained fp32-MFMA rate of the device with NO memory traffic: the ceiling a perfect conv kernel could reach at
// the clock the chip actually holds under matrix load (the datasheet peak assumes 2.4 GHz).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/bin/mfma_peak && tools/bin/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters) {
    f32x16 acc[NACC];
    for (int k = 0; k < NACC; ++k)
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[k], 0, 0, 0);
    }
    float s = 0.f;
    for (int k = 0; k < NACC; ++k)
        for (int r = 0; r < 16; ++r) s += acc[k][r];
    if (s == 123.456f) out[0] = s;
}

template <int NACC>
static void run(int blocks_per_cu, int iters, const char* tag) {
    float* out;
    (void)hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int nblk = 256 * blocks_per_cu;
    mfma_loop<NACC><<<nblk, 256>>>(out, 16);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    mfma_loop<NACC><<<nblk, 256>>>(out, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)nblk * 4 * iters * 8 * NACC * 4096.0;
    printf("%-28s %d waves/SIMD, %d acc: %8.3f ms  %7.1f TFLOP/s  (= %.3f GHz x 64 FLOP/clk x 1024 SIMD)\n", tag, blocks_per_cu,
           NACC, ms, flop / ms / 1e9, flop / ms / 1e9 / (64.0 * 1024) * 1e3);
    (void)hipFree(out);
}

int main() {
    run<2>(1, 2000, "short (~1.5 ms)");
    run<2>(1, 20000, "medium (~15 ms)");
    run<2>(1, 200000, "long (~150 ms)");
    run<2>(2, 100000, "long, 2 waves/SIMD");
    run<2>(4, 50000, "long, 4 waves/SIMD");
    run<1>(1, 40000, "1 acc (dependent chain)");
    run<2>(1, 200000, "long again");
    return 0;
}
