// Sustained fp32-MFMA rate as a function of the OPERAND DATA (no memory traffic): the chip clocks to its power budget, and the power of
// the matrix pipe depends on how many operand bits toggle.  tools/mfma_peak.hip feeds every MFMA the same two registers (the datasheet
// situation); this twin cycles through 16 + 16 per-lane values that are all zero / constant / N(0,1)-like random — the last one is what a
// convolution of real activations looks like, i.e. the ceiling the roofline fraction of the conv kernels should be read against.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak_data.hip -o tools/bin/mfma_peak_data && tools/bin/mfma_peak_data
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float hash_unit(unsigned x) {  // ~N(0,1)-ish: sum of 4 uniforms, centred
    float s = 0.f;
    for (int i = 0; i < 4; ++i) {
        x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
        s += (float)(x & 0xffffff) * (1.0f / 16777216.0f);
    }
    return (s - 2.0f) * 1.7320508f;
}

template <int MODE>  // 0 zeros, 1 one constant pair, 2 random per lane and step
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters) {
    f32x16 acc[2];
    for (int k = 0; k < 2; ++k)
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    float a[16], b[16];
    const unsigned id = blockIdx.x * 256 + threadIdx.x;
    for (int i = 0; i < 16; ++i) {
        a[i] = MODE == 0 ? 0.f : (MODE == 1 ? 0.37f : hash_unit(id * 32 + i));
        b[i] = MODE == 0 ? 0.f : (MODE == 1 ? -1.21f : hash_unit(id * 32 + 16 + i));
    }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int k = 0; k < 2; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[(u + 5 * k) & 15], acc[k], 0, 0, 0);
    }
    float s = 0.f;
    for (int k = 0; k < 2; ++k)
        for (int r = 0; r < 16; ++r) s += acc[k][r];
    if (s == 123.456f) out[0] = s;
}

template <int MODE>
static void run(int blocks_per_cu, int iters, const char* tag) {
    float* out;
    (void)hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int nblk = 256 * blocks_per_cu;
    mfma_loop<MODE><<<nblk, 256>>>(out, 16);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    mfma_loop<MODE><<<nblk, 256>>>(out, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)nblk * 4 * iters * 32 * 4096.0;
    printf("%-34s %d waves/SIMD: %8.3f ms  %7.1f TFLOP/s  (= %.3f GHz x 64 FLOP/clk x 1024 SIMD)\n", tag, blocks_per_cu, ms,
           flop / ms / 1e9, flop / ms / 1e9 / (64.0 * 1024) * 1e3);
    (void)hipFree(out);
}

int main() {
    for (int rep = 0; rep < 2; ++rep) {
        run<0>(2, 4000, "zeros, ~15 ms");
        run<1>(2, 4000, "one constant pair, ~15 ms");
        run<2>(2, 4000, "random operands, ~15 ms");
        run<2>(2, 300, "random operands, ~1 ms");
        run<2>(1, 8000, "random operands, 1 wave/SIMD");
    }
    return 0;
}
