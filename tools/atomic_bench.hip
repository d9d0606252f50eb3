// atomic_bench.hip — what the f64 statistics flush of the persistent kernels costs: B blocks each add 64 doubles (one per lane) to
// table[(block % R) * stride + lane]; R replicas `stride` doubles apart.  hipcc --offload-arch=gfx950 -O3 -o atomic_bench atomic_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void flush_kernel(double* table, int R, long long stride, int per_block) {
    const int l = threadIdx.x;
    if (l >= 64) return;
    double* dst = table + (long long)(blockIdx.x % R) * stride;
    for (int k = 0; k < per_block; ++k) unsafeAtomicAdd(dst + l, 1.0 + k);
}
int main(int argc, char** argv) {
    double* t;
    const long long cap = 64ll << 20;
    hipMalloc(&t, cap * sizeof(double));
    hipMemset(t, 0, cap * sizeof(double));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int Bs[] = {512, 2048};
    const int Rs[] = {1, 2, 4, 8, 16, 64, 512};
    const long long strides[] = {64, 512, 8192, 1 << 17};
    for (int B : Bs)
        for (int R : Rs)
            for (long long st : strides) {
                if (R == 1 && st != 64) continue;
                for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(flush_kernel, dim3(B), dim3(256), 0, 0, t, R, st, 1);
                hipEventRecord(e0, 0);
                for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(flush_kernel, dim3(B), dim3(256), 0, 0, t, R, st, 1);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                printf("blocks %4d replicas %3d stride %7lld doubles: %.2f us per launch\n", B, R, st, ms * 1000 / 20);
            }
    return 0;
}
