// u3d_act.hip — layer orders other than 'gcr' (the order mini-language of buildingblocks.py:10-96): LeakyReLU / ELU
// non-linearities ('l', 'e': buildingblocks.py:47-51; ResNetBlock's LeakyReLU(0.1) / ELU, :270-275) and the post-norm orders
// ('cgr', 'cgl', 'cge', 'cg': GroupNorm AFTER the convolution, :62-66 with is_before_conv False).  The convolutions are the same
// MFMA kernels (epilogue ReLU off); these bandwidth kernels supply what the fused 'gcr' pipeline folds into its epilogues:
//   u3d_act_fwd         y = f(x)                               (in place allowed)
//   u3d_act_bwd         out = g * f'(.) expressed through the OUTPUT y of f (sign(y) = sign(x); ELU': y + 1 for y <= 0)
//   u3d_affine_act_fwd  y = f(a[n,c] * z + b[n,c])             GroupNorm apply (+ non-linearity) of a post-norm layer
//   u3d_affine_add_act_fwd  y = f(a*z + b + add)                the same with a ResNetBlock's `out += residual` before f
//   u3d_pair_stats      stats[n][c] += (sum_v a, sum_v a*b)    the two reductions GroupNorm backward needs
// Activation codes: 0 none, 1 ReLU, 2 LeakyReLU(slope), 3 ELU(alpha = 1).  All tensors NDHWC fp32.
#include "u3d_common.h"

namespace {

__device__ __forceinline__ float act_f(float x, int mode, float slope) {
    if (mode == 1) return fmaxf(x, 0.f);
    if (mode == 2) return x > 0.f ? x : slope * x;
    if (mode == 3) return x > 0.f ? x : expm1f(x);
    return x;
}
// derivative of f at the point whose OUTPUT is y
__device__ __forceinline__ float act_df(float y, int mode, float slope) {
    if (mode == 1) return y > 0.f ? 1.f : 0.f;
    if (mode == 2) return y > 0.f ? 1.f : slope;
    if (mode == 3) return y > 0.f ? 1.f : y + 1.f;
    return 1.f;
}

__global__ void act_fwd_kernel(const float* __restrict__ x, long long n, int mode, float slope, float* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = act_f(x[i], mode, slope);
}

__global__ void act_bwd_kernel(const float* __restrict__ g, const float* __restrict__ y, long long n, int mode, float slope,
                               float* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = g[i] * act_df(y[i], mode, slope);
}

__global__ void affine_act_fwd_kernel(const float* __restrict__ z, const float* __restrict__ affine, const float* __restrict__ add,
                                      int N, long long V, int C, int mode, float slope, float* __restrict__ out) {
    const long long total = (long long)N * V * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int n = (int)(i / ((long long)V * C));
        const float* ab = affine + ((size_t)n * C + c) * 2;
        float v = fmaf(z[i], ab[0], ab[1]);
        if (add) v += add[i];
        out[i] = act_f(v, mode, slope);
    }
}

// grid (splits, ceil(C/64), N), 256 threads = 64 channels x 4 voxel rows
__global__ __launch_bounds__(256) void pair_stats_kernel(const float* __restrict__ a, const float* __restrict__ b, long long V,
                                                         int C, double* __restrict__ stats) {
    __shared__ float red[4][64][2];
    const int t = threadIdx.x, cl = t & 63, row = t >> 6;
    const int c = blockIdx.y * 64 + cl, n = blockIdx.z;
    const long long per = (V + gridDim.x - 1) / gridDim.x;
    const long long v0 = (long long)blockIdx.x * per, v1 = min(V, v0 + per);
    float s1 = 0.f, s2 = 0.f;
    if (c < C) {
        for (long long v = v0 + row; v < v1; v += 4) {
            const size_t o = ((size_t)n * V + v) * C + c;
            const float av = a[o];
            s1 += av;
            s2 = fmaf(av, b[o], s2);
        }
    }
    red[row][cl][0] = s1;
    red[row][cl][1] = s2;
    __syncthreads();
    if (row == 0 && c < C) {
        double d1 = 0.0, d2 = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            d1 += (double)red[r][cl][0];
            d2 += (double)red[r][cl][1];
        }
        u3d_atomic_add_f64(stats + ((size_t)n * C + c) * 2 + 0, d1);
        u3d_atomic_add_f64(stats + ((size_t)n * C + c) * 2 + 1, d2);
    }
}

inline int blocks_for(long long n) {
    long long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

}  // namespace

extern "C" int u3d_act_fwd(int device, u3d_stream_t stream, const float* x, int64_t n, int mode, float slope, float* out) {
    U3D_ENTER(device);
    U3D_REQUIRE(x && out && n > 0 && mode >= 0 && mode <= 3, "u3d_act_fwd: bad argument");
    hipLaunchKernelGGL(act_fwd_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, x, (long long)n, mode, slope, out);
    U3D_LAUNCH_CHECK();
    return 0;
}

extern "C" int u3d_act_bwd(int device, u3d_stream_t stream, const float* g, const float* y, int64_t n, int mode, float slope,
                           float* out) {
    U3D_ENTER(device);
    U3D_REQUIRE(g && y && out && n > 0 && mode >= 0 && mode <= 3, "u3d_act_bwd: bad argument");
    hipLaunchKernelGGL(act_bwd_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, g, y, (long long)n, mode, slope, out);
    U3D_LAUNCH_CHECK();
    return 0;
}

extern "C" int u3d_affine_add_act_fwd(int device, u3d_stream_t stream, const float* z, const float* affine, const float* add, int N,
                                      int64_t V, int C, int mode, float slope, float* out) {
    U3D_ENTER(device);
    U3D_REQUIRE(z && affine && out && N > 0 && V > 0 && C > 0 && mode >= 0 && mode <= 3, "u3d_affine_add_act_fwd: bad argument");
    hipLaunchKernelGGL(affine_act_fwd_kernel, dim3(blocks_for((long long)N * V * C)), dim3(256), 0, (hipStream_t)stream, z, affine,
                       add, N, (long long)V, C, mode, slope, out);
    U3D_LAUNCH_CHECK();
    return 0;
}

extern "C" int u3d_affine_act_fwd(int device, u3d_stream_t stream, const float* z, const float* affine, int N, int64_t V, int C,
                                  int mode, float slope, float* out) {
    return u3d_affine_add_act_fwd(device, stream, z, affine, nullptr, N, V, C, mode, slope, out);
}

extern "C" int u3d_pair_stats(int device, u3d_stream_t stream, const float* a, const float* b, int N, int64_t V, int C,
                              double* stats) {
    U3D_ENTER(device);
    U3D_REQUIRE(a && b && stats && N > 0 && V > 0 && C > 0, "u3d_pair_stats: bad argument");
    long long splits = (V + 1023) / 1024;
    if (splits > 128) splits = 128;
    hipLaunchKernelGGL(pair_stats_kernel, dim3((unsigned)splits, (unsigned)((C + 63) / 64), (unsigned)N), dim3(256), 0,
                       (hipStream_t)stream, a, b, (long long)V, C, stats);
    U3D_LAUNCH_CHECK();
    return 0;
}
