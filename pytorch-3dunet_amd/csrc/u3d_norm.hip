// u3d_norm.hip — the remaining characters of the layer-order mini language (reference
// pytorch3dunet/unet3d/buildingblocks.py:10-96 create_conv): 'b' BatchNorm3d (:78-88), conv WITH bias when the layer has no
// norm ('cr', 'cl', 'ce', 'c': bias = not ('g' in order or 'b' in order), :54-55), 'd' / 'D' dropout (:89-92).
//
// BatchNorm reuses the GroupNorm machinery of u3d_ops.hip: the producers' epilogues (or u3d_chan_stats) deliver per-(sample,
// channel) sums (sum x, sum x^2); u3d_bn_finalize folds them over the batch into per-channel mean / biased variance (what
// nn.BatchNorm3d normalises with in training mode), updates running_mean / running_var exactly like ATen (momentum, UNBIASED
// variance), and writes the same per-(n, c) affine table (a, b) the convolutions apply while staging — identical for every n.
// In eval mode the table comes from the running statistics.  Backward: u3d_bn_bwd_finalize turns the data gradient's sums
// (sum dg, sum dg*x) into dgamma, dbeta and the (p, q, r) coefficients of dx = p*dg + q*x + r consumed by u3d_gn_bwd_apply*:
//     training: p = gamma*rstd,  k = rstd*(S2 - mu*S1)/M,  q = -p*k*rstd,  r = -p*S1/M + p*k*rstd*mu      (M = N * voxels)
//     eval:     p = gamma*rstd (running),  q = r = 0                                   dgamma = rstd*(S2 - mu*S1), dbeta = S1
// Bias: a layer without a norm is "post-norm with the constant affine (1, bias)": u3d_bias_table builds that table,
// u3d_bias_grad reduces the sums of the pre-activation gradient.  Dropout multiplies by a mask drawn by the caller.
#include "u3d_common.h"

namespace {

__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ st0, int C0, double sc0,
                                                          const double* __restrict__ st1, int C1, double sc1, int N, double count,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                          int training, float momentum, float* __restrict__ running_mean,
                                                          float* __restrict__ running_var, float* __restrict__ affine,
                                                          float* __restrict__ mean_rstd) {
    const int C = C0 + C1;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double mean, rstd;
    if (training) {
        double s = 0.0, ss = 0.0;
        for (int n = 0; n < N; ++n) {  // fixed order
            if (c < C0) {
                s += sc0 * st0[((size_t)n * C0 + c) * 2];
                ss += sc0 * st0[((size_t)n * C0 + c) * 2 + 1];
            } else {
                s += sc1 * st1[((size_t)n * C1 + (c - C0)) * 2];
                ss += sc1 * st1[((size_t)n * C1 + (c - C0)) * 2 + 1];
            }
        }
        const double m = count * N;
        mean = s / m;
        double var = ss / m - mean * mean;
        if (var < 0.0) var = 0.0;
        rstd = 1.0 / sqrt(var + (double)eps);
        if (running_mean && running_var) {  // ATen batch_norm_update_stats: unbiased variance in the running estimate
            const double unbiased = m > 1.0 ? var * m / (m - 1.0) : var;
            running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mean);
            running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unbiased);
        }
    } else {
        mean = (double)running_mean[c];
        rstd = 1.0 / sqrt((double)running_var[c] + (double)eps);
    }
    mean_rstd[2 * c] = (float)mean;
    mean_rstd[2 * c + 1] = (float)rstd;
    const double a = rstd * (double)gamma[c];
    const double b = (double)beta[c] - mean * a;
    for (int n = 0; n < N; ++n) {
        affine[((size_t)n * C + c) * 2] = (float)a;
        affine[((size_t)n * C + c) * 2 + 1] = (float)b;
    }
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const double* __restrict__ gs, const float* __restrict__ mean_rstd,
                                                              const float* __restrict__ gamma, int N, int C, double count,
                                                              int training, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              float* __restrict__ coef) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double S1 = 0.0, S2 = 0.0;
    for (int n = 0; n < N; ++n) {
        S1 += gs[((size_t)n * C + c) * 2];
        S2 += gs[((size_t)n * C + c) * 2 + 1];
    }
    const double mean = (double)mean_rstd[2 * c], rstd = (double)mean_rstd[2 * c + 1];
    const double m = count * N;
    const double p = rstd * (double)gamma[c];
    double q = 0.0, r = 0.0;
    if (training) {
        const double k = rstd * (S2 - mean * S1) / m;
        q = -p * k * rstd;
        r = -p * S1 / m + p * k * rstd * mean;
    }
    dgamma[c] = (float)(rstd * (S2 - mean * S1));
    dbeta[c] = (float)S1;
    for (int n = 0; n < N; ++n) {
        coef[((size_t)n * 3 + 0) * C + c] = (float)p;
        coef[((size_t)n * 3 + 1) * C + c] = (float)q;
        coef[((size_t)n * 3 + 2) * C + c] = (float)r;
    }
}

__global__ void bias_table_kernel(const float* __restrict__ bias, int N, int C, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    out[2 * (size_t)i] = 1.f;
    out[2 * (size_t)i + 1] = bias[i % C];
}

__global__ void bias_grad_kernel(const double* __restrict__ stats, int N, int C, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0;
    for (int n = 0; n < N; ++n) s += stats[((size_t)n * C + c) * 2];
    out[c] = (float)s;
}

__global__ void mul_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n, float* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = a[i] * b[i];
}

}  // namespace

extern "C" int u3d_bn_finalize(int device, u3d_stream_t stream, const double* stats0, int C0, double scale0, const double* stats1,
                               int C1, double scale1, int N, double count, const float* gamma, const float* beta, float eps,
                               int training, float momentum, float* running_mean, float* running_var, float* affine,
                               float* mean_rstd) {
    U3D_ENTER(device);
    U3D_REQUIRE(gamma && beta && affine && mean_rstd && N > 0 && C0 > 0 && C1 >= 0 && count > 0, "u3d_bn_finalize: bad argument");
    U3D_REQUIRE(training ? (stats0 && (C1 == 0 || stats1)) : (running_mean && running_var),
                "u3d_bn_finalize: training needs the batch sums, eval the running statistics");
    const int C = C0 + C1;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, stats0, C0, scale0, stats1, C1,
                       scale1, N, count, gamma, beta, eps, training, momentum, running_mean, running_var, affine, mean_rstd);
    U3D_LAUNCH_CHECK();
    return 0;
}

extern "C" int u3d_bn_bwd_finalize(int device, u3d_stream_t stream, const double* gstats, const float* mean_rstd, const float* gamma,
                                   int N, int C, double count, int training, float* dgamma, float* dbeta, float* coef) {
    U3D_ENTER(device);
    U3D_REQUIRE(gstats && mean_rstd && gamma && dgamma && dbeta && coef && N > 0 && C > 0 && count > 0, "u3d_bn_bwd_finalize: bad argument");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gstats, mean_rstd, gamma, N,
                       C, count, training, dgamma, dbeta, coef);
    U3D_LAUNCH_CHECK();
    return 0;
}

extern "C" int u3d_bias_table(int device, u3d_stream_t stream, const float* bias, int N, int C, float* affine) {
    U3D_ENTER(device);
    U3D_REQUIRE(bias && affine && N > 0 && C > 0, "u3d_bias_table: bad argument");
    hipLaunchKernelGGL(bias_table_kernel, dim3((N * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, bias, N, C, affine);
    U3D_LAUNCH_CHECK();
    return 0;
}

extern "C" int u3d_bias_grad(int device, u3d_stream_t stream, const double* stats, int N, int C, float* dbias) {
    U3D_ENTER(device);
    U3D_REQUIRE(stats && dbias && N > 0 && C > 0, "u3d_bias_grad: bad argument");
    hipLaunchKernelGGL(bias_grad_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, stats, N, C, dbias);
    U3D_LAUNCH_CHECK();
    return 0;
}

extern "C" int u3d_mul(int device, u3d_stream_t stream, const float* a, const float* b, int64_t n, float* out) {
    U3D_ENTER(device);
    U3D_REQUIRE(a && b && out && n > 0, "u3d_mul: bad argument");
    long long blocks = (n + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(mul_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, b, (long long)n, out);
    U3D_LAUNCH_CHECK();
    return 0;
}
