// u3d_loss.hip — BCEDiceLoss / DiceLoss(sigmoid) / BCEWithLogitsLoss on the logits, fused.
//
// Reference: pytorch3dunet/unet3d/losses.py — BCEDiceLoss :187-201 (= nn.BCEWithLogitsLoss() + alpha * DiceLoss()),
// DiceLoss / _AbstractDiceLoss :84-127 (sigmoid normalisation, 1 - mean_c dice_c), compute_per_channel_dice :11-37
// (dice_c = 2 * w_c * sum(p*t) / clamp(sum(p^2) + sum(t^2), eps), sums over (N, spatial) per channel, flatten :253-271).
// The stock path launches ~15 elementwise / reduction kernels plus a permute+contiguous copy and keeps 6 full-size
// temporaries for autograd; here:
//   pass 1  one read of (logits, target): 4 sums per block (BCE terms, p*t, p^2, t^2) as per-block partials in double
//   pass 2  one block: the partials summed in a fixed order (bit-reproducible), the scalar loss + per-channel gradient
//           coefficients (dL/dp_c = a_c*t + b_c*p is affine)
//   pass 3  backward: one read of (logits, target), one write of dlogits, scaled by the upstream scalar ON DEVICE
// logits / target are (N, C, V) contiguous fp32 — the reference's NCDHW, which is what the model's head writes.
#include "u3d_common.h"

namespace {

__device__ __forceinline__ void sigmoid_softplus(float x, float& p, float& sp_pos) {
    // p = sigmoid(x); sp_pos = max(x,0) + log1p(exp(-|x|)) = softplus(x)  (BCE-with-logits = softplus(x) - x*t)
    const float e = expf(-fabsf(x));
    const float r = 1.f / (1.f + e);
    p = x >= 0.f ? r : e * r;
    sp_pos = fmaxf(x, 0.f) + log1pf(e);
}

// grid (blocks_per_row, N*C), 256 threads; row = one (n, c) plane of V voxels
__global__ __launch_bounds__(256) void loss_sums_kernel(const float* __restrict__ logits, const float* __restrict__ target,
                                                        int C, long long V, int vec, double* __restrict__ sums) {
    const int row = blockIdx.y;
    const int c = row % C;
    const float* x = logits + (size_t)row * V;
    const float* t = target + (size_t)row * V;
    float s_bce = 0.f, s_pt = 0.f, s_pp = 0.f, s_tt = 0.f;
    auto acc = [&](float xv, float tv) {
        float p, sp;
        sigmoid_softplus(xv, p, sp);
        s_bce += sp - xv * tv;
        s_pt += p * tv;
        s_pp += p * p;
        s_tt += tv * tv;
    };
    const long long stride = (long long)gridDim.x * 256;
    if (vec) {
        const long long nq = V >> 2;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nq; i += stride) {
            const f32x4 xv = reinterpret_cast<const f32x4*>(x)[i];
            const f32x4 tv = reinterpret_cast<const f32x4*>(t)[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc(xv[e], tv[e]);
        }
        for (long long i = (nq << 2) + (long long)blockIdx.x * 256 + threadIdx.x; i < V; i += stride) acc(x[i], t[i]);
    } else {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < V; i += stride) acc(x[i], t[i]);
    }
    // block reduction: wave butterflies, then 4 waves through LDS (fixed order), one f64 atomic per sum and block
    __shared__ float red[4][4];
    float v[4] = {s_bce, s_pt, s_pp, s_tt};
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v[k] += __shfl_xor(v[k], m);
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (l == 0)
#pragma unroll
        for (int k = 0; k < 4; ++k) red[w][k] = v[k];
    __syncthreads();
    if (threadIdx.x < 4) {
        // per-block partial, plain store (round 6: 512 blocks adding to the same four doubles were 10 of the kernel's 15 us — a
        // same-address f64 atomic retires every 19.5 ns, tools/atomic_bench.hip; the one-block second pass sums them in a fixed order)
        const int k = threadIdx.x;
        sums[((size_t)row * gridDim.x + blockIdx.x) * 4 + k] = ((double)red[0][k] + (double)red[1][k]) + ((double)red[2][k] + (double)red[3][k]);
    }
}

// one block of 256 threads: the partials [N*C rows][per_row blocks][4] -> loss[0] and coef[2*C + 1] = {a_c, b_c}_c, k_bce
__global__ __launch_bounds__(256) void loss_finalize_kernel(const double* __restrict__ partial, int N, int per_row,
                                                            const float* __restrict__ weight, int C, double count, float w_bce,
                                                            float w_dice, float eps, float* __restrict__ loss, float* __restrict__ coef) {
    __shared__ double red[256][4];
    __shared__ double acc_bce, acc_dice;
    const int t = threadIdx.x;
    if (t == 0) acc_bce = 0.0, acc_dice = 0.0;
    for (int c = 0; c < C; ++c) {
        double v[4] = {0.0, 0.0, 0.0, 0.0};
        for (int i = t; i < N * per_row; i += 256) {
            const int n = i / per_row, bx = i - n * per_row;
            const double* pp = partial + ((size_t)(n * C + c) * per_row + bx) * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] += pp[k];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) red[t][k] = v[k];
        __syncthreads();
        for (int m = 128; m > 0; m >>= 1) {  // fixed-order tree
            if (t < m)
#pragma unroll
                for (int k = 0; k < 4; ++k) red[t][k] += red[t + m][k];
            __syncthreads();
        }
        if (t == 0) {
            const double I = red[0][1], A = red[0][2], B = red[0][3];
            const double wc = weight ? (double)weight[c] : 1.0;
            const double raw = A + B;
            const bool clamped = raw < (double)eps;  // torch.clamp(min=eps): gradient 0 through the clamped branch
            const double den = clamped ? (double)eps : raw;
            acc_bce += red[0][0];
            acc_dice += 2.0 * wc * I / den;
            // L_dice = w_dice * (1 - (1/C) sum_c dice_c);  d dice_c / dp = 2 wc t / den - (clamped ? 0 : 2 wc I * 2p / den^2)
            const double k = -(double)w_dice / C;
            coef[2 * c + 0] = (float)(k * 2.0 * wc / den);
            coef[2 * c + 1] = clamped ? 0.f : (float)(-k * 4.0 * wc * I / (den * den));
        }
        __syncthreads();
    }
    if (t == 0) {
        coef[2 * C] = (float)((double)w_bce / count);
        loss[0] = (float)((double)w_bce * acc_bce / count + (double)w_dice * (1.0 - acc_dice / C));
    }
}

// dlogits = g * [ k_bce * (p - t) + (a_c * t + b_c * p) * p * (1 - p) ],  g = *grad_out (device scalar) or 1
__global__ __launch_bounds__(256) void loss_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ target,
                                                       const float* __restrict__ coef, const float* __restrict__ grad_out,
                                                       int C, long long V, int vec, float* __restrict__ dlogits) {
    const int row = blockIdx.y;
    const int c = row % C;
    const float g = grad_out ? grad_out[0] : 1.f;
    const float a = coef[2 * c] * g, b = coef[2 * c + 1] * g, kb = coef[2 * C] * g;
    const float* x = logits + (size_t)row * V;
    const float* t = target + (size_t)row * V;
    float* d = dlogits + (size_t)row * V;
    auto one = [&](float xv, float tv) {
        float p, sp;
        sigmoid_softplus(xv, p, sp);
        return kb * (p - tv) + (a * tv + b * p) * p * (1.f - p);
    };
    const long long stride = (long long)gridDim.x * 256;
    if (vec) {
        const long long nq = V >> 2;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nq; i += stride) {
            const f32x4 xv = reinterpret_cast<const f32x4*>(x)[i];
            const f32x4 tv = reinterpret_cast<const f32x4*>(t)[i];
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = one(xv[e], tv[e]);
            reinterpret_cast<f32x4*>(d)[i] = o;
        }
        for (long long i = (nq << 2) + (long long)blockIdx.x * 256 + threadIdx.x; i < V; i += stride) d[i] = one(x[i], t[i]);
    } else {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < V; i += stride) d[i] = one(x[i], t[i]);
    }
}

inline bool rows_vec_ok(const void* a, const void* b, const void* c, long long V) {
    return V % 4 == 0 && (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15) == 0;
}

inline dim3 loss_grid(int rows, long long V) {
    // ~4 float4 per thread; keep >= ~1024 blocks in flight when the tensor is large enough
    long long per_row = (V + 4095) / 4096;
    if (per_row < 1) per_row = 1;
    if (per_row > 4096) per_row = 4096;
    return dim3((unsigned)per_row, (unsigned)rows);
}

}  // namespace

extern "C" long long u3d_bce_dice_scratch_doubles(int N, int C, int64_t V) {
    if (N <= 0 || C <= 0 || V <= 0) return 0;
    return (long long)N * C * loss_grid(N * C, V).x * 4;
}

extern "C" int u3d_bce_dice_fwd(int device, u3d_stream_t stream, const float* logits, const float* target,
                                const float* weight, int N, int C, int64_t V, float w_bce, float w_dice, float eps,
                                double* sums, float* loss, float* coef) {
    U3D_ENTER(device);
    U3D_REQUIRE(logits && target && sums && loss && coef && N > 0 && C > 0 && V > 0, "u3d_bce_dice_fwd: bad argument");
    U3D_REQUIRE((long long)N * C < 65536, "u3d_bce_dice_fwd: N*C must be < 65536");
    hipStream_t st = (hipStream_t)stream;
    const int vec = rows_vec_ok(logits, target, logits, V) ? 1 : 0;
    const dim3 grid = loss_grid(N * C, V);
    hipLaunchKernelGGL(loss_sums_kernel, grid, dim3(256), 0, st, logits, target, C, (long long)V, vec, sums);
    U3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, st, sums, N, (int)grid.x, weight, C, (double)N * C * (double)V,
                       w_bce, w_dice, eps, loss, coef);
    U3D_LAUNCH_CHECK();
    return 0;
}

extern "C" int u3d_bce_dice_bwd(int device, u3d_stream_t stream, const float* logits, const float* target,
                                const float* coef, const float* grad_out, int N, int C, int64_t V, float* dlogits) {
    U3D_ENTER(device);
    U3D_REQUIRE(logits && target && coef && dlogits && N > 0 && C > 0 && V > 0, "u3d_bce_dice_bwd: bad argument");
    U3D_REQUIRE((long long)N * C < 65536, "u3d_bce_dice_bwd: N*C must be < 65536");
    const int vec = rows_vec_ok(logits, target, dlogits, V) ? 1 : 0;
    hipLaunchKernelGGL(loss_bwd_kernel, loss_grid(N * C, V), dim3(256), 0, (hipStream_t)stream, logits, target, coef,
                       grad_out, C, (long long)V, vec, dlogits);
    U3D_LAUNCH_CHECK();
    return 0;
}
