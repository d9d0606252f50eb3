// u3d_optim.hip — the Adam update of ALL parameters of a model in one launch.
//
// Replaces, inside the training step, what the reference's optimizer does after loss.backward() (trainer.py:246 `self.optimizer.step()`
// with the optimizer of create_optimizer, utils.py:246-316: torch.optim.Adam(params, lr, betas, weight_decay)).  torch's multi-tensor
// form is 8 launches / 0.17 ms per step on the 44 parameters (4.08 M elements) of UNet3D f_maps=32, its `fused=True` form 3 launches /
// 0.21 ms (profiles/r06a_step_launches.txt, r06c) — for 16 MB of parameters, i.e. 114 MB of traffic = ~25 us at HBM rate.  Here one
// thread owns four consecutive elements of one parameter: a binary search over the descriptor table (first = padded running element
// offset), 16-byte accesses where the four pointers allow, the update in torch's operation order:
//   g  = grad + weight_decay * p                       (torch/optim/adam.py _single_tensor_adam: grad.add(param, alpha=weight_decay))
//   m  = m + (1 - beta1) * (g - m)                     (exp_avg.lerp_(grad, 1 - beta1))
//   v  = beta2 * v + (1 - beta2) * g * g               (exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2))
//   p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)  (denom = (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps); addcdiv_)
// with bc1 = 1 - beta1^step, bc2 = 1 - beta2^step formed on the host in double like torch does.
#include "u3d_common.h"

namespace {
__global__ __launch_bounds__(256) void adam_step_kernel(const u3d_adam_desc_t* __restrict__ descs, int n, long long total4, float wd,
                                                        float one_m_b1, float b2, float one_m_b2, float step_size, float bc2_sqrt,
                                                        float eps) {
#pragma clang fp contract(off)  // (rounded products and sums like the element-wise ATen kernels)
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total4; q += (long long)gridDim.x * 256) {
        const long long e0 = q << 2;
        int lo = 0, hi = n - 1;  // last descriptor with first <= e0
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (descs[mid].first <= e0) lo = mid; else hi = mid - 1;
        }
        const u3d_adam_desc_t d = descs[lo];
        const long long i0 = e0 - d.first;
        if (i0 >= d.numel) continue;  // padding quads behind a parameter whose size is not a multiple of 4
        const int cnt = (int)(d.numel - i0 < 4 ? d.numel - i0 : 4);
        float p[4], g[4], m[4], v[4];
        const bool vec = cnt == 4 && ((((uintptr_t)d.p | (uintptr_t)d.g | (uintptr_t)d.m | (uintptr_t)d.v) & 15) == 0);
        if (vec) {
            const f32x4 P = *reinterpret_cast<const f32x4*>(d.p + i0), G = *reinterpret_cast<const f32x4*>(d.g + i0);
            const f32x4 M = *reinterpret_cast<const f32x4*>(d.m + i0), V = *reinterpret_cast<const f32x4*>(d.v + i0);
#pragma unroll
            for (int e = 0; e < 4; ++e) p[e] = P[e], g[e] = G[e], m[e] = M[e], v[e] = V[e];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool ok = e < cnt;
                p[e] = ok ? d.p[i0 + e] : 0.f, g[e] = ok ? d.g[i0 + e] : 0.f, m[e] = ok ? d.m[i0 + e] : 0.f, v[e] = ok ? d.v[i0 + e] : 0.f;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gg = wd != 0.f ? g[e] + wd * p[e] : g[e];
            // (ATen lerp: self + w*(end - self) for w < 0.5, else end - (end - self)*(1 - w); addcmul: self + value*t1*t2 left to right)
            m[e] = one_m_b1 < 0.5f ? m[e] + one_m_b1 * (gg - m[e]) : gg - (gg - m[e]) * (1.f - one_m_b1);
            v[e] = b2 * v[e] + (one_m_b2 * gg) * gg;
            const float denom = sqrtf(v[e]) / bc2_sqrt + eps;
            p[e] = p[e] - step_size * (m[e] / denom);
        }
        if (vec) {
            *reinterpret_cast<f32x4*>(d.p + i0) = f32x4{p[0], p[1], p[2], p[3]};
            *reinterpret_cast<f32x4*>(d.m + i0) = f32x4{m[0], m[1], m[2], m[3]};
            *reinterpret_cast<f32x4*>(d.v + i0) = f32x4{v[0], v[1], v[2], v[3]};
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (e < cnt) d.p[i0 + e] = p[e], d.m[i0 + e] = m[e], d.v[i0 + e] = v[e];
        }
    }
}
}  // namespace

extern "C" int u3d_adam_step(int device, u3d_stream_t stream, const u3d_adam_desc_t* descs_device, int n, int64_t total_padded,
                             double lr, double beta1, double beta2, double eps, double weight_decay, int64_t step) {
    U3D_ENTER(device);
    U3D_REQUIRE(descs_device && n > 0 && total_padded > 0 && total_padded % 4 == 0 && step >= 1 && lr >= 0.0 && beta1 >= 0.0 && beta1 < 1.0 &&
                    beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0, "u3d_adam_step: bad argument");
    double p1 = 1.0, p2 = 1.0;  // beta^step in double (torch: python floats)
    {
        double b1 = beta1, b2 = beta2;
        for (int64_t k = step; k > 0; k >>= 1) {
            if (k & 1) p1 *= b1, p2 *= b2;
            b1 *= b1, b2 *= b2;
        }
    }
    const double bc1 = 1.0 - p1, bc2 = 1.0 - p2;
    const long long total4 = total_padded >> 2;
    long long blocks = (total4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, descs_device, n, total4,
                       (float)weight_decay, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)(lr / bc1),
                       (float)sqrt(bc2), (float)eps);
    U3D_LAUNCH_CHECK();
    return 0;
}
