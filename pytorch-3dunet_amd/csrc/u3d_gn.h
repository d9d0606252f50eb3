// u3d_gn.h — the GroupNorm-backward reduction (sums -> dgamma, dbeta, coefficient table), shared by gn_bwd_finalize_kernel
// (csrc/u3d_ops.hip) and the extra block of wgrad_reduce_kernel (csrc/u3d_conv.hip, u3d_conv3d_wgrad_job: the finalize launch of a layer
// rides in the launch that reduces its weight gradient — 13 single-block launches of ~4.8 us less per bench step, round 6).
#pragma once
#include "u3d_common.h"

// Sum of the replica rows of one statistics entry (u3d_conv3d_ex_reps): p[0] + p[stride] + .. + p[(reps - 1) * stride], ascending, with
// eight loads in flight (a dependent chain of `reps` global loads per thread made a 16-row finalize launch slower than the atomics it saves).
__device__ __forceinline__ double u3d_sum_replicas(const double* __restrict__ p, size_t stride, int reps) {
    double v = p[0];
    for (int r0 = 1; r0 < reps; r0 += 8) {
        double t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = r0 + j < reps ? p[(size_t)(r0 + j) * stride] : 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) v += t[j];
    }
    return v;
}

// One block of 256 threads; shb = dynamic LDS of (par ? 2 * 16 * N * C + 16 * N * G : staged ? 16 * N * C : 0) bytes.
__device__ inline void u3d_gn_bwd_finalize_body(const double* __restrict__ gs, const float* __restrict__ mean_rstd,
                                                const float* __restrict__ gamma, int N, int C, int G, double count, int staged, int par,
                                                float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ coef,
                                                const double* __restrict__ gs_hi, int C0, float hi_scale, float* __restrict__ coef_hi,
                                                double* shb, int reps = 1, int reps_hi = 1) {
    // reps > 1 (staged paths only): gs holds `reps` replica rows [reps][N][C0 or C][2] whose sum is the table (u3d_conv3d_ex_reps)

#pragma clang fp contract(off)  // (both paths: products rounded, then added in channel order — identical results)
    const double* src = gs;
    if (staged) {
        if (gs_hi) {
            // (u3d_gn_bwd_finalize_split: the sums of channels [0, C0) and [C0, C) arrive as two tables [N][C0][2] / [N][C - C0][2] —
            // the skip-half and low-res data-gradient kernels of a sub-pixel decoder level each write their own)
            const int C1 = C - C0;
            for (int i = threadIdx.x; i < N * C * 2; i += blockDim.x) {
                const int e = i & 1, nc = i >> 1, n = nc / C, c = nc - n * C;
                shb[i] = c < C0 ? u3d_sum_replicas(gs + ((size_t)n * C0 + c) * 2 + e, (size_t)N * C0 * 2, reps)
                                : u3d_sum_replicas(gs_hi + ((size_t)n * C1 + (c - C0)) * 2 + e, (size_t)N * C1 * 2, reps_hi);
            }
        } else {
            for (int i = threadIdx.x; i < N * C * 2; i += blockDim.x) shb[i] = u3d_sum_replicas(gs + i, (size_t)N * C * 2, reps);
        }
        __syncthreads();
        src = shb;
    }
    const int cpg = C / G;
    const double m = count * cpg;
    double* pa = shb + 2 * (size_t)N * C;  // (par only)
    double* pb = pa + (size_t)N * C;
    double* qr = pb + (size_t)N * C;       // [N*G][2]
    if (par) {
        for (int i = threadIdx.x; i < N * C; i += blockDim.x) {
            const int n = i / C, c = i - n * C, g = c / cpg;
            const double mean = (double)mean_rstd[((size_t)n * G + g) * 2], rstd = (double)mean_rstd[((size_t)n * G + g) * 2 + 1];
            const double S1 = src[(size_t)i * 2], S2 = src[(size_t)i * 2 + 1], gm = (double)gamma[c];
            pa[i] = gm * S1;
            pb[i] = gm * rstd * (S2 - mean * S1);
        }
        __syncthreads();
    }
    for (int pair = threadIdx.x; pair < N * G; pair += blockDim.x) {
        const int n = pair / G, g = pair - n * G;
        const double mean = (double)mean_rstd[(size_t)pair * 2], rstd = (double)mean_rstd[(size_t)pair * 2 + 1];
        double A = 0.0, B = 0.0;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
            if (par) {
                A += pa[(size_t)n * C + c];
                B += pb[(size_t)n * C + c];
            } else {
                const double S1 = src[((size_t)n * C + c) * 2], S2 = src[((size_t)n * C + c) * 2 + 1];
                const double gm = (double)gamma[c];
                A += gm * S1;
                B += gm * rstd * (S2 - mean * S1);
            }
        }
        const double q = -rstd * rstd * B / m;
        const double r = -rstd * A / m + rstd * rstd * mean * B / m;
        if (par) {
            qr[(size_t)pair * 2] = q;
            qr[(size_t)pair * 2 + 1] = r;
        } else {
            for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
                coef[((size_t)n * 3 + 0) * C + c] = (float)(rstd * (double)gamma[c]);
                coef[((size_t)n * 3 + 1) * C + c] = (float)q;
                coef[((size_t)n * 3 + 2) * C + c] = (float)r;
            }
        }
    }
    if (par) {
        __syncthreads();
        for (int i = threadIdx.x; i < N * C; i += blockDim.x) {
            const int n = i / C, c = i - n * C, pair = n * G + c / cpg;
            const double rstd = (double)mean_rstd[(size_t)pair * 2 + 1];
            const float fp = (float)(rstd * (double)gamma[c]), fq = (float)qr[(size_t)pair * 2], fr = (float)qr[(size_t)pair * 2 + 1];
            coef[((size_t)n * 3 + 0) * C + c] = fp;
            coef[((size_t)n * 3 + 1) * C + c] = fq;
            coef[((size_t)n * 3 + 2) * C + c] = fr;
            if (coef_hi && c >= C0) {
                // compact table of the upper channels with (q, r) scaled: a low-res voxel of an exact 2x upsampling stands for 8 children
                const int C1 = C - C0;
                coef_hi[((size_t)n * 3 + 0) * C1 + (c - C0)] = fp;
                coef_hi[((size_t)n * 3 + 1) * C1 + (c - C0)] = fq * hi_scale;
                coef_hi[((size_t)n * 3 + 2) * C1 + (c - C0)] = fr * hi_scale;
            }
        }
    }
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const int g = c / cpg;
        double dg = 0.0, db = 0.0;
        for (int n = 0; n < N; ++n) {
            const double mean = (double)mean_rstd[((size_t)n * G + g) * 2], rstd = (double)mean_rstd[((size_t)n * G + g) * 2 + 1];
            const double S1 = src[((size_t)n * C + c) * 2], S2 = src[((size_t)n * C + c) * 2 + 1];
            dg += rstd * (S2 - mean * S1);
            db += S1;
        }
        dgamma[c] = (float)dg;
        dbeta[c] = (float)db;
    }
}
