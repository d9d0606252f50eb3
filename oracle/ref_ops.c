/*
 * TEST INFRASTRUCTURE ONLY — plain-C restatement of the operators on the reference hot path, independent of
 * torch, used to cross-check the torch-functional oracle (oracle/unet3d_oracle.py) and the HIP kernels at small
 * sizes.  Layout NCDHW fp32 like the reference; all reductions accumulate in double.
 *
 * The reference (wolny/pytorch-3dunet 1.9.6) calls these through torch.nn (third-party, unpinned `torch`,
 * pyproject.toml:17); the algorithms restated here are the published operator definitions:
 *   nn.Conv3d(k=3,padding=1,bias=False)  buildingblocks.py:56     cross-correlation, zero padding
 *   nn.GroupNorm(G,C,eps=1e-5)           buildingblocks.py:75     per-(n,group) mean / biased variance, affine
 *   nn.MaxPool3d(2)                      buildingblocks.py:356    stride 2, floor, first max in scan order
 *   F.interpolate(mode="nearest")        buildingblocks.py:614    src = min(floor(dst*float(in/out)), in-1)
 * Only tests/ may link this.  Build: `make -C oracle` -> oracle/_build/libref_ops.so
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#define IDX5(n, c, z, y, x, C, D, H, W) ((((size_t)(n) * (C) + (c)) * (D) + (z)) * (H) + (y)) * (W) + (x)

void ref_conv3d_fwd(const float* x, const float* w, float* y, int N, int C, int K, int D, int H, int W) {
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k)
            for (int z = 0; z < D; ++z)
                for (int yy = 0; yy < H; ++yy)
                    for (int xx = 0; xx < W; ++xx) {
                        double acc = 0.0;
                        for (int c = 0; c < C; ++c)
                            for (int dz = 0; dz < 3; ++dz)
                                for (int dy = 0; dy < 3; ++dy)
                                    for (int dx = 0; dx < 3; ++dx) {
                                        int zi = z + dz - 1, yi = yy + dy - 1, xi = xx + dx - 1;
                                        if (zi < 0 || zi >= D || yi < 0 || yi >= H || xi < 0 || xi >= W) continue;
                                        acc += (double)w[(((size_t)k * C + c) * 27) + dz * 9 + dy * 3 + dx] *
                                               (double)x[IDX5(n, c, zi, yi, xi, C, D, H, W)];
                                    }
                        y[IDX5(n, k, z, yy, xx, K, D, H, W)] = (float)acc;
                    }
}

/* dx[n,c,u] = sum_{k,t} w[k,c,t] * dy[n,k,u-(t-1)] */
void ref_conv3d_dgrad(const float* dy, const float* w, float* dx, int N, int C, int K, int D, int H, int W) {
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c)
            for (int z = 0; z < D; ++z)
                for (int yy = 0; yy < H; ++yy)
                    for (int xx = 0; xx < W; ++xx) {
                        double acc = 0.0;
                        for (int k = 0; k < K; ++k)
                            for (int dz = 0; dz < 3; ++dz)
                                for (int dyy = 0; dyy < 3; ++dyy)
                                    for (int dxx = 0; dxx < 3; ++dxx) {
                                        int zi = z - (dz - 1), yi = yy - (dyy - 1), xi = xx - (dxx - 1);
                                        if (zi < 0 || zi >= D || yi < 0 || yi >= H || xi < 0 || xi >= W) continue;
                                        acc += (double)w[(((size_t)k * C + c) * 27) + dz * 9 + dyy * 3 + dxx] *
                                               (double)dy[IDX5(n, k, zi, yi, xi, K, D, H, W)];
                                    }
                        dx[IDX5(n, c, z, yy, xx, C, D, H, W)] = (float)acc;
                    }
}

/* dw[k,c,t] = sum_{n,v} dy[n,k,v] * x[n,c,v+(t-1)] */
void ref_conv3d_wgrad(const float* x, const float* dy, float* dw, int N, int C, int K, int D, int H, int W) {
    for (int k = 0; k < K; ++k)
        for (int c = 0; c < C; ++c)
            for (int dz = 0; dz < 3; ++dz)
                for (int dyy = 0; dyy < 3; ++dyy)
                    for (int dxx = 0; dxx < 3; ++dxx) {
                        double acc = 0.0;
                        for (int n = 0; n < N; ++n)
                            for (int z = 0; z < D; ++z)
                                for (int yy = 0; yy < H; ++yy)
                                    for (int xx = 0; xx < W; ++xx) {
                                        int zi = z + dz - 1, yi = yy + dyy - 1, xi = xx + dxx - 1;
                                        if (zi < 0 || zi >= D || yi < 0 || yi >= H || xi < 0 || xi >= W) continue;
                                        acc += (double)dy[IDX5(n, k, z, yy, xx, K, D, H, W)] *
                                               (double)x[IDX5(n, c, zi, yi, xi, C, D, H, W)];
                                    }
                        dw[(((size_t)k * C + c) * 27) + dz * 9 + dyy * 3 + dxx] = (float)acc;
                    }
}

void ref_groupnorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                       int N, int C, int G, size_t V, float eps) {
    int cpg = C / G;
    for (int n = 0; n < N; ++n)
        for (int g = 0; g < G; ++g) {
            const float* p = x + ((size_t)n * C + (size_t)g * cpg) * V;
            size_t m = (size_t)cpg * V;
            double s = 0.0;
            for (size_t i = 0; i < m; ++i) s += p[i];
            double mu = s / (double)m, ss = 0.0;
            for (size_t i = 0; i < m; ++i) ss += ((double)p[i] - mu) * ((double)p[i] - mu);
            double r = 1.0 / sqrt(ss / (double)m + (double)eps);
            mean[n * G + g] = (float)mu;
            rstd[n * G + g] = (float)r;
            for (int c = 0; c < cpg; ++c) {
                int ch = g * cpg + c;
                for (size_t i = 0; i < V; ++i)
                    y[((size_t)n * C + ch) * V + i] = (float)((((double)p[c * V + i] - mu) * r) * gamma[ch] + beta[ch]);
            }
        }
}

void ref_groupnorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                       float* dx, float* dgamma, float* dbeta, int N, int C, int G, size_t V) {
    int cpg = C / G;
    for (int c = 0; c < C; ++c) {
        double dg = 0.0, db = 0.0;
        for (int n = 0; n < N; ++n) {
            double mu = mean[n * G + c / cpg], r = rstd[n * G + c / cpg];
            for (size_t i = 0; i < V; ++i) {
                size_t o = ((size_t)n * C + c) * V + i;
                dg += (double)dy[o] * ((double)x[o] - mu) * r;
                db += (double)dy[o];
            }
        }
        dgamma[c] = (float)dg;
        dbeta[c] = (float)db;
    }
    for (int n = 0; n < N; ++n)
        for (int g = 0; g < G; ++g) {
            double mu = mean[n * G + g], r = rstd[n * G + g];
            size_t m = (size_t)cpg * V;
            double a = 0.0, b = 0.0;
            for (int c = 0; c < cpg; ++c) {
                int ch = g * cpg + c;
                for (size_t i = 0; i < V; ++i) {
                    size_t o = ((size_t)n * C + ch) * V + i;
                    double t = (double)dy[o] * gamma[ch];
                    a += t;
                    b += t * ((double)x[o] - mu) * r;
                }
            }
            for (int c = 0; c < cpg; ++c) {
                int ch = g * cpg + c;
                for (size_t i = 0; i < V; ++i) {
                    size_t o = ((size_t)n * C + ch) * V + i;
                    double xh = ((double)x[o] - mu) * r;
                    dx[o] = (float)(r * ((double)dy[o] * gamma[ch] - a / (double)m - xh * b / (double)m));
                }
            }
        }
}

void ref_maxpool2_fwd(const float* x, float* y, uint8_t* idx, int N, int C, int D, int H, int W) {
    int D2 = D / 2, H2 = H / 2, W2 = W / 2;
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c)
            for (int z = 0; z < D2; ++z)
                for (int yy = 0; yy < H2; ++yy)
                    for (int xx = 0; xx < W2; ++xx) {
                        float best = -INFINITY;
                        int bi = 0;
                        for (int k = 0; k < 8; ++k) {
                            float v = x[IDX5(n, c, 2 * z + (k >> 2), 2 * yy + ((k >> 1) & 1), 2 * xx + (k & 1), C, D, H, W)];
                            if (v > best || v != v) {
                                best = v;
                                bi = k;
                            }
                        }
                        y[IDX5(n, c, z, yy, xx, C, D2, H2, W2)] = best;
                        idx[IDX5(n, c, z, yy, xx, C, D2, H2, W2)] = (uint8_t)bi;
                    }
}

static int nearest_src(int dst, int in, int out) {
    float scale = (float)in / (float)out;
    int s = (int)floorf((float)dst * scale);
    return s < in - 1 ? s : in - 1;
}

void ref_upsample_nearest(const float* x, float* y, int N, int C, int D1, int H1, int W1, int D, int H, int W) {
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c)
            for (int z = 0; z < D; ++z)
                for (int yy = 0; yy < H; ++yy)
                    for (int xx = 0; xx < W; ++xx)
                        y[IDX5(n, c, z, yy, xx, C, D, H, W)] =
                            x[IDX5(n, c, nearest_src(z, D1, D), nearest_src(yy, H1, H), nearest_src(xx, W1, W), C, D1, H1, W1)];
}
