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
 *   (round 2, at the end of the file: nn.BatchNorm3d, F.interpolate trilinear / area, nn.ConvTranspose3d)
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

/* ---- operators added with the round-2 coverage (same conventions) -------------------------------------------------------
 *   nn.BatchNorm3d(C, eps=1e-5, momentum=0.1)   buildingblocks.py:78-88   training: per-channel mean / biased variance over
 *                                               (N, D, H, W), running estimates <- (1-m)*running + m*batch (UNBIASED variance);
 *                                               eval: the running estimates
 *   F.interpolate(mode="trilinear")             buildingblocks.py:612-614 align_corners=False, scale = in/out:
 *                                               src = max(scale*(dst+0.5)-0.5, 0), i0 = floor(src), i1 = min(i0+1, in-1)
 *   F.interpolate(mode="area")                  = adaptive_avg_pool3d: window [floor(o*in/out), ceil((o+1)*in/out))
 *   nn.ConvTranspose3d(k=3, stride=2, padding=1, bias=False)  buildingblocks.py:617-664  out = 2*in - 1 per dimension,
 *                                               y[2i + t - 1] += x[i] * w[ci][co][t]   (weight layout (Cin, Cout, 3,3,3))      */
void ref_batchnorm_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var, float* y,
                       int N, int C, size_t V, float eps, float momentum, int training) {
    for (int c = 0; c < C; ++c) {
        double mean, var;
        if (training) {
            double s = 0.0, ss = 0.0;
            for (int n = 0; n < N; ++n)
                for (size_t v = 0; v < V; ++v) {
                    const double t = x[((size_t)n * C + c) * V + v];
                    s += t;
                    ss += t * t;
                }
            const double m = (double)N * (double)V;
            mean = s / m;
            var = ss / m - mean * mean;
            if (var < 0.0) var = 0.0;
            if (running_mean && running_var) {
                const double unbiased = m > 1.0 ? var * m / (m - 1.0) : var;
                running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
                running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
            }
        } else {
            mean = running_mean[c];
            var = running_var[c];
        }
        const double rstd = 1.0 / sqrt(var + (double)eps);
        for (int n = 0; n < N; ++n)
            for (size_t v = 0; v < V; ++v) {
                const size_t o = ((size_t)n * C + c) * V + v;
                y[o] = (float)(((double)x[o] - mean) * rstd * gamma[c] + beta[c]);
            }
    }
}

static void linear_src(int dst, int in, int out, int* i0, int* i1, float* w1) {
    const float scale = (float)in / (float)out;
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    *i0 = (int)src;
    *i1 = *i0 + (*i0 < in - 1 ? 1 : 0);
    *w1 = src - (float)*i0;
}

void ref_upsample_trilinear(const float* x, float* y, int N, int C, int D1, int H1, int W1, int D, int H, int W) {
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c)
            for (int z = 0; z < D; ++z) {
                int z0, z1;
                float wz;
                linear_src(z, D1, D, &z0, &z1, &wz);
                for (int yy = 0; yy < H; ++yy) {
                    int y0, y1;
                    float wy;
                    linear_src(yy, H1, H, &y0, &y1, &wy);
                    for (int xx = 0; xx < W; ++xx) {
                        int x0, x1;
                        float wx;
                        linear_src(xx, W1, W, &x0, &x1, &wx);
                        double acc = 0.0;
                        for (int a = 0; a < 2; ++a)
                            for (int b = 0; b < 2; ++b)
                                for (int d = 0; d < 2; ++d) {
                                    const double wgt = (a ? wz : 1.f - wz) * (double)(b ? wy : 1.f - wy) * (double)(d ? wx : 1.f - wx);
                                    acc += wgt * x[IDX5(n, c, a ? z1 : z0, b ? y1 : y0, d ? x1 : x0, C, D1, H1, W1)];
                                }
                        y[IDX5(n, c, z, yy, xx, C, D, H, W)] = (float)acc;
                    }
                }
            }
}

void ref_upsample_area(const float* x, float* y, int N, int C, int D1, int H1, int W1, int D, int H, int W) {
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c)
            for (int z = 0; z < D; ++z) {
                const int zs = (int)(((long long)z * D1) / D), ze = (int)((((long long)z + 1) * D1 + D - 1) / D);
                for (int yy = 0; yy < H; ++yy) {
                    const int ys = (int)(((long long)yy * H1) / H), ye = (int)((((long long)yy + 1) * H1 + H - 1) / H);
                    for (int xx = 0; xx < W; ++xx) {
                        const int xs = (int)(((long long)xx * W1) / W), xe = (int)((((long long)xx + 1) * W1 + W - 1) / W);
                        double acc = 0.0;
                        for (int a = zs; a < ze; ++a)
                            for (int b = ys; b < ye; ++b)
                                for (int d = xs; d < xe; ++d) acc += x[IDX5(n, c, a, b, d, C, D1, H1, W1)];
                        y[IDX5(n, c, z, yy, xx, C, D, H, W)] = (float)(acc / ((double)(ze - zs) * (ye - ys) * (xe - xs)));
                    }
                }
            }
}

void ref_conv_transpose3d_fwd(const float* x, const float* w, float* y, int N, int Cin, int Cout, int D1, int H1, int W1) {
    const int D = 2 * D1 - 1, H = 2 * H1 - 1, W = 2 * W1 - 1;
    for (int n = 0; n < N; ++n)
        for (int co = 0; co < Cout; ++co)
            for (int z = 0; z < D; ++z)
                for (int yy = 0; yy < H; ++yy)
                    for (int xx = 0; xx < W; ++xx) {
                        double acc = 0.0;
                        for (int tz = 0; tz < 3; ++tz) {
                            const int iz2 = z + 1 - tz;  /* z = 2*iz + tz - 1 */
                            if (iz2 < 0 || (iz2 & 1) || iz2 / 2 >= D1) continue;
                            for (int ty = 0; ty < 3; ++ty) {
                                const int iy2 = yy + 1 - ty;
                                if (iy2 < 0 || (iy2 & 1) || iy2 / 2 >= H1) continue;
                                for (int tx = 0; tx < 3; ++tx) {
                                    const int ix2 = xx + 1 - tx;
                                    if (ix2 < 0 || (ix2 & 1) || ix2 / 2 >= W1) continue;
                                    for (int ci = 0; ci < Cin; ++ci)
                                        acc += (double)x[IDX5(n, ci, iz2 / 2, iy2 / 2, ix2 / 2, Cin, D1, H1, W1)] *
                                               w[((((size_t)ci * Cout + co) * 3 + tz) * 3 + ty) * 3 + tx];
                                }
                            }
                        }
                        y[IDX5(n, co, z, yy, xx, Cout, D, H, W)] = (float)acc;
                    }
}
