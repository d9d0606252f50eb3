// u3d_smallc.hip — the first convolution of the network (in_channels 1..4: Conv3d(in,f/2,3,padding=1) behind a
// one-group GroupNorm, buildingblocks.py:62-75 + :56).  K = 27*Cin is far too small for the MFMA tiling (Cin would be
// padded to 16 and, in the data gradient, Cout=Cin to 32), so this layer gets its own bandwidth-shaped kernels:
//
//   forward : direct convolution, one voxel per thread, halo tile + weights in LDS, 16 B coalesced stores
//   backward: NO data-gradient pass at all.  With X[n,k,c,t] = sum_u dz[n,u,k] * x[n,u+t-1,c] and
//             T[n,k,t] = sum_{u : u+t-1 in bounds} dz[n,u,k], everything the layer owes is linear in (X,T):
//               dw[k,c,t]        = sum_n a[n,c] * X[n,k,c,t] + b[n,c] * T[n,k,t]        (g = a*x + b, zero padded)
//               sum_v dg[n,v,c]      = sum_{k,t} w[k,c,t] * T[n,k,t]                        (GroupNorm-backward S1)
//               sum_v dg[n,v,c]*x    = sum_{k,t} w[k,c,t] * X[n,k,c,t]                      (GroupNorm-backward S2)
//             so one pass over (dz, x) replaces wgrad + dgrad + their reductions.
#include "u3d_common.h"

namespace sc {
constexpr int TZ = 4, TY = 8, TX = 8;
constexpr int HZ = 6, HY = 10, HX = 10;
constexpr int HV = HZ * HY * HX;  // 600 halo voxels
constexpr int MAXC = 4;
}  // namespace sc

struct SmallFwdParams {
    const float* x;       // (N,D,H,W,Cin)
    const float* affine;  // [N][Cin][2] or null
    const float* w;       // (Cout,Cin,27) reference layout
    float* out;           // (N,D,H,W,Cout)
    int N, D, H, W, Cin, Cout, relu;
    int tz, ty, tx;
};

template <int COUTP>
__global__ __launch_bounds__(256) void conv3d_small_fwd_kernel(const SmallFwdParams p) {
    using namespace sc;
    __shared__ __attribute__((aligned(16))) float xs[HV * MAXC];
    __shared__ __attribute__((aligned(16))) float ws[27 * MAXC * COUTP];
    const int t = threadIdx.x;
    int tile = blockIdx.x;
    const int txi = tile % p.tx;
    tile /= p.tx;
    const int tyi = tile % p.ty;
    tile /= p.ty;
    const int tzi = tile % p.tz;
    const int n = tile / p.tz;
    const int z0 = tzi * TZ, y0 = tyi * TY, x0 = txi * TX;
    const int Cin = p.Cin, D = p.D, H = p.H, W = p.W;
    // weights -> LDS as [tap][c][k] (k padded to COUTP with zeros)
    for (int i = t; i < 27 * Cin * COUTP; i += 256) {
        const int k = i % COUTP;
        const int r = i / COUTP;
        const int c = r % Cin, tap = r / Cin;
        ws[i] = k < p.Cout ? p.w[((size_t)k * Cin + c) * 27 + tap] : 0.f;
    }
    // halo tile with the GroupNorm affine applied, zero padded
    for (int i = t; i < HV * Cin; i += 256) {
        const int c = i % Cin;
        const int hv = i / Cin;
        const int hz = hv / (HY * HX), rem = hv - hz * (HY * HX), hy = rem / HX, hx = rem - hy * HX;
        const int gz = z0 - 1 + hz, gy = y0 - 1 + hy, gx = x0 - 1 + hx;
        float v = 0.f;
        if (gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W) {
            v = p.x[((size_t)((n * D + gz) * H + gy) * W + gx) * Cin + c];
            if (p.affine) v = v * p.affine[((size_t)n * Cin + c) * 2] + p.affine[((size_t)n * Cin + c) * 2 + 1];
        }
        xs[hv * Cin + c] = v;
    }
    __syncthreads();
    const int zl = t >> 6, yl = (t >> 3) & 7, xl = t & 7;
    float acc[COUTP];
#pragma unroll
    for (int k = 0; k < COUTP; ++k) acc[k] = 0.f;
    for (int tap = 0; tap < 27; ++tap) {
        const int hv = (zl + tap / 9) * (HY * HX) + (yl + (tap / 3) % 3) * HX + xl + tap % 3;
        for (int c = 0; c < Cin; ++c) {
            const float xv = xs[hv * Cin + c];
            const f32x4* wr = reinterpret_cast<const f32x4*>(&ws[(tap * Cin + c) * COUTP]);
#pragma unroll
            for (int k4 = 0; k4 < COUTP / 4; ++k4) {
                const f32x4 wv = wr[k4];  // broadcast read
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[4 * k4 + e] = fmaf(xv, wv[e], acc[4 * k4 + e]);
            }
        }
    }
    const int z = z0 + zl, y = y0 + yl, x = x0 + xl;
    if (z < D && y < H && x < W) {
        float* o = p.out + ((size_t)((n * D + z) * H + y) * W + x) * p.Cout;
        if (p.Cout % 4 == 0) {
#pragma unroll
            for (int k4 = 0; k4 < COUTP / 4; ++k4) {
                if (4 * k4 < p.Cout) {
                    f32x4 v = {acc[4 * k4], acc[4 * k4 + 1], acc[4 * k4 + 2], acc[4 * k4 + 3]};
                    if (p.relu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                    *reinterpret_cast<f32x4*>(o + 4 * k4) = v;
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < COUTP; ++k)
                if (k < p.Cout) o[k] = p.relu ? fmaxf(acc[k], 0.f) : acc[k];
        }
    }
}

extern "C" int u3d_conv3d_small_cin_fwd(int device, u3d_stream_t stream, const float* x, const float* affine,
                                        const float* w, float* out, int N, int D, int H, int W, int Cin, int Cout,
                                        int relu) {
    if (int e = u3d_enter(device)) return e;
    U3D_REQUIRE(x && w && out && N > 0 && D > 0 && H > 0 && W > 0, "u3d_conv3d_small_cin_fwd: bad argument");
    U3D_REQUIRE(Cin >= 1 && Cin <= sc::MAXC && Cout >= 1 && Cout <= 32,
                "u3d_conv3d_small_cin_fwd: needs Cin<=4, Cout<=32 (got %d,%d)", Cin, Cout);
    SmallFwdParams p;
    p.x = x, p.affine = affine, p.w = w, p.out = out;
    p.N = N, p.D = D, p.H = H, p.W = W, p.Cin = Cin, p.Cout = Cout, p.relu = relu;
    p.tz = (D + sc::TZ - 1) / sc::TZ, p.ty = (H + sc::TY - 1) / sc::TY, p.tx = (W + sc::TX - 1) / sc::TX;
    const long long nblk = (long long)N * p.tz * p.ty * p.tx;
    U3D_REQUIRE(nblk < (1ll << 31), "u3d_conv3d_small_cin_fwd: grid too large");
    hipStream_t st = (hipStream_t)stream;
    if (Cout <= 8)
        hipLaunchKernelGGL(conv3d_small_fwd_kernel<8>, dim3((unsigned)nblk), dim3(256), 0, st, p);
    else if (Cout <= 16)
        hipLaunchKernelGGL(conv3d_small_fwd_kernel<16>, dim3((unsigned)nblk), dim3(256), 0, st, p);
    else
        hipLaunchKernelGGL(conv3d_small_fwd_kernel<32>, dim3((unsigned)nblk), dim3(256), 0, st, p);
    U3D_LAUNCH_CHECK();
    return 0;
}

// -------------------------------------------------------------------------------------------------------------
// backward: partial[n][b][(k*27 + tap)*(Cin+1) + c]  (c == Cin is the T slot).  grid (B, N); a block walks tiles
// b, b+B, ... of sample n.  256 threads = 16 k-lanes x 16 tap-groups; thread (kq, tg) owns taps {tg, tg+16} and output
// channels {kq, kq+16}.
struct SmallBwdParams {
    const float* x;   // raw input (N,D,H,W,Cin)
    const float* dz;  // (N,D,H,W,Cout)
    float* partial;
    int N, D, H, W, Cin, Cout;
    int tz, ty, tx, B;
};

template <int KH>  // output channels per thread: k = kq + 16*a, a < KH  (KH = 1 for Cout <= 16)
__global__ __launch_bounds__(256) void conv3d_small_bwd_kernel(const SmallBwdParams p) {
    using namespace sc;
    __shared__ float xs[HV * (MAXC + 1)];  // [hv][Cin+1]: raw x, then the in-bounds indicator
    __shared__ float dzs[256 * 16 * KH];   // [voxel][k] (k < 16*KH)
    const int t = threadIdx.x;
    const int n = blockIdx.y;
    const int kq = t & 15, tg = t >> 4;
    const int Cin = p.Cin, C1 = p.Cin + 1, Cout = p.Cout, D = p.D, H = p.H, W = p.W;
    float acc[KH][2][MAXC + 1];  // [k half][tap slot][c]
#pragma unroll
    for (int a = 0; a < KH; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c <= MAXC; ++c) acc[a][b][c] = 0.f;
    const int tap0 = tg, tap1 = tg + 16;  // tap1 valid if < 27
    const int off0 = (tap0 / 9) * (HY * HX) + ((tap0 / 3) % 3) * HX + tap0 % 3;
    const int off1 = tap1 < 27 ? (tap1 / 9) * (HY * HX) + ((tap1 / 3) % 3) * HX + tap1 % 3 : 0;
    const int ntiles = p.tz * p.ty * p.tx;
    for (int tile = blockIdx.x; tile < ntiles; tile += p.B) {
        int tt = tile;
        const int txi = tt % p.tx;
        tt /= p.tx;
        const int tyi = tt % p.ty;
        const int tzi = tt / p.ty;
        const int z0 = tzi * TZ, y0 = tyi * TY, x0 = txi * TX;
        __syncthreads();
        for (int i = t; i < HV; i += 256) {
            const int hz = i / (HY * HX), rem = i - hz * (HY * HX), hy = rem / HX, hx = rem - hy * HX;
            const int gz = z0 - 1 + hz, gy = y0 - 1 + hy, gx = x0 - 1 + hx;
            const bool in = gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W;
            const float* src = p.x + ((size_t)((n * D + (in ? gz : 0)) * H + (in ? gy : 0)) * W + (in ? gx : 0)) * Cin;
            for (int c = 0; c < Cin; ++c) xs[i * C1 + c] = in ? src[c] : 0.f;
            xs[i * C1 + Cin] = in ? 1.f : 0.f;
        }
        constexpr int KW = 16 * KH;
        for (int i = t; i < 256 * KW; i += 256) {
            const int k = i % KW, v = i / KW;
            const int z = z0 + (v >> 6), y = y0 + ((v >> 3) & 7), x = x0 + (v & 7);
            float val = 0.f;
            if (k < Cout && z < D && y < H && x < W) val = p.dz[((size_t)((n * D + z) * H + y) * W + x) * Cout + k];
            dzs[i] = val;
        }
        __syncthreads();
#pragma unroll 8
        for (int v = 0; v < 256; ++v) {
            const int hb = (v >> 6) * (HY * HX) + ((v >> 3) & 7) * HX + (v & 7);
            float d[KH];
#pragma unroll
            for (int a = 0; a < KH; ++a) d[a] = dzs[v * KW + kq + 16 * a];
            const float* x0p = &xs[(hb + off0) * C1];
            const float* x1p = &xs[(hb + off1) * C1];
#pragma unroll
            for (int c = 0; c <= MAXC; ++c) {
                if (c < C1) {
                    const float xa = x0p[c], xb = x1p[c];
#pragma unroll
                    for (int a = 0; a < KH; ++a) {
                        acc[a][0][c] = fmaf(d[a], xa, acc[a][0][c]);
                        acc[a][1][c] = fmaf(d[a], xb, acc[a][1][c]);
                    }
                }
            }
        }
    }
    float* dst = p.partial + ((size_t)n * p.B + blockIdx.x) * ((size_t)Cout * 27 * C1);
#pragma unroll
    for (int a = 0; a < KH; ++a) {
        const int k = kq + 16 * a;
        if (k >= Cout) continue;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int tap = b == 0 ? tap0 : tap1;
            if (tap >= 27) continue;
#pragma unroll
            for (int c = 0; c <= MAXC; ++c)
                if (c < C1) dst[((size_t)k * 27 + tap) * C1 + c] = acc[a][b][c];
        }
    }
}

// finalize: grid = ceil(Cout*27 / 64) blocks of 256 threads = 64 (k,tap) lanes x 4 partial-groups.  Fixed summation
// order; forms dw (summed over n in-thread) and adds the GroupNorm-backward sums (S1,S2) per (n, c).
__global__ __launch_bounds__(256) void conv3d_small_bwd_finalize_kernel(const float* __restrict__ partial,
                                                                        const float* __restrict__ affine,
                                                                        const float* __restrict__ w, int N, int B,
                                                                        int Cin, int Cout, float* __restrict__ dw,
                                                                        double* __restrict__ gstats) {
    __shared__ float red[4][64][sc::MAXC + 1];
    const int C1 = Cin + 1;
    const int nkt = Cout * 27;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int kt = blockIdx.x * 64 + lane;
    const bool valid = kt < nkt;
    const int k = valid ? kt / 27 : 0, tap = valid ? kt % 27 : 0;
    double dwacc[sc::MAXC];
    for (int c = 0; c < sc::MAXC; ++c) dwacc[c] = 0.0;
    for (int n = 0; n < N; ++n) {
        float acc[sc::MAXC + 1];
        for (int c = 0; c <= sc::MAXC; ++c) acc[c] = 0.f;
        if (valid) {
            for (int b = grp; b < B; b += 4) {
                const float* src = partial + ((size_t)n * B + b) * ((size_t)nkt * C1) + (size_t)kt * C1;
                for (int c = 0; c < C1; ++c) acc[c] += src[c];
            }
        }
        for (int c = 0; c < C1; ++c) red[grp][lane][c] = acc[c];
        __syncthreads();
        if (grp == 0) {  // wave 0: lanes = 64 (k,tap) pairs
            double X[sc::MAXC + 1];
            for (int c = 0; c < C1; ++c)
                X[c] = ((double)red[0][lane][c] + (double)red[1][lane][c]) + ((double)red[2][lane][c] + (double)red[3][lane][c]);
            const double T = X[Cin];
            for (int c = 0; c < Cin; ++c) {
                double s1 = 0.0, s2 = 0.0;
                if (valid) {
                    const double a = affine ? (double)affine[((size_t)n * Cin + c) * 2] : 1.0;
                    const double bb = affine ? (double)affine[((size_t)n * Cin + c) * 2 + 1] : 0.0;
                    dwacc[c] += a * X[c] + bb * T;
                    const double wv = (double)w[((size_t)k * Cin + c) * 27 + tap];
                    s1 = wv * T;
                    s2 = wv * X[c];
                }
                for (int m = 32; m > 0; m >>= 1) {
                    s1 += __shfl_xor(s1, m);
                    s2 += __shfl_xor(s2, m);
                }
                if (lane == 0 && gstats) {
                    u3d_atomic_add_f64(&gstats[((size_t)n * Cin + c) * 2], s1);
                    u3d_atomic_add_f64(&gstats[((size_t)n * Cin + c) * 2 + 1], s2);
                }
            }
        }
        __syncthreads();
    }
    if (grp == 0 && valid)
        for (int c = 0; c < Cin; ++c) dw[((size_t)k * Cin + c) * 27 + tap] = (float)dwacc[c];
}

static int small_bwd_blocks(int N, int D, int H, int W) {
    const long long ntiles = (long long)((D + 3) / 4) * ((H + 7) / 8) * ((W + 7) / 8);
    long long B = 1024 / (N > 0 ? N : 1);  // ~4 blocks per CU in total (latency hiding); each walks its share of tiles
    if (B < 1) B = 1;
    if (B > ntiles) B = ntiles;
    return (int)B;
}

extern "C" size_t u3d_small_cin_bwd_workspace_floats(int N, int D, int H, int W, int Cin, int Cout) {
    return (size_t)N * small_bwd_blocks(N, D, H, W) * Cout * 27 * (Cin + 1);
}

extern "C" int u3d_conv3d_small_cin_bwd(int device, u3d_stream_t stream, const float* x, const float* affine,
                                        const float* dz, const float* w, float* dw, double* gstats, int N, int D,
                                        int H, int W, int Cin, int Cout, float* workspace, size_t workspace_floats) {
    if (int e = u3d_enter(device)) return e;
    U3D_REQUIRE(x && dz && w && dw && workspace && N > 0 && D > 0 && H > 0 && W > 0, "u3d_conv3d_small_cin_bwd: bad argument");
    U3D_REQUIRE(Cin >= 1 && Cin <= sc::MAXC && Cout >= 1 && Cout <= 32 && Cout * 27 <= 1024,
                "u3d_conv3d_small_cin_bwd: needs Cin<=4, Cout<=32 (got %d,%d)", Cin, Cout);
    SmallBwdParams p;
    p.x = x, p.dz = dz, p.partial = workspace;
    p.N = N, p.D = D, p.H = H, p.W = W, p.Cin = Cin, p.Cout = Cout;
    p.tz = (D + sc::TZ - 1) / sc::TZ, p.ty = (H + sc::TY - 1) / sc::TY, p.tx = (W + sc::TX - 1) / sc::TX;
    p.B = small_bwd_blocks(N, D, H, W);
    const size_t need = (size_t)N * p.B * Cout * 27 * (Cin + 1);
    if (workspace_floats < need)
        return u3d_set_err(U3D_EWORKSPACE, "u3d_conv3d_small_cin_bwd: workspace %zu < %zu floats", workspace_floats, need);
    if (Cout <= 16)
        hipLaunchKernelGGL(conv3d_small_bwd_kernel<1>, dim3((unsigned)p.B, (unsigned)N), dim3(256), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(conv3d_small_bwd_kernel<2>, dim3((unsigned)p.B, (unsigned)N), dim3(256), 0, (hipStream_t)stream, p);
    U3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(conv3d_small_bwd_finalize_kernel, dim3((Cout * 27 + 63) / 64), dim3(256), 0, (hipStream_t)stream,
                       workspace, affine, w, N, p.B, Cin, Cout, dw, gstats);
    U3D_LAUNCH_CHECK();
    return 0;
}
