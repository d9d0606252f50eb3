// u3d_res.hip — the operators the residual U-Net variants add around the 3x3x3 convolutions
// (ResidualUNet3D / ResidualUNetSE3D):
//   * 1x1x1 convolution WITH bias that widens the block input (ResNetBlock.conv1, buildingblocks.py:248-255) fwd / bwd
//   * ConvTranspose3d(k=3, stride=2, padding=1, bias=False) (TransposeConvUpsampling, buildingblocks.py:617-664) fwd / bwd
//   * nearest resize of its (2n-1) output to the skip's size + summation joining (:650-651, :493) fwd / bwd
// These are <7 % of the model's FLOPs (8.4 + 212 of 3675 GFLOP at BASELINE config 4).  They run as ONE generic
// register-tiled "gather GEMM" on the FP32 vector units (64 rows x 64 outputs per block, 4x4 per thread, K-chunks of 16
// through LDS): rows are output voxels on a regular sub-grid, each tap reads the input voxel row*stride + offset (zero
// outside the tensor).  ConvTranspose3d is 8 such GEMMs, one per output parity class: an even output coordinate 2j
// receives only tap 1 of input j, an odd one 2j+1 taps 0 and 2 of inputs j+1 and j — 1/2/4/8 taps per class, no
// multiplications by the inserted zeros.  (An MFMA version of the 8 parity classes is the next step for this file.)
#include "u3d_common.h"
#include <type_traits>

namespace gc {
constexpr int TM = 64, TN = 64, TK = 16, LD = 68;  // LD: padded LDS row (keeps 16-byte alignment of the quads)
}

struct GConvParams {
    const void* x;      // (N, Di, Hi, Wi, Ci), element type TI of gconv_kernel<TI, TO>
    const float* w;     // element (tap, in-channel i, out-channel j) at w[woff[tap] + i*wsi + j*wsj]
    const float* bias;  // [Cj] or null
    const void* mask;   // output-shaped (TO) or null: out = mask > 0 ? out : 0   (ReLU backward of the producer)
    void* out;          // (N, Do, Ho, Wo, Cj), element type TO
    double* stats;      // [N][Cj][2] += (sum, sum of squares) of the written values, or null
    int N, Di, Hi, Wi, Ci, Do, Ho, Wo, Cj;
    int Rz, Ry, Rx;        // row grid; output coordinate = r * os + oo, input coordinate of tap t = r * is + t?[t]
    int osz, osy, osx, ooz, ooy, oox;
    int isz, isy, isx;
    int ntaps;
    signed char tz[27], ty[27], tx[27];
    int woff[27];
    long long wsi, wsj;
    int avec, ovec, bvec;
};

// MFMA version (v_mfma_f32_32x32x2_f32): a block = 4 waves owns a 64-row x 64-column tile, wave w the 32x32 quadrant
// (w >> 1, w & 1).  The (tap, 16-channel chunk) steps run as ONE flat software pipeline: the A rows (gathered, zero
// outside the tensor) and the B slice of step s+1 are loaded into registers before the 8 MFMAs of step s are issued from
// the LDS buffer s & 1, then stored to buffer (s+1) & 1 — one barrier per step.
template <typename TI = float, typename TO = float>
__global__ __launch_bounds__(256) void gconv_kernel(const GConvParams p) {
    using namespace gc;
    const TI* px = reinterpret_cast<const TI*>(p.x);
    const TO* pmask = reinterpret_cast<const TO*>(p.mask);
    TO* pout = reinterpret_cast<TO*>(p.out);
    __shared__ __attribute__((aligned(16))) float As[2][TK][LD];  // [buffer][k][row]
    __shared__ __attribute__((aligned(16))) float Bs[2][TK][LD];  // [buffer][k][out channel]
    __shared__ int orow[TM];                                      // output voxel index of each tile row, -1 = no row
    __shared__ double sred[TN][2];
    const int t = threadIdx.x, l = t & 63, w = t >> 6;
    const int wr = w >> 1, wc = w & 1, lm = l & 31, lh = l >> 5;
    const int n = blockIdx.z, j0 = blockIdx.y * TN;
    const long long R = (long long)p.Rz * p.Ry * p.Rx;
    const int la = t >> 2, lq = t & 3;     // A-load role: row, channel quad of the 16-channel chunk
    const int bci = t >> 4, bjq = t & 15;  // B-load role: chunk channel, out-channel quad
    const int nchunk = (p.Ci + TK - 1) / TK, nsteps = p.ntaps * nchunk;
    const bool bvec = p.wsj == 1 && p.bvec;
    const int col = j0 + 32 * wc + lm;  // this lane's output channel
    float s1 = 0.f, s2 = 0.f;
    if (t < TN) sred[t][0] = sred[t][1] = 0.0;
    for (long long tile = blockIdx.x; tile * TM < R; tile += gridDim.x) {
        const long long ra = tile * TM + la;
        const bool rok = ra < R;
        int rz = 0, ry = 0, rx = 0;
        if (rok) {
            rx = (int)(ra % p.Rx);
            const long long q = ra / p.Rx;
            ry = (int)(q % p.Ry);
            rz = (int)(q / p.Ry);
        }
        __syncthreads();  // the previous tile's epilogue has read orow; its last MFMAs have read the LDS buffers
        if (lq == 0)
            orow[la] = rok ? ((n * p.Do + rz * p.osz + p.ooz) * p.Ho + ry * p.osy + p.ooy) * p.Wo + rx * p.osx + p.oox : -1;
        auto load_step = [&](int s_, f32x4& av, f32x4& bv) {
            const int tap = s_ / nchunk, c0 = (s_ - tap * nchunk) * TK;
            av = f32x4{0.f, 0.f, 0.f, 0.f};
            bv = f32x4{0.f, 0.f, 0.f, 0.f};
            const int iz = rz * p.isz + p.tz[tap], iy = ry * p.isy + p.ty[tap], ix = rx * p.isx + p.tx[tap];
            const bool inb = rok && iz >= 0 && iz < p.Di && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
            const int ca = c0 + 4 * lq;
            if (inb) {
                const TI* xrow = px + ((size_t)((n * p.Di + iz) * p.Hi + iy) * p.Wi + ix) * p.Ci;
                if (p.avec && ca + 3 < p.Ci) {
                    av = u3d_ldq(xrow + ca);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (ca + e < p.Ci) av[e] = u3d_ld(xrow + ca + e);
                }
            }
            const int cb = c0 + bci, jb_ = j0 + 4 * bjq;
            if (cb < p.Ci) {
                const float* wt = p.w + p.woff[tap] + (size_t)cb * p.wsi;
                if (bvec && jb_ + 3 < p.Cj) {
                    bv = *reinterpret_cast<const f32x4*>(wt + jb_);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (jb_ + e < p.Cj) bv[e] = wt[(size_t)(jb_ + e) * p.wsj];
                }
            }
        };
        auto store_step = [&](int buf, const f32x4& av, const f32x4& bv) {
#pragma unroll
            for (int e = 0; e < 4; ++e) As[buf][4 * lq + e][la] = av[e];
            *reinterpret_cast<f32x4*>(&Bs[buf][bci][4 * bjq]) = bv;
        };
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        f32x4 av, bv;
        load_step(0, av, bv);
        store_step(0, av, bv);
        __syncthreads();
        for (int s_ = 0; s_ < nsteps; ++s_) {
            const int buf = s_ & 1;
            const bool more = s_ + 1 < nsteps;
            if (more) load_step(s_ + 1, av, bv);
#pragma unroll
            for (int kk = 0; kk < TK; kk += 2) {
                const float a = As[buf][kk + lh][32 * wr + lm];
                const float b = Bs[buf][kk + lh][32 * wc + lm];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
            if (more) store_step(buf ^ 1, av, bv);
            __syncthreads();
        }
        // ---- epilogue.  C/D layout: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
        const bool cok = col < p.Cj;
        const float bias = (p.bias && cok) ? p.bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * wr + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int ov = orow[row];
            if (ov < 0 || !cok) continue;
            const size_t o = (size_t)ov * p.Cj + col;
            float val = acc[r] + bias;
            if (pmask && !(u3d_ld(pmask + o) > 0.f)) val = 0.f;
            val = u3d_stored(val, pout);  // (statistics describe the STORED tensor)
            u3d_st(pout + o, val);
            s1 += val;
            s2 += val * val;
        }
    }
    if (p.stats) {
        __syncthreads();
        if (col < p.Cj) {
            __hip_atomic_fetch_add(&sred[32 * wc + lm][0], (double)s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&sred[32 * wc + lm][1], (double)s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __syncthreads();
        if (t < TN && j0 + t < p.Cj) {
            u3d_atomic_add_f64(&p.stats[((size_t)n * p.Cj + j0 + t) * 2], sred[t][0]);
            u3d_atomic_add_f64(&p.stats[((size_t)n * p.Cj + j0 + t) * 2 + 1], sred[t][1]);
        }
    }
}

// ---- weight-gradient twin: acc[tap*dst_t + a*dst_a + b*dst_b] += sum_rows X[row, a] * Y[ycoord(row, tap), b]
struct GWgradParams {
    const void* X;  // (N, Rz, Ry, Rx, Ca), element type T of gwgrad_kernel<T>
    const void* Y;  // (N, Dy, Hy, Wy, Cb); coordinate of tap t = r * ys + t?[t], zero outside
    double* acc;
    double* bias_acc;  // optional [Cb] += sum_rows Y[row, b] (tap 0 only)
    int N, Ca, Cb;
    int Rz, Ry, Rx, Dy, Hy, Wy, ysz, ysy, ysx;
    int ntaps;
    signed char tz[27], ty[27], tx[27];
    long long dst_t, dst_a, dst_b;
    int atiles, btiles;
    long long rows_per_split;
    int xvec, yvec;
};

template <typename TX = float, typename TY = TX>
__global__ __launch_bounds__(256) void gwgrad_kernel(const GWgradParams p) {
    const TX* pX = reinterpret_cast<const TX*>(p.X);
    const TY* pY = reinterpret_cast<const TY*>(p.Y);
    // same MFMA scheme: D[a][b] += X^T Y over 16-row chunks; wave w owns the 32x32 quadrant (w >> 1, w & 1) of the 64x64 tile;
    // register-prefetched double buffering, one barrier per chunk
    using namespace gc;
    __shared__ __attribute__((aligned(16))) float Xs[2][TK][LD];
    __shared__ __attribute__((aligned(16))) float Ys[2][TK][LD];
    const int t = threadIdx.x, l = t & 63, w = t >> 6;
    const int wr = w >> 1, wc = w & 1, lm = l & 31, lh = l >> 5;
    const int a0 = (blockIdx.x % p.atiles) * TM, b0 = (blockIdx.x / p.atiles) * TN;
    const int tap = blockIdx.y;
    const long long R = (long long)p.Rz * p.Ry * p.Rx, total = (long long)p.N * R;
    const long long r_begin = (long long)blockIdx.z * p.rows_per_split;
    const long long r_end = r_begin + p.rows_per_split < total ? r_begin + p.rows_per_split : total;
    const int lr = t >> 4, lqd = t & 15;  // load role: row of the 16-row chunk, channel quad
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float bsum = 0.f;
    const bool want_bias = p.bias_acc != nullptr && tap == 0 && a0 == 0 && wr == 0;
    auto load_chunk = [&](long long rc, f32x4& xv, f32x4& yv) {
        xv = f32x4{0.f, 0.f, 0.f, 0.f};
        yv = f32x4{0.f, 0.f, 0.f, 0.f};
        const long long row = rc + lr;
        if (row >= r_end) return;
        const int n = (int)(row / R);
        const long long r = row - (long long)n * R;
        const int rx = (int)(r % p.Rx);
        const long long q = r / p.Rx;
        const int ry = (int)(q % p.Ry), rz = (int)(q / p.Ry);
        const int ca = a0 + 4 * lqd;
        const TX* xr = pX + (size_t)row * p.Ca;
        if (p.xvec && ca + 3 < p.Ca) {
            xv = u3d_ldq(xr + ca);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (ca + e < p.Ca) xv[e] = u3d_ld(xr + ca + e);
        }
        const int yz = rz * p.ysz + p.tz[tap], yy = ry * p.ysy + p.ty[tap], yx = rx * p.ysx + p.tx[tap];
        if (yz >= 0 && yz < p.Dy && yy >= 0 && yy < p.Hy && yx >= 0 && yx < p.Wy) {
            const TY* yr = pY + ((size_t)((n * p.Dy + yz) * p.Hy + yy) * p.Wy + yx) * p.Cb;
            const int cb = b0 + 4 * lqd;
            if (p.yvec && cb + 3 < p.Cb) {
                yv = u3d_ldq(yr + cb);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (cb + e < p.Cb) yv[e] = u3d_ld(yr + cb + e);
            }
        }
    };
    f32x4 xv, yv;
    if (r_begin < r_end) {
        load_chunk(r_begin, xv, yv);
        *reinterpret_cast<f32x4*>(&Xs[0][lr][4 * lqd]) = xv;
        *reinterpret_cast<f32x4*>(&Ys[0][lr][4 * lqd]) = yv;
    }
    __syncthreads();
    int buf = 0;
    for (long long rc = r_begin; rc < r_end; rc += TK, buf ^= 1) {
        const bool more = rc + TK < r_end;
        if (more) load_chunk(rc + TK, xv, yv);
#pragma unroll
        for (int kk = 0; kk < TK; kk += 2) {
            const float a = Xs[buf][kk + lh][32 * wr + lm];
            const float b = Ys[buf][kk + lh][32 * wc + lm];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            if (want_bias) bsum += b;
        }
        if (more) {
            *reinterpret_cast<f32x4*>(&Xs[buf ^ 1][lr][4 * lqd]) = xv;
            *reinterpret_cast<f32x4*>(&Ys[buf ^ 1][lr][4 * lqd]) = yv;
        }
        __syncthreads();
    }
    const int b = b0 + 32 * wc + lm;
    if (b < p.Cb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int a = a0 + 32 * wr + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (a < p.Ca) u3d_atomic_add_f64(&p.acc[(size_t)tap * p.dst_t + (size_t)a * p.dst_a + (size_t)b * p.dst_b], (double)acc[r]);
        }
        if (want_bias) {
            bsum += __shfl_xor(bsum, 32);  // the two k-halves of the wave hold alternate rows
            if (lh == 0) u3d_atomic_add_f64(&p.bias_acc[b], (double)bsum);
        }
    }
}

// ---- nearest resize + summation joining: out = skip + t[map(voxel)]  (+ per-(n,channel) statistics of out) ----------
template <typename T = float>
__global__ __launch_bounds__(256) void nearest_add_kernel(const T* __restrict__ skip, const T* __restrict__ tt,
                                                          const int* __restrict__ zmap, const int* __restrict__ ymap,
                                                          const int* __restrict__ xmap, int D, int H, int W, int Dt, int Ht,
                                                          int Wt, int C, int Q, int vecw, int t8, T* __restrict__ out,
                                                          double* __restrict__ stats) {
    // thread -> (row = t / Q, unit = t % Q); a unit is vecw (4 or 1) channels; rows stride the block's voxel range
    extern __shared__ double sred[];  // [Q*vecw][2]
    const int n = blockIdx.y;
    const int t = threadIdx.x;
    const int rows = 256 / Q;
    const int unit = t % Q, row = t / Q;
    const bool active = row < rows;
    for (int k = t; k < 2 * Q * vecw; k += 256) sred[k] = 0.0;
    __syncthreads();
    const long long V = (long long)D * H * W;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    if (active) {
        for (long long v = (long long)blockIdx.x * rows + row; v < V; v += (long long)gridDim.x * rows) {
            const int x = (int)(v % W);
            const long long q = v / W;
            const int y = (int)(q % H), z = (int)(q / H);
            const size_t o = ((size_t)n * V + v) * C + (size_t)unit * vecw;
            size_t ti;
            if (t8) {  // space-to-depth layout of the transposed convolution's output (csrc/u3d_bf16.hip): T8[i][parity*C + c]
                const int zt = zmap[z], yt = ymap[y], xt = xmap[x];
                const int D1 = (Dt + 1) >> 1, H1 = (Ht + 1) >> 1, W1 = (Wt + 1) >> 1;
                ti = (((size_t)((n * D1 + (zt >> 1)) * H1 + (yt >> 1)) * W1 + (xt >> 1)) * 8 + ((zt & 1) * 4 + (yt & 1) * 2 + (xt & 1))) * C +
                     (size_t)unit * vecw;
            } else {
                ti = ((size_t)((n * Dt + zmap[z]) * Ht + ymap[y]) * Wt + xmap[x]) * C + (size_t)unit * vecw;
            }
            if (vecw == 4) {
                const f32x4 a = u3d_ldq(skip + o);
                const f32x4 b = u3d_ldq(tt + ti);
                f32x4 r = a + b;
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] = u3d_stored(r[e], out);
                u3d_stq(out + o, r);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s1[e] += r[e];
                    s2[e] += r[e] * r[e];
                }
            } else {
                const float r = u3d_stored(u3d_ld(skip + o) + u3d_ld(tt + ti), out);
                u3d_st(out + o, r);
                s1[0] += r;
                s2[0] += r * r;
            }
        }
    }
    if (stats) {
        if (active) {
            for (int e = 0; e < vecw; ++e) {
                __hip_atomic_fetch_add(&sred[(unit * vecw + e) * 2], (double)s1[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(&sred[(unit * vecw + e) * 2 + 1], (double)s2[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        __syncthreads();
        for (int k = t; k < 2 * C; k += 256) u3d_atomic_add_f64(&stats[(size_t)n * C * 2 + k], sred[k]);
    }
}

// bf16 storage + space-to-depth source (config 4's joins: 0.37 ms per step on the kernel above, which moves 8 bytes per lane behind two
// 64-bit divisions and runs at most 2048 blocks — 2.8 TB/s): a thread owns a channel OCTET (16-byte accesses) of FOUR voxels whose loads
// are all issued before the first use; 32-bit index arithmetic; same values (bf16(skip + t8)) and statistics of the stored tensor.
__global__ __launch_bounds__(256) void nearest_add_t8_b16_oct_kernel(const __bf16* __restrict__ skip, const __bf16* __restrict__ t8,
                                                                     const int* __restrict__ zmap, const int* __restrict__ ymap,
                                                                     const int* __restrict__ xmap, int D, int H, int W, int D1, int H1,
                                                                     int W1, int C, int OCT, __bf16* __restrict__ out,
                                                                     double* __restrict__ stats) {
    typedef __bf16 b16x8 __attribute__((ext_vector_type(8)));
    typedef __bf16 b16x2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    extern __shared__ double sred[];  // [C][2]
    constexpr int U = 4;
    const int n = blockIdx.y, t = threadIdx.x;
    const int rows = 256 / OCT, o = t % OCT, row = t / OCT;
    for (int k = t; k < 2 * C; k += 256) sred[k] = 0.0;
    __syncthreads();
    const int V = D * H * W;
    const __bf16* sp = skip + (size_t)n * V * C + o * 8;
    const __bf16* tp = t8 + (size_t)n * D1 * H1 * W1 * 8 * C + o * 8;
    __bf16* op = out + (size_t)n * V * C + o * 8;
    f2 s1[4], s2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) s1[i] = s2[i] = f2{0.f, 0.f};
    for (int v0 = blockIdx.x * rows * U; v0 < V; v0 += gridDim.x * rows * U) {
        b16x8 a[U], b[U];
        int vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int v = v0 + u * rows + row;
            vv[u] = v < V ? v : -1;
            const int vs = v < V ? v : 0;
            const int q = vs / W, x = vs - q * W, z = q / H, y = q - z * H;
            const int zt = zmap[z], yt = ymap[y], xt = xmap[x];
            const size_t ti = ((size_t)(((zt >> 1) * H1 + (yt >> 1)) * W1 + (xt >> 1)) * 8 + ((zt & 1) * 4 + (yt & 1) * 2 + (xt & 1))) * C;
            a[u] = *reinterpret_cast<const b16x8*>(sp + (size_t)vs * C);
            b[u] = *reinterpret_cast<const b16x8*>(tp + ti);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (vv[u] < 0) continue;
            const u4 au = __builtin_bit_cast(u4, a[u]), bu = __builtin_bit_cast(u4, b[u]);
            u4 ru;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f2 x2 = {__builtin_bit_cast(float, au[i] << 16), __builtin_bit_cast(float, au[i] & 0xffff0000u)};
                const f2 y2 = {__builtin_bit_cast(float, bu[i] << 16), __builtin_bit_cast(float, bu[i] & 0xffff0000u)};
                ru[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(x2 + y2, b16x2));
                const f2 r2 = {__builtin_bit_cast(float, ru[i] << 16), __builtin_bit_cast(float, ru[i] & 0xffff0000u)};  // as stored
                s1[i] += r2;
                s2[i] = __builtin_elementwise_fma(r2, r2, s2[i]);
            }
            *reinterpret_cast<u4*>(op + (size_t)vv[u] * C) = ru;
        }
    }
    if (stats) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                __hip_atomic_fetch_add(&sred[(o * 8 + 2 * i + e) * 2], (double)s1[i][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(&sred[(o * 8 + 2 * i + e) * 2 + 1], (double)s2[i][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        __syncthreads();
        for (int k = t; k < 2 * C; k += 256) u3d_atomic_add_f64(&stats[(size_t)n * C * 2 + k], sred[k]);
    }
}

// dt[s] = sum of dj over the children of s (voxels o with map(o) == s): lo tables of length Dt+1 / Ht+1 / Wt+1
template <typename T = float>
__global__ void nearest_sum_kernel(const T* __restrict__ dj, const int* __restrict__ zlo, const int* __restrict__ ylo,
                                   const int* __restrict__ xlo, int N, int D, int H, int W, int Dt, int Ht, int Wt, int C,
                                   int t8, T* __restrict__ dt) {
    const int D1 = (Dt + 1) >> 1, H1 = (Ht + 1) >> 1, W1 = (Wt + 1) >> 1;
    const long long total = t8 ? (long long)N * D1 * H1 * W1 * 8 * C : (long long)N * Dt * Ht * Wt * C;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C);
        long long v = idx / C;
        int xx, yy, zz, n;
        if (t8) {  // T8[i][parity*C + c] = dt[2i + parity]; parities that fall outside the (2n-1) grid are written as 0
            const int par = (int)(v & 7);
            v >>= 3;
            xx = 2 * (int)(v % W1) + (par & 1);
            v /= W1;
            yy = 2 * (int)(v % H1) + ((par >> 1) & 1);
            v /= H1;
            zz = 2 * (int)(v % D1) + (par >> 2);
            n = (int)(v / D1);
            if (zz >= Dt || yy >= Ht || xx >= Wt) {
                u3d_st(dt + idx, 0.f);
                continue;
            }
        } else {
            xx = (int)(v % Wt);
            v /= Wt;
            yy = (int)(v % Ht);
            v /= Ht;
            zz = (int)(v % Dt);
            n = (int)(v / Dt);
        }
        float sum = 0.f;
        for (int z = zlo[zz]; z < zlo[zz + 1]; ++z)
            for (int y = ylo[yy]; y < ylo[yy + 1]; ++y)
                for (int x = xlo[xx]; x < xlo[xx + 1]; ++x) sum += u3d_ld(dj + ((size_t)((n * D + z) * H + y) * W + x) * C + c);
        u3d_st(dt + idx, sum);
    }
}

// bf16 storage, space-to-depth layout, C % 8 == 0: a thread owns 8 channels (16 bytes) of one dT8 element and adds its (up to 8)
// children in fp32 — the scalar kernel above moves 2 bytes per lane and load (0.63 TB/s on config 4's 262 MB tensors)
typedef __bf16 rs_bf16x8 __attribute__((ext_vector_type(8)));
__global__ void nearest_sum_t8_b16_vec_kernel(const __bf16* __restrict__ dj, const int* __restrict__ zlo, const int* __restrict__ ylo,
                                              const int* __restrict__ xlo, int N, int D, int H, int W, int Dt, int Ht, int Wt, int C,
                                              __bf16* __restrict__ dt) {
    const int D1 = (Dt + 1) >> 1, H1 = (Ht + 1) >> 1, W1 = (Wt + 1) >> 1, Q = C >> 3;
    const long long total = (long long)N * D1 * H1 * W1 * 8 * Q;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int q = (int)(idx % Q);
        long long v = idx / Q;
        const int par = (int)(v & 7);
        v >>= 3;
        const int xx = 2 * (int)(v % W1) + (par & 1);
        v /= W1;
        const int yy = 2 * (int)(v % H1) + ((par >> 1) & 1);
        v /= H1;
        const int zz = 2 * (int)(v % D1) + (par >> 2);
        const int n = (int)(v / D1);
        float sum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (zz < Dt && yy < Ht && xx < Wt) {  // (parities outside the (2n-1) grid are written as 0)
            for (int z = zlo[zz]; z < zlo[zz + 1]; ++z)
                for (int y = ylo[yy]; y < ylo[yy + 1]; ++y)
                    for (int x = xlo[xx]; x < xlo[xx + 1]; ++x) {
                        const rs_bf16x8 c = *reinterpret_cast<const rs_bf16x8*>(dj + ((size_t)((n * D + z) * H + y) * W + x) * C + 8 * q);
#pragma unroll
                        for (int e = 0; e < 8; ++e) sum[e] += (float)c[e];
                    }
        }
        rs_bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (__bf16)sum[e];
        *reinterpret_cast<rs_bf16x8*>(dt + idx * 8) = o;
    }
}

// =====================================================================================================================
static inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

template <typename TI = float, typename TO = float>
static int launch_gconv(GConvParams& p, hipStream_t st) {
    const long long R = (long long)p.Rz * p.Ry * p.Rx;
    if (R <= 0) return 0;
    long long tiles = (R + gc::TM - 1) / gc::TM;
    const int jt = (p.Cj + gc::TN - 1) / gc::TN;
    // persistent over row tiles when statistics are accumulated (few atomics); otherwise one tile per block
    long long gx = tiles;
    const long long cap = p.stats ? (2048 / ((long long)jt * p.N) > 1 ? 2048 / ((long long)jt * p.N) : 1) : 65535 * 16;
    if (gx > cap) gx = cap;
    if (gx > 2147483647ll) gx = 2147483647ll;
    p.avec = (p.Ci % 4 == 0 && ((uintptr_t)p.x & u3d_vec_align<TI>::mask) == 0) ? 1 : 0;
    p.ovec = (p.Cj % 4 == 0 && al16(p.out) && (!p.mask || al16(p.mask))) ? 1 : 0;
    p.bvec = (p.wsj == 1 && p.Cj % 4 == 0 && p.wsi % 4 == 0 && al16(p.w)) ? 1 : 0;
    for (int k = 0; k < p.ntaps; ++k)
        if (p.woff[k] % 4 != 0) p.bvec = 0;
    hipLaunchKernelGGL((gconv_kernel<TI, TO>), dim3((unsigned)gx, (unsigned)jt, (unsigned)p.N), dim3(256), 0, st, p);
    U3D_LAUNCH_CHECK();
    return 0;
}

template <typename TX = float, typename TY = TX>
static int launch_gwgrad(GWgradParams& p, hipStream_t st) {
    const long long total = (long long)p.N * p.Rz * p.Ry * p.Rx;
    if (total <= 0) return 0;
    p.atiles = (p.Ca + gc::TM - 1) / gc::TM;
    p.btiles = (p.Cb + gc::TN - 1) / gc::TN;
    const long long blocks = (long long)p.atiles * p.btiles * p.ntaps;
    long long S = 2048 / blocks;  // ~8 blocks per CU in total
    if (S < 1) S = 1;
    const long long maxS = (total + 255) / 256;  // at least 256 rows per split
    if (S > maxS) S = maxS;
    if (S > 65535) S = 65535;
    long long rps = (total + S - 1) / S;
    rps = (rps + gc::TK - 1) / gc::TK * gc::TK;
    S = (total + rps - 1) / rps;
    p.rows_per_split = rps;
    p.xvec = (p.Ca % 4 == 0 && ((uintptr_t)p.X & u3d_vec_align<TX>::mask) == 0) ? 1 : 0;
    p.yvec = (p.Cb % 4 == 0 && ((uintptr_t)p.Y & u3d_vec_align<TY>::mask) == 0) ? 1 : 0;
    hipLaunchKernelGGL((gwgrad_kernel<TX, TY>), dim3((unsigned)(p.atiles * p.btiles), (unsigned)p.ntaps, (unsigned)S), dim3(256), 0, st, p);
    U3D_LAUNCH_CHECK();
    return 0;
}

// ---- 1x1x1 convolution with bias --------------------------------------------------------------------------------------
// ---- first block of the network with bf16 storage: Cin <= 4 fp32 input channels -> Cout bf16 channels.  Pure streaming (the
// gather-GEMM above spends 0.22 + 0.37 ms per config-4 step on what is a 262 MB write and a 262 MB read): a thread owns 8 output
// channels of a voxel (16-byte store); backward reduces dw[co][ci] = sum_v dy[v][co] x[v][ci] and db[co] = sum_v dy[v][co] with
// per-thread fp32 partials over a strided voxel range, a fixed-order LDS reduction per block and f64 atomics across blocks.
typedef __bf16 c1_bf16x8 __attribute__((ext_vector_type(8)));
template <int CIN>
__global__ __launch_bounds__(256) void conv1x1_smallc_fwd_b16_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                      const float* __restrict__ bias, __bf16* __restrict__ y, int N,
                                                                      long long V, int Cout, double* __restrict__ stats) {
    extern __shared__ float red[];  // [rows][Cout][2]
    const int Q = Cout >> 3, rows = 256 / Q;
    const int t = threadIdx.x, q = t % Q, row = t / Q;
    const int n = blockIdx.y;
    float wv[8][CIN], bv[8], s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        bv[e] = bias ? bias[8 * q + e] : 0.f;
        s1[e] = s2[e] = 0.f;
#pragma unroll
        for (int c = 0; c < CIN; ++c) wv[e][c] = w[(size_t)(8 * q + e) * CIN + c];
    }
    if (row < rows) {
        // four voxels per iteration, their loads in flight together (one dependent 4-byte load per 16-byte store ran this kernel at
        // 3 TB/s of writes); the voxels of a thread are still visited — and summed into the statistics — in ascending order
        const long long stride = (long long)gridDim.x * rows;
        for (long long v0 = (long long)blockIdx.x * rows + row; v0 < V; v0 += 4 * stride) {
            float xv[4][CIN];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long v = v0 + u * stride;
#pragma unroll
                for (int c = 0; c < CIN; ++c) xv[u][c] = v < V ? x[((size_t)n * V + v) * CIN + c] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long v = v0 + u * stride;
                if (v < V) {
                    c1_bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float a = bv[e];
#pragma unroll
                        for (int c = 0; c < CIN; ++c) a = fmaf(xv[u][c], wv[e][c], a);
                        o[e] = (__bf16)a;
                        const float r = (float)o[e];  // (statistics describe the STORED tensor)
                        s1[e] += r;
                        s2[e] = fmaf(r, r, s2[e]);
                    }
                    *reinterpret_cast<c1_bf16x8*>(y + ((size_t)n * V + v) * Cout + 8 * q) = o;
                }
            }
        }
    }
    if (stats == nullptr) return;
    if (row < rows) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[(row * Cout + 8 * q + e) * 2] = s1[e];
            red[(row * Cout + 8 * q + e) * 2 + 1] = s2[e];
        }
    }
    __syncthreads();
    for (int i = t; i < 2 * Cout; i += 256) {
        double sum = 0.0;
        for (int r = 0; r < rows; ++r) sum += (double)red[r * Cout * 2 + i];
        u3d_atomic_add_f64(&stats[(size_t)n * Cout * 2 + i], sum);
    }
}

template <int CIN>
__global__ __launch_bounds__(256) void conv1x1_smallc_bwd_b16_kernel(const __bf16* __restrict__ dy, const float* __restrict__ x, int N,
                                                                      long long V, int Cout, double* __restrict__ acc) {
    extern __shared__ float red[];  // [rows][Cout][CIN + 1]
    const int Q = Cout >> 3, rows = 256 / Q;
    const int t = threadIdx.x, q = t % Q, row = t / Q;
    float aw[8][CIN], ab[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        ab[e] = 0.f;
#pragma unroll
        for (int c = 0; c < CIN; ++c) aw[e][c] = 0.f;
    }
    const long long total = (long long)N * V;
    if (row < rows) {
        // four voxels per iteration with all their loads in flight (the accumulation order per thread is unchanged: ascending voxels)
        const long long stride = (long long)gridDim.x * rows;
        for (long long v0 = (long long)blockIdx.x * rows + row; v0 < total; v0 += 4 * stride) {
            c1_bf16x8 d[4];
            float xv[4][CIN];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long v = v0 + u * stride;
                const long long vc = v < total ? v : v0;
                d[u] = *reinterpret_cast<const c1_bf16x8*>(dy + (size_t)vc * Cout + 8 * q);
#pragma unroll
                for (int c = 0; c < CIN; ++c) xv[u][c] = x[(size_t)vc * CIN + c];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (v0 + u * stride < total) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float de = (float)d[u][e];
                        ab[e] += de;
#pragma unroll
                        for (int c = 0; c < CIN; ++c) aw[e][c] = fmaf(de, xv[u][c], aw[e][c]);
                    }
                }
            }
        }
    }
    constexpr int L = CIN + 1;
    if (row < rows) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
            for (int c = 0; c < CIN; ++c) red[(row * Cout + 8 * q + e) * L + c] = aw[e][c];
            red[(row * Cout + 8 * q + e) * L + CIN] = ab[e];
        }
    }
    __syncthreads();
    for (int i = t; i < Cout * L; i += 256) {
        double sum = 0.0;
        for (int r = 0; r < rows; ++r) sum += (double)red[r * Cout * L + i];
        const int co = i / L, c = i - co * L;
        // acc layout of u3d_conv1x1_bwd: dw[co][ci] at co*Cin + ci, db[co] at Cout*Cin + co
        u3d_atomic_add_f64(&acc[c < CIN ? (size_t)co * CIN + c : (size_t)Cout * CIN + co], sum);
    }
}

static bool smallc_ok(int Cin, int Cout) { return Cin >= 1 && Cin <= 4 && Cout % 8 == 0 && Cout >= 8 && Cout <= 256; }

template <typename TI, typename TO>
static int conv1x1_fwd_impl(int device, u3d_stream_t stream, const void* x, const float* w, const float* bias, void* y, int N,
                            int64_t V, int Cin, int Cout, double* out_stats) {
    U3D_ENTER(device);
    U3D_REQUIRE(x && w && y && N > 0 && V > 0 && Cin > 0 && Cout > 0, "u3d_conv1x1_fwd: bad argument");
    U3D_REQUIRE(V < (1ll << 31) && (long long)N * V < (1ll << 31), "u3d_conv1x1_fwd: N*V must be < 2^31");
    GConvParams p{};
    p.x = x, p.w = w, p.bias = bias, p.mask = nullptr, p.out = y, p.stats = out_stats;
    p.N = N, p.Di = 1, p.Hi = 1, p.Wi = (int)V, p.Ci = Cin, p.Do = 1, p.Ho = 1, p.Wo = (int)V, p.Cj = Cout;
    p.Rz = 1, p.Ry = 1, p.Rx = (int)V;
    p.osz = p.osy = p.osx = 1, p.ooz = p.ooy = p.oox = 0, p.isz = p.isy = p.isx = 1;
    p.ntaps = 1, p.tz[0] = p.ty[0] = p.tx[0] = 0;
    p.woff[0] = 0, p.wsi = 1, p.wsj = Cin;  // w[cout][cin]
    return launch_gconv<TI, TO>(p, (hipStream_t)stream);
}

extern "C" int u3d_conv1x1_fwd(int device, u3d_stream_t stream, const float* x, const float* w, const float* bias, float* y,
                               int N, int64_t V, int Cin, int Cout, double* out_stats) {
    return conv1x1_fwd_impl<float, float>(device, stream, x, w, bias, y, N, V, Cin, Cout, out_stats);
}

// bf16 activation storage: y is bf16; x is bf16, or fp32 when x_is_f32 (the network input feeding the first block)
extern "C" int u3d_conv1x1_fwd_b16(int device, u3d_stream_t stream, const void* x, int x_is_f32, const float* w, const float* bias,
                                   void* y, int N, int64_t V, int Cin, int Cout, double* out_stats) {
    if (x_is_f32 && smallc_ok(Cin, Cout) && x && w && y && N > 0 && V > 0 && ((uintptr_t)y & 15) == 0) {
        U3D_ENTER(device);
        const int rows = 256 / (Cout / 8);
        long long bx = (V + rows - 1) / rows;
        // (every block ends with 2 * Cout same-address f64 atomics, 19.5 ns each: 4096 blocks spent 80 us of this kernel's 117 us on them at
        // config 4's first block; rocprofv3 by total block count: 256: 149 us, 512: 105, 768: 94, 1024: 90, 1536: 88, 2048: 96, 4096: 117)
        const long long cap = 1280 / N > 1 ? 1280 / N : 1;
        if (bx > cap) bx = cap;
        const size_t sh = out_stats ? (size_t)rows * Cout * 2 * sizeof(float) : 0;
#define U3D_C1F(CI)                                                                                                            \
    hipLaunchKernelGGL(conv1x1_smallc_fwd_b16_kernel<CI>, dim3((unsigned)bx, (unsigned)N), dim3(256), sh, (hipStream_t)stream, \
                       (const float*)x, w, bias, (__bf16*)y, N, (long long)V, Cout, out_stats)
        if (Cin == 1) U3D_C1F(1);
        else if (Cin == 2) U3D_C1F(2);
        else if (Cin == 3) U3D_C1F(3);
        else U3D_C1F(4);
#undef U3D_C1F
        U3D_LAUNCH_CHECK();
        return 0;
    }
    if (x_is_f32) return conv1x1_fwd_impl<float, __bf16>(device, stream, x, w, bias, y, N, V, Cin, Cout, out_stats);
    return conv1x1_fwd_impl<__bf16, __bf16>(device, stream, x, w, bias, y, N, V, Cin, Cout, out_stats);
}

template <typename TX, typename T>
static int conv1x1_bwd_impl(int device, u3d_stream_t stream, const void* dy, const void* x, const float* w, int N, int64_t V, int Cin,
                            int Cout, void* dx, double* acc) {
    U3D_ENTER(device);
    U3D_REQUIRE(dy && x && w && acc && N > 0 && V > 0 && Cin > 0 && Cout > 0, "u3d_conv1x1_bwd: bad argument");
    U3D_REQUIRE(V < (1ll << 31) && (long long)N * V < (1ll << 31), "u3d_conv1x1_bwd: N*V must be < 2^31");
    hipStream_t st = (hipStream_t)stream;
    if (dx) {  // dx[v, c] = sum_k dy[v, k] * w[k][c]
        GConvParams p{};
        p.x = dy, p.w = w, p.bias = nullptr, p.mask = nullptr, p.out = dx, p.stats = nullptr;
        p.N = N, p.Di = 1, p.Hi = 1, p.Wi = (int)V, p.Ci = Cout, p.Do = 1, p.Ho = 1, p.Wo = (int)V, p.Cj = Cin;
        p.Rz = 1, p.Ry = 1, p.Rx = (int)V;
        p.osz = p.osy = p.osx = 1, p.ooz = p.ooy = p.oox = 0, p.isz = p.isy = p.isx = 1;
        p.ntaps = 1, p.tz[0] = p.ty[0] = p.tx[0] = 0;
        p.woff[0] = 0, p.wsi = Cin, p.wsj = 1;
        if (int e = launch_gconv<T, T>(p, st)) return e;
    }
    // dw[k][c] = sum_v dy[v,k] * x[v,c] -> acc[k*Cin + c];  db[k] = sum_v dy[v,k] -> acc[Cout*Cin + k]
    GWgradParams g{};
    g.X = x, g.Y = dy, g.acc = acc, g.bias_acc = acc + (size_t)Cout * Cin;
    g.N = N, g.Ca = Cin, g.Cb = Cout;
    g.Rz = 1, g.Ry = 1, g.Rx = (int)V, g.Dy = 1, g.Hy = 1, g.Wy = (int)V, g.ysz = g.ysy = g.ysx = 1;
    g.ntaps = 1, g.tz[0] = g.ty[0] = g.tx[0] = 0;
    g.dst_t = 0, g.dst_a = 1, g.dst_b = Cin;
    return launch_gwgrad<TX, T>(g, st);
}

extern "C" int u3d_conv1x1_bwd(int device, u3d_stream_t stream, const float* dy, const float* x, const float* w, int N,
                               int64_t V, int Cin, int Cout, float* dx, double* acc) {
    return conv1x1_bwd_impl<float, float>(device, stream, dy, x, w, N, V, Cin, Cout, dx, acc);
}

// bf16 activation storage: dy / dx are bf16; x is bf16, or fp32 when x_is_f32 (the network input: first block's conv1)
extern "C" int u3d_conv1x1_bwd_b16(int device, u3d_stream_t stream, const void* dy, const void* x, int x_is_f32, const float* w, int N,
                                   int64_t V, int Cin, int Cout, void* dx, double* acc) {
    if (x_is_f32 && dx == nullptr && smallc_ok(Cin, Cout) && dy && x && acc && N > 0 && V > 0 && ((uintptr_t)dy & 15) == 0) {
        U3D_ENTER(device);  // (no input gradient wanted: the usual case for the network input)
        const int rows = 256 / (Cout / 8);
        long long bx = ((long long)N * V + rows * 64 - 1) / ((long long)rows * 64);
        if (bx > 1024) bx = 1024;  // (512: 89 us, 1024 and beyond: 79 us at config 4's first block)
        if (bx < 1) bx = 1;
        const size_t sh = (size_t)rows * Cout * (Cin + 1) * sizeof(float);
#define U3D_C1B(CI)                                                                                                    \
    hipLaunchKernelGGL(conv1x1_smallc_bwd_b16_kernel<CI>, dim3((unsigned)bx), dim3(256), sh, (hipStream_t)stream,      \
                       (const __bf16*)dy, (const float*)x, N, (long long)V, Cout, acc)
        if (Cin == 1) U3D_C1B(1);
        else if (Cin == 2) U3D_C1B(2);
        else if (Cin == 3) U3D_C1B(3);
        else U3D_C1B(4);
#undef U3D_C1B
        U3D_LAUNCH_CHECK();
        return 0;
    }
    if (x_is_f32) return conv1x1_bwd_impl<float, __bf16>(device, stream, dy, x, w, N, V, Cin, Cout, dx, acc);
    return conv1x1_bwd_impl<__bf16, __bf16>(device, stream, dy, x, w, N, V, Cin, Cout, dx, acc);
}

// ---- ConvTranspose3d(k=3, stride=2, padding=1, bias=False): (N,D1,H1,W1,Cin) -> (N,2D1-1,2H1-1,2W1-1,Cout) ------------
// weight layout of nn.ConvTranspose3d: (Cin, Cout, 3, 3, 3); output coordinate s = 2i - 1 + t per dimension, so an even
// s = 2j takes (input j, tap 1) and an odd s = 2j+1 takes (input j+1, tap 0) and (input j, tap 2).
static int parity_taps(int pz, int py, int px, signed char* tz, signed char* ty, signed char* tx, int* woff) {
    static const int offs[2][2] = {{0, 0}, {1, 0}}, taps[2][2] = {{1, 1}, {0, 2}};
    int n = 0;
    for (int a = 0; a <= pz; ++a)
        for (int b = 0; b <= py; ++b)
            for (int c = 0; c <= px; ++c) {
                tz[n] = (signed char)offs[pz][a], ty[n] = (signed char)offs[py][b], tx[n] = (signed char)offs[px][c];
                woff[n] = (taps[pz][a] * 3 + taps[py][b]) * 3 + taps[px][c];
                ++n;
            }
    return n;
}

// packed image of the transposed-conv weights for coalesced B-tile loads:
//   mode 0 (forward):   out[tap][ci][co] = w[ci][co][tap]        mode 1 (data gradient): out[tap][co][ci] = w[ci][co][tap]
__global__ void pack_convtr_kernel(const float* __restrict__ w, float* __restrict__ out, int Cin, int Cout, int mode) {
    const long long total = (long long)27 * Cin * Cout;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int inner = mode == 0 ? Cout : Cin, outer = mode == 0 ? Cin : Cout;
        const int j = (int)(i % inner);
        const long long q = i / inner;
        const int k = (int)(q % outer), tap = (int)(q / outer);
        const int ci = mode == 0 ? k : j, co = mode == 0 ? j : k;
        out[i] = w[((size_t)ci * Cout + co) * 27 + tap];
    }
}

extern "C" int u3d_pack_convtr_weights(int device, u3d_stream_t stream, const float* w, int Cin, int Cout, int mode,
                                       float* packed) {
    U3D_ENTER(device);
    U3D_REQUIRE(w && packed && Cin > 0 && Cout > 0 && (mode == 0 || mode == 1), "u3d_pack_convtr_weights: bad argument");
    const long long total = (long long)27 * Cin * Cout;
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(pack_convtr_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, packed, Cin, Cout, mode);
    U3D_LAUNCH_CHECK();
    return 0;
}

extern "C" int u3d_convtr3d_fwd(int device, u3d_stream_t stream, const float* x, const float* w, float* t, int N, int D1,
                                int H1, int W1, int Cin, int Cout, const float* packed) {
    U3D_ENTER(device);
    U3D_REQUIRE(x && w && t && N > 0 && D1 > 0 && H1 > 0 && W1 > 0 && Cin > 0 && Cout > 0, "u3d_convtr3d_fwd: bad argument");
    const int Dt = 2 * D1 - 1, Ht = 2 * H1 - 1, Wt = 2 * W1 - 1;
    U3D_REQUIRE((long long)N * Dt * Ht * Wt < (1ll << 31), "u3d_convtr3d_fwd: output voxel count must be < 2^31");
    hipStream_t st = (hipStream_t)stream;
    for (int cls = 0; cls < 8; ++cls) {  // one gather GEMM per output parity class
        const int pz = (cls >> 2) & 1, py = (cls >> 1) & 1, px = cls & 1;
        GConvParams p{};
        p.x = x, p.w = w, p.bias = nullptr, p.mask = nullptr, p.out = t, p.stats = nullptr;
        p.N = N, p.Di = D1, p.Hi = H1, p.Wi = W1, p.Ci = Cin, p.Do = Dt, p.Ho = Ht, p.Wo = Wt, p.Cj = Cout;
        p.Rz = D1 - pz, p.Ry = H1 - py, p.Rx = W1 - px;  // s = 2j + p <= 2n - 2
        if (p.Rz <= 0 || p.Ry <= 0 || p.Rx <= 0) continue;
        p.osz = p.osy = p.osx = 2, p.ooz = pz, p.ooy = py, p.oox = px, p.isz = p.isy = p.isx = 1;
        p.ntaps = parity_taps(pz, py, px, p.tz, p.ty, p.tx, p.woff);
        p.wsi = (long long)Cout * 27, p.wsj = 27;  // element (ci, co, tap) at w[(ci*Cout + co)*27 + tap]
        if (packed) {                               // packed[tap][ci][co]
            p.w = packed, p.wsi = Cout, p.wsj = 1;
            for (int k = 0; k < p.ntaps; ++k) p.woff[k] *= Cin * Cout;
        }
        if (int e = launch_gconv(p, st)) return e;
    }
    return 0;
}

// dt: gradient w.r.t. the transposed conv's output (N,2D1-1,2H1-1,2W1-1,Cout).
//   dx[i, ci]      = sum_{t, co} dt[2i - 1 + t, co] * w[ci, co, t]      (a stride-2 3x3x3 convolution of dt), masked by
//                    x > 0 when relu_mask (x is the post-ReLU output of the block that produced it)
//   acc[(ci*Cout + co)*27 + t] += sum_i x[i, ci] * dt[2i - 1 + t, co]   (zeroed double scratch; u3d_cvt_f64_f32)
extern "C" int u3d_convtr3d_bwd(int device, u3d_stream_t stream, const float* dt, const float* x, const float* w, int N,
                                int D1, int H1, int W1, int Cin, int Cout, int relu_mask, float* dx, double* acc,
                                const float* packed_t) {
    U3D_ENTER(device);
    U3D_REQUIRE(dt && x && w && acc && N > 0 && D1 > 0 && H1 > 0 && W1 > 0 && Cin > 0 && Cout > 0, "u3d_convtr3d_bwd: bad argument");
    const int Dt = 2 * D1 - 1, Ht = 2 * H1 - 1, Wt = 2 * W1 - 1;
    U3D_REQUIRE((long long)N * Dt * Ht * Wt < (1ll << 31), "u3d_convtr3d_bwd: output voxel count must be < 2^31");
    hipStream_t st = (hipStream_t)stream;
    if (dx) {
        GConvParams p{};
        p.x = dt, p.w = w, p.bias = nullptr, p.mask = relu_mask ? x : nullptr, p.out = dx, p.stats = nullptr;
        p.N = N, p.Di = Dt, p.Hi = Ht, p.Wi = Wt, p.Ci = Cout, p.Do = D1, p.Ho = H1, p.Wo = W1, p.Cj = Cin;
        p.Rz = D1, p.Ry = H1, p.Rx = W1;
        p.osz = p.osy = p.osx = 1, p.ooz = p.ooy = p.oox = 0, p.isz = p.isy = p.isx = 2;
        p.ntaps = 27;
        for (int tp = 0; tp < 27; ++tp) {
            p.tz[tp] = (signed char)(tp / 9 - 1), p.ty[tp] = (signed char)((tp / 3) % 3 - 1), p.tx[tp] = (signed char)(tp % 3 - 1);
            p.woff[tp] = tp;
        }
        p.wsi = 27, p.wsj = (long long)Cout * 27;  // (tap, in = co, out = ci) at w[(ci*Cout + co)*27 + tap]
        if (packed_t) {                              // packed_t[tap][co][ci]
            p.w = packed_t, p.wsi = Cin, p.wsj = 1;
            for (int tp = 0; tp < 27; ++tp) p.woff[tp] = tp * Cin * Cout;
        }
        if (int e = launch_gconv(p, st)) return e;
    }
    GWgradParams g{};
    g.X = x, g.Y = dt, g.acc = acc, g.bias_acc = nullptr;
    g.N = N, g.Ca = Cin, g.Cb = Cout;
    g.Rz = D1, g.Ry = H1, g.Rx = W1, g.Dy = Dt, g.Hy = Ht, g.Wy = Wt, g.ysz = g.ysy = g.ysx = 2;
    g.ntaps = 27;
    for (int tp = 0; tp < 27; ++tp)
        g.tz[tp] = (signed char)(tp / 9 - 1), g.ty[tp] = (signed char)((tp / 3) % 3 - 1), g.tx[tp] = (signed char)(tp % 3 - 1);
    g.dst_t = 1, g.dst_a = (long long)Cout * 27, g.dst_b = 27;
    return launch_gwgrad(g, st);
}

// ---- nearest resize to the skip's size + summation joining (buildingblocks.py:650-651 + :493) ---------------------------
template <typename T>
static int nearest_add_impl(int device, u3d_stream_t stream, const T* skip, const T* t, const int32_t* zmap,
                            const int32_t* ymap, const int32_t* xmap, int N, int D, int H, int W, int Dt, int Ht, int Wt, int C,
                            T* out, double* out_stats, int t8);

extern "C" int u3d_nearest_add_fwd(int device, u3d_stream_t stream, const float* skip, const float* t, const int32_t* zmap,
                                   const int32_t* ymap, const int32_t* xmap, int N, int D, int H, int W, int Dt, int Ht,
                                   int Wt, int C, float* out, double* out_stats) {
    return nearest_add_impl<float>(device, stream, skip, t, zmap, ymap, xmap, N, D, H, W, Dt, Ht, Wt, C, out, out_stats, 0);
}

extern "C" int u3d_nearest_add_fwd_t8(int device, u3d_stream_t stream, const float* skip, const float* t8, const int32_t* zmap,
                                      const int32_t* ymap, const int32_t* xmap, int N, int D, int H, int W, int Dt, int Ht,
                                      int Wt, int C, float* out, double* out_stats) {
    return nearest_add_impl<float>(device, stream, skip, t8, zmap, ymap, xmap, N, D, H, W, Dt, Ht, Wt, C, out, out_stats, 1);
}

template <typename T>
static int nearest_add_impl(int device, u3d_stream_t stream, const T* skip, const T* t, const int32_t* zmap,
                            const int32_t* ymap, const int32_t* xmap, int N, int D, int H, int W, int Dt, int Ht, int Wt, int C,
                            T* out, double* out_stats, int t8) {
    U3D_ENTER(device);
    U3D_REQUIRE(skip && t && zmap && ymap && xmap && out && N > 0 && D > 0 && H > 0 && W > 0 && Dt > 0 && Ht > 0 && Wt > 0 &&
                    C > 0 && C <= 1024,
                "u3d_nearest_add_fwd: bad argument (C <= 1024)");
    if constexpr (std::is_same<T, __bf16>::value) {
        const int oct = C / 8;
        const long long V32 = (long long)D * H * W;
        if (t8 && C % 8 == 0 && 256 % oct == 0 && V32 < (1ll << 31) && (((uintptr_t)skip | (uintptr_t)t | (uintptr_t)out) & 15) == 0) {
            const int rows = 256 / oct;
            long long bx = (V32 + rows * 4 - 1) / (rows * 4);
            // (every block ends with 2 C f64 atomics on the same N x C x 2 addresses, ~24 ns each per address: 16384 one-pass blocks made
            // this kernel 0.58 ms per step, slower than the generic one; two to four resident blocks per CU walk the volume instead)
            const long long cap = 1024 / N > 1 ? 1024 / N : 1;
            if (bx > cap) bx = cap;
            hipLaunchKernelGGL(nearest_add_t8_b16_oct_kernel, dim3((unsigned)bx, (unsigned)N), dim3(256), sizeof(double) * 2 * (size_t)C,
                               (hipStream_t)stream, skip, t, zmap, ymap, xmap, D, H, W, (Dt + 1) / 2, (Ht + 1) / 2, (Wt + 1) / 2, C, oct, out,
                               out_stats);
            U3D_LAUNCH_CHECK();
            return 0;
        }
    }
    const int vecw = (C % 4 == 0 && (((uintptr_t)skip | (uintptr_t)t | (uintptr_t)out) & u3d_vec_align<T>::mask) == 0) ? 4 : 1;
    int Q = C / vecw;
    U3D_REQUIRE(Q <= 256, "u3d_nearest_add_fwd: more than 256 channel units per voxel (C %% 4 != 0 with C > 256)");
    const int rows = 256 / Q;
    const long long V = (long long)D * H * W;
    long long bx = (V + rows - 1) / rows;
    const long long cap = 2048 / N > 1 ? 2048 / N : 1;
    if (bx > cap) bx = cap;
    hipLaunchKernelGGL(nearest_add_kernel<T>, dim3((unsigned)bx, (unsigned)N), dim3(256), sizeof(double) * 2 * (size_t)C,
                       (hipStream_t)stream, skip, t, zmap, ymap, xmap, D, H, W, Dt, Ht, Wt, C, Q, vecw, t8, out, out_stats);
    U3D_LAUNCH_CHECK();
    return 0;
}

template <typename T>
static int nearest_sum_impl(int device, u3d_stream_t stream, const T* dj, const int32_t* zlo, const int32_t* ylo,
                            const int32_t* xlo, int N, int D, int H, int W, int Dt, int Ht, int Wt, int C, T* dt, int t8);

extern "C" int u3d_nearest_sum_bwd(int device, u3d_stream_t stream, const float* dj, const int32_t* zlo, const int32_t* ylo,
                                   const int32_t* xlo, int N, int D, int H, int W, int Dt, int Ht, int Wt, int C, float* dt) {
    return nearest_sum_impl<float>(device, stream, dj, zlo, ylo, xlo, N, D, H, W, Dt, Ht, Wt, C, dt, 0);
}

extern "C" int u3d_nearest_sum_bwd_t8(int device, u3d_stream_t stream, const float* dj, const int32_t* zlo, const int32_t* ylo,
                                      const int32_t* xlo, int N, int D, int H, int W, int Dt, int Ht, int Wt, int C, float* dt8) {
    return nearest_sum_impl<float>(device, stream, dj, zlo, ylo, xlo, N, D, H, W, Dt, Ht, Wt, C, dt8, 1);
}

// bf16 activation storage (space-to-depth layout only: the bf16 path's transposed convolutions)
extern "C" int u3d_nearest_add_fwd_t8_b16(int device, u3d_stream_t stream, const void* skip, const void* t8, const int32_t* zmap,
                                          const int32_t* ymap, const int32_t* xmap, int N, int D, int H, int W, int Dt, int Ht, int Wt,
                                          int C, void* out, double* out_stats) {
    return nearest_add_impl<__bf16>(device, stream, (const __bf16*)skip, (const __bf16*)t8, zmap, ymap, xmap, N, D, H, W, Dt, Ht, Wt, C,
                                    (__bf16*)out, out_stats, 1);
}

extern "C" int u3d_nearest_sum_bwd_t8_b16(int device, u3d_stream_t stream, const void* dj, const int32_t* zlo, const int32_t* ylo,
                                          const int32_t* xlo, int N, int D, int H, int W, int Dt, int Ht, int Wt, int C, void* dt8) {
    return nearest_sum_impl<__bf16>(device, stream, (const __bf16*)dj, zlo, ylo, xlo, N, D, H, W, Dt, Ht, Wt, C, (__bf16*)dt8, 1);
}

template <typename T>
static int nearest_sum_impl(int device, u3d_stream_t stream, const T* dj, const int32_t* zlo, const int32_t* ylo,
                            const int32_t* xlo, int N, int D, int H, int W, int Dt, int Ht, int Wt, int C, T* dt, int t8) {
    U3D_ENTER(device);
    U3D_REQUIRE(dj && zlo && ylo && xlo && dt && N > 0 && C > 0, "u3d_nearest_sum_bwd: bad argument");
    const long long total = t8 ? (long long)N * ((Dt + 1) / 2) * ((Ht + 1) / 2) * ((Wt + 1) / 2) * 8 * C : (long long)N * Dt * Ht * Wt * C;
    if constexpr (std::is_same<T, __bf16>::value) {
        if (t8 && C % 8 == 0 && (((uintptr_t)dj | (uintptr_t)dt) & 15) == 0) {
            long long vb = (total / 8 + 255) / 256;
            if (vb > 16384) vb = 16384;
            hipLaunchKernelGGL(nearest_sum_t8_b16_vec_kernel, dim3((unsigned)vb), dim3(256), 0, (hipStream_t)stream, dj, zlo, ylo, xlo, N,
                               D, H, W, Dt, Ht, Wt, C, dt);
            U3D_LAUNCH_CHECK();
            return 0;
        }
    }
    long long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(nearest_sum_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dj, zlo, ylo, xlo, N, D, H,
                       W, Dt, Ht, Wt, C, t8, dt);
    U3D_LAUNCH_CHECK();
    return 0;
}


// ---- concat joining for residual decoders with an EXPLICIT upsample='deconv' (buildingblocks.py:435-468: only 'default' selects
// summation joining; an explicit 'deconv' keeps torch.cat((skip, upsampled)) and the block's 1x1x1 conv maps the 2x channels back).
// The joined tensor feeds a 1x1x1 convolution, so it is materialised: out[..., :Cs] = skip, out[..., Cs:] = t[zmap, ymap, xmap].
namespace {

__global__ void nearest_cat_kernel(const float* __restrict__ skip, const float* __restrict__ t, const int32_t* __restrict__ zmap,
                                   const int32_t* __restrict__ ymap, const int32_t* __restrict__ xmap, int N, int D, int H, int W,
                                   int Dt, int Ht, int Wt, int Cs, int Ct, float* __restrict__ out) {
    const int C = Cs + Ct;
    const long long total = (long long)N * D * H * W * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long long v = i / C;
        if (c < Cs) {
            out[i] = skip[v * Cs + c];
            continue;
        }
        const int x = (int)(v % W);
        long long r = v / W;
        const int y = (int)(r % H);
        r /= H;
        const int z = (int)(r % D);
        const int n = (int)(r / D);
        out[i] = t[((((size_t)n * Dt + zmap[z]) * Ht + ymap[y]) * Wt + xmap[x]) * Ct + (c - Cs)];
    }
}

// channel QUADS (Cs % 4 == 0, Ct % 4 == 0, 16-byte aligned tensors): one (voxel, quad) per thread, 16-byte accesses; the (n, z, y, x)
// decomposition is 32-bit (N*D*H*W < 2^31).  The element-wise kernel above wrote config 2's three decoder concats at 1.5 TB/s.
__global__ __launch_bounds__(256) void nearest_cat_quad_kernel(const float* __restrict__ skip, const float* __restrict__ t, const int32_t* __restrict__ zmap,
                                                               const int32_t* __restrict__ ymap, const int32_t* __restrict__ xmap, int NV, int D,
                                                               int H, int W, int Dt, int Ht, int Wt, int Cs, int Ct, float* __restrict__ out) {
    const int Q = (Cs + Ct) >> 2, Qs = Cs >> 2;
    const long long total = (long long)NV * Q;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int v = (int)(i / Q), q = (int)(i - (long long)v * Q);
        f32x4 val;
        if (q < Qs) {
            val = *reinterpret_cast<const f32x4*>(skip + (size_t)v * Cs + 4 * q);
        } else {
            const int x = v % W;
            int r = v / W;
            const int y = r % H;
            r /= H;
            const int z = r % D, n = r / D;
            val = *reinterpret_cast<const f32x4*>(t + ((((size_t)n * Dt + zmap[z]) * Ht + ymap[y]) * Wt + xmap[x]) * Ct + 4 * (q - Qs));
        }
        *reinterpret_cast<f32x4*>(out + (size_t)i * 4) = val;
    }
}

__global__ void split_channels_kernel(const float* __restrict__ x, long long rows, int C0, int C1, float* __restrict__ out0,
                                      float* __restrict__ out1) {
    const int C = C0 + C1;
    const long long total = rows * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long long v = i / C;
        if (c < C0)
            out0[v * C0 + c] = x[i];
        else
            out1[v * C1 + (c - C0)] = x[i];
    }
}

}  // namespace

extern "C" int u3d_nearest_cat_fwd(int device, u3d_stream_t stream, const float* skip, const float* t, const int32_t* zmap,
                                   const int32_t* ymap, const int32_t* xmap, int N, int D, int H, int W, int Dt, int Ht, int Wt,
                                   int Cs, int Ct, float* out) {
    U3D_ENTER(device);
    U3D_REQUIRE(skip && t && zmap && ymap && xmap && out && N > 0 && D > 0 && H > 0 && W > 0 && Dt > 0 && Ht > 0 && Wt > 0 && Cs > 0 &&
                    Ct > 0, "u3d_nearest_cat_fwd: bad argument");
    const long long total = (long long)N * D * H * W * (Cs + Ct);
    if (Cs % 4 == 0 && Ct % 4 == 0 && (((uintptr_t)skip | (uintptr_t)t | (uintptr_t)out) & 15) == 0 && (long long)N * D * H * W < (1ll << 31)) {
        long long qb = (total / 4 + 255) / 256;
        if (qb > 65536) qb = 65536;
        hipLaunchKernelGGL(nearest_cat_quad_kernel, dim3((unsigned)qb), dim3(256), 0, (hipStream_t)stream, skip, t, zmap, ymap, xmap,
                           (int)((long long)N * D * H * W), D, H, W, Dt, Ht, Wt, Cs, Ct, out);
        U3D_LAUNCH_CHECK();
        return 0;
    }
    long long blocks = (total + 255) / 256;
    if (blocks > 32768) blocks = 32768;
    hipLaunchKernelGGL(nearest_cat_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, skip, t, zmap, ymap, xmap, N, D, H,
                       W, Dt, Ht, Wt, Cs, Ct, out);
    U3D_LAUNCH_CHECK();
    return 0;
}

extern "C" int u3d_split_channels(int device, u3d_stream_t stream, const float* x, int64_t rows, int C0, int C1, float* out0,
                                  float* out1) {
    U3D_ENTER(device);
    U3D_REQUIRE(x && out0 && out1 && rows > 0 && C0 > 0 && C1 > 0, "u3d_split_channels: bad argument");
    const long long total = (long long)rows * (C0 + C1);
    long long blocks = (total + 255) / 256;
    if (blocks > 32768) blocks = 32768;
    hipLaunchKernelGGL(split_channels_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (long long)rows, C0, C1, out0,
                       out1);
    U3D_LAUNCH_CHECK();
    return 0;
}
