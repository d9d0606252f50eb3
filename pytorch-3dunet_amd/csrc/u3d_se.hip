// u3d_se.hip — squeeze-and-excitation gates of ResNetBlockSE.
//
// Reference: pytorch3dunet/unet3d/se.py — ChannelSELayer3D :18-51 (cSE: global average -> fc1 + bias -> ReLU -> fc2 + bias ->
// sigmoid -> per-channel scale), SpatialSELayer3D :54-93 (sSE: 1x1x1 conv C->1 + bias -> sigmoid -> per-voxel scale),
// ChannelSpatialSELayer3D :96-114 (scSE = elementwise max of the two), attached after the residual block by
// ResNetBlockSE (buildingblocks.py:291-307, reduction_ratio = 1).
//
// Both gates are positive, so max(y * gc[n,c], y * a[n,v]) = y * max(gc, a) where y >= 0 (always, for ReLU blocks) and
// y * min(gc, a) where y < 0 (LeakyReLU / ELU blocks).
// Forward: the channel means come for free from conv3's fused statistics; a tiny kernel runs the two FC layers per sample;
// ONE bandwidth pass computes the spatial gate (a dot product over the voxel's channels, lanes of a wave share a voxel)
// and applies max(gc, a).  Backward: one reduction pass (d gc, d ws, d bs, d logit_s per voxel), the FC backward, and one
// elementwise pass that also applies the ReLU mask of the block output.  modes: 0 scSE, 1 cSE only, 2 sSE only.
#include "u3d_common.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float x) {
    const float e = expf(-fabsf(x));
    const float r = 1.f / (1.f + e);
    return x >= 0.f ? r : e * r;
}

// ---- channel gate, two launches of the same row kernel: out[n][r] = act(b[r] + <w[r,:], in[n,:]>).  grid (ceil(R/16), N),
//      256 threads = 4 waves x 4 rows each; lanes stride the row (coalesced), butterfly reduction.
//      stage 0: in = channel means from the fused statistics (also written to s_out), act = ReLU   (fc1, se.py:43-46)
//      stage 1: in = h, act = sigmoid                                                             (fc2, se.py:47)
__global__ __launch_bounds__(256) void se_fc_rows_kernel(const double* __restrict__ ystats, double count,
                                                         const float* __restrict__ in, const float* __restrict__ w,
                                                         const float* __restrict__ b, int K, int R, int stage,
                                                         float* __restrict__ s_out, float* __restrict__ out) {
    __shared__ float sv[1024];
    const int n = blockIdx.y, t = threadIdx.x, l = t & 63, wv = t >> 6;
    for (int c = t; c < K; c += 256) {
        float m;
        if (stage == 0) {
            m = (float)(ystats[((size_t)n * K + c) * 2] / count);  // AdaptiveAvgPool3d(1), se.py:40
            if (blockIdx.x == 0) s_out[(size_t)n * K + c] = m;
        } else {
            m = in[(size_t)n * K + c];
        }
        sv[c] = m;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = blockIdx.x * 16 + wv * 4 + i;
        if (r >= R) break;
        float p = 0.f;
        for (int c = l; c < K; c += 64) p += w[(size_t)r * K + c] * sv[c];
        for (int m = 32; m > 0; m >>= 1) p += __shfl_xor(p, m);
        if (l == 0) out[(size_t)n * R + r] = stage == 0 ? fmaxf(p + b[r], 0.f) : sigmoidf_(p + b[r]);
    }
}

// ---- spatial gate + apply: LPV lanes share one voxel, lane `sub` owns channel quads sub, sub+LPV, ... (<= 4 of them) ----
// (T: storage type of the activation tensors y / out / dout — float, or __bf16 for `activation_dtype: bf16`; gates and sums stay fp32)
template <int LPV, typename T = float>
__global__ __launch_bounds__(256) void se_apply_fwd_kernel(const T* __restrict__ y, const float* __restrict__ gc,
                                                           const float* __restrict__ ws, const float* __restrict__ bs, int N,
                                                           long long V, int C, int mode, T* __restrict__ out,
                                                           float* __restrict__ a_out) {
    const int t = threadIdx.x, sub = t & (LPV - 1);
    const int Q = C >> 2;
    const long long total = (long long)N * V, vpb = 256 / LPV;
    f32x4 wq[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int q = sub + it * LPV;
        wq[it] = (mode != 1 && q < Q) ? *reinterpret_cast<const f32x4*>(ws + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float bias = mode != 1 ? bs[0] : 0.f;
    for (long long idx = (long long)blockIdx.x * vpb + t / LPV; idx < total; idx += (long long)gridDim.x * vpb) {
        const int n = (int)(idx / V);
        f32x4 yv[4];
        float dot = 0.f;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int q = sub + it * LPV;
            yv[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (q < Q) {
                yv[it] = u3d_ldq(y + (size_t)idx * C + 4 * q);
                dot += yv[it][0] * wq[it][0] + yv[it][1] * wq[it][1] + yv[it][2] * wq[it][2] + yv[it][3] * wq[it][3];
            }
        }
#pragma unroll
        for (int m = LPV >> 1; m > 0; m >>= 1) dot += __shfl_xor(dot, m);
        const float a = mode != 1 ? sigmoidf_(dot + bias) : 0.f;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int q = sub + it * LPV;
            if (q < Q) {
                f32x4 g = {a, a, a, a};
                if (mode != 2) {
                    const f32x4 gv = *reinterpret_cast<const f32x4*>(gc + (size_t)n * C + 4 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) g[e] = mode == 0 ? (yv[it][e] >= 0.f ? fmaxf(gv[e], a) : fminf(gv[e], a)) : gv[e];
                }
                u3d_stq(out + (size_t)idx * C + 4 * q, yv[it] * g);
            }
        }
        if (a_out && sub == 0) a_out[idx] = a;
    }
}

// ---- backward reduction: grid (bx, N).  Per voxel: split dout*y between the two gates (torch.max backward: the larger
//      operand takes the gradient, a tie halves it), d logit_s = da * a(1-a); per-thread register accumulators for
//      d gc[n, c], d ws[c] over the block's voxels, folded through LDS (f64) into one global f64 atomic per block --------
template <int LPV, typename T = float>
__global__ __launch_bounds__(256) void se_bwd_reduce_kernel(const T* __restrict__ dout, const T* __restrict__ y,
                                                            const float* __restrict__ gc, const float* __restrict__ a_in,
                                                            const float* __restrict__ ws, long long V, int C, int mode,
                                                            float* __restrict__ dls, double* __restrict__ acc_gc,
                                                            double* __restrict__ acc_ws) {
    __shared__ double sgc[1024], sws[1024];
    __shared__ double sbs;
    const int n = blockIdx.y, t = threadIdx.x, sub = t & (LPV - 1);
    const int Q = C >> 2;
    const long long vpb = 256 / LPV;
    for (int c = t; c < C; c += 256) {
        sgc[c] = 0.0;
        sws[c] = 0.0;
    }
    if (t == 0) sbs = 0.0;
    __syncthreads();
    f32x4 gq[4], dg[4], dw[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int q = sub + it * LPV;
        gq[it] = (mode != 2 && q < Q) ? *reinterpret_cast<const f32x4*>(gc + (size_t)n * C + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
        dg[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        dw[it] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float dbs = 0.f;
    for (long long v = (long long)blockIdx.x * vpb + t / LPV; v < V; v += (long long)gridDim.x * vpb) {
        const size_t idx = (size_t)n * V + v;
        const float a = mode != 1 ? a_in[idx] : 0.f;
        f32x4 yv[4];
        float da = 0.f;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int q = sub + it * LPV;
            yv[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (q < Q) {
                yv[it] = u3d_ldq(y + idx * C + 4 * q);
                const f32x4 d = u3d_ldq(dout + idx * C + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float tv = d[e] * yv[it][e];
                    float to_g = tv, to_a = 0.f;
                    if (mode == 2) {
                        to_g = 0.f, to_a = tv;
                    } else if (mode == 0) {
                        const float g = gq[it][e];
                        const bool pick_g = yv[it][e] >= 0.f ? g > a : g < a;  // which product is the larger one
                        to_g = pick_g ? tv : (g == a ? 0.5f * tv : 0.f);
                        to_a = tv - to_g;
                    }
                    dg[it][e] += to_g;
                    da += to_a;
                }
            }
        }
        if (mode != 1) {
#pragma unroll
            for (int m = LPV >> 1; m > 0; m >>= 1) da += __shfl_xor(da, m);
            const float dl = da * a * (1.f - a);
            if (sub == 0) {
                dls[idx] = dl;
                dbs += dl;
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) dw[it] += yv[it] * dl;
        }
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int q = sub + it * LPV;
        if (q < Q) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (mode != 2) __hip_atomic_fetch_add(&sgc[4 * q + e], (double)dg[it][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (mode != 1) __hip_atomic_fetch_add(&sws[4 * q + e], (double)dw[it][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    if (mode != 1 && sub == 0) __hip_atomic_fetch_add(&sbs, (double)dbs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    for (int c = t; c < C; c += 256) {
        if (mode != 2) u3d_atomic_add_f64(&acc_gc[(size_t)n * C + c], sgc[c]);
        if (mode != 1) u3d_atomic_add_f64(&acc_ws[c], sws[c]);
    }
    if (mode != 1 && t == 0) u3d_atomic_add_f64(&acc_ws[C], sbs);
}

// ---- FC backward, per sample (grid N): dz2, dh -> dz1, ds -------------------------------------------------------------
__global__ __launch_bounds__(256) void se_gate_bwd_kernel(const double* __restrict__ acc_gc, const float* __restrict__ gc,
                                                          const float* __restrict__ h, const float* __restrict__ w1,
                                                          const float* __restrict__ w2, int C, int Cr, double count,
                                                          float* __restrict__ dz2, float* __restrict__ dz1,
                                                          float* __restrict__ ds) {
    __shared__ float s2[1024], s1[1024];
    const int n = blockIdx.x, t = threadIdx.x;
    for (int c = t; c < C; c += 256) {
        const float g = gc[(size_t)n * C + c];
        const float v = (float)acc_gc[(size_t)n * C + c] * g * (1.f - g);
        s2[c] = v;
        dz2[(size_t)n * C + c] = v;
    }
    __syncthreads();
    for (int j = t; j < Cr; j += 256) {  // dh[j] = sum_c w2[c][j] * dz2[c]: consecutive threads read consecutive j
        float p = 0.f;
        for (int c = 0; c < C; ++c) p += w2[(size_t)c * Cr + j] * s2[c];
        const float v = h[(size_t)n * Cr + j] > 0.f ? p : 0.f;
        s1[j] = v;
        dz1[(size_t)n * Cr + j] = v;
    }
    __syncthreads();
    for (int c = t; c < C; c += 256) {  // ds[c] = (1/V) sum_j w1[j][c] * dz1[j]
        float p = 0.f;
        for (int j = 0; j < Cr; ++j) p += w1[(size_t)j * C + c] * s1[j];
        ds[(size_t)n * C + c] = (float)((double)p / count);
    }
}

// parameter gradients of the two FC layers: fixed summation order over n, direct writes
__global__ void se_gate_wgrad_kernel(const float* __restrict__ dz2, const float* __restrict__ dz1, const float* __restrict__ h,
                                     const float* __restrict__ s, int N, int C, int Cr, float* __restrict__ dw1,
                                     float* __restrict__ db1, float* __restrict__ dw2, float* __restrict__ db2) {
    const long long nw = (long long)C * Cr;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * nw + C + Cr; i += (long long)gridDim.x * blockDim.x) {
        float v = 0.f;
        if (i < nw) {  // dw2[c][j]
            const int c = (int)(i / Cr), j = (int)(i % Cr);
            for (int n = 0; n < N; ++n) v += dz2[(size_t)n * C + c] * h[(size_t)n * Cr + j];
            dw2[i] = v;
        } else if (i < 2 * nw) {  // dw1[j][c]
            const long long k = i - nw;
            const int j = (int)(k / C), c = (int)(k % C);
            for (int n = 0; n < N; ++n) v += dz1[(size_t)n * Cr + j] * s[(size_t)n * C + c];
            dw1[k] = v;
        } else if (i < 2 * nw + C) {
            const int c = (int)(i - 2 * nw);
            for (int n = 0; n < N; ++n) v += dz2[(size_t)n * C + c];
            db2[c] = v;
        } else {
            const int j = (int)(i - 2 * nw - C);
            for (int n = 0; n < N; ++n) v += dz1[(size_t)n * Cr + j];
            db1[j] = v;
        }
    }
}

// m = (dout * gate + dls[v] * ws[c] + ds[n,c]) * (y > 0)
template <typename T = float>
__global__ void se_bwd_apply_kernel(const T* __restrict__ dout, const T* __restrict__ y, const float* __restrict__ gc,
                                    const float* __restrict__ a_in, const float* __restrict__ ws, const float* __restrict__ dls,
                                    const float* __restrict__ ds, int N, long long V, int C, int mode, int relu_mask,
                                    T* __restrict__ out) {
    const int Q = C >> 2;
    const long long total = (long long)N * V * Q;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int q = (int)(i % Q);
        const long long idx = i / Q;
        const int n = (int)(idx / V);
        const f32x4 d = u3d_ldq(dout + (size_t)idx * C + 4 * q);
        const f32x4 yv = u3d_ldq(y + (size_t)idx * C + 4 * q);
        const float a = mode != 1 ? a_in[idx] : 0.f;
        const float dl = mode != 1 ? dls[idx] : 0.f;
        f32x4 g = {a, a, a, a}, wv = {0.f, 0.f, 0.f, 0.f}, dsv = {0.f, 0.f, 0.f, 0.f};
        if (mode != 2) {
            const f32x4 gv = *reinterpret_cast<const f32x4*>(gc + (size_t)n * C + 4 * q);
            dsv = *reinterpret_cast<const f32x4*>(ds + (size_t)n * C + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = mode == 0 ? (yv[e] >= 0.f ? fmaxf(gv[e], a) : fminf(gv[e], a)) : gv[e];
        }
        if (mode != 1) wv = *reinterpret_cast<const f32x4*>(ws + 4 * q);
        f32x4 o = d * g + wv * dl + dsv;
        if (relu_mask) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = yv[e] > 0.f ? o[e] : 0.f;
        }
        u3d_stq(out + (size_t)idx * C + 4 * q, o);
    }
}

inline int lanes_per_voxel(int Q) {  // largest power of two <= min(64, Q) (at least 4): ceil(Q / lpv) <= 4 for Q <= 256
    int l = 64;
    while (l > 4 && l > Q) l >>= 1;
    return l;
}

}  // namespace

#define U3D_SE_DISPATCH(KERNEL, T_, LPV_, ...)                                                          \
    do {                                                                                                \
        if (LPV_ == 4)                                                                                  \
            hipLaunchKernelGGL((KERNEL<4, T_>), __VA_ARGS__);                                           \
        else if (LPV_ == 8)                                                                             \
            hipLaunchKernelGGL((KERNEL<8, T_>), __VA_ARGS__);                                           \
        else if (LPV_ == 16)                                                                            \
            hipLaunchKernelGGL((KERNEL<16, T_>), __VA_ARGS__);                                          \
        else if (LPV_ == 32)                                                                            \
            hipLaunchKernelGGL((KERNEL<32, T_>), __VA_ARGS__);                                          \
        else                                                                                            \
            hipLaunchKernelGGL((KERNEL<64, T_>), __VA_ARGS__);                                          \
    } while (0)

static int se_check(int N, int64_t V, int C, int mode, const char* what) {
    U3D_REQUIRE(N > 0 && V > 0 && C >= 4 && C % 4 == 0 && C <= 1024 && mode >= 0 && mode <= 2,
                "%s: needs C %% 4 == 0, 4 <= C <= 1024, mode in {0,1,2}", what);
    U3D_REQUIRE((long long)N * V < (1ll << 31), "%s: N*V must be < 2^31", what);
    return 0;
}

extern "C" int u3d_se_gate_fwd(int device, u3d_stream_t stream, const double* ystats, double count, const float* w1,
                               const float* b1, const float* w2, const float* b2, int N, int C, int Cr, float* s, float* h,
                               float* gc) {
    U3D_ENTER(device);
    U3D_REQUIRE(ystats && w1 && b1 && w2 && b2 && s && h && gc && N > 0 && C > 0 && C <= 1024 && Cr > 0 && Cr <= 1024 && count > 0,
                "u3d_se_gate_fwd: bad argument (C, Cr <= 1024)");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(se_fc_rows_kernel, dim3((Cr + 15) / 16, N), dim3(256), 0, st, ystats, count, nullptr, w1, b1, C, Cr, 0, s, h);
    U3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(se_fc_rows_kernel, dim3((C + 15) / 16, N), dim3(256), 0, st, nullptr, count, h, w2, b2, Cr, C, 1, nullptr, gc);
    U3D_LAUNCH_CHECK();
    return 0;
}

template <typename T>
static int se_apply_fwd_impl(int device, u3d_stream_t stream, const T* y, const float* gc, const float* ws, const float* bs, int N,
                             int64_t V, int C, int mode, T* out, float* a) {
    U3D_ENTER(device);
    if (int e = se_check(N, V, C, mode, "u3d_se_apply_fwd")) return e;
    U3D_REQUIRE(y && out && (mode == 2 || gc) && (mode == 1 || (ws && bs)), "u3d_se_apply_fwd: missing gate inputs");
    const int lpv = lanes_per_voxel(C / 4);
    const long long total = (long long)N * V, vpb = 256 / lpv;
    long long blocks = (total + vpb - 1) / vpb;
    if (blocks > 8192) blocks = 8192;
    U3D_SE_DISPATCH(se_apply_fwd_kernel, T, lpv, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, y, gc, ws, bs, N,
                    (long long)V, C, mode, out, a);
    U3D_LAUNCH_CHECK();
    return 0;
}

extern "C" int u3d_se_apply_fwd(int device, u3d_stream_t stream, const float* y, const float* gc, const float* ws,
                                const float* bs, int N, int64_t V, int C, int mode, float* out, float* a) {
    return se_apply_fwd_impl<float>(device, stream, y, gc, ws, bs, N, V, C, mode, out, a);
}

// bf16 activation storage (y, out: bf16 NDHWC; the spatial gate `a` and every table stay fp32)
extern "C" int u3d_se_apply_fwd_b16(int device, u3d_stream_t stream, const void* y, const float* gc, const float* ws,
                                    const float* bs, int N, int64_t V, int C, int mode, void* out, float* a) {
    return se_apply_fwd_impl<__bf16>(device, stream, (const __bf16*)y, gc, ws, bs, N, V, C, mode, (__bf16*)out, a);
}

template <typename T>
static int se_bwd_reduce_impl(int device, u3d_stream_t stream, const T* dout, const T* y, const float* gc, const float* a,
                              const float* ws, int N, int64_t V, int C, int mode, float* dls, double* acc_gc, double* acc_ws) {
    U3D_ENTER(device);
    if (int e = se_check(N, V, C, mode, "u3d_se_bwd_reduce")) return e;
    U3D_REQUIRE(dout && y && (mode == 2 || (gc && acc_gc)) && (mode == 1 || (a && ws && dls && acc_ws)),
                "u3d_se_bwd_reduce: missing argument");
    const int lpv = lanes_per_voxel(C / 4);
    const long long vpb = 256 / lpv;
    long long bx = (V + vpb - 1) / vpb;
    const long long cap = 1024 / N > 1 ? 1024 / N : 1;
    if (bx > cap) bx = cap;
    U3D_SE_DISPATCH(se_bwd_reduce_kernel, T, lpv, dim3((unsigned)bx, (unsigned)N), dim3(256), 0, (hipStream_t)stream, dout, y, gc, a,
                    ws, (long long)V, C, mode, dls, acc_gc, acc_ws);
    U3D_LAUNCH_CHECK();
    return 0;
}

extern "C" int u3d_se_bwd_reduce(int device, u3d_stream_t stream, const float* dout, const float* y, const float* gc,
                                 const float* a, const float* ws, int N, int64_t V, int C, int mode, float* dls, double* acc_gc,
                                 double* acc_ws) {
    return se_bwd_reduce_impl<float>(device, stream, dout, y, gc, a, ws, N, V, C, mode, dls, acc_gc, acc_ws);
}

extern "C" int u3d_se_bwd_reduce_b16(int device, u3d_stream_t stream, const void* dout, const void* y, const float* gc,
                                     const float* a, const float* ws, int N, int64_t V, int C, int mode, float* dls, double* acc_gc,
                                     double* acc_ws) {
    return se_bwd_reduce_impl<__bf16>(device, stream, (const __bf16*)dout, (const __bf16*)y, gc, a, ws, N, V, C, mode, dls, acc_gc, acc_ws);
}

extern "C" int u3d_se_gate_bwd(int device, u3d_stream_t stream, const double* acc_gc, const float* gc, const float* h,
                               const float* s, const float* w1, const float* w2, int N, int C, int Cr, double count, float* dz2,
                               float* dz1, float* ds, float* dw1, float* db1, float* dw2, float* db2) {
    U3D_ENTER(device);
    U3D_REQUIRE(acc_gc && gc && h && s && w1 && w2 && dz2 && dz1 && ds && dw1 && db1 && dw2 && db2 && N > 0 && C > 0 && C <= 1024 &&
                    Cr > 0 && Cr <= 1024 && count > 0,
                "u3d_se_gate_bwd: bad argument");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(se_gate_bwd_kernel, dim3(N), dim3(256), 0, st, acc_gc, gc, h, w1, w2, C, Cr, count, dz2, dz1, ds);
    U3D_LAUNCH_CHECK();
    const long long total = 2ll * C * Cr + C + Cr;
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(se_gate_wgrad_kernel, dim3((unsigned)blocks), dim3(256), 0, st, dz2, dz1, h, s, N, C, Cr, dw1, db1, dw2, db2);
    U3D_LAUNCH_CHECK();
    return 0;
}

template <typename T>
static int se_bwd_apply_impl(int device, u3d_stream_t stream, const T* dout, const T* y, const float* gc, const float* a,
                             const float* ws, const float* dls, const float* ds, int N, int64_t V, int C, int mode, int relu_mask, T* out) {
    U3D_ENTER(device);
    if (int e = se_check(N, V, C, mode, "u3d_se_bwd_apply")) return e;
    U3D_REQUIRE(dout && y && out && (mode == 2 || (gc && ds)) && (mode == 1 || (a && ws && dls)), "u3d_se_bwd_apply: missing argument");
    const long long total = (long long)N * V * (C / 4);
    long long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(se_bwd_apply_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dout, y, gc, a, ws, dls, ds, N,
                       (long long)V, C, mode, relu_mask, out);
    U3D_LAUNCH_CHECK();
    return 0;
}

extern "C" int u3d_se_bwd_apply(int device, u3d_stream_t stream, const float* dout, const float* y, const float* gc,
                                const float* a, const float* ws, const float* dls, const float* ds, int N, int64_t V, int C,
                                int mode, int relu_mask, float* out) {
    return se_bwd_apply_impl<float>(device, stream, dout, y, gc, a, ws, dls, ds, N, V, C, mode, relu_mask, out);
}

extern "C" int u3d_se_bwd_apply_b16(int device, u3d_stream_t stream, const void* dout, const void* y, const float* gc,
                                    const float* a, const float* ws, const float* dls, const float* ds, int N, int64_t V, int C,
                                    int mode, int relu_mask, void* out) {
    return se_bwd_apply_impl<__bf16>(device, stream, (const __bf16*)dout, (const __bf16*)y, gc, a, ws, dls, ds, N, V, C, mode, relu_mask,
                                     (__bf16*)out);
}
