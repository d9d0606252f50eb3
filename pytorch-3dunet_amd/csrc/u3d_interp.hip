// u3d_interp.hip — the InterpolateUpsampling modes other than 'nearest' that run on 5-D tensors.
//
// Reference: pytorch3dunet/unet3d/buildingblocks.py:598-614 InterpolateUpsampling -> F.interpolate(x, size=skip.shape[2:],
// mode=...), chosen by Decoder.__init__ (:435-455) from the model's `upsample` key.  For 3-D models the modes that run at all are
// 'nearest' (virtual, csrc/u3d_conv.hip), 'trilinear' and 'area' ('linear' / 'bilinear' / 'bicubic' raise inside
// F.interpolate for 5-D inputs, upsample 'none' fails at the concat: tests/test_oracle.py pins that on the live reference).
//
// Both remaining modes are separable linear maps with at most TWO source samples per output index and dimension:
//   trilinear (align_corners=False): src = max(scale*(o + 0.5) - 0.5, 0), i0 = floor(src), i1 = min(i0 + 1, in - 1),
//                                    weights (1 - frac, frac)                      (ATen UpSample.h area_pixel_compute_source_index)
//   area = adaptive_avg_pool3d:      window [floor(o*in/out), ceil((o+1)*in/out)), 1 or 2 samples when out >= in, equal weights
// The host builds, per dimension, the tables idx[2*out] = (i0, i1), wt[2*out] = (w0, w1) with ATen's float32 formulas, and for
// the adjoint rng[2*in] = [lo, hi): the outputs that touch input i.  Forward: 8 gathers per output element.  Backward: every
// input element GATHERS its contributions in a fixed order (no atomics: run-to-run identical), the per-dimension weight of
// output o on input i being (i0[o] == i) * w0[o] + (i1[o] == i) * w1[o]  (both terms where the index was clamped at the edge).
// Bandwidth kernels, NDHWC fp32, not on the measured path.
#include "u3d_common.h"

namespace {

struct resample_params {
    const float* x;  // forward: low-res input; backward: gradient of the full-res output
    float* y;
    const int32_t *iz, *iy, *ix;  // forward tables (2 per output index) / backward ranges (2 per input index)
    const float *wz, *wy, *wx;
    const int32_t *jz, *jy, *jx;  // backward only: the forward index tables
    int N, D1, H1, W1, D, H, W, C;
};

__global__ __launch_bounds__(256) void resample_fwd_kernel(const resample_params p) {
    const long long total = (long long)p.N * p.D * p.H * p.W * p.C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % p.C);
        long long v = i / p.C;
        const int x = (int)(v % p.W);
        v /= p.W;
        const int y = (int)(v % p.H);
        v /= p.H;
        const int z = (int)(v % p.D);
        const int n = (int)(v / p.D);
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int zz = p.iz[2 * z + a];
            const float wa = p.wz[2 * z + a];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int yy = p.iy[2 * y + b];
                const float wb = wa * p.wy[2 * y + b];
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const int xx = p.ix[2 * x + d];
                    const float w = wb * p.wx[2 * x + d];
                    acc = fmaf(w, p.x[((((size_t)n * p.D1 + zz) * p.H1 + yy) * p.W1 + xx) * p.C + c], acc);
                }
            }
        }
        p.y[i] = acc;
    }
}

__device__ __forceinline__ float adj_weight(const int32_t* idx, const float* wt, int o, int i) {
    return (idx[2 * o] == i ? wt[2 * o] : 0.f) + (idx[2 * o + 1] == i ? wt[2 * o + 1] : 0.f);
}

__global__ __launch_bounds__(256) void resample_bwd_kernel(const resample_params p) {
    const long long total = (long long)p.N * p.D1 * p.H1 * p.W1 * p.C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % p.C);
        long long v = i / p.C;
        const int x = (int)(v % p.W1);
        v /= p.W1;
        const int y = (int)(v % p.H1);
        v /= p.H1;
        const int z = (int)(v % p.D1);
        const int n = (int)(v / p.D1);
        float acc = 0.f;
        for (int oz = p.iz[2 * z]; oz < p.iz[2 * z + 1]; ++oz) {
            const float wa = adj_weight(p.jz, p.wz, oz, z);
            for (int oy = p.iy[2 * y]; oy < p.iy[2 * y + 1]; ++oy) {
                const float wb = wa * adj_weight(p.jy, p.wy, oy, y);
                for (int ox = p.ix[2 * x]; ox < p.ix[2 * x + 1]; ++ox) {
                    const float w = wb * adj_weight(p.jx, p.wx, ox, x);
                    acc = fmaf(w, p.x[((((size_t)n * p.D + oz) * p.H + oy) * p.W + ox) * p.C + c], acc);
                }
            }
        }
        p.y[i] = acc;
    }
}

inline int grid_for(long long n) {
    long long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 32768 ? 32768 : b));
}

}  // namespace

extern "C" int u3d_resample2_fwd(int device, u3d_stream_t stream, const float* x, const int32_t* iz, const int32_t* iy,
                                 const int32_t* ix, const float* wz, const float* wy, const float* wx, int N, int D1, int H1, int W1,
                                 int D, int H, int W, int C, float* out) {
    U3D_ENTER(device);
    U3D_REQUIRE(x && iz && iy && ix && wz && wy && wx && out && N > 0 && D1 > 0 && H1 > 0 && W1 > 0 && D > 0 && H > 0 && W > 0 && C > 0,
                "u3d_resample2_fwd: bad argument");
    resample_params p{x, out, iz, iy, ix, wz, wy, wx, nullptr, nullptr, nullptr, N, D1, H1, W1, D, H, W, C};
    hipLaunchKernelGGL(resample_fwd_kernel, dim3(grid_for((long long)N * D * H * W * C)), dim3(256), 0, (hipStream_t)stream, p);
    U3D_LAUNCH_CHECK();
    return 0;
}

extern "C" int u3d_resample2_bwd(int device, u3d_stream_t stream, const float* dout, const int32_t* rz, const int32_t* ry,
                                 const int32_t* rx, const int32_t* iz, const int32_t* iy, const int32_t* ix, const float* wz,
                                 const float* wy, const float* wx, int N, int D1, int H1, int W1, int D, int H, int W, int C,
                                 float* dx) {
    U3D_ENTER(device);
    U3D_REQUIRE(dout && rz && ry && rx && iz && iy && ix && wz && wy && wx && dx && N > 0 && D1 > 0 && H1 > 0 && W1 > 0 && D > 0 &&
                    H > 0 && W > 0 && C > 0,
                "u3d_resample2_bwd: bad argument");
    resample_params p{dout, dx, rz, ry, rx, wz, wy, wx, iz, iy, ix, N, D1, H1, W1, D, H, W, C};
    hipLaunchKernelGGL(resample_bwd_kernel, dim3(grid_for((long long)N * D1 * H1 * W1 * C)), dim3(256), 0, (hipStream_t)stream, p);
    U3D_LAUNCH_CHECK();
    return 0;
}
