// u3d_ops.hip — the bandwidth-bound kernels around the convolutions: GroupNorm statistics / finalize /
// backward, MaxPool3d(2) forward + fused backward merge, the 1x1x1 head with Sigmoid/Softmax, layout
// transposes.  Reference call sites: buildingblocks.py:47 (ReLU), :62-75 (GroupNorm), :356 (MaxPool3d),
// :491 (cat), :614 (nearest interpolate), model.py:88-101,141-147 (final conv + activation).
#include "u3d_common.h"
#include "u3d_gn.h"

#include <string.h>

extern int g_u3d_tune[24];  // u3d_set_tuning (csrc/u3d_conv.hip); key 18 = 1: one-channel input statistics on the general kernel (A/B)

// ---- library plumbing ----------------------------------------------------------------------------
static thread_local char g_err[512] = "";

int u3d_set_err(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int u3d_device_guard::enter(int device) {
    hipError_t e = hipGetDevice(&prev);
    if (e != hipSuccess) return u3d_set_err(U3D_EHIP, "hipGetDevice failed: %s", hipGetErrorString(e));
    if (device >= 0 && prev != device) {
        e = hipSetDevice(device);
        if (e != hipSuccess) return u3d_set_err(U3D_EHIP, "hipSetDevice(%d) failed: %s", device, hipGetErrorString(e));
        switched = true;
    }
    return 0;
}

extern "C" int u3d_version(void) { return U3D_VERSION; }
extern "C" const char* u3d_last_error(void) { return g_err; }

extern "C" int u3d_check_device(int device) {
    hipDeviceProp_t prop;
    U3D_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return u3d_set_err(U3D_EARCH, "device %d is %s, this library is built for gfx950 only", device, prop.gcnArchName);
    return 0;
}

// ---- developer aid: a stand-in for a link-bound collective (tools/overlap_probe.py) ----------------------------------------
// RCCL's ring all-reduce runs a handful of workgroups (one per channel) that move data at the rate of the xGMI links — an
// order of magnitude below HBM — for as long as the links need.  A 1-rank group launches nothing, so the single-GPU overlap probe
// needs a kernel of that shape: `blocks` workgroups of 256 threads stream `n` floats in place (x <- x * 1) in 64 KiB pieces and
// PACE themselves against the constant-rate wall clock so that the launch lasts `min_seconds` however fast memory is (0 = no
// pacing, `passes` repetitions).  Like the collective it occupies few CUs, needs little bandwidth and does not finish early when
// it gets more of either.  Values never change.
__global__ __launch_bounds__(256) void debug_stream_pass_kernel(float* __restrict__ buf, long long n4, int passes, float one,
                                                               long long ticks_total) {
    f32x4* b = reinterpret_cast<f32x4*>(buf);
    constexpr long long PIECE = 4096;  // float4 elements per piece = 64 KiB
    const long long npieces = (n4 + PIECE - 1) / PIECE;
    const long long mine = (npieces - blockIdx.x + gridDim.x - 1) / gridDim.x;  // pieces blockIdx.x, + gridDim.x, ...
    const long long t0 = wall_clock64();
    long long done = 0;
    for (int p = 0; p < passes; ++p)
        for (long long pc = blockIdx.x; pc < npieces; pc += gridDim.x) {
            const long long lo = pc * PIECE, hi = lo + PIECE < n4 ? lo + PIECE : n4;
            for (long long i = lo + threadIdx.x; i < hi; i += 256) {
                f32x4 v = b[i];
                v *= one;  // (a run-time 1.0: a literal would let the compiler drop the load / store pair)
                b[i] = v;
            }
            ++done;
            if (ticks_total > 0) {
                const long long due = t0 + ticks_total * done / (mine * passes);
                while (wall_clock64() < due) __builtin_amdgcn_s_sleep(64);
            }
        }
}

extern "C" int u3d_debug_stream_pass(int device, u3d_stream_t stream, float* buf, long long n, int blocks, int passes,
                                     double min_seconds) {
    U3D_ENTER(device);
    U3D_REQUIRE(buf && n >= 4 && blocks > 0 && passes > 0 && min_seconds >= 0.0 && ((uintptr_t)buf & 15) == 0,
                "u3d_debug_stream_pass: bad argument");
    int khz = 0;
    U3D_HIP(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device));
    const long long ticks = (long long)(min_seconds * 1e3 * (double)khz);
    hipLaunchKernelGGL(debug_stream_pass_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, buf, n / 4, passes, 1.0f,
                       ticks);
    U3D_LAUNCH_CHECK();
    return 0;
}

// ---- the matrix pipe's sustained fp32 rate as a function of the OPERAND DATA (round 6; tools/mfma_peak_data.hip is the stand-alone
// twin).  A launch of 2 blocks per CU that does nothing but v_mfma_f32_32x32x2_f32 on register operands — no memory traffic, no other
// instruction: mode 0 all-zero operands, 1 one constant pair, 2 N(0,1)-like values that differ per lane and step.  The chip clocks the pipe
// by the power its operands' toggling draws: mode 2 (what a convolution of real activations looks like) is the ceiling a roofline
// fraction against the 157.3 TFLOP/s datasheet peak can reach.  bench.py times it beside the step (roofline.mfma_ceiling_*).
__device__ __forceinline__ float dbg_hash_unit(unsigned x) {
    float s = 0.f;
    for (int i = 0; i < 4; ++i) {
        x ^= x >> 16, x *= 0x7feb352dU, x ^= x >> 15, x *= 0x846ca68bU, x ^= x >> 16;
        s += (float)(x & 0xffffff) * (1.0f / 16777216.0f);
    }
    return (s - 2.0f) * 1.7320508f;
}
__global__ __launch_bounds__(256) void debug_mfma_rate_kernel(float* __restrict__ sink, int iters, int mode) {
    f32x16 acc[2];
    for (int k = 0; k < 2; ++k)
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    float a[16], b[16];
    const unsigned id = blockIdx.x * 256 + threadIdx.x;
    for (int i = 0; i < 16; ++i) {
        a[i] = mode == 0 ? 0.f : (mode == 1 ? 0.37f : dbg_hash_unit(id * 32 + i));
        b[i] = mode == 0 ? 0.f : (mode == 1 ? -1.21f : dbg_hash_unit(id * 32 + 16 + i));
    }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int k = 0; k < 2; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[(u + 5 * k) & 15], acc[k], 0, 0, 0);
    }
    float s = 0.f;
    for (int k = 0; k < 2; ++k)
        for (int r = 0; r < 16; ++r) s += acc[k][r];
    if (s == 123.456f) sink[0] = s;
}

// returns (through *flop_out, host memory) the FLOPs the launch executes: blocks x 4 waves x iters x 32 MFMAs x 4096
extern "C" int u3d_debug_mfma_f32_rate(int device, u3d_stream_t stream, float* sink, int iters, int mode, double* flop_out) {
    U3D_ENTER(device);
    U3D_REQUIRE(sink && iters > 0 && mode >= 0 && mode <= 2, "u3d_debug_mfma_f32_rate: bad argument");
    int ncu = 0;
    U3D_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device));
    const int blocks = 2 * ncu;
    hipLaunchKernelGGL(debug_mfma_rate_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, sink, iters, mode);
    U3D_LAUNCH_CHECK();
    if (flop_out) *flop_out = (double)blocks * 4.0 * iters * 32.0 * 4096.0;
    return 0;
}

static inline long long cdivll(long long a, long long b) { return (a + b - 1) / b; }
__device__ __forceinline__ long long cdivll_dev(long long a, long long b) { return (a + b - 1) / b; }
static inline int grid_for(long long total, int cap = 8192) {
    long long b = cdivll(total, 256);
    if (b < 1) b = 1;
    return (int)(b > cap ? cap : b);
}

static bool src_vec_ok(const u3d_src_t* s) {
    if (s->C0 % 4 != 0 || s->C1 % 4 != 0) return false;
    if (((uintptr_t)s->p0 & 15) != 0) return false;
    if (s->C1 > 0 && ((uintptr_t)s->p1 & 15) != 0) return false;
    return true;
}

// =================================================================================================
// per-(n,channel) sum / sum of squares of a (virtual) tensor.  grid (blocks_per_n, N), 256 threads:
// thread -> (row = t / Q, quad = t % Q), Q = ceil(C/4); rows stride the block's voxel range.
__global__ __launch_bounds__(256) void chan_stats_kernel(const u3d_src_t src, int D, int H, int W, int Q, int rows,
                                                         int vec, double* __restrict__ stats, int reps) {
    extern __shared__ float red[];  // [rows][Q][8]
    const int t = threadIdx.x;
    const int n = blockIdx.y;
    const int Ctot = src.C0 + src.C1;
    const long long V = (long long)D * H * W;
    const long long per = cdivll_dev(V, gridDim.x);
    const long long vbeg = (long long)blockIdx.x * per;
    const long long vend = min(V, vbeg + per);
    const int row = t / Q, qd = t - row * Q;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    if (row < rows) {
        long long v = vbeg + row;
        if (src.C1 == 0) {
            // plain tensor: 8 independent 16-byte loads in flight per thread (round 6; one at a time, the pass was latency-bound:
            // 23 us for the 33 MB pooled tensor of the bench workload)
            for (; v + 7LL * rows < vend; v += 8LL * rows) {
                f32x4 q[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) q[k] = u3d_load_quad(src, (int)((long long)n * V + v + (long long)k * rows), 0, 4 * qd, vec != 0);
#pragma unroll
                for (int k = 0; k < 8; ++k)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        s1[e] += q[k][e];
                        s2[e] += q[k][e] * q[k][e];
                    }
            }
        }
        for (; v < vend; v += rows) {
            int v0 = (int)((long long)n * V + v), v1 = 0;
            if (src.C1 > 0) {
                const int x = (int)(v % W);
                const long long r = v / W;
                const int y = (int)(r % H);
                const int z = (int)(r / H);
                u3d_vox_index(src, n, z, y, x, D, H, W, v0, v1);
            }
            const f32x4 q = u3d_load_quad(src, v0, v1, 4 * qd, vec != 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s1[e] += q[e];
                s2[e] += q[e] * q[e];
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[(row * Q + qd) * 8 + e] = s1[e];
            red[(row * Q + qd) * 8 + 4 + e] = s2[e];
        }
    }
    __syncthreads();
    // one channel per thread (strided when C > 256), sum over rows
    for (int c = t; c < Ctot; c += 256) {
        const int qd2 = c >> 2, e = c & 3;
        float a = 0.f, b = 0.f;
        for (int r = 0; r < rows; ++r) {
            a += red[(r * Q + qd2) * 8 + e];
            b += red[(r * Q + qd2) * 8 + 4 + e];
        }
        // (replica row blockIdx.x % reps, u3d_chan_stats_reps: ~1000 blocks per sample on the same 2 C addresses were most of this pass)
        double* dst = stats + ((size_t)(blockIdx.x % (unsigned)reps) * gridDim.y * Ctot + (size_t)n * Ctot + c) * 2;
        u3d_atomic_add_f64(dst, (double)a);
        u3d_atomic_add_f64(dst + 1, (double)b);
    }
}

// ONE channel (the network's input patch, model.py:123): 16-byte loads, wave butterflies, one f64 atomic pair per block.  (The general
// kernel above reads 4 bytes per lane and folds its 256 rows with ONE thread: 23 us for the 8 MB input of the bench workload.)
__global__ __launch_bounds__(256) void scalar_stats_kernel(const float* __restrict__ x, long long V, double* __restrict__ stats) {
    __shared__ float red[2][4];
    const int n = blockIdx.y, t = threadIdx.x;
    const float* xn = x + (size_t)n * V;
    const long long V4 = V >> 2;
    float s1 = 0.f, s2 = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + t; i < V4; i += (long long)gridDim.x * 256) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(xn + 4 * i);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            s1 += q[e];
            s2 += q[e] * q[e];
        }
    }
    if (blockIdx.x == 0 && t < (int)(V & 3)) {  // tail voxels
        const float q = xn[4 * V4 + t];
        s1 += q;
        s2 += q * q;
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        s1 += __shfl_xor(s1, m);
        s2 += __shfl_xor(s2, m);
    }
    if ((t & 63) == 0) {
        red[0][t >> 6] = s1;
        red[1][t >> 6] = s2;
    }
    __syncthreads();
    if (t == 0) {
        u3d_atomic_add_f64(&stats[(size_t)n * 2], (double)(((red[0][0] + red[0][1]) + red[0][2]) + red[0][3]));
        u3d_atomic_add_f64(&stats[(size_t)n * 2 + 1], (double)(((red[1][0] + red[1][1]) + red[1][2]) + red[1][3]));
    }
}

static int chan_stats_impl(int device, u3d_stream_t stream, const u3d_src_t* src, int N, int D, int H, int W, double* stats, int reps) {
    U3D_ENTER(device);
    U3D_REQUIRE(src && src->p0 && stats && N > 0 && D > 0 && H > 0 && W > 0, "u3d_chan_stats: bad argument");
    const int Ctot = src->C0 + src->C1;
    if (Ctot == 1 && src->C1 == 0 && N <= 65535 && ((long long)D * H * W) % 4 == 0 && ((uintptr_t)src->p0 & 15) == 0 &&
        g_u3d_tune[18] != 1) {
        const long long V = (long long)D * H * W;
        long long bpn = (V / 4 + 256 * 8 - 1) / (256 * 8);  // >= 8 loads per thread
        const long long want = (512 + N - 1) / N;
        if (bpn > want) bpn = want;
        if (bpn < 1) bpn = 1;
        hipLaunchKernelGGL(scalar_stats_kernel, dim3((unsigned)bpn, (unsigned)N), dim3(256), 0, (hipStream_t)stream, src->p0, V, stats);
        U3D_LAUNCH_CHECK();
        return 0;
    }
    const int Q = (Ctot + 3) / 4;
    U3D_REQUIRE(Q <= 256, "u3d_chan_stats: at most 1024 channels supported");
    const int rows = 256 / Q;
    const long long V = (long long)D * H * W;
    // >= ~2048 blocks in total when the tensor is large enough (>= 16 voxels per thread-row), <= 4096 per sample
    long long bpn = cdivll(V, (long long)rows * 16);
    const long long want = cdivll(2048, N);
    if (bpn > want) bpn = want;
    if (bpn < 1) bpn = 1;
    if (bpn > 4096) bpn = 4096;
    const size_t shmem = (size_t)rows * Q * 8 * sizeof(float);
    hipLaunchKernelGGL(chan_stats_kernel, dim3((unsigned)bpn, (unsigned)N), dim3(256), shmem, (hipStream_t)stream,
                       *src, D, H, W, Q, rows, src_vec_ok(src) ? 1 : 0, stats, reps);
    U3D_LAUNCH_CHECK();
    return 0;
}

extern "C" int u3d_chan_stats(int device, u3d_stream_t stream, const u3d_src_t* src, int N, int D, int H, int W, double* stats) {
    return chan_stats_impl(device, stream, src, N, D, H, W, stats, 1);
}

// ... into a table of `reps` replica rows [reps][N][C][2] (zeroed by the caller; see u3d_conv3d_ex_reps): block b adds to row b % reps
extern "C" int u3d_chan_stats_reps(int device, u3d_stream_t stream, const u3d_src_t* src, int N, int D, int H, int W, double* stats,
                                   int reps) {
    U3D_REQUIRE(reps >= 1 && reps <= 64, "u3d_chan_stats_reps: reps must be 1 .. 64");
    return chan_stats_impl(device, stream, src, N, D, H, W, stats, reps);
}

// Per-(n, channel) sums of the NEAREST-UPSAMPLED image of a low-res tensor without touching the upsampled grid: along an axis that is
// upsampled n -> 2n + 1 (the pooled size of an odd level, buildingblocks.py:614 to the skip's size) low-res cell 0 has three children,
// every other cell two; along an exact-2x axis every cell has two.  stats[N][C][2] += (sum_v w(v) x, sum_v w(v) x^2), w = the
// number of children of low-res voxel v.  (Round 6: such a level's GroupNorm statistics were taken by u3d_chan_stats over the
// VIRTUAL concat — every full-resolution voxel through the index maps: 164 us at the 40 x 85 x 85 level of the shipped 80 x 170 x 170
// patch, against the 18 MB low-res tensor read once here.)
__global__ __launch_bounds__(256) void chan_stats_children_kernel(const float* __restrict__ x, int D1, int H1, int W1, int C, int Q, int rows,
                                                                  int ez, int ey, int ex, double* __restrict__ stats) {
    extern __shared__ float red[];  // [rows][Q][8]
    const int t = threadIdx.x, n = blockIdx.y;
    const long long V = (long long)D1 * H1 * W1;
    const long long per = cdivll_dev(V, gridDim.x);
    const long long vbeg = (long long)blockIdx.x * per, vend = min(V, vbeg + per);
    const int row = t / Q, qd = t - row * Q;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    if (row < rows) {
        for (long long v = vbeg + row; v < vend; v += rows) {
            const int xi = (int)(v % W1);
            const long long r = v / W1;
            const int yi = (int)(r % H1), zi = (int)(r / H1);
            const float wgt = (float)((2 + (ez && zi == 0)) * (2 + (ey && yi == 0)) * (2 + (ex && xi == 0)));
            const f32x4 q = *reinterpret_cast<const f32x4*>(x + ((size_t)n * V + v) * C + 4 * qd);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s1[e] += wgt * q[e];
                s2[e] += wgt * (q[e] * q[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[(row * Q + qd) * 8 + e] = s1[e];
            red[(row * Q + qd) * 8 + 4 + e] = s2[e];
        }
    }
    __syncthreads();
    for (int c = t; c < C; c += 256) {
        const int qd2 = c >> 2, e = c & 3;
        float a = 0.f, b = 0.f;
        for (int r = 0; r < rows; ++r) {
            a += red[(r * Q + qd2) * 8 + e];
            b += red[(r * Q + qd2) * 8 + 4 + e];
        }
        u3d_atomic_add_f64(&stats[((size_t)n * C + c) * 2 + 0], (double)a);
        u3d_atomic_add_f64(&stats[((size_t)n * C + c) * 2 + 1], (double)b);
    }
}

extern "C" int u3d_chan_stats_children(int device, u3d_stream_t stream, const float* x_low, int N, int D1, int H1, int W1, int C, int ez,
                                       int ey, int ex, double* stats) {
    U3D_ENTER(device);
    U3D_REQUIRE(x_low && stats && N > 0 && N <= 65535 && D1 > 0 && H1 > 0 && W1 > 0 && C > 0 && C % 4 == 0 && C <= 1024 &&
                    ((uintptr_t)x_low & 15) == 0, "u3d_chan_stats_children: bad argument (C a multiple of 4, <= 1024, 16-byte aligned)");
    const int Q = C / 4, rows = 256 / Q;
    const long long V = (long long)D1 * H1 * W1;
    long long bpn = cdivll(V, (long long)rows * 16);
    const long long want = cdivll(1024, N);
    if (bpn > want) bpn = want;
    if (bpn < 1) bpn = 1;
    hipLaunchKernelGGL(chan_stats_children_kernel, dim3((unsigned)bpn, (unsigned)N), dim3(256), (size_t)rows * Q * 8 * sizeof(float),
                       (hipStream_t)stream, x_low, D1, H1, W1, C, Q, rows, ez, ey, ex, stats);
    U3D_LAUNCH_CHECK();
    return 0;
}

// =================================================================================================
// GroupNorm finalize: (n,group) mean / rstd from channel sums, then the per-channel affine table.
// One block per sample: threads = channels (coalesced loads of the per-channel sums into LDS), then one thread per group
// reduces its channels from LDS, then threads = channels again for the affine table.  (The first version ran one thread per
// (n, group) with a serial loop of dependent global loads: ~9 us per launch, 28 launches per step.)
__global__ __launch_bounds__(256) void gn_finalize_kernel(const double* __restrict__ st0, int C0, double sc0,
                                                          const double* __restrict__ st1, int C1, double sc1, int N, int G,
                                                          double count, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps,
                                                          float* __restrict__ affine, float* __restrict__ mean_rstd,
                                                          float* __restrict__ affine_lo, float* __restrict__ affine_hi,
                                                          int Cs, int reps0, int reps1) {
    // reps0 / reps1 > 1 (u3d_gn_finalize_reps): st0 / st1 hold that many replica rows [reps][N][C][2], summed here in ascending order
    extern __shared__ double sh[];  // [C][2] channel sums, then [G][2] mean / rstd
    const int C = C0 + C1, cpg = C / G, n = blockIdx.x, t = threadIdx.x;
    double* gmr = sh + 2 * (size_t)C;
    for (int c = t; c < C; c += blockDim.x) {
        double s, ss;
        if (c < C0) {
            const double* q = st0 + ((size_t)n * C0 + c) * 2;
            s = sc0 * u3d_sum_replicas(q, (size_t)gridDim.x * C0 * 2, reps0);
            ss = sc0 * u3d_sum_replicas(q + 1, (size_t)gridDim.x * C0 * 2, reps0);
        } else {
            const double* q = st1 + ((size_t)n * C1 + (c - C0)) * 2;
            s = sc1 * u3d_sum_replicas(q, (size_t)gridDim.x * C1 * 2, reps1);
            ss = sc1 * u3d_sum_replicas(q + 1, (size_t)gridDim.x * C1 * 2, reps1);
        }
        sh[2 * c] = s;
        sh[2 * c + 1] = ss;
    }
    __syncthreads();
    for (int g = t; g < G; g += blockDim.x) {
        double s = 0.0, ss = 0.0;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) {  // fixed order
            s += sh[2 * c];
            ss += sh[2 * c + 1];
        }
        const double m = count * cpg;
        const double mean = s / m;
        double var = ss / m - mean * mean;
        if (var < 0.0) var = 0.0;
        const double rstd = 1.0 / sqrt(var + (double)eps);
        gmr[2 * g] = mean;
        gmr[2 * g + 1] = rstd;
        mean_rstd[((size_t)n * G + g) * 2] = (float)mean;
        mean_rstd[((size_t)n * G + g) * 2 + 1] = (float)rstd;
    }
    __syncthreads();
    for (int c = t; c < C; c += blockDim.x) {
        const int g = c / cpg;
        const double a = gmr[2 * g + 1] * (double)gamma[c];
        const float fa = (float)a, fb = (float)((double)beta[c] - gmr[2 * g] * a);
        affine[((size_t)n * C + c) * 2] = fa;
        affine[((size_t)n * C + c) * 2 + 1] = fb;
        // (u3d_gn_finalize_split) compact copies of the rows of [0, Cs) / [Cs, C): what a kernel that reads ONE half of a virtual
        // concat as a plain tensor takes as its table
        if (affine_lo && c < Cs) {
            affine_lo[((size_t)n * Cs + c) * 2] = fa;
            affine_lo[((size_t)n * Cs + c) * 2 + 1] = fb;
        }
        if (affine_hi && c >= Cs) {
            affine_hi[((size_t)n * (C - Cs) + (c - Cs)) * 2] = fa;
            affine_hi[((size_t)n * (C - Cs) + (c - Cs)) * 2 + 1] = fb;
        }
    }
}

static int gn_finalize_impl(int device, u3d_stream_t stream, const double* stats0, int C0, double scale0, const double* stats1, int C1,
                            double scale1, int N, int G, double count, const float* gamma, const float* beta, float eps, float* affine,
                            float* mean_rstd, float* affine_lo, float* affine_hi, int Csplit, int reps0 = 1, int reps1 = 1) {
    U3D_ENTER(device);
    const int C = C0 + C1;
    U3D_REQUIRE(reps0 >= 1 && reps0 <= 64 && reps1 >= 1 && reps1 <= 64, "u3d_gn_finalize_reps: replica counts must be 1 .. 64");
    U3D_REQUIRE(stats0 && C0 > 0 && C1 >= 0 && (C1 == 0 || stats1) && gamma && beta && affine && mean_rstd && N > 0 && G > 0 &&
                    count > 0.0 && Csplit >= 0 && Csplit <= C, "u3d_gn_finalize: bad argument");
    U3D_REQUIRE(C % G == 0, "u3d_gn_finalize: channels %d not divisible by groups %d", C, G);
    U3D_REQUIRE(C <= 4096, "u3d_gn_finalize: more than 4096 channels");
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(N), dim3(256), sizeof(double) * 2 * ((size_t)C + G), (hipStream_t)stream,
                       stats0, C0, scale0, stats1, C1, scale1, N, G, count, gamma, beta, eps, affine, mean_rstd, affine_lo, affine_hi, Csplit,
                       reps0, reps1);
    U3D_LAUNCH_CHECK();
    return 0;
}

extern "C" int u3d_gn_finalize(int device, u3d_stream_t stream, const double* stats0, int C0, double scale0,
                               const double* stats1, int C1, double scale1, int N, int G, double count,
                               const float* gamma, const float* beta, float eps, float* affine, float* mean_rstd) {
    return gn_finalize_impl(device, stream, stats0, C0, scale0, stats1, C1, scale1, N, G, count, gamma, beta, eps, affine, mean_rstd,
                            nullptr, nullptr, 0);
}

extern "C" int u3d_gn_finalize_split(int device, u3d_stream_t stream, const double* stats0, int C0, double scale0,
                                     const double* stats1, int C1, double scale1, int N, int G, double count,
                                     const float* gamma, const float* beta, float eps, float* affine, float* mean_rstd,
                                     int Csplit, float* affine_lo, float* affine_hi) {
    return gn_finalize_impl(device, stream, stats0, C0, scale0, stats1, C1, scale1, N, G, count, gamma, beta, eps, affine, mean_rstd,
                            affine_lo, affine_hi, Csplit);
}

extern "C" int u3d_gn_finalize_reps(int device, u3d_stream_t stream, const double* stats0, int C0, double scale0, int reps0,
                                    const double* stats1, int C1, double scale1, int reps1, int N, int G, double count,
                                    const float* gamma, const float* beta, float eps, float* affine, float* mean_rstd,
                                    int Csplit, float* affine_lo, float* affine_hi) {
    return gn_finalize_impl(device, stream, stats0, C0, scale0, stats1, C1, scale1, N, G, count, gamma, beta, eps, affine, mean_rstd,
                            affine_lo, affine_hi, Csplit, reps0, reps1);
}

// GroupNorm backward reductions -> dgamma, dbeta, coefficient table coef[N][3][C] (p,q,r).
// One block.  `staged`: the N*C*2 sums are first copied to LDS with coalesced loads.  `par` (the sums AND two product arrays fit
// LDS: N*C <= 2048, every shipped configuration): the per-channel products of the group sums are formed by all threads, the one
// thread of a (sample, group) pair only adds them up in channel order, and the coefficient table is written by all threads — with
// eight pairs (N = 1, 8 groups) the serial version kept 8 threads busy for 13 us, 18 times per config-4 step.
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(const double* __restrict__ gs, const float* __restrict__ mean_rstd,
                                                              const float* __restrict__ gamma, int N, int C, int G,
                                                              double count, int staged, int par, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, float* __restrict__ coef,
                                                              const double* __restrict__ gs_hi, int C0, float hi_scale,
                                                              float* __restrict__ coef_hi) {
    extern __shared__ double shb[];
    u3d_gn_bwd_finalize_body(gs, mean_rstd, gamma, N, C, G, count, staged, par, dgamma, dbeta, coef, gs_hi, C0, hi_scale, coef_hi, shb);
}

extern "C" int u3d_gn_bwd_finalize(int device, u3d_stream_t stream, const double* gstats, const float* mean_rstd,
                                   const float* gamma, int N, int C, int G, double count, float* dgamma,
                                   float* dbeta, float* coef) {
    U3D_ENTER(device);
    U3D_REQUIRE(gstats && mean_rstd && gamma && dgamma && dbeta && coef && N > 0 && C > 0 && G > 0 && C % G == 0,
                "u3d_gn_bwd_finalize: bad argument");
    const size_t bytes = sizeof(double) * 2 * (size_t)N * C;
    const int staged = bytes <= 48 * 1024 ? 1 : 0;
    const size_t par_bytes = 2 * bytes + sizeof(double) * 2 * (size_t)N * G;
    const int par = staged && par_bytes <= 64 * 1024 ? 1 : 0;
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(1), dim3(256), par ? par_bytes : (staged ? bytes : 0), (hipStream_t)stream, gstats,
                       mean_rstd, gamma, N, C, G, count, staged, par, dgamma, dbeta, coef, (const double*)nullptr, C, 1.0f, (float*)nullptr);
    U3D_LAUNCH_CHECK();
    return 0;
}

extern "C" int u3d_gn_bwd_finalize_split_supported(int N, int C, int G) {
    // the two-table form lives in the LDS-staged path of the kernel (every shipped configuration: N * C <= 2048)
    const size_t bytes = sizeof(double) * 2 * (size_t)N * C;
    return (N > 0 && C > 0 && G > 0 && 2 * bytes + sizeof(double) * 2 * (size_t)N * G <= 64 * 1024) ? 1 : 0;
}

extern "C" int u3d_gn_bwd_finalize_split(int device, u3d_stream_t stream, const double* gstats_lo, int C0, const double* gstats_hi,
                                         int C1, const float* mean_rstd, const float* gamma, int N, int G, double count,
                                         float* dgamma, float* dbeta, float* coef, float hi_scale, float* coef_hi) {
    U3D_ENTER(device);
    const int C = C0 + C1;
    U3D_REQUIRE(gstats_lo && gstats_hi && mean_rstd && gamma && dgamma && dbeta && coef && N > 0 && C0 > 0 && C1 > 0 && G > 0 &&
                    C % G == 0, "u3d_gn_bwd_finalize_split: bad argument");
    U3D_REQUIRE(u3d_gn_bwd_finalize_split_supported(N, C, G) == 1, "u3d_gn_bwd_finalize_split: N * C = %d exceeds the staged path", N * C);
    const size_t bytes = sizeof(double) * 2 * (size_t)N * C;
    const size_t par_bytes = 2 * bytes + sizeof(double) * 2 * (size_t)N * G;
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(1), dim3(256), par_bytes, (hipStream_t)stream, gstats_lo, mean_rstd, gamma, N, C, G,
                       count, 1, 1, dgamma, dbeta, coef, gstats_hi, C0, hi_scale, coef_hi);
    U3D_LAUNCH_CHECK();
    return 0;
}

// out = (p*dg[coff+c] + q*x + r) * mask
template <bool VEC, typename T = float>
__global__ void gn_bwd_apply_kernel(const T* __restrict__ dg, int Cdg, int coff, const T* __restrict__ x,
                                    int Cx, const float* __restrict__ coef, int Ctot, long long Vn, int N,
                                    int relu_mask, const T* __restrict__ add, T* __restrict__ out) {
    if (VEC) {
        const int Q = Cx >> 2;
        const long long total = (long long)N * Vn * Q;
        for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
             idx += (long long)gridDim.x * blockDim.x) {
            const int qd = (int)(idx % Q);
            const long long v = idx / Q;  // n*Vn + voxel
            const int n = (int)(v / Vn);
            const int c = 4 * qd;
            const f32x4 d = u3d_ldq(dg + (size_t)v * Cdg + coff + c);
            const f32x4 xv = u3d_ldq(x + (size_t)v * Cx + c);
            const f32x4 p = *reinterpret_cast<const f32x4*>(coef + ((size_t)n * 3 + 0) * Ctot + coff + c);
            const f32x4 q = *reinterpret_cast<const f32x4*>(coef + ((size_t)n * 3 + 1) * Ctot + coff + c);
            const f32x4 r = *reinterpret_cast<const f32x4*>(coef + ((size_t)n * 3 + 2) * Ctot + coff + c);
            f32x4 o = p * d + q * xv + r;
            if (add) o += u3d_ldq(add + (size_t)v * Cx + c);
            if (relu_mask) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = xv[e] > 0.f ? o[e] : 0.f;
            }
            u3d_stq(out + (size_t)v * Cx + c, o);
        }
    } else {
        const long long total = (long long)N * Vn * Cx;
        for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
             idx += (long long)gridDim.x * blockDim.x) {
            const int c = (int)(idx % Cx);
            const long long v = idx / Cx;
            const int n = (int)(v / Vn);
            const float d = u3d_ld(dg + (size_t)v * Cdg + coff + c);
            const float xv = u3d_ld(x + idx);
            float o = coef[((size_t)n * 3 + 0) * Ctot + coff + c] * d + coef[((size_t)n * 3 + 1) * Ctot + coff + c] * xv +
                      coef[((size_t)n * 3 + 2) * Ctot + coff + c];
            if (add) o += u3d_ld(add + idx);
            if (relu_mask && !(xv > 0.f)) o = 0.f;
            u3d_st(out + idx, o);
        }
    }
}

// bf16 activation storage, channel OCTETS (round 5): the quad kernel above moves 8 bytes per lane and tensor with two 64-bit divisions
// per element — 4.5 TB/s on config 4's 18 launches per step (0.96 ms).  Same arithmetic per element (p * dg + q * x + r [+ add], mask),
// 16 bytes per lane: needs Cdg, Cx, coff, Ctot % 8 == 0 and 16-byte aligned tensors.
__global__ __launch_bounds__(256) void gn_bwd_apply_oct_b16_kernel(const __bf16* __restrict__ dg, int Cdg, int coff,
                                                                   const __bf16* __restrict__ x, int Cx, const float* __restrict__ coef,
                                                                   int Ctot, long long Vn, int N, int relu_mask,
                                                                   const __bf16* __restrict__ add, __bf16* __restrict__ out) {
    typedef __bf16 b16x8 __attribute__((ext_vector_type(8)));
    const int Q = Cx >> 3;
    const long long total = (long long)N * Vn * Q;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int qd = (int)(idx % Q);
        const long long v = idx / Q;  // n*Vn + voxel
        const int n = (int)(v / Vn);
        const int c = 8 * qd;
        const b16x8 d8 = *reinterpret_cast<const b16x8*>(dg + (size_t)v * Cdg + coff + c);
        const b16x8 x8 = *reinterpret_cast<const b16x8*>(x + (size_t)v * Cx + c);
        b16x8 a8 = {};
        if (add) a8 = *reinterpret_cast<const b16x8*>(add + (size_t)v * Cx + c);
        const float* cp = coef + ((size_t)n * 3) * Ctot + coff + c;
        b16x8 o8;
#pragma unroll
        for (int hq = 0; hq < 2; ++hq) {
            const f32x4 p = *reinterpret_cast<const f32x4*>(cp + 4 * hq);
            const f32x4 q = *reinterpret_cast<const f32x4*>(cp + Ctot + 4 * hq);
            const f32x4 r = *reinterpret_cast<const f32x4*>(cp + 2 * (size_t)Ctot + 4 * hq);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xv = (float)x8[4 * hq + e];
                float o = p[e] * (float)d8[4 * hq + e] + q[e] * xv + r[e];
                if (add) o += (float)a8[4 * hq + e];
                if (relu_mask) o = xv > 0.f ? o : 0.f;
                o8[4 * hq + e] = (__bf16)o;
            }
        }
        *reinterpret_cast<b16x8*>(out + (size_t)v * Cx + c) = o8;
    }
}

template <typename T>
static int gn_bwd_apply_impl(int device, u3d_stream_t stream, const T* dg, int Cdg, int coff, const T* x, int Cx,
                             const float* coef, int Ctot, int64_t voxels_per_n, int N, int relu_mask, const T* add,
                             T* out);

extern "C" int u3d_gn_bwd_apply(int device, u3d_stream_t stream, const float* dg, int Cdg, int coff, const float* x,
                                int Cx, const float* coef, int Ctot, int64_t voxels_per_n, int N, int relu_mask,
                                float* out) {
    return gn_bwd_apply_impl<float>(device, stream, dg, Cdg, coff, x, Cx, coef, Ctot, voxels_per_n, N, relu_mask, nullptr, out);
}

// bf16 activation storage: dg, x, add, out are bf16 tensors, the coefficient table stays fp32
extern "C" int u3d_gn_bwd_apply_b16(int device, u3d_stream_t stream, const void* dg, int Cdg, int coff, const void* x, int Cx,
                                    const float* coef, int Ctot, int64_t voxels_per_n, int N, int relu_mask, const void* add,
                                    void* out) {
    return gn_bwd_apply_impl<__bf16>(device, stream, (const __bf16*)dg, Cdg, coff, (const __bf16*)x, Cx, coef, Ctot, voxels_per_n, N,
                                     relu_mask, (const __bf16*)add, (__bf16*)out);
}

extern "C" int u3d_gn_bwd_apply_add(int device, u3d_stream_t stream, const float* dg, int Cdg, int coff, const float* x,
                                    int Cx, const float* coef, int Ctot, int64_t voxels_per_n, int N, int relu_mask,
                                    const float* add, float* out) {
    if (add == nullptr) return u3d_set_err(U3D_EINVAL, "u3d_gn_bwd_apply_add: add is NULL");
    return gn_bwd_apply_impl<float>(device, stream, dg, Cdg, coff, x, Cx, coef, Ctot, voxels_per_n, N, relu_mask, add, out);
}

template <typename T>
static int gn_bwd_apply_impl(int device, u3d_stream_t stream, const T* dg, int Cdg, int coff, const T* x, int Cx,
                             const float* coef, int Ctot, int64_t voxels_per_n, int N, int relu_mask, const T* add,
                             T* out) {
    U3D_ENTER(device);
    U3D_REQUIRE(dg && x && coef && out && Cdg > 0 && Cx > 0 && coff >= 0 && coff + Cx <= Cdg && Ctot >= coff + Cx &&
                    voxels_per_n > 0 && N > 0,
                "u3d_gn_bwd_apply: bad argument");
    const bool vec = (Cdg % 4 == 0) && (Cx % 4 == 0) && (coff % 4 == 0) && (Ctot % 4 == 0) &&
                     (((uintptr_t)dg | (uintptr_t)x | (uintptr_t)out | (uintptr_t)add) & u3d_vec_align<T>::mask) == 0 &&
                     ((uintptr_t)coef & 15) == 0;
    if constexpr (sizeof(T) == 2) {
        if (vec && Cdg % 8 == 0 && Cx % 8 == 0 && coff % 8 == 0 && Ctot % 8 == 0 &&
            (((uintptr_t)dg | (uintptr_t)x | (uintptr_t)out | (uintptr_t)add) & 15) == 0) {
            const long long total = (long long)N * voxels_per_n * (Cx / 8);
            hipLaunchKernelGGL(gn_bwd_apply_oct_b16_kernel, dim3(grid_for(total, 16384)), dim3(256), 0, (hipStream_t)stream,
                               reinterpret_cast<const __bf16*>(dg), Cdg, coff, reinterpret_cast<const __bf16*>(x), Cx, coef, Ctot,
                               (long long)voxels_per_n, N, relu_mask, reinterpret_cast<const __bf16*>(add), reinterpret_cast<__bf16*>(out));
            U3D_LAUNCH_CHECK();
            return 0;
        }
    }
    if (vec) {
        const long long total = (long long)N * voxels_per_n * (Cx / 4);
        hipLaunchKernelGGL((gn_bwd_apply_kernel<true, T>), dim3(grid_for(total, 16384)), dim3(256), 0, (hipStream_t)stream,
                           dg, Cdg, coff, x, Cx, coef, Ctot, (long long)voxels_per_n, N, relu_mask, add, out);
    } else {
        const long long total = (long long)N * voxels_per_n * Cx;
        hipLaunchKernelGGL((gn_bwd_apply_kernel<false, T>), dim3(grid_for(total, 16384)), dim3(256), 0,
                           (hipStream_t)stream, dg, Cdg, coff, x, Cx, coef, Ctot, (long long)voxels_per_n, N,
                           relu_mask, add, out);
    }
    U3D_LAUNCH_CHECK();
    return 0;
}

// upsampled half of a concat: sum over the children of each low-res voxel; a thread owns VW (4 or 1) channels
template <int VW>
__global__ void gn_bwd_apply_up_kernel(const float* __restrict__ dg, int Cdg, int coff, const float* __restrict__ x1,
                                       int C1, const float* __restrict__ coef, int Ctot, int N, int D, int H, int W,
                                       int D1, int H1, int W1, const int* __restrict__ zlo,
                                       const int* __restrict__ ylo, const int* __restrict__ xlo, int relu_mask,
                                       float* __restrict__ out) {
    const int Q = C1 / VW;
    const long long total = (long long)N * D1 * H1 * W1 * Q;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % Q) * VW;
        long long v = idx / Q;
        const int xx = (int)(v % W1);
        v /= W1;
        const int yy = (int)(v % H1);
        v /= H1;
        const int zz = (int)(v % D1);
        const int n = (int)(v / D1);
        float sum[VW];
#pragma unroll
        for (int k = 0; k < VW; ++k) sum[k] = 0.f;
        int cnt = 0;
        for (int z = zlo[zz]; z < zlo[zz + 1]; ++z)
            for (int y = ylo[yy]; y < ylo[yy + 1]; ++y)
                for (int x = xlo[xx]; x < xlo[xx + 1]; ++x) {
                    const float* src = dg + ((size_t)((n * D + z) * H + y) * W + x) * Cdg + coff + c;
                    if (VW == 4) {
                        const f32x4 t = *reinterpret_cast<const f32x4*>(src);
#pragma unroll
                        for (int k = 0; k < 4; ++k) sum[k] += t[k];
                    } else {
                        sum[0] += src[0];
                    }
                    ++cnt;
                }
        const size_t oi = (size_t)(idx / Q) * C1 + c;
#pragma unroll
        for (int k = 0; k < VW; ++k) {
            const float xv = x1[oi + k];
            const float p = coef[((size_t)n * 3 + 0) * Ctot + coff + c + k], q = coef[((size_t)n * 3 + 1) * Ctot + coff + c + k],
                        r = coef[((size_t)n * 3 + 2) * Ctot + coff + c + k];
            float o = p * sum[k] + (float)cnt * (q * xv + r);
            if (relu_mask && !(xv > 0.f)) o = 0.f;
            out[oi + k] = o;
        }
    }
}

extern "C" int u3d_gn_bwd_apply_up(int device, u3d_stream_t stream, const float* dg, int Cdg, int coff,
                                   const float* x1, int C1, const float* coef, int Ctot, int N, int D, int H, int W,
                                   int D1, int H1, int W1, const int32_t* zlo, const int32_t* ylo,
                                   const int32_t* xlo, int relu_mask, float* out) {
    U3D_ENTER(device);
    U3D_REQUIRE(dg && x1 && coef && out && zlo && ylo && xlo && C1 > 0 && coff >= 0 && coff + C1 <= Cdg &&
                    Ctot >= coff + C1 && N > 0,
                "u3d_gn_bwd_apply_up: bad argument");
    const bool vec = C1 % 4 == 0 && Cdg % 4 == 0 && coff % 4 == 0 && ((uintptr_t)dg & 15) == 0;
    const long long total = (long long)N * D1 * H1 * W1 * (vec ? C1 / 4 : C1);
    if (vec)
        hipLaunchKernelGGL(gn_bwd_apply_up_kernel<4>, dim3(grid_for(total, 16384)), dim3(256), 0, (hipStream_t)stream, dg,
                           Cdg, coff, x1, C1, coef, Ctot, N, D, H, W, D1, H1, W1, zlo, ylo, xlo, relu_mask, out);
    else
        hipLaunchKernelGGL(gn_bwd_apply_up_kernel<1>, dim3(grid_for(total, 16384)), dim3(256), 0, (hipStream_t)stream, dg,
                           Cdg, coff, x1, C1, coef, Ctot, N, D, H, W, D1, H1, W1, zlo, ylo, xlo, relu_mask, out);
    U3D_LAUNCH_CHECK();
    return 0;
}

// GroupNorm backward on the LOW-RES producer of a level that upsamples n -> 2n + 1 along the axes with e = 1 (round 5): dlow holds the
// children sums of dg, and a low-res cell has (2 + [e_z && z == 0]) (2 + [e_y && y == 0]) (2 + [e_x && x == 0]) children (the first cell
// of a shifted axis has three), so  out = (p * dlow + children * (q * x + r)) * [x > 0 if relu_mask]  — the exact-2x levels fold the
// constant 8 into the coefficient table instead (u3d_gn_bwd_apply).
__global__ void gn_bwd_apply_cnt_kernel(const float* __restrict__ dlow, const float* __restrict__ x, const float* __restrict__ coef,
                                        int Ctot, int coff, int N, int D1, int H1, int W1, int C, int ez, int ey, int ex, int relu_mask,
                                        float* __restrict__ out) {
    const int Q = C >> 2;
    const long long total = (long long)N * D1 * H1 * W1 * Q;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % Q) * 4;
        long long v = idx / Q;
        const int xx = (int)(v % W1);
        v /= W1;
        const int yy = (int)(v % H1);
        v /= H1;
        const int zz = (int)(v % D1);
        const int n = (int)(v / D1);
        const float cnt = (float)((2 + (ez && zz == 0)) * (2 + (ey && yy == 0)) * (2 + (ex && xx == 0)));
        const size_t oi = (size_t)(idx / Q) * C + c;
        const f32x4 dg = *reinterpret_cast<const f32x4*>(dlow + oi), xv = *reinterpret_cast<const f32x4*>(x + oi);
        const f32x4 pp = *reinterpret_cast<const f32x4*>(coef + ((size_t)n * 3 + 0) * Ctot + coff + c);
        const f32x4 qq = *reinterpret_cast<const f32x4*>(coef + ((size_t)n * 3 + 1) * Ctot + coff + c);
        const f32x4 rr = *reinterpret_cast<const f32x4*>(coef + ((size_t)n * 3 + 2) * Ctot + coff + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[e] = pp[e] * dg[e] + cnt * (qq[e] * xv[e] + rr[e]);
            if (relu_mask && !(xv[e] > 0.f)) o[e] = 0.f;
        }
        *reinterpret_cast<f32x4*>(out + oi) = o;
    }
}

extern "C" int u3d_gn_bwd_apply_children(int device, u3d_stream_t stream, const float* dlow, const float* x, const float* coef, int Ctot,
                                         int coff, int N, int D1, int H1, int W1, int C, int ez, int ey, int ex, int relu_mask,
                                         float* out) {
    U3D_ENTER(device);
    U3D_REQUIRE(dlow && x && coef && out && N > 0 && D1 > 0 && H1 > 0 && W1 > 0 && C > 0 && C % 4 == 0 && coff >= 0 && coff % 4 == 0 &&
                    Ctot % 4 == 0 && coff + C <= Ctot && (((uintptr_t)dlow | (uintptr_t)x | (uintptr_t)coef | (uintptr_t)out) & 15) == 0,
                "u3d_gn_bwd_apply_children: bad argument (channel counts / offsets must be multiples of 4, 16-byte aligned pointers)");
    const long long total = (long long)N * D1 * H1 * W1 * (C / 4);
    hipLaunchKernelGGL(gn_bwd_apply_cnt_kernel, dim3(grid_for(total, 16384)), dim3(256), 0, (hipStream_t)stream, dlow, x, coef, Ctot, coff,
                       N, D1, H1, W1, C, ez, ey, ex, relu_mask, out);
    U3D_LAUNCH_CHECK();
    return 0;
}

// Round 5, decoder levels that upsample n -> 2n + 1 (F.interpolate(nearest) to an odd skip size, buildingblocks.py:598-614): the
// data gradient of the outputs in the near-boundary slab arrives as a FULL-RESOLUTION gradient dv of the upsampled channels (written by
// u3d_conv3d_box inside the slab's one-voxel dilation only).  Backward of the nearest upsampling for exactly those low-res cells whose
// children lie in that region — cell j is AFFECTED if cz && jz < cz || cy && jy < cy || cx && jx < cx (every child of an affected
// cell is inside the region) — ADDED to dlow, with the cell's share of the GroupNorm-backward sums (sum add, sum add * x_low).
__global__ __launch_bounds__(256) void nearest_childsum_add_kernel(const float* __restrict__ dv, const float* __restrict__ xlow,
                                                                   float* __restrict__ dlow, double* __restrict__ gstats, int N, int D,
                                                                   int H, int W, int D1, int H1, int W1, int C,
                                                                   const int* __restrict__ zlo, const int* __restrict__ ylo,
                                                                   const int* __restrict__ xlo, int cz, int cy, int cx) {
    extern __shared__ __attribute__((aligned(16))) unsigned char childsum_lds[];
    double* red = reinterpret_cast<double*>(childsum_lds);  // [C][2]: this block's partial sums of the sample it is working on
    const int Q = C >> 2;
    const long long per_n = (long long)D1 * H1 * W1 * Q;
    for (int n = 0; n < N; ++n) {
        for (int k = threadIdx.x; k < 2 * C; k += blockDim.x) red[k] = 0.0;
        __syncthreads();
        for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < per_n; idx += (long long)gridDim.x * blockDim.x) {
            const int c = (int)(idx % Q) * 4;
            long long v = idx / Q;
            const int xx = (int)(v % W1);
            v /= W1;
            const int yy = (int)(v % H1);
            const int zz = (int)(v / H1);
            if (!((cz && zz < cz) || (cy && yy < cy) || (cx && xx < cx))) continue;
            f32x4 sum = {0.f, 0.f, 0.f, 0.f};
            for (int z = zlo[zz]; z < zlo[zz + 1]; ++z)
                for (int y = ylo[yy]; y < ylo[yy + 1]; ++y)
                    for (int x = xlo[xx]; x < xlo[xx + 1]; ++x)
                        sum += *reinterpret_cast<const f32x4*>(dv + ((size_t)((n * D + z) * H + y) * W + x) * C + c);
            const size_t oi = ((size_t)((n * D1 + zz) * H1 + yy) * W1 + xx) * C + c;
            f32x4* dst = reinterpret_cast<f32x4*>(dlow + oi);
            *dst = *dst + sum;
            if (gstats) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(xlow + oi);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    __hip_atomic_fetch_add(&red[2 * (c + e)], (double)sum[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(&red[2 * (c + e) + 1], (double)sum[e] * (double)xv[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        __syncthreads();
        if (gstats)
            for (int k = threadIdx.x; k < 2 * C; k += blockDim.x)
                if (red[k] != 0.0) u3d_atomic_add_f64(&gstats[(size_t)n * C * 2 + k], red[k]);
        __syncthreads();
    }
}

extern "C" int u3d_nearest_childsum_add(int device, u3d_stream_t stream, const float* dv, const float* x_low, float* dlow,
                                        double* gstats, int N, int D, int H, int W, int D1, int H1, int W1, int C, const int32_t* zlo,
                                        const int32_t* ylo, const int32_t* xlo, int cz, int cy, int cx) {
    U3D_ENTER(device);
    U3D_REQUIRE(dv && dlow && zlo && ylo && xlo && N > 0 && D > 0 && H > 0 && W > 0 && D1 > 0 && H1 > 0 && W1 > 0 && C > 0 && C % 4 == 0 &&
                    C <= 2048 && (gstats == nullptr || x_low != nullptr) && (cz > 0 || cy > 0 || cx > 0),
                "u3d_nearest_childsum_add: bad argument (C must be a multiple of 4, <= 2048)");
    U3D_REQUIRE((((uintptr_t)dv | (uintptr_t)dlow | (uintptr_t)x_low) & 15) == 0, "u3d_nearest_childsum_add: 16-byte alignment");
    const long long per_n = (long long)D1 * H1 * W1 * (C / 4);
    long long blocks = (per_n + 255) / 256;
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(nearest_childsum_add_kernel, dim3((unsigned)blocks), dim3(256), (size_t)2 * C * sizeof(double), (hipStream_t)stream,
                       dv, x_low, dlow, gstats, N, D, H, W, D1, H1, W1, C, zlo, ylo, xlo, cz, cy, cx);
    U3D_LAUNCH_CHECK();
    return 0;
}

// =================================================================================================
// MaxPool3d(2): stride 2, floor.  One thread per (n, out voxel, channel).
template <typename T = float>
__global__ void maxpool2_fwd_kernel(const T* __restrict__ x, int N, int D, int H, int W, int C,
                                    T* __restrict__ out, uint8_t* __restrict__ argmax) {
    const int D2 = D >> 1, H2 = H >> 1, W2 = W >> 1;
    const long long total = (long long)N * D2 * H2 * W2 * C;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C);
        long long v = idx / C;
        const int xo = (int)(v % W2);
        v /= W2;
        const int yo = (int)(v % H2);
        v /= H2;
        const int zo = (int)(v % D2);
        const int n = (int)(v / D2);
        float best = -INFINITY;
        int bi = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int z = 2 * zo + (k >> 2), y = 2 * yo + ((k >> 1) & 1), xx = 2 * xo + (k & 1);
            const float val = u3d_ld(x + ((size_t)((n * D + z) * H + y) * W + xx) * C + c);
            if (val > best || val != val) {  // first max in scan order; NaN propagates (ATen max_pool3d)
                best = val;
                bi = k;
            }
        }
        u3d_st(out + idx, best);
        argmax[idx] = (uint8_t)bi;
    }
}

// 16-byte version (bf16 storage: 0.142 -> 0.086 ms per step on config 4): a thread owns VW consecutive channels (8 bf16) of one output voxel — eight 16-byte loads in flight, one
// 16-byte store and VW arg-max bytes; same scan order and NaN rule per element as the scalar kernel above (which moves 4 / 2 bytes per lane)
template <typename T, int VW>
__global__ __launch_bounds__(256) void maxpool2_fwd_vec_kernel(const T* __restrict__ x, int N, int D, int H, int W, int C, T* __restrict__ out,
                                                               uint8_t* __restrict__ argmax) {
    typedef T vec_t __attribute__((ext_vector_type(VW)));
    typedef uint8_t idx_t __attribute__((ext_vector_type(VW)));
    const int D2 = D >> 1, H2 = H >> 1, W2 = W >> 1, Q = C / VW;
    const long long total = (long long)N * D2 * H2 * W2 * Q;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int qd = (int)(idx % Q);
        long long v = idx / Q;
        const int xo = (int)(v % W2);
        v /= W2;
        const int yo = (int)(v % H2);
        v /= H2;
        const int zo = (int)(v % D2);
        const int n = (int)(v / D2);
        const T* base = x + ((size_t)((n * D + 2 * zo) * H + 2 * yo) * W + 2 * xo) * C + qd * VW;
        vec_t val[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            val[k] = *reinterpret_cast<const vec_t*>(base + ((size_t)((k >> 2) * H + ((k >> 1) & 1)) * W + (k & 1)) * C);
        vec_t best;
        idx_t bi;
#pragma unroll
        for (int e = 0; e < VW; ++e) {
            float b = -INFINITY;
            int i = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float f = (float)val[k][e];
                if (f > b || f != f) {
                    b = f;
                    i = k;
                }
            }
            best[e] = (T)b;  // (one of the inputs: exact in T)
            bi[e] = (uint8_t)i;
        }
        *reinterpret_cast<vec_t*>(out + idx * VW) = best;
        *reinterpret_cast<idx_t*>(argmax + idx * VW) = bi;
    }
}

extern "C" int u3d_maxpool2_fwd(int device, u3d_stream_t stream, const float* x, int N, int D, int H, int W, int C,
                                float* out, uint8_t* argmax, double* out_stats) {
    U3D_ENTER(device);
    U3D_REQUIRE(x && out && argmax && N > 0 && D >= 2 && H >= 2 && W >= 2 && C > 0, "u3d_maxpool2_fwd: bad argument");
    const long long total = (long long)N * (D / 2) * (H / 2) * (W / 2) * C;
    // (the 16-byte variant below is for bf16 storage: with fp32 tensors it measured 0.131 against this kernel's 0.116 ms per step on config 2)
    hipLaunchKernelGGL(maxpool2_fwd_kernel<float>, dim3(grid_for(total, 16384)), dim3(256), 0, (hipStream_t)stream, x, N, D,
                       H, W, C, out, argmax);
    U3D_LAUNCH_CHECK();
    if (out_stats) {
        u3d_src_t s = {};
        s.p0 = out;
        s.C0 = C;
        return u3d_chan_stats(device, stream, &s, N, D / 2, H / 2, W / 2, out_stats);
    }
    return 0;
}

extern "C" int u3d_maxpool2_fwd_b16(int device, u3d_stream_t stream, const void* x, int N, int D, int H, int W, int C, void* out,
                                    uint8_t* argmax) {
    U3D_ENTER(device);
    U3D_REQUIRE(x && out && argmax && N > 0 && D >= 2 && H >= 2 && W >= 2 && C > 0, "u3d_maxpool2_fwd_b16: bad argument");
    const long long total = (long long)N * (D / 2) * (H / 2) * (W / 2) * C;
    if (C % 8 == 0 && (((uintptr_t)x | (uintptr_t)out) & 15) == 0 && ((uintptr_t)argmax & 7) == 0)
        hipLaunchKernelGGL((maxpool2_fwd_vec_kernel<__bf16, 8>), dim3(grid_for(total / 8, 16384)), dim3(256), 0, (hipStream_t)stream,
                           (const __bf16*)x, N, D, H, W, C, (__bf16*)out, argmax);
    else
        hipLaunchKernelGGL(maxpool2_fwd_kernel<__bf16>, dim3(grid_for(total, 16384)), dim3(256), 0, (hipStream_t)stream,
                           (const __bf16*)x, N, D, H, W, C, (__bf16*)out, argmax);
    U3D_LAUNCH_CHECK();
    return 0;
}

// dz_e = (skip_grad + scatter(dpool)) * (e > 0); one thread per (n, 2x2x2 window, channel unit), windows cover ceil dims.
// A unit is VW (4 or 1) channels.  The skip gradient is either absent, a plain tensor (sdg with Csdg channels per voxel),
// or — fused — the GroupNorm backward of the decoder's first conv restricted to the skip channels:
//     skip = ps*sdg[v, c] + qs*e[v, c] + rs      (scoef[N][3][Cstot], skip channels first => no channel offset)
// which removes one full write + read of the skip gradient per decoder level.
template <int VW, typename T = float>
__global__ void maxpool2_bwd_merge_kernel(const T* __restrict__ dg, const T* __restrict__ pooled,
                                          const uint8_t* __restrict__ argmax, const float* __restrict__ coef,
                                          const T* __restrict__ sdg, int Csdg, const float* __restrict__ scoef,
                                          int Cstot, const T* __restrict__ e, int N, int D, int H, int W, int C,
                                          int relu_mask, T* __restrict__ out) {
    const int D2 = D >> 1, H2 = H >> 1, W2 = W >> 1;
    const int Dc = (D + 1) >> 1, Hc = (H + 1) >> 1, Wc = (W + 1) >> 1;
    const int Q = C / VW;
    const long long total = (long long)N * Dc * Hc * Wc * Q;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % Q) * VW;
        long long v = idx / Q;
        const int xo = (int)(v % Wc);
        v /= Wc;
        const int yo = (int)(v % Hc);
        v /= Hc;
        const int zo = (int)(v % Dc);
        const int n = (int)(v / Dc);
        const bool pv = zo < D2 && yo < H2 && xo < W2;
        float dp[VW], ps[VW], qs[VW], rs[VW];
        int am[VW];
#pragma unroll
        for (int k = 0; k < VW; ++k) {
            dp[k] = 0.f;
            am[k] = -1;
            ps[k] = 1.f, qs[k] = 0.f, rs[k] = 0.f;
            if (scoef) {
                ps[k] = scoef[((size_t)n * 3 + 0) * Cstot + c + k];
                qs[k] = scoef[((size_t)n * 3 + 1) * Cstot + c + k];
                rs[k] = scoef[((size_t)n * 3 + 2) * Cstot + c + k];
            }
        }
        if (pv) {
            const size_t pi = ((size_t)((n * D2 + zo) * H2 + yo) * W2 + xo) * C + c;
#pragma unroll
            for (int k = 0; k < VW; ++k) {
                dp[k] = u3d_ld(dg + pi + k);
                if (coef)
                    dp[k] = coef[((size_t)n * 3 + 0) * C + c + k] * dp[k] + coef[((size_t)n * 3 + 1) * C + c + k] * u3d_ld(pooled + pi + k) +
                            coef[((size_t)n * 3 + 2) * C + c + k];
                am[k] = argmax[pi + k];
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int z = 2 * zo + (j >> 2), y = 2 * yo + ((j >> 1) & 1), xx = 2 * xo + (j & 1);
            if (z < D && y < H && xx < W) {
                const size_t vi = (size_t)((n * D + z) * H + y) * W + xx;
                const size_t ei = vi * C + c;
                float ev[VW], sv[VW], o[VW];
                if (VW == 4) {
                    const f32x4 t = (relu_mask || scoef) ? u3d_ldq(e + ei) : f32x4{1.f, 1.f, 1.f, 1.f};
                    const f32x4 u = sdg ? u3d_ldq(sdg + vi * Csdg + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < 4; ++k) ev[k] = t[k], sv[k] = u[k];
                } else {
                    ev[0] = (relu_mask || scoef) ? u3d_ld(e + ei) : 1.f;
                    sv[0] = sdg ? u3d_ld(sdg + vi * Csdg + c) : 0.f;
                }
#pragma unroll
                for (int k = 0; k < VW; ++k) {
                    float g = sdg ? (scoef ? ps[k] * sv[k] + qs[k] * ev[k] + rs[k] : sv[k]) : 0.f;
                    if (j == am[k]) g += dp[k];
                    if (relu_mask && !(ev[k] > 0.f)) g = 0.f;
                    o[k] = g;
                }
                if (VW == 4)
                    u3d_stq(out + ei, f32x4{o[0], o[1], o[2], o[3]});
                else
                    u3d_st(out + ei, o[0]);
            }
        }
    }
}

template <typename T>
static int maxpool2_bwd_merge_impl(int device, u3d_stream_t stream, const T* dg, const T* pooled,
                                   const uint8_t* argmax, const float* coef, const T* sdg, int Csdg, const float* scoef,
                                   int Cstot, const T* e, int N, int D, int H, int W, int C, int relu_mask, T* out) {
    U3D_ENTER(device);
    U3D_REQUIRE(dg && argmax && out && (coef == nullptr || pooled) && (!(relu_mask || scoef) || e) && N > 0 && C > 0 &&
                    (!sdg || Csdg >= C) && (!scoef || (sdg && Cstot >= C)),
                "u3d_maxpool2_bwd_merge: bad argument");
    const bool vec = C % 4 == 0 && (!sdg || Csdg % 4 == 0) &&
                     (((uintptr_t)e | (uintptr_t)sdg | (uintptr_t)out) & u3d_vec_align<T>::mask) == 0;
    const long long total = (long long)N * ((D + 1) / 2) * ((H + 1) / 2) * ((W + 1) / 2) * (vec ? C / 4 : C);
    if (vec)
        hipLaunchKernelGGL((maxpool2_bwd_merge_kernel<4, T>), dim3(grid_for(total, 16384)), dim3(256), 0, (hipStream_t)stream, dg,
                           pooled, argmax, coef, sdg, Csdg, scoef, Cstot, e, N, D, H, W, C, relu_mask, out);
    else
        hipLaunchKernelGGL((maxpool2_bwd_merge_kernel<1, T>), dim3(grid_for(total, 16384)), dim3(256), 0, (hipStream_t)stream, dg,
                           pooled, argmax, coef, sdg, Csdg, scoef, Cstot, e, N, D, H, W, C, relu_mask, out);
    U3D_LAUNCH_CHECK();
    return 0;
}

extern "C" int u3d_maxpool2_bwd_merge(int device, u3d_stream_t stream, const float* dg, const float* pooled,
                                      const uint8_t* argmax, const float* coef, const float* skip_grad,
                                      const float* e, int N, int D, int H, int W, int C, int relu_mask, float* out) {
    return maxpool2_bwd_merge_impl<float>(device, stream, dg, pooled, argmax, coef, skip_grad, C, nullptr, 0, e, N, D, H, W, C,
                                          relu_mask, out);
}

extern "C" int u3d_maxpool2_bwd_merge_b16(int device, u3d_stream_t stream, const void* dg, const void* pooled, const uint8_t* argmax,
                                          const float* coef, const void* skip_grad, const void* e, int N, int D, int H, int W, int C,
                                          int relu_mask, void* out) {
    return maxpool2_bwd_merge_impl<__bf16>(device, stream, (const __bf16*)dg, (const __bf16*)pooled, argmax, coef, (const __bf16*)skip_grad,
                                           C, nullptr, 0, (const __bf16*)e, N, D, H, W, C, relu_mask, (__bf16*)out);
}

extern "C" int u3d_maxpool2_bwd_merge_gn(int device, u3d_stream_t stream, const float* dg, const float* pooled,
                                         const uint8_t* argmax, const float* coef, const float* skip_dg, int Cdg,
                                         const float* skip_coef, int Ctot, const float* e, int N, int D, int H, int W, int C,
                                         int relu_mask, float* out) {
    if (!skip_dg || !skip_coef) return u3d_set_err(U3D_EINVAL, "u3d_maxpool2_bwd_merge_gn: skip_dg / skip_coef are NULL");
    return maxpool2_bwd_merge_impl<float>(device, stream, dg, pooled, argmax, coef, skip_dg, Cdg, skip_coef, Ctot, e, N, D, H, W, C,
                                          relu_mask, out);
}

// =================================================================================================
// head: 1x1x1 conv + bias + activation; x NDHWC (N,V,Cin) -> logits/probs NCDHW (N,Cout,V)
constexpr int HEAD_MAXCO = 16;
constexpr int HEAD_MAXCI = 256;

__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ b, int N, long long V, int Cin,
                                                       int Cout, int act, float* __restrict__ logits,
                                                       float* __restrict__ probs) {
    __shared__ float ws[HEAD_MAXCO * HEAD_MAXCI + HEAD_MAXCO];
    for (int i = threadIdx.x; i < Cout * Cin; i += blockDim.x) ws[i] = w[i];
    for (int i = threadIdx.x; i < Cout; i += blockDim.x) ws[HEAD_MAXCO * HEAD_MAXCI + i] = b[i];
    __syncthreads();
    const long long total = (long long)N * V;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(idx / V);
        const long long v = idx - (long long)n * V;
        float acc[HEAD_MAXCO];
#pragma unroll
        for (int o = 0; o < HEAD_MAXCO; ++o) acc[o] = o < Cout ? ws[HEAD_MAXCO * HEAD_MAXCI + o] : 0.f;
        const float* xr = x + (size_t)idx * Cin;
        for (int c = 0; c < Cin; ++c) {
            const float xv = xr[c];
#pragma unroll
            for (int o = 0; o < HEAD_MAXCO; ++o)
                if (o < Cout) acc[o] = fmaf(xv, ws[o * Cin + c], acc[o]);
        }
        float mx = -INFINITY, den = 0.f;
        if (act == 2) {
#pragma unroll
            for (int o = 0; o < HEAD_MAXCO; ++o)
                if (o < Cout) mx = fmaxf(mx, acc[o]);
#pragma unroll
            for (int o = 0; o < HEAD_MAXCO; ++o)
                if (o < Cout) den += expf(acc[o] - mx);
        }
#pragma unroll
        for (int o = 0; o < HEAD_MAXCO; ++o) {
            if (o < Cout) {
                const size_t oi = ((size_t)n * Cout + o) * V + v;
                logits[oi] = acc[o];
                if (probs) {
                    float pr = acc[o];
                    if (act == 1) pr = 1.f / (1.f + expf(-acc[o]));
                    if (act == 2) pr = expf(acc[o] - mx) / den;
                    probs[oi] = pr;
                }
            }
        }
    }
}

// Vectorised variant: G = Cin/4 lanes (power of two, <= 64) share one voxel, each loads one float4 of the row
// (fully coalesced 16 B/lane), partial dot products are combined with a butterfly shuffle.
template <int G, typename T = float>
__global__ __launch_bounds__(256) void head_fwd_vec_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ b, int N, long long V, int Cin,
                                                           int Cout, int act, float* __restrict__ logits,
                                                           float* __restrict__ probs) {
    const int t = threadIdx.x;
    const int sub = t & (G - 1);
    const long long total = (long long)N * V;
    const long long vpb = 256 / G;
    for (long long idx = (long long)blockIdx.x * vpb + t / G; idx < total; idx += (long long)gridDim.x * vpb) {
        const f32x4 xv = u3d_ldq(x + (size_t)idx * Cin + 4 * sub);
        float acc[HEAD_MAXCO];
#pragma unroll
        for (int o = 0; o < HEAD_MAXCO; ++o) {
            acc[o] = 0.f;
            if (o < Cout) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (size_t)o * Cin + 4 * sub);
                float p = xv[0] * wv[0] + xv[1] * wv[1] + xv[2] * wv[2] + xv[3] * wv[3];
#pragma unroll
                for (int m = G >> 1; m > 0; m >>= 1) p += __shfl_xor(p, m);
                acc[o] = p + b[o];
            }
        }
        if (sub == 0) {
            const int n = (int)(idx / V);
            const long long v = idx - (long long)n * V;
            float mx = -INFINITY, den = 0.f;
            if (act == 2) {
#pragma unroll
                for (int o = 0; o < HEAD_MAXCO; ++o)
                    if (o < Cout) mx = fmaxf(mx, acc[o]);
#pragma unroll
                for (int o = 0; o < HEAD_MAXCO; ++o)
                    if (o < Cout) den += expf(acc[o] - mx);
            }
#pragma unroll
            for (int o = 0; o < HEAD_MAXCO; ++o) {
                if (o < Cout) {
                    const size_t oi = ((size_t)n * Cout + o) * V + v;
                    logits[oi] = acc[o];
                    if (probs) {
                        float pr = acc[o];
                        if (act == 1) pr = 1.f / (1.f + expf(-acc[o]));
                        if (act == 2) pr = expf(acc[o] - mx) / den;
                        probs[oi] = pr;
                    }
                }
            }
        }
    }
}

// bf16 storage, Cin = 32 / 64 / 128 and at most 4 outputs (the segmentation heads of the shipped configs): the kernel above lets
// Cin/4 lanes share a voxel — 8-byte loads, four shuffle rounds per output and a 4-byte store from every 16th lane (1.4 TB/s on
// config 4's 262 MB input).  Here a wave copies 64 consecutive voxels with 16-byte loads into its own LDS rows, every lane then
// owns ONE voxel (row reads are conflict-free at a 16-byte row pad), and a wave-instruction stores 64 consecutive voxels.
// The sum is formed in the SAME order as above (four rounded products added left to right, then the butterfly's pairing s ^ G/2, .., s ^ 1), so
// the two kernels agree bit for bit.
// four products, each rounded, summed left to right: what head_fwd_vec_kernel's vectorised multiply compiles to (v_pk_mul_f32 + adds)
__device__ __forceinline__ float head_dot4(float x0, float x1, float x2, float x3, const f32x4& w4) {
#pragma clang fp contract(off)
    return ((x0 * w4[0] + x1 * w4[1]) + x2 * w4[2]) + x3 * w4[3];
}

template <int CIN, int COUT>
__global__ __launch_bounds__(256) void head_fwd_b16_rows_kernel(const __bf16* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ b, int N, long long V, int act,
                                                                float* __restrict__ logits, float* __restrict__ probs) {
    constexpr int ROW = CIN * 2 + 16, OPV = CIN / 8, NLD = OPV, G = CIN / 4;  // octets per voxel = 16-byte loads per lane and round
    __shared__ __attribute__((aligned(16))) char rows[4][64 * ROW];
    __shared__ __attribute__((aligned(16))) float wsh[COUT * CIN];  // (uniform reads broadcast; scalar loads of 64-512 weights spill SGPRs)
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    for (int i = t; i < COUT * CIN; i += 256) wsh[i] = w[i];
    __syncthreads();
    char* mine = rows[wv];
    const long long total = (long long)N * V;
    typedef __bf16 b16x8 __attribute__((ext_vector_type(8)));
    for (long long v0 = ((long long)blockIdx.x * 4 + wv) * 64; v0 < total; v0 += (long long)gridDim.x * 256) {
        const long long left = total - v0;  // (>= 1; the last round may hold fewer than 64 voxels)
        b16x8 it[NLD];
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int item = i * 64 + lane, vox = item / OPV;
            it[i] = b16x8{};
            if (vox < left) it[i] = *reinterpret_cast<const b16x8*>(x + (size_t)v0 * CIN + (size_t)item * 8);
        }
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int item = i * 64 + lane, vox = item / OPV, oc = item - vox * OPV;
            *reinterpret_cast<b16x8*>(mine + vox * ROW + oc * 16) = it[i];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float xs[CIN];
#pragma unroll
        for (int j = 0; j < OPV; ++j) {
            const b16x8 r = *reinterpret_cast<const b16x8*>(mine + lane * ROW + j * 16);
#pragma unroll
            for (int e = 0; e < 8; ++e) xs[8 * j + e] = (float)r[e];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float acc[COUT];
#pragma unroll
        for (int o = 0; o < COUT; ++o) {
            asm volatile("" ::: "memory");  // (keeps the weight reads inside the round: hoisted, COUT * CIN of them live in VGPRs)
            float pr[G];
#pragma unroll
            for (int s_ = 0; s_ < G; ++s_) {
                const f32x4 wr = *reinterpret_cast<const f32x4*>(wsh + o * CIN + 4 * s_);
                pr[s_] = head_dot4(xs[4 * s_], xs[4 * s_ + 1], xs[4 * s_ + 2], xs[4 * s_ + 3], wr);
            }
#pragma unroll
            for (int m = G >> 1; m > 0; m >>= 1)
#pragma unroll
                for (int s_ = 0; s_ < m; ++s_) pr[s_] = pr[s_] + pr[s_ + m];
            acc[o] = pr[0] + b[o];
        }
        if (lane < left) {
            const long long idx = v0 + lane;
            const int n = (int)(idx / V);
            const long long v = idx - (long long)n * V;
            float mx = -INFINITY, den = 0.f;
            if (act == 2) {
#pragma unroll
                for (int o = 0; o < COUT; ++o) mx = fmaxf(mx, acc[o]);
#pragma unroll
                for (int o = 0; o < COUT; ++o) den += expf(acc[o] - mx);
            }
#pragma unroll
            for (int o = 0; o < COUT; ++o) {
                const size_t oi = ((size_t)n * COUT + o) * V + v;
                logits[oi] = acc[o];
                if (probs) {
                    float p_ = acc[o];
                    if (act == 1) p_ = 1.f / (1.f + expf(-acc[o]));
                    if (act == 2) p_ = expf(acc[o] - mx) / den;
                    probs[oi] = p_;
                }
            }
        }
    }
}

template <int CIN, int COUT>
static void launch_head_rows(const __bf16* x, const float* w, const float* b, int N, long long V, int act, float* logits, float* probs,
                             hipStream_t st) {
    long long rounds = ((long long)N * V + 255) / 256;
    if (rounds > 4096) rounds = 4096;
    hipLaunchKernelGGL((head_fwd_b16_rows_kernel<CIN, COUT>), dim3((unsigned)rounds), dim3(256), 0, st, x, w, b, N, V, act, logits, probs);
}

// bf16 activation storage: x is a bf16 tensor (vector path only: Cin/4 a power of two <= 64), logits / probabilities stay fp32
extern "C" int u3d_conv1x1_head_fwd_b16(int device, u3d_stream_t stream, const void* x, const float* w, const float* b, int N,
                                        int64_t V, int Cin, int Cout, int act, float* logits, float* probs) {
    U3D_ENTER(device);
    const int G = Cin / 4;
    U3D_REQUIRE(x && w && b && logits && N > 0 && V > 0 && Cout >= 1 && Cout <= HEAD_MAXCO && Cin % 4 == 0 && G >= 1 && G <= 64 &&
                    (G & (G - 1)) == 0 && ((uintptr_t)x & 7) == 0 && ((uintptr_t)w & 15) == 0,
                "u3d_conv1x1_head_fwd_b16: needs Cin/4 a power of two <= 64 and Cout <= %d (got %d, %d)", HEAD_MAXCO, Cin, Cout);
    hipStream_t st = (hipStream_t)stream;
    const long long tot = (long long)N * V;
    const __bf16* xb = (const __bf16*)x;
    // (one or two outputs from 32 / 64 channels: the binary heads of the shipped configs; wider ones would keep Cout * Cin weights
    // in registers)
    if (((uintptr_t)x & 15) == 0 && ((Cin == 32 && Cout <= 2) || (Cin == 64 && Cout == 1))) {
        if (Cin == 64) launch_head_rows<64, 1>(xb, w, b, N, (long long)V, act, logits, probs, st);
        else if (Cout == 1) launch_head_rows<32, 1>(xb, w, b, N, (long long)V, act, logits, probs, st);
        else launch_head_rows<32, 2>(xb, w, b, N, (long long)V, act, logits, probs, st);
        U3D_LAUNCH_CHECK();
        return 0;
    }
#define U3D_HEAD_FWD16(GG)                                                                                                 \
    hipLaunchKernelGGL((head_fwd_vec_kernel<GG, __bf16>), dim3(grid_for(tot * GG, 8192)), dim3(256), 0, st, xb, w, b, N, \
                       (long long)V, Cin, Cout, act, logits, probs)
    if (G == 1) U3D_HEAD_FWD16(1);
    else if (G == 2) U3D_HEAD_FWD16(2);
    else if (G == 4) U3D_HEAD_FWD16(4);
    else if (G == 8) U3D_HEAD_FWD16(8);
    else if (G == 16) U3D_HEAD_FWD16(16);
    else if (G == 32) U3D_HEAD_FWD16(32);
    else U3D_HEAD_FWD16(64);
#undef U3D_HEAD_FWD16
    U3D_LAUNCH_CHECK();
    return 0;
}

// ---- wide heads (round 5): more than HEAD_MAXCO outputs (model.py:88-91 allows any out_channels — multi-class segmentation).  Outputs
// are processed in tiles of HEAD_MAXCO (grid.y): a thread owns one voxel and one tile, the tile's weights sit in LDS; Sigmoid is
// applied in the same pass, Softmax(dim=1) needs every logit of the voxel and runs as a second pass over the logits (coalesced:
// the voxel index is the fastest dimension of NCDHW).
constexpr int HEAD_WIDE_MAXCO = 1024;

__global__ __launch_bounds__(256) void head_fwd_wide_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                            int N, long long V, int Cin, int Cout, int act, float* __restrict__ logits,
                                                            float* __restrict__ probs) {
    __shared__ float ws[HEAD_MAXCO * HEAD_MAXCI + HEAD_MAXCO];
    const int o0 = blockIdx.y * HEAD_MAXCO, no = min(HEAD_MAXCO, Cout - o0);
    for (int i = threadIdx.x; i < no * Cin; i += blockDim.x) ws[i] = w[(size_t)o0 * Cin + i];
    for (int i = threadIdx.x; i < no; i += blockDim.x) ws[HEAD_MAXCO * HEAD_MAXCI + i] = b[o0 + i];
    __syncthreads();
    const long long total = (long long)N * V;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(idx / V);
        const long long v = idx - (long long)n * V;
        float acc[HEAD_MAXCO];
#pragma unroll
        for (int o = 0; o < HEAD_MAXCO; ++o) acc[o] = o < no ? ws[HEAD_MAXCO * HEAD_MAXCI + o] : 0.f;
        const float* xr = x + (size_t)idx * Cin;
        for (int c = 0; c < Cin; ++c) {
            const float xv = xr[c];
#pragma unroll
            for (int o = 0; o < HEAD_MAXCO; ++o)
                if (o < no) acc[o] = fmaf(xv, ws[o * Cin + c], acc[o]);
        }
#pragma unroll
        for (int o = 0; o < HEAD_MAXCO; ++o) {
            if (o < no) {
                const size_t oi = ((size_t)n * Cout + o0 + o) * V + v;
                logits[oi] = acc[o];
                if (probs && act != 2) probs[oi] = act == 1 ? 1.f / (1.f + expf(-acc[o])) : acc[o];
            }
        }
    }
}

// probs[n,:,v] = softmax(logits[n,:,v]) — max, denominator and quotient exactly as the fused kernels compute them
__global__ __launch_bounds__(256) void head_softmax_wide_kernel(const float* __restrict__ logits, int N, long long V, int Cout,
                                                                float* __restrict__ probs) {
    const long long total = (long long)N * V;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(idx / V);
        const long long v = idx - (long long)n * V;
        const float* lp = logits + (size_t)n * Cout * V + v;
        float mx = -INFINITY, den = 0.f;
        for (int o = 0; o < Cout; ++o) mx = fmaxf(mx, lp[(size_t)o * V]);
        for (int o = 0; o < Cout; ++o) den += expf(lp[(size_t)o * V] - mx);
        float* pp = probs + (size_t)n * Cout * V + v;
        for (int o = 0; o < Cout; ++o) pp[(size_t)o * V] = expf(lp[(size_t)o * V] - mx) / den;
    }
}

// fp32 twin of head_fwd_b16_rows_kernel (round 6): the vectorised kernel lets Cin / 4 lanes share a voxel — one 16-byte load per lane and
// round, three shuffle rounds per output and a 4-byte store from every 8th lane: 3.1 TB/s on the 268 MB input of the bench workload.  Here
// a wave copies 64 consecutive voxels with CIN / 4 independent 16-byte loads per lane into its own LDS rows, every lane then owns ONE voxel
// and a wave-instruction stores 64 consecutive voxels.  Same summation order (head_dot4, then the butterfly's pairing): bit-identical.
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void head_fwd_f32_rows_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ b, int N, long long V, int act,
                                                                float* __restrict__ logits, float* __restrict__ probs) {
    constexpr int ROW = CIN * 4 + 16, QPV = CIN / 4, G = CIN / 4;  // bytes per LDS row (16-byte pad); quads per voxel = loads per lane
    __shared__ __attribute__((aligned(16))) char rows[4][64 * ROW];
    __shared__ __attribute__((aligned(16))) float wsh[COUT * CIN];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    for (int i = t; i < COUT * CIN; i += 256) wsh[i] = w[i];
    __syncthreads();
    char* mine = rows[wv];
    const long long total = (long long)N * V;
    for (long long v0 = ((long long)blockIdx.x * 4 + wv) * 64; v0 < total; v0 += (long long)gridDim.x * 256) {
        const long long left = total - v0;  // (>= 1; the last round may hold fewer than 64 voxels)
        f32x4 it[QPV];
#pragma unroll
        for (int i = 0; i < QPV; ++i) {
            const int item = i * 64 + lane, vox = item / QPV;
            it[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (vox < left) it[i] = *reinterpret_cast<const f32x4*>(x + (size_t)v0 * CIN + (size_t)item * 4);
        }
#pragma unroll
        for (int i = 0; i < QPV; ++i) {
            const int item = i * 64 + lane, vox = item / QPV, qc = item - vox * QPV;
            *reinterpret_cast<f32x4*>(mine + vox * ROW + qc * 16) = it[i];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float xs[CIN];
#pragma unroll
        for (int j = 0; j < QPV; ++j) {
            const f32x4 r = *reinterpret_cast<const f32x4*>(mine + lane * ROW + j * 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) xs[4 * j + e] = r[e];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float acc[COUT];
#pragma unroll
        for (int o = 0; o < COUT; ++o) {
            asm volatile("" ::: "memory");  // (keeps the weight reads inside the round)
            float pr[G];
#pragma unroll
            for (int s_ = 0; s_ < G; ++s_) {
                const f32x4 wr = *reinterpret_cast<const f32x4*>(wsh + o * CIN + 4 * s_);
                pr[s_] = head_dot4(xs[4 * s_], xs[4 * s_ + 1], xs[4 * s_ + 2], xs[4 * s_ + 3], wr);
            }
#pragma unroll
            for (int m = G >> 1; m > 0; m >>= 1)
#pragma unroll
                for (int s_ = 0; s_ < m; ++s_) pr[s_] = pr[s_] + pr[s_ + m];
            acc[o] = pr[0] + b[o];
        }
        if (lane < left) {
            const long long idx = v0 + lane;
            const int n = (int)(idx / V);
            const long long v = idx - (long long)n * V;
            float mx = -INFINITY, den = 0.f;
            if (act == 2) {
#pragma unroll
                for (int o = 0; o < COUT; ++o) mx = fmaxf(mx, acc[o]);
#pragma unroll
                for (int o = 0; o < COUT; ++o) den += expf(acc[o] - mx);
            }
#pragma unroll
            for (int o = 0; o < COUT; ++o) {
                const size_t oi = ((size_t)n * COUT + o) * V + v;
                logits[oi] = acc[o];
                if (probs) {
                    float p_ = acc[o];
                    if (act == 1) p_ = 1.f / (1.f + expf(-acc[o]));
                    if (act == 2) p_ = expf(acc[o] - mx) / den;
                    probs[oi] = p_;
                }
            }
        }
    }
}
template <int CIN, int COUT>
static void launch_head_rows_f32(const float* x, const float* w, const float* b, int N, long long V, int act, float* logits, float* probs,
                                 hipStream_t st) {
    long long rounds = ((long long)N * V + 255) / 256;
    if (rounds > 4096) rounds = 4096;
    hipLaunchKernelGGL((head_fwd_f32_rows_kernel<CIN, COUT>), dim3((unsigned)rounds), dim3(256), 0, st, x, w, b, N, V, act, logits, probs);
}

extern "C" int u3d_conv1x1_head_fwd(int device, u3d_stream_t stream, const float* x, const float* w, const float* b,
                                    int N, int64_t V, int Cin, int Cout, int act, float* logits, float* probs) {
    U3D_ENTER(device);
    U3D_REQUIRE(x && w && b && logits && N > 0 && V > 0, "u3d_conv1x1_head_fwd: bad argument");
    U3D_REQUIRE(Cout >= 1 && Cout <= HEAD_WIDE_MAXCO && Cin >= 1 && Cin <= HEAD_MAXCI,
                "u3d_conv1x1_head_fwd: supports Cout<=%d, Cin<=%d (got %d,%d)", HEAD_WIDE_MAXCO, HEAD_MAXCI, Cout, Cin);
    if (Cout > HEAD_MAXCO) {
        const long long tot_ = (long long)N * V;
        hipLaunchKernelGGL(head_fwd_wide_kernel, dim3(grid_for(tot_, 2048), (unsigned)((Cout + HEAD_MAXCO - 1) / HEAD_MAXCO)), dim3(256), 0,
                           (hipStream_t)stream, x, w, b, N, (long long)V, Cin, Cout, act, logits, probs);
        U3D_LAUNCH_CHECK();
        if (probs && act == 2) {
            hipLaunchKernelGGL(head_softmax_wide_kernel, dim3(grid_for(tot_, 4096)), dim3(256), 0, (hipStream_t)stream, logits, N,
                               (long long)V, Cout, probs);
            U3D_LAUNCH_CHECK();
        }
        return 0;
    }
    const int G = Cin / 4;
    const bool vec = (Cin % 4 == 0) && (G & (G - 1)) == 0 && G >= 1 && G <= 64 && (((uintptr_t)x | (uintptr_t)w) & 15) == 0;
    hipStream_t st = (hipStream_t)stream;
    const long long tot = (long long)N * V;
    // the segmentation heads of the shipped configurations (32 or 64 channels -> 1 or 2 outputs): one voxel per lane through LDS rows
    // (key 21 = 1: the vectorised kernel, A/B; results are bit-identical)
    if (vec && (Cin == 32 || Cin == 64) && Cout <= 2 && g_u3d_tune[21] != 1) {
        if (Cin == 64 && Cout == 1) launch_head_rows_f32<64, 1>(x, w, b, N, (long long)V, act, logits, probs, st);
        else if (Cin == 64) launch_head_rows_f32<64, 2>(x, w, b, N, (long long)V, act, logits, probs, st);
        else if (Cout == 1) launch_head_rows_f32<32, 1>(x, w, b, N, (long long)V, act, logits, probs, st);
        else launch_head_rows_f32<32, 2>(x, w, b, N, (long long)V, act, logits, probs, st);
        U3D_LAUNCH_CHECK();
        return 0;
    }
#define U3D_HEAD_FWD(GG)                                                                                            \
    hipLaunchKernelGGL((head_fwd_vec_kernel<GG, float>), dim3(grid_for(tot * GG, 8192)), dim3(256), 0, st, x, w, b, N, \
                       (long long)V, Cin, Cout, act, logits, probs)
    if (vec && G == 1) U3D_HEAD_FWD(1);
    else if (vec && G == 2) U3D_HEAD_FWD(2);
    else if (vec && G == 4) U3D_HEAD_FWD(4);
    else if (vec && G == 8) U3D_HEAD_FWD(8);
    else if (vec && G == 16) U3D_HEAD_FWD(16);
    else if (vec && G == 32) U3D_HEAD_FWD(32);
    else if (vec && G == 64) U3D_HEAD_FWD(64);
    else
        hipLaunchKernelGGL(head_fwd_kernel, dim3(grid_for(tot, 4096)), dim3(256), 0, st, x, w, b, N, (long long)V, Cin,
                           Cout, act, logits, probs);
#undef U3D_HEAD_FWD
    U3D_LAUNCH_CHECK();
    return 0;
}

// dx[n,v,c] = sum_o dlogits[n,o,v] * w[o,c]  (masked by x>0)
__global__ __launch_bounds__(256) void head_bwd_dx_kernel(const float* __restrict__ dl, const float* __restrict__ x,
                                                          const float* __restrict__ w, int N, long long V, int Cin,
                                                          int Cout, int relu_mask, float* __restrict__ dx) {
    __shared__ float ws[HEAD_MAXCO * HEAD_MAXCI];
    for (int i = threadIdx.x; i < Cout * Cin; i += blockDim.x) ws[i] = w[i];
    __syncthreads();
    // thread -> (voxel, channel) with channel fastest: coalesced dx/x, broadcast dlogits
    const long long total = (long long)N * V * Cin;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % Cin);
        const long long nv = idx / Cin;
        const int n = (int)(nv / V);
        const long long v = nv - (long long)n * V;
        float s = 0.f;
        for (int o = 0; o < Cout; ++o) s = fmaf(dl[((size_t)n * Cout + o) * V + v], ws[o * Cin + c], s);
        if (relu_mask && !(x[idx] > 0.f)) s = 0.f;
        dx[idx] = s;
    }
}

// acc[o*Cin + c] += sum_v dl[o,v] * x[v,c];  acc[Cout*Cin + o] += sum_v dl[o,v]
__global__ __launch_bounds__(256) void head_bwd_dw_kernel(const float* __restrict__ dl, const float* __restrict__ x,
                                                          int N, long long V, int Cin, int Cout,
                                                          double* __restrict__ acc) {
    extern __shared__ float red[];  // [rows][Cin][HEAD_MAXCO+... ] reduced via atomics in LDS instead
    const int t = threadIdx.x;
    const int rows = 256 / Cin;
    const int row = t / Cin, c = t - row * Cin;
    float a[HEAD_MAXCO];
    float bsum[HEAD_MAXCO];
#pragma unroll
    for (int o = 0; o < HEAD_MAXCO; ++o) {
        a[o] = 0.f;
        bsum[o] = 0.f;
    }
    const long long total = (long long)N * V;
    const long long per = (total + gridDim.x - 1) / gridDim.x;
    const long long beg = (long long)blockIdx.x * per, end = min(total, beg + per);
    if (row < rows) {
        for (long long nv = beg + row; nv < end; nv += rows) {
            const int n = (int)(nv / V);
            const long long v = nv - (long long)n * V;
            const float xv = x[(size_t)nv * Cin + c];
#pragma unroll
            for (int o = 0; o < HEAD_MAXCO; ++o) {
                if (o < Cout) {
                    const float d = dl[((size_t)n * Cout + o) * V + v];
                    a[o] = fmaf(d, xv, a[o]);
                    if (c == 0) bsum[o] += d;
                }
            }
        }
    }
    // reduce the rows through LDS in a FIXED order (red[row][(Cout+1)*Cin]; no floating-point atomics: the block's
    // partial is bit-reproducible, the cross-block sum is carried in f64)
    const int L = Cout * Cin + Cout;  // a row = [Cout x Cin | Cout biases] (round 5: (Cout + 1) * Cin overlapped the next row when Cout > Cin)
    if (row < rows) {
#pragma unroll
        for (int o = 0; o < HEAD_MAXCO; ++o) {
            if (o < Cout) {
                red[row * L + o * Cin + c] = a[o];
                if (c == 0) red[row * L + Cout * Cin + o] = bsum[o];
            }
        }
    }
    __syncthreads();
    for (int i = t; i < Cout * Cin + Cout; i += 256) {
        double sum = 0.0;
        for (int r = 0; r < rows; ++r) sum += (double)red[r * L + i];
        u3d_atomic_add_f64(&acc[i], sum);
    }
}

// Fused vectorised backward: thread -> (voxel, channel quad) with a FIXED quad per thread (grid stride is a
// multiple of Q), so dw partials stay in registers; x is read once for both dx (ReLU mask) and dw.
constexpr int HEAD_VEC_MAXCO = 4;  // the vector kernels keep per-output partials in registers
template <typename T = float>
__global__ __launch_bounds__(256) void head_bwd_vec_kernel(const float* __restrict__ dl, const T* __restrict__ x,
                                                           const float* __restrict__ w, int N, long long V, int Cin,
                                                           int Cout, int relu_mask, T* __restrict__ dx,
                                                           double* __restrict__ acc, int reps) {
    extern __shared__ float red[];  // [(Cout+1)][Cin]
    const int t = threadIdx.x;
    const int Q = Cin >> 2;
    const int rows = 256 / Q;  // voxels per block iteration
    const int q = t % Q, row = t / Q;
    f32x4 wv[HEAD_VEC_MAXCO];
    f32x4 aw[HEAD_VEC_MAXCO];
    float ab[HEAD_VEC_MAXCO];
#pragma unroll
    for (int o = 0; o < HEAD_VEC_MAXCO; ++o) {
        wv[o] = f32x4{0.f, 0.f, 0.f, 0.f};
        aw[o] = f32x4{0.f, 0.f, 0.f, 0.f};
        ab[o] = 0.f;
        if (o < Cout) wv[o] = *reinterpret_cast<const f32x4*>(w + (size_t)o * Cin + 4 * q);
    }
    if (row < rows) {
        // per sample, two voxels per iteration (both loads in flight before the first use); no 64-bit division
        const long long stride = (long long)gridDim.x * rows;
        for (int n = 0; n < N; ++n) {
            const T* xn = x + (size_t)n * V * Cin + 4 * q;
            const float* dn = dl + (size_t)n * Cout * V;
            T* dxn = dx ? dx + (size_t)n * V * Cin + 4 * q : nullptr;
            for (long long v = (long long)blockIdx.x * rows + row; v < V; v += 2 * stride) {
                const long long v2 = v + stride;
                const bool two = v2 < V;
                const long long vb = two ? v2 : v;
                const f32x4 xa = u3d_ldq(xn + (size_t)v * Cin);
                const f32x4 xb = u3d_ldq(xn + (size_t)vb * Cin);
                float da[HEAD_VEC_MAXCO], db[HEAD_VEC_MAXCO];
#pragma unroll
                for (int o = 0; o < HEAD_VEC_MAXCO; ++o) {
                    da[o] = o < Cout ? dn[(size_t)o * V + v] : 0.f;
                    db[o] = (o < Cout && two) ? dn[(size_t)o * V + vb] : 0.f;
                }
                f32x4 sa = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int o = 0; o < HEAD_VEC_MAXCO; ++o) {
                    if (o < Cout) {
                        sa += da[o] * wv[o];
                        sb += db[o] * wv[o];
                        aw[o] += da[o] * xa + db[o] * xb;
                        if (q == 0) ab[o] += da[o] + db[o];
                    }
                }
                if (relu_mask) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        sa[e] = xa[e] > 0.f ? sa[e] : 0.f;
                        sb[e] = xb[e] > 0.f ? sb[e] : 0.f;
                    }
                }
                if (dxn) {
                    u3d_stq(dxn + (size_t)v * Cin, sa);
                    if (two) u3d_stq(dxn + (size_t)vb * Cin, sb);
                }
            }
        }
    }
    if (acc == nullptr) return;
    // fixed-order reduction of the rows (red[row][(Cout+1)*Cin]), no floating-point atomics inside the block
    const int L = (Cout + 1) * Cin;
    if (row < rows) {
#pragma unroll
        for (int o = 0; o < HEAD_VEC_MAXCO; ++o) {
            if (o < Cout) {
#pragma unroll
                for (int e = 0; e < 4; ++e) red[row * L + o * Cin + 4 * q + e] = aw[o][e];
                if (q == 0) red[row * L + Cout * Cin + o] = ab[o];
            }
        }
    }
    __syncthreads();
    for (int i = t; i < Cout * Cin + Cout; i += 256) {
        double sum = 0.0;
        for (int r = 0; r < rows; ++r) sum += (double)red[r * L + i];
        // (replica row blockIdx.x % reps of acc[reps][Cout * Cin + Cout], u3d_conv1x1_head_bwd_reps: 2048 blocks on the same 33 doubles)
        u3d_atomic_add_f64(&acc[(size_t)(blockIdx.x % (unsigned)reps) * (Cout * Cin + Cout) + i], sum);
    }
}

extern "C" int u3d_conv1x1_head_bwd_b16(int device, u3d_stream_t stream, const float* dlogits, const void* x, const float* w, int N,
                                        int64_t V, int Cin, int Cout, int relu_mask, void* dx, double* acc) {
    U3D_ENTER(device);
    U3D_REQUIRE(dlogits && x && w && N > 0 && V > 0 && Cin % 4 == 0 && Cin <= 256 && Cout >= 1 && Cout <= HEAD_VEC_MAXCO &&
                    (((uintptr_t)x | (uintptr_t)dx) & 7) == 0 && ((uintptr_t)w & 15) == 0,
                "u3d_conv1x1_head_bwd_b16: needs Cin %% 4 == 0, Cin <= 256, Cout <= %d (got %d, %d)", HEAD_VEC_MAXCO, Cin, Cout);
    const int rows = 256 / (Cin / 4);
    long long blocks = cdivll((long long)V, (long long)rows * 32);
    // (one accumulator row: every block ends with Cout * (Cin + 1) same-address f64 atomics of 19.5 ns — config 4's head by block count,
    // rocprofv3: 2048: 119 us, 1024: 105, 768: 109, 512: 141; the fp32 entry point spreads them over replica rows instead)
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(head_bwd_vec_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), (size_t)rows * (Cout + 1) * Cin * sizeof(float),
                       (hipStream_t)stream, dlogits, (const __bf16*)x, w, N, (long long)V, Cin, Cout, relu_mask, (__bf16*)dx, acc, 1);
    U3D_LAUNCH_CHECK();
    return 0;
}

// ---- wide heads: dx reads the (cached) weights from global memory instead of a Cout x Cin LDS copy; dw / db run per tile of
// HEAD_MAXCO outputs (grid.y) with the same fixed-order block reduction and f64 accumulation as the narrow kernel
__global__ __launch_bounds__(256) void head_bwd_dx_wide_kernel(const float* __restrict__ dl, const float* __restrict__ x,
                                                               const float* __restrict__ w, int N, long long V, int Cin, int Cout,
                                                               int relu_mask, float* __restrict__ dx) {
    const long long total = (long long)N * V * Cin;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % Cin);
        const long long nv = idx / Cin;
        const int n = (int)(nv / V);
        const long long v = nv - (long long)n * V;
        float s_ = 0.f;
        for (int o = 0; o < Cout; ++o) s_ = fmaf(dl[((size_t)n * Cout + o) * V + v], w[(size_t)o * Cin + c], s_);
        if (relu_mask && !(x[idx] > 0.f)) s_ = 0.f;
        dx[idx] = s_;
    }
}

__global__ __launch_bounds__(256) void head_bwd_dw_wide_kernel(const float* __restrict__ dl, const float* __restrict__ x, int N, long long V,
                                                               int Cin, int Cout, double* __restrict__ acc) {
    extern __shared__ float red[];
    const int t = threadIdx.x;
    const int o0 = blockIdx.y * HEAD_MAXCO, no = min(HEAD_MAXCO, Cout - o0);
    const int rows = 256 / Cin;
    const int row = t / Cin, c = t - row * Cin;
    float a[HEAD_MAXCO], bsum[HEAD_MAXCO];
#pragma unroll
    for (int o = 0; o < HEAD_MAXCO; ++o) a[o] = bsum[o] = 0.f;
    const long long total = (long long)N * V;
    const long long per = (total + gridDim.x - 1) / gridDim.x;
    const long long beg = (long long)blockIdx.x * per, end = min(total, beg + per);
    if (row < rows) {
        for (long long nv = beg + row; nv < end; nv += rows) {
            const int n = (int)(nv / V);
            const long long v = nv - (long long)n * V;
            const float xv = x[(size_t)nv * Cin + c];
#pragma unroll
            for (int o = 0; o < HEAD_MAXCO; ++o) {
                if (o < no) {
                    const float d = dl[((size_t)n * Cout + o0 + o) * V + v];
                    a[o] = fmaf(d, xv, a[o]);
                    if (c == 0) bsum[o] += d;
                }
            }
        }
    }
    const int L = no * Cin + no;  // a row = [no x Cin weight partials | no bias partials]  (no may exceed Cin)
    if (row < rows) {
#pragma unroll
        for (int o = 0; o < HEAD_MAXCO; ++o) {
            if (o < no) {
                red[row * L + o * Cin + c] = a[o];
                if (c == 0) red[row * L + no * Cin + o] = bsum[o];
            }
        }
    }
    __syncthreads();
    for (int i = t; i < no * Cin + no; i += 256) {
        double sum = 0.0;
        for (int r = 0; r < rows; ++r) sum += (double)red[r * L + i];
        // acc = [Cout x Cin weights | Cout biases]
        double* dst = i < no * Cin ? &acc[(size_t)o0 * Cin + i] : &acc[(size_t)Cout * Cin + o0 + (i - no * Cin)];
        u3d_atomic_add_f64(dst, sum);
    }
}

static int head_bwd_impl(int device, u3d_stream_t stream, const float* dlogits, const float* x, const float* w, int N, int64_t V, int Cin,
                         int Cout, int relu_mask, float* dx, double* acc, int reps);

extern "C" int u3d_conv1x1_head_bwd(int device, u3d_stream_t stream, const float* dlogits, const float* x,
                                    const float* w, int N, int64_t V, int Cin, int Cout, int relu_mask, float* dx,
                                    double* acc) {
    return head_bwd_impl(device, stream, dlogits, x, w, N, V, Cin, Cout, relu_mask, dx, acc, 1);
}

// ... with acc as `reps` replica rows [reps][Cout * Cin + Cout] (zeroed by the caller; the vectorised kernel's block b adds to row
// b % reps, every other variant to row 0); u3d_cvt_f64_f32_sum folds the rows into the float gradient
extern "C" int u3d_conv1x1_head_bwd_reps(int device, u3d_stream_t stream, const float* dlogits, const float* x, const float* w, int N,
                                         int64_t V, int Cin, int Cout, int relu_mask, float* dx, double* acc, int reps) {
    U3D_REQUIRE(reps >= 1 && reps <= 64, "u3d_conv1x1_head_bwd_reps: reps must be 1 .. 64");
    return head_bwd_impl(device, stream, dlogits, x, w, N, V, Cin, Cout, relu_mask, dx, acc, reps);
}

static int head_bwd_impl(int device, u3d_stream_t stream, const float* dlogits, const float* x, const float* w, int N, int64_t V, int Cin,
                         int Cout, int relu_mask, float* dx, double* acc, int reps) {
    U3D_ENTER(device);
    U3D_REQUIRE(dlogits && x && w && N > 0 && V > 0, "u3d_conv1x1_head_bwd: bad argument");
    U3D_REQUIRE(Cout >= 1 && Cout <= HEAD_WIDE_MAXCO && Cin >= 1 && Cin <= HEAD_MAXCI,
                "u3d_conv1x1_head_bwd: supports Cout<=%d, Cin<=%d (got %d,%d)", HEAD_WIDE_MAXCO, HEAD_MAXCI, Cout, Cin);
    if (Cout > HEAD_MAXCO) {
        if (dx) {
            hipLaunchKernelGGL(head_bwd_dx_wide_kernel, dim3(grid_for((long long)N * V * Cin, 16384)), dim3(256), 0, (hipStream_t)stream,
                               dlogits, x, w, N, (long long)V, Cin, Cout, relu_mask, dx);
            U3D_LAUNCH_CHECK();
        }
        if (acc) {
            long long blocks = cdivll((long long)N * V, 4096);
            if (blocks > 512) blocks = 512;
            if (blocks < 1) blocks = 1;
            const size_t shmem = (size_t)(256 / Cin) * (HEAD_MAXCO * Cin + HEAD_MAXCO) * sizeof(float);
            hipLaunchKernelGGL(head_bwd_dw_wide_kernel, dim3((unsigned)blocks, (unsigned)((Cout + HEAD_MAXCO - 1) / HEAD_MAXCO)), dim3(256),
                               shmem, (hipStream_t)stream, dlogits, x, N, (long long)V, Cin, Cout, acc);
            U3D_LAUNCH_CHECK();
        }
        return 0;
    }
    const bool vec = (Cin % 4 == 0) && Cin <= 1024 && Cout <= HEAD_VEC_MAXCO &&
                     (((uintptr_t)x | (uintptr_t)w | (uintptr_t)dx) & 15) == 0;
    if (vec) {
        const int rows = 256 / (Cin / 4);
        long long blocks = cdivll((long long)V, (long long)rows * 32);
        if (blocks > 2048) blocks = 2048;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(head_bwd_vec_kernel<float>, dim3((unsigned)blocks), dim3(256), (size_t)rows * (Cout + 1) * Cin * sizeof(float),
                           (hipStream_t)stream, dlogits, x, w, N, (long long)V, Cin, Cout, relu_mask, dx, acc, reps);
        U3D_LAUNCH_CHECK();
        return 0;
    }
    if (dx) {
        hipLaunchKernelGGL(head_bwd_dx_kernel, dim3(grid_for((long long)N * V * Cin, 16384)), dim3(256), 0,
                           (hipStream_t)stream, dlogits, x, w, N, (long long)V, Cin, Cout, relu_mask, dx);
        U3D_LAUNCH_CHECK();
    }
    if (acc) {
        long long blocks = cdivll((long long)N * V, 4096);
        if (blocks > 1024) blocks = 1024;
        if (blocks < 1) blocks = 1;
        const size_t shmem = (size_t)(256 / Cin) * (Cout * Cin + Cout) * sizeof(float);
        hipLaunchKernelGGL(head_bwd_dw_kernel, dim3((unsigned)blocks), dim3(256), shmem, (hipStream_t)stream, dlogits,
                           x, N, (long long)V, Cin, Cout, acc);
        U3D_LAUNCH_CHECK();
    }
    return 0;
}

__global__ void cvt_f64_f32_kernel(const double* __restrict__ s, float* __restrict__ d, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        d[i] = (float)s[i];
}

__global__ void cvt_f64_f32_sum_kernel(const double* __restrict__ s, float* __restrict__ d, long long n, int reps) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        d[i] = (float)u3d_sum_replicas(s + i, (size_t)n, reps);
}

// dst[i] = float(src[0][i] + src[1][i] + .. + src[reps - 1][i]) (ascending): the fold of a replicated f64 accumulator
extern "C" int u3d_cvt_f64_f32_sum(int device, u3d_stream_t stream, const double* src, float* dst, int64_t n, int reps) {
    U3D_ENTER(device);
    U3D_REQUIRE(src && dst && n > 0 && reps >= 1 && reps <= 64, "u3d_cvt_f64_f32_sum: bad argument");
    hipLaunchKernelGGL(cvt_f64_f32_sum_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, (hipStream_t)stream, src, dst, (long long)n, reps);
    U3D_LAUNCH_CHECK();
    return 0;
}

extern "C" int u3d_cvt_f64_f32(int device, u3d_stream_t stream, const double* src, float* dst, int64_t n) {
    U3D_ENTER(device);
    U3D_REQUIRE(src && dst && n > 0, "u3d_cvt_f64_f32: bad argument");
    hipLaunchKernelGGL(cvt_f64_f32_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, (hipStream_t)stream, src, dst,
                       (long long)n);
    U3D_LAUNCH_CHECK();
    return 0;
}

// =================================================================================================
// layout transposes through a 32x32 LDS tile: src (N, R, S) -> dst (N, S, R)
__global__ void transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, long long R, long long S) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const long long r0 = (long long)blockIdx.y * 32, s0 = (long long)blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const float* sp = src + (size_t)n * R * S;
    float* dp = dst + (size_t)n * R * S;
    for (int k = ty; k < 32; k += 8)
        if (r0 + k < R && s0 + tx < S) tile[k][tx] = sp[(size_t)(r0 + k) * S + s0 + tx];
    __syncthreads();
    for (int k = ty; k < 32; k += 8)
        if (s0 + k < S && r0 + tx < R) dp[(size_t)(s0 + k) * R + r0 + tx] = tile[tx][k];
}

static int launch_transpose(u3d_stream_t stream, const float* src, float* dst, int N, long long R, long long S) {
    const long long gx = cdivll(S, 32), gy = cdivll(R, 32);
    // grid.y is limited to 65535: the long (voxel) dimension must be S
    U3D_REQUIRE(gy <= 65535, "transpose: row dimension too large");
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)gx, (unsigned)gy, (unsigned)N), dim3(256), 0,
                       (hipStream_t)stream, src, dst, R, S);
    U3D_LAUNCH_CHECK();
    return 0;
}

extern "C" int u3d_ncdhw_to_ndhwc(int device, u3d_stream_t stream, const float* src, float* dst, int N, int C,
                                  int64_t V) {
    U3D_ENTER(device);
    U3D_REQUIRE(src && dst && N > 0 && C > 0 && V > 0, "u3d_ncdhw_to_ndhwc: bad argument");
    return launch_transpose(stream, src, dst, N, C, V);  // (N,C,V) -> (N,V,C)
}

// (N,V,C) -> (N,C,V): rows = V can exceed 65535*32, so run the transposed problem with swapped roles
__global__ void transpose_vc_kernel(const float* __restrict__ src, float* __restrict__ dst, long long V, int C) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const long long v0 = (long long)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* sp = src + (size_t)n * V * C;
    float* dp = dst + (size_t)n * V * C;
    for (int k = ty; k < 32; k += 8)
        if (v0 + k < V && c0 + tx < C) tile[k][tx] = sp[(size_t)(v0 + k) * C + c0 + tx];
    __syncthreads();
    for (int k = ty; k < 32; k += 8)
        if (c0 + k < C && v0 + tx < V) dp[(size_t)(c0 + k) * V + v0 + tx] = tile[tx][k];
}

extern "C" int u3d_ndhwc_to_ncdhw(int device, u3d_stream_t stream, const float* src, float* dst, int N, int C,
                                  int64_t V) {
    U3D_ENTER(device);
    U3D_REQUIRE(src && dst && N > 0 && C > 0 && V > 0, "u3d_ndhwc_to_ncdhw: bad argument");
    hipLaunchKernelGGL(transpose_vc_kernel, dim3((unsigned)cdivll(V, 32), (unsigned)cdivll(C, 32), (unsigned)N),
                       dim3(256), 0, (hipStream_t)stream, src, dst, (long long)V, C);
    U3D_LAUNCH_CHECK();
    return 0;
}
