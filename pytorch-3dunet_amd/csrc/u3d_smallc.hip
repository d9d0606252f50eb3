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

extern int g_u3d_tune[24];  // u3d_set_tuning (csrc/u3d_conv.hip); key 13 = 1: first-layer forward on the direct (non-MFMA) kernel, for A/B

typedef float f32x4s __attribute__((ext_vector_type(4)));

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
    double* out_stats;    // optional [N][Cout][2] += (sum, sum of squares) of the written values
    int N, D, H, W, Cin, Cout, relu;
    int tz, ty, tx, B;
    int reps;             // out_stats has this many replica rows [reps][N][Cout][2]; block b adds to row b % reps (u3d_conv3d_small_cin_fwd_reps)
};

// grid (B, N): a block walks tiles b, b+B, ... of sample n (weights staged in LDS once per block; the statistics of the
// written values — the next GroupNorm's input — are carried in registers over all its tiles and leave the block as ONE
// f64 atomic per channel, so the 16-channel output is not re-read by a separate statistics pass)
template <int COUTP>
__global__ __launch_bounds__(256) void conv3d_small_fwd_kernel(const SmallFwdParams p) {
    using namespace sc;
    __shared__ __attribute__((aligned(16))) float xs[HV * MAXC];
    __shared__ __attribute__((aligned(16))) float ws[27 * MAXC * COUTP];
    __shared__ double sred[COUTP][2];
    const int t = threadIdx.x;
    const int n = blockIdx.y;
    const int Cin = p.Cin, D = p.D, H = p.H, W = p.W;
    // weights -> LDS as [tap][c][k] (k padded to COUTP with zeros)
    for (int i = t; i < 27 * Cin * COUTP; i += 256) {
        const int k = i % COUTP;
        const int r = i / COUTP;
        const int c = r % Cin, tap = r / Cin;
        ws[i] = k < p.Cout ? p.w[((size_t)k * Cin + c) * 27 + tap] : 0.f;
    }
    if (t < COUTP) sred[t][0] = sred[t][1] = 0.0;
    float s1[COUTP], s2[COUTP];
#pragma unroll
    for (int k = 0; k < COUTP; ++k) s1[k] = s2[k] = 0.f;
    const int zl = t >> 6, yl = (t >> 3) & 7, xl = t & 7;
    const int ntiles = p.tz * p.ty * p.tx;
    for (int tile = blockIdx.x; tile < ntiles; tile += p.B) {
        int tt = tile;
        const int txi = tt % p.tx;
        tt /= p.tx;
        const int tyi = tt % p.ty;
        const int tzi = tt / p.ty;
        const int z0 = tzi * TZ, y0 = tyi * TY, x0 = txi * TX;
        __syncthreads();  // the previous tile's reads of xs are done (and ws / sred are initialised)
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
#pragma unroll
            for (int k = 0; k < COUTP; ++k) {
                if (p.relu) acc[k] = fmaxf(acc[k], 0.f);
                s1[k] += acc[k];  // padded channels (k >= Cout) are exactly 0
                s2[k] += acc[k] * acc[k];
            }
            if (p.Cout % 4 == 0) {
#pragma unroll
                for (int k4 = 0; k4 < COUTP / 4; ++k4)
                    if (4 * k4 < p.Cout)
                        *reinterpret_cast<f32x4*>(o + 4 * k4) = f32x4{acc[4 * k4], acc[4 * k4 + 1], acc[4 * k4 + 2], acc[4 * k4 + 3]};
            } else {
#pragma unroll
                for (int k = 0; k < COUTP; ++k)
                    if (k < p.Cout) o[k] = acc[k];
            }
        }
    }
    if (p.out_stats) {
        // wave butterflies, then the four waves through LDS (f64), one global f64 atomic per channel and block
#pragma unroll
        for (int k = 0; k < COUTP; ++k) {
            float a = s1[k], b2 = s2[k];
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) {
                a += __shfl_xor(a, m);
                b2 += __shfl_xor(b2, m);
            }
            if ((t & 63) == 0) {
                __hip_atomic_fetch_add(&sred[k][0], (double)a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(&sred[k][1], (double)b2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        __syncthreads();
        if (t < p.Cout) {
            double* dst = p.out_stats + ((size_t)(blockIdx.x % (unsigned)p.reps) * p.N * p.Cout + (size_t)n * p.Cout + t) * 2;
            u3d_atomic_add_f64(dst, sred[t][0]);
            u3d_atomic_add_f64(dst + 1, sred[t][1]);
        }
    }
}

// ---- forward on the matrix pipe (round 4) ---------------------------------------------------------------------------------------
// The direct kernel above spends its time in LDS broadcast reads of the weights (4 x ds_read_b128 + 1 x ds_read_b32 per 16 FMAs and
// thread): 0.136 ms per step of the bench workload for a layer that moves 142 MB (1 TB/s).  With K = 27*Cin padded to a multiple of
// 4 the layer is a GEMM  out[k][v] = sum_kk W[k][kk] * xs[v + tap(kk)][c(kk)]  on v_mfma_f32_16x16x4_f32 with the WEIGHTS as the A
// operand (rows = output channels: KS registers per lane, loaded once per block) and the halo tile as B (columns = 16 voxels, one
// ds_read_b32 per MFMA and lane).  The accumulator then holds, per lane, FOUR CONSECUTIVE output channels (rows 4*(lane>>4) .. +3)
// of ONE voxel (column lane&15): the 16-channel record of a voxel leaves as four 16-byte lane stores, 8 voxels of a row = 512
// contiguous bytes, no transposition.  Wave w owns z-plane w of the 4x8x8 tile = 4 M-tiles of 2 rows x 8 voxels.
template <int CIN>
__global__ __launch_bounds__(256) void conv3d_small_fwd_mfma_kernel(const SmallFwdParams p) {
    using namespace sc;
    constexpr int K = 27 * CIN;
    constexpr int KS = (K + 3) / 4;  // k-steps of 4
    constexpr int NH = (HV * CIN + 255) / 256;  // halo elements per thread
    __shared__ __attribute__((aligned(16))) float xsb[2][HV * CIN];
    __shared__ double sred[16][2];
    const int t = threadIdx.x, l = t & 63, w = t >> 6;
    const int j = l & 15, kq = l >> 4;  // B column (voxel of the M-tile) / A row (output channel); k index within a step
    const int n = blockIdx.y;
    const int D = p.D, H = p.H, W = p.W;
    // A fragments: W[k = j][kk = 4*s + kq], kk = tap*CIN + c
    float wa[KS];
    int boff[KS];  // LDS offset of this lane's B element of step s, relative to the voxel's halo origin
#pragma unroll
    for (int s_ = 0; s_ < KS; ++s_) {
        const int kk = 4 * s_ + kq;
        const int tap = kk / CIN, c = kk - tap * CIN;
        const bool ok = kk < K;
        wa[s_] = (ok && j < p.Cout) ? p.w[((size_t)j * CIN + c) * 27 + tap] : 0.f;
        boff[s_] = ok ? ((tap / 9) * (HY * HX) + ((tap / 3) % 3) * HX + tap % 3) * CIN + c : 0;  // dead k: any valid slot (A is 0)
    }
    if (t < 16) sred[t][0] = sred[t][1] = 0.0;
    f32x4s s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};  // this lane's channels 4*kq .. +3 over its voxels
    // voxel of column j in M-tile mt: z = w, y = 2*mt + (j >> 3), x = j & 7
    const int vbase = (w * HY + (j >> 3)) * HX + (j & 7);
    const int ntiles = p.tz * p.ty * p.tx;
    // this thread's halo elements (element i = hv * CIN + c): constant coordinates inside the tile, and the GroupNorm affine of its channel
    int hrel[NH];  // (hz * H + hy) * W + hx relative to the halo origin, in voxels; -1: beyond the 600 halo voxels
    int hco[NH];   // hz | hy << 8 | hx << 16 | c << 24
    float ha[NH], hb[NH];
#pragma unroll
    for (int it = 0; it < NH; ++it) {
        const int i = t + 256 * it;
        const int c = i % CIN, hv = i / CIN;
        const int hz = hv / (HY * HX), rem = hv - hz * (HY * HX), hy = rem / HX, hx = rem - hy * HX;
        hrel[it] = i < HV * CIN ? (hz * H + hy) * W + hx : -1;
        hco[it] = hz | (hy << 8) | (hx << 16) | (c << 24);
        ha[it] = p.affine ? p.affine[((size_t)n * CIN + c) * 2] : 1.f;
        hb[it] = p.affine ? p.affine[((size_t)n * CIN + c) * 2 + 1] : 0.f;
    }
    auto origin = [&](int tile, int& z0, int& y0, int& x0) {
        const int txi = tile % p.tx;
        const int tt = tile / p.tx;
        z0 = (tt / p.ty) * TZ, y0 = (tt % p.ty) * TY, x0 = txi * TX;
    };
    // raw halo values of a tile into registers (loads only: the affine and the LDS stores follow after the current tile's MFMAs, so the
    // memory round trip of tile t+1 runs under the arithmetic of tile t — round 6; fill -> barrier -> compute -> barrier per tile
    // left every load latency exposed: 93 us for a layer that moves 142 MB)
    auto halo_load = [&](int tile, float (&hv)[NH], unsigned& inside) {
        int z0, y0, x0;
        origin(tile, z0, y0, x0);
        const int base = ((n * D + z0 - 1) * H + y0 - 1) * W + x0 - 1;
        inside = 0;
#pragma unroll
        for (int it = 0; it < NH; ++it) {
            const int gz = z0 - 1 + (hco[it] & 255), gy = y0 - 1 + ((hco[it] >> 8) & 255), gx = x0 - 1 + ((hco[it] >> 16) & 255);
            const bool in = hrel[it] >= 0 && gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W;
            inside |= (in ? 1u : 0u) << it;
            hv[it] = in ? p.x[(size_t)(base + hrel[it]) * CIN + (hco[it] >> 24)] : 0.f;
        }
    };
    auto halo_store = [&](float* xs, const float (&hv)[NH], unsigned inside) {
#pragma unroll
        for (int it = 0; it < NH; ++it)
            if (t + 256 * it < HV * CIN) xs[t + 256 * it] = ((inside >> it) & 1u) ? hv[it] * ha[it] + hb[it] : 0.f;  // zero padding stays 0
    };
    float hv[NH];
    unsigned hin = 0;
    int cur = 0;
    if ((int)blockIdx.x < ntiles) {
        halo_load(blockIdx.x, hv, hin);
        halo_store(xsb[0], hv, hin);
    }
    __syncthreads();  // (also: sred initialised)
    for (int tile = blockIdx.x; tile < ntiles; tile += p.B) {
        int z0, y0, x0;
        origin(tile, z0, y0, x0);
        const float* xs = xsb[cur];
        const bool has_next = tile + p.B < ntiles;
        if (has_next) halo_load(tile + p.B, hv, hin);
        f32x4s acc[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = f32x4s{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const float b = xs[(vbase + 2 * mt * HX) * CIN + boff[s_]];
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s_], b, acc[mt], 0, 0, 0);
            }
        }
        const int z = z0 + w;
        const bool cok = 4 * kq < p.Cout;  // (Cout % 4 == 0 on this path)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int y = y0 + 2 * mt + (j >> 3), x = x0 + (j & 7);
            f32x4s v = acc[mt];
            if (p.relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if (cok && z < D && y < H && x < W) {
                *reinterpret_cast<f32x4s*>(p.out + ((size_t)((n * D + z) * H + y) * W + x) * p.Cout + 4 * kq) = v;
                s1 += v;
                s2 += v * v;
            }
        }
        if (has_next) halo_store(xsb[cur ^ 1], hv, hin);
        __syncthreads();  // the next tile's halo is complete; nobody reads the current buffer any more
        cur ^= 1;
    }
    if (p.out_stats) {
        // the 16 lanes j of a k-group hold 16 voxel columns of the same four channels: butterfly over j, then the four waves
        // through LDS (f64), one global f64 atomic per channel and block
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float a = s1[e], b2 = s2[e];
#pragma unroll
            for (int m = 8; m > 0; m >>= 1) {
                a += __shfl_xor(a, m);
                b2 += __shfl_xor(b2, m);
            }
            if (j == 0 && 4 * kq + e < p.Cout) {
                __hip_atomic_fetch_add(&sred[4 * kq + e][0], (double)a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(&sred[4 * kq + e][1], (double)b2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        __syncthreads();
        if (t < p.Cout) {
            double* dst = p.out_stats + ((size_t)(blockIdx.x % (unsigned)p.reps) * p.N * p.Cout + (size_t)n * p.Cout + t) * 2;
            u3d_atomic_add_f64(dst, sred[t][0]);
            u3d_atomic_add_f64(dst + 1, sred[t][1]);
        }
    }
}

static int small_cin_fwd_impl(int device, u3d_stream_t stream, const float* x, const float* affine, const float* w, float* out, int N,
                              int D, int H, int W, int Cin, int Cout, int relu, double* out_stats, int reps);

extern "C" int u3d_conv3d_small_cin_fwd(int device, u3d_stream_t stream, const float* x, const float* affine,
                                        const float* w, float* out, int N, int D, int H, int W, int Cin, int Cout,
                                        int relu, double* out_stats) {
    return small_cin_fwd_impl(device, stream, x, affine, w, out, N, D, H, W, Cin, Cout, relu, out_stats, 1);
}

// ... with out_stats as `reps` replica rows [reps][N][Cout][2] (zeroed by the caller; see u3d_conv3d_ex_reps)
extern "C" int u3d_conv3d_small_cin_fwd_reps(int device, u3d_stream_t stream, const float* x, const float* affine, const float* w,
                                             float* out, int N, int D, int H, int W, int Cin, int Cout, int relu, double* out_stats,
                                             int reps) {
    U3D_REQUIRE(reps >= 1 && reps <= 64, "u3d_conv3d_small_cin_fwd_reps: reps must be 1 .. 64");
    return small_cin_fwd_impl(device, stream, x, affine, w, out, N, D, H, W, Cin, Cout, relu, out_stats, reps);
}

static int small_cin_fwd_impl(int device, u3d_stream_t stream, const float* x, const float* affine, const float* w, float* out, int N,
                              int D, int H, int W, int Cin, int Cout, int relu, double* out_stats, int reps) {
    U3D_ENTER(device);
    U3D_REQUIRE(x && w && out && N > 0 && D > 0 && H > 0 && W > 0, "u3d_conv3d_small_cin_fwd: bad argument");
    U3D_REQUIRE(Cin >= 1 && Cin <= sc::MAXC && Cout >= 1 && Cout <= 32,
                "u3d_conv3d_small_cin_fwd: needs Cin<=4, Cout<=32 (got %d,%d)", Cin, Cout);
    SmallFwdParams p;
    p.x = x, p.affine = affine, p.w = w, p.out = out, p.out_stats = out_stats;
    p.N = N, p.D = D, p.H = H, p.W = W, p.Cin = Cin, p.Cout = Cout, p.relu = relu;
    p.reps = reps;
    p.tz = (D + sc::TZ - 1) / sc::TZ, p.ty = (H + sc::TY - 1) / sc::TY, p.tx = (W + sc::TX - 1) / sc::TX;
    const long long ntiles = (long long)p.tz * p.ty * p.tx;
    // 2 blocks per CU in total: every block ends with one f64 atomic pair per output channel on the SAME 2*Cout addresses per sample, and
    // same-address atomics retire at ~24 ns each — measured on the bench workload (tools/small_bench.py, round 6): 2048 blocks 86 us
    // (38 us without the statistics), 1024 blocks 62, 512 blocks 57, 4096 blocks 138 (key 19: A/B of the block count)
    long long B = (g_u3d_tune[19] > 0 ? g_u3d_tune[19] : 512) / N;
    if (B < 1) B = 1;
    if (B > ntiles) B = ntiles;
    p.B = (int)B;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)B, (unsigned)N);
    // <= 16 output channels in whole quads (the network's first layer: Cin -> f_maps/2): matrix pipe; key 13 = 1: the direct kernel
    const bool mfma = Cout <= 16 && Cout % 4 == 0 && ((uintptr_t)out & 15) == 0 && g_u3d_tune[13] != 1;
    if (mfma && Cin == 1)
        hipLaunchKernelGGL(conv3d_small_fwd_mfma_kernel<1>, grid, dim3(256), 0, st, p);
    else if (mfma && Cin == 2)
        hipLaunchKernelGGL(conv3d_small_fwd_mfma_kernel<2>, grid, dim3(256), 0, st, p);
    else if (mfma && Cin == 3)
        hipLaunchKernelGGL(conv3d_small_fwd_mfma_kernel<3>, grid, dim3(256), 0, st, p);
    else if (mfma && Cin == 4)
        hipLaunchKernelGGL(conv3d_small_fwd_mfma_kernel<4>, grid, dim3(256), 0, st, p);
    else if (Cout <= 8)
        hipLaunchKernelGGL(conv3d_small_fwd_kernel<8>, grid, dim3(256), 0, st, p);
    else if (Cout <= 16)
        hipLaunchKernelGGL(conv3d_small_fwd_kernel<16>, grid, dim3(256), 0, st, p);
    else
        hipLaunchKernelGGL(conv3d_small_fwd_kernel<32>, grid, dim3(256), 0, st, p);
    U3D_LAUNCH_CHECK();
    return 0;
}

// -------------------------------------------------------------------------------------------------------------
// backward: partial[n][b][(k*27 + tap)*(Cin+1) + c]  (c == Cin is the T slot).  grid (B, N); a block walks tiles
// b, b+B, ... of sample n.  The contraction  P[k][(tap,c)] = sum_u dz[u,k] * xs[u + tap][c]  (xs = raw x with an
// in-bounds indicator as channel Cin, zero outside the volume) is a GEMM with M = Cout, N = 27*(Cin+1), K = voxels: it
// runs on v_mfma_f32_16x16x4_f32 (A[i = k][kk = voxel] straight from global dz, B[kk = voxel][j = column] from the LDS
// halo tile).  Wave w owns z-plane w of the 4x8x8 tile = 16 k-steps of 4 consecutive x; the four waves' sums are folded in a
// fixed order at the end of the block, the finalize kernel adds the blocks' partials in a fixed order.
struct SmallBwdParams {
    const float* x;   // raw input (N,D,H,W,Cin)
    const float* dz;  // (N,D,H,W,Cout)
    float* partial;
    int N, D, H, W, Cin, Cout;
    int tz, ty, tx, B;
};


// VA (round 6; RT = 1: two 16 KB dz buffers fit the static LDS limit): Cout == 16, every tile inside the volume, dz 16-byte aligned — the dz tile goes through LDS with 16-byte loads at
// per-thread CONSTANT offsets from a scalar tile base (the first form fetched every A operand as a 4-byte global load with its own
// coordinate arithmetic: 330 vector instructions per tile and wave beside 64 MFMAs that run on the same lanes — 91 us for a pass whose
// 134 MB take 25 us at HBM rate), and the halo indices are per-thread constants too (coordinates only on tiles that touch a face).
template <int CIN, int RT, bool VA = false>  // RT = row tiles of 16 output channels (Cout <= 16*RT)
__global__ __launch_bounds__(256) void conv3d_small_bwd_kernel(const SmallBwdParams p) {
    using namespace sc;
    constexpr int C1 = CIN + 1;
    constexpr int NCOL = 27 * C1;
    constexpr int NCT = (NCOL + 15) / 16;
    constexpr int NH = (HV + 255) / 256;  // halo voxels per thread (3)
    constexpr int NQ = 4 * RT;            // VA: channel quads per voxel = 16-byte items per thread and tile
    __shared__ float xsb[2][HV * C1];  // [hv][CIN+1]: raw x, then the in-bounds indicator; two buffers (round 6, see below)
    __shared__ __attribute__((aligned(16))) float dzs[VA ? 2 : 1][VA ? 256 * 16 * RT : 4];  // VA: [voxel][16 * RT channels], two buffers
    const int t = threadIdx.x, l = t & 63, w = t >> 6;
    const int n = blockIdx.y;
    const int j = l & 15, kk = l >> 4;
    const int Cout = p.Cout, D = p.D, H = p.H, W = p.W;
    int boff[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        const int col = ct * 16 + j;
        const int tap = col / C1, c = col - tap * C1;
        const int off = (tap / 9) * (HY * HX) + ((tap / 3) % 3) * HX + tap % 3;
        boff[ct] = col < NCOL ? off * C1 + c : 0;  // dead columns read column 0; never written out
    }
    const int bbase = (w * (HY * HX) + kk) * C1;
    f32x4s acc[RT][NCT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[rt][ct] = f32x4s{0.f, 0.f, 0.f, 0.f};
    const int ntiles = p.tz * p.ty * p.tx;
    auto origin = [&](int tile, int& z0, int& y0, int& x0) {
        const int txi = tile % p.tx;
        const int tt = tile / p.tx;
        z0 = (tt / p.ty) * TZ, y0 = (tt % p.ty) * TY, x0 = txi * TX;
    };
    // A operands of a whole tile: dz[voxel(step, kk)][k = j + 16*rt], zero outside the volume / beyond Cout
    auto a_load = [&](int tile, float (&a)[RT][16]) {
        int z0, y0, x0;
        origin(tile, z0, y0, x0);
        const int z = z0 + w;
#pragma unroll
        for (int s_ = 0; s_ < 16; ++s_) {
            const int y = y0 + (s_ >> 1), x = x0 + (s_ & 1) * 4 + kk;
            const bool vin = z < D && y < H && x < W;
            const size_t vox = (size_t)((n * D + (vin ? z : 0)) * H + (vin ? y : 0)) * W + (vin ? x : 0);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const int k = j + 16 * rt;
                const float v = p.dz[vox * Cout + (k < Cout ? k : 0)];
                a[rt][s_] = (vin && k < Cout) ? v : 0.f;
            }
        }
    };
    // VA: item i = t + 256 * it of a tile = (voxel i / NQ, channel quad i % NQ): constant element offset from the tile's first voxel
    int drel[VA ? NQ : 1];
    if constexpr (VA) {
#pragma unroll
        for (int it = 0; it < NQ; ++it) {
            const int i = t + 256 * it, vox = i / NQ, q = i - vox * NQ;
            drel[it] = (((vox >> 6) * H + ((vox >> 3) & 7)) * W + (vox & 7)) * (16 * RT) + 4 * q;
        }
    }
    auto dz_load = [&](int tile, f32x4 (&v)[VA ? NQ : 1]) {
        int z0, y0, x0;
        origin(tile, z0, y0, x0);
        const float* base = p.dz + ((size_t)((n * D + z0) * H + y0) * W + x0) * (16 * RT);
#pragma unroll
        for (int it = 0; it < NQ; ++it) v[it] = *reinterpret_cast<const f32x4*>(base + drel[it]);
    };
    auto dz_store = [&](float* buf, const f32x4 (&v)[VA ? NQ : 1]) {
#pragma unroll
        for (int it = 0; it < NQ; ++it) *reinterpret_cast<f32x4*>(buf + (size_t)(t + 256 * it) * 4) = v[it];
    };
    const int abase = ((w * 64 + kk) * 16 * RT) + j;  // VA: LDS offset of this lane's A element at step 0, row tile 0
    // halo item `it` of a thread: packed halo coordinates and the voxel offset from the tile's first voxel (constants)
    int hpk[NH], hrel[NH];
#pragma unroll
    for (int it = 0; it < NH; ++it) {
        const int i = t + 256 * it;
        const int hz = i / (HY * HX), rem = i - hz * (HY * HX), hy = rem / HX, hxx = rem - hy * HX;
        hpk[it] = i < HV ? (hz | (hy << 8) | (hxx << 16)) : -1;
        hrel[it] = ((hz - 1) * H + (hy - 1)) * W + (hxx - 1);
    }
    // raw halo voxels of a tile into registers; the LDS stores follow after the current tile's MFMAs (round 6: both operand fetches of
    // tile t+1 run under the arithmetic of tile t — with fill -> barrier -> compute -> barrier per tile every load latency was exposed)
    auto halo_load = [&](int tile, float (&hx)[NH][CIN], unsigned& inside) {
        int z0, y0, x0;
        origin(tile, z0, y0, x0);
        inside = 0;
        const int vbase = ((n * D + z0) * H + y0) * W + x0;
        const bool interior = z0 > 0 && z0 + TZ < D && y0 > 0 && y0 + TY < H && x0 > 0 && x0 + TX < W;  // (uniform)
#pragma unroll
        for (int it = 0; it < NH; ++it) {
            bool in = hpk[it] >= 0;
            if (!interior) {
                const int gz = z0 - 1 + (hpk[it] & 255), gy = y0 - 1 + ((hpk[it] >> 8) & 255), gx = x0 - 1 + ((hpk[it] >> 16) & 255);
                in = in && gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W;
            }
            inside |= (in ? 1u : 0u) << it;
            const float* src = p.x + (size_t)(in ? vbase + hrel[it] : 0) * CIN;
#pragma unroll
            for (int c = 0; c < CIN; ++c) hx[it][c] = in ? src[c] : 0.f;
        }
    };
    auto halo_store = [&](float* xs, const float (&hx)[NH][CIN], unsigned inside) {
#pragma unroll
        for (int it = 0; it < NH; ++it) {
            const int i = t + 256 * it;
            if (i < HV) {
#pragma unroll
                for (int c = 0; c < CIN; ++c) xs[i * C1 + c] = hx[it][c];
                xs[i * C1 + CIN] = ((inside >> it) & 1u) ? 1.f : 0.f;
            }
        }
    };
    float a[VA ? 1 : RT][16], hx[NH][CIN];
    f32x4 dv[VA ? NQ : 1];
    unsigned hin = 0;
    int cur = 0;
    if ((int)blockIdx.x < ntiles) {
        if constexpr (VA) {
            dz_load(blockIdx.x, dv);
            dz_store(dzs[0], dv);
        } else {
            a_load(blockIdx.x, a);
        }
        halo_load(blockIdx.x, hx, hin);
        halo_store(xsb[0], hx, hin);
    }
    __syncthreads();
    for (int tile = blockIdx.x; tile < ntiles; tile += p.B) {
        const float* xs = xsb[cur];
        const bool has_next = tile + p.B < ntiles;
        float an[VA ? 1 : RT][16];
        if (has_next) {
            if constexpr (VA) dz_load(tile + p.B, dv);
            else a_load(tile + p.B, an);
            halo_load(tile + p.B, hx, hin);
        }
#pragma unroll
        for (int s_ = 0; s_ < 16; ++s_) {
            const int so = ((s_ >> 1) * HX + (s_ & 1) * 4) * C1;
            float b[NCT], as[RT];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) b[ct] = xs[bbase + so + boff[ct]];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                if constexpr (VA) as[rt] = dzs[cur][abase + ((s_ >> 1) * 8 + (s_ & 1) * 4) * 16 * RT + 16 * rt];
                else as[rt] = a[rt][s_];
            }
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(as[rt], b[ct], acc[rt][ct], 0, 0, 0);
        }
        if (has_next) {
            halo_store(xsb[cur ^ 1], hx, hin);
            if constexpr (VA) {
                dz_store(dzs[cur ^ 1], dv);
            } else {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int s_ = 0; s_ < 16; ++s_) a[rt][s_] = an[rt][s_];
            }
        }
        __syncthreads();  // the next tile's halo is complete; nobody reads the current buffer any more
        cur ^= 1;
    }
    // fold the four z-plane partials (fixed order 0+1+2+3) through LDS, then write the block's partial.
    // D layout of 16x16x4: col = lane & 15, row = 4*(lane >> 4) + reg
    __shared__ float red[RT * NCT * 4 * 64];
    for (int src = 1; src < 4; ++src) {
        __syncthreads();
        if (w == src) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[((rt * NCT + ct) * 4 + r) * 64 + l] = acc[rt][ct][r];
        }
        __syncthreads();
        if (w == 0) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[rt][ct][r] += red[((rt * NCT + ct) * 4 + r) * 64 + l];
        }
    }
    if (w != 0) return;
    float* dst = p.partial + (size_t)(n * p.B + blockIdx.x) * ((size_t)Cout * NCOL);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const int col = ct * 16 + j;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = rt * 16 + 4 * kk + r;
                if (k < Cout && col < NCOL) dst[(size_t)k * NCOL + col] = acc[rt][ct][r];
            }
        }
}

// finalize: grid = ceil(Cout*27 / 64) blocks of 1024 threads = 64 (k,tap) lanes x 16 partial-groups; every thread keeps
// four independent loads in flight (the pass is latency-bound: a few MB through 7 blocks).  Fixed summation order;
// forms dw (summed over n in-thread) and adds the GroupNorm-backward sums (S1,S2) per (n, c).
constexpr int FIN_GROUPS = 16, FIN_UNROLL = 4;
__global__ __launch_bounds__(64 * FIN_GROUPS) void conv3d_small_bwd_finalize_kernel(const float* __restrict__ partial,
                                                                                   const float* __restrict__ affine,
                                                                                   const float* __restrict__ w, int N,
                                                                                   int B, int Cin, int Cout,
                                                                                   float* __restrict__ dw,
                                                                                   double* __restrict__ gstats) {
    __shared__ float red[FIN_GROUPS][64][sc::MAXC + 1];
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
#pragma unroll
        for (int c = 0; c <= sc::MAXC; ++c) acc[c] = 0.f;
        if (valid) {
            const float* base = partial + (size_t)n * B * ((size_t)nkt * C1) + (size_t)kt * C1;
            for (int b = grp; b < B; b += FIN_GROUPS * FIN_UNROLL) {
                float tmp[FIN_UNROLL][sc::MAXC + 1];
#pragma unroll
                for (int u = 0; u < FIN_UNROLL; ++u) {
                    const int bb = b + u * FIN_GROUPS;
                    const float* src = base + (size_t)(bb < B ? bb : b) * ((size_t)nkt * C1);
#pragma unroll
                    for (int c = 0; c <= sc::MAXC; ++c) tmp[u][c] = (c < C1 && bb < B) ? src[c] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < FIN_UNROLL; ++u)
#pragma unroll
                    for (int c = 0; c <= sc::MAXC; ++c) acc[c] += tmp[u][c];
            }
        }
#pragma unroll
        for (int c = 0; c <= sc::MAXC; ++c)
            if (c < C1) red[grp][lane][c] = acc[c];
        __syncthreads();
        if (grp == 0) {  // wave 0: lanes = 64 (k,tap) pairs
            double X[sc::MAXC + 1];
            for (int c = 0; c < C1; ++c) {
                double sum = 0.0;
                for (int g = 0; g < FIN_GROUPS; ++g) sum += (double)red[g][lane][c];
                X[c] = sum;
            }
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

static int small_bwd_blocks(int N, int D, int H, int W, bool va = false) {
    const long long ntiles = (long long)((D + 3) / 4) * ((H + 7) / 8) * ((W + 7) / 8);
    // ~4 blocks per CU in total (latency hiding), each walks its share of tiles (key 20: A/B).  The LDS-staged form (va; 46 KB per block:
    // three per CU) with 2 per CU: 48 + 14 us (kernel + finalize, whose work is proportional to the block count) against 45 + 18 at 3 per
    // CU and 56 + 22 at 4
    long long B = (g_u3d_tune[20] > 0 ? g_u3d_tune[20] : (va ? 512 : 1024)) / (N > 0 ? N : 1);
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
    U3D_ENTER(device);
    U3D_REQUIRE(x && dz && w && dw && workspace && N > 0 && D > 0 && H > 0 && W > 0, "u3d_conv3d_small_cin_bwd: bad argument");
    U3D_REQUIRE(Cin >= 1 && Cin <= sc::MAXC && Cout >= 1 && Cout <= 32 && Cout * 27 <= 1024,
                "u3d_conv3d_small_cin_bwd: needs Cin<=4, Cout<=32 (got %d,%d)", Cin, Cout);
    SmallBwdParams p;
    p.x = x, p.dz = dz, p.partial = workspace;
    p.N = N, p.D = D, p.H = H, p.W = W, p.Cin = Cin, p.Cout = Cout;
    p.tz = (D + sc::TZ - 1) / sc::TZ, p.ty = (H + sc::TY - 1) / sc::TY, p.tx = (W + sc::TX - 1) / sc::TX;
    // whole tiles, whole row tiles of 16 channels, aligned dz: the dz tile through LDS (VA); key 20 = -1: the first form (A/B)
    const bool va = Cout == 16 && D % sc::TZ == 0 && H % sc::TY == 0 && W % sc::TX == 0 && ((uintptr_t)dz & 15) == 0 &&
                    g_u3d_tune[20] != -1;
    p.B = small_bwd_blocks(N, D, H, W, va);  // (<= the count u3d_small_cin_bwd_workspace_floats sizes for)
    const size_t need = (size_t)N * p.B * Cout * 27 * (Cin + 1);
    if (workspace_floats < need)
        return u3d_set_err(U3D_EWORKSPACE, "u3d_conv3d_small_cin_bwd: workspace %zu < %zu floats", workspace_floats, need);
    const dim3 grid((unsigned)p.B, (unsigned)N), block(256);
    hipStream_t st = (hipStream_t)stream;
#define U3D_SMALL_BWD(CIN_)                                                                          \
    do {                                                                                             \
        if (Cout <= 16 && va)                                                                        \
            hipLaunchKernelGGL((conv3d_small_bwd_kernel<CIN_, 1, true>), grid, block, 0, st, p);     \
        else if (Cout <= 16)                                                                         \
            hipLaunchKernelGGL((conv3d_small_bwd_kernel<CIN_, 1>), grid, block, 0, st, p);           \
        else                                                                                         \
            hipLaunchKernelGGL((conv3d_small_bwd_kernel<CIN_, 2>), grid, block, 0, st, p);           \
    } while (0)
    if (Cin == 1)
        U3D_SMALL_BWD(1);
    else if (Cin == 2)
        U3D_SMALL_BWD(2);
    else if (Cin == 3)
        U3D_SMALL_BWD(3);
    else
        U3D_SMALL_BWD(4);
#undef U3D_SMALL_BWD
    U3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(conv3d_small_bwd_finalize_kernel, dim3((Cout * 27 + 63) / 64), dim3(64 * FIN_GROUPS), 0, (hipStream_t)stream,
                       workspace, affine, w, N, p.B, Cin, Cout, dw, gstats);
    U3D_LAUNCH_CHECK();
    return 0;
}
