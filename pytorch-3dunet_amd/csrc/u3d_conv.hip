// u3d_conv.hip — Conv3d 3x3x3 (stride 1, pad 1, bias-free) forward / data-gradient / weight-gradient as
// implicit GEMM on the gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32), plus weight packing and a naive
// direct convolution used only to cross-check on the device.
//
// Replaces the ATen kernels behind nn.Conv3d(in,out,3,padding=1,bias=False) of the reference
// (pytorch3dunet/unet3d/buildingblocks.py:56) and its autograd (trainer.py:245).
//
// Data layout: activations NDHWC fp32.  A block owns a 4x8x8 output tile (256 voxels) and
// BN = 32*NT output channels.  Per 16-channel input chunk the 6x10x10 halo tile is staged through LDS with
// the GroupNorm affine applied on the way (zero padding stays exactly zero), layout
// [hz][hy][hx][16] with a 4-float row pad (row stride 164) — conflict-free for the ds_read_b128 A-fragment
// reads (tools/lds_bank_model.py).  An MFMA M-tile is 4(y) x 8(x) voxels at one z; wave w owns z = z0+w and
// both y-halves (MT = 2).  B fragments (weights) are read straight from the packed global image (1 KiB
// contiguous per wave-load, L1/L2 resident), software-prefetched two k-steps ahead.
#include "u3d_common.h"

// run-time tuning knobs (u3d_set_tuning): [0] stagger on/off
int g_u3d_tune[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // [1] = ablation mask for timing experiments (wrong results!)

namespace cv {
constexpr int TZ = 4, TY = 8, TX = 8;
constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2;
constexpr int CC = 16;             // input channels per chunk
constexpr int CS = 16;             // voxel stride (floats)
constexpr int RS = HX * CS + 4;    // row stride 164 (bank-conflict-free, see header)
constexpr int PS = HY * RS;        // plane stride 1640
constexpr int LDS_FLOATS = HZ * PS + 4;           // 9840 floats + one dummy float4 slot = 39376 B
constexpr int NITEMS = HZ * HY * HX * (CC / 4);   // 2400 float4 items per chunk
constexpr int NIT = (NITEMS + 255) / 256;         // 10
constexpr int NSTEP = 27 * (CC / 8);              // 54 k-steps of 8 channels per chunk
}  // namespace cv

struct ConvParams {
    u3d_src_t src;
    u3d_src_t gx;
    const float* wp;
    float* out;
    double* out_stats;
    double* gstats;
    int N, D, H, W, Cout;
    int nchunks, ncb, ntot;
    int tz, ty, tx;
    int relu, vec, has_gx;
    int stagger;  // shader cycles of one phase step (0 = off)
};

// ABL: timing-only ablation mask (tools/conv_microbench.py): 1 no B loads in the k-loop, 2 no A LDS reads in the
// k-loop, 4 no re-staging after the first chunk, 8 no epilogue.  ABL != 0 produces wrong results by design.
template <int NT, bool VEC, int ABL = 0>
__global__ __launch_bounds__(256) void conv3d_mfma_kernel(const ConvParams p) {
    using namespace cv;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t = threadIdx.x;
    const int l = t & 63, w = t >> 6, m = l & 31, h = l >> 5;

    // ---- block -> (tile, cout block) with XCD-contiguous ordering
    const int logical = u3d_xcd_remap(blockIdx.x, gridDim.x);
    const int cb = logical % p.ncb;
    int tile = logical / p.ncb;
    const int txi = tile % p.tx;
    tile /= p.tx;
    const int tyi = tile % p.ty;
    tile /= p.ty;
    const int tzi = tile % p.tz;
    const int n = tile / p.tz;
    const int z0 = tzi * TZ, y0 = tyi * TY, x0 = txi * TX;
    const int D = p.D, H = p.H, W = p.W;
    const int Ctot = p.src.C0 + p.src.C1;

    // ---- de-synchronise the first wave of workgroups.  All resident blocks are dispatched together and do
    //      identical work, so left alone they run in lockstep: every chunk they all stage their halo tiles at the
    //      same moment (a tens-of-MB burst with every MFMA pipe idle) and then all compute with memory idle.  A
    //      hashed start phase makes one block's staging overlap its co-residents' MFMA phase; blocks dispatched
    //      later inherit the spread.
    if (p.stagger > 0 && blockIdx.x < 256u * 4u) {
        const unsigned hsh = (blockIdx.x * 2654435761u) >> 30;  // 0..3
        if (hsh) {
            const long long until = clock64() + (long long)hsh * p.stagger;
            while (clock64() < until) __builtin_amdgcn_s_sleep(64);
        }
    }

    // ---- per-thread staging descriptors (constant across chunks)
    int ldsoff[NIT], gv0[NIT], gv1[NIT];
    const int q = t & 3;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int item = t + 256 * it;
        const int vox = item >> 2;
        const bool in = item < NITEMS;
        const int hz = vox / (HY * HX);
        const int rem = vox - hz * (HY * HX);
        const int hy = rem / HX;
        const int hx = rem - hy * HX;
        ldsoff[it] = in ? hz * PS + hy * RS + hx * CS + 4 * q : HZ * PS;  // tail items -> dummy slot
        const int gz = z0 - 1 + hz, gy = y0 - 1 + hy, gxx = x0 - 1 + hx;
        const bool ok = in && gz >= 0 && gz < D && gy >= 0 && gy < H && gxx >= 0 && gxx < W;
        gv0[it] = -1;
        gv1[it] = 0;
        if (ok) u3d_vox_index(p.src, n, gz, gy, gxx, D, H, W, gv0[it], gv1[it]);
    }

    f32x16 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    // A-fragment base: lane (m,h) -> voxel (zl = w, yl = (m>>3) [+4 for mt=1], xl = m&7), channels 4h..4h+3
    const int abase = w * PS + (m >> 3) * RS + (m & 7) * CS + 4 * h;

    for (int ch = 0; ch < p.nchunks; ++ch) {
        // weights of the first two k-steps: issued with the staging loads so their latency is shared
        const f32x4* wq = reinterpret_cast<const f32x4*>(p.wp) + ((size_t)ch * NSTEP * p.ntot + cb * NT) * 64 + l;
        const size_t wstep = (size_t)p.ntot * 64;
        f32x4 bq[3][NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            bq[0][nt] = wq[nt * 64];
            bq[1][nt] = wq[wstep + nt * 64];
        }
        // ---- stage the halo tile of this 16-channel chunk: global -> regs -> (affine) -> LDS.
        //      Fast path is branch-free: every load is issued unconditionally from a clamped (always valid)
        //      address so the 10 loads of a thread are in flight together; validity is applied by select.
        if (!(ABL & 4) || ch == 0) {
            const int cq = ch * CC + 4 * q;
            f32x4 v[NIT];
            f32x4 ga = {1.f, 1.f, 1.f, 1.f}, gb = {0.f, 0.f, 0.f, 0.f};
            if constexpr (VEC) {
                const bool cok = cq < Ctot;
                const bool from0 = cq < p.src.C0;
                const float* base = !cok ? p.src.p0 : (from0 ? p.src.p0 + cq : p.src.p1 + (cq - p.src.C0));
                const int Cs = (from0 || !cok) ? p.src.C0 : p.src.C1;
                if (p.src.affine) {
                    const float* ap = p.src.affine + ((size_t)n * Ctot + (cok ? cq : 0)) * 2;
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(ap);
                    const f32x4 hi = *reinterpret_cast<const f32x4*>(ap + 4);
                    ga = f32x4{lo[0], lo[2], hi[0], hi[2]};
                    gb = f32x4{lo[1], lo[3], hi[1], hi[3]};
                }
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const bool ok = cok && gv0[it] >= 0;
                    const int idx = ok ? (from0 ? gv0[it] : gv1[it]) : 0;
                    v[it] = *reinterpret_cast<const f32x4*>(base + (size_t)idx * Cs);
                }
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const bool ok = cok && gv0[it] >= 0;
                    f32x4 val = v[it] * ga + gb;
#pragma unroll
                    for (int e = 0; e < 4; ++e) val[e] = ok ? val[e] : 0.f;  // padding stays exactly 0
                    *reinterpret_cast<f32x4*>(&lds[ldsoff[it]]) = val;
                }
            } else {
                u3d_load_affine(p.src.affine, n, Ctot, cq, false, ga, gb);
#pragma unroll 1
                for (int it = 0; it < NIT; ++it) {
                    f32x4 val = {0.f, 0.f, 0.f, 0.f};
                    if (gv0[it] >= 0) val = u3d_load_quad(p.src, gv0[it], gv1[it], cq, false) * ga + gb;
                    *reinterpret_cast<f32x4*>(&lds[ldsoff[it]]) = val;
                }
            }
        }
        __syncthreads();

        // ---- 54 k-steps (27 taps x 2 channel-octets): 8*NT MFMAs each.  Software pipeline pinned with
        //      sched_barrier: B (weights, global/L1) is fetched two steps ahead, A (LDS) one step ahead; the
        //      compiler inserts the matching counted vmcnt/lgkmcnt waits.
        f32x4 aq[2][2];
        aq[0][0] = *reinterpret_cast<const f32x4*>(&lds[abase]);
        aq[0][1] = *reinterpret_cast<const f32x4*>(&lds[abase + 4 * RS]);
#pragma unroll
        for (int st = 0; st < NSTEP; ++st) {
            if (st + 2 < NSTEP) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if (ABL & 1) {
                        bq[(st + 2) % 3][nt] = bq[st % 3][nt];
                        asm volatile("" : "+v"(bq[(st + 2) % 3][nt]));
                    } else {
                        bq[(st + 2) % 3][nt] = wq[(size_t)(st + 2) * wstep + nt * 64];
                    }
                }
            }
            if ((ABL & 2) && st + 1 < NSTEP) {
                aq[(st + 1) & 1][0] = aq[st & 1][0];
                aq[(st + 1) & 1][1] = aq[st & 1][1];
                asm volatile("" : "+v"(aq[(st + 1) & 1][0]), "+v"(aq[(st + 1) & 1][1]));
            } else if (st + 1 < NSTEP) {
                const int tap = (st + 1) >> 1, s1_ = (st + 1) & 1;
                const int aoff = (tap / 9) * PS + ((tap / 3) % 3) * RS + (tap % 3) * CS + 8 * s1_;
                aq[(st + 1) & 1][0] = *reinterpret_cast<const f32x4*>(&lds[abase + aoff]);
                aq[(st + 1) & 1][1] = *reinterpret_cast<const f32x4*>(&lds[abase + 4 * RS + aoff]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[st & 1][0][j], bq[st % 3][nt][j], acc[0][nt], 0, 0, 0);
                    acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[st & 1][1][j], bq[st % 3][nt][j], acc[1][nt], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }

    // ---- epilogue.  C/D layout of 32x32 MFMA: col = lane&31 (cout), row = (r&3) + 8*(r>>2) + 4*(lane>>5);
    //      M-tile row -> (y = row>>3, x = row&7)  =>  reg r of lane (m,h): y = r>>2, x = (r&3) + 4h.
    if (ABL & 8) {
        // keep the accumulators alive without storing them
        float keep = 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) keep += acc[mt][nt][r];
        if (keep == 123.456f) p.out[0] = keep;
        return;
    }
    const int z = z0 + w;
    float s1[NT], s2[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        s1[nt] = 0.f;
        s2[nt] = 0.f;
    }
    const bool want_stats = p.out_stats != nullptr;
    const bool want_g = p.gstats != nullptr;
    // dgrad epilogue: x of the layer input at (voxel, cout); per-lane source select is fixed per nt
    const float* xb[NT];
    int xcs[NT];
    bool xfrom0[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = (cb * NT + nt) * 32 + m;
        const bool cok = co < p.Cout;
        xfrom0[nt] = co < p.gx.C0 || !cok;
        xb[nt] = !cok ? p.gx.p0 : (xfrom0[nt] ? p.gx.p0 + co : p.gx.p1 + (co - p.gx.C0));
        xcs[nt] = xfrom0[nt] ? p.gx.C0 : p.gx.C1;
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int y = y0 + mt * 4 + (r >> 2);
            const int x = x0 + (r & 3) + 4 * h;
            const bool vok = z < D && y < H && x < W;
            int v0 = 0, v1 = 0;
            if (want_g) {
                // clamped coordinates: always a valid address, masked below
                u3d_vox_index(p.gx, n, min(z, D - 1), min(y, H - 1), min(x, W - 1), D, H, W, v0, v1);
            }
            const size_t vidx = (size_t)((n * D + z) * H + y) * W + x;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int co = (cb * NT + nt) * 32 + m;
                const bool ok = vok && co < p.Cout;
                float val = acc[mt][nt][r];
                if (p.relu) val = fmaxf(val, 0.f);
                if (ok) p.out[vidx * p.Cout + co] = val;
                const float vv = ok ? val : 0.f;
                if (want_g) {
                    const float xv = xb[nt][(size_t)(xfrom0[nt] ? v0 : v1) * xcs[nt]];
                    s1[nt] += vv;
                    s2[nt] += vv * xv;
                } else {
                    s1[nt] += vv;
                    s2[nt] += vv * vv;
                }
            }
        }
    }
    if (want_stats || want_g) {
        // reduce over the two half-waves (same cout), then over the 4 waves through LDS, then one f64 atomic
        float* red = lds;  // [4][NT][32][2]
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            s1[nt] += __shfl_xor(s1[nt], 32);
            s2[nt] += __shfl_xor(s2[nt], 32);
            if (h == 0) {
                red[((w * NT + nt) * 32 + m) * 2 + 0] = s1[nt];
                red[((w * NT + nt) * 32 + m) * 2 + 1] = s2[nt];
            }
        }
        __syncthreads();
        if (t < NT * 32) {
            const int nt = t >> 5, mm = t & 31;
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) {
                a += red[((ww * NT + nt) * 32 + mm) * 2 + 0];
                b += red[((ww * NT + nt) * 32 + mm) * 2 + 1];
            }
            const int co = (cb * NT + nt) * 32 + mm;
            if (co < p.Cout) {
                double* dst = (want_stats ? p.out_stats : p.gstats) + ((size_t)n * p.Cout + co) * 2;
                u3d_atomic_add_f64(dst, (double)a);
                u3d_atomic_add_f64(dst + 1, (double)b);
            }
        }
    }
}

// =================================================================================================
// Weight gradient.  GEMM view: M = 32 input channels of one tap, N = 32 output channels, K = voxels.
// A block owns (split s, 32-channel input chunk, 32-channel output block), walks its share of 2x8x8 voxel
// tiles and keeps all 27 taps x 32 x 32 partial sums in registers (wave w owns taps w, w+4, ..: 7 x 16 regs).
namespace wg {
constexpr int TZ = 2, TY = 8, TX = 8;
constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2;
constexpr int CC = 32;
constexpr int CSg = 32, RSg = HX * CSg, PSg = HY * RSg;  // 32, 320, 3200
constexpr int G_FLOATS = HZ * PSg;                       // 12800
constexpr int TV = TZ * TY * TX;                         // 128 voxels
constexpr int DZ_FLOATS = TV * 32;                       // 4096
constexpr int LDS_FLOATS = G_FLOATS + DZ_FLOATS;         // 16896 floats = 67584 B
constexpr int NITEMS_G = HZ * HY * HX * (CC / 4);        // 3200
constexpr int NIT_G = (NITEMS_G + 255) / 256;            // 13
constexpr int NIT_DZ = TV * 8 / 256;                     // 4
}  // namespace wg

struct WgradParams {
    u3d_src_t src;
    const float* dz;
    float* partial;
    int N, D, H, W, Cout;
    int nchunks, nkb, S;
    int tz, ty, tx, ntiles, tps;
    int vec, dzvec;
    int stagger;
};

template <bool VEC>
__global__ __launch_bounds__(256, 2) void conv3d_wgrad_kernel(const WgradParams p) {
    using namespace wg;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* gl = lds;
    float* dzl = lds + G_FLOATS;
    const int t = threadIdx.x;
    const int l = t & 63, i = l & 31, h = l >> 5;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);

    const int logical = u3d_xcd_remap(blockIdx.x, gridDim.x);
    const int kb = logical % p.nkb;
    const int chunk = (logical / p.nkb) % p.nchunks;
    const int s = logical / (p.nkb * p.nchunks);
    const int D = p.D, H = p.H, W = p.W;
    const int Ctot = p.src.C0 + p.src.C1;

    // start-phase offset against lockstep staging (see conv3d_mfma_kernel)
    if (p.stagger > 0) {
        const unsigned hsh = (blockIdx.x * 2654435761u) >> 31;  // 0..1: two blocks per CU
        if (hsh) {
            const long long until = clock64() + (long long)p.stagger;
            while (clock64() < until) __builtin_amdgcn_s_sleep(64);
        }
    }

    int toff[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const int tap = w + 4 * k;
        toff[k] = tap < 27 ? (tap / 9) * PSg + ((tap / 3) % 3) * RSg + (tap % 3) * CSg : 0;
    }
    f32x16 acc[7];
#pragma unroll
    for (int k = 0; k < 7; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;

    // staging descriptors: g halo items (voxel = item>>3, quad = t&7), packed halo coords
    const int q = t & 7;
    int gpk[NIT_G], goff[NIT_G];
#pragma unroll
    for (int it = 0; it < NIT_G; ++it) {
        const int item = t + 256 * it;
        const int vox = item >> 3;
        const bool in = item < NITEMS_G;
        const int hz = vox / (HY * HX);
        const int rem = vox - hz * (HY * HX);
        const int hy = rem / HX;
        const int hx = rem - hy * HX;
        gpk[it] = in ? (hz | (hy << 8) | (hx << 16)) : (int)0x80000000;
        goff[it] = hz * PSg + hy * RSg + hx * CSg + 4 * q;
    }

    const int tile_end = min(p.ntiles, (s + 1) * p.tps);
    for (int tile = s * p.tps; tile < tile_end; ++tile) {
        int tt = tile;
        const int txi = tt % p.tx;
        tt /= p.tx;
        const int tyi = tt % p.ty;
        tt /= p.ty;
        const int tzi = tt % p.tz;
        const int n = tt / p.tz;
        const int z0 = tzi * TZ, y0 = tyi * TY, x0 = txi * TX;

        // ---- stage g (GroupNorm-affine input, zero padded) and dz; fast path is branch-free (clamped
        //      addresses + select) so all loads of a thread are in flight together
        {
            const int cq = chunk * CC + 4 * q;
            const int co = kb * 32 + 4 * q;
            if constexpr (VEC) {
                const bool cok = cq < Ctot;
                const bool from0 = cq < p.src.C0;
                const float* base = !cok ? p.src.p0 : (from0 ? p.src.p0 + cq : p.src.p1 + (cq - p.src.C0));
                const int Cs = (from0 || !cok) ? p.src.C0 : p.src.C1;
                f32x4 ga = {1.f, 1.f, 1.f, 1.f}, gb = {0.f, 0.f, 0.f, 0.f};
                if (p.src.affine) {
                    const float* ap = p.src.affine + ((size_t)n * Ctot + (cok ? cq : 0)) * 2;
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(ap);
                    const f32x4 hi = *reinterpret_cast<const f32x4*>(ap + 4);
                    ga = f32x4{lo[0], lo[2], hi[0], hi[2]};
                    gb = f32x4{lo[1], lo[3], hi[1], hi[3]};
                }
                // dz first (4 loads), then g in two batches of 7/6 to bound live registers
                f32x4 dv[NIT_DZ];
                const bool dcok = co < p.Cout;
#pragma unroll
                for (int it = 0; it < NIT_DZ; ++it) {
                    const int vox = (t >> 3) + 32 * it;
                    const int z = z0 + (vox >> 6), y = y0 + ((vox >> 3) & 7), x = x0 + (vox & 7);
                    const bool ok = dcok && z < D && y < H && x < W;
                    const int vi = ok ? ((n * D + z) * H + y) * W + x : 0;
                    dv[it] = *reinterpret_cast<const f32x4*>(p.dz + (size_t)vi * p.Cout + (dcok ? co : 0));
#pragma unroll
                    for (int e = 0; e < 4; ++e) dv[it][e] = ok ? dv[it][e] : 0.f;
                }
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    constexpr int HB = (NIT_G + 1) / 2;  // 7
                    f32x4 v[HB];
                    bool ok[HB];
#pragma unroll
                    for (int k = 0; k < HB; ++k) {
                        const int it = half * HB + k;
                        if (it < NIT_G) {
                            const int gz = z0 - 1 + (gpk[it] & 0xff), gy = y0 - 1 + ((gpk[it] >> 8) & 0xff),
                                      gxx = x0 - 1 + ((gpk[it] >> 16) & 0xff);
                            ok[k] = cok && gpk[it] >= 0 && gz >= 0 && gz < D && gy >= 0 && gy < H && gxx >= 0 && gxx < W;
                            int v0, v1;
                            u3d_vox_index(p.src, n, min(max(gz, 0), D - 1), min(max(gy, 0), H - 1),
                                          min(max(gxx, 0), W - 1), D, H, W, v0, v1);
                            const int idx = ok[k] ? (from0 ? v0 : v1) : 0;
                            v[k] = *reinterpret_cast<const f32x4*>(base + (size_t)idx * Cs);
                        }
                    }
#pragma unroll
                    for (int k = 0; k < HB; ++k) {
                        const int it = half * HB + k;
                        if (it < NIT_G) {
                            f32x4 val = v[k] * ga + gb;
#pragma unroll
                            for (int e = 0; e < 4; ++e) val[e] = ok[k] ? val[e] : 0.f;
                            if (gpk[it] >= 0) *reinterpret_cast<f32x4*>(&gl[goff[it]]) = val;
                        }
                    }
                }
#pragma unroll
                for (int it = 0; it < NIT_DZ; ++it)
                    *reinterpret_cast<f32x4*>(&dzl[((t >> 3) + 32 * it) * 32 + 4 * q]) = dv[it];
            } else {
                f32x4 ga, gb;
                u3d_load_affine(p.src.affine, n, Ctot, cq, false, ga, gb);
#pragma unroll 1
                for (int it = 0; it < NIT_G; ++it) {
                    if (gpk[it] < 0) continue;
                    const int gz = z0 - 1 + (gpk[it] & 0xff), gy = y0 - 1 + ((gpk[it] >> 8) & 0xff),
                              gxx = x0 - 1 + ((gpk[it] >> 16) & 0xff);
                    f32x4 val = {0.f, 0.f, 0.f, 0.f};
                    if (gz >= 0 && gz < D && gy >= 0 && gy < H && gxx >= 0 && gxx < W) {
                        int v0, v1;
                        u3d_vox_index(p.src, n, gz, gy, gxx, D, H, W, v0, v1);
                        val = u3d_load_quad(p.src, v0, v1, cq, false) * ga + gb;
                    }
                    *reinterpret_cast<f32x4*>(&gl[goff[it]]) = val;
                }
#pragma unroll 1
                for (int it = 0; it < NIT_DZ; ++it) {
                    const int vox = (t >> 3) + 32 * it;
                    const int z = z0 + (vox >> 6), y = y0 + ((vox >> 3) & 7), x = x0 + (vox & 7);
                    f32x4 val = {0.f, 0.f, 0.f, 0.f};
                    if (z < D && y < H && x < W) {
                        const float* sp = p.dz + ((size_t)((n * D + z) * H + y) * W + x) * p.Cout + co;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (co + e < p.Cout) val[e] = sp[e];
                    }
                    *reinterpret_cast<f32x4*>(&dzl[vox * 32 + 4 * q]) = val;
                }
            }
        }
        __syncthreads();

        // ---- 64 voxel pairs x 7 taps.  A[i=c][k=h] = g[voxel 2t+h shifted by tap][c], B[k=h][j] = dz[voxel][j]
        const int abase = i + h * CSg;
#pragma unroll 2
        for (int row = 0; row < 16; ++row) {
            const int rowbase = (row >> 3) * PSg + (row & 7) * RSg + abase;
#pragma unroll
            for (int tq = 0; tq < 4; ++tq) {
                const float b = dzl[(row * 8 + 2 * tq + h) * 32 + i];
#pragma unroll
                for (int k = 0; k < 7; ++k) {
                    const float a = gl[rowbase + 2 * tq * CSg + toff[k]];
                    acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[k], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // ---- partial[s][chunk][kb][tap][c][k]; D rows = c, cols = k
    float* dst = p.partial + ((size_t)((s * p.nchunks + chunk) * p.nkb + kb) * 27) * 1024;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const int tap = w + 4 * k;
        if (tap < 27) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = (r & 3) + 8 * (r >> 2) + 4 * h;
                dst[((size_t)tap * 32 + c) * 32 + i] = acc[k][r];
            }
        }
    }
}

// deterministic second pass of the split-K: block = 64 outputs x 4 split-groups, fixed summation order
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                           int S, int nchunks, int nkb, int Cin, int Cout) {
    __shared__ float red[4][64];
    const long long total = (long long)Cin * 27 * Cout;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const long long idx = (long long)blockIdx.x * 64 + lane;  // (c, tap, k) with k fastest: coalesced partial reads
    float sum = 0.f;
    int k = 0, tap = 0, c = 0;
    if (idx < total) {
        k = (int)(idx % Cout);
        const long long r = idx / Cout;
        tap = (int)(r % 27);
        c = (int)(r / 27);
        const int chunk = c >> 5, kb = k >> 5;
        const size_t off = ((size_t)(chunk * nkb + kb) * 27 + tap) * 1024 + (c & 31) * 32 + (k & 31);
        const size_t sstride = (size_t)nchunks * nkb * 27 * 1024;
        const int per = (S + 3) / 4;
        const int s0 = grp * per, s1 = min(S, s0 + per);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int s = s0;
        for (; s + 3 < s1; s += 4) {
            a0 += partial[(size_t)s * sstride + off];
            a1 += partial[(size_t)(s + 1) * sstride + off];
            a2 += partial[(size_t)(s + 2) * sstride + off];
            a3 += partial[(size_t)(s + 3) * sstride + off];
        }
        for (; s < s1; ++s) a0 += partial[(size_t)s * sstride + off];
        sum = (a0 + a1) + (a2 + a3);
    }
    red[grp][lane] = sum;
    __syncthreads();
    if (grp == 0 && idx < total)
        dw[((size_t)k * Cin + c) * 27 + tap] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// =================================================================================================
// weight packing: packed f32x4 index (((ch*54 + st)*ntot + ntg)*64 + lane), element j:
//   k-channel  c  = ch*16 + 8*(st&1) + 4*(lane>>5) + j,  tap = st>>1,  n-channel = ntg*32 + (lane&31)
__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin,
                                    int mode, int nchunks, int ntot) {
    const long long total = (long long)nchunks * cv::NSTEP * ntot * 256;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(idx & 3);
        const int lane = (int)((idx >> 2) & 63);
        long long r = idx >> 8;
        const int ntg = (int)(r % ntot);
        r /= ntot;
        const int st = (int)(r % cv::NSTEP);
        const int ch = (int)(r / cv::NSTEP);
        const int kc = ch * 16 + 8 * (st & 1) + 4 * (lane >> 5) + j;
        const int tap = st >> 1;
        const int nc = ntg * 32 + (lane & 31);
        float v = 0.f;
        if (mode == 0) {
            if (kc < Cin && nc < Cout) v = w[((size_t)nc * Cin + kc) * 27 + tap];
        } else {
            // dgrad: contraction over original cout (kc), output = original cin (nc), flipped taps
            if (kc < Cout && nc < Cin) v = w[((size_t)kc * Cin + nc) * 27 + (26 - tap)];
        }
        out[idx] = v;
    }
}

// =================================================================================================
__global__ void conv3d_naive_kernel(const u3d_src_t src, const float* __restrict__ w, float* __restrict__ out,
                                    int N, int D, int H, int W, int Cin, int Cout, int relu, int flip) {
    const long long total = (long long)N * D * H * W * Cout;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int co = (int)(idx % Cout);
        long long v = idx / Cout;
        const int x = (int)(v % W);
        v /= W;
        const int y = (int)(v % H);
        v /= H;
        const int z = (int)(v % D);
        const int n = (int)(v / D);
        float sum = 0.f;
        for (int tap = 0; tap < 27; ++tap) {
            const int zz = z + tap / 9 - 1, yy = y + (tap / 3) % 3 - 1, xx = x + tap % 3 - 1;
            if (zz < 0 || zz >= D || yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
            int v0, v1;
            u3d_vox_index(src, n, zz, yy, xx, D, H, W, v0, v1);
            for (int c = 0; c < Cin; ++c) {
                float g = u3d_load_elem(src, v0, v1, c);
                if (src.affine) {
                    const float* ab = src.affine + ((size_t)n * Cin + c) * 2;
                    g = g * ab[0] + ab[1];
                }
                const float wt = flip ? w[((size_t)c * Cout + co) * 27 + (26 - tap)] : w[((size_t)co * Cin + c) * 27 + tap];
                sum += g * wt;
            }
        }
        if (relu) sum = fmaxf(sum, 0.f);
        out[idx] = sum;
    }
}

// =================================================================================================
// host side
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

static bool src_vec_ok(const u3d_src_t* s) {
    if (s->C0 % 4 != 0 || s->C1 % 4 != 0) return false;
    if (((uintptr_t)s->p0 & 15) != 0) return false;
    if (s->C1 > 0 && ((uintptr_t)s->p1 & 15) != 0) return false;
    if (s->affine && ((uintptr_t)s->affine & 15) != 0) return false;
    return true;
}

static int check_src(const u3d_src_t* s, const char* what) {
    U3D_REQUIRE(s != nullptr && s->p0 != nullptr && s->C0 > 0, "%s: null source", what);
    U3D_REQUIRE(s->C1 >= 0, "%s: negative C1", what);
    if (s->C1 > 0)
        U3D_REQUIRE(s->p1 && s->zmap && s->ymap && s->xmap && s->D1 > 0 && s->H1 > 0 && s->W1 > 0,
                    "%s: low-res source needs p1, index maps and dims", what);
    return 0;
}

extern "C" int u3d_set_tuning(int key, int value) {
    if (key < 0 || key >= 8) return u3d_set_err(U3D_EINVAL, "u3d_set_tuning: key out of range");
    g_u3d_tune[key] = value;
    return 0;
}

extern "C" size_t u3d_packed_weight_floats(int Cin, int Cout, int mode) {
    const int K = mode == 0 ? Cin : Cout, Nn = mode == 0 ? Cout : Cin;
    return (size_t)cdiv(K, 16) * cv::NSTEP * cdiv(Nn, 32) * 256;
}

extern "C" int u3d_pack_weights(int device, u3d_stream_t stream, const float* w, int Cout, int Cin, int mode,
                                float* packed) {
    if (int e = u3d_enter(device)) return e;
    U3D_REQUIRE(w && packed && Cout > 0 && Cin > 0 && (mode == 0 || mode == 1), "u3d_pack_weights: bad argument");
    const int K = mode == 0 ? Cin : Cout, Nn = mode == 0 ? Cout : Cin;
    const int nchunks = cdiv(K, 16), ntot = cdiv(Nn, 32);
    const long long total = (long long)nchunks * cv::NSTEP * ntot * 256;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, packed, Cout, Cin,
                       mode, nchunks, ntot);
    U3D_LAUNCH_CHECK();
    return 0;
}

extern "C" int u3d_conv3d(int device, u3d_stream_t stream, const u3d_src_t* src, const float* packed_w, float* out,
                          int N, int D, int H, int W, int Cout, int relu, double* out_stats, const u3d_src_t* gx,
                          double* gstats) {
    if (int e = u3d_enter(device)) return e;
    if (int e = check_src(src, "u3d_conv3d")) return e;
    U3D_REQUIRE(packed_w && out && N > 0 && D > 0 && H > 0 && W > 0 && Cout > 0, "u3d_conv3d: bad argument");
    U3D_REQUIRE((long long)N * D * H * W < (1ll << 31), "u3d_conv3d: N*D*H*W must be < 2^31");
    U3D_REQUIRE(!(out_stats && gstats), "u3d_conv3d: out_stats and gstats are mutually exclusive");
    U3D_REQUIRE((gstats == nullptr) == (gx == nullptr), "u3d_conv3d: gx and gstats go together");
    U3D_REQUIRE(((uintptr_t)packed_w & 15) == 0, "u3d_conv3d: packed weights must be 16-byte aligned");
    ConvParams p;
    p.src = *src;
    if (gx) {
        if (int e = check_src(gx, "u3d_conv3d(gx)")) return e;
        U3D_REQUIRE(gx->C0 + gx->C1 == Cout, "u3d_conv3d: gx must have Cout channels");
        p.gx = *gx;
    } else {
        p.gx = *src;
    }
    p.has_gx = gx != nullptr;
    p.wp = packed_w;
    p.out = out;
    p.out_stats = out_stats;
    p.gstats = gstats;
    p.N = N, p.D = D, p.H = H, p.W = W, p.Cout = Cout;
    const int Cin = src->C0 + src->C1;
    p.nchunks = cdiv(Cin, 16);
    p.ntot = cdiv(Cout, 32);
    p.tz = cdiv(D, cv::TZ), p.ty = cdiv(H, cv::TY), p.tx = cdiv(W, cv::TX);
    p.relu = relu;
    p.vec = src_vec_ok(src) ? 1 : 0;
    p.stagger = 0;
    const long long ntiles = (long long)N * p.tz * p.ty * p.tx;
    // BN = 64 halves the A-tile restaging; use it when there are enough blocks to fill 256 CUs anyway
    const bool nt2 = (p.ntot % 2 == 0) && (ntiles * (p.ntot / 2) >= 512);
    p.ncb = nt2 ? p.ntot / 2 : p.ntot;
    const long long nblk = ntiles * p.ncb;
    U3D_REQUIRE(nblk < (1ll << 31), "u3d_conv3d: grid too large");
    // quarter of a chunk period when the CU is full: one wave's MFMA time per chunk = 54 steps * 8*NT MFMAs * 64 cycles
    if (g_u3d_tune[0] && nblk >= 512 && p.nchunks >= 1) p.stagger = 54 * 8 * (nt2 ? 2 : 1) * 64;
    const size_t shmem = cv::LDS_FLOATS * sizeof(float);
    const bool vec = p.vec != 0;
    const dim3 grid((unsigned)nblk), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (nt2 && vec)
        hipLaunchKernelGGL((conv3d_mfma_kernel<2, true>), grid, block, shmem, st, p);
    else if (nt2)
        hipLaunchKernelGGL((conv3d_mfma_kernel<2, false>), grid, block, shmem, st, p);
    else if (vec && g_u3d_tune[1] == 1)
        hipLaunchKernelGGL((conv3d_mfma_kernel<1, true, 1>), grid, block, shmem, st, p);
    else if (vec && g_u3d_tune[1] == 2)
        hipLaunchKernelGGL((conv3d_mfma_kernel<1, true, 2>), grid, block, shmem, st, p);
    else if (vec && g_u3d_tune[1] == 4)
        hipLaunchKernelGGL((conv3d_mfma_kernel<1, true, 4>), grid, block, shmem, st, p);
    else if (vec && g_u3d_tune[1] == 8)
        hipLaunchKernelGGL((conv3d_mfma_kernel<1, true, 8>), grid, block, shmem, st, p);
    else if (vec && g_u3d_tune[1] == 7)
        hipLaunchKernelGGL((conv3d_mfma_kernel<1, true, 7>), grid, block, shmem, st, p);
    else if (vec && g_u3d_tune[1] == 15)
        hipLaunchKernelGGL((conv3d_mfma_kernel<1, true, 15>), grid, block, shmem, st, p);
    else if (vec)
        hipLaunchKernelGGL((conv3d_mfma_kernel<1, true>), grid, block, shmem, st, p);
    else
        hipLaunchKernelGGL((conv3d_mfma_kernel<1, false>), grid, block, shmem, st, p);
    U3D_LAUNCH_CHECK();
    return 0;
}

static void wgrad_plan(int N, int D, int H, int W, int Cin, int Cout, WgradParams& p) {
    p.nchunks = cdiv(Cin, 32);
    p.nkb = cdiv(Cout, 32);
    p.tz = cdiv(D, wg::TZ), p.ty = cdiv(H, wg::TY), p.tx = cdiv(W, wg::TX);
    p.ntiles = N * p.tz * p.ty * p.tx;
    int S = 512 / (p.nchunks * p.nkb);
    if (S < 1) S = 1;
    if (S > p.ntiles) S = p.ntiles;
    p.tps = cdiv(p.ntiles, S);
    p.S = cdiv(p.ntiles, p.tps);
}

extern "C" size_t u3d_wgrad_workspace_floats(int N, int D, int H, int W, int Cin, int Cout) {
    WgradParams p;
    wgrad_plan(N, D, H, W, Cin, Cout, p);
    return (size_t)p.S * p.nchunks * p.nkb * 27 * 1024;
}

static int wgrad_set_lds_once(int device) {
    static bool done[64] = {false};
    if (device >= 0 && device < 64 && done[device]) return 0;
    U3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_wgrad_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, wg::LDS_FLOATS * sizeof(float)));
    U3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_wgrad_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, wg::LDS_FLOATS * sizeof(float)));
    if (device >= 0 && device < 64) done[device] = true;
    return 0;
}

extern "C" int u3d_conv3d_wgrad(int device, u3d_stream_t stream, const u3d_src_t* src, const float* dz, float* dw,
                                int N, int D, int H, int W, int Cout, float* workspace, size_t workspace_floats) {
    if (int e = u3d_enter(device)) return e;
    if (int e = check_src(src, "u3d_conv3d_wgrad")) return e;
    U3D_REQUIRE(dz && dw && workspace && N > 0 && D > 0 && H > 0 && W > 0 && Cout > 0, "u3d_conv3d_wgrad: bad argument");
    U3D_REQUIRE((long long)N * D * H * W < (1ll << 31), "u3d_conv3d_wgrad: N*D*H*W must be < 2^31");
    WgradParams p;
    const int Cin = src->C0 + src->C1;
    wgrad_plan(N, D, H, W, Cin, Cout, p);
    const size_t need = (size_t)p.S * p.nchunks * p.nkb * 27 * 1024;
    if (workspace_floats < need)
        return u3d_set_err(U3D_EWORKSPACE, "u3d_conv3d_wgrad: workspace %zu < %zu floats", workspace_floats, need);
    p.src = *src;
    p.dz = dz;
    p.partial = workspace;
    p.N = N, p.D = D, p.H = H, p.W = W, p.Cout = Cout;
    p.vec = src_vec_ok(src) ? 1 : 0;
    p.dzvec = (Cout % 4 == 0 && ((uintptr_t)dz & 15) == 0) ? 1 : 0;
    // half a tile period: one wave's MFMA time per tile = 64 voxel pairs * 7 taps * 64 cycles
    p.stagger = (g_u3d_tune[0] && p.tps >= 2) ? 64 * 7 * 64 : 0;
    if (int e = wgrad_set_lds_once(device)) return e;
    const int nblk = p.S * p.nchunks * p.nkb;
    if (p.vec && p.dzvec)
        hipLaunchKernelGGL(conv3d_wgrad_kernel<true>, dim3(nblk), dim3(256), wg::LDS_FLOATS * sizeof(float),
                           (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(conv3d_wgrad_kernel<false>, dim3(nblk), dim3(256), wg::LDS_FLOATS * sizeof(float),
                           (hipStream_t)stream, p);
    U3D_LAUNCH_CHECK();
    const long long total = (long long)Cin * 27 * Cout;
    const int rblocks = (int)((total + 63) / 64);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rblocks), dim3(256), 0, (hipStream_t)stream, workspace, dw, p.S,
                       p.nchunks, p.nkb, Cin, Cout);
    U3D_LAUNCH_CHECK();
    return 0;
}

extern "C" int u3d_conv3d_naive(int device, u3d_stream_t stream, const u3d_src_t* src, const float* w, float* out,
                                int N, int D, int H, int W, int Cin, int Cout, int relu, int flip) {
    if (int e = u3d_enter(device)) return e;
    if (int e = check_src(src, "u3d_conv3d_naive")) return e;
    U3D_REQUIRE(w && out && Cin == src->C0 + src->C1, "u3d_conv3d_naive: bad argument");
    const long long total = (long long)N * D * H * W * Cout;
    const int blocks = (int)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
    hipLaunchKernelGGL(conv3d_naive_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *src, w, out, N, D, H,
                       W, Cin, Cout, relu, flip);
    U3D_LAUNCH_CHECK();
    return 0;
}
