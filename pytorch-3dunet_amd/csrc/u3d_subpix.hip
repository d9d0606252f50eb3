// u3d_subpix.hip — 3x3x3 convolution over a NEAREST-2x-UPSAMPLED tensor without the upsampled work.
//
// The decoder's first SingleConv convolves cat(skip, interpolate(low, nearest)) (buildingblocks.py:491,:614,:56).  For the
// upsampled half, full-res input voxel v reads low-res voxel v >> 1, so for an output voxel 2j + p (p = parity, per
// dimension) the three taps t = 0,1,2 at full-res positions 2j + p + t - 1 hit only TWO low-res voxels:
//      p = 0:  t=0 -> j-1,  t=1,2 -> j          p = 1:  t=0,1 -> j,  t=2 -> j+1
// Taps that hit the same low-res voxel are added up ONCE in the weights (pack kernel), so each of the 8 output parity
// classes is a 2x2x2 convolution over the low-res grid: 8 instead of 27 multiply-adds per (voxel, cin, cout) — 0.296 of
// the FLOPs, same result up to fp32 association.  Zero padding carries over: full-res positions outside the volume
// map exactly to low-res positions outside the low-res volume.
//
// Kernel (implicit GEMM on v_mfma_f32_32x32x2_f32, same pipeline as conv3d_mfma_kernel in u3d_conv.hip): a block of 4 waves
// owns a 4x4x8 LOW-RES tile (wave w: the 4(y) x 8(x) voxels at z0 + w) x 32 output channels and keeps the accumulators
// of ALL EIGHT parity classes (8 x 16 registers): the 6x6x10 low-res halo tile of a 16-channel chunk is staged once (two
// LDS buffers, LDS flags instead of barriers, GroupNorm affine fused) and serves every class.  The k-loop walks the 27
// halo offsets x 2 channel octets; at offset h only the classes that use it issue MFMAs (1, 2, 4 or 8 of them: per
// dimension offset 0 belongs to parity 0, offset 2 to parity 1, offset 1 to both), each with its own combined-weight
// fragment: 512 MFMAs and 54 A-fragment reads per chunk and wave.  The B fragments form one flat stream of 128 per
// chunk, fetched 7 fragments (1800 MFMA cycles) ahead through an 8-slot register ring.
//
// Output: plain partial sums at full resolution (no ReLU, no statistics) — the caller adds the skip half with
// u3d_conv3d_residual, whose epilogue applies ReLU and the GroupNorm statistics to the sum.
#include "u3d_subpix.h"

extern int g_u3d_tune[24];  // u3d_set_tuning (csrc/u3d_conv.hip); key 15 = 1: sub-pixel weight gradient without the constant-offset B loads


struct SubpixParams {
    const float* low;     // (N, D1, H1, W1, C1)
    const float* affine;  // optional GroupNorm (a,b) rows of the C1 channels: affine[n * aff_nstride + 2*c]
    long long aff_nstride;
    const float* wp;
    float* out;           // (N, Do, Ho, Wo, Cout): 2*D1.. (nearest upsampling) or 2*D1-1.. (transposed convolution)
    int N, D1, H1, W1, C1, Cout, Do, Ho, Wo;
    int tz, ty, tx, nchunks, ncb;
    int ksplit, cps;        // split-K on small grids: ksplit blocks per (tile, channel block), each over a run of cps chunks,
    long long part_stride;  // writing its partial sums to out + run * part_stride (summed by sum_partials_kernel)
    int oz, oy, ox;         // round 5 (u3d_subpixel_conv_fwd_win): full-res voxel u of the 2x grid is written to out[u + o] — a decoder
                            // level that upsamples n -> 2n + 1 is the exact 2x upsampling shifted by one (see the entry point)
};

__device__ __forceinline__ void sp_flag_signal(int* c, int lane) {
    asm volatile("" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void sp_flag_wait(int* c, int target) {
    while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(2);
    asm volatile("" ::: "memory");
}

template <class S>
__global__ __launch_bounds__(256, 2) void subpixel_fwd_kernel(const SubpixParams p) {
    using namespace sp;
    using SC = Sch<S>;
    constexpr int NFRAG = NFRAG_OF<S>, RING = S::RING, NACT = NACT_OF<S>;
    // prefetch schedule in units of ACTIVE k-steps: halo loads early, halo stores late, affine rows a few steps before
    constexpr int LE = NACT >= 2 * NIT + 8 ? 2 : 1;
    constexpr int KS0 = NACT >= 40 ? 30 : NACT - NIT - 1;
    constexpr int KA = KS0 >= 6 ? KS0 - 6 : 0;
    static_assert(LE * (NIT - 1) < KS0 && KS0 + NIT <= NACT, "prefetch schedule must fit the k-loop");
    __shared__ __attribute__((aligned(16))) float lds[2 * TILE_FLOATS];
    __shared__ int cnt[16];  // [0,1] full[buf], [2,3] freed[buf]
    const int t = threadIdx.x;
    const int l = t & 63, w = t >> 6, m = l & 31, h = l >> 5;
    __builtin_amdgcn_s_setprio(3);
    if (t < 16) cnt[t] = 0;
    __syncthreads();  // the only rendezvous of the kernel

    int logical = u3d_xcd_remap(blockIdx.x, gridDim.x);
    int ch0 = 0, nch = p.nchunks;  // this block's run of chunks
    float* const outp = p.out + (size_t)(p.ksplit > 1 ? logical % p.ksplit : 0) * p.part_stride;
    if (p.ksplit > 1) {
        ch0 = (logical % p.ksplit) * p.cps;
        nch = min(p.cps, p.nchunks - ch0);
        logical /= p.ksplit;
    }
    const int cb = logical % p.ncb;
    int tile = logical / p.ncb;
    const int txi = tile % p.tx;
    tile /= p.tx;
    const int tyi = tile % p.ty;
    tile /= p.ty;
    const int tzi = tile % p.tz;
    const int n = tile / p.tz;
    const int z0 = tzi * TZ, y0 = tyi * TY, x0 = txi * TX;
    const int D1 = p.D1, H1 = p.H1, W1 = p.W1, C1 = p.C1;

    // ---- per-thread staging descriptors (constant across chunks)
    int ldsoff[NIT], gv[NIT];
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
        const int gz = z0 - 1 + hz, gy = y0 - 1 + hy, gx = x0 - 1 + hx;
        const bool ok = in && gz >= 0 && gz < D1 && gy >= 0 && gy < H1 && gx >= 0 && gx < W1;
        gv[it] = ok ? ((n * D1 + gz) * H1 + gy) * W1 + gx : -1;
    }

    f32x16 acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

    // A-fragment base: lane (m,h) -> low-res voxel (zl = w, yl = m>>3, xl = m&7), channels 4h..4h+3 of an octet
    const int abase = w * PS + (m >> 3) * RS + (m & 7) * CS + 4 * h;
    const int wstep = p.ncb * 64;
    const f32x4* wq = reinterpret_cast<const f32x4*>(p.wp) + (size_t)cb * 64 + (size_t)ch0 * NFRAG * wstep;  // uniform
    f32x4 bq[RING];
#pragma unroll
    for (int k = 0; k < RING - 1; ++k) bq[k] = wq[(size_t)k * wstep + l];

    struct ChunkSrc {
        const float* base;
        bool cok;
        const float* ap;
        f32x4 lo, hi;  // raw affine rows (a0,b0,a1,b1), (a2,b2,a3,b3)
    };
    auto chunk_src = [&](int ch, bool live) {
        ChunkSrc c;
        const int cq = ch * CC + 4 * q;
        c.cok = live && cq < C1;
        c.base = c.cok ? p.low + cq : p.low;
        c.ap = p.affine ? p.affine + (size_t)n * p.aff_nstride + (size_t)(c.cok ? cq : 0) * 2 : nullptr;
        c.lo = f32x4{1.f, 0.f, 1.f, 0.f};
        c.hi = f32x4{1.f, 0.f, 1.f, 0.f};
        return c;
    };
    auto load_affine_rows = [&](ChunkSrc& c) {
        if (c.ap) {
            c.lo = *reinterpret_cast<const f32x4*>(c.ap);
            c.hi = *reinterpret_cast<const f32x4*>(c.ap + 4);
        }
    };
    auto halo_load = [&](const ChunkSrc& c, int it) {  // branch-free: clamped address, validity applied at the store
        const int idx = (c.cok && gv[it] >= 0) ? gv[it] : 0;
        return *reinterpret_cast<const f32x4*>(c.base + (size_t)idx * C1);
    };
    auto halo_store = [&](float* buf, const ChunkSrc& c, int it, f32x4 raw) {
        const bool ok = c.cok && gv[it] >= 0;
        f32x4 val = {fmaf(raw[0], c.lo[0], c.lo[1]), fmaf(raw[1], c.lo[2], c.lo[3]), fmaf(raw[2], c.hi[0], c.hi[1]),
                     fmaf(raw[3], c.hi[2], c.hi[3])};
#pragma unroll
        for (int e = 0; e < 4; ++e) val[e] = ok ? val[e] : 0.f;  // padding stays exactly 0
        *reinterpret_cast<f32x4*>(&buf[ldsoff[it]]) = val;
    };

    // ---- prologue: stage chunk 0 into buffer 0
    {
        ChunkSrc c0 = chunk_src(ch0, true);
        load_affine_rows(c0);
        f32x4 v[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) v[it] = halo_load(c0, it);
#pragma unroll
        for (int it = 0; it < NIT; ++it) halo_store(lds, c0, it, v[it]);
    }
    sp_flag_signal(&cnt[0], l);

    for (int ch = 0; ch < nch; ++ch) {  // ch counts within the block's run; the source chunk is ch0 + ch
        const bool has_next = ch + 1 < nch;
        const int b = ch & 1;
        const float* cur = lds + b * TILE_FLOATS;
        float* nxt = lds + (b ^ 1) * TILE_FLOATS;
        ChunkSrc cn = chunk_src(ch0 + ch + 1, has_next);
        f32x4 v[NIT];
        sp_flag_wait(&cnt[b], 4 * (ch / 2 + 1));  // all four waves have staged chunk ch
        __builtin_amdgcn_s_setprio(0);

        f32x4 aq[2];
        {
            constexpr int st0 = SC::act(0);
            constexpr int aoff0 = ((st0 >> 1) / 9) * PS + (((st0 >> 1) / 3) % 3) * RS + ((st0 >> 1) % 3) * CS + 8 * (st0 & 1);
            aq[0] = *reinterpret_cast<const f32x4*>(&cur[abase + aoff0]);
        }
        const f32x4* wch = wq + ((size_t)ch * NFRAG + RING - 1) * wstep;  // fragment (chunk, f) + RING-1
        static_for<0, NACT>([&](auto ic) {
            constexpr int k = decltype(ic)::value;
            constexpr int st = SC::act(k);
            constexpr int tap = st >> 1;
            constexpr int NC = SC::ncls(tap);
            constexpr int pre = SC::prefix(st);
            if constexpr (k % LE == 0 && k / LE < NIT) v[k / LE] = halo_load(cn, k / LE);
            if constexpr (k == KA) load_affine_rows(cn);
            static_for<0, NC>([&](auto ii) {
                constexpr int i = decltype(ii)::value;
                constexpr int f = pre + i;
                constexpr int ci = SC::cls(tap, i);
                bq[(f + RING - 1) % RING] = wch[(size_t)f * wstep + l];
                __builtin_amdgcn_sched_barrier(0);
                acc[ci] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[k & 1][0], bq[f % RING][0], acc[ci], 0, 0, 0);
                acc[ci] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[k & 1][1], bq[f % RING][1], acc[ci], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (i == 0) {
                    // the non-MFMA work of the step issues while the MFMA pipe is busy with the two above
                    if constexpr (k + 1 < NACT) {
                        constexpr int st1 = SC::act(k + 1);
                        constexpr int tap1 = st1 >> 1, s1 = st1 & 1;
                        constexpr int aoff = (tap1 / 9) * PS + ((tap1 / 3) % 3) * RS + (tap1 % 3) * CS + 8 * s1;
                        aq[(k + 1) & 1] = *reinterpret_cast<const f32x4*>(&cur[abase + aoff]);
                    }
                    if constexpr (k >= KS0 && k < KS0 + NIT) {
                        if (has_next) {
                            // the other buffer is free once all four waves have finished the k-loop of chunk ch-1
                            if constexpr (k == KS0) sp_flag_wait(&cnt[2 + (b ^ 1)], 4 * ((ch + 1) / 2));
                            halo_store(nxt, cn, k - KS0, v[k - KS0]);
                            if constexpr (k == KS0 + NIT - 1) sp_flag_signal(&cnt[b ^ 1], l);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                acc[ci] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[k & 1][2], bq[f % RING][2], acc[ci], 0, 0, 0);
                acc[ci] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[k & 1][3], bq[f % RING][3], acc[ci], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        sp_flag_signal(&cnt[2 + b], l);  // this wave no longer reads buffer b
    }
    __builtin_amdgcn_s_setprio(3);

    // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane&31 (cout), row = (r&3) + 8*(r>>2) + 4*(lane>>5); M-tile
    //      row -> (y = row>>3, x = row&7).  A 4x4 transpose inside every lane quad (two DPP stages) leaves lane j of
    //      quad k with the 4 consecutive channels 4k..4k+3 of voxel x = j + 4h: 16-byte stores.
    const int cq = (l >> 2) & 7, vl = (l & 3) + 4 * h;
    const bool odd = (l & 1) != 0, hi2 = (l & 2) != 0;
    auto xlane = [](float v, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), decltype(ctrl)::value, 0xF, 0xF, true));
    };
    using X1 = std::integral_constant<int, 0xB1>;  // quad_perm [1,0,3,2]: value of lane ^ 1
    using X2 = std::integral_constant<int, 0x4E>;  // quad_perm [2,3,0,1]: value of lane ^ 2
    const int co = cb * 32 + 4 * cq;
    const bool cok = co < p.Cout;
    const int z = z0 + w, x = x0 + vl;
    const int D = p.Do, H = p.Ho, W = p.Wo;
    // Fast path (tile and channel block fully inside): store the accumulators in their native layout — lane (m, h) holds
    // output channel m of voxel x = (r&3) + 4h, so one 4-byte store per register writes two full 128-byte channel rows.  The
    // address is a UNIFORM base (scalar arithmetic) plus a per-lane constant: no VALU at all, against 4 VALU per value for
    // the in-register transposition below (a VALU instruction issued beside another wave's MFMA stream waits ~45 cycles).
    const bool full = 2 * (z0 + TZ) - 1 + p.oz < D && 2 * (y0 + TY) - 1 + p.oy < H && 2 * (x0 + TX) - 1 + p.ox < W && cb * 32 + 32 <= p.Cout;
    if (full) {
        const int wz = __builtin_amdgcn_readfirstlane(z);
        const int lane_off = 8 * h * p.Cout + m;  // x = x0 + (r&3) + 4h  ->  full-res 2x: 8h voxels further
        static_for<0, 8>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            constexpr int pz = c >> 2, py = (c >> 1) & 1, px = c & 1;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const size_t vu = ((size_t)(n * D + 2 * wz + pz + p.oz) * H + 2 * (y0 + (r >> 2)) + py + p.oy) * W + 2 * (x0 + (r & 3)) + px + p.ox;
                float* ob = outp + vu * p.Cout + cb * 32;
                ob[lane_off] = acc[c][r];
            }
        });
        return;
    }
    static_for<0, 8>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        constexpr int pz = c >> 2, py = (c >> 1) & 1, px = c & 1;
#pragma unroll
        for (int bi = 0; bi < 4; ++bi) {
            const float a0 = acc[c][4 * bi + 0], a1 = acc[c][4 * bi + 1], a2 = acc[c][4 * bi + 2], a3 = acc[c][4 * bi + 3];
            const float t0 = xlane(a1, X1{}), t1 = xlane(a0, X1{}), t2 = xlane(a3, X1{}), t3 = xlane(a2, X1{});
            const float c0 = odd ? t0 : a0, c1 = odd ? a1 : t1, c2 = odd ? t2 : a2, c3 = odd ? a3 : t3;
            const float u0 = xlane(c2, X2{}), u2 = xlane(c0, X2{}), u1 = xlane(c3, X2{}), u3 = xlane(c1, X2{});
            const f32x4 val = {hi2 ? u0 : c0, hi2 ? u1 : c1, hi2 ? c2 : u2, hi2 ? c3 : u3};
            const int y = y0 + bi;
            if (cok && 2 * z + pz + p.oz < D && 2 * y + py + p.oy < H && 2 * x + px + p.ox < W) {
                const size_t vidx = ((size_t)(n * D + 2 * z + pz + p.oz) * H + 2 * y + py + p.oy) * W + 2 * x + px + p.ox;
                *reinterpret_cast<f32x4*>(outp + vidx * p.Cout + co) = val;
            }
        }
    });
}

// out = sum of the split-K runs, in a fixed order
__global__ void sum_partials_kernel(const float* __restrict__ part, long long stride, int ks, float* __restrict__ out,
                                    long long n4) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        f32x4 a = reinterpret_cast<const f32x4*>(part)[i];
        for (int k = 1; k < ks; ++k) a += reinterpret_cast<const f32x4*>(part + (size_t)k * stride)[i];
        reinterpret_cast<f32x4*>(out)[i] = a;
    }
}

// =================================================================================================================
// Data gradient with respect to the low-res tensor.  Block = 4 waves on a 2x4x8 low-res tile x 64 low-res channels: wave
// w owns z-plane (w & 1) and the 32-channel half (w >> 1).  The 6x10x18 full-res dz region of a 16-channel chunk lives
// in ONE LDS buffer (70 KB, two blocks per CU) laid out [z][y parity][y/2][x parity][x/2][16]: for a fixed tap the 32
// voxels 2i + d of an M-tile then sit at unit stride in y/2 and x/2, so the A-fragment reads are the same conflict-free
// pattern as in the forward kernels.  The next chunk's 17 items per thread are fetched into registers during the 128
// fragment steps (4 MFMAs each) of the current chunk and written to LDS between two k-loops, after all four waves have
// signalled that they are done reading.  Epilogue: transposed 16-byte stores of dlow and the two GroupNorm-backward sums
// (sum dlow, sum dlow * x_low — equal to the full-resolution sums of the reference's formulation).
struct SubpixDgradParams {
    const float* dz;    // (N, 2*D1, 2*H1, 2*W1, K)
    const float* wp;
    const float* xlow;  // (N, D1, H1, W1, C1): forward input of the upsampled half (for the sums), may be null
    float* out;         // (N, D1, H1, W1, C1)
    double* gstats;     // [reps][N][C1][2] += (sum dlow, sum dlow*x), may be null
    int greps;          // replica rows of gstats (block b adds to row b % greps; u3d_subpixel_conv_dgrad_reps)
    int N, D1, H1, W1, C1, K;
    int tz, ty, tx, nchunks, ncb, ntot;
    // round 5 (u3d_subpixel_conv_dgrad_win): voxel u of the 2x grid is dz[u + o] of a (Dd, Hd, Wd) tensor and counts only if u >= l
    // (below l: zero — those outputs belong to the boundary slab, whose gradient comes from the box launches)
    int Dd, Hd, Wd, oz, oy, ox, lz, ly, lx;
};

__global__ __launch_bounds__(256, 2) void subpixel_dgrad_kernel(const SubpixDgradParams p) {
    using namespace spd;
    using sp::static_for;
    extern __shared__ __attribute__((aligned(16))) float lds[];  // REGION_FLOATS + 4 (dummy slot) + red[4][32][2] + cnt[16]
    float* red = lds + REGION_FLOATS + 4;
    int* cnt = reinterpret_cast<int*>(red + 4 * 32 * 2);
    const int t = threadIdx.x;
    const int l = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), m = l & 31, h = l >> 5;
    const int zi = w & 1, ni = w >> 1;
    __builtin_amdgcn_s_setprio(3);
    if (t < 16) cnt[t] = 0;
    __syncthreads();

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
    const int D1 = p.D1, H1 = p.H1, W1 = p.W1, K = p.K;
    const int D = 2 * D1, H = 2 * H1, W = 2 * W1;

    int ldsoff[NIT], gv[NIT];
    const int q = t & 3;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int item = t + 256 * it;
        const int vox = item >> 2;
        const bool in = item < NITEMS;
        const int rz = vox / (RY * RX);
        const int rem = vox - rz * (RY * RX);
        const int ry = rem / RX;
        const int rx = rem - ry * RX;
        ldsoff[it] = in ? ((rz * 2 + (ry & 1)) * (RY / 2) + (ry >> 1)) * ROW + ((rx & 1) * (RX / 2) + (rx >> 1)) * 16 + 4 * q
                        : REGION_FLOATS;
        const int gz = 2 * z0 - 1 + rz, gy = 2 * y0 - 1 + ry, gx = 2 * x0 - 1 + rx;
        const bool ok = in && gz >= p.lz && gz < D && gy >= p.ly && gy < H && gx >= p.lx && gx < W;
        gv[it] = ok ? ((n * p.Dd + gz + p.oz) * p.Hd + gy + p.oy) * p.Wd + gx + p.ox : -1;
    }

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    // lane (m,h) -> low-res voxel (zl = zi, yl = m>>3, xl = m&7); tap d adds a compile-time offset
    const int abase = zi * (2 * RY) * ROW + (m >> 3) * ROW + (m & 7) * 16 + 4 * h;
    const int ntg = cb * 2 + ni;
    const int wstep = p.ntot * 64;
    const f32x4* wq = reinterpret_cast<const f32x4*>(p.wp) + (size_t)ntg * 64;  // wave-uniform base; lanes add l
    f32x4 bq[RING];
#pragma unroll
    for (int k = 0; k < RING - 1; ++k) bq[k] = wq[(size_t)k * wstep + l];

    auto halo_load = [&](int ch, bool live, int it) {
        const int cq = ch * 16 + 4 * q;
        const bool ok = live && cq < K && gv[it] >= 0;
        return *reinterpret_cast<const f32x4*>(p.dz + (size_t)(ok ? gv[it] : 0) * K + (ok ? cq : 0));
    };
    auto halo_store = [&](int ch, int it, f32x4 raw) {
        const bool ok = ch * 16 + 4 * q < K && gv[it] >= 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) raw[e] = ok ? raw[e] : 0.f;  // outside the volume / beyond the last channel: 0
        *reinterpret_cast<f32x4*>(&lds[ldsoff[it]]) = raw;
    };
    auto aoff = [](int f) constexpr {
        const int tap = f >> 1, s = f & 1;
        const int dz = tap >> 4, dy = (tap >> 2) & 3, dx = tap & 3;  // d + 1
        return ((dz * 2 + (dy & 1)) * (RY / 2) + (dy >> 1)) * ROW + ((dx & 1) * (RX / 2) + (dx >> 1)) * 16 + 8 * s;
    };

    {
        f32x4 v[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) v[it] = halo_load(0, true, it);
#pragma unroll
        for (int it = 0; it < NIT; ++it) halo_store(0, it, v[it]);
    }
    sp_flag_signal(&cnt[0], l);

    for (int ch = 0; ch < p.nchunks; ++ch) {
        const bool has_next = ch + 1 < p.nchunks;
        f32x4 v[NIT];
        sp_flag_wait(&cnt[0], 4 * (ch + 1));  // all four waves have staged chunk ch
        __builtin_amdgcn_s_setprio(0);
        f32x4 aq[2];
        aq[0] = *reinterpret_cast<const f32x4*>(&lds[abase + aoff(0)]);
        const f32x4* wch = wq + ((size_t)ch * NFRAG + RING - 1) * wstep;
        static_for<0, NFRAG>([&](auto ic) {
            constexpr int f = decltype(ic)::value;
            if constexpr (f % 4 == 0 && f / 4 < NIT) v[f / 4] = halo_load(ch + 1, has_next, f / 4);
            bq[(f + RING - 1) % RING] = wch[(size_t)f * wstep + l];
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[f & 1][0], bq[f % RING][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[f & 1][1], bq[f % RING][1], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (f + 1 < NFRAG) aq[(f + 1) & 1] = *reinterpret_cast<const f32x4*>(&lds[abase + aoff(f + 1)]);
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[f & 1][2], bq[f % RING][2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[f & 1][3], bq[f % RING][3], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        });
        sp_flag_signal(&cnt[1], l);  // this wave no longer reads the buffer
        if (has_next) {
            __builtin_amdgcn_s_setprio(3);
            sp_flag_wait(&cnt[1], 4 * (ch + 1));
#pragma unroll
            for (int it = 0; it < NIT; ++it) halo_store(ch + 1, it, v[it]);
            sp_flag_signal(&cnt[0], l);
        }
    }
    __builtin_amdgcn_s_setprio(3);

    // ---- epilogue (C/D layout and the in-register 4x4 transpose as in the forward kernel)
    const int cq = (l >> 2) & 7, vl = (l & 3) + 4 * h;
    const bool odd = (l & 1) != 0, hi2 = (l & 2) != 0;
    auto xlane = [](float v_, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v_), decltype(ctrl)::value, 0xF, 0xF, true));
    };
    using X1 = std::integral_constant<int, 0xB1>;
    using X2 = std::integral_constant<int, 0x4E>;
    const int co = ntg * 32 + 4 * cq;
    const bool cok = co < p.C1;
    const int z = z0 + zi, x = x0 + vl;
    const bool want = p.gstats != nullptr;
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int bi = 0; bi < 4; ++bi) {
        const float a0 = acc[4 * bi + 0], a1 = acc[4 * bi + 1], a2 = acc[4 * bi + 2], a3 = acc[4 * bi + 3];
        const float t0 = xlane(a1, X1{}), t1 = xlane(a0, X1{}), t2 = xlane(a3, X1{}), t3 = xlane(a2, X1{});
        const float c0 = odd ? t0 : a0, c1 = odd ? a1 : t1, c2 = odd ? t2 : a2, c3 = odd ? a3 : t3;
        const float u0 = xlane(c2, X2{}), u2 = xlane(c0, X2{}), u1 = xlane(c3, X2{}), u3 = xlane(c1, X2{});
        const f32x4 val = {hi2 ? u0 : c0, hi2 ? u1 : c1, hi2 ? c2 : u2, hi2 ? c3 : u3};
        const int y = y0 + bi;
        if (cok && z < D1 && y < H1 && x < W1) {
            const size_t off = ((size_t)((n * D1 + z) * H1 + y) * W1 + x) * p.C1 + co;
            *reinterpret_cast<f32x4*>(p.out + off) = val;
            if (want) {
                s1 += val;
                s2 += val * *reinterpret_cast<const f32x4*>(p.xlow + off);
            }
        }
    }
    if (want) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float a = s1[e], b = s2[e];
#pragma unroll
            for (int mask : {1, 2, 32}) {  // lanes of one channel quad: x = (l&3) + 4*(l>>5)
                a += __shfl_xor(a, mask);
                b += __shfl_xor(b, mask);
            }
            if ((l & 35) == 0) {
                red[(w * 32 + 4 * cq + e) * 2 + 0] = a;
                red[(w * 32 + 4 * cq + e) * 2 + 1] = b;
            }
        }
        __syncthreads();
        if (t < 128) {  // (channel half, channel, which sum): the two z-planes in a fixed order, one f64 atomic each
            const int nh = t >> 6, c = (t >> 1) & 31, j = t & 1;
            const int cc = (cb * 2 + nh) * 32 + c;
            if (cc < p.C1) {
                const float sum = red[((nh * 2 + 0) * 32 + c) * 2 + j] + red[((nh * 2 + 1) * 32 + c) * 2 + j];
                u3d_atomic_add_f64(&p.gstats[((size_t)(blockIdx.x % (unsigned)p.greps) * p.N * p.C1 + (size_t)n * p.C1 + cc) * 2 + j], (double)sum);
            }
        }
    }
}

// =================================================================================================================
// Weight gradient of the upsampled half.  With v = 2j + p (parity class p) reading low-res voxel j + p - 1 + e, e in {0,1}
// per dimension:   dWc[p][e][c][k] = sum_j g_low[j + p - 1 + e][c] * dz[2j + p][k]      (64 matrices (p, e))
// and the reference's dw[k][c][t] = sum_p dWc[p][e(p,t)]  (p=0: t=0 -> e=0, t=1,2 -> e=1;  p=1: t=0,1 -> e=0, t=2 -> e=1), 8 of
// the 64 per tap — formed by the reduce kernel, which also sums the split-K partials in a fixed order.
//
// Block = 8 waves = the 8 parity classes on a 2x4x8 low-res tile x 32 low-res channels x 32 dz channels.  A dz voxel
// belongs to exactly one class, so the B operand (dz[2j+p][k], one float per lane) never goes through LDS: every wave
// fetches its own 32 voxel pairs of the NEXT tile straight into registers while it computes the current one.  The A operand
// (g_low shifted by the tap, GroupNorm affine fused) is read from a double-buffered 4x6x10 halo tile in LDS like in
// conv3d_wgrad_kernel.  Per tile and wave: 32 voxel-pair groups x 8 taps = 256 MFMAs, one barrier.
namespace spw {
constexpr int TZ = 2, TY = 4, TX = 8;
constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2;   // 4 x 6 x 10 low-res halo tile
constexpr int CS = 32, RS = HX * CS, PS = HY * RS;      // [hz][hy][hx][32 channels]
constexpr int G_FLOATS = HZ * PS;                       // 7680 floats = 30 KB per buffer
constexpr int NTHR = 512;
constexpr int NITEMS = HZ * HY * HX * 8;                // 1920 float4 items
constexpr int NIT = (NITEMS + NTHR - 1) / NTHR;         // 4
constexpr int NGRP = TZ * TY * TX / 2;                  // 32 voxel pairs
}  // namespace spw

struct SubpixWgradParams {
    const float* low;     // (N, D1, H1, W1, C1)
    const float* affine;  // optional, sample stride aff_nstride floats
    long long aff_nstride;
    const float* dz;      // (N, 2*D1, 2*H1, 2*W1, K)
    float* partial;       // [S][nchunks][nkb][64 (p, e)][32 c][32 k]
    int N, D1, H1, W1, C1, K;
    int nchunks, nkb, S, tz, ty, tx, ntiles, tps;
    int Dd, Hd, Wd, oz, oy, ox, lz, ly, lx;  // dz window as in SubpixDgradParams (general B loads only: never with FULL)
};

// FULL (round 4): every tile lies inside the volume and K is a multiple of 32 — the address of a lane's dz element is then
// (uniform tile / group offset, SALU) + (per-lane constant): ONE vector add per B load instead of the ~13 vector instructions of the
// coordinate arithmetic and bounds tests (the kernel issued 2.4 vector instructions per MFMA, the plain weight gradient 1.9; fp32
// MFMAs lose ~5 pipe cycles per vector instruction of any co-resident wave).
template <bool FULL>
__global__ __launch_bounds__(512, 2) void subpixel_wgrad_kernel(const SubpixWgradParams p) {
    using namespace spw;
    using sp::static_for;
    __shared__ __attribute__((aligned(16))) float lds[2 * G_FLOATS + 4];  // two halo buffers + a dummy float4 slot
    __builtin_amdgcn_s_setprio(3);
    const int t = threadIdx.x;
    const int l = t & 63, i = l & 31, h = l >> 5;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int pz = w >> 2, py = (w >> 1) & 1, px = w & 1;

    const int logical = u3d_xcd_remap(blockIdx.x, gridDim.x);
    const int kb = logical % p.nkb;
    const int chunk = (logical / p.nkb) % p.nchunks;
    const int s = logical / (p.nkb * p.nchunks);
    const int D1 = p.D1, H1 = p.H1, W1 = p.W1, C1 = p.C1, K = p.K;
    const int H = 2 * H1, W = 2 * W1;

    f32x16 acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;

    // ---- staging of the g halo tile: this thread's items (halo voxel, channel quad), constant across tiles
    const int q = t & 7;
    const int cq = chunk * 32 + 4 * q;
    const bool cok = cq < C1;
    int ihv[NIT], loff[NIT];  // halo coordinates packed as hz | hy << 8 | hx << 16 (hz = 100: tail item, never inside)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int vox = (t >> 3) + 64 * it;
        const bool in = vox < HZ * HY * HX;
        const int hz = vox / (HY * HX);
        const int rem = vox - hz * (HY * HX);
        const int hy = rem / HX;
        const int hx = rem - hy * HX;
        loff[it] = in ? vox * CS + 4 * q : 2 * G_FLOATS;  // tail items -> dummy slot
        ihv[it] = (in ? hz : 100) | (hy << 8) | (hx << 16);
    }
    struct Tile {
        int n, z0, y0, x0;
    };
    auto tile_of = [&](int tile) {
        Tile c;
        c.x0 = (tile % p.tx) * TX;
        tile /= p.tx;
        c.y0 = (tile % p.ty) * TY;
        tile /= p.ty;
        c.z0 = (tile % p.tz) * TZ;
        c.n = tile / p.tz;
        return c;
    };
    auto g_load = [&](const Tile& c, int it) {
        const int gz = c.z0 - 1 + (ihv[it] & 255), gy = c.y0 - 1 + ((ihv[it] >> 8) & 255), gx = c.x0 - 1 + (ihv[it] >> 16);
        const bool ok = cok & ((unsigned)gz < (unsigned)D1) & ((unsigned)gy < (unsigned)H1) & ((unsigned)gx < (unsigned)W1);
        const int idx = ok ? ((c.n * D1 + gz) * H1 + gy) * W1 + gx : 0;
        return *reinterpret_cast<const f32x4*>(p.low + (size_t)idx * C1 + (cok ? cq : 0));
    };
    auto g_store = [&](float* buf, const Tile& c, int it, f32x4 raw, const f32x4& ga, const f32x4& gb) {
        const int gz = c.z0 - 1 + (ihv[it] & 255), gy = c.y0 - 1 + ((ihv[it] >> 8) & 255), gx = c.x0 - 1 + (ihv[it] >> 16);
        const bool ok = cok & ((unsigned)gz < (unsigned)D1) & ((unsigned)gy < (unsigned)H1) & ((unsigned)gx < (unsigned)W1);
        f32x4 val = raw * ga + gb;
#pragma unroll
        for (int e = 0; e < 4; ++e) val[e] = ok ? val[e] : 0.f;
        *reinterpret_cast<f32x4*>(&buf[loff[it]]) = val;
    };
    auto load_affine = [&](int n, f32x4& ga, f32x4& gb) {
        ga = f32x4{1.f, 1.f, 1.f, 1.f};
        gb = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.affine && cok) {
            const float* ap = p.affine + (size_t)n * p.aff_nstride + (size_t)cq * 2;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(ap), hi = *reinterpret_cast<const f32x4*>(ap + 4);
            ga = f32x4{lo[0], lo[2], hi[0], hi[2]};
            gb = f32x4{lo[1], lo[3], hi[1], hi[3]};
        }
    };
    // ---- B operand: lane (i = dz channel, h = voxel of the pair) of group g reads dz[2(j) + p][kb*32 + i] of its own class
    const int kch = kb * 32 + i;
    const bool kok = kch < K;
    // FULL: byte offset of this lane's element relative to the (tile, group) origin dz[n][2*z0 + 2*zl][2*y0 + 2*yl][2*x0 + 4*(g&3)][0]
    const unsigned lane_off = 4u * ((unsigned)K * (unsigned)((pz * H + py) * W + 2 * h + px) + (unsigned)kch);
    const __amdgpu_buffer_rsrc_t dz_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.dz), 0, FULL ? (int)((long long)p.N * 8 * D1 * H1 * W1 * K * 4) : 0, 0x00020000);  // (FULL: < 2^31 bytes, host check)
    auto b_load = [&](const Tile& c, int g) {
        if constexpr (FULL) {
            // buffer load: (resource = dz, scalar offset = the tile / group origin, vector offset = the lane constant) — no vector
            // instruction at all for the address (a flat `base + lane_off` made the compiler form 64-bit addresses per lane)
            const int zl = g >> 4, yl = (g >> 2) & 3;
            const int vox = ((c.n * 2 * D1 + 2 * (c.z0 + zl)) * H + 2 * (c.y0 + yl)) * W + 2 * c.x0 + 4 * (g & 3);  // uniform
            return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dz_rsrc, (int)lane_off, vox * K * 4, 0));
        }
        const int zl = g >> 4, yl = (g >> 2) & 3, xl = 2 * (g & 3) + h;
        const int vz = 2 * (c.z0 + zl) + pz, vy = 2 * (c.y0 + yl) + py, vx = 2 * (c.x0 + xl) + px;
        const bool ok = kok & (c.z0 + zl < D1) & (c.y0 + yl < H1) & (c.x0 + xl < W1) & (vz >= p.lz) & (vy >= p.ly) & (vx >= p.lx);
        int vi = ok ? ((c.n * p.Dd + vz + p.oz) * p.Hd + vy + p.oy) * p.Wd + vx + p.ox : 0;
        asm volatile("" : "+v"(vi));  // opaque: an unconditional load from a clamped address, no exec-masked branch
        const float v = p.dz[(size_t)vi * K + (kok ? kch : 0)];
        return ok ? v : 0.f;
    };

    const int tile_begin = s * p.tps, tile_end = min(p.ntiles, (s + 1) * p.tps);
    float bcur[NGRP];
    f32x4 gan, gbn;
    int n_aff = -1;
    if (tile_begin < tile_end) {
        const Tile c = tile_of(tile_begin);
        load_affine(c.n, gan, gbn);
        n_aff = c.n;
        f32x4 v[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) v[it] = g_load(c, it);
#pragma unroll
        for (int g = 0; g < NGRP; ++g) bcur[g] = b_load(c, g);
#pragma unroll
        for (int it = 0; it < NIT; ++it) g_store(lds, c, it, v[it], gan, gbn);
    }
    __syncthreads();

    // A base: lane (i = channel, h = voxel of the pair) + the class offset p; tap e and group add compile-time offsets
    const int ab0 = (pz * HY + py) * RS + px * CS + h * CS + i;
    auto toff = [](int k) constexpr { return ((k >> 2) * HY + ((k >> 1) & 1)) * RS + (k & 1) * CS; };
    int cur = 0;
    __builtin_amdgcn_s_setprio(0);
    for (int tile = tile_begin; tile < tile_end; ++tile) {
        const bool has_next = tile + 1 < tile_end;
        const Tile cn = tile_of(has_next ? tile + 1 : tile);
        float* nbuf = lds + (cur ^ 1) * G_FLOATS;
        const float* gl = lds + cur * G_FLOATS;
        if (cn.n != n_aff) {
            load_affine(cn.n, gan, gbn);
            n_aff = cn.n;
        }
        f32x4 v[NIT];
        float aop[2][8];
#pragma unroll
        for (int k = 0; k < 8; ++k) aop[0][k] = gl[ab0 + toff(k)];
        static_for<0, NGRP>([&](auto gg) {
            constexpr int g = decltype(gg)::value;
            if constexpr (g % 2 == 0 && g / 2 < NIT) v[g / 2] = g_load(cn, g / 2);
            if constexpr (g >= 20 && g - 20 < NIT) g_store(nbuf, cn, g - 20, v[g - 20], gan, gbn);
            if constexpr (g + 1 < NGRP) {
                constexpr int g1 = g + 1;
                constexpr int goff = ((g1 >> 4) * HY + ((g1 >> 2) & 3)) * RS + 2 * (g1 & 3) * CS;
#pragma unroll
                for (int k = 0; k < 8; ++k) aop[g1 & 1][k] = gl[ab0 + toff(k) + goff];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(aop[g & 1][k], bcur[g], acc[k], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            bcur[g] = b_load(cn, g);  // the same pair of the NEXT tile: in flight for a whole tile before its use
        });
        __syncthreads();
        cur ^= 1;
    }
    __builtin_amdgcn_s_setprio(3);
    // partial[s][chunk][kb][w*8 + e][c][k]: D rows = c, cols = k
    float* dst = p.partial + ((size_t)((s * p.nchunks + chunk) * p.nkb + kb) * 64 + w * 8) * 1024;
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            dst[((size_t)k * 32 + row) * 32 + i] = acc[k][r];
        }
}

// fixed-order sum over the splits and over the 8 (class, tap-half) matrices that make up one original tap; writes
// dw[k][c_off + c][t] of the (K, cstride, 3,3,3) weight gradient
__global__ __launch_bounds__(256) void subpixel_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                                    int S, int nchunks, int nkb, int C1, int K,
                                                                    int cstride) {
    __shared__ float red[4][64];
    const long long total = (long long)C1 * 27 * K;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const long long idx = (long long)blockIdx.x * 64 + lane;  // (c, tap, k), k fastest
    float sum = 0.f;
    int k = 0, tap = 0, c = 0;
    if (idx < total) {
        k = (int)(idx % K);
        const long long r = idx / K;
        tap = (int)(r % 27);
        c = (int)(r / 27);
        const int t3[3] = {tap / 9, (tap / 3) % 3, tap % 3};
        const size_t base = ((size_t)((c >> 5) * nkb + (k >> 5)) * 64) * 1024 + (c & 31) * 32 + (k & 31);
        const size_t sstride = (size_t)nchunks * nkb * 64 * 1024;
        const int per = (S + 3) / 4;
        const int s0 = grp * per, s1 = min(S, s0 + per);
        for (int s = s0; s < s1; ++s) {
            float a = 0.f;
            for (int pc = 0; pc < 8; ++pc) {  // parity class (pz, py, px) and the tap half e it pairs with this tap
                int e = 0;
                for (int d = 0; d < 3; ++d) {
                    const int pd = (pc >> (2 - d)) & 1;
                    const int ed = pd == 0 ? (t3[d] == 0 ? 0 : 1) : (t3[d] == 2 ? 1 : 0);
                    e = e * 2 + ed;
                }
                a += partial[(size_t)s * sstride + base + (size_t)(pc * 8 + e) * 1024];
            }
            sum += a;
        }
    }
    red[grp][lane] = sum;
    __syncthreads();
    if (grp == 0 && idx < total)
        dw[((size_t)k * cstride + c) * 27 + tap] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// Round 6: the same reduction with every partial read ONCE.  The kernel above reads each of the 64 (class, tap-half) matrices once per
// tap it belongs to (3.4 reads per element: 216 MB of L2 traffic for 64 MB of partials, 27-32 us per level).  Here a thread owns one
// (input channel c, output channel k) element and the splits g, g + G, ..: it sums the 64 matrices over its splits in registers, folds them
// into the 27 taps (8 matrices each, ascending parity class), and the G groups of a block are added in ascending order through LDS.
// block = 32 output channels x G split groups, grid = C1 x ceil(K / 32).  Fixed order: results are run-to-run identical (they differ from
// the kernel above in the last bits: splits are summed before the fold, not after).
__global__ __launch_bounds__(512) void subpixel_wgrad_reduce_once_kernel(const float* __restrict__ partial, float* __restrict__ dw, int S,
                                                                         int nchunks, int nkb, int C1, int K, int cstride) {
    extern __shared__ float red[];  // [G][27][32]
    const int k32 = threadIdx.x & 31, grp = threadIdx.x >> 5, G = blockDim.x >> 5;
    const int c = blockIdx.x / nkb, kb = blockIdx.x - c * nkb;
    const int k = kb * 32 + k32;
    const float* base = partial + ((size_t)((c >> 5) * nkb + kb) * 64) * 1024 + (c & 31) * 32 + k32;
    const size_t sstride = (size_t)nchunks * nkb * 64 * 1024;
    float m[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) m[j] = 0.f;
    for (int sp_ = grp; sp_ < S; sp_ += G) {
        const float* ps = base + (size_t)sp_ * sstride;
#pragma unroll
        for (int j = 0; j < 64; ++j) m[j] += ps[(size_t)j * 1024];
    }
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
        const int t3[3] = {tap / 9, (tap / 3) % 3, tap % 3};
        float a = 0.f;
#pragma unroll
        for (int pc = 0; pc < 8; ++pc) {
            int e = 0;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int pd = (pc >> (2 - d)) & 1;
                e = e * 2 + (pd == 0 ? (t3[d] == 0 ? 0 : 1) : (t3[d] == 2 ? 1 : 0));
            }
            a += m[pc * 8 + e];
        }
        red[(grp * 27 + tap) * 32 + k32] = a;
    }
    __syncthreads();
    for (int o = threadIdx.x; o < 27 * 32; o += blockDim.x) {
        const int tap = o >> 5, kk = o & 31;
        float sum = 0.f;
        for (int g = 0; g < G; ++g) sum += red[(g * 27 + tap) * 32 + kk];
        if (c < C1 && kb * 32 + kk < K) dw[((size_t)(kb * 32 + kk) * cstride + c) * 27 + tap] = sum;
    }
    (void)k;
}

__global__ void pack_subpixel_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int cstride, int C1,
                                     int nchunks, int ncb, long long total) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x)
        out[idx] = sp::pack_elem(w, Cout, cstride, C1, nchunks, ncb, idx);
}

static inline int sp_cdiv(int a, int b) { return (a + b - 1) / b; }

extern "C" long long u3d_subpixel_packed_floats(int C1, int Cout) {
    if (C1 <= 0 || Cout <= 0) return 0;
    return sp::packed_floats(C1, Cout);
}

extern "C" int u3d_pack_subpixel_weights(int device, u3d_stream_t stream, const float* w, int Cout, int Cin_total,
                                         int c_off, int C1, float* packed) {
    U3D_ENTER(device);
    U3D_REQUIRE(w && packed && Cout > 0 && C1 > 0 && c_off >= 0 && c_off + C1 <= Cin_total,
                "u3d_pack_subpixel_weights: bad argument");
    const long long total = u3d_subpixel_packed_floats(C1, Cout);
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_subpixel_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w + (size_t)c_off * 27,
                       packed, Cout, Cin_total, C1, sp_cdiv(C1, 16), sp_cdiv(Cout, 32), total);
    U3D_LAUNCH_CHECK();
    return 0;
}

static void subpixel_fwd_split(long long items, int nchunks, int* ksplit, int* cps) {
    // fewer than one block per CU: split the channel reduction until there are ~2 blocks per CU (256 CUs)
    *ksplit = 1, *cps = nchunks;
    if (items >= 256 || nchunks < 2) return;
    long long ks = 512 / items;
    if (ks > nchunks) ks = nchunks;
    if (ks > 8) ks = 8;
    if (ks < 2) return;
    *cps = sp_cdiv(nchunks, (int)ks);
    *ksplit = sp_cdiv(nchunks, *cps);
}

extern "C" long long u3d_subpixel_fwd_workspace_floats(int N, int D1, int H1, int W1, int C1, int Cout) {
    if (N <= 0 || D1 <= 0 || H1 <= 0 || W1 <= 0 || C1 <= 0 || Cout <= 0) return 0;
    const long long items = (long long)N * sp_cdiv(D1, sp::TZ) * sp_cdiv(H1, sp::TY) * sp_cdiv(W1, sp::TX) * sp_cdiv(Cout, 32);
    int ks, cps;
    subpixel_fwd_split(items, sp_cdiv(C1, 16), &ks, &cps);
    return ks > 1 ? (long long)ks * N * D1 * H1 * W1 * 8 * Cout : 0;
}

static int subpixel_conv_fwd_impl(int device, u3d_stream_t stream, const float* low, const float* affine,
                                  long long affine_sample_stride, const float* packed, float* out, int N, int D1, int H1, int W1,
                                  int C1, int Cout, float* workspace, long long workspace_floats, const int* win);

extern "C" int u3d_subpixel_conv_fwd(int device, u3d_stream_t stream, const float* low, const float* affine,
                                     long long affine_sample_stride, const float* packed, float* out, int N, int D1,
                                     int H1, int W1, int C1, int Cout, float* workspace, long long workspace_floats) {
    return subpixel_conv_fwd_impl(device, stream, low, affine, affine_sample_stride, packed, out, N, D1, H1, W1, C1, Cout, workspace,
                                  workspace_floats, nullptr);
}

// ... into a WINDOW of a larger output (round 5).  F.interpolate(x, size=skip, mode="nearest") from n to 2n + 1 voxels along an axis
// (buildingblocks.py:598-614; the shipped 80 x 170 x 170 patch pools 85 -> 42 and upsamples 42 -> 85) reads src = floor(dst * n / (2n+1))
// = (dst - 1) >> 1 for dst >= 1 and 0 for dst = 0: the exact 2x upsampling SHIFTED BY ONE, with low[0] once more in front.  For every
// output voxel d >= 2 along such an axis the three taps of the convolution see exactly the shifted 2x tensor, so
//     out[d] = (exact-2x sub-pixel convolution)[d - 1],
// zero padding at the far end included; only the slab d < 2 needs the general kernel (u3d_conv3d_box).  win = {Do, Ho, Wo, oz, oy, ox}:
// dims of `out` and the shift (1 on an n -> 2n + 1 axis, 0 on an exact one).  Voxels d = o (u = 0) are written too and are WRONG on
// shifted axes (they miss the extra copy of low[0]): the caller overwrites the slab afterwards.  No split-K (workspace unused).
extern "C" int u3d_subpixel_conv_fwd_win(int device, u3d_stream_t stream, const float* low, const float* affine,
                                         long long affine_sample_stride, const float* packed, float* out, int N, int D1, int H1,
                                         int W1, int C1, int Cout, const int* win) {
    U3D_REQUIRE(win != nullptr, "u3d_subpixel_conv_fwd_win: win is NULL");
    return subpixel_conv_fwd_impl(device, stream, low, affine, affine_sample_stride, packed, out, N, D1, H1, W1, C1, Cout, nullptr, 0,
                                  win);
}

static int subpixel_conv_fwd_impl(int device, u3d_stream_t stream, const float* low, const float* affine,
                                  long long affine_sample_stride, const float* packed, float* out, int N, int D1, int H1, int W1,
                                  int C1, int Cout, float* workspace, long long workspace_floats, const int* win) {
    U3D_ENTER(device);
    U3D_REQUIRE(low && packed && out && N > 0 && D1 > 0 && H1 > 0 && W1 > 0 && C1 > 0 && Cout > 0,
                "u3d_subpixel_conv_fwd: bad argument");
    U3D_REQUIRE(C1 % 4 == 0 && Cout % 4 == 0, "u3d_subpixel_conv_fwd: C1 and Cout must be multiples of 4 (got %d,%d)", C1,
                Cout);
    U3D_REQUIRE((((uintptr_t)low | (uintptr_t)packed | (uintptr_t)out | (uintptr_t)affine) & 15) == 0 &&
                    (affine == nullptr || affine_sample_stride % 4 == 0),
                "u3d_subpixel_conv_fwd: pointers must be 16-byte aligned");
    U3D_REQUIRE((long long)N * D1 * H1 * W1 * 8 < (1ll << 31), "u3d_subpixel_conv_fwd: volume too large");
    SubpixParams p;
    p.low = low, p.affine = affine, p.aff_nstride = affine_sample_stride, p.wp = packed, p.out = out;
    p.N = N, p.D1 = D1, p.H1 = H1, p.W1 = W1, p.C1 = C1, p.Cout = Cout;
    p.Do = 2 * D1, p.Ho = 2 * H1, p.Wo = 2 * W1;
    p.oz = p.oy = p.ox = 0;
    if (win) {
        p.Do = win[0], p.Ho = win[1], p.Wo = win[2], p.oz = win[3], p.oy = win[4], p.ox = win[5];
        U3D_REQUIRE(p.oz >= 0 && p.oy >= 0 && p.ox >= 0 && 2 * D1 + p.oz <= p.Do && 2 * H1 + p.oy <= p.Ho && 2 * W1 + p.ox <= p.Wo,
                    "u3d_subpixel_conv_fwd_win: the shifted 2x grid does not fit the output");
        U3D_REQUIRE((long long)N * p.Do * p.Ho * p.Wo < (1ll << 31), "u3d_subpixel_conv_fwd_win: volume too large");
    }
    p.tz = sp_cdiv(D1, sp::TZ), p.ty = sp_cdiv(H1, sp::TY), p.tx = sp_cdiv(W1, sp::TX);
    p.nchunks = sp_cdiv(C1, 16), p.ncb = sp_cdiv(Cout, 32);
    long long nblk = (long long)N * p.tz * p.ty * p.tx * p.ncb;
    U3D_REQUIRE(nblk < (1ll << 31), "u3d_subpixel_conv_fwd: grid too large");
    const long long out_elems = (long long)N * D1 * H1 * W1 * 8 * Cout;
    subpixel_fwd_split(nblk, p.nchunks, &p.ksplit, &p.cps);
    if (win) p.ksplit = 1;  // (the split partials are laid out like an exact-2x output)
    if (p.ksplit > 1 && workspace && ((uintptr_t)workspace & 15) == 0 && (long long)p.ksplit * out_elems <= workspace_floats) {
        p.out = workspace, p.part_stride = out_elems;
        nblk *= p.ksplit;
    } else {
        p.ksplit = 1, p.cps = p.nchunks, p.part_stride = 0;
    }
    hipLaunchKernelGGL(subpixel_fwd_kernel<sp::Nearest2x>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, p);
    U3D_LAUNCH_CHECK();
    if (p.ksplit > 1) {
        long long blocks = (out_elems / 4 + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, workspace, out_elems,
                           p.ksplit, out, out_elems / 4);
        U3D_LAUNCH_CHECK();
    }
    return 0;
}

// ---- ConvTranspose3d(k=3, stride=2, padding=1, bias=False) forward on the same kernel (scheme Deconv3s2): 8 parity classes
// with 1/2/4/8 single taps each — exactly 27*Cin*Cout multiply-adds per input voxel, no zero-stuffed work.
__global__ void pack_deconv_subpixel_kernel(const float* __restrict__ w, float* __restrict__ out, int Cin, int Cout, int nchunks,
                                            int ncb, long long total) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x)
        out[idx] = sp::pack_elem_deconv(w, Cin, Cout, nchunks, ncb, idx);
}

extern "C" long long u3d_convtr3d_subpixel_packed_floats(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0) return 0;
    return sp::packed_floats_deconv(Cin, Cout);
}

extern "C" int u3d_pack_convtr3d_subpixel(int device, u3d_stream_t stream, const float* w, int Cin, int Cout, float* packed) {
    U3D_ENTER(device);
    U3D_REQUIRE(w && packed && Cin > 0 && Cout > 0, "u3d_pack_convtr3d_subpixel: bad argument");
    const long long total = sp::packed_floats_deconv(Cin, Cout);
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_deconv_subpixel_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, packed, Cin,
                       Cout, sp_cdiv(Cin, 16), sp_cdiv(Cout, 32), total);
    U3D_LAUNCH_CHECK();
    return 0;
}

extern "C" int u3d_convtr3d_fwd_subpixel(int device, u3d_stream_t stream, const float* x, const float* packed, float* t, int N,
                                         int D1, int H1, int W1, int Cin, int Cout) {
    U3D_ENTER(device);
    U3D_REQUIRE(x && packed && t && N > 0 && D1 > 0 && H1 > 0 && W1 > 0 && Cin > 0 && Cout > 0,
                "u3d_convtr3d_fwd_subpixel: bad argument");
    U3D_REQUIRE(Cin % 4 == 0 && Cout % 4 == 0, "u3d_convtr3d_fwd_subpixel: Cin and Cout must be multiples of 4 (got %d,%d)",
                Cin, Cout);
    U3D_REQUIRE((((uintptr_t)x | (uintptr_t)packed | (uintptr_t)t) & 15) == 0,
                "u3d_convtr3d_fwd_subpixel: pointers must be 16-byte aligned");
    U3D_REQUIRE((long long)N * D1 * H1 * W1 * 8 < (1ll << 31), "u3d_convtr3d_fwd_subpixel: volume too large");
    SubpixParams p;
    p.low = x, p.affine = nullptr, p.aff_nstride = 0, p.wp = packed, p.out = t;
    p.N = N, p.D1 = D1, p.H1 = H1, p.W1 = W1, p.C1 = Cin, p.Cout = Cout;
    p.Do = 2 * D1 - 1, p.Ho = 2 * H1 - 1, p.Wo = 2 * W1 - 1;
    p.oz = p.oy = p.ox = 0;
    p.tz = sp_cdiv(D1, sp::TZ), p.ty = sp_cdiv(H1, sp::TY), p.tx = sp_cdiv(W1, sp::TX);
    p.nchunks = sp_cdiv(Cin, 16), p.ncb = sp_cdiv(Cout, 32);
    p.ksplit = 1, p.cps = p.nchunks, p.part_stride = 0;
    const long long nblk = (long long)N * p.tz * p.ty * p.tx * p.ncb;
    U3D_REQUIRE(nblk < (1ll << 31), "u3d_convtr3d_fwd_subpixel: grid too large");
    hipLaunchKernelGGL(subpixel_fwd_kernel<sp::Deconv3s2>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, p);
    U3D_LAUNCH_CHECK();
    return 0;
}

__global__ void pack_subpixel_dgrad_kernel(const float* __restrict__ w, float* __restrict__ out, int K, int cstride, int C1,
                                           int nchunks, int ntot, long long total) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x)
        out[idx] = spd::pack_elem(w, K, cstride, C1, nchunks, ntot, idx);
}

extern "C" long long u3d_subpixel_dgrad_packed_floats(int Cout, int C1) {
    if (C1 <= 0 || Cout <= 0) return 0;
    return spd::packed_floats(Cout, C1);
}

extern "C" int u3d_pack_subpixel_dgrad_weights(int device, u3d_stream_t stream, const float* w, int Cout, int Cin_total,
                                               int c_off, int C1, float* packed) {
    U3D_ENTER(device);
    U3D_REQUIRE(w && packed && Cout > 0 && C1 > 0 && c_off >= 0 && c_off + C1 <= Cin_total,
                "u3d_pack_subpixel_dgrad_weights: bad argument");
    const long long total = spd::packed_floats(Cout, C1);
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_subpixel_dgrad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       w + (size_t)c_off * 27, packed, Cout, Cin_total, C1, sp_cdiv(Cout, 16), sp_cdiv(C1, 32), total);
    U3D_LAUNCH_CHECK();
    return 0;
}

static int subpixel_conv_dgrad_impl(int device, u3d_stream_t stream, const float* dz, const float* packed, const float* x_low,
                                    float* dlow, double* gstats, int N, int D1, int H1, int W1, int C1, int Cout, const int* win,
                                    int greps = 1);

extern "C" int u3d_subpixel_conv_dgrad(int device, u3d_stream_t stream, const float* dz, const float* packed,
                                       const float* x_low, float* dlow, double* gstats, int N, int D1, int H1, int W1, int C1,
                                       int Cout) {
    return subpixel_conv_dgrad_impl(device, stream, dz, packed, x_low, dlow, gstats, N, D1, H1, W1, C1, Cout, nullptr);
}

// ... with gstats as `reps` replica rows [reps][N][C1][2] (zeroed by the caller; u3d_conv3d_ex_reps): the last wave of resident blocks
// flushes onto the same 2 C1 doubles at the end of the launch
extern "C" int u3d_subpixel_conv_dgrad_reps(int device, u3d_stream_t stream, const float* dz, const float* packed, const float* x_low,
                                            float* dlow, double* gstats, int N, int D1, int H1, int W1, int C1, int Cout, int reps) {
    U3D_REQUIRE(reps >= 1 && reps <= 64, "u3d_subpixel_conv_dgrad_reps: reps must be 1 .. 64");
    return subpixel_conv_dgrad_impl(device, stream, dz, packed, x_low, dlow, gstats, N, D1, H1, W1, C1, Cout, nullptr, reps);
}

// ... reading dz through the window of u3d_subpixel_conv_fwd_win: win = {Dd, Hd, Wd, oz, oy, ox, lz, ly, lx} — dz is (N, Dd, Hd, Wd,
// Cout), voxel u of the 2x grid is dz[u + o], and voxels u < l do not count (l = 1 on an n -> 2n + 1 axis: output d = 1 belongs to the
// boundary slab, whose share of the gradient the box launches + u3d_nearest_childsum_add deliver).  dlow / gstats: the share of the
// outputs d >= 2.
extern "C" int u3d_subpixel_conv_dgrad_win(int device, u3d_stream_t stream, const float* dz, const float* packed, const float* x_low,
                                           float* dlow, double* gstats, int N, int D1, int H1, int W1, int C1, int Cout,
                                           const int* win) {
    U3D_REQUIRE(win != nullptr, "u3d_subpixel_conv_dgrad_win: win is NULL");
    return subpixel_conv_dgrad_impl(device, stream, dz, packed, x_low, dlow, gstats, N, D1, H1, W1, C1, Cout, win);
}

static int subpixel_conv_dgrad_impl(int device, u3d_stream_t stream, const float* dz, const float* packed, const float* x_low,
                                    float* dlow, double* gstats, int N, int D1, int H1, int W1, int C1, int Cout, const int* win,
                                    int greps) {
    U3D_ENTER(device);
    U3D_REQUIRE(dz && packed && dlow && N > 0 && D1 > 0 && H1 > 0 && W1 > 0 && C1 > 0 && Cout > 0,
                "u3d_subpixel_conv_dgrad: bad argument");
    U3D_REQUIRE((gstats == nullptr) || x_low != nullptr, "u3d_subpixel_conv_dgrad: gstats needs x_low");
    U3D_REQUIRE(C1 % 4 == 0 && Cout % 4 == 0, "u3d_subpixel_conv_dgrad: C1 and Cout must be multiples of 4 (got %d,%d)", C1,
                Cout);
    U3D_REQUIRE((((uintptr_t)dz | (uintptr_t)packed | (uintptr_t)dlow | (uintptr_t)x_low) & 15) == 0,
                "u3d_subpixel_conv_dgrad: pointers must be 16-byte aligned");
    U3D_REQUIRE((long long)N * D1 * H1 * W1 * 8 < (1ll << 31), "u3d_subpixel_conv_dgrad: volume too large");
    SubpixDgradParams p;
    p.dz = dz, p.wp = packed, p.xlow = x_low, p.out = dlow, p.gstats = gstats;
    p.greps = greps;
    p.N = N, p.D1 = D1, p.H1 = H1, p.W1 = W1, p.C1 = C1, p.K = Cout;
    p.Dd = 2 * D1, p.Hd = 2 * H1, p.Wd = 2 * W1, p.oz = p.oy = p.ox = 0, p.lz = p.ly = p.lx = 0;
    if (win) {
        p.Dd = win[0], p.Hd = win[1], p.Wd = win[2], p.oz = win[3], p.oy = win[4], p.ox = win[5], p.lz = win[6], p.ly = win[7], p.lx = win[8];
        U3D_REQUIRE(p.oz >= 0 && p.oy >= 0 && p.ox >= 0 && 2 * D1 + p.oz <= p.Dd && 2 * H1 + p.oy <= p.Hd && 2 * W1 + p.ox <= p.Wd &&
                        p.lz >= 0 && p.ly >= 0 && p.lx >= 0 && (long long)N * p.Dd * p.Hd * p.Wd < (1ll << 31),
                    "u3d_subpixel_conv_dgrad_win: bad window");
    }
    p.tz = sp_cdiv(D1, spd::TZ), p.ty = sp_cdiv(H1, spd::TY), p.tx = sp_cdiv(W1, spd::TX);
    p.nchunks = sp_cdiv(Cout, 16), p.ntot = sp_cdiv(C1, 32), p.ncb = sp_cdiv(p.ntot, 2);
    const long long nblk = (long long)N * p.tz * p.ty * p.tx * p.ncb;
    U3D_REQUIRE(nblk < (1ll << 31), "u3d_subpixel_conv_dgrad: grid too large");
    const size_t shmem = (spd::REGION_FLOATS + 4 + 4 * 32 * 2 + 16) * sizeof(float);
    static bool attr_done[64] = {false};
    if (device < 0 || device >= 64 || !attr_done[device]) {
        U3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(subpixel_dgrad_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        if (device >= 0 && device < 64) attr_done[device] = true;
    }
    hipLaunchKernelGGL(subpixel_dgrad_kernel, dim3((unsigned)nblk), dim3(256), shmem, (hipStream_t)stream, p);
    U3D_LAUNCH_CHECK();
    return 0;
}

static void subpixel_wgrad_plan(int N, int D1, int H1, int W1, int C1, int K, SubpixWgradParams& p) {
    p.nchunks = sp_cdiv(C1, 32), p.nkb = sp_cdiv(K, 32);
    p.tz = sp_cdiv(D1, spw::TZ), p.ty = sp_cdiv(H1, spw::TY), p.tx = sp_cdiv(W1, spw::TX);
    p.ntiles = N * p.tz * p.ty * p.tx;
    // one 8-wave block per CU: the split count whose grid fills whole rounds of 256 CUs best (as u3d_conv3d_wgrad)
    const int pairs = p.nchunks * p.nkb;
    long long best = -1;
    int best_S = 1;
    for (int rounds = 1; rounds <= 8; ++rounds) {
        int S = (rounds * 256) / pairs;
        if (S < 1) S = 1;
        if (S > p.ntiles) S = p.ntiles;
        const int tps = sp_cdiv(p.ntiles, S);
        S = sp_cdiv(p.ntiles, tps);
        const long long cost = (long long)sp_cdiv(S * pairs, 256) * (tps + 2);
        if (best < 0 || cost < best) best = cost, best_S = S;
    }
    p.tps = sp_cdiv(p.ntiles, best_S);
    p.S = sp_cdiv(p.ntiles, p.tps);
}

extern "C" long long u3d_subpixel_wgrad_workspace_floats(int N, int D1, int H1, int W1, int C1, int Cout) {
    if (N <= 0 || D1 <= 0 || H1 <= 0 || W1 <= 0 || C1 <= 0 || Cout <= 0) return 0;
    SubpixWgradParams p;
    subpixel_wgrad_plan(N, D1, H1, W1, C1, Cout, p);
    return (long long)p.S * p.nchunks * p.nkb * 64 * 1024;
}

static int subpixel_conv_wgrad_impl(int device, u3d_stream_t stream, const float* low, const float* affine, long long affine_sample_stride,
                                    const float* dz, float* dw, int dw_cin_stride, int N, int D1, int H1, int W1, int C1, int Cout,
                                    float* workspace, long long workspace_floats, const int* win);

extern "C" int u3d_subpixel_conv_wgrad(int device, u3d_stream_t stream, const float* low, const float* affine,
                                       long long affine_sample_stride, const float* dz, float* dw, int dw_cin_stride, int N,
                                       int D1, int H1, int W1, int C1, int Cout, float* workspace,
                                       long long workspace_floats) {
    return subpixel_conv_wgrad_impl(device, stream, low, affine, affine_sample_stride, dz, dw, dw_cin_stride, N, D1, H1, W1, C1, Cout,
                                    workspace, workspace_floats, nullptr);
}

// ... with dz read through the window of u3d_subpixel_conv_dgrad_win (same 9 integers): the weight gradient of the outputs d >= 2 of a
// level that upsamples n -> 2n + 1; the boundary slab's share comes from u3d_conv3d_wgrad_box.
extern "C" int u3d_subpixel_conv_wgrad_win(int device, u3d_stream_t stream, const float* low, const float* affine,
                                           long long affine_sample_stride, const float* dz, float* dw, int dw_cin_stride, int N, int D1,
                                           int H1, int W1, int C1, int Cout, float* workspace, long long workspace_floats,
                                           const int* win) {
    U3D_REQUIRE(win != nullptr, "u3d_subpixel_conv_wgrad_win: win is NULL");
    return subpixel_conv_wgrad_impl(device, stream, low, affine, affine_sample_stride, dz, dw, dw_cin_stride, N, D1, H1, W1, C1, Cout,
                                    workspace, workspace_floats, win);
}

static int subpixel_conv_wgrad_impl(int device, u3d_stream_t stream, const float* low, const float* affine, long long affine_sample_stride,
                                    const float* dz, float* dw, int dw_cin_stride, int N, int D1, int H1, int W1, int C1, int Cout,
                                    float* workspace, long long workspace_floats, const int* win) {
    U3D_ENTER(device);
    U3D_REQUIRE(low && dz && dw && workspace && N > 0 && D1 > 0 && H1 > 0 && W1 > 0 && C1 > 0 && Cout > 0 &&
                    dw_cin_stride >= C1,
                "u3d_subpixel_conv_wgrad: bad argument");
    U3D_REQUIRE(C1 % 4 == 0, "u3d_subpixel_conv_wgrad: C1 must be a multiple of 4 (got %d)", C1);
    U3D_REQUIRE((((uintptr_t)low | (uintptr_t)affine) & 15) == 0 && (affine == nullptr || affine_sample_stride % 4 == 0),
                "u3d_subpixel_conv_wgrad: low / affine must be 16-byte aligned");
    U3D_REQUIRE((long long)N * D1 * H1 * W1 * 8 < (1ll << 31), "u3d_subpixel_conv_wgrad: volume too large");
    SubpixWgradParams p;
    subpixel_wgrad_plan(N, D1, H1, W1, C1, Cout, p);
    const long long need = (long long)p.S * p.nchunks * p.nkb * 64 * 1024;
    if (workspace_floats < need)
        return u3d_set_err(U3D_EWORKSPACE, "u3d_subpixel_conv_wgrad: workspace %lld < %lld floats", workspace_floats, need);
    p.low = low, p.affine = affine, p.aff_nstride = affine_sample_stride, p.dz = dz, p.partial = workspace;
    p.N = N, p.D1 = D1, p.H1 = H1, p.W1 = W1, p.C1 = C1, p.K = Cout;
    p.Dd = 2 * D1, p.Hd = 2 * H1, p.Wd = 2 * W1, p.oz = p.oy = p.ox = 0, p.lz = p.ly = p.lx = 0;
    if (win) {
        p.Dd = win[0], p.Hd = win[1], p.Wd = win[2], p.oz = win[3], p.oy = win[4], p.ox = win[5], p.lz = win[6], p.ly = win[7], p.lx = win[8];
        U3D_REQUIRE(p.oz >= 0 && p.oy >= 0 && p.ox >= 0 && 2 * D1 + p.oz <= p.Dd && 2 * H1 + p.oy <= p.Hd && 2 * W1 + p.ox <= p.Wd &&
                        p.lz >= 0 && p.ly >= 0 && p.lx >= 0 && (long long)N * p.Dd * p.Hd * p.Wd < (1ll << 31),
                    "u3d_subpixel_conv_wgrad_win: bad window");
    }
    // every tile inside the volume, whole 32-channel dz blocks: constant-offset B loads (key 15 = 1: the general kernel, for A/B)
    const bool full = !win && D1 % spw::TZ == 0 && H1 % spw::TY == 0 && W1 % spw::TX == 0 && Cout % 32 == 0 && g_u3d_tune[15] != 1 &&
                      (long long)N * 8 * D1 * H1 * W1 * Cout * 4 < (1ll << 31);
    if (full)
        hipLaunchKernelGGL(subpixel_wgrad_kernel<true>, dim3((unsigned)(p.S * p.nchunks * p.nkb)), dim3(spw::NTHR), 0,
                           (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(subpixel_wgrad_kernel<false>, dim3((unsigned)(p.S * p.nchunks * p.nkb)), dim3(spw::NTHR), 0,
                           (hipStream_t)stream, p);
    U3D_LAUNCH_CHECK();
    // (the read-once kernel has C1 * nkb blocks: with 64 of them — the 64 -> 32-channel top level — it is latency-bound at 75 us against
    // 28 us of the one-thread-per-output kernel; from 256 blocks on it wins, 32 -> 23 and 27 -> 17 us on the two levels below)
    if (g_u3d_tune[22] == 1 || (g_u3d_tune[22] != 2 && C1 * p.nkb < 256)) {  // key 22 = 1 / 2: always the round-2 / the read-once reduction (A/B)
        const long long total = (long long)C1 * 27 * Cout;
        hipLaunchKernelGGL(subpixel_wgrad_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, (hipStream_t)stream,
                           workspace, dw, p.S, p.nchunks, p.nkb, C1, Cout, dw_cin_stride);
    } else {
        int G = 16;  // split groups per block: every group at least one split, at most 512 threads
        while (G > 1 && G > p.S) G >>= 1;
        hipLaunchKernelGGL(subpixel_wgrad_reduce_once_kernel, dim3((unsigned)(C1 * p.nkb)), dim3(32 * G), (size_t)G * 27 * 32 * sizeof(float),
                           (hipStream_t)stream, workspace, dw, p.S, p.nchunks, p.nkb, C1, Cout, dw_cin_stride);
    }
    U3D_LAUNCH_CHECK();
    return 0;
}
