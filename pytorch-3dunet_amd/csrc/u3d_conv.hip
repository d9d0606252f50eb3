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
#include <type_traits>
#include <utility>

#include "u3d_common.h"
#include "u3d_gn.h"
#include "u3d_subpix.h"

// run-time tuning knobs (u3d_set_tuning), for A/B measurements only — results never change:
//   [0] forced N-tiles per block of u3d_conv3d (1,2,3; 0 = automatic)   [1] wgrad split override (0 = automatic)
//   [2] ablation mask of the instrumented conv twin (timing experiments, wrong results)
//   [3] 1 = never use the persistent fast variant of u3d_conv3d (A/B against the generic kernel)
//   [4] <= 16 output channels: 0 = 16-column variant (v_mfma_f32_16x16x4_f32), 2 = paired-y variant (round 1), 1 = padded 32-column kernel
//   [5] start-phase stagger of the persistent kernel in units of 1024 cycles (0 = off)
//   [6] 1 = one persistent block per CU instead of two (occupancy experiment)
//   [7] bf16-storage weight gradient: 1 = the round-3 kernel, 16 / 8 = force the 16- / 8-wide tile of conv3d_wgrad_b16v2_kernel (A/B);
//       2 = u3d_conv3d_ex never splits the channel reduction (A/B of the split-K path)
//   [8] bf16 weight gradient: target number of blocks (0 = default)   [9] 1 = bf16 weight gradient without the XCD-aware block order
//   [10] 1 = bf16-storage convolutions on 4-plane tiles only (no 8-plane tiles); 2 = transposed-convolution 2x2x2 kernels without
//        skipping their structurally zero weight blocks   [11] bf16 kernels: 1 = x-y-z raster tile order; 2 / 3 = the tile index never / always runs
//        fastest over the block ids (default: on small volumes with wide layers)
//   [12] block slots (of 2 per CU) that the persistent convolution grids leave FREE for kernels of other streams (RCCL's gradient
//        all-reduce: parallel.cu_budget).  A CU-MASKED compute queue was measured instead and rejected: the same kernels run 40-75 %
//        slower on a queue masked to 248 of 256 CUs (profiles/r04_cu_mask_*.txt)
//   [13] 1 = first-layer forward on the direct kernel instead of the matrix-pipe one (csrc/u3d_smallc.hip, A/B)
//   [14] 1 = persistent convolution kernel walks its tiles x-fastest (rounds 1-3) instead of z-fastest
//   [15] 1 = sub-pixel weight gradient with per-element coordinate arithmetic for its B loads (A/B of the constant-offset path)
//   [16] bf16-storage 3x3x3 convolution, flat 5 x 10 x 10 tile of the small wide levels: 1 = never, >= 2 = force that split count
//   [17] 1 = bf16-storage weight gradient with ONE split still goes through the workspace + reduction kernel (A/B of the direct dw write)
//   [18] 1 = the one-channel input statistics run on the general u3d_chan_stats kernel — the round-5 form (A/B of the 16-byte kernel,
//        csrc/u3d_ops.hip).  (A max-pool with the statistics fused into its pass was built and measured in round 6: 77 / 43 / 12 us per level
//        against 52 + 23 / 14 + 11 / 5 + 8 for the massively parallel pool + a statistics pass — the pool alone moves 6 TB/s; not kept.)
//   [21] 1 = head forward of the 32 / 64-channel -> 1 / 2-output heads on the vectorised kernel instead of the LDS-rows kernel (A/B)
//   [22] sub-pixel weight gradient: 1 = always reduced by the one-thread-per-output kernel of round 2, 2 = always by the read-once kernel
//        (default: read-once from 256 blocks on, csrc/u3d_subpix.hip)
//   [23] 1 = the weight-gradient reduction always with four split groups per output (A/B of the one-thread-per-output form for <= 16 splits)
//   [19] / [20] total block count of the first-layer forward / backward kernels (csrc/u3d_smallc.hip; 0 = default 512 / 1024); [20] = -1: the
//        backward kernel's first form (4-byte operand loads from global memory) at its default block count (A/B)
int g_u3d_tune[24] = {0};

namespace cv {
constexpr int TZ = 4, TY = 8, TX = 8;
constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2;
constexpr int CC = 16;             // input channels per chunk
constexpr int CS = 16;             // voxel stride (floats)
constexpr int RS = HX * CS + 4;    // row stride 164 (bank-conflict-free, see header)
constexpr int PS = HY * RS;        // plane stride 1640
constexpr int TILE_FLOATS = HZ * PS + 4;          // 9840 floats + one dummy float4 slot = 39376 B per staging buffer
constexpr int CNT_OFF = 2 * TILE_FLOATS;          // 16 ints: [0,1] full[buf], [2,3] freed[buf], [4] statistics arrivals
constexpr int RED_OFF = CNT_OFF + 16;             // [4 waves][NT <= 3][32][2] partial statistics
constexpr int LDS_FLOATS = RED_OFF + 4 * 3 * 32 * 2;  // 20472 floats = 81888 B -> two blocks per CU
constexpr int NITEMS = HZ * HY * HX * (CC / 4);   // 2400 float4 items per chunk
constexpr int NIT = (NITEMS + 255) / 256;         // 10
constexpr int NSTEP = 27 * (CC / 8);              // 54 k-steps of 8 channels per chunk
constexpr int NSTEP_PAIRY = 36 * (CC / 8);        // 72 k-steps: 3 x 4 x 3 tap window of the paired-y variant
constexpr int NSTEP_N16 = 27;                     // 16-column variant: one k-step per tap over the chunk's 16 channels
constexpr int ST0 = 30;                           // k-step of the first halo store into the other buffer
constexpr int ISTORE = 5;                         // persistent variant: tap row (of 9) whose k-steps store the halo
constexpr int PACK_PAD = 5;                       // zero k-steps appended to the packed weight image (B prefetch overrun)
static_assert(2 * (NIT - 1) < ST0 - 8 && ST0 + NIT <= NSTEP, "prefetch schedule must fit the k-loop");
}  // namespace cv

struct ConvParams {
    u3d_src_t src;
    u3d_src_t gx;
    const float* wp;
    float* out;
    double* out_stats;
    double* gstats;
    const float* res;  // optional residual (N,D,H,W,Cout) added before the ReLU (ResNetBlock, buildingblocks.py:285)
    int N, D, H, W, Cout;
    int nchunks, ncb, ntot;
    int tz, ty, tx;
    int relu, vec, has_gx, ovec;
    int total;   // REG kernel: work items = tiles * ncb
    int gx_x2;   // REG kernel: gx's low-res half is an exact 2x upsampling (index = i >> 1, no table)
    int stagger; // REG kernel: start delay of the second half of the grid, in units of 1024 cycles
    int stat_reps;  // REG kernel: the statistics table has this many replica rows [reps][N][Cout][2]; block b adds to row b % reps (u3d_conv3d_ex_reps)
    int zfast;   // REG kernel: tiles are walked z-fastest (1) or x-fastest (0)
    int ksplit, cps;        // generic kernel, split-K: the chunk range is cut into ksplit runs of cps chunks, one per block,
    long long part_stride;  // each run writing its partial sums to out + run * part_stride (summed by splitk_reduce_kernel)
    long long* dbg;  // optional per-wave timeline records (u3d_set_profile_buffer), 24 int64 per wave
    // generic kernel only (round 5, u3d_conv3d_box): OUTPUT BOX [o*0, o*1) — tiles are enumerated from its origin and only voxels inside
    // are stored — and an INPUT SLAB MASK (mz, my, mx; all 0 = none): a source voxel counts only if mz && z < mz || my && y < my ||
    // mx && x < mx, everything else reads as zero.  Defaults: the whole volume, no mask.
    int oz0, oy0, ox0, oz1, oy1, ox1;
    int mz, my, mx;
};

// timeline record of one wave (DBG kernels only), 24 int64: [0] block, [1] HW_ID, [2] XCC_ID, [3] t_entry,
// [5] t_epilogue_start, [6] t_exit, [7] nchunks, [8+2c] / [9+2c] start / end of the k-loop of chunk c < 8
#define U3D_DBG_STAMP(slot)                                                   \
    do {                                                                      \
        if constexpr (DBG) {                                                  \
            if (l == 0) dbgw[slot] = (long long)__builtin_readcyclecounter(); \
        }                                                                     \
    } while (0)

// ---- intra-block flags in LDS (no rendezvous): the LDS unit executes a wave's DS instructions in order, so a
// counter increment issued after a wave's tile stores is observed only after those stores; the compiler is held to
// program order by the asm memory clobbers.  No fences: a workgroup-scope fence would drain vmcnt, i.e. stall on
// the in-flight weight / halo prefetch loads.
__device__ __forceinline__ void u3d_flag_signal(int* c, int lane) {
    asm volatile("" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void u3d_flag_wait(int* c, int target) {
    while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(2);
    asm volatile("" ::: "memory");
}

// Asynchronous software pipeline.  A block = 4 waves on the 4 SIMDs of a CU sharing a 4x8x8 output tile; its 6x10x10
// halo tile of the current 16-channel chunk lives in one of TWO LDS buffers.  During the 54 k-steps of chunk c a wave
//   steps 0,2,..,18   issues its 10 halo loads of chunk c+1 into registers (branch-free, clamped addresses),
//   every step        streams the B fragments (packed weights, one contiguous image over (chunk, step)) two steps ahead
//                     and reads the A fragments of the next step from LDS,
//   steps 30..39      applies the GroupNorm affine and writes one halo item per step into the OTHER buffer,
//   step 39           signals full[other]; at the end of the k-loop it signals freed[current].
// Waves never meet at a barrier inside the chunk loop: a wave enters chunk c+1 as soon as all four have signalled
// full, and overwrites a buffer only after all four signalled freed.  The four waves of a block share their SIMDs
// with waves of another block in arbitrary phases and therefore progress at different speeds; with rendezvous
// barriers that skew cost ~10 k cycles per chunk (measured with the timeline twin), with flags it is absorbed as
// long as a wave is not more than ~half a chunk ahead of the slowest.
// ABL (instrumented twin only, u3d_set_tuning key 2): timing-only ablation mask — 1 no halo prefetch/restaging,
// 2 no B loads in the k-loop, 4 no A reads in the k-loop.  ABL != 0 produces wrong results by design.
template <int NT, bool VEC, bool DBG = false, int ABL = 0>
__global__ __launch_bounds__(256, 2) void conv3d_mfma_kernel(const ConvParams p) {
    using namespace cv;
    constexpr int RB = NT == 1 ? 6 : 3;  // B ring depth: fragments are fetched RB-1 k-steps ahead (54 % RB == 0)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int* cnt = reinterpret_cast<int*>(lds + CNT_OFF);
    const int t = threadIdx.x;
    const int l = t & 63, w = t >> 6, m = l & 31, h = l >> 5;
    // Non-MFMA phases (prologue, epilogue) run at priority 3: co-resident waves in their k-loop always have an MFMA
    // pending and, being older, would otherwise win the issue arbitration every cycle.
    __builtin_amdgcn_s_setprio(3);
    long long* dbgw = nullptr;
    if constexpr (DBG) {
        dbgw = p.dbg + ((size_t)blockIdx.x * 4 + w) * 24;
        if (l == 0) {
            dbgw[0] = blockIdx.x;
            dbgw[1] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID
            dbgw[2] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
            dbgw[7] = p.nchunks;
        }
    }
    U3D_DBG_STAMP(3);
    if (t < 16) cnt[t] = 0;
    __syncthreads();  // the only rendezvous of the kernel

    // ---- block -> (tile, cout block) with XCD-contiguous ordering
    int logical = u3d_xcd_remap(blockIdx.x, gridDim.x);
    int ch0 = 0, nch = p.nchunks;  // this block's chunk run (split-K: one of ksplit runs)
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
    const int z0 = p.oz0 + tzi * TZ, y0 = p.oy0 + tyi * TY, x0 = p.ox0 + txi * TX;
    const int D = p.D, H = p.H, W = p.W;
    const int Ctot = p.src.C0 + p.src.C1;
    const bool slab_mask = (p.mz | p.my | p.mx) != 0;

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
        bool ok = in && gz >= 0 && gz < D && gy >= 0 && gy < H && gxx >= 0 && gxx < W;
        if (slab_mask) ok = ok && ((p.mz && gz < p.mz) || (p.my && gy < p.my) || (p.mx && gxx < p.mx));
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

    // B stream: packed f32x4 index ((g*ntot + ntile)*64 + lane) for the global step g = chunk*54 + step; the image
    // carries PACK_PAD zero steps past the end, so the (RB-1)-steps-ahead fetch needs no clamp.  vmcnt retires in
    // order: a B fragment can only be consumed once every OLDER load has landed, including a halo prefetch load
    // (HBM latency), hence the depth: RB-1 steps x 512*NT MFMA cycles must cover an HBM miss.
    const int wstep = p.ntot * 64;
    const f32x4* wq = reinterpret_cast<const f32x4*>(p.wp) + (size_t)cb * NT * 64 + l + (size_t)ch0 * NSTEP * wstep;
    f32x4 bq[RB][NT];
#pragma unroll
    for (int k = 0; k < RB - 1; ++k)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bq[k][nt] = wq[(size_t)k * wstep + nt * 64];

    // per-chunk source selection of this thread's channel quad.  The GroupNorm affine rows are NOT loaded here: a load
    // at the head of a chunk is waited for at once (s_waitcnt vmcnt(0), a full memory round trip per chunk) — see the
    // persistent kernel; load_affine_rows is called a few k-steps before the first halo store instead.
    struct ChunkSrc {
        const float* base;
        int Cs;
        bool cok, from0;
        const float* ap;  // this thread's rows of the affine table (nullptr: identity)
        f32x4 lo, hi;     // raw rows (a0,b0,a1,b1), (a2,b2,a3,b3)
    };
    auto chunk_src = [&](int ch, bool live) {
        ChunkSrc c;
        const int cq = ch * CC + 4 * q;
        c.cok = live && cq < Ctot;
        c.from0 = cq < p.src.C0;
        c.base = !c.cok ? p.src.p0 : (c.from0 ? p.src.p0 + cq : p.src.p1 + (cq - p.src.C0));
        c.Cs = (c.from0 || !c.cok) ? p.src.C0 : p.src.C1;
        c.ap = p.src.affine ? p.src.affine + ((size_t)n * Ctot + (c.cok ? cq : 0)) * 2 : nullptr;
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
    // branch-free halo load: always a valid address (clamped), validity is applied by select at the LDS write
    auto halo_load = [&](const ChunkSrc& c, int it) {
        const bool ok = c.cok && gv0[it] >= 0;
        const int idx = ok ? (c.from0 ? gv0[it] : gv1[it]) : 0;
        return *reinterpret_cast<const f32x4*>(c.base + (size_t)idx * c.Cs);
    };
    auto halo_store = [&](float* buf, const ChunkSrc& c, int it, f32x4 raw) {
        const bool ok = c.cok && gv0[it] >= 0;
        f32x4 val = {fmaf(raw[0], c.lo[0], c.lo[1]), fmaf(raw[1], c.lo[2], c.lo[3]), fmaf(raw[2], c.hi[0], c.hi[1]),
                     fmaf(raw[3], c.hi[2], c.hi[3])};
#pragma unroll
        for (int e = 0; e < 4; ++e) val[e] = ok ? val[e] : 0.f;  // padding stays exactly 0
        *reinterpret_cast<f32x4*>(&buf[ldsoff[it]]) = val;
    };
    // scalar-load staging of one chunk (channel counts that are not multiples of 4)
    auto stage_scalar = [&](float* buf, int ch) {
        const int cq = ch * CC + 4 * q;
        f32x4 ga, gb;
        u3d_load_affine(p.src.affine, n, Ctot, cq, false, ga, gb);
#pragma unroll 1
        for (int it = 0; it < NIT; ++it) {
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (gv0[it] >= 0) val = u3d_load_quad(p.src, gv0[it], gv1[it], cq, false) * ga + gb;
            *reinterpret_cast<f32x4*>(&buf[ldsoff[it]]) = val;
        }
    };

    // ---- prologue: stage chunk 0 into buffer 0
    if constexpr (VEC) {
        ChunkSrc c0 = chunk_src(ch0, true);
        load_affine_rows(c0);
        f32x4 v[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) v[it] = halo_load(c0, it);
#pragma unroll
        for (int it = 0; it < NIT; ++it) halo_store(lds, c0, it, v[it]);
    } else {
        stage_scalar(lds, ch0);
    }
    u3d_flag_signal(&cnt[0], l);

    for (int ch = 0; ch < nch; ++ch) {  // ch counts within the block's run; the source chunk is ch0 + ch
        const bool has_next = ch + 1 < nch;
        const int b = ch & 1;
        const float* cur = lds + b * TILE_FLOATS;
        float* nxt = lds + (b ^ 1) * TILE_FLOATS;
        ChunkSrc cn;
        f32x4 v[NIT];
        if constexpr (VEC) cn = chunk_src(ch0 + ch + 1, has_next);
        u3d_flag_wait(&cnt[b], 4 * (ch / 2 + 1));  // all four waves have staged chunk ch
        __builtin_amdgcn_s_setprio(0);
        if (ch < 8) U3D_DBG_STAMP(8 + 2 * ch);

        // ---- 54 k-steps (27 taps x 2 channel-octets), 8*NT MFMAs each, in two halves so that the non-MFMA
        //      instructions in front of each half issue while the previous MFMA still occupies the pipe; the compiler
        //      inserts the matching counted vmcnt/lgkmcnt waits (the loop is fully unrolled).
        f32x4 aq[2][2];
        aq[0][0] = *reinterpret_cast<const f32x4*>(&cur[abase]);
        aq[0][1] = *reinterpret_cast<const f32x4*>(&cur[abase + 4 * RS]);
        const f32x4* wch = wq + (size_t)(ch * NSTEP + RB - 1) * wstep;
#pragma unroll
        for (int st = 0; st < NSTEP; ++st) {
            if constexpr (VEC && !(ABL & 1)) {
                if (st % 2 == 0 && st / 2 < NIT) v[st / 2] = halo_load(cn, st / 2);
                if (st == ST0 - 6) load_affine_rows(cn);  // 6 k-steps ahead of the first halo store
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if constexpr (ABL & 2) {
                    bq[(st + RB - 1) % RB][nt] = bq[st % RB][nt];
                    asm volatile("" : "+v"(bq[(st + RB - 1) % RB][nt]));
                } else {
                    bq[(st + RB - 1) % RB][nt] = wch[(size_t)st * wstep + nt * 64];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[st & 1][0][j], bq[st % RB][nt][j], acc[0][nt], 0, 0, 0);
                    acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[st & 1][1][j], bq[st % RB][nt][j], acc[1][nt], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if ((ABL & 4) && st + 1 < NSTEP) {
                aq[(st + 1) & 1][0] = aq[st & 1][0];
                aq[(st + 1) & 1][1] = aq[st & 1][1];
                asm volatile("" : "+v"(aq[(st + 1) & 1][0]), "+v"(aq[(st + 1) & 1][1]));
            } else if (st + 1 < NSTEP) {
                const int tap = (st + 1) >> 1, s1_ = (st + 1) & 1;
                const int aoff = (tap / 9) * PS + ((tap / 3) % 3) * RS + (tap % 3) * CS + 8 * s1_;
                aq[(st + 1) & 1][0] = *reinterpret_cast<const f32x4*>(&cur[abase + aoff]);
                aq[(st + 1) & 1][1] = *reinterpret_cast<const f32x4*>(&cur[abase + 4 * RS + aoff]);
            }
            if (st >= ST0 && st < ST0 + NIT && has_next && !(ABL & 1)) {
                // the other buffer is free once all four waves have finished the k-loop of chunk ch-1
                if (st == ST0) u3d_flag_wait(&cnt[2 + (b ^ 1)], 4 * ((ch + 1) / 2));
                if constexpr (VEC) {
                    halo_store(nxt, cn, st - ST0, v[st - ST0]);
                } else {
                    if (st == ST0) stage_scalar(nxt, ch0 + ch + 1);
                }
                if (st == ST0 + NIT - 1) u3d_flag_signal(&cnt[b ^ 1], l);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 2; j < 4; ++j) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[st & 1][0][j], bq[st % RB][nt][j], acc[0][nt], 0, 0, 0);
                    acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[st & 1][1][j], bq[st % RB][nt][j], acc[1][nt], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ch < 8) U3D_DBG_STAMP(9 + 2 * ch);
        if constexpr ((ABL & 1) != 0) {
            if (has_next) u3d_flag_signal(&cnt[b ^ 1], l);
        }
        u3d_flag_signal(&cnt[2 + b], l);  // this wave no longer reads buffer b
    }
    __builtin_amdgcn_s_setprio(3);

    // ---- epilogue.  C/D layout of 32x32 MFMA: col = lane&31 (cout), row = (r&3) + 8*(r>>2) + 4*(lane>>5);
    //      M-tile row -> (y = row>>3, x = row&7)  =>  reg r of lane (m,h): y = r>>2, x = (r&3) + 4h.
    U3D_DBG_STAMP(5);
    const int z = z0 + w;
    const bool want_stats = p.out_stats != nullptr;
    const bool want_g = p.gstats != nullptr;
    float* red = lds + RED_OFF;  // [4 waves][NT][32][2] partial statistics
    if (p.ovec) {
        // ---- wide epilogue.  In the MFMA C layout lane (m, h) holds channel m of 16 voxels; the 4 rows r&3 of a
        //      register quad are the voxels x = (r&3) + 4h of one y.  A 4x4 transpose inside every lane quad (two DPP
        //      butterfly stages, no LDS) leaves lane j of quad k with the 4 consecutive channels 4k..4k+3 of voxel
        //      x = j + 4h: 16-byte stores (8 voxels x 128 B = 1 KiB contiguous per instruction when Cout == 32) and
        //      16-byte loads of x for the GroupNorm-backward sums instead of 4-byte ones.
        const int cq = (l >> 2) & 7, vl = (l & 3) + 4 * h;  // after the transpose: channel quad, x within the row
        const bool odd = (l & 1) != 0, hi = (l & 2) != 0;
        auto xlane = [](float v, auto ctrl) {
            return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), decltype(ctrl)::value, 0xF, 0xF, true));
        };
        using X1 = std::integral_constant<int, 0xB1>;  // quad_perm [1,0,3,2]: value of lane ^ 1
        using X2 = std::integral_constant<int, 0x4E>;  // quad_perm [2,3,0,1]: value of lane ^ 2
        f32x4 q1[NT], q2[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            q1[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            q2[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 tq[8];  // row st = 4*mt + bi: the lane's channel quad at voxel (y0 + st, x0 + vl)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int bi = 0; bi < 4; ++bi) {
                    const float a0 = acc[mt][nt][4 * bi + 0], a1 = acc[mt][nt][4 * bi + 1];
                    const float a2 = acc[mt][nt][4 * bi + 2], a3 = acc[mt][nt][4 * bi + 3];
                    const float t0 = xlane(a1, X1{}), t1 = xlane(a0, X1{}), t2 = xlane(a3, X1{}), t3 = xlane(a2, X1{});
                    const float c0 = odd ? t0 : a0, c1 = odd ? a1 : t1, c2 = odd ? t2 : a2, c3 = odd ? a3 : t3;
                    const float u0 = xlane(c2, X2{}), u2 = xlane(c0, X2{}), u1 = xlane(c3, X2{}), u3 = xlane(c1, X2{});
                    tq[4 * mt + bi] = f32x4{hi ? u0 : c0, hi ? u1 : c1, hi ? c2 : u2, hi ? c3 : u3};
                }
            const int co = (cb * NT + nt) * 32 + 4 * cq;
            const bool cok = co < p.Cout;
            const int x = x0 + vl;
            const bool xfrom0 = co < p.gx.C0 || !cok;
            const float* xb = !cok ? p.gx.p0 : (xfrom0 ? p.gx.p0 + co : p.gx.p1 + (co - p.gx.C0));
            const int xcs = xfrom0 ? p.gx.C0 : p.gx.C1;
#pragma unroll
            for (int half = 0; half < 2; ++half) {  // two batches of 4 rows bound the live registers
                f32x4 xv[4];
                if (want_g) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        int v0, v1;  // clamped coordinates: always a valid address, masked below
                        u3d_vox_index(p.gx, n, min(z, D - 1), min(y0 + 4 * half + k, H - 1), min(x, W - 1), D, H, W, v0, v1);
                        xv[k] = *reinterpret_cast<const f32x4*>(xb + (size_t)(xfrom0 ? v0 : v1) * xcs);
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int st = 4 * half + k;
                    const int y = y0 + st;
                    const bool ok = cok && z < p.oz1 && y < p.oy1 && x < p.ox1;
                    f32x4 val = tq[st];
                    const size_t vidx = (size_t)((n * D + z) * H + y) * W + x;
                    if (p.res && ok) val += *reinterpret_cast<const f32x4*>(p.res + vidx * p.Cout + co);
                    if (p.relu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) val[e] = fmaxf(val[e], 0.f);
                    }
                    if (ok) *reinterpret_cast<f32x4*>(outp + vidx * p.Cout + co) = val;
#pragma unroll
                    for (int e = 0; e < 4; ++e) val[e] = ok ? val[e] : 0.f;
                    q1[nt] += val;
                    q2[nt] += want_g ? val * xv[k] : val * val;
                }
            }
        }
        if (want_stats || want_g) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float a = q1[nt][e], b = q2[nt][e];
#pragma unroll
                    for (int mask : {1, 2, 32}) {  // lanes of one channel quad: x = (l&3) + 4*(l>>5)
                        a += __shfl_xor(a, mask);
                        b += __shfl_xor(b, mask);
                    }
                    if ((l & 35) == 0) {
                        red[((w * NT + nt) * 32 + 4 * cq + e) * 2 + 0] = a;
                        red[((w * NT + nt) * 32 + 4 * cq + e) * 2 + 1] = b;
                    }
                }
            }
        }
    } else {
        float s1[NT], s2[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            s1[nt] = 0.f;
            s2[nt] = 0.f;
        }
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
                const bool vok = z < p.oz1 && y < p.oy1 && x < p.ox1;
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
                    if (p.res && ok) val += p.res[vidx * p.Cout + co];
                    if (p.relu) val = fmaxf(val, 0.f);
                    if (ok) outp[vidx * p.Cout + co] = val;
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
            // reduce over the two half-waves (same cout)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                s1[nt] += __shfl_xor(s1[nt], 32);
                s2[nt] += __shfl_xor(s2[nt], 32);
                if (h == 0) {
                    red[((w * NT + nt) * 32 + m) * 2 + 0] = s1[nt];
                    red[((w * NT + nt) * 32 + m) * 2 + 1] = s2[nt];
                }
            }
        }
    }
    if (want_stats || want_g) {
        // over the 4 waves through LDS: the LAST wave to arrive adds the four partials (fixed order) and issues one
        // f64 atomic per (n, channel) and block — nobody waits
        int arrived = 0;
        asm volatile("" ::: "memory");
        if (l == 0) arrived = __hip_atomic_fetch_add(&cnt[4], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        arrived = __builtin_amdgcn_readfirstlane(arrived);
        asm volatile("" ::: "memory");
        if (arrived == 3) {
            for (int k = l; k < NT * 32; k += 64) {
                const int nt = k >> 5, mm = k & 31;
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
    U3D_DBG_STAMP(6);
}

// ---- split-K: small volumes with many channels (the bottom of the U: 16 tiles x 4 channel blocks on 256 CUs) do not
// fill the chip with one block per (tile, channel block).  The generic kernel then runs ksplit blocks per item, each
// over a run of input-channel chunks, writing plain partial sums; this kernel adds the runs in a fixed order and applies
// the whole epilogue (residual, ReLU, GroupNorm statistics / GroupNorm-backward sums) — the result does not depend on
// which block finished first.
struct SplitKParams {
    const float* part;
    long long stride;
    float* out;
    const float* res;
    double* stats;
    u3d_src_t gx;
    int ksplit, relu, want_g, V, D, H, W, C;
};

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const SplitKParams p) {
    __shared__ double red[256 * 4 * 2];  // [C/4 quads <= 256][4][2]
    const int Q = p.C >> 2, per = blockDim.x / Q;
    const int t = threadIdx.x, cq = t % Q, vl = t / Q, n = blockIdx.y;
    for (int k = t; k < Q * 8; k += blockDim.x) red[k] = 0.0;
    __syncthreads();
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    const bool xfrom0 = 4 * cq < p.gx.C0;
    for (int v = blockIdx.x * per + vl; v < p.V; v += gridDim.x * per) {
        const size_t off = ((size_t)n * p.V + v) * p.C + 4 * cq;
        f32x4 a = *reinterpret_cast<const f32x4*>(p.part + off);
        for (int k = 1; k < p.ksplit; ++k) a += *reinterpret_cast<const f32x4*>(p.part + (size_t)k * p.stride + off);
        if (p.res) a += *reinterpret_cast<const f32x4*>(p.res + off);
        if (p.relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] = fmaxf(a[e], 0.f);
        }
        *reinterpret_cast<f32x4*>(p.out + off) = a;
        if (p.stats) {
            s1 += a;
            if (p.want_g) {
                const int x = v % p.W, y = (v / p.W) % p.H, z = v / (p.W * p.H);
                int v0, v1;
                u3d_vox_index(p.gx, n, z, y, x, p.D, p.H, p.W, v0, v1);
                const f32x4 xv = xfrom0 ? *reinterpret_cast<const f32x4*>(p.gx.p0 + (size_t)v0 * p.gx.C0 + 4 * cq)
                                        : *reinterpret_cast<const f32x4*>(p.gx.p1 + (size_t)v1 * p.gx.C1 + (4 * cq - p.gx.C0));
                s2 += a * xv;
            } else {
                s2 += a * a;
            }
        }
    }
    if (p.stats) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            __hip_atomic_fetch_add(&red[(cq * 4 + e) * 2], (double)s1[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&red[(cq * 4 + e) * 2 + 1], (double)s2[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __syncthreads();
        for (int k = t; k < p.C * 2; k += blockDim.x) u3d_atomic_add_f64(&p.stats[(size_t)n * p.C * 2 + k], red[k]);
    }
}

// compile-time row ranges of the persistent kernel's k-loop: f(integral_constant<int, B>), ..., f(integral_constant<int, E - 1>)
template <int B, int... I, class F>
__device__ __forceinline__ void u3d_for_rows_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, B + I>{}), ...);
}
template <int B, int E, class F>
__device__ __forceinline__ void u3d_for_rows(F&& f) {
    if constexpr (E > B) u3d_for_rows_impl<B>(std::make_integer_sequence<int, E - B>{}, static_cast<F&&>(f));
}

// =================================================================================================
// The fast variant of the same convolution: PERSISTENT blocks and a VALU diet.
//
// f32 MFMA executes on the SIMD's FMA lanes: tools/mfma_mix.hip measures ~5 cycles of lost MFMA time per VALU
// instruction that ANY co-resident wave issues (SALU, s_waitcnt and LDS reads are free).  With the timeline twin the
// model "64 cycles per MFMA + 5 per VALU instruction" reproduces the measured pipe utilisation of the generic kernel
// (87 % on the 6-chunk 96->32 layer, 80 % on the 2-chunk 32->32 layer, 69 % on the 1-chunk 16->32 layer): what is
// lost is the prologue (index arithmetic of 10 staging items, ~800 VALU), the epilogue and the per-item arithmetic
// of the staging, not memory latency.  This kernel therefore
//   * is launched as 2 blocks per CU that walk the work items slot, slot + grid, ...: (tile, 16-channel chunk) is one
//     flat sequence of chunks alternating between the two LDS buffers, the next tile's first chunk is staged under
//     the current tile's last k-loop and the prologue is paid once per block instead of once per tile;
//   * requires every tile to lie fully inside the volume (D % 4 == H % 8 == W % 8 == 0) and a plain or exact-2x
//     virtual source, so that the global index of a staging item is (uniform tile base) + (per-thread constant) and
//     its validity one bit of a mask assembled from six per-face item masks by uniform tests: ~7 VALU per item and
//     chunk instead of ~45 per item and tile, and no masking at all on the 2/3 of the tiles that touch no face;
//   * keeps the B stream on a scalar base pointer (no VALU in the 54 k-steps);
//   * needs no LDS in the epilogue (DPP transposition), accumulates the GroupNorm statistics per wave in LDS
//     (ds_add_f64) and flushes them with f64 atomics when the block's sample changes.
// Everything else (flags instead of barriers, halo prefetch schedule, fragment layouts) is the generic kernel's.
//
// PAIRY (Cout <= 16, NT = 1): half of a 32-column N-tile would be padding.  Columns 16-31 instead compute the SAME 16
// output channels for the voxel one row further in y: out[y+1] = sum_t x[y + (t_y+1)] w[t], i.e. the shared A row of
// voxel y serves both outputs when the tap window is 3 x 4 x 3 (36 taps, the weights of the second half shifted by one
// in y, zero where that leaves the 3^3 kernel).  A wave then needs only the even rows y = 0,2,4,6 of its z-plane (ONE
// M-tile): 72 k-steps x 4 MFMAs per chunk instead of 54 x 8 — 2/3 of the padded work.
//
// N16 (round 4; Cout <= 16, NT = 1, supersedes PAIRY): the same GEMM on v_mfma_f32_16x16x4_f32 — 16 output channels are exactly
// one MFMA tile, no padded or zero work at all (PAIRY still executes 4/3 of the algorithmic multiply-adds).  The WEIGHT fragment
// is the A operand (rows = output channels, one packed f32x4 per lane and TAP: channels 4*(lane>>4) .. +3 of column lane&15), the
// activation fragment the B operand (columns = 16 voxels: two y rows x 8 x; a wave's z-plane is 4 M-tiles), so a lane's four
// accumulator registers of an M-tile are FOUR CONSECUTIVE output channels of ONE voxel: 16-byte stores / side loads without the
// DPP transposition.  A k-step is one tap over the chunk's 16 channels: 4 A reads, one B fetch, 16 MFMAs of 32 cycles.
//
// RAG (round 5): volumes whose sizes are NOT multiples of the 4 x 8 x 8 tile (the reference's shipped patch is 80 x 170 x 170 and its
// pooled levels 85, 42, 21: resources/3DUnet_confocal_boundary/train_config.yml:94).  The last tile of an axis overhangs the volume:
// its far-face item mask covers every halo plane from the first one beyond the volume on (still six per-thread masks built once per
// block, nothing per tile), so the staged values there are the convolution's zero padding; the k-loop is unchanged (it computes
// values for the overhanging voxels that nobody reads) and only the epilogue differs — rows / planes beyond the volume are skipped
// by uniform branches, lanes beyond W by the store mask, and neither enters the GroupNorm sums.  A separate instantiation, so the
// aligned launches keep their exact code.
template <int NT, bool VIRT, bool DBG = false, bool PAIRY = false, bool AFF = true, bool N16 = false, bool RAG = false>
__global__ __launch_bounds__(256, 2) void conv3d_mfma_reg_kernel(const ConvParams p) {
    using namespace cv;
    static_assert(!RAG || (!PAIRY && !DBG), "ragged tiles: standard and 16-column variants only");
    static_assert(!PAIRY || NT == 1, "the paired-y variant has a single N-tile");
    static_assert(!N16 || (NT == 1 && !PAIRY), "the 16-column variant has a single N-tile");
    // B ring depth: fragments are fetched RB-1 k-steps ahead; slots are indexed by the k-step within the chunk (54 % RB == 0,
    // 72 % RB == 0).  NT = 1 has the registers for 9 (round 4: 6 -> 9, 8 steps = 4 k cycles of slack behind a halo load)
    constexpr int RB = NT == 1 ? 9 : 3;
    constexpr int MT = N16 ? 4 : (PAIRY ? 1 : 2);  // M-tiles per wave (32 voxels = 4 y-rows x 8 x; N16: 16 voxels = 2 y-rows x 8 x)
    constexpr int RY = PAIRY ? 4 : 3;    // taps along y
    constexpr int NROWS = 3 * RY;        // tap rows (z, y) of 3 x-taps
    constexpr int SPR = N16 ? 3 : 6;     // k-steps per tap row: 3 taps x 2 channel octets (N16: 3 taps over all 16 channels)
    constexpr int NSTEPL = NROWS * SPR;  // k-steps per chunk
    constexpr int DBG_TILE = 1;          // timeline twin: which tile of a block is stamped (1 = steady state, not the cold first)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int* cnt = reinterpret_cast<int*>(lds + CNT_OFF);
    // [2 sample parities][NT*32][2] block-level partial statistics, accumulated in f64 (ds_add_f64): the arrival order
    // of the waves then perturbs the sums below 1e-15 relative, i.e. results are run-to-run reproducible
    double* red = reinterpret_cast<double*>(lds + RED_OFF);
    static_assert((RED_OFF * 4) % 8 == 0 && 2 * 3 * 32 * 2 * 8 <= 4 * 3 * 32 * 2 * 4, "f64 statistics rows must fit");
    const int t = threadIdx.x;
    const int l = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), m = l & 31, h = l >> 5;
    __builtin_amdgcn_s_setprio(3);
    long long* dbgw = nullptr;
    if constexpr (DBG) {
        dbgw = p.dbg + ((size_t)blockIdx.x * 4 + w) * 24;
        if (l == 0) {
            dbgw[0] = blockIdx.x;
            dbgw[1] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID
            dbgw[2] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
            dbgw[7] = p.nchunks;
        }
    }
    U3D_DBG_STAMP(3);
    if (t < 16) cnt[t] = 0;
    for (int k = t; k < 2 * 3 * 32 * 2; k += 256) red[k] = 0.0;
    __syncthreads();  // the only rendezvous of the kernel

    const int D = p.D, H = p.H, W = p.W;
    const int Ctot = p.src.C0 + p.src.C1;
    const int G = gridDim.x;
    const int D1 = p.src.D1, H1 = p.src.H1, W1 = p.src.W1;

    // ---- work item -> (tile, cout block); slot s of the grid runs on XCD s % 8 (observed, speed only): within one
    //      round of G items every XCD gets a contiguous run of logical ids, so neighbouring tiles share an L2
    struct Item {
        int cb, n, z0, y0, x0;
        int base0, base1;  // voxel index of the halo origin (z0-1, y0-1, x0-1) in the full-res / low-res source
        unsigned inv;      // per-thread bit mask of staging items that are zero padding (or dead)
        bool border;       // uniform: the tile touches a face of the volume
    };
    const int q = t & 3, tv = t >> 2;
    int rel0[NIT], rel1[VIRT ? NIT : 1], loff[NIT];
    unsigned fz0 = 0, fz1 = 0, fy0 = 0, fy1 = 0, fx0 = 0, fx1 = 0, fdead = 0;
    // valid voxels of the LAST tile of each axis (RAG: 1 .. tile size; otherwise the tile size: far face = halo plane HZ-1 / HY-1 / HX-1)
    const int vz = RAG && D % TZ ? D % TZ : TZ, vy = RAG && H % TY ? H % TY : TY, vx = RAG && W % TX ? W % TX : TX;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int vox = tv + 64 * it;
        const bool in = vox < HZ * HY * HX;
        const int hz = vox / (HY * HX);
        const int rem = vox - hz * (HY * HX);
        const int hy = rem / HX;
        const int hx = rem - hy * HX;
        // dead tail items (the 10th round covers 2560 > 2400 items) read the halo origin and write the dummy slot
        rel0[it] = in ? (hz * H + hy) * W + hx : 0;
        // exact 2x, tile origins even: ((z0 - 1 + hz) >> 1) = z0/2 + ((hz - 1) >> 1), relative to the low-res origin
        if constexpr (VIRT) rel1[it] = in ? (((hz - 1) >> 1) * H1 + ((hy - 1) >> 1)) * W1 + ((hx - 1) >> 1) : 0;
        loff[it] = in ? hz * PS + hy * RS + hx * CS + 4 * q : HZ * PS;  // tail items -> dummy slot
        fz0 |= (in && hz == 0 ? 1u : 0u) << it;
        fz1 |= (in && hz > vz ? 1u : 0u) << it;
        fy0 |= (in && hy == 0 ? 1u : 0u) << it;
        fy1 |= (in && hy > vy ? 1u : 0u) << it;
        fx0 |= (in && hx == 0 ? 1u : 0u) << it;
        fx1 |= (in && hx > vx ? 1u : 0u) << it;
        fdead |= (in ? 0u : 1u) << it;
    }
    // Work items are walked with stride G.  Decoding an item index costs five integer divisions (VALU sequences — and a
    // VALU instruction of a wave whose SIMD partner is in its k-loop waits ~45 cycles for an MFMA boundary, measured with
    // the timeline twin: tile-switch code that takes 2.5 k cycles alone takes 10 k beside a busy partner).  So the index
    // is decoded ONCE (mixed-radix digits cb, x, y, z, n) and then advanced digit-wise by the digits of G with carries:
    // scalar adds / compares only.
    struct Digits {
        int cb, xi, yi, zi, n;
    };
    // Order of the tiles inside the flat item sequence.  At any time the 512 blocks work on 512 CONSECUTIVE items and every XCD on a
    // contiguous run of 64 of them, i.e. that run is what shares an L2.  x-fastest (rounds 1-3) makes it a 16 x 4 patch of (x, y) tiles
    // at ONE z: the z halos (6 planes fetched for 4 computed) are shared with nobody and most of them come from HBM again — 1.65x the
    // algorithmic traffic.  z-fastest (round 4, default; key 14 = 1: x-fastest) walks all z tiles of an (x, y) column first: the run is
    // a slab whose z halos are its own neighbours' interiors.
    const bool zfast = p.zfast != 0;
    const int r1 = zfast ? p.tz : p.tx, r3 = zfast ? p.tx : p.tz;  // radices of the first / third tile digit (the second is y)
    auto decode = [&](int idx) {
        Digits d;  // (xi, yi, zi) hold the first / second / third digit; item_of maps them to coordinates
        d.cb = idx % p.ncb;
        int tile = idx / p.ncb;
        d.xi = tile % r1;
        tile /= r1;
        d.yi = tile % p.ty;
        tile /= p.ty;
        d.zi = tile % r3;
        d.n = tile / r3;
        return d;
    };
    const Digits dG = decode(G);
    auto advance = [&](const Digits& a) {  // a + G in the mixed radix (ncb, r1, ty, r3, unbounded)
        Digits r;
        int c;
        r.cb = a.cb + dG.cb;
        c = r.cb >= p.ncb ? 1 : 0;
        r.cb -= c ? p.ncb : 0;
        r.xi = a.xi + dG.xi + c;
        c = r.xi >= r1 ? 1 : 0;
        r.xi -= c ? r1 : 0;
        r.yi = a.yi + dG.yi + c;
        c = r.yi >= p.ty ? 1 : 0;
        r.yi -= c ? p.ty : 0;
        r.zi = a.zi + dG.zi + c;
        c = r.zi >= r3 ? 1 : 0;
        r.zi -= c ? r3 : 0;
        r.n = a.n + dG.n + c;
        return r;
    };
    auto item_of = [&](const Digits& d) {
        Item c;
        c.cb = d.cb;
        c.x0 = (zfast ? d.zi : d.xi) * TX;
        c.y0 = d.yi * TY;
        c.z0 = (zfast ? d.xi : d.zi) * TZ;
        c.n = d.n;
        c.base0 = ((c.n * D + c.z0 - 1) * H + c.y0 - 1) * W + c.x0 - 1;
        c.base1 = VIRT ? ((c.n * D1 + (c.z0 >> 1)) * H1 + (c.y0 >> 1)) * W1 + (c.x0 >> 1) : 0;
        const bool bz0 = c.z0 == 0, bz1 = c.z0 + TZ >= D, by0 = c.y0 == 0, by1 = c.y0 + TY >= H, bx0 = c.x0 == 0,
                   bx1 = c.x0 + TX >= W;
        c.border = bz0 || bz1 || by0 || by1 || bx0 || bx1;
        c.inv = fdead | (bz0 ? fz0 : 0u) | (bz1 ? fz1 : 0u) | (by0 ? fy0 : 0u) | (by1 ? fy1 : 0u) | (bx0 ? fx0 : 0u) |
                (bx1 ? fx1 : 0u);
        return c;
    };
    int wi = u3d_xcd_remap(blockIdx.x, G);
    Digits dT = decode(wi);
    Item T = item_of(dT);

    // NT <= 2: the accumulators are written first by the C = 0 MFMAs of every tile's first k-step (ROW_LOAD_FIRST); NT = 3
    // keeps explicit zeroing (the extra row variant costs it registers it does not have: +90 B of spills, -3 %)
    constexpr bool ZEROC = NT < 3 && !N16;
    f32x16 acc[N16 ? 1 : MT][NT];
    f32x4 acc16[N16 ? MT : 1];  // N16: per M-tile, rows = channels 4*(lane>>4) .. +3, column = voxel lane & 15
    if constexpr (N16) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc16[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else if constexpr (!ZEROC) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    }

    // A-fragment base: lane (m,h) -> voxel (zl = w, yl = (m>>3) [+4 for mt=1], xl = m&7), channels 4h..4h+3;
    // PAIRY: yl = 2*(m>>3) (even rows only); N16: lane (v = l&15, kq = l>>4) -> voxel (yl = v>>3 [+2 per M-tile], xl = v&7),
    // channels 4kq..4kq+3 (all 16 channels of the chunk in one read)
    const int v16 = l & 15, kq = l >> 4;
    // (N16: an M-tile pairs the rows y and y + 2 — with the row stride of 164 floats that is the pairing whose 16 x 4 lanes hit
    // distinct banks in every ds_read_b128 service group, tools/lds_bank_model.py; rows y, y + 1 are 2-way conflicted)
    const int abase = N16 ? w * PS + 2 * (v16 >> 3) * RS + (v16 & 7) * CS + 4 * kq
                          : w * PS + (m >> 3) * (PAIRY ? 2 : 1) * RS + (m & 7) * CS + 4 * h;
    auto mrow = [](int mt) { return N16 ? (mt >> 1) * 4 + (mt & 1) : 4 * mt; };  // first y row of M-tile mt

    // B stream: uniform base pointer of the block's channel block + the lane's 16-byte slot (no VALU per step)
    const f32x4* wimg = reinterpret_cast<const f32x4*>(p.wp);
    const f32x4* wq = wimg + (size_t)T.cb * NT * 64;
    const int wstep = p.ntot * 64;
    f32x4 bq[RB][NT];
#pragma unroll
    for (int k = 0; k < RB - 1; ++k)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bq[k][nt] = wq[(size_t)k * wstep + nt * 64 + l];

    // per-chunk source selection of this thread's channel quad
    struct ChunkSrc {
        const float* base;
        int Cs;
        bool cok, from0;
        const float* ap;  // this thread's rows of the GroupNorm affine table (nullptr: identity)
        f32x4 lo, hi;     // the raw rows (a0,b0,a1,b1), (a2,b2,a3,b3)
    };
    // chunk_src does NOT load the GroupNorm affine: a load placed at the head of a chunk makes the compiler wait for it
    // (s_waitcnt vmcnt(0): ALL outstanding loads, a full memory round trip of 3-4 k cycles per chunk, measured as the
    // "restage gap" of the timeline twin) because the (a,b) de-interleave uses it at once.  The affine rows are loaded
    // one tap row before the halo stores that need them (load_affine_rows in the k-loop).
    auto chunk_src = [&](int ch, int n, bool live) {
        ChunkSrc c;
        const int cq = ch * CC + 4 * q;
        c.cok = live && cq < Ctot;
        c.from0 = !VIRT || cq < p.src.C0;
        c.base = !c.cok ? p.src.p0 : (c.from0 ? p.src.p0 + cq : p.src.p1 + (cq - p.src.C0));
        c.Cs = (c.from0 || !c.cok) ? p.src.C0 : p.src.C1;
        c.ap = p.src.affine ? p.src.affine + ((size_t)n * Ctot + (c.cok ? cq : 0)) * 2 : nullptr;
        c.lo = f32x4{1.f, 0.f, 1.f, 0.f};  // (a,b) pairs of channels 0,1 / 2,3 of the quad: identity
        c.hi = f32x4{1.f, 0.f, 1.f, 0.f};
        return c;
    };
    auto load_affine_rows = [&](ChunkSrc& c) {  // raw rows only: their first USE is a tap row later (halo_store)
        if constexpr (AFF) {
            if (c.ap) {
                c.lo = *reinterpret_cast<const f32x4*>(c.ap);
                c.hi = *reinterpret_cast<const f32x4*>(c.ap + 4);
            }
        }
    };
    // `masked`: the staged tile has padding items (border tile) or this thread's channel quad is dead
    auto halo_load = [&](const ChunkSrc& c, const Item& S, int it, bool masked) {
        int idx = c.from0 ? S.base0 + rel0[it] : S.base1 + rel1[VIRT ? it : 0];
        if (masked) idx = (c.cok && ((S.inv >> it) & 1u) == 0) ? idx : 0;
        return *reinterpret_cast<const f32x4*>(c.base + (size_t)idx * c.Cs);
    };
    auto halo_store = [&](float* buf, const ChunkSrc& c, const Item& S, int it, f32x4 raw, bool masked) {
        f32x4 val = raw;  // AFF=false (source without a GroupNorm affine, i.e. every dgrad launch): no per-element FMA
        if constexpr (AFF)
            val = f32x4{fmaf(raw[0], c.lo[0], c.lo[1]), fmaf(raw[1], c.lo[2], c.lo[3]), fmaf(raw[2], c.hi[0], c.hi[1]),
                        fmaf(raw[3], c.hi[2], c.hi[3])};
        if (masked) {
            const bool ok = c.cok && ((S.inv >> it) & 1u) == 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) val[e] = ok ? val[e] : 0.f;  // padding stays exactly 0
        }
        *reinterpret_cast<f32x4*>(&buf[loff[it]]) = val;
    };
    // GroupNorm statistics: the four waves add their 64-voxel sums of every tile into ONE block-level LDS row per
    // sample parity (ds_add_f64); when a wave has finished the block's last tile of a sample it arrives on that
    // parity's counter and the LAST of the four flushes the row with f64 atomics and clears it — 64*NT global atomics
    // per block and sample instead of per wave and tile (same-address f64 atomics retire at ~24 ns each: 2048 waves
    // flushing at once cost 100 us on the 32^3-tile layers).  A wave is never more than one tile ahead of the slowest,
    // so two rows suffice.
    const bool want_stats = p.out_stats != nullptr;
    const bool want_g = p.gstats != nullptr;
    auto flush_stats = [&](int n, int cb) {
        if (!(want_stats || want_g)) return;
        const int par = n & 1;
        int arrived = 0;
        asm volatile("" ::: "memory");
        if (l == 0) arrived = __hip_atomic_fetch_add(&cnt[5 + par], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        arrived = __builtin_amdgcn_readfirstlane(arrived);
        asm volatile("" ::: "memory");
        if ((arrived & 3) != 3) return;
        const int rep = p.stat_reps > 1 ? (int)(blockIdx.x % (unsigned)p.stat_reps) : 0;
        for (int k = l; k < NT * 32; k += 64) {
            double* r = &red[(par * NT * 32 + k) * 2];
            const double a = r[0], b = r[1];
            r[0] = 0.0;
            r[1] = 0.0;
            const int co = cb * NT * 32 + k;
            if (co < p.Cout) {
                // (replica row: a same-address f64 atomic retires every 19.5 ns and all 512 blocks flush a sample at the same time —
                // 10 us behind the last tile of every launch on ONE row, 1.3 us on eight; tools/atomic_bench.hip)
                double* dst = (want_stats ? p.out_stats : p.gstats) + ((size_t)rep * p.N * p.Cout + (size_t)n * p.Cout + co) * 2;
                u3d_atomic_add_f64(dst, a);
                u3d_atomic_add_f64(dst + 1, b);
            }
        }
        asm volatile("" ::: "memory");
    };
    const bool dead_quads = (Ctot % CC) != 0;  // uniform: some thread quads lie beyond the last channel

    // ---- prologue: stage chunk 0 of the first tile into buffer 0
    {
        ChunkSrc c0 = chunk_src(0, T.n, true);
        load_affine_rows(c0);
        const bool masked = T.border || dead_quads || fdead != 0;
        f32x4 v[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) v[it] = halo_load(c0, T, it, true);
#pragma unroll
        for (int it = 0; it < NIT; ++it) halo_store(lds, c0, T, it, v[it], true);
        (void)masked;
    }
    u3d_flag_signal(&cnt[0], l);

    // Phase-stagger experiment (u3d_set_tuning key 5, default off).  Blocks b and b + G/2 share a CU (observed with the
    // timeline twin: every SIMD holds one wave of each, starting within ~40 cycles of each other) and, doing identical work,
    // run their k-loops AND their epilogues at the same time.  Delaying the second half of the grid once would interleave
    // the phases — measured: no gain (see host side).
    if (p.stagger > 0 && (int)blockIdx.x >= (G + 1) / 2) {
        for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(16);
    }

    int gch = 0;     // chunks done by this wave over all tiles: buffer parity and flag targets
    int ntiles = 0;  // tiles done (timeline record)
    // Statistics of the written values.  After the epilogue's transposition lane (l&3, l>>5) of a channel quad holds one of
    // its 8 x-positions: the 4 lanes of a quad are summed with two DPP steps (no LDS round trips as __shfl_xor would
    // take), the two half-waves (and, PAIRY, the two y-row column groups) each add their sum to the block's f64 LDS row.
    // NT = 1 carries the per-lane sums in registers across the tiles of one (sample, channel block) and reduces only when
    // that changes: the per-tile epilogue is then transposition + stores only.
    constexpr bool CARRY = NT == 1;
    f32x4 sq1 = {0.f, 0.f, 0.f, 0.f}, sq2 = {0.f, 0.f, 0.f, 0.f};
    auto quad_sum = [](float v) {
        v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));  // lane ^ 1
        v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));  // lane ^ 2
        return v;
    };
    auto stat_reduce = [&](const f32x4& a1, const f32x4& a2, int n_, int nt_, int c0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = quad_sum(a1[e]), b2 = quad_sum(a2[e]);
            if ((l & 3) == 0) {
                double* r = &red[(((n_ & 1) * NT + nt_) * 32 + c0 + e) * 2];
                __hip_atomic_fetch_add(r, (double)a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(r + 1, (double)b2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    };
    while (true) {
        const int nwi = wi + G;
        const bool has_next_tile = nwi < p.total;
        const Digits dTN = has_next_tile ? advance(dT) : dT;
        const Item TN = item_of(dTN);
        const f32x4* wqn = wimg + (size_t)TN.cb * NT * 64;

        for (int ch = 0; ch < p.nchunks; ++ch, ++gch) {
            const bool last = ch + 1 == p.nchunks;
            const bool has_next = !last || has_next_tile;
            const int b = gch & 1;
            const float* cur = lds + b * TILE_FLOATS;
            float* nxt = lds + (b ^ 1) * TILE_FLOATS;
            // the chunk staged during this k-loop: the next chunk of this tile, or chunk 0 of the next tile
            const int nch_ = last ? 0 : ch + 1;
            const Item S = last ? TN : T;
            ChunkSrc cn = chunk_src(nch_, S.n, has_next);
            // uniform: does any lane have to mask an item of the staged chunk?  (the tail items of the 10th round are
            // dead in 3 of 4 waves only; they go to the dummy slot and need no masking)
            const bool masked = S.border || !has_next || (dead_quads && nch_ + 1 == p.nchunks);
            f32x4 v[NIT];
            u3d_flag_wait(&cnt[b], 4 * (gch / 2 + 1));  // all four waves have staged this chunk
            __builtin_amdgcn_s_setprio(0);
            if (ntiles == DBG_TILE && ch < 7) U3D_DBG_STAMP(8 + 2 * ch);
            if (ntiles == DBG_TILE + 1 && ch == 0) U3D_DBG_STAMP(23);  // first k-loop of the following tile starts

            // ---- 54 k-steps as 9 tap rows x 6 steps (3 taps x 2 channel-octets) of 8*NT MFMAs; tap row 0 carries
            //      the 10 halo loads of the next chunk, tap row ISTORE their LDS stores
            f32x4 aq[2][MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) aq[0][mt] = *reinterpret_cast<const f32x4*>(&cur[abase + mrow(mt) * RS]);
            const f32x4* wch0 = wq + (size_t)(ch * NSTEPL + RB - 1) * wstep;  // B fragments of step (row 0, 0) + RB-1
            // One tap row = 6 k-steps.  The rows differ in what rides along with the MFMAs (halo loads in row 0, the affine
            // rows in row ISTORE-1, the halo stores in row ISTORE, the switch to the next tile's weight image in the last
            // row); each kind is its own compile-time specialisation, so that the plain rows — 5 of 9 — carry nothing but
            // the B load, two A reads and 8*NT MFMAs per step.  (With every condition evaluated at run time in one loop body
            // the ~30 scalar instructions and 3 branches between two MFMA groups did not fit under one 64-cycle MFMA: a wave
            // alone on its SIMD ran its k-loop at 81 % of the pipe, tools/wave_timeline.py with U3D_TUNE=6:1.)
            // ROW_LOAD_FIRST: row 0 of a tile's first chunk — its very first MFMA per accumulator takes C = 0 (inline constant),
            // which replaces 32*NT v_mov of accumulator zeroing per tile
            enum { ROW_PLAIN = 0, ROW_LOAD = 1, ROW_AFFINE = 2, ROW_STORE = 3, ROW_LAST = 4, ROW_LOAD_FIRST = 5 };
            auto row_body = [&](auto kind_c, auto row_c) __attribute__((always_inline)) {
                constexpr int KIND = decltype(kind_c)::value;
                constexpr int row = decltype(row_c)::value;
                const int rz = row / RY, ry = row - RY * rz;
                const float* arow = cur + abase + rz * PS + ry * RS;
                const int nrow = row + 1;
                const float* anext = cur + abase + (nrow / RY) * PS + (nrow % RY) * RS;
                const bool do_store = KIND == ROW_STORE && has_next;
                if constexpr (KIND == ROW_AFFINE) load_affine_rows(cn);  // a whole tap row ahead of its first use
                if constexpr (KIND == ROW_STORE) {
                    if (do_store) u3d_flag_wait(&cnt[2 + (b ^ 1)], 4 * ((gch + 1) / 2));  // other buffer free
                }
#pragma unroll
                for (int s6 = 0; s6 < SPR; ++s6) {
                    constexpr int STORES_PER_STEP = (NIT + SPR - 2) / (SPR - 1);  // the halo stores of a row: spread over its first steps
                    {
                        // B fragments of step + RB-1; past the end of a tile's image continue with the next tile's.
                        // (recomputed from uniform scalars, not carried as a mutated pointer: keeps the address in SGPRs)
                        const f32x4* wsrc = wch0 + (size_t)(row * SPR + s6) * wstep;
                        if (row * SPR + s6 + RB - 1 >= NSTEPL && last) wsrc = wqn + (size_t)(row * SPR + s6 + RB - 1 - NSTEPL) * wstep;
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) bq[(row * SPR + s6 + RB - 1) % RB][nt] = wsrc[nt * 64 + l];
                    }
                    // vmcnt retires in order: a B fragment fetched AFTER a halo load cannot be consumed before that halo load (an
                    // HBM / remote-L2 round trip) has landed.  Spread two per k-step, every B fragment of the row sat behind the halo
                    // loads of its own step with only RB-1 steps of slack; issued as ONE burst right after step 0's B fetch, the first
                    // B fragment behind them is step 1's — RB steps of slack — and the slack grows by a step from there on.
                    if constexpr (KIND == ROW_LOAD || KIND == ROW_LOAD_FIRST) {
                        if (s6 == 0) {
                            if (masked) {
#pragma unroll
                                for (int it = 0; it < NIT; ++it) v[it] = halo_load(cn, S, it, true);
                            } else {
#pragma unroll
                                for (int it = 0; it < NIT; ++it) v[it] = halo_load(cn, S, it, false);
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if constexpr (N16) {
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt)
                                acc16[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[(row * SPR + s6) % RB][0][j], aq[(row * SPR + s6) & 1][mt][j], acc16[mt], 0, 0, 0);
                        } else {
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                                for (int mt = 0; mt < MT; ++mt) {
                                    if (KIND == ROW_LOAD_FIRST && s6 == 0 && j == 0) {
                                        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[0][mt][0], bq[0][nt][0], zero, 0, 0, 0);
                                    } else {
                                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[(row * SPR + s6) & 1][mt][j], bq[(row * SPR + s6) % RB][nt][j], acc[mt][nt], 0, 0, 0);
                                    }
                                }
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (s6 < SPR - 1) {
                        // next k-step of this tap row: (tap, channel octet) -> (s+1)>>1, (s+1)&1; N16: the next tap, all 16 channels
                        const int aoff = N16 ? (s6 + 1) * CS : ((s6 + 1) >> 1) * CS + 8 * ((s6 + 1) & 1);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) aq[(row * SPR + s6 + 1) & 1][mt] = *reinterpret_cast<const f32x4*>(&arow[mrow(mt) * RS + aoff]);
                    } else if constexpr (KIND != ROW_LAST) {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) aq[(row * SPR + s6 + 1) & 1][mt] = *reinterpret_cast<const f32x4*>(&anext[mrow(mt) * RS]);
                    }
                    if constexpr (KIND == ROW_STORE) {
                        if (do_store && s6 * STORES_PER_STEP < NIT) {
#pragma unroll
                            for (int k = 0; k < STORES_PER_STEP; ++k) {
                                const int it = s6 * STORES_PER_STEP + k;
                                if (it < NIT) {
                                    if (masked)
                                        halo_store(nxt, cn, S, it, v[it], true);
                                    else
                                        halo_store(nxt, cn, S, it, v[it], false);
                                }
                            }
                            if ((s6 + 1) * STORES_PER_STEP >= NIT) u3d_flag_signal(&cnt[b ^ 1], l);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 2; j < 4; ++j) {
                        if constexpr (N16) {
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt)
                                acc16[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[(row * SPR + s6) % RB][0][j], aq[(row * SPR + s6) & 1][mt][j], acc16[mt], 0, 0, 0);
                        } else {
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                                for (int mt = 0; mt < MT; ++mt)
                                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[(row * SPR + s6) & 1][mt][j], bq[(row * SPR + s6) % RB][nt][j], acc[mt][nt], 0, 0, 0);
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            static_assert(ISTORE - 1 > 1 && ISTORE + 1 < NROWS - 1, "row kinds must not collide");
            static_assert(NSTEPL % RB == 0, "B ring slots are indexed by the k-step within the chunk");
            using R0 = std::integral_constant<int, 0>;
            if (ZEROC && ch == 0)
                row_body(std::integral_constant<int, ROW_LOAD_FIRST>{}, R0{});
            else
                row_body(std::integral_constant<int, ROW_LOAD>{}, R0{});
            // plain rows as straight-line code (compile-time row: LDS offsets and ring slots become immediates, no loop-carried
            // scalars): measured per row alone on a SIMD 3140-3260 cycles straight-line vs 3440 inside a run-time loop (ideal 3072)
            u3d_for_rows<1, ISTORE - 1>([&](auto r) { row_body(std::integral_constant<int, ROW_PLAIN>{}, r); });
            row_body(std::integral_constant<int, ROW_AFFINE>{}, std::integral_constant<int, ISTORE - 1>{});
            row_body(std::integral_constant<int, ROW_STORE>{}, std::integral_constant<int, ISTORE>{});
            u3d_for_rows<ISTORE + 1, NROWS - 1>([&](auto r) { row_body(std::integral_constant<int, ROW_PLAIN>{}, r); });
            row_body(std::integral_constant<int, ROW_LAST>{}, std::integral_constant<int, NROWS - 1>{});
            if (ntiles == DBG_TILE && ch < 7) U3D_DBG_STAMP(9 + 2 * ch);
            u3d_flag_signal(&cnt[2 + b], l);  // this wave no longer reads buffer b
        }
        __builtin_amdgcn_s_setprio(3);

        // ---- epilogue of tile T (fully inside the volume: no voxel masks).  C/D layout of 32x32 MFMA: col = lane&31
        //      (cout), row = (r&3) + 8*(r>>2) + 4*(lane>>5); M-tile row -> (y = row>>3, x = row&7).  A 4x4 transpose
        //      inside every lane quad (two DPP butterfly stages) leaves lane j of quad k with the 4 consecutive
        //      channels 4k..4k+3 of voxel x = j + 4h: 16-byte stores and 16-byte loads of x for the GroupNorm sums.
        if (ntiles == DBG_TILE) U3D_DBG_STAMP(5);
        if constexpr (N16) {
            // accumulator of M-tile mt, lane (v = l&15, kq = l>>4): channels 4kq .. 4kq+3 of voxel (z, y0 + mrow(mt) + 2*(v>>3), x0 + (v&7))
            const int n = T.n;
            const int z = T.z0 + w;
            const int co = 4 * kq;
            const bool cok = co < p.Cout;
            const int vlane = ((n * D + z) * H + T.y0 + 2 * (v16 >> 3)) * W + T.x0 + (v16 & 7);  // this lane's voxel of M-tile 0
            float* orow = p.out + (size_t)vlane * p.Cout + co;
            const size_t rstep = (size_t)W * p.Cout;  // one y row of the output
            // RAG: is this lane's voxel of M-tile mt inside the volume?
            auto vin = [&](int mt) { return !RAG || (z < D && T.y0 + mrow(mt) + 2 * (v16 >> 3) < H && T.x0 + (v16 & 7) < W); };
            f32x4 xv[MT];
            if (want_g) {
                // gx (the layer's input, for the GroupNorm-backward sums): plain, or its channels beyond C0 from the exact-2x low-res half
                const bool from0 = co < p.gx.C0 || !cok;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int y = T.y0 + mrow(mt) + 2 * (v16 >> 3), x = T.x0 + (v16 & 7);
                    const int xi1 = ((n * p.gx.D1 + (z >> 1)) * p.gx.H1 + (y >> 1)) * p.gx.W1 + (x >> 1);
                    const float* xp = !(cok && vin(mt)) ? p.gx.p0
                                           : (from0 ? p.gx.p0 + (size_t)(vlane + mrow(mt) * W) * p.gx.C0 + co
                                                    : p.gx.p1 + (size_t)xi1 * p.gx.C1 + (co - p.gx.C0));
                    xv[mt] = *reinterpret_cast<const f32x4*>(xp);
                }
            } else if (p.res) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    xv[mt] = *reinterpret_cast<const f32x4*>(p.res + (vin(mt) ? (size_t)(vlane + mrow(mt) * W) * p.Cout + (cok ? co : 0) : (size_t)0));
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f32x4 val = acc16[mt];
                if (p.res) val += xv[mt];
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) val[e] = fmaxf(val[e], 0.f);
                }
                if (cok && vin(mt)) *reinterpret_cast<f32x4*>(orow + mrow(mt) * rstep) = val;
                if (RAG || p.Cout % 16 != 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) val[e] = (cok && vin(mt)) ? val[e] : 0.f;
                }
                sq1 += val;  // (CARRY: NT == 1) running sums of this lane's four channels over its voxels and tiles
                sq2 += want_g ? val * xv[mt] : val * val;
                acc16[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        } else
        if (p.stagger != -1) {  // (u3d_set_tuning key 5 = -1: TIMING-ONLY ablation without the epilogue — wrong results by design)
            const int n = T.n, cb = T.cb;
            const int z = T.z0 + w;
            const int cq = (l >> 2) & 7, vl = (l & 3) + 4 * h;
            const bool odd = (l & 1) != 0, hi = (l & 2) != 0;
            auto xlane = [](float v, auto ctrl) {
                return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), decltype(ctrl)::value, 0xF, 0xF, true));
            };
            using X1 = std::integral_constant<int, 0xB1>;  // quad_perm [1,0,3,2]: value of lane ^ 1
            using X2 = std::integral_constant<int, 0x4E>;  // quad_perm [2,3,0,1]: value of lane ^ 2
            const int vrow = ((n * D + z) * H + T.y0) * W + T.x0 + vl;  // voxel of row 0; row st adds st * W
            // low-res (exact 2x) voxel of row 0 of gx's upsampled half; row st is (st >> 1) low-res rows further
            const int xrow = ((n * p.gx.D1 + (z >> 1)) * p.gx.H1 + (T.y0 >> 1)) * p.gx.W1 + ((T.x0 + vl) >> 1);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x4 q1 = {0.f, 0.f, 0.f, 0.f}, q2 = {0.f, 0.f, 0.f, 0.f};
                f32x4 tq[4 * MT];  // row st = 4*mt + bi: the lane's channel quad at voxel (y0 + yoff(st), x0 + vl)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int bi = 0; bi < 4; ++bi) {
                        const float a0 = acc[mt][nt][4 * bi + 0], a1 = acc[mt][nt][4 * bi + 1];
                        const float a2 = acc[mt][nt][4 * bi + 2], a3 = acc[mt][nt][4 * bi + 3];
                        const float t0 = xlane(a1, X1{}), t1 = xlane(a0, X1{}), t2 = xlane(a3, X1{}), t3 = xlane(a2, X1{});
                        const float c0 = odd ? t0 : a0, c1 = odd ? a1 : t1, c2 = odd ? t2 : a2, c3 = odd ? a3 : t3;
                        const float u0 = xlane(c2, X2{}), u2 = xlane(c0, X2{}), u1 = xlane(c3, X2{}), u3 = xlane(c1, X2{});
                        tq[4 * mt + bi] = f32x4{hi ? u0 : c0, hi ? u1 : c1, hi ? c2 : u2, hi ? c3 : u3};
                    }
                // PAIRY: column quad cq = (second-row flag, channel quad): columns 16-31 are the voxels one row below
                const int ysh = PAIRY ? (cq >> 2) : 0;
                const int co = PAIRY ? 4 * (cq & 3) : (cb * NT + nt) * 32 + 4 * cq;
                auto yoff = [&](int st) { return PAIRY ? 2 * st + ysh : st; };
                // RAG: a lane beyond W neither stores nor counts (folded into cok); rows beyond H / planes beyond D are uniform skips
                const bool cok = co < p.Cout && (!RAG || T.x0 + vl < W);
                auto row_in = [&](int st) { return !RAG || (z < D && T.y0 + yoff(st) < H); };  // uniform
                float* orow = p.out + (size_t)vrow * p.Cout + co;
                const bool xfrom0 = co < p.gx.C0 || !cok;
                const float* xb = !cok ? p.gx.p0 : (xfrom0 ? p.gx.p0 + co : p.gx.p1 + (co - p.gx.C0));
                const int xcs = xfrom0 ? p.gx.C0 : p.gx.C1;
#pragma unroll
                for (int half = 0; half < MT; ++half) {  // batches of 4 rows bound the live registers
                    f32x4 xv[4];
                    if (want_g) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int yo = yoff(4 * half + k);
                            int xi = xfrom0 ? vrow + yo * W : xrow + (yo >> 1) * p.gx.W1;
                            if constexpr (RAG) xi = (cok && row_in(4 * half + k)) ? xi : 0;
                            xv[k] = *reinterpret_cast<const f32x4*>(xb + (size_t)xi * xcs);
                        }
                    } else if (p.res) {  // residual rows (same voxels / channels as the output rows)
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            size_t ri = (size_t)(vrow + yoff(4 * half + k) * W) * p.Cout + (cok ? co : 0);
                            if constexpr (RAG) ri = (cok && row_in(4 * half + k)) ? ri : 0;
                            xv[k] = *reinterpret_cast<const f32x4*>(p.res + ri);
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int st = 4 * half + k;
                        if (!row_in(st)) continue;
                        f32x4 val = tq[st];
                        if (p.res) val += xv[k];
                        if (p.relu) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) val[e] = fmaxf(val[e], 0.f);
                        }
                        if (cok) *reinterpret_cast<f32x4*>(orow + (size_t)yoff(st) * W * p.Cout) = val;
                        if (RAG || p.Cout % 32 != 0) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) val[e] = cok ? val[e] : 0.f;
                        }
                        q1 += val;
                        q2 += want_g ? val * xv[k] : val * val;
                    }
                }
                if (want_stats || want_g) {
                    if constexpr (CARRY) {  // NT = 1: keep per-lane running sums across the tiles, reduce at the flush
                        sq1 += q1;
                        sq2 += q2;
                    } else {
                        stat_reduce(q1, q2, n, nt, co - (PAIRY ? 0 : (cb * NT + nt) * 32));
                    }
                }
            }
        }
        if (ntiles == DBG_TILE) U3D_DBG_STAMP(22);  // epilogue of the stamped tile done
        ++ntiles;
        // statistics are per (sample, channel): flush this wave's LDS row when the sample or the channel block changes
        if (!has_next_tile || TN.n != T.n || TN.cb != T.cb) {
            if constexpr (CARRY) {
                if (want_stats || want_g) {
                    if constexpr (N16) {
                        // the 16 lanes v of a k-group hold 16 voxels of the same four channels
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float a = sq1[e], b2 = sq2[e];
#pragma unroll
                            for (int msk = 8; msk > 0; msk >>= 1) {
                                a += __shfl_xor(a, msk);
                                b2 += __shfl_xor(b2, msk);
                            }
                            if (v16 == 0) {
                                double* r = &red[(((T.n & 1) * NT) * 32 + 4 * kq + e) * 2];
                                __hip_atomic_fetch_add(r, (double)a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                __hip_atomic_fetch_add(r + 1, (double)b2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                        }
                    } else {
                        const int cq_ = (l >> 2) & 7;
                        stat_reduce(sq1, sq2, T.n, 0, PAIRY ? 4 * (cq_ & 3) : 4 * cq_);
                    }
                    sq1 = f32x4{0.f, 0.f, 0.f, 0.f};
                    sq2 = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            flush_stats(T.n, T.cb);
        }
        if (!has_next_tile) break;
        if constexpr (!ZEROC) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
        }
        wi = nwi;
        dT = dTN;
        T = TN;
        wq = wqn;
    }
    if constexpr (DBG) {
        if (l == 0) dbgw[4] = ntiles;
    }
    U3D_DBG_STAMP(6);
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
constexpr int DUMMY_OFF = G_FLOATS + DZ_FLOATS;          // one float4 slot that swallows the tail staging items
constexpr int BUF_FLOATS = G_FLOATS + DZ_FLOATS + 4;     // 16900 floats = 67600 B per staging buffer
constexpr int LDS_FLOATS = 2 * BUF_FLOATS;               // double buffered: 135200 B -> one 8-wave block per CU
constexpr int NTHR = 512;
constexpr int NITEMS_G = HZ * HY * HX * (CC / 4);        // 3200
constexpr int NIT_G = (NITEMS_G + NTHR - 1) / NTHR;      // 7
constexpr int NIT_DZ = TV * 8 / NTHR;                    // 2
constexpr int MAX_MAP_INTS = 4096;                       // LDS copy of the nearest-upsample index maps (D + H + W)
}  // namespace wg

struct WgradParams {
    u3d_src_t src;
    const float* dz;
    float* partial;
    int N, D, H, W, Cout;
    int nchunks, nkb, S;
    int tz, ty, tx, ntiles, tps;
    int vec, dzvec;
    // generic staging only (round 5, u3d_conv3d_wgrad_box): the voxels v of dz that take part — tiles are enumerated over this box
    // and dz outside it reads as zero; g is read from the whole volume.  Default: the whole volume.
    int bz0, by0, bx0, bz1, by1, bx1;
};

// One block = 8 waves (512 threads) per CU: wave w owns the 7 taps {tg, tg+4, ..} (tg = w & 3) on voxel half
// hf = w >> 2 (z-plane hf of the 2x8x8 tile), i.e. the block carries TWO independent 27-tap accumulator sets that
// are written out as two split-K partials.  LDS holds two staging buffers: while the 224 MFMAs per wave of tile i
// run on buffer b, the 7 + 2 staging loads of tile i+1 are issued (one every two voxel-pair groups) and written to
// buffer b^1 later in the same loop with the GroupNorm affine applied — one barrier per tile, nothing else exposed.
// The nearest-upsample index maps of a virtual source are copied to LDS once per block so that the prefetch
// address arithmetic never waits on a dependent global load inside the MFMA stream.
//
// f32 MFMA executes on the SIMD's FMA lanes: every VALU instruction any co-resident wave issues costs ~5 cycles of
// MFMA time (tools/mfma_mix.hip; SALU, s_waitcnt and LDS reads are free).  REG (every tile fully inside the volume,
// plain or exact-2x source) therefore replaces the per-item index arithmetic (~40 VALU per staged item and tile) by
// per-thread constants computed once per block: index = tile base + rel[item], validity = one bit of a mask built
// from six per-face item masks and uniform tile-position tests.
//
// PAIR (Cin <= 16: the single input chunk fills only half of the 32 MFMA rows): rows 0-15 carry the 16 channels at
// tap 2q, rows 16-31 the same channels at tap 2q+1 — one MFMA serves two taps, 14 tap pairs instead of 27 taps
// (wave w owns pairs {w&3, +4, ..}: 4 accumulators instead of 7).
template <bool VEC, bool REG = false, bool PAIR = false>
__global__ __launch_bounds__(512, 2) void conv3d_wgrad_kernel(const WgradParams p) {
    using namespace wg;
    constexpr int NA = PAIR ? 4 : 7;  // accumulators (taps / tap pairs) per wave
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int* zmapl = reinterpret_cast<int*>(lds + 2 * BUF_FLOATS);  // [D | H | W] (virtual source only)
    __builtin_amdgcn_s_setprio(3);  // non-MFMA phases outrank co-resident k-loops (see conv3d_mfma_kernel)
    const int t = threadIdx.x;
    const int l = t & 63, i = l & 31, h = l >> 5;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int tg = w & 3, hf = w >> 2;

    const int logical = u3d_xcd_remap(blockIdx.x, gridDim.x);
    const int kb = logical % p.nkb;
    const int chunk = (logical / p.nkb) % p.nchunks;
    const int s = logical / (p.nkb * p.nchunks);
    const int D = p.D, H = p.H, W = p.W;
    const int Ctot = p.src.C0 + p.src.C1;
    int* ymapl = zmapl + D;
    int* xmapl = ymapl + H;
    if (p.src.C1 > 0) {
        for (int k = t; k < D; k += NTHR) zmapl[k] = p.src.zmap[k];
        for (int k = t; k < H; k += NTHR) ymapl[k] = p.src.ymap[k];
        for (int k = t; k < W; k += NTHR) xmapl[k] = p.src.xmap[k];
        __syncthreads();
    }

    f32x16 acc[NA];
#pragma unroll
    for (int k = 0; k < NA; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;

    // this thread's channel quad of g (fixed per block) and of dz
    const int q = t & 7, tv = t >> 3;  // tv in [0,64)
    const int cq = chunk * CC + 4 * q;
    const int co = kb * 32 + 4 * q;
    const bool cok = cq < Ctot, from0 = cq < p.src.C0, dcok = co < p.Cout;
    const float* gbase = !cok ? p.src.p0 : (from0 ? p.src.p0 + cq : p.src.p1 + (cq - p.src.C0));
    const int Cs = (from0 || !cok) ? p.src.C0 : p.src.C1;
    const float* dzbase = p.dz + (dcok ? co : 0);

    // REG: per-thread constants of the 7 + 2 staged items
    constexpr int NPFC = NIT_G + NIT_DZ;
    int rel[NPFC], loff[NPFC];
    unsigned fz0 = 0, fz1 = 0, fy0 = 0, fy1 = 0, fx0 = 0, fx1 = 0, fdead = 0;
    if constexpr (REG) {
        // RAGGED volumes (round 5: D % 2, H % 8, W % 8 need not vanish — the reference's shipped patch is 80 x 170 x 170,
        // resources/3DUnet_confocal_boundary/train_config.yml:94): the LAST tile of an axis overhangs the volume.  Its far-face
        // mask simply covers every item from the first plane beyond the volume on (halo coordinate >= valid voxels + 1; for an
        // exact fit that is the halo plane HZ-1 as before); the staged values there are the convolution's zero padding / dz = 0,
        // so the overhanging voxels add nothing.  Still six per-thread masks built once per block, no per-tile arithmetic.
        const int vz = D % TZ ? D % TZ : TZ, vy = H % TY ? H % TY : TY, vx = W % TX ? W % TX : TX;  // valid voxels of the last tile
#pragma unroll
        for (int it = 0; it < NIT_G; ++it) {
            const int vox = tv + 64 * it;
            const int hz = vox / (HY * HX);
            const int rem = vox - hz * (HY * HX);
            const int hy = rem / HX;
            const int hx = rem - hy * HX;
            const bool in = vox < HZ * HY * HX;
            // exact 2x: tile origins are even, so ((z0 - 1 + hz) >> 1) = z0/2 + ((hz - 1) >> 1)
            rel[it] = from0 ? ((hz - 1) * H + (hy - 1)) * W + (hx - 1)
                            : (((hz - 1) >> 1) * p.src.H1 + ((hy - 1) >> 1)) * p.src.W1 + ((hx - 1) >> 1);
            loff[it] = in ? hz * PSg + hy * RSg + hx * CSg + 4 * q : DUMMY_OFF;
            fz0 |= (in && hz == 0 ? 1u : 0u) << it;
            fz1 |= (in && hz > vz ? 1u : 0u) << it;
            fy0 |= (in && hy == 0 ? 1u : 0u) << it;
            fy1 |= (in && hy > vy ? 1u : 0u) << it;
            fx0 |= (in && hx == 0 ? 1u : 0u) << it;
            fx1 |= (in && hx > vx ? 1u : 0u) << it;
            fdead |= (in && cok ? 0u : 1u) << it;
        }
#pragma unroll
        for (int it = 0; it < NIT_DZ; ++it) {
            const int vox = tv + 64 * it;
            rel[NIT_G + it] = ((vox >> 6) * H + ((vox >> 3) & 7)) * W + (vox & 7);
            loff[NIT_G + it] = G_FLOATS + vox * 32 + 4 * q;
            fz1 |= ((vox >> 6) >= vz ? 1u : 0u) << (NIT_G + it);
            fy1 |= (((vox >> 3) & 7) >= vy ? 1u : 0u) << (NIT_G + it);
            fx1 |= ((vox & 7) >= vx ? 1u : 0u) << (NIT_G + it);
        }
        fdead |= (dcok ? 0u : 3u) << NIT_G;
    }
    // invalid-item mask of a tile: uniform face tests select the per-thread face masks
    auto tile_invalid = [&](int z0, int y0, int x0) {
        unsigned inv = fdead;
        inv |= z0 == 0 ? fz0 : 0u;
        inv |= z0 + TZ >= D ? fz1 : 0u;
        inv |= y0 == 0 ? fy0 : 0u;
        inv |= y0 + TY >= H ? fy1 : 0u;
        inv |= x0 == 0 ? fx0 : 0u;
        inv |= x0 + TX >= W ? fx1 : 0u;
        return inv;
    };

    struct TileC {
        int n, z0, y0, x0;
        int base, dzb;  // REG: voxel index of the tile origin in the g source of this thread / in dz
        unsigned inv;   // REG: invalid-item bits
    };
    // Tiles are walked consecutively: the index is decoded once (three integer divisions) and then advanced digit-wise with
    // carries — scalar adds / compares only.  (A VALU instruction issued while the SIMD's other wave is in its MFMA stream
    // waits ~45 cycles for an MFMA boundary: 100 VALU of divisions per 14 k-cycle tile were a double-digit percentage.)
    struct TileIdx {
        int xi, yi, zi, n;
    };
    auto tile_decode = [&](int tile) {
        TileIdx d;
        d.xi = tile % p.tx;
        tile /= p.tx;
        d.yi = tile % p.ty;
        tile /= p.ty;
        d.zi = tile % p.tz;
        d.n = tile / p.tz;
        return d;
    };
    auto tile_next = [&](const TileIdx& a) {
        TileIdx r = a;
        r.xi += 1;
        int c = r.xi >= p.tx ? 1 : 0;
        r.xi = c ? 0 : r.xi;
        r.yi += c;
        c = r.yi >= p.ty ? 1 : 0;
        r.yi = c ? 0 : r.yi;
        r.zi += c;
        c = r.zi >= p.tz ? 1 : 0;
        r.zi = c ? 0 : r.zi;
        r.n += c;
        return r;
    };
    auto tile_coords = [&](const TileIdx& d) {
        TileC c;
        c.x0 = (REG ? 0 : p.bx0) + d.xi * TX;
        c.y0 = (REG ? 0 : p.by0) + d.yi * TY;
        c.z0 = (REG ? 0 : p.bz0) + d.zi * TZ;
        c.n = d.n;
        c.base = c.dzb = 0;
        c.inv = 0;
        if constexpr (REG) {
            c.dzb = ((c.n * D + c.z0) * H + c.y0) * W + c.x0;
            const int b1 = ((c.n * p.src.D1 + (c.z0 >> 1)) * p.src.H1 + (c.y0 >> 1)) * p.src.W1 + (c.x0 >> 1);
            c.base = from0 ? c.dzb : b1;
            c.inv = tile_invalid(c.z0, c.y0, c.x0);
        }
        return c;
    };
    // g halo item `it` of a tile: LDS offset, validity and (clamped, always valid) global voxel index
    auto g_item = [&](const TileC& c, int it, int& off, bool& ok, int& idx) {
        const int vox = tv + 64 * it;
        const int hz = vox / (HY * HX);
        const int rem = vox - hz * (HY * HX);
        const int hy = rem / HX;
        const int hx = rem - hy * HX;
        off = vox < HZ * HY * HX ? hz * PSg + hy * RSg + hx * CSg + 4 * q : DUMMY_OFF;  // tail items -> dummy slot
        const int gz = c.z0 - 1 + hz, gy = c.y0 - 1 + hy, gxx = c.x0 - 1 + hx;
        ok = cok && vox < HZ * HY * HX && gz >= 0 && gz < D && gy >= 0 && gy < H && gxx >= 0 && gxx < W;
        const int zc = min(max(gz, 0), D - 1), yc = min(max(gy, 0), H - 1), xc = min(max(gxx, 0), W - 1);
        idx = ((c.n * D + zc) * H + yc) * W + xc;
        if (p.src.C1 > 0) {  // uniform branch; the select below is per lane (a chunk may straddle C0)
            const int idx1 = ((c.n * p.src.D1 + zmapl[zc]) * p.src.H1 + ymapl[yc]) * p.src.W1 + xmapl[xc];
            idx = from0 ? idx : idx1;
        }
        idx = ok ? idx : 0;
    };
    auto dz_item = [&](const TileC& c, int it, bool& ok, int& idx) {
        const int vox = tv + 64 * it;
        const int z = c.z0 + (vox >> 6), y = c.y0 + ((vox >> 3) & 7), x = c.x0 + (vox & 7);
        ok = dcok && z < p.bz1 && y < p.by1 && x < p.bx1;
        idx = ok ? ((c.n * D + z) * H + y) * W + x : 0;
    };
    auto load_affine = [&](int n, f32x4& ga, f32x4& gb) {
        ga = f32x4{1.f, 1.f, 1.f, 1.f};
        gb = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.src.affine) {
            const float* ap = p.src.affine + ((size_t)n * Ctot + (cok ? cq : 0)) * 2;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(ap);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(ap + 4);
            ga = f32x4{lo[0], lo[2], hi[0], hi[2]};
            gb = f32x4{lo[1], lo[3], hi[1], hi[3]};
        }
    };
    auto pf_load = [&](const TileC& c, int k) {  // k < NIT_G: g item, else dz item
        bool ok;
        int idx, off;
        if constexpr (REG) {
            ok = ((c.inv >> k) & 1u) == 0;
            if (k < NIT_G) return *reinterpret_cast<const f32x4*>(gbase + (size_t)(ok ? c.base + rel[k] : 0) * Cs);
            return *reinterpret_cast<const f32x4*>(dzbase + (size_t)(ok ? c.dzb + rel[k] : 0) * p.Cout);
        }
        if (k < NIT_G) {
            g_item(c, k, off, ok, idx);
            return *reinterpret_cast<const f32x4*>(gbase + (size_t)idx * Cs);
        }
        dz_item(c, k - NIT_G, ok, idx);
        return *reinterpret_cast<const f32x4*>(dzbase + (size_t)idx * p.Cout);
    };
    auto pf_store = [&](float* buf, const TileC& c, int k, f32x4 raw, const f32x4& ga, const f32x4& gb) {
        bool ok;
        int idx, off;
        if constexpr (REG) {
            ok = ((c.inv >> k) & 1u) == 0;
            f32x4 val = k < NIT_G ? raw * ga + gb : raw;
#pragma unroll
            for (int e = 0; e < 4; ++e) val[e] = ok ? val[e] : 0.f;  // zero padding AFTER the affine
            *reinterpret_cast<f32x4*>(&buf[loff[k]]) = val;
            return;
        }
        if (k < NIT_G) {
            g_item(c, k, off, ok, idx);
            f32x4 val = raw * ga + gb;
#pragma unroll
            for (int e = 0; e < 4; ++e) val[e] = ok ? val[e] : 0.f;  // zero padding AFTER the affine
            *reinterpret_cast<f32x4*>(&buf[off]) = val;
        } else {
            dz_item(c, k - NIT_G, ok, idx);
#pragma unroll
            for (int e = 0; e < 4; ++e) raw[e] = ok ? raw[e] : 0.f;
            *reinterpret_cast<f32x4*>(&buf[G_FLOATS + (tv + 64 * (k - NIT_G)) * 32 + 4 * q]) = raw;
        }
    };
    // scalar-load staging (channel counts that are not multiples of 4): synchronous
    auto stage_scalar = [&](float* buf, const TileC& c) {
        f32x4 ga, gb;
        u3d_load_affine(p.src.affine, c.n, Ctot, cq, false, ga, gb);
#pragma unroll 1
        for (int it = 0; it < NIT_G; ++it) {
            const int vox = tv + 64 * it;
            if (vox >= HZ * HY * HX) continue;
            const int hz = vox / (HY * HX), rem = vox - hz * (HY * HX), hy = rem / HX, hx = rem - hy * HX;
            const int gz = c.z0 - 1 + hz, gy = c.y0 - 1 + hy, gxx = c.x0 - 1 + hx;
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (gz >= 0 && gz < D && gy >= 0 && gy < H && gxx >= 0 && gxx < W) {
                int v0, v1;
                u3d_vox_index(p.src, c.n, gz, gy, gxx, D, H, W, v0, v1);
                val = u3d_load_quad(p.src, v0, v1, cq, false) * ga + gb;
            }
            *reinterpret_cast<f32x4*>(&buf[hz * PSg + hy * RSg + hx * CSg + 4 * q]) = val;
        }
#pragma unroll 1
        for (int it = 0; it < NIT_DZ; ++it) {
            const int vox = tv + 64 * it;
            const int z = c.z0 + (vox >> 6), y = c.y0 + ((vox >> 3) & 7), x = c.x0 + (vox & 7);
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (z < p.bz1 && y < p.by1 && x < p.bx1) {
                const float* sp = p.dz + ((size_t)((c.n * D + z) * H + y) * W + x) * p.Cout + co;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (co + e < p.Cout) val[e] = sp[e];
            }
            *reinterpret_cast<f32x4*>(&buf[G_FLOATS + vox * 32 + 4 * q]) = val;
        }
    };

    constexpr int NPF = NIT_G + NIT_DZ;  // 9 prefetch loads per thread and tile
    constexpr int ST0 = 20;              // first voxel-pair group that writes a prefetched item to the other buffer
    static_assert(2 * (NPF - 1) < ST0 && ST0 + NPF <= 32, "prefetch schedule must fit the 32 groups of a tile");
    const int tile_begin = s * p.tps, tile_end = min(p.ntiles, (s + 1) * p.tps);

    // ---- prologue: stage the first tile into buffer 0
    TileIdx tix = tile_decode(tile_begin);
    if (tile_begin < tile_end) {
        const TileC c = tile_coords(tix);
        if constexpr (VEC) {
            f32x4 ga, gb;
            load_affine(c.n, ga, gb);
            f32x4 v[NPF];
#pragma unroll
            for (int k = 0; k < NPF; ++k) v[k] = pf_load(c, k);
#pragma unroll
            for (int k = 0; k < NPF; ++k) pf_store(lds, c, k, v[k], ga, gb);
        } else {
            stage_scalar(lds, c);
        }
    }
    __syncthreads();

    // A-operand bases: lane (i = channel, h = voxel parity) + the wave's z-plane + its 7 tap offsets
    int abase[NA];
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        const int tap = PAIR ? 2 * (tg + 4 * k) + (i >> 4) : tg + 4 * k;
        const int toff = tap < 27 ? (tap / 9) * PSg + ((tap / 3) % 3) * RSg + (tap % 3) * CSg : 0;
        abase[k] = (PAIR ? (i & 15) : i) + h * CSg + hf * PSg + toff;
    }
    int bbase = G_FLOATS + (hf * 64 + h) * 32 + i;
    int cur = 0;  // buffer holding the current tile
    __builtin_amdgcn_s_setprio(0);

    // GroupNorm affine of this block's channel quad: depends on the sample only.  Re-loading it at the head of every
    // tile costs a full memory round trip per tile (the compiler waits for ALL outstanding loads before the (a,b)
    // de-interleave): keep it in registers and reload on a sample change (a uniform, rare branch).
    f32x4 gan = {1.f, 1.f, 1.f, 1.f}, gbn = {0.f, 0.f, 0.f, 0.f};
    int n_aff = -1;
    for (int tile = tile_begin; tile < tile_end; ++tile) {
        const bool has_next = tile + 1 < tile_end;
        if (has_next) tix = tile_next(tix);
        const TileC cn = tile_coords(tix);
        float* nbuf = lds + (cur ^ 1) * BUF_FLOATS;
        f32x4 v[NPF];
        if constexpr (VEC) {
            if (cn.n != n_aff) {
                load_affine(cn.n, gan, gbn);
                n_aff = cn.n;
            }
        }
        const float* gl = lds + cur * BUF_FLOATS;

        // ---- 32 voxel-pair groups x 7 taps.  A[i=c][k=h] = g[voxel 2t+h shifted by tap][c], B[k=h][j] = dz[voxel][j]
        float aop[2][NA], bop[2];
        bop[0] = gl[bbase];
#pragma unroll
        for (int k = 0; k < NA; ++k) aop[0][k] = gl[abase[k]];
#pragma unroll
        for (int g = 0; g < 32; ++g) {
            if constexpr (VEC) {
                if (g % 2 == 0 && g / 2 < NPF) v[g / 2] = pf_load(cn, g / 2);
                // (after the last tile this re-stages it into the idle buffer: harmless and branch-free)
                if (g >= ST0 && g - ST0 < NPF) pf_store(nbuf, cn, g - ST0, v[g - ST0], gan, gbn);
            }
            if (g + 1 < 32) {
                const int row = (g + 1) >> 2, tq = (g + 1) & 3;
                const int goff = row * RSg + 2 * tq * CSg;
                bop[(g + 1) & 1] = gl[bbase + (row * 8 + 2 * tq) * 32];
#pragma unroll
                for (int k = 0; k < NA; ++k) aop[(g + 1) & 1][k] = gl[abase[k] + goff];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < NA; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(aop[g & 1][k], bop[g & 1], acc[k], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (!VEC) {
            if (has_next) stage_scalar(nbuf, cn);
        }
        __syncthreads();
        cur ^= 1;
    }

    __builtin_amdgcn_s_setprio(3);
    // ---- fold the two voxel halves (fixed order: hf 0 + hf 1) through LDS, then
    //      partial[s][chunk][kb][tap][c][k]; D rows = c, cols = k
    float* red = lds;  // [tg][k][r][lane]: 4 * 7 * 16 * 64 floats = 112 KiB of the (now idle) staging buffers
    if (hf == 1) {
#pragma unroll
        for (int k = 0; k < NA; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((tg * 7 + k) * 16 + r) * 64 + l] = acc[k][r];
    }
    __syncthreads();
    if (hf == 0) {
        float* dst = p.partial + ((size_t)((s * p.nchunks + chunk) * p.nkb + kb) * 27) * 1024;
#pragma unroll
        for (int k = 0; k < NA; ++k) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;  // D row of this register
                const int tap = PAIR ? 2 * (tg + 4 * k) + (row >> 4) : tg + 4 * k;
                const int c = PAIR ? (row & 15) : row;
                if (tap < 27) dst[((size_t)tap * 32 + c) * 32 + i] = acc[k][r] + red[((tg * 7 + k) * 16 + r) * 64 + l];
            }
        }
    }
}

// deterministic second pass of the split-K: block = 64 outputs x 4 split-groups (G = 4) or 256 outputs, every split in one thread (G = 1:
// few splits — the deep levels, where four groups of two splits each spent their time on block scheduling and the LDS fold); fixed order
// (has_job: the grid carries ONE extra block — block 0 — that runs the GroupNorm-backward reduction of the layer's input — u3d_conv3d_wgrad_job)
template <int G>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                           int S, int nchunks, int nkb, int Cin, int Cout, int cstride,
                                                           int has_job, u3d_gn_bwd_job_t job) {
    __shared__ float red[4][64];
    // (the job is block 0: dispatched first, its ~5 us dependent chain runs beside the reduction instead of behind its last block)
    if (has_job && blockIdx.x == 0) {
        extern __shared__ double shb[];
        u3d_gn_bwd_finalize_body(job.gstats_lo, job.mean_rstd, job.gamma, job.N, job.C0 + job.C1, job.G, job.count, 1, 1, job.dgamma,
                                 job.dbeta, job.coef, job.gstats_hi, job.C0, job.hi_scale, job.coef_hi, shb, job.reps_lo, job.reps_hi);
        return;
    }
    const long long total = (long long)Cin * 27 * Cout;
    const int lane = G == 1 ? (int)threadIdx.x : (int)(threadIdx.x & 63), grp = G == 1 ? 0 : (int)(threadIdx.x >> 6);
    // (c, tap, k) with k fastest: coalesced partial reads
    const long long idx = (long long)(blockIdx.x - (has_job ? 1 : 0)) * (G == 1 ? 256 : 64) + lane;
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
        const int per = (S + G - 1) / G;
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
    if constexpr (G == 1) {
        if (idx < total) dw[((size_t)k * cstride + c) * 27 + tap] = sum;
    } else {
        red[grp][lane] = sum;
        __syncthreads();
        if (grp == 0 && idx < total)
            dw[((size_t)k * cstride + c) * 27 + tap] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    }
}

// the packed image(s) of one layer: the standard image, then — for <= 16 produced channels — the paired-y image and the 16-column
// image (conv3d_mfma_reg_kernel PAIRY / N16).  Splits a flat element index into (image, index inside it).
__device__ __forceinline__ int pack_image_of(long long idx, int nchunks, int ntot, long long& id) {
    const long long std_total = ((long long)nchunks * cv::NSTEP + cv::PACK_PAD) * ntot * 256;
    const long long pair_total = ((long long)nchunks * cv::NSTEP_PAIRY + cv::PACK_PAD) * 256;
    if (idx < std_total) {
        id = idx;
        return 0;
    }
    if (idx < std_total + pair_total) {
        id = idx - std_total;
        return 1;
    }
    id = idx - std_total - pair_total;
    return 2;
}

// element j of the 16-column image: f32x4 index ((ch*27 + tap)*64 + lane): contraction channel ch*16 + 4*(lane>>4) + j, produced
// channel lane & 15
__device__ __forceinline__ float pack_elem_n16(const float* __restrict__ w, int Cout, int Cin, int cstride, int mode, int nchunks,
                                               long long id) {
    const int j = (int)(id & 3);
    const int lane = (int)((id >> 2) & 63);
    const long long r = id >> 8;
    const int tap = (int)(r % cv::NSTEP_N16), ch = (int)(r / cv::NSTEP_N16);
    const int kc = ch * 16 + 4 * (lane >> 4) + j, nc = lane & 15;
    if (ch >= nchunks) return 0.f;
    if (mode == 0) return (kc < Cin && nc < Cout) ? w[((size_t)nc * cstride + kc) * 27 + tap] : 0.f;
    return (kc < Cout && nc < Cin) ? w[((size_t)kc * cstride + nc) * 27 + (26 - tap)] : 0.f;
}

// =================================================================================================
// weight packing: packed f32x4 index (((ch*54 + st)*ntot + ntg)*64 + lane), element j:
//   k-channel  c  = ch*16 + 8*(st&1) + 4*(lane>>5) + j,  tap = st>>1,  n-channel = ntg*32 + (lane&31)
// one element of the packed image(s) of one layer: `idx` counts through the normal image, then (<= 16 output channels) the
// paired-y image
// cstride: channels per output-channel row of `w` (Cin, or the parent's channel count when w points into a channel slice)
__device__ __forceinline__ float pack_elem(const float* __restrict__ w, int Cout, int Cin, int cstride, int mode, int nchunks,
                                           int ntot, long long idx) {
    // narrow outputs (<= 16 channels) get a second image for the paired-y kernel variant: 72 k-steps per chunk over the
    // 3 x 4 x 3 tap window, columns 16-31 = the same channels with the kernel shifted by one row in y — and a third one for the
    // 16-column variant
    long long id;
    const int image = pack_image_of(idx, nchunks, ntot, id);
    if (image == 2) return pack_elem_n16(w, Cout, Cin, cstride, mode, nchunks, id);
    const bool pair = image == 1;
    const int j = (int)(id & 3);
    const int lane = (int)((id >> 2) & 63);
    long long r = id >> 8;
    const int nstep = pair ? cv::NSTEP_PAIRY : cv::NSTEP;
    const int ntg = pair ? 0 : (int)(r % ntot);
    if (!pair) r /= ntot;
    const int st = (int)(r % nstep);
    const int ch = (int)(r / nstep);
    const int kc = ch * 16 + 8 * (st & 1) + 4 * (lane >> 5) + j;
    int tap = st >> 1;
    int nc = ntg * 32 + (lane & 31);
    bool tap_ok = true;
    if (pair) {
        const int tz = tap / 12, ty4 = (tap / 3) % 4, tx = tap % 3;
        const int ty = ty4 - ((lane & 31) >> 4);  // second half: kernel shifted by one row
        tap_ok = ty >= 0 && ty <= 2;
        tap = (tz * 3 + ty) * 3 + tx;
        nc = lane & 15;
    }
    float v = 0.f;
    if (ch >= nchunks || !tap_ok) {
        // the trailing zero steps / taps outside the 3^3 kernel
    } else if (mode == 0) {
        if (kc < Cin && nc < Cout) v = w[((size_t)nc * cstride + kc) * 27 + tap];
    } else {
        // dgrad: contraction over original cout (kc), output = original cin (nc), flipped taps
        if (kc < Cout && nc < Cin) v = w[((size_t)kc * cstride + nc) * 27 + (26 - tap)];
    }
    return v;
}

static long long pack_total_floats(int Cin, int Cout, int mode, int* nchunks_out, int* ntot_out) {
    const int K = mode == 0 ? Cin : Cout, Nn = mode == 0 ? Cout : Cin;
    const int nchunks = (K + 15) / 16, ntot = (Nn + 31) / 32;
    long long total = ((long long)nchunks * cv::NSTEP + cv::PACK_PAD) * ntot * 256;
    if (Nn <= 16) total += ((long long)nchunks * cv::NSTEP_PAIRY + cv::PACK_PAD) * 256 + ((long long)nchunks * cv::NSTEP_N16 + cv::PACK_PAD) * 256;
    if (nchunks_out) *nchunks_out = nchunks;
    if (ntot_out) *ntot_out = ntot;
    return total;
}

__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin,
                                    int mode, int nchunks, int ntot, long long total) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x)
        out[idx] = pack_elem(w, Cout, Cin, Cin, mode, nchunks, ntot, idx);
}

// the four elements of one packed f32x4 (j = 0..3: four consecutive contraction channels of one (step, n-tile, lane) slot) share
// every index but the channel: decode once, gather four values
__device__ __forceinline__ f32x4 pack_quad(const float* __restrict__ w, int Cout, int Cin, int cstride, int mode, int nchunks,
                                           int ntot, long long idx4) {
    long long id;
    const int image = pack_image_of(idx4 * 4, nchunks, ntot, id);
    if (image == 2) {
        f32x4 q;
#pragma unroll
        for (int j = 0; j < 4; ++j) q[j] = pack_elem_n16(w, Cout, Cin, cstride, mode, nchunks, id + j);
        return q;
    }
    const bool pair = image == 1;
    const int lane = (int)((id >> 2) & 63);
    long long r = id >> 8;
    const int nstep = pair ? cv::NSTEP_PAIRY : cv::NSTEP;
    const int ntg = pair ? 0 : (int)(r % ntot);
    if (!pair) r /= ntot;
    const int st = (int)(r % nstep);
    const int ch = (int)(r / nstep);
    const int kc0 = ch * 16 + 8 * (st & 1) + 4 * (lane >> 5);
    int tap = st >> 1;
    int nc = ntg * 32 + (lane & 31);
    bool tap_ok = true;
    if (pair) {
        const int tz = tap / 12, ty4 = (tap / 3) % 4, tx = tap % 3;
        const int ty = ty4 - ((lane & 31) >> 4);  // second half: kernel shifted by one row
        tap_ok = ty >= 0 && ty <= 2;
        tap = (tz * 3 + ty) * 3 + tx;
        nc = lane & 15;
    }
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (ch >= nchunks || !tap_ok) return v;  // the trailing zero steps / taps outside the 3^3 kernel
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int kc = kc0 + j;
        if (mode == 0) {
            if (kc < Cin && nc < Cout) v[j] = w[((size_t)nc * cstride + kc) * 27 + tap];
        } else {  // dgrad: contraction over original cout (kc), output = original cin (nc), flipped taps
            if (kc < Cout && nc < Cin) v[j] = w[((size_t)kc * cstride + nc) * 27 + (26 - tap)];
        }
    }
    return v;
}

// all layers of a model in ONE launch: descs (device memory) hold cumulative element offsets in `first` (multiples of 4: every image
// is a whole number of f32x4 slots).  One thread per f32x4 slot: 16-byte stores, index arithmetic once per four elements (round 4:
// 0.105 -> 0.0xx ms per step of the bench workload, see DESIGN_HISTORY.md 6; superseded by the LDS cell kernel below).
__global__ void pack_weights_batch_kernel(const u3d_pack_desc_t* __restrict__ descs, int n, long long total) {
    const long long total4 = total >> 2;
    for (long long g4 = (long long)blockIdx.x * blockDim.x + threadIdx.x; g4 < total4; g4 += (long long)gridDim.x * blockDim.x) {
        const long long g = g4 << 2;
        int lo = 0, hi = n - 1;  // last descriptor with first <= g
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (descs[mid].first <= g) lo = mid; else hi = mid - 1;
        }
        const u3d_pack_desc_t d = descs[lo];
        const int cstride = d.cin_stride > 0 ? d.cin_stride : d.Cin;
        const long long e = g - d.first;
        f32x4 v;
        if (d.mode == 2) {  // sub-pixel image of the channel slice [w, w + Cin) (csrc/u3d_subpix.h)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                v[j] = sp::pack_elem(d.w, d.Cout, cstride, d.Cin, (d.Cin + 15) / 16, (d.Cout + 31) / 32, e + j);
        } else if (d.mode == 3) {  // its data-gradient image (contraction over the layer's output channels)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                v[j] = spd::pack_elem(d.w, d.Cout, cstride, d.Cin, (d.Cout + 15) / 16, (d.Cin + 31) / 32, e + j);
        } else {
            const int K = d.mode == 0 ? d.Cin : d.Cout, Nn = d.mode == 0 ? d.Cout : d.Cin;
            v = pack_quad(d.w, d.Cout, d.Cin, cstride, d.mode, (K + 15) / 16, (Nn + 31) / 32, e >> 2);
        }
        *reinterpret_cast<f32x4*>(d.packed + e) = v;
    }
}

// ---- the same images through LDS (round 6) --------------------------------------------------------------------------------------------
// The thread-per-slot kernel above gathers every 4-byte element at a stride of 27 floats (and another 27 * cstride between lanes): 64
// cache lines per wave load, 0.10 ms per bench step for 33 MB of images.  Here a block owns one CELL of one image — a 16-channel
// contraction chunk x a 32-channel n-tile: 13,824 floats of the master weight that are 32 (contraction over the weight's input
// channels: modes 0 / 2) or 16 (over its output channels: modes 1 / 3) CONTIGUOUS runs of the reference layout — reads them with
// 16-byte loads into an odd-stride LDS tile and writes all fragments of the cell 16 bytes per lane: the standard image, the paired-y
// and 16-column images of a layer with <= 16 produced channels, or the pre-summed sub-pixel images (same summation order as
// sp::pack_elem / spd::pack_elem: bit-identical).  One extra block per image writes its zero prefetch tail.  Needs 16-byte aligned
// runs: cstride, Cin and Cout multiples of 4 and an aligned base (u3d_pack_weights_cells_blocks returns 0 otherwise: those images
// stay on the kernel above).
namespace pk {
constexpr int RS0 = 433;  // [32 n rows][16 k x 27 taps + 1]  (odd: the 32 lanes of a fragment read hit 32 banks)
constexpr int RS1 = 865;  // [16 k rows][32 n x 27 taps + 1]
constexpr int TILE = 32 * RS0;  // 13,856 floats >= 16 * RS1
}  // namespace pk

// A sub-pixel cell carries 128 pre-summed fragments (up to 8 LDS reads per value) against the 54 plain ones of a standard cell: it is dealt
// to PK_SUBQ blocks of 32 fragments each (every one re-reads the 54 KB cell from L2), so that the launch does not wait for its heaviest cells
constexpr int PK_SUBQ = 4;
__host__ __device__ inline long long pack_cells_of(int Cin, int Cout, int mode) {  // blocks of one image (without the tail block)
    const int K = (mode == 0 || mode == 2) ? Cin : Cout, Nn = (mode == 0 || mode == 2) ? Cout : Cin;
    return (long long)((K + 15) / 16) * ((Nn + 31) / 32) * (mode >= 2 ? PK_SUBQ : 1);
}

__global__ __launch_bounds__(256) void pack_weights_cells_kernel(const u3d_pack_desc_t* __restrict__ descs, int n) {
    __shared__ float tile[pk::TILE];
    const int t = threadIdx.x;
    int di = 0;
    {
        int lo = 0, hi = n - 1;  // last descriptor with first <= blockIdx.x
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (descs[mid].first <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
        }
        di = lo;
    }
    const u3d_pack_desc_t d = descs[di];
    const int mode = d.mode, Cin = d.Cin, Cout = d.Cout;
    const int cstride = d.cin_stride > 0 ? d.cin_stride : d.Cin;
    const bool geo0 = mode == 0 || mode == 2;       // rows = produced channels of the image (weight's OUTPUT channels), k = its input channels
    const int K = geo0 ? Cin : Cout, Nn = geo0 ? Cout : Cin;
    const int nchunks = (K + 15) / 16, ntot = (Nn + 31) / 32;
    const int b = (int)((long long)blockIdx.x - d.first);
    f32x4* out4 = reinterpret_cast<f32x4*>(d.packed);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const bool narrow = mode <= 1 && Nn <= 16;  // the paired-y and 16-column images follow the standard one
    const long long std4 = ((long long)nchunks * cv::NSTEP + cv::PACK_PAD) * ntot * 64;  // f32x4 slots of the standard image
    const long long pair4 = ((long long)nchunks * cv::NSTEP_PAIRY + cv::PACK_PAD) * 64;
    const int subq = mode >= 2 ? PK_SUBQ : 1;
    if (b >= nchunks * ntot * subq) {
        // tail block: the zero k-steps / fragments behind the last chunk of every image of this descriptor
        if (mode <= 1) {
            for (int i = t; i < cv::PACK_PAD * ntot * 64; i += 256) out4[(long long)nchunks * cv::NSTEP * ntot * 64 + i] = zero4;
            if (narrow) {
                for (int i = t; i < cv::PACK_PAD * 64; i += 256) {
                    out4[std4 + (long long)nchunks * cv::NSTEP_PAIRY * 64 + i] = zero4;
                    out4[std4 + pair4 + (long long)nchunks * cv::NSTEP_N16 * 64 + i] = zero4;
                }
            }
        } else {
            const int pad = mode == 2 ? sp::PACK_PAD : spd::PACK_PAD;
            const int nfrag = mode == 2 ? sp::NFRAG : spd::NFRAG;
            for (int i = t; i < pad * ntot * 64; i += 256) out4[(long long)nchunks * nfrag * ntot * 64 + i] = zero4;
        }
        return;
    }
    const int cell = b / subq, fq = b - cell * subq;  // (sub-pixel images: this block's quarter of the cell's fragments)
    const int ch = cell / ntot, ntg = cell - ch * ntot;
    // ---- the cell's runs -> LDS (zero where the cell overhangs the channel counts)
    const int kval = min(16, K - ch * 16), nval = min(32, Nn - ntg * 32);  // valid contraction / produced channels of this cell
    const int rows = geo0 ? nval : kval, run = (geo0 ? kval : nval) * 27;   // runs and their length in floats (a multiple of 4)
    const int RS = geo0 ? pk::RS0 : pk::RS1;
    if (kval < 16 || nval < 32) {
        for (int i = t; i < pk::TILE; i += 256) tile[i] = 0.f;
        __syncthreads();
    }
    {
        const float* base = geo0 ? d.w + ((size_t)(ntg * 32) * cstride + ch * 16) * 27 : d.w + ((size_t)(ch * 16) * cstride + ntg * 32) * 27;
        const int run4 = run >> 2, total4 = rows * run4;
        for (int i0 = t; i0 < total4; i0 += 256 * 4) {  // four loads in flight per thread
            f32x4 v[4];
            int rr[4], oo[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + 256 * u;
                rr[u] = i / run4;
                oo[u] = i - rr[u] * run4;
                v[u] = i < total4 ? *reinterpret_cast<const f32x4*>(base + (size_t)rr[u] * cstride * 27 + 4 * oo[u]) : zero4;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (i0 + 256 * u < total4) {
                    float* dst = tile + rr[u] * RS + 4 * oo[u];
                    dst[0] = v[u][0], dst[1] = v[u][1], dst[2] = v[u][2], dst[3] = v[u][3];
                }
            }
        }
    }
    __syncthreads();
    // 27 taps of (produced channel nl, contraction channel kl) of this cell; tap order of the image: forward t, data gradient 26 - t
    auto row_of = [&](int nl, int kl) { return geo0 ? tile + nl * pk::RS0 + kl * 27 : tile + kl * pk::RS1 + nl * 27; };
    if (mode <= 1) {
        const bool flip = mode == 1;
        for (int i = t; i < cv::NSTEP * 64; i += 256) {
            const int st = i >> 6, lane = i & 63;
            const int tap = st >> 1, kl = 8 * (st & 1) + 4 * (lane >> 5), nl = lane & 31;
            const float* r0 = row_of(nl, kl) + (flip ? 26 - tap : tap);
            const int ks = geo0 ? 27 : pk::RS1;  // stride between consecutive contraction channels
            out4[(((long long)ch * cv::NSTEP + st) * ntot + ntg) * 64 + lane] = f32x4{r0[0], r0[ks], r0[2 * ks], r0[3 * ks]};
        }
        if (narrow) {
            const int ks = geo0 ? 27 : pk::RS1;
            for (int i = t; i < cv::NSTEP_PAIRY * 64; i += 256) {  // paired-y image: 3 x 4 x 3 tap window, columns 16-31 shifted by one row
                const int st = i >> 6, lane = i & 63;
                const int tp = st >> 1, tz = tp / 12, ty4 = (tp / 3) % 4, tx = tp % 3;
                const int ty = ty4 - ((lane & 31) >> 4);
                f32x4 v = zero4;
                if (ty >= 0 && ty <= 2) {
                    const int tap = (tz * 3 + ty) * 3 + tx, kl = 8 * (st & 1) + 4 * (lane >> 5), nl = lane & 15;
                    const float* r0 = row_of(nl, kl) + (flip ? 26 - tap : tap);
                    v = f32x4{r0[0], r0[ks], r0[2 * ks], r0[3 * ks]};
                }
                out4[std4 + ((long long)ch * cv::NSTEP_PAIRY + st) * 64 + lane] = v;
            }
            for (int i = t; i < cv::NSTEP_N16 * 64; i += 256) {  // 16-column image: one k-step per tap over the chunk's 16 channels
                const int tap = i >> 6, lane = i & 63;
                const int kl = 4 * (lane >> 4), nl = lane & 15;
                const float* r0 = row_of(nl, kl) + (flip ? 26 - tap : tap);
                out4[std4 + pair4 + ((long long)ch * cv::NSTEP_N16 + tap) * 64 + lane] = f32x4{r0[0], r0[ks], r0[2 * ks], r0[3 * ks]};
            }
        }
    } else if (mode == 2) {
        static_assert(sp::NFRAG % PK_SUBQ == 0 && spd::NFRAG % PK_SUBQ == 0, "fragments per sub-block");
        for (int i = t; i < sp::NFRAG / PK_SUBQ * 64; i += 256) {
            const int f = fq * (sp::NFRAG / PK_SUBQ) + (i >> 6), lane = i & 63;
            const int st = sp::frag_tab().st[f];
            const int kl = 8 * (st & 1) + 4 * (lane >> 5), nl = lane & 31;
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = sp::frag_value(row_of(nl, kl + j), f);  // (a zero row where the cell overhangs: sums of zeros)
            out4[(((long long)ch * sp::NFRAG + f) * ntot + ntg) * 64 + lane] = v;
        }
    } else {
        for (int i = t; i < spd::NFRAG / PK_SUBQ * 64; i += 256) {
            const int f = fq * (spd::NFRAG / PK_SUBQ) + (i >> 6), lane = i & 63;
            const int kl = 8 * (f & 1) + 4 * (lane >> 5), nl = lane & 31;
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = spd::frag_value(row_of(nl, kl + j), f);
            out4[(((long long)ch * spd::NFRAG + f) * ntot + ntg) * 64 + lane] = v;
        }
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
    // (C0 == 0: only the upsampled half — the box launches of round 5; p0 must still be a readable pointer: dead lanes load from it)
    U3D_REQUIRE(s != nullptr && s->p0 != nullptr && (s->C0 > 0 || s->C1 > 0) && s->C0 >= 0, "%s: null source", what);
    U3D_REQUIRE(s->C1 >= 0, "%s: negative C1", what);
    if (s->C1 > 0)
        U3D_REQUIRE(s->p1 && s->zmap && s->ymap && s->xmap && s->D1 > 0 && s->H1 > 0 && s->W1 > 0,
                    "%s: low-res source needs p1, index maps and dims", what);
    return 0;
}

static long long* g_u3d_prof_buf = nullptr;
static size_t g_u3d_prof_records = 0;

extern "C" int u3d_set_profile_buffer(void* device_buffer, size_t bytes) {
    g_u3d_prof_buf = static_cast<long long*>(device_buffer);
    g_u3d_prof_records = device_buffer ? bytes / (24 * sizeof(long long)) : 0;
    return 0;
}

extern "C" int u3d_set_tuning(int key, int value) {
    if (key < 0 || key >= 24) return u3d_set_err(U3D_EINVAL, "u3d_set_tuning: key out of range");
    g_u3d_tune[key] = value;
    return 0;
}

extern "C" size_t u3d_packed_weight_floats(int Cin, int Cout, int mode) {
    return (size_t)pack_total_floats(Cin, Cout, mode, nullptr, nullptr);  // incl. zero steps (prefetch overrun) and pair image
}

extern "C" int u3d_pack_weights(int device, u3d_stream_t stream, const float* w, int Cout, int Cin, int mode,
                                float* packed) {
    U3D_ENTER(device);
    U3D_REQUIRE(w && packed && Cout > 0 && Cin > 0 && (mode == 0 || mode == 1), "u3d_pack_weights: bad argument");
    int nchunks = 0, ntot = 0;
    const long long total = pack_total_floats(Cin, Cout, mode, &nchunks, &ntot);
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, packed, Cout, Cin,
                       mode, nchunks, ntot, total);
    U3D_LAUNCH_CHECK();
    return 0;
}

extern "C" int u3d_pack_weights_batch(int device, u3d_stream_t stream, const u3d_pack_desc_t* descs_device, int n,
                                      int64_t total_floats) {
    U3D_ENTER(device);
    U3D_REQUIRE(descs_device && n > 0 && total_floats > 0, "u3d_pack_weights_batch: bad argument");
    U3D_REQUIRE(total_floats % 4 == 0, "u3d_pack_weights_batch: image sizes are multiples of 4 floats");
    long long blocks = (total_floats / 4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(pack_weights_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, descs_device, n,
                       (long long)total_floats);
    U3D_LAUNCH_CHECK();
    return 0;
}

extern "C" long long u3d_pack_weights_cells_blocks(const float* w, int Cin, int Cout, int mode, int cin_stride) {
    // blocks one image takes in u3d_pack_weights_batch_cells (its cells + the tail block); 0 = this image cannot go through the cell
    // kernel (runs not 16-byte aligned) and stays on u3d_pack_weights_batch
    const int cs = cin_stride > 0 ? cin_stride : Cin;
    if (Cin <= 0 || Cout <= 0 || mode < 0 || mode > 3 || Cin % 4 != 0 || Cout % 4 != 0 || cs % 4 != 0 || ((uintptr_t)w & 15) != 0) return 0;
    return pack_cells_of(Cin, Cout, mode) + 1;
}

extern "C" int u3d_pack_weights_batch_cells(int device, u3d_stream_t stream, const u3d_pack_desc_t* descs_device, int n,
                                            long long total_blocks) {
    U3D_ENTER(device);
    U3D_REQUIRE(descs_device && n > 0 && total_blocks > 0 && total_blocks < 0x7fffffffLL, "u3d_pack_weights_batch_cells: bad argument");
    hipLaunchKernelGGL(pack_weights_cells_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, descs_device, n);
    U3D_LAUNCH_CHECK();
    return 0;
}

static int device_cu_count(int device, int* out) {
    static int cached[64] = {0};
    if (device >= 0 && device < 64 && cached[device] > 0) {
        *out = cached[device];
        return 0;
    }
    int n = 0;
    U3D_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device));
    if (n <= 0) n = 256;
    if (device >= 0 && device < 64) cached[device] = n;
    *out = n;
    return 0;
}

template <int NT>
static int conv_set_lds_nt() {
    const int bytes = cv::LDS_FLOATS * sizeof(float);
    U3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_mfma_reg_kernel<NT, false, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    U3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_mfma_reg_kernel<NT, true, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    U3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_mfma_reg_kernel<NT, false, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    U3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_mfma_reg_kernel<NT, true, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    U3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_mfma_reg_kernel<NT, false, false, false, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    U3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_mfma_reg_kernel<NT, true, false, false, true, false, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    U3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_mfma_reg_kernel<NT, false, false, false, false, false, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    U3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_mfma_reg_kernel<NT, false, false, false, true, false, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    if (NT == 1) {
        U3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_mfma_reg_kernel<1, true, false, false, true, true, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        U3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_mfma_reg_kernel<1, false, false, false, false, true, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        U3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_mfma_reg_kernel<1, false, false, false, true, true, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        U3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_mfma_reg_kernel<1, false, false, true, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        U3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_mfma_reg_kernel<1, false, false, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        U3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_mfma_reg_kernel<1, true, false, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        U3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_mfma_reg_kernel<1, true, false, false, true, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        U3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_mfma_reg_kernel<1, false, false, false, false, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        U3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_mfma_reg_kernel<1, false, false, false, true, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    }
    U3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_mfma_kernel<NT, true, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    U3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_mfma_kernel<NT, false, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    U3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_mfma_kernel<NT, true, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    if (NT == 1) {
        U3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_mfma_kernel<1, true, true, 1>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        U3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_mfma_kernel<1, true, true, 2>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        U3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_mfma_kernel<1, true, true, 3>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        U3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_mfma_kernel<1, true, true, 7>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    }
    return 0;
}

static int conv_set_lds_once(int device) {
    static bool done[64] = {false};
    if (device >= 0 && device < 64 && done[device]) return 0;
    if (int e = conv_set_lds_nt<1>()) return e;
    if (int e = conv_set_lds_nt<2>()) return e;
    if (int e = conv_set_lds_nt<3>()) return e;
    if (device >= 0 && device < 64) done[device] = true;
    return 0;
}

static int conv3d_impl(int device, u3d_stream_t stream, const u3d_src_t* src, const float* packed_w, float* out, int N,
                       int D, int H, int W, int Cout, int relu, double* out_stats, const u3d_src_t* gx, double* gstats,
                       const float* residual, float* ws = nullptr, long long ws_floats = 0, const int* out_box = nullptr,
                       const int* in_mask = nullptr, int stat_reps = 1);

extern "C" int u3d_conv3d(int device, u3d_stream_t stream, const u3d_src_t* src, const float* packed_w, float* out,
                          int N, int D, int H, int W, int Cout, int relu, double* out_stats, const u3d_src_t* gx,
                          double* gstats) {
    return conv3d_impl(device, stream, src, packed_w, out, N, D, H, W, Cout, relu, out_stats, gx, gstats, nullptr);
}

// split-K (see splitk_reduce_kernel): used when one block per (tile, 32-channel block) leaves most CUs idle
// Plain convolution (no ReLU / statistics / residual) restricted to a BOX of output voxels, optionally reading only an input SLAB
// (round 5: the near-boundary slab of a decoder level that upsamples n -> 2n + 1, see u3d_subpixel_conv_fwd_win).
extern "C" int u3d_conv3d_box(int device, u3d_stream_t stream, const u3d_src_t* src, const float* packed_w, float* out, int N, int D,
                              int H, int W, int Cout, const int* out_box, const int* in_mask) {
    return conv3d_impl(device, stream, src, packed_w, out, N, D, H, W, Cout, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0, out_box,
                       in_mask);
}

constexpr int SPLITK_MAX = 16;
// ONE statement of the split-K decision, used by the launcher (conv3d_impl) and by the host-only query u3d_conv3d_variant (ADVICE r05:
// the query restated it with 256 CUs hard-coded): ksplit blocks per (tile, 32-channel block) so that ~2 blocks per CU exist, over runs
// of cps chunks; 1 = no split.
static int g_u3d_ncu_seen = 256;  // CU count of the last device a launcher ran on (the query has no device argument; MI355X: 256)
static int splitk_count(int ncu, long long ntiles, int ntot, int nchunks, int* cps_out) {
    long long ks = (2ll * ncu) / (ntiles * ntot);
    if (ks > nchunks) ks = nchunks;
    if (ks > SPLITK_MAX) ks = SPLITK_MAX;
    int cps = nchunks, ksplit = 1;
    if (ks >= 2) {
        cps = (nchunks + (int)ks - 1) / (int)ks;
        ksplit = (nchunks + cps - 1) / cps;
    }
    if (cps_out) *cps_out = cps;
    return ksplit;
}

static bool splitk_shape(int N, int D, int H, int W, int Cin, int Cout) {
    const long long items = (long long)N * cdiv(D, cv::TZ) * cdiv(H, cv::TY) * cdiv(W, cv::TX) * cdiv(Cout, 32);
    return Cin > 16 && Cout % 4 == 0 && Cout <= 1024 && items < 256;
}

extern "C" long long u3d_conv3d_workspace_floats(int N, int D, int H, int W, int Cin, int Cout) {
    if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || !splitk_shape(N, D, H, W, Cin, Cout)) return 0;
    const int ks = cdiv(Cin, 16) < SPLITK_MAX ? cdiv(Cin, 16) : SPLITK_MAX;
    return (long long)ks * N * D * H * W * Cout;
}

extern "C" int u3d_conv3d_ex(int device, u3d_stream_t stream, const u3d_src_t* src, const float* packed_w, float* out,
                             int N, int D, int H, int W, int Cout, int relu, double* out_stats, const u3d_src_t* gx,
                             double* gstats, const float* residual, float* workspace, long long workspace_floats) {
    U3D_REQUIRE(!(residual && gx), "u3d_conv3d_ex: residual and gx are mutually exclusive");
    return conv3d_impl(device, stream, src, packed_w, out, N, D, H, W, Cout, relu, out_stats, gx, gstats, residual,
                       workspace, workspace_floats);
}

// ... with a statistics table of stat_reps replica rows [stat_reps][N][Cout][2] (zeroed by the caller): the persistent kernels' blocks
// spread their per-sample flush over the rows, every other variant adds to row 0 — the true sums are the sums over the rows
// (u3d_gn_finalize_reps, u3d_gn_bwd_job_t::reps_lo).
extern "C" int u3d_conv3d_ex_reps(int device, u3d_stream_t stream, const u3d_src_t* src, const float* packed_w, float* out,
                                  int N, int D, int H, int W, int Cout, int relu, double* out_stats, const u3d_src_t* gx,
                                  double* gstats, const float* residual, float* workspace, long long workspace_floats, int stat_reps) {
    U3D_REQUIRE(!(residual && gx), "u3d_conv3d_ex_reps: residual and gx are mutually exclusive");
    return conv3d_impl(device, stream, src, packed_w, out, N, D, H, W, Cout, relu, out_stats, gx, gstats, residual,
                       workspace, workspace_floats, nullptr, nullptr, stat_reps);
}

extern "C" int u3d_conv3d_residual(int device, u3d_stream_t stream, const u3d_src_t* src, const float* packed_w,
                                   float* out, int N, int D, int H, int W, int Cout, int relu, double* out_stats,
                                   const float* residual) {
    if (residual == nullptr) return u3d_set_err(U3D_EINVAL, "u3d_conv3d_residual: residual is NULL");
    return conv3d_impl(device, stream, src, packed_w, out, N, D, H, W, Cout, relu, out_stats, nullptr, nullptr, residual);
}

static int conv3d_impl(int device, u3d_stream_t stream, const u3d_src_t* src, const float* packed_w, float* out, int N,
                       int D, int H, int W, int Cout, int relu, double* out_stats, const u3d_src_t* gx, double* gstats,
                       const float* residual, float* ws, long long ws_floats, const int* out_box, const int* in_mask, int stat_reps) {
    U3D_ENTER(device);
    if (int e = check_src(src, "u3d_conv3d")) return e;
    U3D_REQUIRE(stat_reps >= 1 && stat_reps <= 64, "u3d_conv3d: stat_reps must be 1 .. 64");
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
    p.res = residual;
    p.N = N, p.D = D, p.H = H, p.W = W, p.Cout = Cout;
    const int Cin = src->C0 + src->C1;
    p.nchunks = cdiv(Cin, 16);
    p.ntot = cdiv(Cout, 32);
    p.tz = cdiv(D, cv::TZ), p.ty = cdiv(H, cv::TY), p.tx = cdiv(W, cv::TX);
    p.oz0 = p.oy0 = p.ox0 = 0, p.oz1 = D, p.oy1 = H, p.ox1 = W;
    p.mz = p.my = p.mx = 0;
    const bool boxed = out_box != nullptr || in_mask != nullptr;  // u3d_conv3d_box: the generic kernel on a sub-box of the volume
    if (out_box) {
        U3D_REQUIRE(out_box[0] >= 0 && out_box[1] >= 0 && out_box[2] >= 0 && out_box[3] <= D && out_box[4] <= H && out_box[5] <= W &&
                        out_box[0] < out_box[3] && out_box[1] < out_box[4] && out_box[2] < out_box[5],
                    "u3d_conv3d_box: output box outside the volume or empty");
        p.oz0 = out_box[0], p.oy0 = out_box[1], p.ox0 = out_box[2], p.oz1 = out_box[3], p.oy1 = out_box[4], p.ox1 = out_box[5];
        p.tz = cdiv(p.oz1 - p.oz0, cv::TZ), p.ty = cdiv(p.oy1 - p.oy0, cv::TY), p.tx = cdiv(p.ox1 - p.ox0, cv::TX);
    }
    if (in_mask) p.mz = in_mask[0], p.my = in_mask[1], p.mx = in_mask[2];
    p.relu = relu;
    p.vec = src_vec_ok(src) ? 1 : 0;
    // wide (16-byte) epilogue: whole channel quads, aligned output and (for dgrad) an x source readable in quads
    p.ovec = (Cout % 4 == 0 && ((uintptr_t)out & 15) == 0 && (!gx || src_vec_ok(gx)) &&
              (!residual || ((uintptr_t)residual & 15) == 0)) ? 1 : 0;
    const long long ntiles = (long long)N * p.tz * p.ty * p.tx;
    // N-tiles per block: BN = 32*NT output channels share one staged A tile.  Larger NT = fewer re-stagings of the
    // same halo tile and fewer LDS reads per MFMA, at the price of registers (NT=1: 3 blocks/CU, NT>=2: 2 blocks/CU);
    // take the largest NT in {3,2} that divides the N-tile count and still leaves >= 2 blocks per CU.
    int nt = 1;
    if (p.ntot % 3 == 0 && ntiles * (p.ntot / 3) >= 512) nt = 3;
    else if (p.ntot % 2 == 0 && ntiles * (p.ntot / 2) >= 512) nt = 2;
    if (g_u3d_tune[0] >= 1 && g_u3d_tune[0] <= 3 && p.ntot % g_u3d_tune[0] == 0) nt = g_u3d_tune[0];
    p.ncb = p.ntot / nt;
    const long long nblk = ntiles * p.ncb;
    U3D_REQUIRE(nblk < (1ll << 31), "u3d_conv3d: grid too large");
    const size_t shmem = cv::LDS_FLOATS * sizeof(float);
    if (int e = conv_set_lds_once(device)) return e;
    hipStream_t st = (hipStream_t)stream;
    p.ksplit = 1, p.cps = p.nchunks, p.part_stride = 0;
    // ---- split-K on small volumes: ksplit blocks per (tile, 32-channel block), partial sums in the caller's workspace,
    //      summed in a fixed order by splitk_reduce_kernel together with the epilogue (key 7 = 2 turns it off)
    if (!boxed && ws && p.vec && p.ovec && ((uintptr_t)ws & 15) == 0 && splitk_shape(N, D, H, W, Cin, Cout) && g_u3d_tune[7] != 2) {
        int ncu = 0;
        if (int e = device_cu_count(device, &ncu)) return e;
        g_u3d_ncu_seen = ncu;
        const long long items = ntiles * p.ntot, out_elems = (long long)N * D * H * W * Cout;
        p.ksplit = splitk_count(ncu, ntiles, p.ntot, p.nchunks, &p.cps);
        if (p.ksplit >= 2 && (long long)p.ksplit * out_elems <= ws_floats) {
            p.ncb = p.ntot;  // NT = 1: most blocks
            p.part_stride = out_elems;
            p.out = ws;
            p.out_stats = nullptr, p.gstats = nullptr, p.res = nullptr, p.relu = 0, p.has_gx = 0, p.dbg = nullptr;
            const dim3 kgrid((unsigned)(items * p.ksplit)), kblock(256);
            hipLaunchKernelGGL((conv3d_mfma_kernel<1, true>), kgrid, kblock, shmem, st, p);
            U3D_LAUNCH_CHECK();
            SplitKParams r;
            r.part = ws, r.stride = out_elems, r.out = out, r.res = residual;
            r.stats = out_stats ? out_stats : gstats;
            r.gx = gx ? *gx : *src;
            r.ksplit = p.ksplit, r.relu = relu, r.want_g = gstats != nullptr;
            r.V = D * H * W, r.D = D, r.H = H, r.W = W, r.C = Cout;
            const int Q = Cout / 4, per = 256 / Q;
            int gxb = cdiv(r.V, per * 4);  // >= 4 voxels per thread
            const int cap = 256 / N > 1 ? 256 / N : 1;
            if (gxb > cap) gxb = cap;
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)gxb, (unsigned)N), dim3((unsigned)(Q * per)), 0, st, r);
            U3D_LAUNCH_CHECK();
            return 0;
        }
        p.ksplit = 1, p.cps = p.nchunks;
    }
    // ---- fast variant: persistent blocks, constant-offset staging (every tile fully inside, no table look-ups)
    auto plain_or_x2 = [&](const u3d_src_t& s_) {
        return s_.C1 == 0 || (D == 2 * s_.D1 && H == 2 * s_.H1 && W == 2 * s_.W1);
    };
    // ragged volumes (last tiles overhang) run the RAG instantiations of the same kernel (key 3 = 2: the round-4 gate, generic kernel)
    const bool ragged = D % cv::TZ != 0 || H % cv::TY != 0 || W % cv::TX != 0;
    const bool reg = p.vec && p.ovec && plain_or_x2(p.src) && (!gx || plain_or_x2(p.gx)) && g_u3d_tune[3] == 0 ||
                     (g_u3d_tune[3] == 2 && !ragged && p.vec && p.ovec && plain_or_x2(p.src) && (!gx || plain_or_x2(p.gx)));
    if (reg && !boxed) {
        p.total = (int)nblk;
        p.gx_x2 = (gx && p.gx.C1 > 0) ? 1 : 0;
        // experiment knob, default off: a sweep of 8..64 k cycles changed no layer by more than noise (profiles/r01q) — a
        // wave that is alone on its SIMD does not run at twice the shared rate, so interleaving the epilogues buys nothing
        p.stagger = g_u3d_tune[5];
        p.stat_reps = stat_reps;
        p.zfast = g_u3d_tune[14] == 1 ? 0 : 1;
        int ncu = 0;
        if (int e = device_cu_count(device, &ncu)) return e;
        long long slots = (g_u3d_tune[6] == 1 ? 1ll : 2ll) * ncu;  // two blocks per CU (LDS); key 6 = 1: one (experiment)
        if (g_u3d_tune[12] > 0 && g_u3d_tune[12] < slots / 2) slots -= g_u3d_tune[12];  // slots left to other streams (key 12)
        if (slots >= nblk)
            slots = nblk;
        else if (slots > p.ncb)
            slots -= slots % p.ncb;  // a block stays on one channel block: one statistics flush per sample
        const dim3 rgrid((unsigned)slots), rblock(256);
        const bool virt = p.src.C1 > 0;
        // a source without a GroupNorm affine (every dgrad launch: dz is plain) runs the variant compiled without the
        // per-element FMA of the halo stores (key 7 = 1 turns it off for A/B runs)
        const bool noaff = p.src.affine == nullptr && g_u3d_tune[7] == 0;
        p.dbg = (!ragged && g_u3d_prof_buf && (size_t)slots * 4 <= g_u3d_prof_records) ? g_u3d_prof_buf : nullptr;
        if (ragged) {
            if (Cout <= 16 && nt == 1 && p.ntot == 1) {  // 16-column variant on the third packed image, as below
                p.wp = packed_w + ((size_t)p.nchunks * cv::NSTEP + cv::PACK_PAD) * 256 + ((size_t)p.nchunks * cv::NSTEP_PAIRY + cv::PACK_PAD) * 256;
                if (virt)
                    hipLaunchKernelGGL((conv3d_mfma_reg_kernel<1, true, false, false, true, true, true>), rgrid, rblock, shmem, st, p);
                else if (noaff)
                    hipLaunchKernelGGL((conv3d_mfma_reg_kernel<1, false, false, false, false, true, true>), rgrid, rblock, shmem, st, p);
                else
                    hipLaunchKernelGGL((conv3d_mfma_reg_kernel<1, false, false, false, true, true, true>), rgrid, rblock, shmem, st, p);
                U3D_LAUNCH_CHECK();
                return 0;
            }
#define U3D_RAG_LAUNCH(NT_)                                                                                                        \
    do {                                                                                                                           \
        if (virt)                                                                                                                  \
            hipLaunchKernelGGL((conv3d_mfma_reg_kernel<NT_, true, false, false, true, false, true>), rgrid, rblock, shmem, st, p);  \
        else if (noaff)                                                                                                            \
            hipLaunchKernelGGL((conv3d_mfma_reg_kernel<NT_, false, false, false, false, false, true>), rgrid, rblock, shmem, st, p); \
        else                                                                                                                       \
            hipLaunchKernelGGL((conv3d_mfma_reg_kernel<NT_, false, false, false, true, false, true>), rgrid, rblock, shmem, st, p); \
    } while (0)
            if (nt == 3)
                U3D_RAG_LAUNCH(3);
            else if (nt == 2)
                U3D_RAG_LAUNCH(2);
            else
                U3D_RAG_LAUNCH(1);
#undef U3D_RAG_LAUNCH
            U3D_LAUNCH_CHECK();
            return 0;
        }
        if (Cout <= 16 && nt == 1 && p.ntot == 1 && !p.dbg && g_u3d_tune[4] != 1) {
            // <= 16 output channels (the 32 -> 16 data gradient at full resolution): the 16-column variant on the THIRD packed image
            // (u3d_pack_weights appends the paired-y and the 16-column image for <= 16 produced channels); key 4 = 2: the round-1
            // paired-y variant on the second image (A/B), key 4 = 1: the padded 32-column kernel
            const size_t std_floats = ((size_t)p.nchunks * cv::NSTEP + cv::PACK_PAD) * 256;
            if (g_u3d_tune[4] == 2) {
                p.wp = packed_w + std_floats;
                if (virt)
                    hipLaunchKernelGGL((conv3d_mfma_reg_kernel<1, true, false, true>), rgrid, rblock, shmem, st, p);
                else if (noaff)
                    hipLaunchKernelGGL((conv3d_mfma_reg_kernel<1, false, false, true, false>), rgrid, rblock, shmem, st, p);
                else
                    hipLaunchKernelGGL((conv3d_mfma_reg_kernel<1, false, false, true>), rgrid, rblock, shmem, st, p);
            } else {
                p.wp = packed_w + std_floats + ((size_t)p.nchunks * cv::NSTEP_PAIRY + cv::PACK_PAD) * 256;
                if (virt)
                    hipLaunchKernelGGL((conv3d_mfma_reg_kernel<1, true, false, false, true, true>), rgrid, rblock, shmem, st, p);
                else if (noaff)
                    hipLaunchKernelGGL((conv3d_mfma_reg_kernel<1, false, false, false, false, true>), rgrid, rblock, shmem, st, p);
                else
                    hipLaunchKernelGGL((conv3d_mfma_reg_kernel<1, false, false, false, true, true>), rgrid, rblock, shmem, st, p);
            }
            U3D_LAUNCH_CHECK();
            return 0;
        }
#define U3D_REG_LAUNCH(NT_)                                                                                \
    do {                                                                                                   \
        if (p.dbg && virt)                                                                                 \
            hipLaunchKernelGGL((conv3d_mfma_reg_kernel<NT_, true, true>), rgrid, rblock, shmem, st, p);    \
        else if (p.dbg)                                                                                    \
            hipLaunchKernelGGL((conv3d_mfma_reg_kernel<NT_, false, true>), rgrid, rblock, shmem, st, p);   \
        else if (virt)                                                                                     \
            hipLaunchKernelGGL((conv3d_mfma_reg_kernel<NT_, true, false>), rgrid, rblock, shmem, st, p);   \
        else if (noaff)                                                                                    \
            hipLaunchKernelGGL((conv3d_mfma_reg_kernel<NT_, false, false, false, false>), rgrid, rblock,   \
                               shmem, st, p);                                                              \
        else                                                                                               \
            hipLaunchKernelGGL((conv3d_mfma_reg_kernel<NT_, false, false>), rgrid, rblock, shmem, st, p);  \
    } while (0)
        if (nt == 3)
            U3D_REG_LAUNCH(3);
        else if (nt == 2)
            U3D_REG_LAUNCH(2);
        else
            U3D_REG_LAUNCH(1);
#undef U3D_REG_LAUNCH
        U3D_LAUNCH_CHECK();
        return 0;
    }
    const bool vec = p.vec != 0;
    const dim3 grid((unsigned)nblk), block(256);
    p.dbg = (vec && g_u3d_prof_buf && (size_t)nblk * 4 <= g_u3d_prof_records) ? g_u3d_prof_buf : nullptr;
#define U3D_CONV_LAUNCH(NT_)                                                                     \
    do {                                                                                         \
        if (p.dbg && NT_ == 1 && g_u3d_tune[2] == 1)                                             \
            hipLaunchKernelGGL((conv3d_mfma_kernel<1, true, true, 1>), grid, block, shmem, st, p); \
        else if (p.dbg && NT_ == 1 && g_u3d_tune[2] == 2)                                        \
            hipLaunchKernelGGL((conv3d_mfma_kernel<1, true, true, 2>), grid, block, shmem, st, p); \
        else if (p.dbg && NT_ == 1 && g_u3d_tune[2] == 3)                                        \
            hipLaunchKernelGGL((conv3d_mfma_kernel<1, true, true, 3>), grid, block, shmem, st, p); \
        else if (p.dbg && NT_ == 1 && g_u3d_tune[2] == 7)                                        \
            hipLaunchKernelGGL((conv3d_mfma_kernel<1, true, true, 7>), grid, block, shmem, st, p); \
        else if (p.dbg)                                                                          \
            hipLaunchKernelGGL((conv3d_mfma_kernel<NT_, true, true>), grid, block, shmem, st, p); \
        else if (vec)                                                                            \
            hipLaunchKernelGGL((conv3d_mfma_kernel<NT_, true>), grid, block, shmem, st, p);      \
        else                                                                                     \
            hipLaunchKernelGGL((conv3d_mfma_kernel<NT_, false>), grid, block, shmem, st, p);     \
    } while (0)
    if (nt == 3)
        U3D_CONV_LAUNCH(3);
    else if (nt == 2)
        U3D_CONV_LAUNCH(2);
    else
        U3D_CONV_LAUNCH(1);
#undef U3D_CONV_LAUNCH
    U3D_LAUNCH_CHECK();
    return 0;
}

static void wgrad_plan(int N, int D, int H, int W, int Cin, int Cout, WgradParams& p) {  // (D, H, W: of the dz box)
    p.nchunks = cdiv(Cin, 32);
    p.nkb = cdiv(Cout, 32);
    p.tz = cdiv(D, wg::TZ), p.ty = cdiv(H, wg::TY), p.tx = cdiv(W, wg::TX);
    p.ntiles = N * p.tz * p.ty * p.tx;
    // One 8-wave block per CU (135 KB of LDS): pick the split count S whose grid S*pairs fills whole rounds of the
    // 256 CUs best.  cost = rounds * (tiles per split + ~2 tiles of prologue / partial-sum write per block).
    const int pairs = p.nchunks * p.nkb;
    const int ncu = 256;  // (MI355X)
    long long best_cost = -1;
    int best_S = 1;
    for (int rounds = 1; rounds <= 8; ++rounds) {
        int S = (rounds * ncu) / pairs;
        if (S < 1) S = 1;
        if (S > p.ntiles) S = p.ntiles;
        const int tps = cdiv(p.ntiles, S);
        S = cdiv(p.ntiles, tps);
        const long long cost = (long long)cdiv(S * pairs, ncu) * (tps + 2);
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best_S = S;
        }
    }
    if (g_u3d_tune[1] > 0) best_S = g_u3d_tune[1] > p.ntiles ? p.ntiles : g_u3d_tune[1];
    p.tps = cdiv(p.ntiles, best_S);
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
    U3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_wgrad_kernel<true, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (wg::LDS_FLOATS + wg::MAX_MAP_INTS) * sizeof(float)));
    U3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_wgrad_kernel<true, true, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (wg::LDS_FLOATS + wg::MAX_MAP_INTS) * sizeof(float)));
    U3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_wgrad_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (wg::LDS_FLOATS + wg::MAX_MAP_INTS) * sizeof(float)));
    U3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_wgrad_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (wg::LDS_FLOATS + wg::MAX_MAP_INTS) * sizeof(float)));
    if (device >= 0 && device < 64) done[device] = true;
    return 0;
}

static int conv3d_wgrad_impl(int device, u3d_stream_t stream, const u3d_src_t* src, const float* dz, float* dw, int cstride,
                             int N, int D, int H, int W, int Cout, float* workspace, size_t workspace_floats, const int* box = nullptr,
                             const u3d_gn_bwd_job_t* job = nullptr);

// Weight gradient over a BOX of dz voxels only (dz outside the box counts as zero; g is read from the whole volume): round 5, the
// near-boundary slab of a decoder level that upsamples n -> 2n + 1.  Workspace: u3d_wgrad_workspace_floats of the BOX dims.
extern "C" int u3d_conv3d_wgrad_box(int device, u3d_stream_t stream, const u3d_src_t* src, const float* dz, float* dw, int N, int D,
                                    int H, int W, int Cout, float* workspace, size_t workspace_floats, const int* box) {
    U3D_REQUIRE(box != nullptr, "u3d_conv3d_wgrad_box: box is NULL");
    return conv3d_wgrad_impl(device, stream, src, dz, dw, 0, N, D, H, W, Cout, workspace, workspace_floats, box);
}

extern "C" int u3d_conv3d_wgrad(int device, u3d_stream_t stream, const u3d_src_t* src, const float* dz, float* dw,
                                int N, int D, int H, int W, int Cout, float* workspace, size_t workspace_floats) {
    return conv3d_wgrad_impl(device, stream, src, dz, dw, 0, N, D, H, W, Cout, workspace, workspace_floats);
}

extern "C" int u3d_conv3d_wgrad_strided(int device, u3d_stream_t stream, const u3d_src_t* src, const float* dz, float* dw,
                                        int dw_cin_stride, int N, int D, int H, int W, int Cout, float* workspace,
                                        size_t workspace_floats) {
    U3D_REQUIRE(src && dw_cin_stride >= src->C0 + src->C1, "u3d_conv3d_wgrad_strided: dw_cin_stride < channels of src");
    return conv3d_wgrad_impl(device, stream, src, dz, dw, dw_cin_stride, N, D, H, W, Cout, workspace, workspace_floats);
}

static size_t wgrad_job_lds_bytes(int N, int C, int G) { return sizeof(double) * (4 * (size_t)N * C + 2 * (size_t)N * G); }

extern "C" int u3d_conv3d_wgrad_job_supported(int N, int C, int G) {
    // (the reduce kernel's own 1 KB of static LDS sits beside the job's tables)
    return (N > 0 && C > 0 && G > 0 && C % G == 0 && wgrad_job_lds_bytes(N, C, G) + 1024 <= 64 * 1024) ? 1 : 0;
}

extern "C" int u3d_conv3d_wgrad_job(int device, u3d_stream_t stream, const u3d_src_t* src, const float* dz, float* dw,
                                    int dw_cin_stride, int N, int D, int H, int W, int Cout, float* workspace,
                                    size_t workspace_floats, const u3d_gn_bwd_job_t* job) {
    U3D_REQUIRE(src && (dw_cin_stride == 0 || dw_cin_stride >= src->C0 + src->C1), "u3d_conv3d_wgrad_job: dw_cin_stride < channels of src");
    if (job) {
        U3D_REQUIRE(job->gstats_lo && job->mean_rstd && job->gamma && job->dgamma && job->dbeta && job->coef && job->C0 > 0 &&
                        job->C1 >= 0 && (job->C1 == 0) == (job->gstats_hi == nullptr) && (job->coef_hi == nullptr || job->C1 > 0) &&
                        job->reps_lo >= 0 && job->reps_lo <= 64 && job->reps_hi >= 0 && job->reps_hi <= 64,
                    "u3d_conv3d_wgrad_job: bad job");
        U3D_REQUIRE(u3d_conv3d_wgrad_job_supported(job->N, job->C0 + job->C1, job->G) == 1,
                    "u3d_conv3d_wgrad_job: the reduction of %d x %d channels in %d groups does not fit one block's LDS", job->N,
                    job->C0 + job->C1, job->G);
    }
    return conv3d_wgrad_impl(device, stream, src, dz, dw, dw_cin_stride, N, D, H, W, Cout, workspace, workspace_floats, nullptr, job);
}

static int conv3d_wgrad_impl(int device, u3d_stream_t stream, const u3d_src_t* src, const float* dz, float* dw, int cstride,
                             int N, int D, int H, int W, int Cout, float* workspace, size_t workspace_floats, const int* box,
                             const u3d_gn_bwd_job_t* job) {
    U3D_ENTER(device);
    if (int e = check_src(src, "u3d_conv3d_wgrad")) return e;
    U3D_REQUIRE(dz && dw && workspace && N > 0 && D > 0 && H > 0 && W > 0 && Cout > 0, "u3d_conv3d_wgrad: bad argument");
    U3D_REQUIRE((long long)N * D * H * W < (1ll << 31), "u3d_conv3d_wgrad: N*D*H*W must be < 2^31");
    WgradParams p;
    const int Cin = src->C0 + src->C1;
    p.bz0 = p.by0 = p.bx0 = 0, p.bz1 = D, p.by1 = H, p.bx1 = W;
    if (box) {
        U3D_REQUIRE(box[0] >= 0 && box[1] >= 0 && box[2] >= 0 && box[3] <= D && box[4] <= H && box[5] <= W && box[0] < box[3] &&
                        box[1] < box[4] && box[2] < box[5], "u3d_conv3d_wgrad_box: box outside the volume or empty");
        p.bz0 = box[0], p.by0 = box[1], p.bx0 = box[2], p.bz1 = box[3], p.by1 = box[4], p.bx1 = box[5];
    }
    wgrad_plan(N, p.bz1 - p.bz0, p.by1 - p.by0, p.bx1 - p.bx0, Cin, Cout, p);
    const size_t need = (size_t)p.S * p.nchunks * p.nkb * 27 * 1024;
    if (workspace_floats < need)
        return u3d_set_err(U3D_EWORKSPACE, "u3d_conv3d_wgrad: workspace %zu < %zu floats", workspace_floats, need);
    p.src = *src;
    p.dz = dz;
    p.partial = workspace;
    p.N = N, p.D = D, p.H = H, p.W = W, p.Cout = Cout;
    p.vec = src_vec_ok(src) ? 1 : 0;
    p.dzvec = (Cout % 4 == 0 && ((uintptr_t)dz & 15) == 0) ? 1 : 0;
    if (int e = wgrad_set_lds_once(device)) return e;
    const int nblk = p.S * p.nchunks * p.nkb;
    U3D_REQUIRE(D + H + W <= wg::MAX_MAP_INTS, "u3d_conv3d_wgrad: D+H+W must be <= %d", wg::MAX_MAP_INTS);
    const size_t shmem = (wg::LDS_FLOATS + (size_t)(D + H + W)) * sizeof(float);
    // no table look-ups (plain or exact-2x source) -> constant-offset staging (REG); ragged last tiles are part of its face masks
    // (key 3 = 2: the round-4 gate "every tile fully inside the volume", for A/B runs)
    const bool reg = !box && (g_u3d_tune[3] != 2 || (D % wg::TZ == 0 && H % wg::TY == 0 && W % wg::TX == 0)) &&
                     (src->C1 == 0 || (D == 2 * src->D1 && H == 2 * src->H1 && W == 2 * src->W1));
    if (p.vec && p.dzvec && reg && Cin <= 16)
        hipLaunchKernelGGL((conv3d_wgrad_kernel<true, true, true>), dim3(nblk), dim3(wg::NTHR), shmem, (hipStream_t)stream, p);
    else if (p.vec && p.dzvec && reg)
        hipLaunchKernelGGL((conv3d_wgrad_kernel<true, true>), dim3(nblk), dim3(wg::NTHR), shmem, (hipStream_t)stream, p);
    else if (p.vec && p.dzvec)
        hipLaunchKernelGGL(conv3d_wgrad_kernel<true>, dim3(nblk), dim3(wg::NTHR), shmem, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(conv3d_wgrad_kernel<false>, dim3(nblk), dim3(wg::NTHR), shmem, (hipStream_t)stream, p);
    U3D_LAUNCH_CHECK();
    const long long total = (long long)Cin * 27 * Cout;
    const bool one_group = p.S <= 16 && g_u3d_tune[23] != 1;  // key 23 = 1: always four split groups (A/B)
    const int rblocks = (int)((total + (one_group ? 255 : 63)) / (one_group ? 256 : 64));
    u3d_gn_bwd_job_t jb = {};
    size_t job_lds = 0;
    if (job) {
        jb = *job;
        if (!jb.gstats_hi) jb.C1 = 0, jb.hi_scale = 1.0f, jb.coef_hi = nullptr;
        if (jb.reps_lo < 1) jb.reps_lo = 1;
        if (jb.reps_hi < 1 || !jb.gstats_hi) jb.reps_hi = 1;
        job_lds = wgrad_job_lds_bytes(jb.N, jb.C0 + jb.C1, jb.G);
    }
    if (one_group)
        hipLaunchKernelGGL(wgrad_reduce_kernel<1>, dim3(rblocks + (job ? 1 : 0)), dim3(256), job_lds, (hipStream_t)stream, workspace, dw, p.S,
                           p.nchunks, p.nkb, Cin, Cout, cstride > 0 ? cstride : Cin, job ? 1 : 0, jb);
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel<4>, dim3(rblocks + (job ? 1 : 0)), dim3(256), job_lds, (hipStream_t)stream, workspace, dw, p.S,
                           p.nchunks, p.nkb, Cin, Cout, cstride > 0 ? cstride : Cin, job ? 1 : 0, jb);
    U3D_LAUNCH_CHECK();
    return 0;
}

// Host-only: which kernel variant a shape runs (the gates of conv3d_impl / conv3d_wgrad_impl restated for 16-byte aligned tensors with
// channel counts that are multiples of 4).  src_kind: 0 = plain source, 1 = virtual source whose low-res half is an exact 2x upsampling,
// 2 = virtual source through general index maps.
extern "C" int u3d_conv3d_variant(int N, int D, int H, int W, int Cin, int Cout, int src_kind, int has_workspace) {
    if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return -1;
    if (Cin % 4 != 0 || Cout % 4 != 0) return 0;
    const long long ntiles = (long long)N * cdiv(D, cv::TZ) * cdiv(H, cv::TY) * cdiv(W, cv::TX);
    const int nchunks = cdiv(Cin, 16), ntot = cdiv(Cout, 32);
    if (has_workspace && splitk_shape(N, D, H, W, Cin, Cout) && g_u3d_tune[7] != 2) {
        // (the launcher additionally needs the workspace to hold ksplit partial tensors: u3d_conv3d_workspace_floats sizes it for that)
        if (splitk_count(g_u3d_ncu_seen, ntiles, ntot, nchunks, nullptr) >= 2) return 3;
    }
    const bool ragged = D % cv::TZ != 0 || H % cv::TY != 0 || W % cv::TX != 0;
    if (src_kind == 2 || g_u3d_tune[3] == 1 || (g_u3d_tune[3] == 2 && ragged)) return 0;
    return ragged ? 2 : 1;
}

extern "C" int u3d_conv3d_wgrad_variant(int N, int D, int H, int W, int Cin, int Cout, int src_kind) {
    if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return -1;
    if (Cin % 4 != 0 || Cout % 4 != 0) return 0;
    const bool ragged = D % wg::TZ != 0 || H % wg::TY != 0 || W % wg::TX != 0;
    if (src_kind == 2 || (g_u3d_tune[3] == 2 && ragged)) return 0;
    return (ragged ? 2 : 1) | (Cin <= 16 ? 4 : 0);
}

extern "C" int u3d_conv3d_naive(int device, u3d_stream_t stream, const u3d_src_t* src, const float* w, float* out,
                                int N, int D, int H, int W, int Cin, int Cout, int relu, int flip) {
    U3D_ENTER(device);
    if (int e = check_src(src, "u3d_conv3d_naive")) return e;
    U3D_REQUIRE(w && out && Cin == src->C0 + src->C1, "u3d_conv3d_naive: bad argument");
    const long long total = (long long)N * D * H * W * Cout;
    const int blocks = (int)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
    hipLaunchKernelGGL(conv3d_naive_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *src, w, out, N, D, H,
                       W, Cin, Cout, relu, flip);
    U3D_LAUNCH_CHECK();
    return 0;
}
