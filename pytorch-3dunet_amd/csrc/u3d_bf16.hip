// u3d_bf16.hip — opt-in bf16-operand convolutions for BASELINE config 4 (ResidualUNet3D f_maps=64, "bf16 compute, fp32
// master weights"): the 3x3x3 Conv3d of buildingblocks.py:56 and its data gradient as an implicit GEMM on
// v_mfma_f32_32x32x16_bf16 — bf16 operands, FP32 accumulation, everything around it (activations in HBM, GroupNorm
// statistics, residual add, ReLU, parameter gradients) stays fp32.  Operands are converted while staging: activations
// (GroupNorm affine applied in fp32 first) in the global->LDS pass, weights by u3d_pack_weights_bf16 from the fp32 master
// copy (round-to-nearest-even, v_cvt_pk_bf16_f32).
//
// GEMM view: M = voxels, N = output channels, K = 27 taps x Cin.  Block = 4 waves, output tile (4*ZW) x 8 x 8 voxels x
// 32*NT channels; wave w owns z-planes [w*ZW, w*ZW+ZW) x two y-halves = 2*ZW M-tiles of 32 voxels (4 y x 8 x), so every
// A fragment (one ds_read_b128) feeds NT MFMAs and every B fragment (16 B per lane of packed weights, L1/L2 resident)
// feeds 2*ZW MFMAs: 4*ZW*NT MFMAs of 32 cycles per (tap, 16-channel chunk) against 2*ZW LDS reads and NT global loads.
// The (TZ+2) x 10 x 10 halo tile of a chunk lives in LDS as two channel-half planes [kh][hz][hy][hx pad 12][8 bf16]:
// a lane's fragment (8 channels of one voxel) is one 16-byte record and, with M-tile row r -> (y = r & 3, x = r >> 2),
// the 16 lanes of every ds_read_b128 service group hit 16 distinct 16-byte slots (row stride 12 records: checked by
// brute force against the lane groups of MI355X_MICROARCH.md §LDS).  Two buffers: chunk c+1 is fetched into registers in
// three batches under the three z-tap groups of chunk c and written to the other buffer after each group; one barrier
// per chunk.
#include "u3d_common.h"
#include "u3d_gn.h"
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// One staged bf16 item (8 channels of a halo voxel): GroupNorm affine in fp32, back to bf16, zero outside the volume (the padding
// applies AFTER the affine).  Written on dword pairs so that it compiles to 8 unpack + 4 v_pk_fma_f32 + 4 v_cvt_pk_bf16_f32 + 4
// v_and (20 VALU; the element-wise form cost ~50: single-source conversions, v_perm repacking, a select per element).
__device__ __forceinline__ bf16x8 u3d_stage_b16(const bf16x8& v, bool has_aff, bool ok, const f32x4& a0, const f32x4& b0, const f32x4& a1,
                                                const f32x4& b1) {
    const u32x4 raw = __builtin_bit_cast(u32x4, v);
    u32x4 o = raw;
    if (has_aff) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const f32x2 x = {__builtin_bit_cast(float, raw[e] << 16), __builtin_bit_cast(float, raw[e] & 0xffff0000u)};
            const f32x2 a = e < 2 ? f32x2{a0[2 * e], a0[2 * e + 1]} : f32x2{a1[2 * e - 4], a1[2 * e - 3]};
            const f32x2 b = e < 2 ? f32x2{b0[2 * e], b0[2 * e + 1]} : f32x2{b1[2 * e - 4], b1[2 * e - 3]};
            o[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(__builtin_elementwise_fma(x, a, b), bf16x2));
        }
    }
    const unsigned m = ok ? 0xffffffffu : 0u;
    return __builtin_bit_cast(bf16x8, u32x4{o[0] & m, o[1] & m, o[2] & m, o[3] & m});
}

extern int g_u3d_tune[24];  // csrc/u3d_conv.hip: run-time A/B knobs (u3d_set_tuning); results never change

namespace {

constexpr int HS = 12;  // padded halo row stride (16-byte records): conflict-free ds_read_b128 (see the header), >= 8 + KS - 1

struct bf16_conv_params {
    const float* x;        // (N,D,H,W,C) fp32
    const float* affine;   // (N,C,2) GroupNorm (a,b) per sample and channel, or null
    const bf16x8* wpk;     // packed weights [chunk][tap][n-tile][lane][8]
    float* y;              // (N,D,H,W,K) fp32
    const float* residual; // (N,D,H,W,K) or null: added before the ReLU
    const float* gx;       // (N,D,H,W,K) or null: forward input of the layer whose data gradient this is
    double* out_stats;     // [N][K][2] += (sum y, sum y^2) or null
    double* gstats;        // [N][K][2] += (sum y, sum y*gx) or null
    int N, D, H, W, C, K, relu;
    int tz, ty, tx;        // tiles per dimension
    int off;               // halo origin = tile origin - off: 1 for the 3x3x3 'same' convolution; the 2x2x2 kernels of the
                           // transposed convolution (§ below) read [i, i+1] (off 0, forward) or [i-1, i] (off 1, data gradient)
    const float* maskx;    // (N,D,H,W,K) or null: out = maskx > 0 ? out : 0 (ReLU mask of the tensor this gradient flows into)
    int ksplit;            // > 1: the channel reduction is split over `ksplit` blocks per (tile, channel block); raw partial
    float* ws;             //      sums go to ws[split][voxel][K] and splitk_bf16_reduce_kernel owns the epilogue
    long long wpart;       // split-fp32 kernels: distance (in bf16x8 records) between the high / middle / low weight images
    int order;             // bit 0: tile order of the grid: 1 (default) z fastest, then x, then y; 0 (u3d_set_tuning key 11 = 1) x-y-z raster;
                           // bit 1: the tile index runs fastest over the block ids (see the kernel)
    int b16;               // 1: x, y, residual, gx, maskx are bf16 tensors (activation storage, `_b16` entry points); the pointer
                           //    fields keep their float* type and are reinterpreted by the kernels' storage type T
    int t8mode, t8cs;      // 2x2x2 kernels of the transposed convolution (space-to-depth form, below): 1 forward / 2 data gradient,
                           //    Cs — which (tap, parity) weight blocks are structurally zero and are skipped; 0: nothing is skipped
};

template <int ZW, int KS>
struct tile_geom {
    // ZW = 3 is the FLAT tile (round 5): 5 x 10 x 10 voxels whose 500 voxels ARE the GEMM rows in raster order (row v = (z*10 + y)*10 + x,
    // 16 M-tiles of 32 rows, the last 12 rows idle) — the shape of config 4's bottom level (5 x 10 x 10: ONE tile instead of eight
    // 4 x 8 x 8 tiles that are 3/4 padding) and an exact divisor of the 10 x 20 x 20 level above it (8 tiles at 98 % instead of 27 at 58 %)
    static constexpr bool FLAT = ZW == 3;
    static constexpr int TZ = FLAT ? 5 : 4 * ZW, TY = FLAT ? 10 : 8, TX = FLAT ? 10 : 8;
    static constexpr int HZ = TZ + KS - 1, HY = TY + KS - 1, HX = TX + KS - 1, MT = FLAT ? 4 : 2 * ZW;
    static_assert(HX <= HS, "halo rows are HS records long");
    static constexpr int NTAPS = KS * KS * KS;
    // B-fragment ring: NTAPS % RING == 0 keeps the slots aligned across chunks (the packed image is linear in (chunk, tap));
    // fragments are fetched BDIST taps ahead — a tap is only 4*ZW*NT MFMAs = 128-256 cycles against ~500 cycles of L2 latency for
    // the first wave that touches a weight fragment.  The image carries BDIST taps of tail padding so that the prefetch never
    // needs a bounds branch.
    static constexpr int RING = KS == 3 ? 9 : 8;
    static constexpr int BDIST = 6;
    static constexpr int ADIST = 2;                       // A fragments (LDS) two taps ahead, ring of ADIST + 1 sets
    static constexpr int PLANE = HZ * HY * HS * 16 + 64;  // bytes; +64: the two planes' writes land on different banks; the
                                                          // pad also serves as the dump slot of items past the tile
    static constexpr int BUF = 2 * PLANE;
    static constexpr int ITEMS = HZ * HY * HX * 4;        // (halo voxel, channel quad) float4 items per chunk
    static constexpr int ITERS = (ITEMS + 255) / 256;
    static constexpr int NPARTS = KS * KS;                // staging parts per chunk = (z tap, y tap) groups of KS taps
    static constexpr int PER_PART = (ITERS + NPARTS - 1) / NPARTS;
    // bf16 activation storage: a staging item is a channel OCTET (16 bytes, one halo voxel x one channel-half plane) — half the
    // vector-memory instructions, half the LDS stores, full 16-byte lanes
    static constexpr int ITEMS8 = HZ * HY * HX * 2;
    static constexpr int ITERS8 = (ITEMS8 + 255) / 256;
    static constexpr int PER_PART8 = (ITERS8 + NPARTS - 1) / NPARTS;
};

// Epilogue shared by the bf16-operand and the split-fp32 kernels: residual, ReLU, fp32 store, per-(n,channel) statistics (or, with
// ksplit > 1, the raw partial sums of this block's chunk range).  `lds` is free for the block reduction when this runs.
template <int NT, int ZW, int KS, typename T = float>
__device__ __forceinline__ void conv_tile_epilogue(const bf16_conv_params& p, f32x16 (&acc)[tile_geom<ZW, KS>::MT][NT], char* lds, int n, int nb,
                                                   int split, int z0, int y0, int x0, int t, int lane, int w) {
    using G = tile_geom<ZW, KS>;
    // ---- epilogue: residual, ReLU, store, per-(n,channel) statistics.  C/D layout of the 32x32 MFMA: column = lane & 31,
    // row = (e & 3) + 8*(e >> 2) + 4*(lane >> 5); with row -> (yy = row & 3, xx = row >> 2): yy = e & 3, xx = 2*(e >> 2) + (lane >> 5)
    const int col = lane & 31, half = lane >> 5;
    if (p.ksplit > 1) {
        // raw partial sums of this chunk range; residual / ReLU / statistics happen in the fixed-order reduction
        float* wsp = p.ws + (size_t)split * p.N * p.D * p.H * p.W * p.K;
        if constexpr (std::is_same<T, __bf16>::value) {
            // (bf16 storage runs the MFMAs with swapped operands: lane = voxel (yy = col & 3, xx = col >> 2), register quad g = four
            // consecutive channels 8 g + 4 half .. + 3 — see the epilogue below)
#pragma unroll
            for (int m = 0; m < G::MT; ++m) {
                int z, y, xx;
                bool row_ok = true;
                if constexpr (G::FLAT) {  // GEMM row = raster index of the voxel inside the 5 x 10 x 10 tile
                    const int v = (w * G::MT + m) * 32 + col, vz = v / (G::TY * G::TX), vr = v - vz * (G::TY * G::TX), vy = vr / G::TX;
                    z = z0 + vz, y = y0 + vy, xx = x0 + vr - vy * G::TX;
                    row_ok = v < G::TZ * G::TY * G::TX;
                } else {
                    z = z0 + w * ZW + (m >> 1), y = y0 + (m & 1) * 4 + (col & 3), xx = x0 + (col >> 2);
                }
                if (row_ok && z < p.D && y < p.H && xx < p.W) {
                    const size_t vox = (((size_t)n * p.D + z) * p.H + y) * p.W + xx;
#pragma unroll
                    for (int j = 0; j < NT; ++j)
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            *reinterpret_cast<f32x4*>(wsp + vox * p.K + (size_t)(nb * NT + j) * 32 + 8 * g + 4 * half) =
                                f32x4{acc[m][j][4 * g], acc[m][j][4 * g + 1], acc[m][j][4 * g + 2], acc[m][j][4 * g + 3]};
                }
            }
            return;
        }
        if constexpr (G::FLAT) return;  // (bf16 storage only)
#pragma unroll
        for (int m = 0; m < G::MT; ++m) {
            const int z = z0 + w * ZW + (m >> 1);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int y = y0 + (m & 1) * 4 + (e & 3), xx = x0 + 2 * (e >> 2) + half;
                if (z < p.D && y < p.H && xx < p.W) {
                    const size_t vox = (((size_t)n * p.D + z) * p.H + y) * p.W + xx;
#pragma unroll
                    for (int j = 0; j < NT; ++j) wsp[vox * p.K + (size_t)(nb * NT + j) * 32 + col] = acc[m][j][e];
                }
            }
        }
        return;
    }
    if constexpr (G::FLAT) {
        return;  // the flat tile is launched with ksplit > 1 only (the fixed-order reduction owns the epilogue)
    } else {
    float s1[NT], s2[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) s1[j] = s2[j] = 0.f;
    const bool want_stats = p.out_stats != nullptr, want_g = p.gstats != nullptr;
    const T* side = reinterpret_cast<const T*>(p.residual ? p.residual : (want_g ? p.gx : p.maskx));  // the one tensor the epilogue reads (exclusive)
    if constexpr (std::is_same<T, __bf16>::value) {
        // bf16 storage: in the accumulator layout a lane owns ONE channel, i.e. 2-byte stores and 2-byte side loads — 64 + 64
        // vector-memory instructions per lane for 8 KB per wave (measured: the 64-channel layers 20-30 % slower than with fp32
        // storage).  The tile therefore goes through LDS (the halo buffers are free now) as [voxel][channel] bf16: the side
        // tensor comes in and the result goes out as 16-byte items (8 channels of a voxel per lane, 8 + 8 instructions per lane);
        // residual / mask / ReLU / rounding / statistics happen in between, in the accumulator layout, on 2-byte LDS accesses.
        constexpr int RSB = NT * 64 + 16;            // bytes per voxel row (+16: the two half-waves' rows land on different banks)
        constexpr int OCT = NT * 4;                  // 16-byte items per voxel
        constexpr int VPK = 256 / OCT;               // voxels per round of 256 items (32 or 64: divides a z-plane of 64)
        constexpr int NITEM = 64 * G::TZ / VPK;      // items per thread (the tile has 64 * TZ voxels)
        const int K = p.K;
        T* yout = reinterpret_cast<T*>(p.y);
        // item t + 256 k = channel octet o of voxel v0 + VPK k: everything that depends on k is a compile-time multiple of two strides
        // (this epilogue was ~1200 of the ~3300 VALU instructions a 64-channel tile executed; see DESIGN_HISTORY.md 4.7b)
        const int v0 = t / OCT, o = t - v0 * OCT, yv = v0 >> 3, xv = v0 & 7;
        const size_t sY = (size_t)p.W * K, sZ = (size_t)p.H * sY;
        const size_t gbase = ((((size_t)n * p.D + z0) * p.H + y0 + yv) * p.W + x0 + xv) * K + (size_t)nb * NT * 32 + o * 8;
        const int l0 = v0 * RSB + o * 16;
        const bool full = z0 + G::TZ <= p.D && y0 + 8 <= p.H && x0 + 8 <= p.W;
        auto item_in = [&](int k) {
            const int zk = (VPK * k) >> 6, yk = ((VPK * k) & 63) >> 3;
            return full || (z0 + zk < p.D && y0 + yk + yv < p.H && x0 + xv < p.W);
        };
        auto item_goff = [&](int k) { return gbase + (size_t)((VPK * k) >> 6) * sZ + (size_t)(((VPK * k) & 63) >> 3) * sY; };
        __syncthreads();  // every wave has left the k-loop: the halo buffers are free
        bf16x8 sv8[NITEM];
#pragma unroll
        for (int k = 0; k < NITEM; ++k) sv8[k] = bf16x8{};
        if (side) {
#pragma unroll
            for (int k = 0; k < NITEM; ++k)
                if (item_in(k)) sv8[k] = *reinterpret_cast<const bf16x8*>(side + item_goff(k));
            if (!want_g) {  // (residual / mask: needed per element, in the accumulator layout; gx only by the statistics, from sv8)
#pragma unroll
                for (int k = 0; k < NITEM; ++k) *reinterpret_cast<bf16x8*>(lds + l0 + VPK * k * RSB) = sv8[k];
                __syncthreads();
            }
        }
        // ---- in the accumulator layout.  The bf16-storage kernels run their MFMAs with SWAPPED operands (weights = A, voxels = B):
        // a lane then owns ONE voxel (column col: yy = col & 3, xx = col >> 2 of the M-tile) and its register quad g holds FOUR
        // CONSECUTIVE channels 8 g + 4 half .. + 3 of the n-tile — one 8-byte LDS access per quad for the side value and one for the
        // result (32 + 32 per lane) instead of the 128 + 128 two-byte accesses of the channel-per-lane layout, which were 19 % of the
        // 64-channel layers (profiles/r04_conv_b16_ablation.txt).  SIDE: 0 none, 1 residual (added), 2 ReLU mask of the tensor this
        // gradient flows into, 3 gx (GroupNorm-backward sums).  Statistics are taken in the output pass below, from the 16-byte items.
        char* const cb = lds + ((w * ZW * 8 + (col & 3)) * 8 + (col >> 2)) * RSB + half * 8;
        const float lowest = p.relu ? 0.f : -__builtin_inff();
        auto elements = [&](auto SIDE_) {
            constexpr int SIDE = decltype(SIDE_)::value;
#pragma unroll
            for (int m = 0; m < G::MT; ++m) {
#pragma unroll
                for (int j = 0; j < NT; ++j) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        char* c0 = cb + (((m >> 1) * 8 + (m & 1) * 4) * 8) * RSB + j * 64 + g * 16;
                        f32x4 v = {acc[m][j][4 * g], acc[m][j][4 * g + 1], acc[m][j][4 * g + 2], acc[m][j][4 * g + 3]};
                        if constexpr (SIDE == 1 || SIDE == 2) {
                            const u32x2 raw = *reinterpret_cast<const u32x2*>(c0);  // four bf16 (integer lanes: no float semantics on bit pairs)
                            const unsigned u0 = raw[0], u1 = raw[1];
                            const f32x4 sv = {__builtin_bit_cast(float, u0 << 16), __builtin_bit_cast(float, u0 & 0xffff0000u),
                                              __builtin_bit_cast(float, u1 << 16), __builtin_bit_cast(float, u1 & 0xffff0000u)};
                            if constexpr (SIDE == 1) v = v + sv;
                            if constexpr (SIDE == 2) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = sv[e] > 0.f ? v[e] : 0.f;
                            }
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], lowest);
                        const bf16x2 r0 = __builtin_convertvector(f32x2{v[0], v[1]}, bf16x2), r1 = __builtin_convertvector(f32x2{v[2], v[3]}, bf16x2);
                        *reinterpret_cast<u32x2*>(c0) = u32x2{__builtin_bit_cast(unsigned, r0), __builtin_bit_cast(unsigned, r1)};
                    }
                }
            }
        };
        // (gx is only needed by the statistics: it stays in the sv8 registers; the accumulators are dead after this pass)
        if (p.residual) elements(std::integral_constant<int, 1>{});
        else if (side && !want_g) elements(std::integral_constant<int, 2>{});
        else elements(std::integral_constant<int, 0>{});
        __syncthreads();
        // ---- output pass: 16-byte items (8 channels of a voxel) to global; per-channel sums of this thread's channel octet
        f32x2 q1[4], q2[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) q1[i] = q2[i] = f32x2{0.f, 0.f};
        const bool st = want_stats || want_g;
#pragma unroll
        for (int k = 0; k < NITEM; ++k) {
            if (item_in(k)) {
                const bf16x8 o8 = *reinterpret_cast<const bf16x8*>(lds + l0 + VPK * k * RSB);
                *reinterpret_cast<bf16x8*>(yout + item_goff(k)) = o8;
                if (st) {  // (statistics describe the STORED tensor)
                    const u32x4 ou = __builtin_bit_cast(u32x4, o8), gu = __builtin_bit_cast(u32x4, sv8[k]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const f32x2 vr = {__builtin_bit_cast(float, ou[i] << 16), __builtin_bit_cast(float, ou[i] & 0xffff0000u)};
                        const f32x2 gr = {__builtin_bit_cast(float, gu[i] << 16), __builtin_bit_cast(float, gu[i] & 0xffff0000u)};
                        q1[i] += vr;
                        q2[i] = __builtin_elementwise_fma(vr, want_g ? gr : vr, q2[i]);
                    }
                }
            }
        }
        if (st) {
            // fixed-order block reduction: the 256 / OCT threads of a channel octet, then one f64 atomic per (n, channel)
            __syncthreads();
            float* red = reinterpret_cast<float*>(lds);  // [thread][8 channels][2]
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<f32x4*>(red + t * 16 + 4 * i) = f32x4{q1[i][0], q2[i][0], q1[i][1], q2[i][1]};
            }
            __syncthreads();
            if (t < NT * 32 * 2) {
                const int ch = t >> 1, o_ = ch >> 3;
                double sum = 0.0;
#pragma unroll 4
                for (int i = 0; i < 256 / OCT; ++i) sum += (double)red[(i * OCT + o_) * 16 + (ch & 7) * 2 + (t & 1)];
                double* dst = (want_stats ? p.out_stats : p.gstats) + ((size_t)n * p.K + (size_t)nb * NT * 32) * 2;
                u3d_atomic_add_f64(dst + t, sum);
            }
        }
        return;
    } else {
    // Addressing is hoisted: element e of an accumulator tile sits (e & 3) rows and 2*(e >> 2) voxels from the tile's first
    // voxel, so one 64-bit base per (m, j) plus sixteen 32-bit offsets replaces a five-term index per element; tiles that lie
    // wholly inside the volume (all but the ragged rim) skip the per-element bounds tests.
    const int K = p.K, rowK = p.W * K;
    const bool full = z0 + G::TZ <= p.D && y0 + 8 <= p.H && x0 + 8 <= p.W;
    auto emit_tile = [&](auto FULL, int m, int j) {
        const int z = z0 + w * ZW + (m >> 1), yb = y0 + (m & 1) * 4, xb = x0 + half;
        const size_t base = ((((size_t)n * p.D + z) * p.H + yb) * p.W + xb) * K + (size_t)(nb * NT + j) * 32 + col;
        T* yp = reinterpret_cast<T*>(p.y) + base;
        const T* sp = side ? side + base : nullptr;
        auto inside = [&](int e) { return FULL.value || (z < p.D && yb + (e & 3) < p.H && xb + 2 * (e >> 2) < p.W); };
        f32x16 sv;  // all 16 side loads of this accumulator tile in flight before the first use
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            sv[e] = 0.f;
            if (sp && inside(e)) sv[e] = u3d_ld(sp + (e & 3) * rowK + (e >> 2) * 2 * K);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            if (inside(e)) {
                float v = acc[m][j][e];
                if (p.residual) v += sv[e];
                if (p.maskx && !(sv[e] > 0.f)) v = 0.f;
                if (p.relu) v = fmaxf(v, 0.f);
                v = u3d_stored(v, yp);  // (statistics describe the STORED tensor)
                u3d_st(yp + (e & 3) * rowK + (e >> 2) * 2 * K, v);
                if (want_stats) {
                    s1[j] += v;
                    s2[j] = fmaf(v, v, s2[j]);
                } else if (want_g) {
                    s1[j] += v;
                    s2[j] = fmaf(v, sv[e], s2[j]);
                }
            }
        }
    };
#pragma unroll
    for (int m = 0; m < G::MT; ++m) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if (full)
                emit_tile(std::true_type{}, m, j);
            else
                emit_tile(std::false_type{}, m, j);
        }
    }
    }  // (fp32 storage)
    if (want_stats || want_g) {
        // fixed-order block reduction through LDS (the halo buffers are free now), then one f64 atomic per (n, channel)
        __syncthreads();
        float* red = reinterpret_cast<float*>(lds);  // [8 = wave*2+half][NT*32][2]
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            red[((w * 2 + half) * NT * 32 + j * 32 + col) * 2 + 0] = s1[j];
            red[((w * 2 + half) * NT * 32 + j * 32 + col) * 2 + 1] = s2[j];
        }
        __syncthreads();
        if (t < NT * 32 * 2) {
            double sum = 0.0;
#pragma unroll
            for (int i = 0; i < 8; ++i) sum += (double)red[i * NT * 32 * 2 + t];
            double* dst = (want_stats ? p.out_stats : p.gstats) + ((size_t)n * p.K + (size_t)nb * NT * 32) * 2;
            u3d_atomic_add_f64(dst + t, sum);
        }
    }
    }  // (!FLAT)
}

// ABL: timing-only ablation bits that attributed the loop's cost (1 no B loads, 2 no A reads, 4 no staging, 8 no barrier, 16 no epilogue,
// 32 no prologue staging; results in DESIGN_HISTORY.md 4.7).  Only ABL = 0 is instantiated (-DU3D_CONV_ABL=.. builds: tools/ab_libs.sh).
template <int NT, int ZW, int KS, int ABL = 0, typename T = float>
__global__ __launch_bounds__(256, ZW == 3 ? 1 : (NT == 2 && ZW == 1 && KS == 3) ? 3 : 2) void conv3d_bf16_kernel(const bf16_conv_params p) {
    using G = tile_geom<ZW, KS>;
    // The 64-channel 3x3x3 tile with the full B ring (9 slots, 6 taps ahead) needs 202 VGPRs: two blocks per CU.  With fragments
    // only 2 taps ahead in a ring of 3 it fits 168 — THREE blocks per CU (LDS 46 KB each), and the third wave per SIMD hides more
    // than the shorter lead exposes: config 4 19.5 -> 19.2 ms per step (same-box pairs, profiles/r03_cfg4_ab.txt)
    constexpr bool SHORT = NT == 2 && KS == 3 && (ZW == 1 || std::is_same<T, __bf16>::value);
    // FLAT (the small wide levels: a few hundred blocks, each streaming ITS OWN slice of 14-57 MB of weights, most of it from HBM): ONE
    // block per CU with the whole register file — the ring holds U3D_FLAT_BDIST taps of fragments in flight per wave (the short ring's two
    // taps = 4 KB per block left the weight stream latency-bound: 0.9 k cycles per tap measured against 256 of MFMA work)
#ifndef U3D_FLAT_BRING
#define U3D_FLAT_BRING 9
#define U3D_FLAT_BDIST 8
#endif
    constexpr int B_RING = G::FLAT ? (KS == 3 ? U3D_FLAT_BRING : 8) : SHORT ? (KS == 3 ? 3 : 4) : G::RING;
    constexpr int B_DIST = G::FLAT ? (KS == 3 ? U3D_FLAT_BDIST : 7) : SHORT ? 2 : G::BDIST;
    static_assert(G::NTAPS % B_RING == 0 && B_DIST < B_RING || B_RING == G::NTAPS, "ring slots stay aligned across chunks");
    // (bf16 storage: A fragments one tap ahead — the 8-plane tile reads 4 per tap, the 4-plane tile sits at the 168-register line of
    // three blocks per CU and spilled 8 dwords with two taps of them)
    constexpr int A_DIST = (NT == 2 && std::is_same<T, __bf16>::value && KS == 3 && !G::FLAT) ? 1 : G::ADIST;
    constexpr int HY = G::HY;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int nblk = p.K / (32 * NT);
    const int bid = u3d_xcd_remap(blockIdx.x, gridDim.x);
    // Which index runs fastest over consecutive block ids (= blocks that run on one XCD at one time and share its L2):
    //   order bit 1 clear: the output-channel block, then the split, then the tile — neighbouring blocks share the ACTIVATION halo (large
    //     volumes: the halo is the traffic, a layer's weights stay L2-resident);
    //   order bit 1 set: the tile, then the split, then the channel block — neighbouring blocks stream the SAME weight fragments (small
    //     volumes with wide layers: config 4's 512 / 1024-channel levels have 27 / 8 tiles and 14 / 57 MB of bf16 weights, which every tile
    //     used to pull through the fabric again: 380-450 MB per launch).
    int nb, tile, split;
    if (p.order & 2) {
        const int ntile = p.N * p.tz * p.ty * p.tx;
        tile = bid % ntile;
        const int rest = bid / ntile;
        split = rest % p.ksplit;
        nb = rest / p.ksplit;
    } else {
        nb = bid % nblk;
        tile = bid / nblk;
        split = tile % p.ksplit;  // (ksplit == 1: 0)
        tile /= p.ksplit;
    }
    int txi, tyi, tzi, n;
    // z fastest: the tiles an XCD works on at one time (32 CUs x 3 blocks, consecutive ids) then share their z halos — the widest
    // (6 planes for 4) — through that XCD's L2: 0.49 -> 0.40 GB fetched per launch on config 4 (profiles/r03_tile_order.txt)
    if (p.order & 1) {
        tzi = tile % p.tz;
        tile /= p.tz;
        txi = tile % p.tx;
        tile /= p.tx;
        tyi = tile % p.ty;
        n = tile / p.ty;
    } else {
        txi = tile % p.tx;
        tile /= p.tx;
        tyi = tile % p.ty;
        tile /= p.ty;
        tzi = tile % p.tz;
        n = tile / p.tz;
    }
    const int z0 = tzi * G::TZ, y0 = tyi * G::TY, x0 = txi * G::TX;
    const int nch_all = p.C >> 4;
    const int cps = (nch_all + p.ksplit - 1) / p.ksplit;          // chunks per split
    const int cbeg = split * cps, nch = min(nch_all, cbeg + cps);  // this block's chunk range [cbeg, nch)
    const int ntiles = p.K >> 5;
    constexpr bool B16 = std::is_same<T, __bf16>::value;
    // staging role of this thread: fp32 storage -> channel QUAD q = t & 3 of halo voxel t >> 2 (+64 per item);
    // bf16 storage -> channel OCTET (= channel-half plane) q = t & 1 of halo voxel t >> 1 (+128 per item)
    constexpr int QS = B16 ? 1 : 2, VSTEP = 256 >> QS, NIT = B16 ? G::ITERS8 : G::ITERS, PER = B16 ? G::PER_PART8 : G::PER_PART;
    const int q = t & ((1 << QS) - 1);

    // A-fragment base of this lane: row r = lane & 31 -> (yy = r & 3, xx = r >> 2), channel half kh = lane >> 5
    const int r = lane & 31, kh = lane >> 5;
    const int a_base = kh * G::PLANE + (((w * ZW) * HY + (r & 3)) * HS + (r >> 2)) * 16;
    // FLAT: row r of M-tile m of wave w is voxel v = (4 w + m) * 32 + r of the tile in raster order (rows past the 500th read voxel 0:
    // computed, never stored) — one base per M-tile instead of one per lane
    int a_flat[G::FLAT ? G::MT : 1];
    if constexpr (G::FLAT) {
#pragma unroll
        for (int m = 0; m < G::MT; ++m) {
            int v = (w * G::MT + m) * 32 + r;
            v = v < G::TZ * G::TY * G::TX ? v : 0;
            const int vz = v / (G::TY * G::TX), vr = v - vz * (G::TY * G::TX), vy = vr / G::TX, vx = vr - vy * G::TX;
            a_flat[m] = kh * G::PLANE + ((vz * HY + vy) * HS + vx) * 16;
        }
    }
    auto a_addr = [&](int m, int tzz, int tyy, int txx) {  // byte offset of lane's A record for M-tile m at tap (tzz, tyy, txx)
        if constexpr (G::FLAT) return a_flat[m] + ((tzz * HY + tyy) * HS + txx) * 16;
        else return a_base + ((((m >> 1) + tzz) * HY + ((m & 1) * 4 + tyy)) * HS + txx) * 16;
    };

    // Staging descriptors, computed ONCE per block: per item the element offset from the tile's halo origin, a validity bit and
    // the LDS byte offset.  The hot loop is then BRANCH-FREE: out-of-volume items load a harmless in-tensor address (the tile's
    // own first voxel) and are zeroed by a select (zero padding applies AFTER the GroupNorm affine, as Conv3d(padding=1) on the
    // normalised tensor requires); items past the tile's halo write into the plane's pad bytes.  (Measured before: ~10 VALU
    // instructions per MFMA and a scalar branch per item and tap; every branch in the unrolled loop cost 1-2 % of the kernel.)
    int rel[NIT], lo[NIT];
    unsigned okmask = 0;
    {
        const int hv0 = t >> QS;
        const int bz = hv0 / (G::HY * G::HX), brem = hv0 - bz * (G::HY * G::HX), by = brem / G::HX, bx = brem - by * G::HX;
        const int rel_safe = ((p.off * p.H + p.off) * p.W + p.off) * p.C;  // the tile's first output voxel: always inside the volume
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            constexpr int HYX = G::HY * G::HX;
            const int dz = (VSTEP * it) / HYX, dy = ((VSTEP * it) % HYX) / G::HX, dx = (VSTEP * it) % G::HX;  // compile-time
            int hx = bx + dx, hy = by + dy, hz = bz + dz;
            if (hx >= G::HX) hx -= G::HX, hy += 1;
            if (hy >= G::HY) hy -= G::HY, hz += 1;
            if (hy >= G::HY) hy -= G::HY, hz += 1;  // (a 128-voxel step can carry twice into z on the small 2x2x2 halo)
            const int z = z0 - p.off + hz, y = y0 - p.off + hy, xx = x0 - p.off + hx;
            const bool ok = hz < G::HZ && (unsigned)z < (unsigned)p.D && (unsigned)y < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
            rel[it] = ok ? ((hz * p.H + hy) * p.W + hx) * p.C : rel_safe;
            okmask |= (ok ? 1u : 0u) << it;
            if constexpr (B16)
                lo[it] = hz < G::HZ ? q * G::PLANE + ((hz * G::HY + hy) * HS + hx) * 16 : G::PLANE - 64 + (t & 3) * 16;
            else
                lo[it] = hz < G::HZ ? (q >> 1) * G::PLANE + ((hz * G::HY + hy) * HS + hx) * 16 + (q & 1) * 8 : G::PLANE - 64 + (t & 7) * 8;
        }
    }
    // halo origin of the tile (may lie outside the tensor: only dereferenced through `rel`), this thread's channel quad / octet
    const T* xo = reinterpret_cast<const T*>(p.x) + ((((long long)n * p.D + (z0 - p.off)) * p.H + (y0 - p.off)) * p.W + (x0 - p.off)) * (long long)p.C + (B16 ? 8 : 4) * q;
    const int rel_dump = ((p.off * p.H + p.off) * p.W + p.off) * p.C;
    // one staged item in registers: 4 fp32 channels, or 8 bf16 channels (both 16 bytes)
    using item_t = typename std::conditional<B16, bf16x8, f32x4>::type;
    // the GroupNorm affine of this thread's channels in the staged chunk: (a, b) x 4 (quad) or x 8 (octet)
    struct aff_t {
        f32x4 a0, b0, a1, b1;
    };
    const bool has_aff = p.affine != nullptr;  // uniform; the data-gradient launches have none: bf16 items are then copied as they are
    // live = false (the last chunk has nothing to stage): every item re-reads one cached address instead of branching
    auto load_item = [&](int c, int it, item_t& v, bool live = true) {
        v = *reinterpret_cast<const item_t*>(xo + (live ? rel[it] : rel_dump) + (c << 4));
    };
    auto store_item = [&](char* buf, int it, const item_t& v, const aff_t& g, auto AFF) {
        const bool ok = (okmask >> it) & 1u;
        if constexpr (B16) {
            *reinterpret_cast<bf16x8*>(buf + lo[it]) = u3d_stage_b16(v, decltype(AFF)::value, ok, g.a0, g.b0, g.a1, g.b1);
        } else {
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (__bf16)(ok ? fmaf(v[e], g.a0[e], g.b0[e]) : 0.f);
            *reinterpret_cast<bf16x4*>(buf + lo[it]) = o;
        }
    };
    auto chunk_affine = [&](int c, aff_t& g) {
        if constexpr (B16) {
            u3d_load_affine(p.affine, n, p.C, (c << 4) + 8 * q, true, g.a0, g.b0);
            u3d_load_affine(p.affine, n, p.C, (c << 4) + 8 * q + 4, true, g.a1, g.b1);
        } else {
            u3d_load_affine(p.affine, n, p.C, (c << 4) + 4 * q, true, g.a0, g.b0);
        }
    };

    // ---- prologue: the first chunk into its buffer, 8 loads in flight per thread (the accumulators are not live yet)
    if (!(ABL & 32) && cbeg < nch) {
        aff_t g0;
        chunk_affine(cbeg, g0);
#pragma unroll
        for (int it0 = 0; it0 < NIT; it0 += 8) {
            item_t v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (it0 + i < NIT) load_item(cbeg, it0 + i, v[i]);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (it0 + i < NIT) {
                    if (has_aff) store_item(lds + (cbeg & 1) * G::BUF, it0 + i, v[i], g0, std::true_type{});
                    else store_item(lds + (cbeg & 1) * G::BUF, it0 + i, v[i], g0, std::false_type{});
                }
        }
    }
    f32x16 acc[G::MT][NT];
#pragma unroll
    for (int m = 0; m < G::MT; ++m)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][j][e] = 0.f;

    // (Spreading a tap's prefetches between its 4 MFMAs with sched_group_barrier — +8-12 % on the split-fp32 kernel below, whose taps
    // carry 12-24 MFMAs — measured -2 % here.)
    // Software pipeline (hipcc's own schedule issues every load right before its use): B fragments BDIST taps ahead in a register
    // ring that runs across chunk boundaries, A fragments ADIST taps ahead (restarted per chunk: the LDS buffer changes), the next
    // chunk's halo items fetched at the start of a KS-tap part and written at its end; sched_barriers pin "issue the prefetches,
    // then the 2*ZW*NT MFMAs of the tap".  The last chunk prefetches a clamped (repeated) chunk instead of branching.
    bf16x8 bq[B_RING][NT];
    if (cbeg < nch) {
        const bf16x8* wp0 = p.wpk + ((size_t)cbeg * G::NTAPS * ntiles + (size_t)nb * NT) * 64;  // wave-uniform base + lane
#pragma unroll
        for (int d = 0; d < B_DIST; ++d)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                if (B_DIST <= G::BDIST || cbeg * G::NTAPS + d < nch_all * G::NTAPS + G::BDIST) bq[d][j] = wp0[((size_t)d * ntiles + j) * 64 + lane];
    }
    // (two copies of the chunk loop, with and without the GroupNorm affine in the staging: the data-gradient launches have none
    // and copy their bf16 items as they are — a select per dword otherwise)
    // Transposed convolution in space-to-depth form (KS = 2): of the 64 (tap, output parity) weight blocks only 27 are non-zero.  A
    // block's output channels (forward) / a chunk's input channels (data gradient) belong to one parity p = pz*4 + py*2 + px when
    // Cs is a multiple of the channel run (else the union of the parities it touches): forward tap a is live iff a & ~p == 0, data-
    // gradient tap tau iff tau | p == 7.  Dead taps are skipped whole — B load, A reads, MFMAs: uniform branches.
    auto t8_mask_of = [&](int ch0, int nchan) -> unsigned {  // live-tap mask of the channel run [ch0, ch0 + nchan)
        if (KS != 2 || p.t8mode == 0) return 0xffu;
        const int p0 = min(ch0 / p.t8cs, 7), p1 = min((ch0 + nchan - 1) / p.t8cs, 7);
        unsigned m = 0;
        for (int pp = p0; pp <= p1; ++pp)
            for (int a = 0; a < 8; ++a)
                if (p.t8mode == 1 ? (a & ~pp & 7) == 0 : (a | pp) == 7) m |= 1u << a;
        return m;
    };
    const unsigned fwd_mask = (KS == 2 && p.t8mode == 1) ? t8_mask_of(nb * NT * 32, NT * 32) : 0xffu;
    auto chunk_loop = [&](auto AFF) {
    for (int c = cbeg; c < nch; ++c) {
        if constexpr (!(ABL & 8)) __syncthreads();  // buffer (c&1) is complete; everyone is done reading buffer ((c+1)&1)
        const char* cur = lds + (c & 1) * G::BUF;
        char* nxt = lds + ((c + 1) & 1) * G::BUF;
        const bool more = c + 1 < nch;
        const int cn = more ? c + 1 : c;  // the chunk staged under this one (the last chunk stages a dummy: never read)
        aff_t gaff;
        chunk_affine(cn, gaff);
        const bf16x8* wp = p.wpk + ((size_t)c * G::NTAPS * ntiles + (size_t)nb * NT) * 64;  // wave-uniform (scalar) base
        unsigned tm = 0xffffffffu, tm1 = 0xffffffffu;  // live taps of this chunk / of the next (the B ring runs ahead across chunks)
        if constexpr (KS == 2) {
            tm = p.t8mode == 2 ? t8_mask_of(c << 4, 16) : fwd_mask;
            tm1 = p.t8mode == 2 ? t8_mask_of((c + 1) << 4, 16) : fwd_mask;
        }
        auto live = [&](int tap_) { return KS != 2 || (((tap_ < G::NTAPS ? tm : tm1) >> (tap_ % G::NTAPS)) & 1u) != 0; };
        item_t st[PER];
        bf16x8 aq[A_DIST + 1][G::MT] = {};
#pragma unroll
        for (int d = 0; d < A_DIST; ++d) {
            const int tzz = d / (KS * KS), tyy = (d / KS) % KS, txx = d % KS;
#pragma unroll
            for (int m = 0; m < G::MT; ++m)
                aq[d][m] = *reinterpret_cast<const bf16x8*>(cur + a_addr(m, tzz, tyy, txx));
        }
#pragma unroll
        for (int part = 0; part < G::NPARTS; ++part) {  // part = (z tap, y tap)
#pragma unroll
            for (int i = 0; i < PER; ++i)
                if (!(ABL & 4) && part * PER + i < NIT) load_item(cn, part * PER + i, st[i], more);
#pragma unroll
            for (int t3 = 0; t3 < KS; ++t3) {
                const int tap = part * KS + t3;
                // (the image carries G::BDIST taps of tail padding: a longer lead stops at its end — uniform, last chunk only)
                if (live(tap + B_DIST) && (B_DIST <= G::BDIST || c * G::NTAPS + tap + B_DIST < nch_all * G::NTAPS + G::BDIST)) {
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        if constexpr (!(ABL & 1)) bq[(tap + B_DIST) % B_RING][j] = wp[((size_t)(tap + B_DIST) * ntiles + j) * 64 + lane];
                }
                if (!(ABL & 2) && tap + A_DIST < G::NTAPS && live(tap + A_DIST)) {
                    const int nt_ = tap + A_DIST, tzz = nt_ / (KS * KS), tyy = (nt_ / KS) % KS, txx = nt_ % KS;
#pragma unroll
                    for (int m = 0; m < G::MT; ++m)
                        aq[nt_ % (A_DIST + 1)][m] = *reinterpret_cast<const bf16x8*>(cur + a_addr(m, tzz, tyy, txx));
                }
                __builtin_amdgcn_sched_barrier(0);
                if (live(tap)) {
#pragma unroll
                    for (int m = 0; m < G::MT; ++m)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[m][j] = B16 ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[tap % B_RING][j], aq[tap % (A_DIST + 1)][m], acc[m][j], 0, 0, 0)
                                            : __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[tap % (A_DIST + 1)][m], bq[tap % B_RING][j], acc[m][j], 0, 0, 0);  // (B16: D^T, see the epilogue)
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < PER; ++i)
                if (!(ABL & 4) && part * PER + i < NIT) store_item(nxt, part * PER + i, st[i], gaff, AFF);
        }
    }
    };
    if (B16 && !has_aff) chunk_loop(std::false_type{});
    else chunk_loop(std::true_type{});

    if constexpr ((ABL & 16) != 0) {  // (timing: no epilogue; one store keeps the accumulators alive)
        float s_ = 0.f;
#pragma unroll
        for (int m = 0; m < G::MT; ++m)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) s_ += acc[m][j][e];
        if (s_ == 123.456f) p.y[0] = s_;
        return;
    }
    conv_tile_epilogue<NT, ZW, KS, T>(p, acc, lds, n, nb, split, z0, y0, x0, t, lane, w);
}

// out = [relu](sum over splits (fixed order) + residual), statistics like the fused epilogue.
// grid (voxel blocks, ceil(K/256), N); thread = one channel, loops over the block's voxels
template <typename T = float>
__global__ __launch_bounds__(256) void splitk_bf16_reduce_kernel(const bf16_conv_params p, long long V, int vper) {
    const T* presidual = reinterpret_cast<const T*>(p.residual);
    const T* pmaskx = reinterpret_cast<const T*>(p.maskx);
    const T* pgx = reinterpret_cast<const T*>(p.gx);
    T* py = reinterpret_cast<T*>(p.y);
    // block = 64 channels (16 float4 quads) x 16 voxel slots; a thread adds the splits of its quad for vper/16 voxels (fixed order),
    // the 16 slots' statistics meet in LDS (fixed order) and the block issues ONE f64 atomic per channel and sum — the first
    // version (thread = channel, one atomic pair per 16 voxels) spent most of its 60 us on 128-way contended atomics
    __shared__ float red[16][64][2];
    const int q = threadIdx.x & 15, slot = threadIdx.x >> 4;
    const int c = blockIdx.y * 64 + 4 * q, n = blockIdx.z;
    const long long v0 = (long long)blockIdx.x * vper, v1 = min(V, v0 + vper);
    const size_t split_stride = (size_t)p.N * V * p.K;
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    if (c < p.K) {
        for (long long v = v0 + slot; v < v1; v += 16) {
            const size_t o = ((size_t)n * V + v) * p.K + c;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
            for (int s = 0; s < p.ksplit; ++s) acc += *reinterpret_cast<const f32x4*>(p.ws + s * split_stride + o);
            if (p.residual) acc += u3d_ldq(presidual + o);
            if (p.maskx) {
                const f32x4 mx = u3d_ldq(pmaskx + o);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = mx[e] > 0.f ? acc[e] : 0.f;
            }
            if (p.relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = fmaxf(acc[e], 0.f);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = u3d_stored(acc[e], py);
            u3d_stq(py + o, acc);
            s1 += acc;
            if (p.out_stats)
                s2 += acc * acc;
            else if (p.gstats)
                s2 += acc * u3d_ldq(pgx + o);
        }
    }
    double* dst = p.out_stats ? p.out_stats : p.gstats;
    if (!dst) return;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        red[slot][4 * q + e][0] = s1[e];
        red[slot][4 * q + e][1] = s2[e];
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int cc = threadIdx.x >> 1, which = threadIdx.x & 1;
        const int ch = blockIdx.y * 64 + cc;
        if (ch < p.K) {
            double sum = 0.0;
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += (double)red[r][cc][which];
            u3d_atomic_add_f64(dst + ((size_t)n * p.K + ch) * 2 + which, sum);
        }
    }
}

// fp32 master weights (Cout,Cin,3,3,3) -> bf16 fragment image [chunk][tap][n-tile][lane][8]:
// mode 0 (forward):       B[k = input channel ][col = output channel] = w[col][k][tap]
// mode 1 (data gradient): B[k = output channel][col = input channel ] = w[k][col][26 - tap]   (taps flipped, roles swapped)
// lane l of a fragment holds column (l & 31) of n-tile `nt` and the 8 consecutive k = chunk*16 + 8*(l >> 5) + 0..7.
__global__ void pack_weights_bf16_kernel(const float* __restrict__ w, int Cout, int Cin, int mode, __bf16* __restrict__ out,
                                         long long total) {
    const int Kc = mode == 0 ? Cin : Cout;   // contraction channels
    const int Nc = mode == 0 ? Cout : Cin;   // produced channels
    const int ntiles = Nc >> 5;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(i & 7);
        long long rest = i >> 3;
        const int l = (int)(rest & 63);
        rest >>= 6;
        const int nt = (int)(rest % ntiles);
        rest /= ntiles;
        const int tap = (int)(rest % 27);
        const int c = (int)(rest / 27);
        const int k = c * 16 + 8 * (l >> 5) + e, col = nt * 32 + (l & 31);
        float v = 0.f;
        if (k < Kc && col < Nc) {
            if (mode == 0)
                v = w[((size_t)col * Cin + k) * 27 + tap];
            else
                v = w[((size_t)k * Cin + col) * 27 + (26 - tap)];
        }
        out[i] = (__bf16)v;
    }
}

// The same image for MANY weights in one launch, at HBM rate: a block owns one (16-channel chunk, 32-column n-tile) cell of one
// descriptor's image — 13,824 floats of the master weight that are 32 (mode 0) or 16 (mode 1) CONTIGUOUS runs in the reference
// layout — reads them coalesced into LDS and writes its 27 fragments (1 KiB each) 16 bytes per thread.  (The one-weight kernel
// above reads 4 bytes per thread at a stride of 27 floats: 16x read amplification, 28 us per image, 36 + 16 launches per
// config-4 step.)  desc.first = first block of the image; the block after an image's cells zeroes its 6-tap prefetch tail.
// Modes 4 / 5 (round 5): the 2x2x2 images of a ConvTranspose3d(k3, s2, p1) weight (Cl, Cs, 3,3,3) in space-to-depth form (forward /
// data gradient; see "Transposed convolution in SPACE-TO-DEPTH form" below: V[a][ci][p*Cs + co] = w[ci][co][s(a,p)], 27 of the 64 (tap,
// parity) blocks non-zero).  desc.Cin = Cl, desc.Cout = Cs, Cs % 32 == 0: a cell's 32 columns (mode 4) / 16 contraction rows (mode 5)
// then lie inside ONE parity and are the same contiguous runs of the reference layout as in modes 1 / 0 — the per-weight kernel
// (pack_convtr_t8_kernel: 4 bytes per thread at a stride of 27 floats, 8 launches of ~33 us per config-4 step) is only the fallback.
constexpr int PK_RS0 = 433;  // mode 0: [32 columns][16 k x 27 taps + 1]   (odd stride: the 32 lanes of a store hit 32 banks)
constexpr int PK_RS1 = 865;  // mode 1: [16 k][32 columns x 27 taps + 1]
__device__ __forceinline__ int pk_t8_sidx(int tap, int pp, bool fwd) {  // 3x3x3 tap index carried by (2x2x2 tap, parity), or -1
    int sidx = 0;
#pragma unroll
    for (int dd = 0; dd < 3; ++dd) {  // z (bit 2), y, x
        const int bit = 2 - dd;
        const int tb = (tap >> bit) & 1, pb = (pp >> bit) & 1, a = fwd ? tb : 1 - tb;
        const int sd = a == 0 ? (pb == 0 ? 1 : 2) : (pb == 1 ? 0 : -1);
        if (sd < 0) return -1;
        sidx = sidx * 3 + sd;
    }
    return sidx;
}
// Mode 6 (round 6): BOTH 3x3x3 images of a weight from ONE read of it.  Modes 0 and 1 each read the whole master weight (their cells are
// different 16 x 32 cuts of it): 1.13 GB of reads + 0.57 GB of writes per config-4 step at 5.9 TB/s.  A block of mode 6 owns a (32 output
// channels x 32 input channels) region — 32 contiguous runs of 864 floats — keeps it in LDS as bf16 (the same round-to-nearest-even
// conversion the fragment writes of modes 0 / 1 apply, so the images are bit for bit theirs) and writes the 2 x 27 fragments of the forward
// image (chunks 2*cib, 2*cib + 1 of n-tile cob) and the 2 x 27 of the data-gradient image (chunks 2*cob, 2*cob + 1 of n-tile cib, taps
// flipped).  The data-gradient image starts u3d_packed_weight_bf16_elems(Cin, Cout, 0) elements behind desc.packed.
constexpr int PK_RS6 = 868;  // bf16 elements per region row: 8-byte aligned rows, 434 words = 50 mod 64: 32 rows -> 32 distinct banks
__device__ __forceinline__ void pack_weights_bf16_both(const u3d_pack_desc_t& ds, int b, float* tile_f) {
    typedef __bf16 b16x4 __attribute__((ext_vector_type(4)));
    __bf16* tile = reinterpret_cast<__bf16*>(tile_f);
    const int t = threadIdx.x;
    const int Cin = ds.Cin, Cout = ds.Cout;
    const int nt0 = Cout >> 5, nt1 = Cin >> 5, nch0 = Cin >> 4, nch1 = Cout >> 4;
    __bf16* out0 = reinterpret_cast<__bf16*>(ds.packed);
    __bf16* out1 = out0 + ((size_t)nch0 * 27 + 6) * nt0 * 512;
    const int regions = nt0 * nt1;
    if (b >= regions) {  // two tail blocks: BDIST = 6 taps of zero fragments after the last chunk of each image
        bf16x8 z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = (__bf16)0.f;
        const bool second = b > regions;
        bf16x8* o8 = reinterpret_cast<bf16x8*>(second ? out1 + (size_t)nch1 * 27 * nt1 * 512 : out0 + (size_t)nch0 * 27 * nt0 * 512);
        const int cnt = 6 * (second ? nt1 : nt0) * 64;
        for (int i = t; i < cnt; i += 256) o8[i] = z;
        return;
    }
    const int cob = b / nt1, cib = b - cob * nt1;
    const float* base = ds.w + ((size_t)(cob * 32) * Cin + cib * 32) * 27;
    const size_t run_stride = (size_t)Cin * 27;
    // 6912 float4 of the region (32 runs of 216), 27 per thread, in two rounds of loads in flight
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        f32x4 v[14];
#pragma unroll
        for (int j = 0; j < 14; ++j) {
            const int i = t + 256 * (14 * half + j);
            v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (i < 6912) {
                const int r = i / 216, o = i - r * 216;
                v[j] = *reinterpret_cast<const f32x4*>(base + (size_t)r * run_stride + 4 * o);
            }
        }
#pragma unroll
        for (int j = 0; j < 14; ++j) {
            const int i = t + 256 * (14 * half + j);
            if (i < 6912) {
                const int r = i / 216, o = i - r * 216;
                *reinterpret_cast<b16x4*>(tile + r * PK_RS6 + 4 * o) = b16x4{(__bf16)v[j][0], (__bf16)v[j][1], (__bf16)v[j][2], (__bf16)v[j][3]};
            }
        }
    }
    __syncthreads();
    // forward image: lane l of fragment (chunk c2, tap) holds column co = l & 31 and k = input channels 16*c2 + 8*(l >> 5) + 0..7
    for (int i = t; i < 2 * 27 * 64; i += 256) {
        const int l = i & 63, tap = (i >> 6) % 27, c2 = i / (27 * 64);
        const __bf16* src = tile + (l & 31) * PK_RS6 + (16 * c2 + 8 * (l >> 5)) * 27 + tap;
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = src[e * 27];
        *reinterpret_cast<bf16x8*>(out0 + ((((size_t)(2 * cib + c2) * 27 + tap) * nt0 + cob) * 64 + l) * 8) = v;
    }
    // data-gradient image: column ci = l & 31, k = output channels 16*c2 + 8*(l >> 5) + 0..7, taps flipped
    for (int i = t; i < 2 * 27 * 64; i += 256) {
        const int l = i & 63, tap = (i >> 6) % 27, c2 = i / (27 * 64);
        const __bf16* src = tile + (16 * c2 + 8 * (l >> 5)) * PK_RS6 + (l & 31) * 27 + (26 - tap);
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = src[e * PK_RS6];
        *reinterpret_cast<bf16x8*>(out1 + ((((size_t)(2 * cob + c2) * 27 + tap) * nt1 + cib) * 64 + l) * 8) = v;
    }
}

__global__ __launch_bounds__(256) void pack_weights_bf16_batch_kernel(const u3d_pack_desc_t* __restrict__ descs, int n) {
    __shared__ __attribute__((aligned(16))) float tile[16 * PK_RS6];  // 13,888 floats (>= 32 * PK_RS0 = 13,856, 16 * PK_RS1 = 13,840; mode 6: 32 x 868 bf16)
    const int t = threadIdx.x;
    int d = 0;
    while (d + 1 < n && (long long)blockIdx.x >= descs[d + 1].first) ++d;
    const u3d_pack_desc_t ds = descs[d];
    const int Cin = ds.Cin, Cout = ds.Cout, mode = ds.mode;
    if (mode == 6) {
        pack_weights_bf16_both(ds, (int)((long long)blockIdx.x - ds.first), tile);
        return;
    }
    const bool t8 = mode >= 4;  // 4: T8 forward (Kc = Cl, Nc = 8 Cs), 5: T8 data gradient (Kc = 8 Cs, Nc = Cl)
    const int Kc = t8 ? (mode == 4 ? Cin : 8 * Cout) : (mode == 0 ? Cin : Cout), Nc = t8 ? (mode == 4 ? 8 * Cout : Cin) : (mode == 0 ? Cout : Cin);
    const int NTAPS = t8 ? 8 : 27;
    const int ntiles = Nc >> 5, nch = Kc >> 4;
    // T8: a block owns one SOURCE region — (16 input channels, 32 output channels) in mode 4, (16 output channels, 32 input channels) in
    // mode 5 — and writes the cells of all 8 parities that are made of it (one read of the weight per image instead of eight)
    const int src_tiles = t8 ? (mode == 4 ? Cout >> 5 : Cin >> 5) : ntiles, src_ch = t8 ? (mode == 4 ? Cin >> 4 : Cout >> 4) : nch;
    const int b = (int)((long long)blockIdx.x - ds.first);
    __bf16* out = reinterpret_cast<__bf16*>(ds.packed);
    if (b >= src_ch * src_tiles) {  // tail: BDIST = 6 taps of zero fragments after the last chunk
        bf16x8 z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = (__bf16)0.f;
        bf16x8* o8 = reinterpret_cast<bf16x8*>(out + (size_t)nch * NTAPS * ntiles * 512);
        for (int i = t; i < 6 * ntiles * 64; i += 256) o8[i] = z;
        return;
    }
    const int c = b / src_tiles, nt = b - c * src_tiles;  // (T8: indices of the source region)
    const float* w = ds.w;
    const int co0 = t8 ? (mode == 4 ? nt * 32 : c * 16) : 0;  // T8: first output channel (inside a parity) of the region
    const bool geo0 = mode == 0 || mode == 5;  // [32 runs (columns)][16 k x 27]; else [16 runs (k)][32 columns x 27]
    // 3456 float4 of the cell, 13.5 per thread, all loads in flight before the first LDS store (runs start at multiples of
    // 16 * 27 floats = 1728 bytes: 16-byte aligned whenever the parameter is)
    {
        const int RUN4 = geo0 ? 108 : 216;           // float4 per run: 432 / 864 floats
        const int RS = geo0 ? PK_RS0 : PK_RS1;
        const size_t run_stride = t8 ? (size_t)Cout * 27 : (size_t)Cin * 27;
        const float* base = mode == 0 ? w + ((size_t)(nt * 32) * Cin + c * 16) * 27
                          : mode == 1 ? w + ((size_t)(c * 16) * Cin + nt * 32) * 27
                          : mode == 4 ? w + ((size_t)(c * 16) * Cout + co0) * 27     // rows = 16 input channels, 32 output channels x 27 each
                                      : w + ((size_t)(nt * 32) * Cout + co0) * 27;   // rows = 32 input channels (columns), 16 output channels x 27 each
        f32x4 v[14];
#pragma unroll
        for (int j = 0; j < 14; ++j) {
            const int i = t + 256 * j;
            v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (i < 3456) {
                const int r = i / RUN4, o = i - r * RUN4;
                v[j] = *reinterpret_cast<const f32x4*>(base + (size_t)r * run_stride + 4 * o);
            }
        }
#pragma unroll
        for (int j = 0; j < 14; ++j) {
            const int i = t + 256 * j;
            if (i < 3456) {
                const int r = i / RUN4, o = i - r * RUN4;
                float* dst = tile + r * RS + 4 * o;
                dst[0] = v[j][0];
                dst[1] = v[j][1];
                dst[2] = v[j][2];
                dst[3] = v[j][3];
            }
        }
    }
    __syncthreads();
    if (t8) {
        // 8 parities x 8 taps = 64 fragments of this region: parity pp places it at column tile (pp*Cs + co0) / 32 (mode 4) or at
        // chunk (pp*Cs + co0) / 16 (mode 5); 27 of the 64 carry a tap, the others are written as zeros
        for (int i = t; i < 64 * 64; i += 256) {
            const int l = i & 63, tap = (i >> 6) & 7, pp = i >> 9;
            const int sidx = pk_t8_sidx(tap, pp, mode == 4);  // uniform per wave
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (__bf16)0.f;
            if (sidx >= 0 && mode == 4) {
                const float* src = tile + (8 * (l >> 5)) * PK_RS1 + (l & 31) * 27 + sidx;  // k = input channel, column = output channel
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (__bf16)src[e * PK_RS1];
            } else if (sidx >= 0) {
                const float* src = tile + (l & 31) * PK_RS0 + (8 * (l >> 5)) * 27 + sidx;  // column = input channel, k = output channel
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (__bf16)src[e * 27];
            }
            const int oc = mode == 4 ? c : (pp * Cout + co0) >> 4, ont = mode == 4 ? (pp * Cout + co0) >> 5 : nt;
            *reinterpret_cast<bf16x8*>(out + ((((size_t)oc * 8 + tap) * ntiles + ont) * 64 + l) * 8) = v;
        }
        return;
    }
    for (int i = t; i < 27 * 64; i += 256) {
        const int tap = i >> 6, l = i & 63;
        bf16x8 v;
        if (mode == 0) {
            const float* src = tile + (l & 31) * PK_RS0 + (8 * (l >> 5)) * 27 + tap;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (__bf16)src[e * 27];
        } else {
            const float* src = tile + (8 * (l >> 5)) * PK_RS1 + (l & 31) * 27 + (26 - tap);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (__bf16)src[e * PK_RS1];
        }
        *reinterpret_cast<bf16x8*>(out + ((((size_t)c * 27 + tap) * ntiles + nt) * 64 + l) * 8) = v;
    }
}

}  // namespace

extern "C" long long u3d_pack_weights_bf16_blocks(int Cin, int Cout, int mode) {
    if (mode == 6) {  // both 3x3x3 images from one read: (32 x 32)-channel regions + the two tails; 0 = not eligible
        if (Cin <= 0 || Cout <= 0 || Cin % 32 != 0 || Cout % 32 != 0) return 0;
        return (long long)(Cin / 32) * (Cout / 32) + 2;
    }
    if (mode == 4 || mode == 5) {  // T8 images of a transposed-convolution weight (Cin = Cl, Cout = Cs); 0 = not batchable
        if (Cin <= 0 || Cout <= 0 || Cin % 32 != 0 || Cout % 32 != 0) return 0;
        return mode == 4 ? (long long)(Cin / 16) * (Cout / 32) + 1 : (long long)(Cout / 16) * (Cin / 32) + 1;  // source regions + tail
    }
    const int Kc = mode == 0 ? Cin : Cout, Nc = mode == 0 ? Cout : Cin;
    if (Kc <= 0 || Nc <= 0 || Kc % 16 != 0 || Nc % 32 != 0 || (mode != 0 && mode != 1)) return 0;
    return (long long)(Kc / 16) * (Nc / 32) + 1;
}

extern "C" int u3d_pack_weights_bf16_batch(int device, u3d_stream_t stream, const u3d_pack_desc_t* descs_device, int n,
                                           long long total_blocks) {
    U3D_ENTER(device);
    U3D_REQUIRE(descs_device && n > 0 && total_blocks > 0 && total_blocks < 0x7fffffffLL, "u3d_pack_weights_bf16_batch: bad argument");
    hipLaunchKernelGGL(pack_weights_bf16_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, descs_device, n);
    U3D_LAUNCH_CHECK();
    return 0;
}

extern "C" long long u3d_packed_weight_bf16_elems(int Cin, int Cout, int mode) {
    const int Kc = mode == 0 ? Cin : Cout, Nc = mode == 0 ? Cout : Cin;
    if (Kc <= 0 || Nc <= 0 || Kc % 16 != 0 || Nc % 32 != 0) return 0;
    return ((long long)(Kc / 16) * 27 + 6) * (Nc / 32) * 64 * 8;  // + BDIST taps of tail padding (prefetched, never used)
}

extern "C" int u3d_pack_weights_bf16(int device, u3d_stream_t stream, const float* w, int Cout, int Cin, int mode,
                                     void* packed) {
    U3D_ENTER(device);
    const long long total = u3d_packed_weight_bf16_elems(Cin, Cout, mode);
    U3D_REQUIRE(w && packed && (mode == 0 || mode == 1) && total > 0,
                "u3d_pack_weights_bf16: needs contraction channels %% 16 == 0 and produced channels %% 32 == 0 (Cin %d, Cout %d)",
                Cin, Cout);
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_weights_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, Cout, Cin, mode,
                       reinterpret_cast<__bf16*>(packed), total);
    U3D_LAUNCH_CHECK();
    return 0;
}

extern "C" int u3d_conv3d_bf16_supported(int C, int K) { return (C > 0 && K > 0 && C % 16 == 0 && K % 32 == 0) ? 1 : 0; }

template <int NT, int ZW, int KS, int ABL = 0, typename T = float>
static int launch_bf16(const bf16_conv_params& p, hipStream_t stream) {
    using G = tile_geom<ZW, KS>;
    bf16_conv_params q = p;
    q.order = g_u3d_tune[11] == 1 ? 0 : 1;
    q.tz = (p.D + G::TZ - 1) / G::TZ;
    q.ty = (p.H + G::TY - 1) / G::TY;
    q.tx = (p.W + G::TX - 1) / G::TX;
    // few tiles, many channel blocks: blocks of one (channel block, split) — the same weights — next to each other (key 11 = 2 / 3: never / always)
    const long long ntile = (long long)p.N * q.tz * q.ty * q.tx;
    // (measured, profiles/r04_block_order_ab.txt: 3x3x3 at 10x20x20 / 5x10x10 -8...-10 %; the 2x2x2 kernels mixed: left on the old order)
    if (g_u3d_tune[11] == 3 || (g_u3d_tune[11] != 2 && KS == 3 && ntile <= 64 && (long long)p.C * p.K >= 256 * 256)) q.order |= 2;
    const long long blocks = (long long)p.N * q.tz * q.ty * q.tx * (p.K / (32 * NT)) * p.ksplit;
    if (blocks > 0x7fffffffLL) return u3d_set_err(U3D_EINVAL, "u3d_conv3d_bf16: grid too large");
    size_t shmem = 2 * (size_t)G::BUF;
    if (std::is_same<T, __bf16>::value && !G::FLAT && shmem < (size_t)64 * G::TZ * (NT * 64 + 16)) shmem = (size_t)64 * G::TZ * (NT * 64 + 16);  // epilogue tile
    // (per device, cheap: set on every launch so that every device of a multi-GPU process has it)
    U3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_bf16_kernel<NT, ZW, KS, ABL, T>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipLaunchKernelGGL((conv3d_bf16_kernel<NT, ZW, KS, ABL, T>), dim3((unsigned)blocks), dim3(256), shmem, stream, q);
    U3D_LAUNCH_CHECK();
    if (p.ksplit > 1) {
        const long long V = (long long)p.D * p.H * p.W;
        // voxels per block: 64 on big volumes; 16 (one per thread slot) where that would leave most CUs without a block — the 5 x 10 x 10
        // level ran this pass on 128 blocks, 4 voxels x ksplit serial 16-byte loads per thread: 24 us for 32 MB (rocprofv3, round 5)
        const int vper = (V + 63) / 64 * ((p.K + 63) / 64) * p.N >= 1024 ? 64 : 16;
        hipLaunchKernelGGL(splitk_bf16_reduce_kernel<T>, dim3((unsigned)((V + vper - 1) / vper), (unsigned)((p.K + 63) / 64), (unsigned)p.N),
                           dim3(256), 0, stream, q, V, vper);
        U3D_LAUNCH_CHECK();
    }
    return 0;
}

// how many ways the channel reduction is split: only when the natural grid (4-plane tiles) leaves most of the 256 CUs idle
// Tile variant of conv3d_bf16_kernel for a shape — ONE decision used by the launcher and by u3d_conv3d_bf16_tile_variant():
//   nt      64 output channels per block when possible, else 32;
//   planes  fp32 storage: 4-plane tiles everywhere (8-plane tiles — twice the B-fragment reuse — timed the same, profiles/r02i, and
//           have no registers left for the staging descriptors and the deeper rings: 223 spilled VGPRs; removed in round 5);
//           bf16 storage: 8-plane tiles where they still give two blocks per CU.  A wave then issues 8 MFMAs per pair of B fragments
//           instead of 4: the B stream (1 KiB per fragment and wave, from L2 through the CU's vector L1) is what bounds this kernel —
//           without it the same code runs 14-32 % faster (profiles/r03_bf16_ablation.txt) — and the short B ring, the one-deep A
//           ring and bf16 halo planes (77 KB of LDS per block) make the taller tile fit 256 VGPRs without spills.  +4-7 % on the 64-
//           and 128-channel layers, config 4 19.4 -> 18.9 ms (u3d_set_tuning key 10 = 1: 4-plane tiles everywhere);
//   blocks_per_cu  the __launch_bounds__ of the instantiation: three for the 64-channel 4-plane tile (168 VGPRs), else two.
struct Bf16Tile {
    int nt, planes, blocks_per_cu;
};
static Bf16Tile bf16_tile_choice(int N, int D, int H, int W, int K, bool b16, int ksplit) {
    Bf16Tile t;
    const bool nt2 = K % 64 == 0;
    const long long big = (long long)N * ((D + 7) / 8) * ((H + 7) / 8) * ((W + 7) / 8) * (K / (nt2 ? 64 : 32));
    const bool fits8 = big >= 512 && D >= 8 && ksplit == 1;
    // (round 5: the fp32-storage 8-plane instantiations — an A/B experiment behind tuning key 7 = 2 that never beat the 4-plane tile and
    // spilled 223 VGPRs — are gone: no shipped instantiation of conv3d_bf16_kernel spills inside its k-loop)
    const bool zw2 = b16 && nt2 && g_u3d_tune[10] != 1 && fits8;
    t.nt = nt2 ? 2 : 1;
    t.planes = zw2 ? 8 : 4;
    t.blocks_per_cu = (t.nt == 2 && t.planes == 4) ? 3 : 2;
    return t;
}

static int bf16_ksplit(int N, int D, int H, int W, int C, int K) {
    const bool nt2 = K % 64 == 0;
    const long long natural = (long long)N * ((D + 3) / 4) * ((H + 7) / 8) * ((W + 7) / 8) * (K / (nt2 ? 64 : 32));
    const int nch = C / 16;
    if (natural >= 384 || nch < 4) return 1;
    long long ks = (1024 + natural - 1) / natural;
    if (ks > nch / 2) ks = nch / 2;
    if (ks > 16) ks = 16;
    return ks < 2 ? 1 : (int)ks;
}

// Round 5: the FLAT 5 x 10 x 10 tile (tile_geom<3, 3>) for the small, wide levels that already split their channel reduction — bf16 storage,
// 64-channel blocks.  Chosen where it executes at most 3/4 of the padded GEMM rows of the 4 x 8 x 8 tiling (config 4: 512 instead of 2048
// rows at 5 x 10 x 10, 4096 instead of 6912 at 10 x 20 x 20); its split count aims at one block per CU (the kernel's occupancy) with at least two chunks per block.
// Returns the split count, 0 = not used.  u3d_set_tuning key 16: 1 = never, >= 2 = that split count (A/B).
// (need_split: the 3x3x3 convolutions and the transposed convolution's data gradient take it only where the 4 x 8 x 8 plan already splits its
// channel reduction; the transposed convolution's FORWARD — 8 Cs output columns, never split before — takes it wherever one block per CU needs
// at least two splits, i.e. on grids of at most 128 (tile, channel block) items)
static int bf16_flat_ksplit(int N, int D, int H, int W, int C, int K, bool need_split = true) {
    if (g_u3d_tune[16] == 1 || K % 64 != 0 || (need_split && bf16_ksplit(N, D, H, W, C, K) < 2)) return 0;
    const long long tiles4 = (long long)N * ((D + 3) / 4) * ((H + 7) / 8) * ((W + 7) / 8);
    const long long tilesf = (long long)N * ((D + 4) / 5) * ((H + 9) / 10) * ((W + 9) / 10);
    if (tilesf * 512 * 4 > tiles4 * 256 * 3) return 0;
    const int nch = C / 16;
    const long long natural = tilesf * (K / 64);
    long long ks = (256 + natural - 1) / natural;  // one block per CU
    if (g_u3d_tune[16] >= 2) ks = g_u3d_tune[16];
    if (ks > nch / 2) ks = nch / 2;
    if (ks > 32) ks = 32;
    return ks < 2 ? 0 : (int)ks;
}

extern "C" long long u3d_conv3d_bf16_workspace_floats(int N, int D, int H, int W, int C, int K) {
    if (!u3d_conv3d_bf16_supported(C, K) || N <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    int ks = bf16_ksplit(N, D, H, W, C, K);
    const int kf = bf16_flat_ksplit(N, D, H, W, C, K);  // (the b16 entry points may take the flat tile's plan: room for either)
    if (kf > ks) ks = kf;
    return ks > 1 ? (long long)ks * N * D * H * W * K : 0;
}

extern "C" int u3d_conv3d_bf16_ex(int device, u3d_stream_t stream, const float* x, const float* affine, const void* packed_w,
                                  float* out, int N, int D, int H, int W, int C, int K, int relu, double* out_stats,
                                  const float* gx, double* gstats, const float* residual, float* workspace,
                                  long long workspace_floats);

extern "C" int u3d_conv3d_bf16(int device, u3d_stream_t stream, const float* x, const float* affine, const void* packed_w,
                               float* out, int N, int D, int H, int W, int C, int K, int relu, double* out_stats,
                               const float* gx, double* gstats, const float* residual) {
    return u3d_conv3d_bf16_ex(device, stream, x, affine, packed_w, out, N, D, H, W, C, K, relu, out_stats, gx, gstats, residual,
                              nullptr, 0);
}

static int conv3d_bf16_impl(int device, u3d_stream_t stream, const float* x, const float* affine, const void* packed_w, float* out,
                           int N, int D, int H, int W, int C, int K, int relu, double* out_stats, const float* gx, double* gstats,
                           const float* residual, float* workspace, long long workspace_floats, int b16);

extern "C" int u3d_conv3d_bf16_ex(int device, u3d_stream_t stream, const float* x, const float* affine, const void* packed_w,
                                  float* out, int N, int D, int H, int W, int C, int K, int relu, double* out_stats,
                                  const float* gx, double* gstats, const float* residual, float* workspace,
                                  long long workspace_floats) {
    return conv3d_bf16_impl(device, stream, x, affine, packed_w, out, N, D, H, W, C, K, relu, out_stats, gx, gstats, residual, workspace,
                            workspace_floats, 0);
}

// bf16 ACTIVATION STORAGE (`activation_dtype: bf16`): x, out, gx and residual are bf16 NDHWC tensors (half the HBM bytes of every
// halo read, output write and epilogue side read); arithmetic, statistics (taken over the stored values), the affine table and
// the split-K scratch are unchanged
extern "C" int u3d_conv3d_bf16_ex_b16(int device, u3d_stream_t stream, const void* x, const float* affine, const void* packed_w,
                                      void* out, int N, int D, int H, int W, int C, int K, int relu, double* out_stats, const void* gx,
                                      double* gstats, const void* residual, float* workspace, long long workspace_floats) {
    return conv3d_bf16_impl(device, stream, (const float*)x, affine, packed_w, (float*)out, N, D, H, W, C, K, relu, out_stats,
                            (const float*)gx, gstats, (const float*)residual, workspace, workspace_floats, 1);
}

static int conv3d_bf16_impl(int device, u3d_stream_t stream, const float* x, const float* affine, const void* packed_w, float* out,
                           int N, int D, int H, int W, int C, int K, int relu, double* out_stats, const float* gx, double* gstats,
                           const float* residual, float* workspace, long long workspace_floats, int b16) {
    U3D_ENTER(device);
    U3D_REQUIRE(x && packed_w && out && N > 0 && D > 0 && H > 0 && W > 0, "u3d_conv3d_bf16: bad argument");
    U3D_REQUIRE(u3d_conv3d_bf16_supported(C, K), "u3d_conv3d_bf16: needs Cin %% 16 == 0 and Cout %% 32 == 0 (got %d, %d)", C, K);
    U3D_REQUIRE(!(out_stats && gstats), "u3d_conv3d_bf16: out_stats and gstats are mutually exclusive");
    U3D_REQUIRE(!gstats || gx, "u3d_conv3d_bf16: gstats needs gx");
    U3D_REQUIRE(!(residual && gstats), "u3d_conv3d_bf16: residual and gx/gstats are mutually exclusive");
    U3D_REQUIRE((((uintptr_t)x | (uintptr_t)packed_w | (uintptr_t)affine) & 15) == 0, "u3d_conv3d_bf16: 16-byte alignment");
    bf16_conv_params p{x, affine, reinterpret_cast<const bf16x8*>(packed_w), out, residual, gx, out_stats, gstats,
                       N, D, H, W, C, K, relu, 0, 0, 0, 1, nullptr, 1, nullptr};
    p.b16 = b16;
    const int ks = bf16_ksplit(N, D, H, W, C, K);
    const int kf = b16 ? bf16_flat_ksplit(N, D, H, W, C, K) : 0;
    hipStream_t s = (hipStream_t)stream;
    if (kf > 1 && workspace && workspace_floats >= (long long)kf * N * D * H * W * K) {
        p.ksplit = kf;
        p.ws = workspace;
        return launch_bf16<2, 3, 3, 0, __bf16>(p, s);
    }
    if (ks > 1 && workspace && workspace_floats >= (long long)ks * N * D * H * W * K) {
        p.ksplit = ks;
        p.ws = workspace;
    }
    const Bf16Tile tc = bf16_tile_choice(N, D, H, W, K, b16, p.ksplit);
#ifndef U3D_CONV_ABL
#define U3D_CONV_ABL 0  // timing experiments on the 8-plane bf16-storage tile (tools/ab_libs.sh; WRONG results): see the kernel's ABL bits
#endif
    if (b16 && tc.planes == 8) return launch_bf16<2, 2, 3, U3D_CONV_ABL, __bf16>(p, s);
    if (b16) return tc.nt == 2 ? launch_bf16<2, 1, 3, 0, __bf16>(p, s) : launch_bf16<1, 1, 3, 0, __bf16>(p, s);
    return tc.nt == 2 ? launch_bf16<2, 1, 3>(p, s) : launch_bf16<1, 1, 3>(p, s);
}

// host-only query of that choice (tests assert that the shapes they pin really run the variants the benchmarks run)
extern "C" int u3d_conv3d_bf16_tile_variant(int N, int D, int H, int W, int C, int K, int b16) {
    if (!u3d_conv3d_bf16_supported(C, K) || N <= 0 || D <= 0 || H <= 0 || W <= 0) return -1;
    const int ks = bf16_ksplit(N, D, H, W, C, K);
    const int kf = b16 ? bf16_flat_ksplit(N, D, H, W, C, K) : 0;
    if (kf > 1) return (kf << 16) | (5 << 8) | (2 << 4) | 2;  // planes = 5: the flat 5 x 10 x 10 tile
    const Bf16Tile tc = bf16_tile_choice(N, D, H, W, K, b16 != 0, ks);
    return (ks << 16) | (tc.planes << 8) | (tc.nt << 4) | tc.blocks_per_cu;
}

// =====================================================================================================================
// Weight gradient on v_mfma_f32_32x32x16_bf16:  dw[co][ci][tap] = sum_{n,v} g[n, v + tap - 1, ci] * dz[n, v, co],
// g = GroupNorm-affine(x) zero padded.  GEMM view: rows = 32 input channels, columns = 32 output channels, K = voxels
// (16 consecutive x positions per MFMA).  The contraction runs over VOXELS while both tensors are stored channel-
// contiguous (NDHWC), so each operand fragment (8 voxels of one channel per lane) is a transposed read: the tiles sit in
// LDS as [voxel][channel] bf16 and fragments are fetched with ds_read_b64_tr_b16, a pure shuffle inside 16-lane groups
// (out[l][j] = in[16*(l>>4) + 4*j + ((l&15)>>2)][(l&15)&3], tools/tr_probe.hip): source lane s of a group supplies the
// address of 4 channels of voxel (s >> 2), lane l receives channel (l & 15) of voxels j = 0..3.  The 32 lanes of one LDS
// cycle cover 4 voxels x 64 bytes = 256 contiguous bytes: conflict-free.
//
// Block = 8 waves owns (split s, 32 input channels, 64 output channels): wave w takes output-channel half (w >> 2) and
// the taps {w&3, (w&3)+4, ...} (7,7,7,6) — 7 accumulators of 16 registers.  Per 2 x 8 x 16 voxel tile the g halo tile
// (4 x 10 x 18 x 32 ch) and the dz tile (2 halves x 256 x 32 ch) are staged once (fp32 -> affine -> bf16); per 16-voxel
// row a wave reads its dz fragment once and one g fragment per tap.  Partial sums of the splits are written to a
// workspace and reduced in a FIXED order by wgrad_bf16_reduce_kernel straight into the reference layout.
namespace {

constexpr int WG_TZ = 2, WG_TY = 8, WG_TX = 16;
constexpr int WG_DZ_HALF = WG_TZ * WG_TY * WG_TX * 64;       // [z][y][x][32 co] bf16, two halves
constexpr int WG_DZ_ITEMS = WG_TZ * WG_TY * WG_TX * 16;      // (voxel, channel quad of 64)
constexpr int WG_DZ_ITERS = (WG_DZ_ITEMS + 511) / 512;       // 8 dz items per thread

// KS = 3: the 3x3x3 'same' convolution; KS = 2: the 2x2x2 kernels of the transposed convolution in space-to-depth form
template <int KS>
struct wg_geom {
    static constexpr int HZ = WG_TZ + KS - 1, HY = WG_TY + KS - 1, HX = WG_TX + KS - 1;
    static constexpr int G_BYTES = HZ * HY * HX * 64;         // [hz][hy][hx][32 ci] bf16 (KS 3: 46080)
    static constexpr int LDS = G_BYTES + 2 * WG_DZ_HALF;      // KS 3: 78848 bytes
    static constexpr int G_ITEMS = HZ * HY * HX * 8;          // (halo voxel, channel quad)
    static constexpr int G_ITERS = (G_ITEMS + 511) / 512;
    static constexpr int PARTS = 4;                            // 4 rows of 16 voxels each
    static constexpr int ITERS = ((G_ITERS + WG_DZ_ITERS + PARTS - 1) / PARTS) * PARTS;
    static constexpr int PER_PART = ITERS / PARTS;
    static constexpr int NTAPS = KS * KS * KS;
    static constexpr int NTW = (NTAPS + 3) / 4;                // taps (= accumulators) per wave: 7 / 2
    // bf16 activation storage: items are channel OCTETS (16 bytes): half the loads and LDS stores
    static constexpr int G_ITEMS8 = HZ * HY * HX * 4;
    static constexpr int G_ITERS8 = (G_ITEMS8 + 511) / 512;
    static constexpr int DZ_ITERS8 = (WG_TZ * WG_TY * WG_TX * 8 + 511) / 512;
    static constexpr int ITERS8 = ((G_ITERS8 + DZ_ITERS8 + PARTS - 1) / PARTS) * PARTS;
    static constexpr int PER_PART8 = ITERS8 / PARTS;
};

struct bf16_wgrad_params {
    const float* x;       // (N,D,H,W,C)
    const float* affine;  // (N,C,2) or null
    const float* dz;      // (N,D,H,W,K)
    float* ws;            // [S][P][KS^3][32][64] partial sums
    int N, D, H, W, C, K;
    int off;              // g halo origin = tile origin - off (1: 3x3x3 'same'; 0: the forward-looking 2x2x2 kernel)
    int tz, ty, tx;       // tiles per dimension
    int tiles;            // N*tz*ty*tx
    int per_block;        // tiles per split
    int pco;              // K / 64
    int xcd;              // 1: XCD-aware block order
    int t8cs;             // 2x2x2 weight gradient of the transposed convolution (KS = 2, space-to-depth form): Cs if a block's 64 output
                          // columns lie inside ONE output parity (Cs % 64 == 0), else 0.  Only 27 of the 64 (tap, parity) blocks carry a tap
                          // (tap a is live for parity p iff a & ~p == 0): a wave skips the fragment reads, MFMAs and stores of its dead taps
    float* dw;            // conv3d_wgrad_b16v2_kernel with ONE split (every block owns all tiles of its pair): the block writes its 27 x 32 x 64
                          // sums straight into dw[co][ci][tap] (through LDS, 3456-byte runs) and no reduction kernel follows; else null
};

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8 tr_frag(const char* lds_addr) {
    // two transposed reads: voxels +0..3 and +4..7 of this lane's 8-voxel half (4 voxels x 64 B apart)
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(uintptr_t)(uint32_t)(uintptr_t)lds_addr);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(uintptr_t)(uint32_t)(uintptr_t)(lds_addr + 4 * 64));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

struct wg_tile {
    int n, z0, y0, x0;
};

// One block per CU (8 waves, 163 VGPRs): the staging of tile i+1 must overlap the MFMAs of tile i INSIDE the block — two LDS
// buffers; the next tile's 20 items per thread are fetched in four batches of five at the start of each 4-row part of the
// current tile and written (affine, bf16) to the other buffer at the part's end; one barrier per tile.
template <int KS, typename T = float>
__global__ __launch_bounds__(512, 2) void conv3d_wgrad_bf16_kernel(const bf16_wgrad_params p) {
    const T* px = reinterpret_cast<const T*>(p.x);    // (the parameter block keeps float* fields; T is the storage type)
    const T* pdz = reinterpret_cast<const T*>(p.dz);
    using G = wg_geom<KS>;
    constexpr int WG_HY = G::HY, WG_HX = G::HX, WG_G_BYTES = G::G_BYTES, WG_LDS = G::LDS, WG_G_ITEMS = G::G_ITEMS,
                  WG_G_ITERS = G::G_ITERS, WG_ITERS = G::ITERS, WG_PARTS = G::PARTS, WG_PER_PART = G::PER_PART, NTW = G::NTW;
    extern __shared__ __attribute__((aligned(256))) char lds[];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int P = (p.C >> 5) * p.pco;
    // XCD-aware order: the P (input-chunk, output-block) pairs of one split read the SAME voxels — consecutive logical ids, i.e.
    // one XCD's L2, so that an x / dz tile comes from HBM once per split instead of once per pair (block b runs on XCD b % 8)
    const int bid = p.xcd ? u3d_xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int pair = bid % P, split = bid / P;
    const int cib = pair / p.pco, cob = pair % p.pco;
    const int c0 = cib * 32, k0 = cob * 64;
    const int h = w >> 2, wq = w & 3;
    // (round 5) live tap slots of this wave, wave-uniform; all of them except in the transposed convolution's 2x2x2 weight gradient,
    // whose LDS pipe — 3 transposed fragment reads per 2 MFMAs — was busy with blocks that are structurally zero: 0.075 of the bf16 peak
    unsigned live = (1u << NTW) - 1;
    if constexpr (KS == 2) {
        if (p.t8cs > 0) {
            const int pp = k0 / p.t8cs;
            live = 0;
#pragma unroll
            for (int i = 0; i < NTW; ++i)
                if (wq + 4 * i < G::NTAPS && ((wq + 4 * i) & ~pp & 7) == 0) live |= 1u << i;
        }
        live = __builtin_amdgcn_readfirstlane(live);
    }
    auto slot_live = [&](int i) { return KS != 2 || ((live >> i) & 1u) != 0; };

    auto decode = [&](int tile) {
        wg_tile r;
        int tt = tile;
        r.x0 = (tt % p.tx) * WG_TX;
        tt /= p.tx;
        r.y0 = (tt % p.ty) * WG_TY;
        tt /= p.ty;
        r.z0 = (tt % p.tz) * WG_TZ;
        r.n = tt / p.tz;
        return r;
    };
    // staging item `it` of this thread: it < GI: (halo voxel, channel quad t & 7) of the g tile, else (voxel, quad t & 15) of dz;
    // bf16 storage: channel OCTETS (t & 3 of g, t & 7 of dz), 16 bytes each, dz copied without a conversion
    constexpr bool B16 = std::is_same<T, __bf16>::value;
    constexpr int GI = B16 ? G::G_ITERS8 : WG_G_ITERS, DI = B16 ? G::DZ_ITERS8 : WG_DZ_ITERS, NITEMS_G = B16 ? G::G_ITEMS8 : WG_G_ITEMS;
    constexpr int NIT = B16 ? G::ITERS8 : WG_ITERS, PER = B16 ? G::PER_PART8 : WG_PER_PART;
    constexpr int GS = B16 ? 2 : 3, DS = B16 ? 3 : 4;  // log2(items per voxel) of g / dz
    using item_t = typename std::conditional<B16, bf16x8, f32x4>::type;
    struct aff_t {
        f32x4 a0, b0, a1, b1;
    };
    const bool has_aff = p.affine != nullptr;
    auto load_aff = [&](int n_, aff_t& g) {
        if constexpr (B16) {
            u3d_load_affine(p.affine, n_, p.C, c0 + 8 * (t & 3), true, g.a0, g.b0);
            u3d_load_affine(p.affine, n_, p.C, c0 + 8 * (t & 3) + 4, true, g.a1, g.b1);
        } else {
            u3d_load_affine(p.affine, n_, p.C, c0 + 4 * (t & 7), true, g.a0, g.b0);
        }
    };
    auto load_item = [&](const wg_tile& tl, int it, item_t& v) {
        v = item_t{};
        if (it < GI) {
            const int item = t + it * 512;
            if (item < NITEMS_G) {
                const int q = item & ((1 << GS) - 1), hv = item >> GS;
                const int hz = hv / (WG_HY * WG_HX), rem = hv - hz * (WG_HY * WG_HX);
                const int hy = rem / WG_HX, hx = rem - hy * WG_HX;
                const int z = tl.z0 - p.off + hz, y = tl.y0 - p.off + hy, xx = tl.x0 - p.off + hx;
                if ((unsigned)z < (unsigned)p.D && (unsigned)y < (unsigned)p.H && (unsigned)xx < (unsigned)p.W)
                    v = *reinterpret_cast<const item_t*>(px + ((((size_t)tl.n * p.D + z) * p.H + y) * p.W + xx) * p.C + c0 + (B16 ? 8 : 4) * q);
            }
        } else if (it < GI + DI) {
            const int item = t + (it - GI) * 512;
            const int q = item & ((1 << DS) - 1), vv = item >> DS;
            const int zl = vv / (WG_TY * WG_TX), rem = vv - zl * (WG_TY * WG_TX);
            const int yl = rem / WG_TX, xl = rem - yl * WG_TX;
            const int z = tl.z0 + zl, y = tl.y0 + yl, xx = tl.x0 + xl;
            if (z < p.D && y < p.H && xx < p.W && k0 + (B16 ? 8 : 4) * q < p.K)  // (K % 64 == 32: the block's upper columns do not exist)
                v = *reinterpret_cast<const item_t*>(pdz + ((((size_t)tl.n * p.D + z) * p.H + y) * p.W + xx) * p.K + k0 + (B16 ? 8 : 4) * q);
        }
    };
    auto store_item = [&](char* buf, const wg_tile& tl, int it, const item_t& v, const aff_t& g) {
        if (it < GI) {
            const int item = t + it * 512;
            if (item < NITEMS_G) {
                const int q = item & ((1 << GS) - 1), hv = item >> GS;
                const int hz = hv / (WG_HY * WG_HX), rem = hv - hz * (WG_HY * WG_HX);
                const int hy = rem / WG_HX, hx = rem - hy * WG_HX;
                const int z = tl.z0 - p.off + hz, y = tl.y0 - p.off + hy, xx = tl.x0 - p.off + hx;
                const bool ok = (unsigned)z < (unsigned)p.D && (unsigned)y < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
                if constexpr (B16) {  // (zero padding applies AFTER the affine; without one, outside the volume was loaded as zero)
                    *reinterpret_cast<bf16x8*>(buf + hv * 64 + q * 16) = u3d_stage_b16(v, has_aff, ok || !has_aff, g.a0, g.b0, g.a1, g.b1);
                } else {
                    bf16x4 o = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
                    if (ok) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (__bf16)fmaf(v[e], g.a0[e], g.b0[e]);
                    }
                    *reinterpret_cast<bf16x4*>(buf + hv * 64 + q * 8) = o;
                }
            }
        } else if (it < GI + DI) {
            const int item = t + (it - GI) * 512;
            const int q = item & ((1 << DS) - 1), vv = item >> DS;
            if constexpr (B16) {
                *reinterpret_cast<bf16x8*>(buf + WG_G_BYTES + (q >> 2) * WG_DZ_HALF + vv * 64 + (q & 3) * 16) = v;
            } else {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];  // (zero outside the volume: loaded as zero)
                *reinterpret_cast<bf16x4*>(buf + WG_G_BYTES + (q >> 3) * WG_DZ_HALF + vv * 64 + (q & 7) * 8) = o;
            }
        }
    };

    const int first = split * p.per_block, last = min(p.tiles, first + p.per_block);
    // ---- prologue: the first tile into buffer 0 (the accumulators are not live yet: 10 loads in flight per thread)
    if (first < last) {
        const wg_tile tl = decode(first);
        aff_t g0;
        load_aff(tl.n, g0);
#pragma unroll 1
        for (int it0 = 0; it0 < NIT; it0 += NIT / 2) {
            item_t v[NIT / 2];
#pragma unroll
            for (int i = 0; i < NIT / 2; ++i) load_item(tl, it0 + i, v[i]);
#pragma unroll
            for (int i = 0; i < NIT / 2; ++i) store_item(lds, tl, it0 + i, v[i], g0);
        }
    }

    f32x16 acc[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    // fragment addressing (see the header comment): group g4 = lane >> 4 -> channel half (g4 & 1), voxel half kh = g4 >> 1;
    // as a SOURCE lane, sidx = lane & 15 supplies voxel (sidx >> 2), channel quad (sidx & 3)
    const int g4 = lane >> 4, sidx = lane & 15;
    const int lane_off = (8 * (g4 >> 1) + (sidx >> 2)) * 64 + (16 * (g4 & 1) + 4 * (sidx & 3)) * 2;
    int a_off[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int tap = min(wq + 4 * i, G::NTAPS - 1);
        const int tz = tap / (KS * KS), ty = (tap / KS) % KS, tx = tap % KS;
        a_off[i] = lane_off + ((tz * WG_HY + ty) * WG_HX + tx) * 64;
    }
    const int b_off = WG_G_BYTES + h * WG_DZ_HALF + lane_off;

    for (int tile = first; tile < last; ++tile) {
        __syncthreads();  // buffer (tile - first) & 1 is complete; everyone is done reading the other one
        const char* cur = lds + ((tile - first) & 1) * WG_LDS;
        char* nxt = lds + ((tile - first + 1) & 1) * WG_LDS;
        const bool more = tile + 1 < last;
        const wg_tile tn = decode(more ? tile + 1 : tile);
        aff_t gaff;
        if (more) load_aff(tn.n, gaff);
        // bf16 storage (DEEP): the loads of part p are written at the end of part p + 1 — two batches in flight, twice the time for a
        // load to come back (the loop is fetch-latency bound: with cache-resident reads it runs 20 % faster)
        constexpr bool DEEP = B16 && WG_PARTS == 4;
        item_t stq[DEEP ? 2 : 1][PER];
        static_assert(WG_TZ * WG_TY == 16 && WG_PARTS == 4, "16 rows of 16 voxels per tile, 4 per part");
        constexpr int NS = 16 * NTW, AD = (NTW >= 2 && (B16 || KS != 3)) ? 2 : 1;  // steps per tile; fragments AD steps ahead (more spills)
        auto a_addr = [&](int s_) {  // step -> (row, tap slot) -> LDS offset of the g fragment
            const int rw = s_ / NTW, i = s_ - rw * NTW;
            const int zl = rw / WG_TY, yl = rw % WG_TY;
            return a_off[i] + (zl * WG_HY + yl) * WG_HX * 64;
        };
        auto b_addr = [&](int rw) { return b_off + rw * WG_TX * 64; };
        bf16x8 af[AD + 1] = {}, bfr[2] = {};
#pragma unroll
        for (int s_ = 0; s_ < AD; ++s_)
            if (slot_live(s_ % NTW)) af[s_] = tr_frag(cur + a_addr(s_));
        if (KS != 2 || live != 0) bfr[0] = tr_frag(cur + b_addr(0));
#pragma unroll
        for (int part = 0; part < WG_PARTS; ++part) {
            item_t(&st)[PER] = stq[DEEP ? (part & 1) : 0];
            if (more) {
#pragma unroll
                for (int i = 0; i < PER; ++i) load_item(tn, part * PER + i, st[i]);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- 4 rows of 16 voxels: one dz fragment per row, one g fragment + MFMA per tap of this wave.  The fragments run in a
            // ring AD steps ahead of their MFMA across the whole tile, pinned by sched_barriers: left to itself the compiler reuses
            // ONE register quadruple for every g fragment — read, wait for the LDS (100+ cycles), MFMA (32), read, wait, ... — which
            // is what held this kernel at 0.4 MFMA-busy with two waves per SIMD.
#pragma unroll
            for (int q4 = 0; q4 < 4 * NTW; ++q4) {
                const int s_ = part * 4 * NTW + q4, rw = s_ / NTW, i = s_ - rw * NTW;
                if (s_ + AD < NS && slot_live((s_ + AD) % NTW)) af[(s_ + AD) % (AD + 1)] = tr_frag(cur + a_addr(s_ + AD));
                if (i == 0 && rw + 1 < 16 && (KS != 2 || live != 0)) bfr[(rw + 1) & 1] = tr_frag(cur + b_addr(rw + 1));
                __builtin_amdgcn_sched_barrier(0);
                if (wq + 4 * i < G::NTAPS && slot_live(i))
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s_ % (AD + 1)], bfr[rw & 1], acc[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (more) {
                if constexpr (DEEP) {
                    if (part >= 1) {
#pragma unroll
                        for (int i = 0; i < PER; ++i) store_item(nxt, tn, (part - 1) * PER + i, stq[(part - 1) & 1][i], gaff);
                    }
                    if (part == WG_PARTS - 1) {
#pragma unroll
                        for (int i = 0; i < PER; ++i) store_item(nxt, tn, part * PER + i, st[i], gaff);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < PER; ++i) store_item(nxt, tn, part * PER + i, st[i], gaff);
                }
            }
        }
    }
    // ---- partial sums: ws[split][pair][tap][ci 32][co 64]; D layout: column = lane & 31 (co), row = ci
    float* dst = p.ws + ((size_t)split * P + pair) * G::NTAPS * 2048;
    const int col = lane & 31, half = lane >> 5;
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int tap = wq + 4 * i;
        if (tap < G::NTAPS && slot_live(i)) {  // (a dead block is never read by wgrad_t8_reduce_kernel)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ci = (e & 3) + 8 * (e >> 2) + 4 * half;
                dst[(size_t)tap * 2048 + ci * 64 + h * 32 + col] = acc[i][e];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 4: the bf16-STORAGE 3x3x3 weight gradient, second form (same tile, LDS layout, splits, workspace layout and — operand for
// operand, in the same order — the same MFMA sequence per accumulator as the kernel above: results are bit-identical).  What round
// 3's counters said about the first form: 5.4 vector instructions per MFMA in the tile loop and an LDS pipe busier than the matrix
// pipe (8 waves x 128 fragments x 8 cycles = 8192 cycles per tile against 7168 of MFMA).  Changes:
//   * a wave owns taps {w, w + 8, w + 16, w + 24} and BOTH output-channel halves: a g fragment feeds two MFMAs, 43 fragment reads
//     per row and block instead of 64 (LDS pipe 8192 -> 5504 cycles per tile);
//   * staging without coordinate arithmetic: item k of a thread is (halo voxel (t >> 2) + 128 k, channel octet t & 3) — its offset
//     inside the tile's halo box is ONE register computed before the loop, its LDS address t * 16 + 8192 k an immediate, the tile
//     origin the scalar offset of a buffer load.  Border / ragged tiles derive a validity bit per item from the packed halo
//     coordinates and send invalid lanes out of the buffer's range (they read zero); interior tiles skip even that;
//   * loads in three batches (4 + 3 + 3 items) that are written one part LATER than they were issued (>= 2 parts in flight);
//   * the tile index is advanced digit-wise (scalar), not decoded by divisions.
#ifndef U3D_WG_ABLATE
#define U3D_WG_ABLATE 0  // timing experiments (tools/ab_libs.sh; WRONG results): 1 no global loads, 2 no LDS stores, 4 no g fragment
#endif                   // reads in the loop, 8 no dz fragment reads, 16 no per-tile barrier
// X8: tile 4 x 8 x 8 voxels (an MFMA's 16 voxels = two y rows of 8) instead of 2 x 8 x 16 — a halo of 600 instead of 720 voxels, and
// no 16-voxel rounding of W (config 4's 40-, 20- and 10-wide levels).  The summation order over voxels differs from the 16-wide tile's.
template <bool X8>
struct wg2_geom {
    static constexpr int TZ = X8 ? 4 : 2, TY = 8, TX = X8 ? 8 : 16;
    static constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2, NH = HZ * HY * HX;
    static constexpr int GB = NH * 64, LDSB = GB + 2 * WG_DZ_HALF;   // g halo tile + the two dz halves
    static constexpr int NG = (NH * 4 + 511) / 512, ND = 4;          // g / dz items (16 bytes) per thread and tile
    static constexpr int GLAST = NH * 4 - 512 * (NG - 1);            // threads that own a last g item (320 / 352)
    static constexpr int LDS_TOTAL = 2 * LDSB + 16;                  // + the dump slot of the others
};
template <bool X8>
__global__ __launch_bounds__(512, 2) void conv3d_wgrad_b16v2_kernel(const bf16_wgrad_params p) {
    using G = wg2_geom<X8>;
    constexpr int TZ = G::TZ, TY = G::TY, TX = G::TX, HZ = G::HZ, HY = G::HY, HX = G::HX, GB = G::GB, LDSB = G::LDSB;
    static_assert(TZ * TY * TX == 256 && G::NG <= 6 && G::NG >= 5, "256 voxels per tile = 16 rows of 16; 5 or 6 g items");
    extern __shared__ __attribute__((aligned(256))) char lds[];
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int P = (p.C >> 5) * p.pco;
    const int bid = p.xcd ? u3d_xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int pair = bid % P, split = bid / P;
    const int cib = pair / p.pco, cob = pair % p.pco;
    const int c0 = cib * 32, k0 = cob * 64;
    const int D = p.D, H = p.H, W = p.W;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)((long long)p.N * D * H * W * p.C * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rdz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dz), 0, (int)((long long)p.N * D * H * W * p.K * 2), 0x00020000);
    constexpr int NG = G::NG, ND = G::ND, NTOT = NG + ND;
    // ---- per-thread staging constants
    int relg[NG];       // byte offset of g item k relative to the halo origin
    unsigned gpos[3] = {0, 0, 0};  // 16 bits per item: hz | hy << 4 | hx << 8 of its halo voxel (hz >= 4: dead tail item)
#pragma unroll
    for (int k = 0; k < NG; ++k) {
        const int hv = (t >> 2) + 128 * k;
        const int hz = hv / (HY * HX), rem = hv - hz * (HY * HX), hy = rem / HX, hx = rem - hy * HX;
        relg[k] = (((hz * H + hy) * W + hx) * p.C + c0 + 8 * (t & 3)) * 2;
        gpos[k >> 1] |= ((unsigned)hz | ((unsigned)hy << 4) | ((unsigned)hx << 8)) << (16 * (k & 1));
    }
    // dz item k: voxel (t >> 3) + 64 k, channel octet t & 7; 16-wide tile: (z k >> 1, y 4 (k & 1) + (t >> 7), x (t >> 3) & 15),
    // 8-wide tile: (z k, y t >> 6, x (t >> 3) & 7)
    const int dyl = X8 ? t >> 6 : t >> 7, dxl = (t >> 3) & (TX - 1);
    auto dz_zk = [](int k) { return X8 ? k : k >> 1; };
    auto dz_yk = [](int k) { return X8 ? 0 : 4 * (k & 1); };
    const unsigned imask = 0xf00u | ((1u << NG) - 1u) & ~(t < G::GLAST ? 0u : 1u << (NG - 1));  // interior tile: this thread's live items
    const int reld = ((dyl * W + dxl) * p.K + k0 + 8 * (t & 7)) * 2;
    const int ldsg = t * 16;
    const int ldsd = GB + ((t & 7) >> 2) * WG_DZ_HALF + (t >> 3) * 64 + (t & 3) * 16;
    const bool has_aff = p.affine != nullptr;
    struct tile_t {
        int n, zi, yi, xi;
    };
    struct tinfo {         // everything a tile's staging needs
        int baseg, based;  // byte offsets of the halo origin in x / of the tile origin in dz (uniform)
        unsigned mask;     // bits 0..5: g item k of THIS thread is inside the volume; bits 8..11: dz item k is
    };
    auto info_of = [&](const tile_t& c, bool exists) {
        tinfo r;
        const int z0 = c.zi * TZ, y0 = c.yi * TY, x0 = c.xi * TX;
        r.baseg = ((((c.n * D + z0 - 1) * H + y0 - 1) * W + x0 - 1) * p.C) * 2;
        r.based = ((((c.n * D + z0) * H + y0) * W + x0) * p.K) * 2;
        const bool interior = z0 >= 1 && z0 + TZ + 1 <= D && y0 >= 1 && y0 + TY + 1 <= H && x0 >= 1 && x0 + TX + 1 <= W;
        if (!exists) {
            r.mask = 0;  // past the block's last tile: every lane reads beyond the buffers' range (zeros, no traffic)
        } else if (interior) {
            r.mask = imask;
        } else {
            // halo coordinate h is inside iff lo <= h < hi with lo = max(0, 1 - origin), hi = min(extent, size + 1 - origin)
            const unsigned zlo = z0 == 0 ? 1u : 0u, zn = (unsigned)min(HZ, D + 1 - z0) - zlo;
            const unsigned ylo = y0 == 0 ? 1u : 0u, yn = (unsigned)min(HY, H + 1 - y0) - ylo;
            const unsigned xlo = x0 == 0 ? 1u : 0u, xn = (unsigned)min(HX, W + 1 - x0) - xlo;
            r.mask = 0;
#pragma unroll
            for (int k = 0; k < NG; ++k) {
                const unsigned gp = gpos[k >> 1] >> (16 * (k & 1));
                const unsigned hz = gp & 0xfu, hy = (gp >> 4) & 0xfu, hx = (gp >> 8) & 0xffu;
                const bool ok = hz - zlo < zn && hy - ylo < yn && hx - xlo < xn;
                r.mask |= (ok ? 1u : 0u) << k;
            }
#pragma unroll
            for (int k = 0; k < ND; ++k) {
                const bool ok = dz_zk(k) < D - z0 && dz_yk(k) + dyl < H - y0 && dxl < W - x0;
                r.mask |= (ok ? 1u : 0u) << (8 + k);
            }
        }
        return r;
    };
    auto advance = [&](tile_t& c) {
        if (++c.xi == p.tx) {
            c.xi = 0;
            if (++c.yi == p.ty) {
                c.yi = 0;
                if (++c.zi == p.tz) {
                    c.zi = 0;
                    ++c.n;
                }
            }
        }
    };
    // item slot j of a tile: j < 6: g item j, else dz item j - 6.  Invalid lanes read at offset -1: beyond the range, zero.
    auto load_slot = [&](const tinfo& ti, int j) -> bf16x8 {
        if constexpr ((U3D_WG_ABLATE & 1) != 0) return bf16x8{};
        if (j < NG) {
            const int off = ((ti.mask >> j) & 1u) ? relg[j] + ti.baseg : -1;
            return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
        }
        const int k = j - NG;
        const int sk = ti.based + ((dz_zk(k) * H + dz_yk(k)) * W * p.K) * 2;  // uniform
        const int off = ((ti.mask >> (8 + k)) & 1u) ? reld + sk : -1;
        return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rdz, off, 0, 0));
    };
    struct aff_t {
        f32x4 a0, b0, a1, b1;
    };
    auto store_slot = [&](char* buf, const tinfo& ti, int j, const bf16x8& v, const aff_t& g) {
        if constexpr ((U3D_WG_ABLATE & 2) != 0) return;
        if (j < NG) {
            char* dst = buf + ldsg + 8192 * j;
            if (j == NG - 1) dst = t < G::GLAST ? dst : lds + 2 * LDSB;  // (the others' last slot would land in the dz tile: dump slot)
            *reinterpret_cast<bf16x8*>(dst) = u3d_stage_b16(v, has_aff, ((ti.mask >> j) & 1u) != 0, g.a0, g.b0, g.a1, g.b1);
        } else {
            *reinterpret_cast<bf16x8*>(buf + ldsd + 4096 * (j - NG)) = v;
        }
    };
    auto load_aff = [&](int n_, aff_t& g) {
        u3d_load_affine(p.affine, n_, p.C, c0 + 8 * (t & 3), true, g.a0, g.b0);
        u3d_load_affine(p.affine, n_, p.C, c0 + 8 * (t & 3) + 4, true, g.a1, g.b1);
    };

    const int first = split * p.per_block, last = min(p.tiles, first + p.per_block);
    tile_t t1;  // the tile AFTER the one being computed (first + 1 at loop entry)
    {
        int tt = first;
        t1.xi = tt % p.tx;
        tt /= p.tx;
        t1.yi = tt % p.ty;
        tt /= p.ty;
        t1.zi = tt % p.tz;
        t1.n = tt / p.tz;
    }
    aff_t gaff;
    int aff_n = -1;
    // Staging schedule (steps of the 64-step tile loop; every item waits 36 steps = ~half a tile between its load and its LDS write;
    // one load and one store every six steps instead of batches — a batch of four 1 KiB loads issued by all eight waves at the same
    // step blocked each of them for 400-1300 cycles, profiles/r04_wgrad_b16v2_ablation.txt):
    //   items 0-4 of tile i+2: loaded at steps 34, 40, .., 58 of iteration i, written at steps 6, 12, .., 30 of iteration i+1
    //   items 5-9 of tile i+1: loaded at steps  2,  8, .., 26 of iteration i, written at steps 38, 44, .., 62 of iteration i
    bf16x8 ra[5], rb[5];
    tinfo i1;  // of tile i + 1 (whose items are written during iteration i)
    if (first < last) {  // prologue: the first tile into buffer 0, items 0-4 of the second into registers
        const tinfo ti = info_of(t1, true);
        load_aff(t1.n, gaff);
        aff_n = t1.n;
#pragma unroll 1
        for (int j0 = 0; j0 < 10; j0 += 5) {
            bf16x8 v[5];
#pragma unroll
            for (int i = 0; i < 5; ++i)
                if (j0 + i < NTOT) v[i] = load_slot(ti, j0 + i);
#pragma unroll
            for (int i = 0; i < 5; ++i)
                if (j0 + i < NTOT) store_slot(lds, ti, j0 + i, v[i], gaff);
        }
        advance(t1);
        i1 = info_of(t1, first + 1 < last);
#pragma unroll
        for (int i = 0; i < 5; ++i) ra[i] = load_slot(i1, i);
    }

    constexpr int NTW = 4;
    f32x16 acc[NTW][2];
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][hh][e] = 0.f;

    const int g4 = lane >> 4, sidx = lane & 15;
    // (voxel half g4 >> 1 of an MFMA's 16 voxels: x + 8 on the 16-wide tile, the next y row on the 8-wide one)
    const int lane_off = ((X8 ? HX : 8) * (g4 >> 1) + (sidx >> 2)) * 64 + (16 * (g4 & 1) + 4 * (sidx & 3)) * 2;
    int a_off[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int tap = min(w + 8 * i, 26);
        const int tz = tap / 9, ty = (tap / 3) % 3, tx = tap % 3;
        a_off[i] = lane_off + ((tz * HY + ty) * HX + tx) * 64;
    }
    const int b_off = GB + (8 * (g4 >> 1) + (sidx >> 2)) * 64 + (16 * (g4 & 1) + 4 * (sidx & 3)) * 2;  // (the dz tile's 16 voxels of a row are contiguous)
    const bool four = w < 3;  // this wave's fourth tap exists (w + 24 < 27)

#ifdef U3D_WG_TRACE  // timeline build (tools/wgrad_timeline.py): s_memtime stamps of tiles 4 .. 15 of every wave of the middle block -> its ws region
    unsigned* trace = reinterpret_cast<unsigned*>(lds + 2 * LDSB + 16);
#define WG_STAMP(k)                                                                             \
    do {                                                                                        \
        const int tr_ = tile - first - 4;                                                       \
        if (tr_ >= 0 && tr_ < 12) {                                                             \
            const unsigned c_ = (unsigned)__builtin_amdgcn_s_memtime();                         \
            if (lane == 0) trace[(w * 12 + tr_) * 14 + (k)] = c_;                               \
        }                                                                                       \
    } while (0)
#else
#define WG_STAMP(k)
#endif
    for (int tile = first; tile < last; ++tile) {
        WG_STAMP(0);
        if constexpr ((U3D_WG_ABLATE & 16) == 0) __syncthreads();  // buffer (tile - first) & 1 is complete; everyone is done reading the other one
        WG_STAMP(1);
        const char* cur = lds + ((tile - first) & 1) * LDSB;
        char* nxt = lds + ((tile - first + 1) & 1) * LDSB;
        if (tile + 1 < last && t1.n != aff_n) {  // (every LDS write of this iteration belongs to tile + 1)
            load_aff(t1.n, gaff);
            aff_n = t1.n;
        }
        tile_t t2 = t1;
        tinfo i2;
        constexpr int NS = 16 * NTW, AD = 2;
        auto a_addr = [&](int s_) {
            const int rw = s_ / NTW, i = s_ - rw * NTW;
            return a_off[i] + (X8 ? ((rw >> 2) * HY + 2 * (rw & 3)) : ((rw >> 3) * HY + (rw & 7))) * HX * 64;
        };
        auto b_addr = [&](int rw, int hh) { return b_off + hh * WG_DZ_HALF + rw * 16 * 64; };
        bf16x8 af[AD + 1], bfr[2][2];
#pragma unroll
        for (int s_ = 0; s_ < AD; ++s_) af[s_] = tr_frag(cur + a_addr(s_));
        bfr[0][0] = tr_frag(cur + b_addr(0, 0));
        bfr[0][1] = tr_frag(cur + b_addr(0, 1));
#pragma unroll
        for (int part = 0; part < 4; ++part) {
            WG_STAMP(2 + 3 * part);
#pragma unroll
            for (int q4 = 0; q4 < 4 * NTW; ++q4) {
                const int s_ = part * 4 * NTW + q4, rw = s_ / NTW, i = s_ - rw * NTW;
                if (s_ + AD < NS) {
                    const int i2_ = (s_ + AD) % NTW;
                    if (((U3D_WG_ABLATE & 4) == 0) && (i2_ < 3 || four)) af[(s_ + AD) % (AD + 1)] = tr_frag(cur + a_addr(s_ + AD));
                }
                if (((U3D_WG_ABLATE & 8) == 0) && i == 0 && rw + 1 < 16) {
                    bfr[(rw + 1) & 1][0] = tr_frag(cur + b_addr(rw + 1, 0));
                    bfr[(rw + 1) & 1][1] = tr_frag(cur + b_addr(rw + 1, 1));
                }
                // (A progress-fair s_setprio between the two waves of a SIMD — each wave publishing its row count in LDS and raising its
                // priority while it was behind — did balance them (barrier waits 2000 -> 300 cycles) and made the kernel 8 % SLOWER:
                // profiles/r04_wgrad_b16v2_ablation.txt.  The older wave wins the pipe by default, its partner finishes the tile alone.)
                // ---- the tile after next: coordinates and validity masks, computed under this tile's MFMAs (right after the barrier they
                // were ~900 exposed cycles per border tile)
                if (s_ == 30) {
                    advance(t2);
                    i2 = info_of(t2, tile + 2 < last);
                }
                // ---- one staging operation every third step (see the schedule above)
                if (s_ < 32) {
                    if (s_ % 6 == 2 && 5 + s_ / 6 < NTOT) rb[s_ / 6] = load_slot(i1, 5 + s_ / 6);
                    if (s_ % 6 == 0 && s_ > 0) store_slot(nxt, i1, s_ / 6 - 1, ra[s_ / 6 - 1], gaff);
                } else {
                    if ((s_ - 32) % 6 == 2) ra[(s_ - 32) / 6] = load_slot(i2, (s_ - 32) / 6);
                    if ((s_ - 32) % 6 == 0 && s_ > 32 && 5 + (s_ - 32) / 6 - 1 < NTOT) store_slot(nxt, i1, 5 + (s_ - 32) / 6 - 1, rb[(s_ - 32) / 6 - 1], gaff);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (i < 3 || four) {
                    acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s_ % (AD + 1)], bfr[rw & 1][0], acc[i][0], 0, 0, 0);
                    acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s_ % (AD + 1)], bfr[rw & 1][1], acc[i][1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            WG_STAMP(3 + 3 * part);
            WG_STAMP(4 + 3 * part);
        }
        t1 = t2;
        i1 = i2;
    }
#ifdef U3D_WG_TRACE
    if (bid == (int)gridDim.x / 2) {
        __syncthreads();
        unsigned* out = reinterpret_cast<unsigned*>(p.ws + ((size_t)split * P + pair) * 27 * 2048);  // (this block's own region)
        for (int i = t; i < 8 * 12 * 14; i += 512) out[i] = trace[i];
        return;
    }
#endif
    const int col = lane & 31, half = lane >> 5;
    if (p.dw) {
        // ---- one split: this block's sums ARE the gradient.  The reference layout dw[co][ci][tap] makes a block's output 64 rows (co) of
        // 32 x 27 = 864 consecutive floats; the accumulators hold (ci, co) tiles per tap.  Two passes (one 32-column half each) through
        // LDS [co 32][864 + 1] (the staging buffers are free now): 16-byte coalesced stores instead of the 113 MB of partial sums +
        // a transposing reduction pass (51 us at 1024 channels, rocprofv3) that the workspace path costs when there is nothing to reduce
        constexpr int RS = 27 * 32 + 1;
        static_assert(32 * RS * 4 <= G::LDS_TOTAL, "the transposition tile must fit the staging buffers");
        float* tl = reinterpret_cast<float*>(lds);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < NTW; ++i) {
                const int tap = w + 8 * i;
                if (tap < 27) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int ci = (e & 3) + 8 * (e >> 2) + 4 * half;
                        tl[col * RS + ci * 27 + tap] = acc[i][hh][e];
                    }
                }
            }
            __syncthreads();
            for (int idx = t; idx < 32 * 216; idx += 512) {
                const int r = idx / 216, q4 = idx - r * 216, co = k0 + hh * 32 + r;
                if (co < p.K) {
                    const float* src = tl + r * RS + 4 * q4;
                    *reinterpret_cast<f32x4*>(p.dw + ((size_t)co * p.C + c0) * 27 + 4 * q4) = f32x4{src[0], src[1], src[2], src[3]};
                }
            }
        }
        return;
    }
    // ---- partial sums: ws[split][pair][tap][ci 32][co 64]; D layout: column = lane & 31 (co), row = ci
    float* dst = p.ws + ((size_t)split * P + pair) * 27 * 2048;
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int tap = w + 8 * i;
        if (tap < 27) {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int ci = (e & 3) + 8 * (e >> 2) + 4 * half;
                    dst[(size_t)tap * 2048 + ci * 64 + hh * 32 + col] = acc[i][hh][e];
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 5: the SAME staging scheme for the 2x2x2 weight gradient of the transposed convolution in space-to-depth form (KS = 2, bf16
// storage).  It ran on the round-3 kernel (conv3d_wgrad_bf16_kernel<2>): per-item coordinate arithmetic with integer divisions for
// every staged item of every tile, and only 2 accumulators per wave to hide it behind — 8 us per 256-voxel tile, 0.075 of the bf16 peak,
// 1.1 ms per config-4 step; skipping its structurally zero blocks changed nothing (3 %): it is the staging, not the MFMAs.  Here: 4 x 8
// x 8 tiles with a 5 x 9 x 9 forward-looking halo (the 2x2x2 taps read x[i + a], a in {0,1}^3), staging items as per-thread constants +
// buffer loads (invalid lanes read beyond the range), wave w = tap w with BOTH 32-column halves (one g fragment feeds two MFMAs), a tile
// loop of 16 steps in which item j of tile i + 2 is loaded at step 2j right after the same register's item of tile i + 1 was written to
// LDS — every load a whole tile period in flight — and dead (tap, parity) blocks skipped (p.t8cs).
struct wgt_geom {
    static constexpr int TZ = 4, TY = 8, TX = 8, HZ = 5, HY = 9, HX = 9, NH = HZ * HY * HX;  // 405 halo voxels
    static constexpr int GB = NH * 64, LDSB = GB + 2 * WG_DZ_HALF;
    static constexpr int NG = (NH * 4 + 511) / 512, ND = 4, NTOT = NG + ND;                   // 4 + 4 items of 16 bytes per thread and tile
    static constexpr int GLAST = NH * 4 - 512 * (NG - 1);                                     // threads that own a last g item (84)
    static constexpr int LDS_TOTAL = 2 * LDSB + 16;
};
__global__ __launch_bounds__(512, 2) void conv3d_wgrad_t8v2_kernel(const bf16_wgrad_params p) {
    using G = wgt_geom;
    constexpr int TZ = G::TZ, TY = G::TY, TX = G::TX, HZ = G::HZ, HY = G::HY, HX = G::HX, GB = G::GB, LDSB = G::LDSB;
    constexpr int NG = G::NG, ND = G::ND, NTOT = G::NTOT;
    static_assert(TZ * TY * TX == 256 && NTOT == 8, "256 voxels per tile = 16 rows of 16; 8 staging items per thread");
    extern __shared__ __attribute__((aligned(256))) char lds[];
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int P = (p.C >> 5) * p.pco;
    const int bid = p.xcd ? u3d_xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int pair = bid % P, split = bid / P;
    const int cib = pair / p.pco, cob = pair % p.pco;
    const int c0 = cib * 32, k0 = cob * 64;
    const int D = p.D, H = p.H, W = p.W;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)((long long)p.N * D * H * W * p.C * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rdz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dz), 0, (int)((long long)p.N * D * H * W * p.K * 2), 0x00020000);
    // tap of this wave (8 waves = the 8 taps) and whether its (tap, parity) block carries a tap at all
    bool live = true;
    if (p.t8cs > 0) live = (w & ~(k0 / p.t8cs) & 7) == 0;
    // ---- per-thread staging constants
    int relg[NG];
    unsigned gpos[2] = {0, 0};  // 16 bits per item: hz | hy << 4 | hx << 8 of its halo voxel (beyond the halo: a dead tail item)
#pragma unroll
    for (int k = 0; k < NG; ++k) {
        const int hv = (t >> 2) + 128 * k;
        const int hz = hv / (HY * HX), rem = hv - hz * (HY * HX), hy = rem / HX, hx = rem - hy * HX;
        relg[k] = (((hz * H + hy) * W + hx) * p.C + c0 + 8 * (t & 3)) * 2;
        gpos[k >> 1] |= ((unsigned)hz | ((unsigned)hy << 4) | ((unsigned)hx << 8)) << (16 * (k & 1));
    }
    // dz item k: voxel (t >> 3) + 64 k = (z k, y t >> 6, x (t >> 3) & 7), channel octet t & 7
    const int dyl = t >> 6, dxl = (t >> 3) & 7;
    const unsigned imask = 0xf00u | (((1u << NG) - 1u) & ~(t < G::GLAST ? 0u : 1u << (NG - 1)));
    const int reld = ((dyl * W + dxl) * p.K + k0 + 8 * (t & 7)) * 2;
    const int ldsg = t * 16;
    const int ldsd = GB + ((t & 7) >> 2) * WG_DZ_HALF + (t >> 3) * 64 + (t & 3) * 16;
    struct tile_t {
        int n, zi, yi, xi;
    };
    struct tinfo {
        int baseg, based;
        unsigned mask;  // bits 0..3: g item k of THIS thread is inside the volume; bits 8..11: dz item k is
    };
    auto info_of = [&](const tile_t& c, bool exists) {
        tinfo r;
        const int z0 = c.zi * TZ, y0 = c.yi * TY, x0 = c.xi * TX;
        r.baseg = ((((c.n * D + z0) * H + y0) * W + x0) * p.C) * 2;  // (forward-looking halo: its origin is the tile's)
        r.based = ((((c.n * D + z0) * H + y0) * W + x0) * p.K) * 2;
        const bool interior = z0 + TZ + 1 <= D && y0 + TY + 1 <= H && x0 + TX + 1 <= W;
        if (!exists) {
            r.mask = 0;
        } else if (interior) {
            r.mask = imask;
        } else {
            const unsigned zn = (unsigned)min(HZ, D - z0), yn = (unsigned)min(HY, H - y0), xn = (unsigned)min(HX, W - x0);
            r.mask = 0;
#pragma unroll
            for (int k = 0; k < NG; ++k) {
                const unsigned gp = gpos[k >> 1] >> (16 * (k & 1));
                const unsigned hz = gp & 0xfu, hy = (gp >> 4) & 0xfu, hx = (gp >> 8) & 0xffu;
                r.mask |= ((hz < zn && hy < yn && hx < xn) ? 1u : 0u) << k;
            }
#pragma unroll
            for (int k = 0; k < ND; ++k) r.mask |= ((k < D - z0 && dyl < H - y0 && dxl < W - x0) ? 1u : 0u) << (8 + k);
        }
        return r;
    };
    auto advance = [&](tile_t& c) {
        if (++c.xi == p.tx) {
            c.xi = 0;
            if (++c.yi == p.ty) {
                c.yi = 0;
                if (++c.zi == p.tz) {
                    c.zi = 0;
                    ++c.n;
                }
            }
        }
    };
    auto load_slot = [&](const tinfo& ti, int j) -> bf16x8 {
        if (j < NG) {
            const int off = ((ti.mask >> j) & 1u) ? relg[j] + ti.baseg : -1;
            return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
        }
        const int k = j - NG;
        const int off = ((ti.mask >> (8 + k)) & 1u) ? reld + ti.based + (k * H * W * p.K) * 2 : -1;
        return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rdz, off, 0, 0));
    };
    auto store_slot = [&](char* buf, int j, const bf16x8& v) {  // (no GroupNorm affine on this path: beyond the volume was loaded as zero)
        if (j < NG) {
            char* dst = buf + ldsg + 8192 * j;
            if (j == NG - 1) dst = t < G::GLAST ? dst : lds + 2 * LDSB;  // (the others' last slot would land in the dz tile: dump slot)
            *reinterpret_cast<bf16x8*>(dst) = v;
        } else {
            *reinterpret_cast<bf16x8*>(buf + ldsd + 4096 * (j - NG)) = v;
        }
    };

    const int first = split * p.per_block, last = min(p.tiles, first + p.per_block);
    tile_t tc;
    {
        int tt = first;
        tc.xi = tt % p.tx;
        tt /= p.tx;
        tc.yi = tt % p.ty;
        tt /= p.ty;
        tc.zi = tt % p.tz;
        tc.n = tt / p.tz;
    }
    bf16x8 ra[NTOT];
    tinfo i1, i2;  // of tiles i + 1 (its items sit in ra and are written during iteration i) and i + 2 (loaded during iteration i)
    if (first < last) {
        const tinfo ti = info_of(tc, true);
#pragma unroll
        for (int j = 0; j < NTOT; ++j) ra[j] = load_slot(ti, j);
#pragma unroll
        for (int j = 0; j < NTOT; ++j) store_slot(lds, j, ra[j]);
        advance(tc);
        i1 = info_of(tc, first + 1 < last);
#pragma unroll
        for (int j = 0; j < NTOT; ++j) ra[j] = load_slot(i1, j);
        advance(tc);
        i2 = info_of(tc, first + 2 < last);
    }
    f32x16 acc[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[hh][e] = 0.f;

    const int g4 = lane >> 4, sidx = lane & 15;
    // (voxel half g4 >> 1 of an MFMA's 16 voxels: the next y row of the 8-wide tile)
    const int lane_off = (HX * (g4 >> 1) + (sidx >> 2)) * 64 + (16 * (g4 & 1) + 4 * (sidx & 3)) * 2;
    const int a_off = lane_off + (((w >> 2) * HY + ((w >> 1) & 1)) * HX + (w & 1)) * 64;
    const int b_off = GB + (8 * (g4 >> 1) + (sidx >> 2)) * 64 + (16 * (g4 & 1) + 4 * (sidx & 3)) * 2;
    for (int tile = first; tile < last; ++tile) {
        __syncthreads();  // buffer (tile - first) & 1 is complete; everyone is done reading the other one
        const char* cur = lds + ((tile - first) & 1) * LDSB;
        char* nxt = lds + ((tile - first + 1) & 1) * LDSB;
        tinfo i3;
        constexpr int AD = 2;
        auto a_addr = [&](int rw) { return a_off + ((rw >> 2) * HY + 2 * (rw & 3)) * HX * 64; };
        auto b_addr = [&](int rw, int hh) { return b_off + hh * WG_DZ_HALF + rw * 16 * 64; };
        bf16x8 af[AD + 1] = {}, bfr[2][2] = {};
        if (live) {
#pragma unroll
            for (int s_ = 0; s_ < AD; ++s_) af[s_] = tr_frag(cur + a_addr(s_));
            bfr[0][0] = tr_frag(cur + b_addr(0, 0));
            bfr[0][1] = tr_frag(cur + b_addr(0, 1));
        }
#pragma unroll
        for (int rw = 0; rw < 16; ++rw) {
            if (live) {
                if (rw + AD < 16) af[(rw + AD) % (AD + 1)] = tr_frag(cur + a_addr(rw + AD));
                if (rw + 1 < 16) {
                    bfr[(rw + 1) & 1][0] = tr_frag(cur + b_addr(rw + 1, 0));
                    bfr[(rw + 1) & 1][1] = tr_frag(cur + b_addr(rw + 1, 1));
                }
            }
            // ---- staging: at even steps the register of item j hands tile i + 1's item to LDS and takes tile i + 2's
            if ((rw & 1) == 0) {
                const int j = rw >> 1;
                store_slot(nxt, j, ra[j]);
                ra[j] = load_slot(i2, j);
            }
            if (rw == 9) {  // the tile after those: coordinates and masks under this tile's MFMAs
                advance(tc);
                i3 = info_of(tc, tile + 3 < last);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (live) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[rw % (AD + 1)], bfr[rw & 1][0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[rw % (AD + 1)], bfr[rw & 1][1], acc[1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        i1 = i2;
        i2 = i3;
    }
    // ---- partial sums: ws[split][pair][tap 8][ci 32][co 64]; D layout: column = lane & 31 (co), row = ci
    if (live) {
        float* dst = p.ws + (((size_t)split * P + pair) * 8 + w) * 2048;
        const int col = lane & 31, half = lane >> 5;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ci = (e & 3) + 8 * (e >> 2) + 4 * half;
                dst[ci * 64 + hh * 32 + col] = acc[hh][e];
            }
    }
}

// sum over the splits of one workspace element in split order (double accumulation), eight loads in flight: the plain loop is a chain of
// dependent 4-byte loads as far as the compiler is concerned (~25 us for 56 MB of partial sums, rocprofv3 on config 4); same order, same bits
__device__ __forceinline__ double u3d_sum_splits(const float* __restrict__ p, size_t stride, int S) {
    double sum = 0.0;
    int s = 0;
    for (; s + 8 <= S; s += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p[(size_t)(s + j) * stride];
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += (double)v[j];
    }
    for (; s < S; ++s) sum += (double)p[(size_t)s * stride];
    return sum;
}

// ... four consecutive workspace elements at once (16-byte loads, round 6): the 4-byte form kept 8 x 4 bytes in flight per thread and read
// config 4's 56 MB of partial sums per layer at 1.9 TB/s (30 us per launch, 10 + 8 launches per step); same order per element, same bits
struct f64x4s {
    double v[4];
};
__device__ __forceinline__ f64x4s u3d_sum_splits4(const float* __restrict__ p, size_t stride, int S) {
    f64x4s sum = {{0.0, 0.0, 0.0, 0.0}};
    int s = 0;
    for (; s + 8 <= S; s += 8) {
        f32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const f32x4*>(p + (size_t)(s + j) * stride);
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) sum.v[e] += (double)v[j][e];
    }
    for (; s < S; ++s) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p + (size_t)s * stride);
#pragma unroll
        for (int e = 0; e < 4; ++e) sum.v[e] += (double)v[e];
    }
    return sum;
}

// dw[co][ci][tap] = sum over splits, fixed order.  One block per (pair, input channel): the 27 x 64 partial sums of that row are
// read coalesced over the output channel (256-byte runs), transposed through LDS, and written as 64 runs of 27 consecutive
// floats (the reference layout has the tap innermost) — a thread-per-element version writes 4 bytes every 27*Cin floats and
// costs more than the GEMM at 1024 channels.
// (has_job, round 6: the grid carries ONE extra block — block 0 — that runs the GroupNorm-backward reduction of the layer's input, as in
// wgrad_reduce_kernel of csrc/u3d_conv.hip: u3d_conv3d_wgrad_bf16_job)
__global__ __launch_bounds__(256) void wgrad_bf16_reduce_kernel(const float* __restrict__ ws, int S, int C, int K, float* __restrict__ dw,
                                                                int has_job, u3d_gn_bwd_job_t job) {
    __shared__ float tile[64][28];
    if (has_job && blockIdx.x == 0) {
        extern __shared__ double shb[];
        u3d_gn_bwd_finalize_body(job.gstats_lo, job.mean_rstd, job.gamma, job.N, job.C0 + job.C1, job.G, job.count, 1, 1, job.dgamma,
                                 job.dbeta, job.coef, job.gstats_hi, job.C0, job.hi_scale, job.coef_hi, shb, job.reps_lo, job.reps_hi);
        return;
    }
    const int bx = (int)blockIdx.x - (has_job ? 1 : 0);
    const int pco = (K + 63) >> 6, P = (C >> 5) * pco;  // (K % 64 == 32: the last block's upper 32 columns are zeros, not written)
    const int pair = bx >> 5, cil = bx & 31;
    const int cib = pair / pco, cob = pair - cib * pco;
    const int t = threadIdx.x;
    const size_t split_stride = (size_t)P * 27 * 2048;
    for (int i = t; i < 27 * 16; i += 256) {  // (tap, quad of output channels): 16-byte loads
        const int tap = i >> 4, co = (i & 15) * 4;
        if (cob * 64 + co >= K) continue;  // K % 64 == 32: those workspace columns were never written (and are never stored)
        const size_t off = ((size_t)pair * 27 + tap) * 2048 + cil * 64 + co;
        const f64x4s sum = u3d_sum_splits4(ws + off, split_stride, S);
#pragma unroll
        for (int e = 0; e < 4; ++e) tile[co + e][tap] = (float)sum.v[e];
    }
    __syncthreads();
    const int ci = cib * 32 + cil;
    for (int i = t; i < 27 * 64; i += 256) {
        const int co = i / 27, tap = i - co * 27;
        if (cob * 64 + co < K) dw[((size_t)(cob * 64 + co) * C + ci) * 27 + tap] = tile[co][tap];
    }
}

// few (pair, channel) rows but many splits (64-channel layers at full resolution): one thread per output element instead, so
// that the sum over hundreds of splits is spread over the whole chip (the scattered 4-byte writes are few there)
__global__ __launch_bounds__(256) void wgrad_bf16_reduce_flat_kernel(const float* __restrict__ ws, int S, int C, int K, float* __restrict__ dw,
                                                                     int has_job, u3d_gn_bwd_job_t job) {
    if (has_job && blockIdx.x == 0) {
        extern __shared__ double shb[];
        u3d_gn_bwd_finalize_body(job.gstats_lo, job.mean_rstd, job.gamma, job.N, job.C0 + job.C1, job.G, job.count, 1, 1, job.dgamma,
                                 job.dbeta, job.coef, job.gstats_hi, job.C0, job.hi_scale, job.coef_hi, shb, job.reps_lo, job.reps_hi);
        return;
    }
    const long long bx = (long long)blockIdx.x - (has_job ? 1 : 0), nb = (long long)gridDim.x - (has_job ? 1 : 0);
    const int pco = (K + 63) >> 6, P = (C >> 5) * pco;
    const int K4 = K >> 2;  // (K % 32 == 0)
    const long long total = (long long)C * K4 * 27;
    for (long long i = bx * blockDim.x + threadIdx.x; i < total; i += nb * blockDim.x) {
        const int co = (int)(i % K4) * 4;  // read-coalesced order: quads of co fastest, then ci, then tap
        long long r = i / K4;
        const int ci = (int)(r % C);
        const int tap = (int)(r / C);
        const int pair = (ci >> 5) * pco + (co >> 6);
        const size_t off = ((size_t)pair * 27 + tap) * 2048 + (ci & 31) * 64 + (co & 63);
        const f64x4s sum = u3d_sum_splits4(ws + off, (size_t)P * 27 * 2048, S);
#pragma unroll
        for (int e = 0; e < 4; ++e) dw[((size_t)(co + e) * C + ci) * 27 + tap] = (float)sum.v[e];
    }
}

struct wgrad_plan {
    int tz, ty, tx, tiles, per_block, S, P;
};
wgrad_plan plan_wgrad(int N, int D, int H, int W, int C, int K, bool x8 = false) {  // x8: the 4 x 8 x 8 tile of conv3d_wgrad_b16v2_kernel<true>
    wgrad_plan q;
    q.tz = x8 ? (D + 3) / 4 : (D + WG_TZ - 1) / WG_TZ;
    q.ty = (H + WG_TY - 1) / WG_TY;
    q.tx = x8 ? (W + 7) / 8 : (W + WG_TX - 1) / WG_TX;
    q.tiles = N * q.tz * q.ty * q.tx;
    q.P = (C / 32) * ((K + 63) / 64);
    // every block writes its 27 x 32 x 64 partial sums (221 KB) and the reduction reads them back: the block count is the
    // split traffic.  Measured on config 4 (profiles/r03e): 1024 blocks 7.2 ms per step, 512 5.75, 256 (one per CU — the kernel
    // double-buffers inside the block) 5.13
    const int blocks = g_u3d_tune[8] > 0 ? g_u3d_tune[8] : 256;
    int target = blocks / q.P;
    if (target < 1) target = 1;
    q.per_block = (q.tiles + target - 1) / target;
    if (q.per_block < 1) q.per_block = 1;
    q.S = (q.tiles + q.per_block - 1) / q.per_block;
    return q;
}

}  // namespace

// (K % 64 == 32 — e.g. the 32-output-channel layers of config 2's model — runs the 64-column blocks with the upper half's dz columns read
// as zero: twice the MFMAs those 32 columns need, still several times the fp32 kernel's rate)
extern "C" int u3d_conv3d_wgrad_bf16_supported(int C, int K) { return (C > 0 && K > 0 && C % 32 == 0 && K % 32 == 0) ? 1 : 0; }

extern "C" long long u3d_wgrad_bf16_workspace_floats(int N, int D, int H, int W, int C, int K) {
    if (!u3d_conv3d_wgrad_bf16_supported(C, K) || N <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    const wgrad_plan q = plan_wgrad(N, D, H, W, C, K), q8 = plan_wgrad(N, D, H, W, C, K, true);  // (either tile shape of the bf16-storage kernel)
    return (long long)(q.S > q8.S ? q.S : q8.S) * q.P * 27 * 2048;
}

// Which kernel u3d_conv3d_wgrad_bf16_b16 runs for a shape: 0 the round-3 kernel (u3d_set_tuning key 7 = 1, or tensors beyond 2 GiB),
// 16 / 8: conv3d_wgrad_b16v2_kernel on 2 x 8 x 16 / 4 x 8 x 8 tiles — the shape that wastes fewer voxels on ragged extents, the
// 8-wide one on a tie (smaller halo; key 7 = 16 / 8 force one).  The 16-wide variant is bit-identical to the fp32-storage kernel fed
// with the same values; the 8-wide one sums the voxels in another order.
static int wgrad_b16_variant(int N, int D, int H, int W, int C, int K) {
    const long long vox = (long long)N * D * H * W;
    if (g_u3d_tune[7] == 1 || K % 64 != 0 || vox * (C > K ? C : K) * 2 >= (1ll << 31)) return 0;  // (buffer offsets are 32-bit; whole 64-column blocks)
    if (g_u3d_tune[7] == 16 || g_u3d_tune[7] == 8) return g_u3d_tune[7];
    const long long v16 = (long long)((D + 1) / 2) * ((H + 7) / 8) * ((W + 15) / 16), v8 = (long long)((D + 3) / 4) * ((H + 7) / 8) * ((W + 7) / 8);
    return v8 <= v16 ? 8 : 16;  // (tiles of 256 voxels each)
}
extern "C" int u3d_conv3d_wgrad_bf16_b16_variant(int N, int D, int H, int W, int C, int K) {
    if (!u3d_conv3d_wgrad_bf16_supported(C, K) || N <= 0 || D <= 0 || H <= 0 || W <= 0) return -1;
    return wgrad_b16_variant(N, D, H, W, C, K);
}

static int conv3d_wgrad_bf16_impl(int device, u3d_stream_t stream, const float* x, const float* affine, const float* dz, float* dw,
                                  int N, int D, int H, int W, int C, int K, float* workspace, long long workspace_floats, int b16,
                                  const u3d_gn_bwd_job_t* job = nullptr);
// does the launch that writes dw for this shape have a reduce pass (one split: the round-4 kernels write dw themselves)?
static bool wgrad_bf16_has_reduce(int N, int D, int H, int W, int C, int K, int b16, const float* dw) {
    const int variant = b16 ? wgrad_b16_variant(N, D, H, W, C, K) : 0;
    const wgrad_plan q = plan_wgrad(N, D, H, W, C, K, variant == 8);
    return !((variant == 8 || variant == 16) && q.S == 1 && g_u3d_tune[17] != 1 && ((uintptr_t)dw & 15) == 0);
}
static size_t wgrad_bf16_job_lds_bytes(int N, int C, int G) { return sizeof(double) * (4 * (size_t)N * C + 2 * (size_t)N * G); }

// The GroupNorm-backward reduction of the layer's INPUT (u3d_gn_bwd_job_t, see u3d_conv3d_wgrad_job) as one extra block of the launch that
// reduces the weight gradient's splits: config 4 ran 18 single-block finalize launches of ~7 us per step behind these.  Only shapes whose
// weight gradient HAS a reduce launch can carry it (one split — the 1024-channel level — writes dw from the main kernel): ask first.
extern "C" int u3d_conv3d_wgrad_bf16_job_supported(int N, int D, int H, int W, int C, int K, int b16, const float* dw, int jobN, int jobC,
                                                   int jobG) {
    if (!u3d_conv3d_wgrad_bf16_supported(C, K) || N <= 0 || D <= 0 || H <= 0 || W <= 0 || jobN <= 0 || jobC <= 0 || jobG <= 0 ||
        jobC % jobG != 0)
        return 0;
    // (the reduce kernel's own 7 KB of static LDS sits beside the job's tables)
    if (wgrad_bf16_job_lds_bytes(jobN, jobC, jobG) + 8 * 1024 > 64 * 1024) return 0;
    return wgrad_bf16_has_reduce(N, D, H, W, C, K, b16, dw) ? 1 : 0;
}
extern "C" int u3d_conv3d_wgrad_bf16_job(int device, u3d_stream_t stream, const float* x, const float* affine, const float* dz, float* dw,
                                         int N, int D, int H, int W, int C, int K, float* workspace, long long workspace_floats,
                                         const u3d_gn_bwd_job_t* job) {
    return conv3d_wgrad_bf16_impl(device, stream, x, affine, dz, dw, N, D, H, W, C, K, workspace, workspace_floats, 0, job);
}
extern "C" int u3d_conv3d_wgrad_bf16_b16_job(int device, u3d_stream_t stream, const void* x, const float* affine, const void* dz, float* dw,
                                             int N, int D, int H, int W, int C, int K, float* workspace, long long workspace_floats,
                                             const u3d_gn_bwd_job_t* job) {
    return conv3d_wgrad_bf16_impl(device, stream, (const float*)x, affine, (const float*)dz, dw, N, D, H, W, C, K, workspace,
                                  workspace_floats, 1, job);
}

extern "C" int u3d_conv3d_wgrad_bf16(int device, u3d_stream_t stream, const float* x, const float* affine, const float* dz,
                                     float* dw, int N, int D, int H, int W, int C, int K, float* workspace,
                                     long long workspace_floats) {
    return conv3d_wgrad_bf16_impl(device, stream, x, affine, dz, dw, N, D, H, W, C, K, workspace, workspace_floats, 0);
}

// bf16 activation storage: x and dz are bf16 tensors (half the bytes of a kernel whose operand re-reads make it HBM-bound)
extern "C" int u3d_conv3d_wgrad_bf16_b16(int device, u3d_stream_t stream, const void* x, const float* affine, const void* dz, float* dw,
                                         int N, int D, int H, int W, int C, int K, float* workspace, long long workspace_floats) {
    return conv3d_wgrad_bf16_impl(device, stream, (const float*)x, affine, (const float*)dz, dw, N, D, H, W, C, K, workspace,
                                  workspace_floats, 1);
}

static int conv3d_wgrad_bf16_impl(int device, u3d_stream_t stream, const float* x, const float* affine, const float* dz, float* dw,
                                  int N, int D, int H, int W, int C, int K, float* workspace, long long workspace_floats, int b16,
                                  const u3d_gn_bwd_job_t* job) {
    U3D_ENTER(device);
    if (job) {
        U3D_REQUIRE(job->gstats_lo && job->mean_rstd && job->gamma && job->dgamma && job->dbeta && job->coef && job->C0 > 0 &&
                        job->C1 >= 0 && (job->C1 == 0) == (job->gstats_hi == nullptr) && (job->coef_hi == nullptr || job->C1 > 0) &&
                        job->reps_lo >= 0 && job->reps_lo <= 64 && job->reps_hi >= 0 && job->reps_hi <= 64,
                    "u3d_conv3d_wgrad_bf16_job: bad job");
        U3D_REQUIRE(u3d_conv3d_wgrad_bf16_job_supported(N, D, H, W, C, K, b16, dw, job->N, job->C0 + job->C1, job->G) == 1,
                    "u3d_conv3d_wgrad_bf16_job: this shape's weight gradient has no reduce launch to carry the job (or %d x %d channels in "
                    "%d groups do not fit one block's LDS): ask u3d_conv3d_wgrad_bf16_job_supported",
                    job->N, job->C0 + job->C1, job->G);
    }
    U3D_REQUIRE(x && dz && dw && N > 0 && D > 0 && H > 0 && W > 0, "u3d_conv3d_wgrad_bf16: bad argument");
    U3D_REQUIRE(u3d_conv3d_wgrad_bf16_supported(C, K), "u3d_conv3d_wgrad_bf16: needs Cin %% 32 == 0 and Cout %% 32 == 0 (got %d, %d)", C, K);
    U3D_REQUIRE((((uintptr_t)x | (uintptr_t)dz | (uintptr_t)affine) & 15) == 0, "u3d_conv3d_wgrad_bf16: 16-byte alignment");
    const int variant = b16 ? wgrad_b16_variant(N, D, H, W, C, K) : 0;
    const wgrad_plan q = plan_wgrad(N, D, H, W, C, K, variant == 8);
    const long long need = (long long)q.S * q.P * 27 * 2048;
    if (!workspace || workspace_floats < need)
        return u3d_set_err(U3D_EWORKSPACE, "u3d_conv3d_wgrad_bf16: workspace of %lld floats needed, %lld given", need, workspace_floats);
    bf16_wgrad_params p{x, affine, dz, workspace, N, D, H, W, C, K, 1, q.tz, q.ty, q.tx, q.tiles, q.per_block, (K + 63) / 64,
                        g_u3d_tune[9] == 1 ? 0 : 1};
    // one split (every block owns all tiles of its (32 input, 64 output channels) pair: config 4's 1024-channel level): the round-4 kernel
    // writes dw itself (u3d_set_tuning key 17 = 1: through the workspace and the reduction, as before; same bits)
    const bool direct = (variant == 8 || variant == 16) && q.S == 1 && g_u3d_tune[17] != 1 && ((uintptr_t)dw & 15) == 0;
    p.dw = direct ? dw : nullptr;
#ifdef U3D_WG_TRACE
    constexpr int v2_extra = 8 * 12 * 14 * 4;
#else
    constexpr int v2_extra = 0;
#endif
    if (variant == 8) {
        constexpr int v2_lds = wg2_geom<true>::LDS_TOTAL + v2_extra;
        U3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_wgrad_b16v2_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, v2_lds));
        hipLaunchKernelGGL(conv3d_wgrad_b16v2_kernel<true>, dim3((unsigned)(q.S * q.P)), dim3(512), v2_lds, (hipStream_t)stream, p);
    } else if (variant == 16) {
        constexpr int v2_lds = wg2_geom<false>::LDS_TOTAL + v2_extra;
        U3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_wgrad_b16v2_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, v2_lds));
        hipLaunchKernelGGL(conv3d_wgrad_b16v2_kernel<false>, dim3((unsigned)(q.S * q.P)), dim3(512), v2_lds, (hipStream_t)stream, p);
    } else if (b16) {
        U3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_wgrad_bf16_kernel<3, __bf16>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 2 * wg_geom<3>::LDS));
        hipLaunchKernelGGL((conv3d_wgrad_bf16_kernel<3, __bf16>), dim3((unsigned)(q.S * q.P)), dim3(512), 2 * wg_geom<3>::LDS,
                           (hipStream_t)stream, p);
    } else {
        U3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_wgrad_bf16_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    2 * wg_geom<3>::LDS));
        hipLaunchKernelGGL(conv3d_wgrad_bf16_kernel<3>, dim3((unsigned)(q.S * q.P)), dim3(512), 2 * wg_geom<3>::LDS, (hipStream_t)stream, p);
    }
    U3D_LAUNCH_CHECK();
    if (direct) return 0;
    u3d_gn_bwd_job_t jb = {};
    size_t job_lds = 0;
    if (job) {
        jb = *job;
        if (jb.reps_lo < 1) jb.reps_lo = 1;
        if (jb.reps_hi < 1) jb.reps_hi = 1;
        job_lds = wgrad_bf16_job_lds_bytes(jb.N, jb.C0 + jb.C1, jb.G);
    }
    const unsigned jx = job ? 1u : 0u;
    if (q.P * 32 >= 1024) {
        hipLaunchKernelGGL(wgrad_bf16_reduce_kernel, dim3((unsigned)(q.P * 32) + jx), dim3(256), job_lds, (hipStream_t)stream, workspace, q.S, C,
                           K, dw, (int)jx, jb);
    } else {
        long long rb = ((long long)C * (K / 4) * 27 + 255) / 256;  // one thread per (tap, input channel, quad of output channels)
        if (rb > 8192) rb = 8192;
        hipLaunchKernelGGL(wgrad_bf16_reduce_flat_kernel, dim3((unsigned)rb + jx), dim3(256), job_lds, (hipStream_t)stream, workspace, q.S, C, K,
                           dw, (int)jx, jb);
    }
    U3D_LAUNCH_CHECK();
    return 0;
}


// =====================================================================================================================
// ConvTranspose3d(k=3, stride=2, padding=1, bias=False) (buildingblocks.py:653-662) on the bf16 kernels, in SPACE-TO-DEPTH form.
// t[o] = sum_j sum_s [o = 2j - 1 + s] x[j] w[s]: an output voxel o = 2i + p (parity p per dimension) takes, per dimension,
//   p = 0: tap s = 1 of input i;          p = 1: tap s = 2 of input i and tap s = 0 of input i + 1.
// Store the (2n-1)^3 output as T8 (N, D1, H1, W1, 8*Cs) with T8[i][p*Cs + co] = t[2i + p][co] (p = pz*4 + py*2 + px; entries with
// 2i + p = 2n - 1 do not exist and are never read).  Then
//   forward        T8 = conv2x2x2(x; taps a in {0,1}^3 read x[i + a])            Cl -> 8*Cs channels
//   data gradient  dx = conv2x2x2(dT8; taps read dT8[j - a])                     8*Cs -> Cl channels (+ ReLU mask of x)
//   weight grad    dV[a][ci][p,co] = sum_i x[i + a][ci] dT8[i][p,co]             the 2x2x2 weight gradient
// with the weight V[a][ci][p*Cs + co] = w[ci][co][s(a,p)] where per dimension s(0,0) = 1, s(0,1) = 2, s(1,1) = 0 and (a,p) = (1,0)
// is structurally zero: 27 of the 64 (a,p) combinations carry the 27 taps — 64/27 = 2.4x the minimal multiply-adds, on the
// low-res grid, at bf16 MFMA rates, with the SAME kernels as the 3x3x3 convolutions (template KS = 2).  The nearest resize +
// join kernels read / write the T8 layout directly (u3d_nearest_add_fwd_t8 / u3d_nearest_sum_bwd_t8, csrc/u3d_res.hip).
namespace {

__device__ __host__ inline int t8_tap_of(int a, int p) {  // per dimension: the 3-tap index s(a,p), or -1
    return a == 0 ? (p == 0 ? 1 : 2) : (p == 1 ? 0 : -1);
}

// mode 0 (forward):       B[k = ci][col = p*Cs + co] for tap a      = w[ci][co][s(a,p)]
// mode 1 (data gradient): B[k = p*Cs + co][col = ci] for tap tau    = w[ci][co][s(1 - tau, p)]   (tap tau reads dT8[j + tau - 1])
// w: (Cl, Cs, 3,3,3) fp32, the nn.ConvTranspose3d layout.  Image [chunk][tap 8][n-tile][lane][8] like pack_weights_bf16_kernel.
__global__ void pack_convtr_t8_kernel(const float* __restrict__ w, int Cl, int Cs, int mode, __bf16* __restrict__ out, long long total) {
    const int Kc = mode == 0 ? Cl : 8 * Cs, Nc = mode == 0 ? 8 * Cs : Cl;
    const int ntiles = Nc >> 5;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(i & 7);
        long long rest = i >> 3;
        const int l = (int)(rest & 63);
        rest >>= 6;
        const int nt = (int)(rest % ntiles);
        rest /= ntiles;
        const int tap = (int)(rest & 7);
        const int c = (int)(rest >> 3);
        const int k = c * 16 + 8 * (l >> 5) + e, col = nt * 32 + (l & 31);
        float v = 0.f;
        if (k < Kc && col < Nc) {
            const int ci = mode == 0 ? k : col, pc = mode == 0 ? col : k;
            const int pp = pc / Cs, co = pc - pp * Cs;
            int sidx = 0;
            bool ok = true;
#pragma unroll
            for (int d = 0; d < 3; ++d) {  // d = 0: z (bit 2), 1: y, 2: x
                const int bit = 2 - d;
                const int tb = (tap >> bit) & 1, pb = (pp >> bit) & 1;
                const int sd = t8_tap_of(mode == 0 ? tb : 1 - tb, pb);
                ok = ok && sd >= 0;
                sidx = sidx * 3 + (sd < 0 ? 0 : sd);
            }
            if (ok) v = w[((size_t)ci * Cs + co) * 27 + sidx];
        }
        out[i] = (__bf16)v;
    }
}

// dW[ci][co][s] = sum over splits of D[a][ci][p*Cs + co] with (a,p) = the one combination that carries tap s (fixed order).
// One block per (input channel, run of 64 output channels): reads coalesced over co, writes runs of 27 consecutive floats.
__global__ __launch_bounds__(256) void wgrad_t8_reduce_kernel(const float* __restrict__ ws, int S, int Cl, int Cs, float* __restrict__ dw) {
    __shared__ float tile[64][28];
    const int Kp = 8 * Cs, pco = Kp >> 6, P = (Cl >> 5) * pco;
    const int cblocks = (Cs + 63) >> 6;
    const int ci = blockIdx.x / cblocks, co0 = (blockIdx.x - ci * cblocks) * 64;
    const int t = threadIdx.x;
    const size_t split_stride = (size_t)P * 8 * 2048;
    for (int i = t; i < 27 * 64; i += 256) {
        const int s = i >> 6, col = i & 63, co = co0 + col;
        float v = 0.f;
        if (co < Cs) {
            int tap = 0, pp = 0;
            const int sd[3] = {s / 9, (s / 3) % 3, s % 3};
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int a = sd[d] == 0 ? 1 : 0, pb = sd[d] == 1 ? 0 : 1;  // s = 0 -> (1,1); 1 -> (0,0); 2 -> (0,1)
                tap = tap * 2 + a;
                pp = pp * 2 + pb;
            }
            const int kp = pp * Cs + co;
            const int pair = (ci >> 5) * pco + (kp >> 6);
            const size_t off = ((size_t)pair * 8 + tap) * 2048 + (ci & 31) * 64 + (kp & 63);
            v = (float)u3d_sum_splits(ws + off, split_stride, S);
        }
        tile[col][s] = v;
    }
    __syncthreads();
    for (int i = t; i < 27 * 64; i += 256) {
        const int col = i / 27, s = i - col * 27;
        if (co0 + col < Cs) dw[((size_t)ci * Cs + co0 + col) * 27 + s] = tile[col][s];
    }
}

int launch_t8_conv(const bf16_conv_params& p0, hipStream_t s) {
    bf16_conv_params p = p0;
    const bool nt2 = p.K % 64 == 0;
    // (8-plane tiles measured no gain on these 2x2x2 kernels: 0.77 vs 0.80 ms for config 4's data gradients, forward equal)
    if (p.b16) return nt2 ? launch_bf16<2, 1, 2, 0, __bf16>(p, s) : launch_bf16<1, 1, 2, 0, __bf16>(p, s);
    return nt2 ? launch_bf16<2, 1, 2>(p, s) : launch_bf16<1, 1, 2>(p, s);
}

}  // namespace

extern "C" int u3d_convtr3d_t8_supported(int Cl, int Cs) { return (Cl > 0 && Cs > 0 && Cl % 32 == 0 && Cs % 8 == 0) ? 1 : 0; }

extern "C" long long u3d_convtr3d_t8_packed_elems(int Cl, int Cs, int mode) {
    if (!u3d_convtr3d_t8_supported(Cl, Cs)) return 0;
    const int Kc = mode == 0 ? Cl : 8 * Cs, Nc = mode == 0 ? 8 * Cs : Cl;
    return ((long long)(Kc / 16) * 8 + 6) * (Nc / 32) * 64 * 8;  // + BDIST taps of tail padding (prefetched, never used)
}

extern "C" int u3d_pack_convtr3d_t8(int device, u3d_stream_t stream, const float* w, int Cl, int Cs, int mode, void* packed) {
    U3D_ENTER(device);
    const long long total = u3d_convtr3d_t8_packed_elems(Cl, Cs, mode);
    U3D_REQUIRE(w && packed && (mode == 0 || mode == 1) && total > 0, "u3d_pack_convtr3d_t8: needs Cin %% 32 == 0, Cout %% 8 == 0 (%d, %d)", Cl, Cs);
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_convtr_t8_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, Cl, Cs, mode,
                       reinterpret_cast<__bf16*>(packed), total);
    U3D_LAUNCH_CHECK();
    return 0;
}

static int convtr3d_fwd_t8_impl(int device, u3d_stream_t stream, const float* x, const void* packed, float* t8, int N, int D1, int H1,
                                int W1, int Cl, int Cs, int b16, float* workspace = nullptr, long long workspace_floats = 0) {
    U3D_ENTER(device);
    U3D_REQUIRE(x && packed && t8 && N > 0 && D1 > 0 && H1 > 0 && W1 > 0 && u3d_convtr3d_t8_supported(Cl, Cs), "u3d_convtr3d_fwd_t8: bad argument");
    bf16_conv_params p{x, nullptr, reinterpret_cast<const bf16x8*>(packed), t8, nullptr, nullptr, nullptr, nullptr,
                       N, D1, H1, W1, Cl, 8 * Cs, 0, 0, 0, 0, 0, nullptr, 1, nullptr};
    p.b16 = b16;
    p.t8mode = g_u3d_tune[10] == 2 ? 0 : 1;  // (key 10 = 2: A/B without the zero-block skipping)
    p.t8cs = Cs;
    const int kf = b16 ? bf16_flat_ksplit(N, D1, H1, W1, Cl, 8 * Cs, false) : 0;
    if (kf > 1 && workspace && workspace_floats >= (long long)kf * N * D1 * H1 * W1 * 8 * Cs) {  // the flat 5 x 10 x 10 tile (see u3d_conv3d_bf16_ex_b16)
        p.ksplit = kf;
        p.ws = workspace;
        return launch_bf16<2, 3, 2, 0, __bf16>(p, (hipStream_t)stream);
    }
    return launch_t8_conv(p, (hipStream_t)stream);
}

// bf16 storage at the bottom of the U (config 4: 1024 -> 512 channels on 5 x 10 x 10): the flat tile with a split channel reduction needs
// scratch; 0 floats where the plain entry point's plan stays
extern "C" long long u3d_convtr3d_fwd_t8_workspace_floats(int N, int D1, int H1, int W1, int Cl, int Cs) {
    if (!u3d_convtr3d_t8_supported(Cl, Cs) || N <= 0 || D1 <= 0 || H1 <= 0 || W1 <= 0) return 0;
    const int kf = bf16_flat_ksplit(N, D1, H1, W1, Cl, 8 * Cs, false);
    return kf > 1 ? (long long)kf * N * D1 * H1 * W1 * 8 * Cs : 0;
}

extern "C" int u3d_convtr3d_fwd_t8_b16_ex(int device, u3d_stream_t stream, const void* x, const void* packed, void* t8, int N, int D1,
                                          int H1, int W1, int Cl, int Cs, float* workspace, long long workspace_floats) {
    return convtr3d_fwd_t8_impl(device, stream, (const float*)x, packed, (float*)t8, N, D1, H1, W1, Cl, Cs, 1, workspace, workspace_floats);
}

extern "C" int u3d_convtr3d_fwd_t8(int device, u3d_stream_t stream, const float* x, const void* packed, float* t8, int N, int D1,
                                   int H1, int W1, int Cl, int Cs) {
    return convtr3d_fwd_t8_impl(device, stream, x, packed, t8, N, D1, H1, W1, Cl, Cs, 0);
}

extern "C" int u3d_convtr3d_fwd_t8_b16(int device, u3d_stream_t stream, const void* x, const void* packed, void* t8, int N, int D1,
                                       int H1, int W1, int Cl, int Cs) {
    return convtr3d_fwd_t8_impl(device, stream, (const float*)x, packed, (float*)t8, N, D1, H1, W1, Cl, Cs, 1);
}

// The data gradient contracts over 8*Cs channels (hundreds of 16-channel chunks at the bottom of the U) into Cl outputs on the low-res
// grid: at config 4's two bottom levels that is 128 / 216 blocks of 256 / 128 serial chunks each (0.37 / 0.19 ms).  The `_ex` entry
// points take the split-K scratch of u3d_conv3d_bf16_ex (same rule, bf16_ksplit; same fixed-order reduction kernel, which owns the
// ReLU mask); without a workspace they run unsplit, like the plain entry points.
extern "C" long long u3d_convtr3d_dgrad_t8_workspace_floats(int N, int D1, int H1, int W1, int Cl, int Cs) {
    if (!u3d_convtr3d_t8_supported(Cl, Cs) || N <= 0 || D1 <= 0 || H1 <= 0 || W1 <= 0) return 0;
    int ks = bf16_ksplit(N, D1, H1, W1, 8 * Cs, Cl);
    const int kf = bf16_flat_ksplit(N, D1, H1, W1, 8 * Cs, Cl);  // (bf16 storage may take the flat tile's plan: room for either)
    if (kf > ks) ks = kf;
    return ks > 1 ? (long long)ks * N * D1 * H1 * W1 * Cl : 0;
}

static int convtr3d_dgrad_t8_impl(int device, u3d_stream_t stream, const float* dt8, const void* packed, const float* x_mask, float* dx,
                                  int N, int D1, int H1, int W1, int Cl, int Cs, int b16, float* workspace, long long workspace_floats) {
    U3D_ENTER(device);
    U3D_REQUIRE(dt8 && packed && dx && N > 0 && D1 > 0 && H1 > 0 && W1 > 0 && u3d_convtr3d_t8_supported(Cl, Cs), "u3d_convtr3d_dgrad_t8: bad argument");
    bf16_conv_params p{dt8, nullptr, reinterpret_cast<const bf16x8*>(packed), dx, nullptr, nullptr, nullptr, nullptr,
                       N, D1, H1, W1, 8 * Cs, Cl, 0, 0, 0, 0, 1, x_mask, 1, nullptr};
    p.b16 = b16;
    p.t8mode = g_u3d_tune[10] == 2 ? 0 : 2;
    p.t8cs = Cs;
    const int ks = bf16_ksplit(N, D1, H1, W1, 8 * Cs, Cl);
    const int kf = b16 ? bf16_flat_ksplit(N, D1, H1, W1, 8 * Cs, Cl) : 0;
    if (kf > 1 && workspace && workspace_floats >= (long long)kf * N * D1 * H1 * W1 * Cl) {  // the flat 5 x 10 x 10 tile (see u3d_conv3d_bf16_ex_b16)
        p.ksplit = kf;
        p.ws = workspace;
        return launch_bf16<2, 3, 2, 0, __bf16>(p, (hipStream_t)stream);
    }
    if (ks > 1 && workspace && workspace_floats >= (long long)ks * N * D1 * H1 * W1 * Cl) {
        p.ksplit = ks;
        p.ws = workspace;
    }
    return launch_t8_conv(p, (hipStream_t)stream);
}

extern "C" int u3d_convtr3d_dgrad_t8(int device, u3d_stream_t stream, const float* dt8, const void* packed, const float* x_mask,
                                     float* dx, int N, int D1, int H1, int W1, int Cl, int Cs) {
    return convtr3d_dgrad_t8_impl(device, stream, dt8, packed, x_mask, dx, N, D1, H1, W1, Cl, Cs, 0, nullptr, 0);
}

extern "C" int u3d_convtr3d_dgrad_t8_b16(int device, u3d_stream_t stream, const void* dt8, const void* packed, const void* x_mask,
                                         void* dx, int N, int D1, int H1, int W1, int Cl, int Cs) {
    return convtr3d_dgrad_t8_impl(device, stream, (const float*)dt8, packed, (const float*)x_mask, (float*)dx, N, D1, H1, W1, Cl, Cs, 1, nullptr, 0);
}

extern "C" int u3d_convtr3d_dgrad_t8_ex(int device, u3d_stream_t stream, const float* dt8, const void* packed, const float* x_mask,
                                        float* dx, int N, int D1, int H1, int W1, int Cl, int Cs, float* workspace, long long workspace_floats) {
    return convtr3d_dgrad_t8_impl(device, stream, dt8, packed, x_mask, dx, N, D1, H1, W1, Cl, Cs, 0, workspace, workspace_floats);
}

extern "C" int u3d_convtr3d_dgrad_t8_b16_ex(int device, u3d_stream_t stream, const void* dt8, const void* packed, const void* x_mask,
                                            void* dx, int N, int D1, int H1, int W1, int Cl, int Cs, float* workspace,
                                            long long workspace_floats) {
    return convtr3d_dgrad_t8_impl(device, stream, (const float*)dt8, packed, (const float*)x_mask, (float*)dx, N, D1, H1, W1, Cl, Cs, 1, workspace,
                                  workspace_floats);
}

extern "C" long long u3d_convtr3d_wgrad_t8_workspace_floats(int N, int D1, int H1, int W1, int Cl, int Cs) {
    if (!u3d_convtr3d_t8_supported(Cl, Cs) || N <= 0 || D1 <= 0 || H1 <= 0 || W1 <= 0) return 0;
    const wgrad_plan q = plan_wgrad(N, D1, H1, W1, Cl, 8 * Cs), q8 = plan_wgrad(N, D1, H1, W1, Cl, 8 * Cs, true);  // (either kernel's tiling)
    return (long long)(q.S > q8.S ? q.S : q8.S) * q.P * 8 * 2048;
}

static int convtr3d_wgrad_t8_impl(int device, u3d_stream_t stream, const float* x, const float* dt8, float* dw, int N, int D1, int H1,
                                  int W1, int Cl, int Cs, float* workspace, long long workspace_floats, int b16);

extern "C" int u3d_convtr3d_wgrad_t8(int device, u3d_stream_t stream, const float* x, const float* dt8, float* dw, int N, int D1,
                                     int H1, int W1, int Cl, int Cs, float* workspace, long long workspace_floats) {
    return convtr3d_wgrad_t8_impl(device, stream, x, dt8, dw, N, D1, H1, W1, Cl, Cs, workspace, workspace_floats, 0);
}

extern "C" int u3d_convtr3d_wgrad_t8_b16(int device, u3d_stream_t stream, const void* x, const void* dt8, float* dw, int N, int D1,
                                         int H1, int W1, int Cl, int Cs, float* workspace, long long workspace_floats) {
    return convtr3d_wgrad_t8_impl(device, stream, (const float*)x, (const float*)dt8, dw, N, D1, H1, W1, Cl, Cs, workspace,
                                  workspace_floats, 1);
}

static int convtr3d_wgrad_t8_impl(int device, u3d_stream_t stream, const float* x, const float* dt8, float* dw, int N, int D1, int H1,
                                  int W1, int Cl, int Cs, float* workspace, long long workspace_floats, int b16) {
    U3D_ENTER(device);
    U3D_REQUIRE(x && dt8 && dw && N > 0 && D1 > 0 && H1 > 0 && W1 > 0 && u3d_convtr3d_t8_supported(Cl, Cs), "u3d_convtr3d_wgrad_t8: bad argument");
    const wgrad_plan q = plan_wgrad(N, D1, H1, W1, Cl, 8 * Cs);
    const long long need = (long long)q.S * q.P * 8 * 2048;
    if (!workspace || workspace_floats < need)
        return u3d_set_err(U3D_EWORKSPACE, "u3d_convtr3d_wgrad_t8: workspace of %lld floats needed, %lld given", need, workspace_floats);
    bf16_wgrad_params p{x, nullptr, dt8, workspace, N, D1, H1, W1, Cl, 8 * Cs, 0, q.tz, q.ty, q.tx, q.tiles, q.per_block, 8 * Cs / 64, g_u3d_tune[9] == 1 ? 0 : 1};
    p.t8cs = (Cs % 64 == 0 && g_u3d_tune[10] != 2) ? Cs : 0;  // (key 10 = 2: no structurally-zero blocks skipped, as for the forward / data gradient)
    // bf16 storage: the round-5 kernel (4 x 8 x 8 tiles, constant-offset staging; key 7 = 1: the round-3 kernel; buffer offsets are 32-bit)
    if (b16 && g_u3d_tune[7] != 1 && (long long)N * D1 * H1 * W1 * (Cl > 8 * Cs ? Cl : 8 * Cs) * 2 < (1ll << 31)) {
        const wgrad_plan q8 = plan_wgrad(N, D1, H1, W1, Cl, 8 * Cs, true);
        if ((long long)q8.S * q8.P * 8 * 2048 <= workspace_floats) {
            p.tz = q8.tz, p.ty = q8.ty, p.tx = q8.tx, p.tiles = q8.tiles, p.per_block = q8.per_block;
            U3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_wgrad_t8v2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        wgt_geom::LDS_TOTAL));
            hipLaunchKernelGGL(conv3d_wgrad_t8v2_kernel, dim3((unsigned)(q8.S * q8.P)), dim3(512), wgt_geom::LDS_TOTAL, (hipStream_t)stream, p);
            U3D_LAUNCH_CHECK();
            hipLaunchKernelGGL(wgrad_t8_reduce_kernel, dim3((unsigned)(Cl * ((Cs + 63) / 64))), dim3(256), 0, (hipStream_t)stream, workspace,
                               q8.S, Cl, Cs, dw);
            U3D_LAUNCH_CHECK();
            return 0;
        }
    }
    if (b16) {
        U3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_wgrad_bf16_kernel<2, __bf16>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 2 * wg_geom<2>::LDS));
        hipLaunchKernelGGL((conv3d_wgrad_bf16_kernel<2, __bf16>), dim3((unsigned)(q.S * q.P)), dim3(512), 2 * wg_geom<2>::LDS,
                           (hipStream_t)stream, p);
    } else {
        U3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_wgrad_bf16_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    2 * wg_geom<2>::LDS));
        hipLaunchKernelGGL(conv3d_wgrad_bf16_kernel<2>, dim3((unsigned)(q.S * q.P)), dim3(512), 2 * wg_geom<2>::LDS, (hipStream_t)stream, p);
    }
    U3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(wgrad_t8_reduce_kernel, dim3((unsigned)(Cl * ((Cs + 63) / 64))), dim3(256), 0, (hipStream_t)stream, workspace,
                       q.S, Cl, Cs, dw);
    U3D_LAUNCH_CHECK();
    return 0;
}


namespace {

// =====================================================================================================================
// 1x1x1 convolution WITH bias (ResNetBlock.conv1, buildingblocks.py:248-255) under bf16 activation storage, on
// v_mfma_f32_32x32x16_bf16.  A plain GEMM per sample, out[v][n] = sum_k in[v][k] * B[k][n] (+ bias[n]):
//   forward        in = block input (V x Cin),  B[k][n] = w[n][k]   (w: (Cout, Cin) fp32, rounded to bf16 like every MFMA operand)
//   data gradient  in = dr (V x Cout),          B[k][n] = w[k][n]
// (the fp32 gather-GEMM of csrc/u3d_res.hip ran these at 13 TF — 1.5 ms of config 4's 20.4 ms step for 25 GFLOP; here they are
// bandwidth: every tensor once).  A wave keeps its B fragments (KS k-steps x NTW n-tiles) in registers and walks 32-voxel M-tiles
// persistently: per tile KS 16-byte loads of A straight from global memory (a lane reads 8 channels of its voxel), KS x NTW MFMAs,
// and an epilogue through a wave-private LDS region ([voxel][channel] bf16: 16-byte stores instead of 2-byte ones), with the
// per-(sample, channel) statistics of the stored values carried in registers across tiles.
struct c1_params {
    const __bf16* in;
    const float* w;
    const float* bias;  // [Nc] or null
    __bf16* out;
    double* stats;      // [N][Nc][2] += (sum, sum of squares) of the stored values, or null
    int N;
    long long V;
    int K, Nc;
    long long wsk, wsn;  // B[k][n] = w[k * wsk + n * wsn]
};

// BL (K = 1024, the bottom of a 5-level net: a few hundred voxels): the B fragments do not fit the register file; they stay in LDS
// (the weight tile keeps its region, the epilogue gets its own) and are read per k-step — slower per MFMA, irrelevant at that size.
template <int KS, int NTW, bool BL>
struct c1_geom {
    static constexpr int RSB = NTW * 64 + 16;  // bytes per voxel row of the epilogue region
    static constexpr int K_ = KS * 16, NT_ = NTW * 32, BROW = K_ * 2 + 16;  // the block's weight tile in LDS: [n][k] bf16, padded rows
    static constexpr int EPI = 4 * 32 * RSB, WT = NT_ * BROW;
    static constexpr int LDS_BYTES = BL ? WT + EPI : (EPI > WT ? EPI : WT);
};

template <int KS, int NTW, bool BL = false>
__global__ __launch_bounds__(256, (BL || KS * NTW > 16 ? 1 : NTW == 4 ? 2 : 3)) void conv1x1_bf16_kernel(const c1_params p) {
    using CG = c1_geom<KS, NTW, BL>;
    constexpr int RSB = CG::RSB, K_ = CG::K_, NT_ = CG::NT_, BROW = CG::BROW;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    __shared__ float red[4][NTW * 32][2];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, col = lane & 31, kh = lane >> 5;
    const int n = blockIdx.z, nb0 = blockIdx.y * NTW * 32;
    // ---- the weight tile once per block: coalesced float4 reads along whichever index is contiguous in memory, bf16 into LDS,
    // then every wave takes its KS x NTW fragments with 16-byte LDS reads (the region is reused by the epilogue afterwards)
    const bool wal = ((uintptr_t)p.w & 15) == 0;  // (a parameter inside a flat buffer need not be 16-byte aligned)
    auto ld4 = [&](const float* src) {
        if (wal) return *reinterpret_cast<const f32x4*>(src);
        return f32x4{src[0], src[1], src[2], src[3]};
    };
    if (p.wsk == 1) {  // forward: w[n][k], k contiguous
        for (int i = t; i < NT_ * (K_ / 4); i += 256) {
            const int nn = i / (K_ / 4), k4 = i - nn * (K_ / 4);
            const f32x4 v = ld4(p.w + (size_t)(nb0 + nn) * p.wsn + 4 * k4);
            const bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
            *reinterpret_cast<bf16x4*>(lds + nn * BROW + k4 * 8) = o;
        }
    } else {  // data gradient: w[k][n], n contiguous
        for (int i = t; i < K_ * (NT_ / 4); i += 256) {
            const int kk = i / (NT_ / 4), n4 = i - kk * (NT_ / 4);
            const f32x4 v = ld4(p.w + (size_t)kk * p.wsk + nb0 + 4 * n4);
#pragma unroll
            for (int e = 0; e < 4; ++e) *reinterpret_cast<__bf16*>(lds + (4 * n4 + e) * BROW + kk * 2) = (__bf16)v[e];
        }
    }
    __syncthreads();
    bf16x8 B[BL ? 1 : KS][NTW];
    if constexpr (!BL) {
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_)
#pragma unroll
            for (int j = 0; j < NTW; ++j) B[s_][j] = *reinterpret_cast<const bf16x8*>(lds + (j * 32 + col) * BROW + (16 * s_ + 8 * kh) * 2);
        __syncthreads();
    }
    float bias[NTW], s1[NTW], s2[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        bias[j] = p.bias ? p.bias[nb0 + j * 32 + col] : 0.f;
        s1[j] = s2[j] = 0.f;
    }
    const __bf16* in = p.in + (size_t)n * p.V * p.K + 8 * kh;
    __bf16* out = p.out + (size_t)n * p.V * p.Nc + nb0;
    char* reg = lds + (BL ? CG::WT : 0) + w * 32 * RSB;
    const long long mtiles = (p.V + 31) / 32;
    constexpr int RING = KS < 4 ? KS : 4;
    bf16x8 a[RING];
    auto fetch = [&](long long mt_) {  // the first RING k-steps of tile mt_ (issued before the previous tile's epilogue)
        const long long v_ = mt_ * 32 + col;
        const __bf16* arow_ = in + (size_t)(v_ < p.V ? v_ : 0) * p.K;
#pragma unroll
        for (int s_ = 0; s_ < RING; ++s_) a[s_] = *reinterpret_cast<const bf16x8*>(arow_ + 16 * s_);
    };
    const long long mt0 = (long long)blockIdx.x * 4 + w, mstep = (long long)gridDim.x * 4;
    if (mt0 < mtiles) fetch(mt0);
    for (long long mt = mt0; mt < mtiles; mt += mstep) {
        const long long v = mt * 32 + col;
        const bool vok = v < p.V;
        const __bf16* arow = in + (size_t)(vok ? v : 0) * p.K;
        f32x16 acc[NTW];
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_) {
            bf16x8 cur = a[s_ & 3];
            if (!vok) cur = bf16x8{};
            if (s_ + 4 < KS) a[s_ & 3] = *reinterpret_cast<const bf16x8*>(arow + 16 * (s_ + 4));
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                bf16x8 bb;
                if constexpr (BL) bb = *reinterpret_cast<const bf16x8*>(lds + (j * 32 + col) * BROW + (16 * s_ + 8 * kh) * 2);
                else bb = B[s_][j];
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur, bb, acc[j], 0, 0, 0);
            }
        }
        if (mt + mstep < mtiles) fetch(mt + mstep);
        // C/D layout: column = lane & 31 (channel), row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5) (voxel of the M-tile)
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * kh;
                const __bf16 vb = (__bf16)(acc[j][e] + bias[j]);
                *reinterpret_cast<__bf16*>(reg + row * RSB + (j * 32 + col) * 2) = vb;
                if (mt * 32 + row < p.V) {
                    const float r = (float)vb;  // (statistics describe the STORED tensor)
                    s1[j] += r;
                    s2[j] = fmaf(r, r, s2[j]);
                }
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 2 * NTW; ++i) {
            const int item = lane + 64 * i, row = item / (NTW * 4), o = item - row * (NTW * 4);
            if (mt * 32 + row < p.V)
                *reinterpret_cast<bf16x8*>(out + (size_t)(mt * 32 + row) * p.Nc + o * 8) = *reinterpret_cast<const bf16x8*>(reg + row * RSB + o * 16);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (p.stats == nullptr) return;
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        s1[j] += __shfl_xor(s1[j], 32);
        s2[j] += __shfl_xor(s2[j], 32);
        if (kh == 0) {
            red[w][j * 32 + col][0] = s1[j];
            red[w][j * 32 + col][1] = s2[j];
        }
    }
    __syncthreads();
    for (int i = t; i < NTW * 32 * 2; i += 256) {
        const int ch = i >> 1, which = i & 1;
        const double sum = (double)red[0][ch][which] + (double)red[1][ch][which] + (double)red[2][ch][which] + (double)red[3][ch][which];
        u3d_atomic_add_f64(p.stats + ((size_t)n * p.Nc + nb0 + ch) * 2 + which, sum);
    }
}

static bool c1_mfma_shape(int K, int Nc) { return K % 16 == 0 && K >= 64 && K <= 1024 && (K & (K - 1)) == 0 && Nc % 32 == 0 && Nc >= 32; }

static int launch_conv1x1_bf16(const c1_params& p, hipStream_t st) {
    const int KS = p.K / 16, nt = p.Nc / 32;
    // B fragments in registers: KS * NTW * 4 VGPRs — at most 64 (3 waves per SIMD: a streaming kernel lives on loads in flight)
    int ntw = KS >= 16 ? 1 : 16 / KS;
    while (ntw > 1 && nt % ntw != 0) ntw >>= 1;
    const int gy = nt / ntw;
    long long gx = (p.V + 255) / 256;  // at least two M-tiles per wave (the next tile's loads are issued before the epilogue)
    const long long cap = 2048 / ((long long)gy * p.N) > 8 ? 2048 / ((long long)gy * p.N) : 8;
    if (gx > cap) gx = cap;
    if (gx > 8) gx &= ~7LL;  // blocks (x, y) and (x, y + 1) read the same voxels: linear ids a multiple of 8 apart = the same XCD's L2
    const dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)p.N);
#define U3D_C1(KS_, NTW_, BL_)                                                                                             \
    if (KS == KS_ && ntw == NTW_) {                                                                                         \
        constexpr int bytes = c1_geom<KS_, NTW_, BL_>::LDS_BYTES;                                                          \
        U3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_bf16_kernel<KS_, NTW_, BL_>),                   \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, bytes));                                   \
        hipLaunchKernelGGL((conv1x1_bf16_kernel<KS_, NTW_, BL_>), grid, dim3(256), bytes, st, p);                          \
        U3D_LAUNCH_CHECK();                                                                                                 \
        return 0;                                                                                                           \
    }
    U3D_C1(4, 4, false) U3D_C1(4, 2, false) U3D_C1(4, 1, false) U3D_C1(8, 2, false) U3D_C1(8, 1, false) U3D_C1(16, 1, false)
    U3D_C1(32, 1, false) U3D_C1(64, 1, true)
#undef U3D_C1
    return u3d_set_err(U3D_EINVAL, "u3d_conv1x1_bf16: no kernel for K = %d, Nc = %d", p.K, p.Nc);
}

// dw[co][ci] = sum over splits (fixed order) of the 1-tap partial sums ws[split][pair][ci 32][co 64]
__global__ void wgrad_k1_reduce_kernel(const float* __restrict__ ws, int S, int C, int K, float* __restrict__ dw) {
    const int pco = K >> 6, P = (C >> 5) * pco;
    const long long total = (long long)C * K;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int co = (int)(i % K), ci = (int)(i / K);
        const int pair = (ci >> 5) * pco + (co >> 6);
        const size_t off = (size_t)pair * 2048 + (ci & 31) * 64 + (co & 63);
        dw[(size_t)co * C + ci] = (float)u3d_sum_splits(ws + off, (size_t)P * 2048, S);
    }
}

// db[co] = sum_v dy[v][co]: per-block partial sums (fixed order inside the block), then one fixed-order pass over the blocks
__global__ __launch_bounds__(256) void colsum_b16_partial_kernel(const __bf16* __restrict__ dy, long long rows, int K, float* __restrict__ part) {
    extern __shared__ float cred[];  // [rws][K]
    const int Q = K >> 3, rws = 256 / Q;
    const int t = threadIdx.x, q = t % Q, row = t / Q;
    float s_[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (row < rws)
        for (long long v = (long long)blockIdx.x * rws + row; v < rows; v += (long long)gridDim.x * rws) {
            const bf16x8 d = *reinterpret_cast<const bf16x8*>(dy + (size_t)v * K + 8 * q);
#pragma unroll
            for (int e = 0; e < 8; ++e) s_[e] += (float)d[e];
        }
    if (row < rws) {
#pragma unroll
        for (int e = 0; e < 8; ++e) cred[row * K + 8 * q + e] = s_[e];
    }
    __syncthreads();
    for (int i = t; i < K; i += 256) {
        float sum = 0.f;
        for (int r = 0; r < rws; ++r) sum += cred[r * K + i];
        part[(size_t)blockIdx.x * K + i] = sum;
    }
}
// block = 16 channels x 16 row groups (a single thread walking all partial rows measured 60 us: 256 dependent loads)
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, int B, int K, float* __restrict__ db) {
    __shared__ double red[16][17];
    const int t = threadIdx.x, cl = t & 15, g = t >> 4, c = blockIdx.x * 16 + cl;
    double sum = 0.0;
    if (c < K)
        for (int b = g; b < B; b += 16) sum += (double)part[(size_t)b * K + c];
    red[g][cl] = sum;
    __syncthreads();
    if (t < 16 && c < K) {
        double tot = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) tot += red[i][t];
        db[c] = (float)tot;
    }
}

}  // namespace

extern "C" int u3d_conv1x1_mfma_b16_supported(int Cin, int Cout) {
    return (c1_mfma_shape(Cin, Cout) && c1_mfma_shape(Cout, Cin) && Cin % 32 == 0 && Cout % 64 == 0) ? 1 : 0;
}

extern "C" int u3d_conv1x1_fwd_mfma_b16(int device, u3d_stream_t stream, const void* x, const float* w, const float* bias, void* y, int N,
                                        int64_t V, int Cin, int Cout, double* out_stats) {
    U3D_ENTER(device);
    U3D_REQUIRE(x && w && y && N > 0 && V > 0 && c1_mfma_shape(Cin, Cout) && (((uintptr_t)x | (uintptr_t)y) & 15) == 0,
                "u3d_conv1x1_fwd_mfma_b16: bad argument (Cin %d, Cout %d)", Cin, Cout);
    c1_params p{(const __bf16*)x, w, bias, (__bf16*)y, out_stats, N, (long long)V, Cin, Cout, 1, Cin};
    return launch_conv1x1_bf16(p, (hipStream_t)stream);
}

extern "C" long long u3d_conv1x1_bwd_mfma_b16_workspace_floats(int N, int D, int H, int W, int Cin, int Cout) {
    if (!u3d_conv1x1_mfma_b16_supported(Cin, Cout) || N <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    const wgrad_plan q = plan_wgrad(N, D, H, W, Cin, Cout);
    return (long long)q.S * q.P * 2048 + 256LL * Cout;
}

// dx (optional) = dy W;  dw (Cout, Cin) fp32 and db (Cout) fp32 written directly (fixed-order reductions: run-to-run identical)
extern "C" int u3d_conv1x1_bwd_mfma_b16(int device, u3d_stream_t stream, const void* dy, const void* x, const float* w, int N, int D, int H,
                                        int W, int Cin, int Cout, void* dx, float* dw, float* db, float* workspace,
                                        long long workspace_floats) {
    U3D_ENTER(device);
    U3D_REQUIRE(dy && x && w && dw && db && N > 0 && D > 0 && H > 0 && W > 0 && u3d_conv1x1_mfma_b16_supported(Cin, Cout) &&
                    (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 15) == 0,
                "u3d_conv1x1_bwd_mfma_b16: bad argument (Cin %d, Cout %d)", Cin, Cout);
    const wgrad_plan q = plan_wgrad(N, D, H, W, Cin, Cout);
    const long long need = (long long)q.S * q.P * 2048 + 256LL * Cout;
    if (!workspace || workspace_floats < need)
        return u3d_set_err(U3D_EWORKSPACE, "u3d_conv1x1_bwd_mfma_b16: workspace of %lld floats needed, %lld given", need, workspace_floats);
    hipStream_t st = (hipStream_t)stream;
    const long long V = (long long)D * H * W;
    if (dx) {
        c1_params p{(const __bf16*)dy, w, nullptr, (__bf16*)dx, nullptr, N, V, Cout, Cin, Cin, 1};
        if (int e = launch_conv1x1_bf16(p, st)) return e;
    }
    // weight gradient: the transposed-read kernel of the 3x3x3 convolutions with ONE tap
    bf16_wgrad_params g{(const float*)x, nullptr, (const float*)dy, workspace, N, D, H, W, Cin, Cout, 0, q.tz, q.ty, q.tx, q.tiles, q.per_block,
                        Cout / 64, 1};
    U3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_wgrad_bf16_kernel<1, __bf16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                2 * wg_geom<1>::LDS));
    hipLaunchKernelGGL((conv3d_wgrad_bf16_kernel<1, __bf16>), dim3((unsigned)(q.S * q.P)), dim3(512), 2 * wg_geom<1>::LDS, st, g);
    U3D_LAUNCH_CHECK();
    long long rb = ((long long)Cin * Cout + 255) / 256;
    hipLaunchKernelGGL(wgrad_k1_reduce_kernel, dim3((unsigned)rb), dim3(256), 0, st, workspace, q.S, Cin, Cout, dw);
    U3D_LAUNCH_CHECK();
    float* part = workspace + (long long)q.S * q.P * 2048;
    const int rws = 256 / (Cout / 8);
    hipLaunchKernelGGL(colsum_b16_partial_kernel, dim3(256), dim3(256), (size_t)rws * Cout * sizeof(float), st, (const __bf16*)dy, (long long)N * V,
                       Cout, part);
    U3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((Cout + 15) / 16)), dim3(256), 0, st, part, 256, Cout, db);
    U3D_LAUNCH_CHECK();
    return 0;
}

// =====================================================================================================================
// FP32 convolution on the bf16 matrix pipe ("split fp32", opt-in: compute_dtype 'fp32_split').
//
// On gfx950 v_mfma_f32_32x32x16_bf16 delivers 16x the multiply-adds per cycle of v_mfma_f32_32x32x2_f32.  An fp32 value splits
// EXACTLY into three bf16 values, a = a_h + a_m + a_l (8 + 8 + 8 mantissa bits; a_h = bf16(a), a_m = bf16(a - a_h),
// a_l = bf16(a - a_h - a_m), every subtraction exact), each bf16 x bf16 product is exact in fp32, and the products whose
// relative weight is >= 2^-16 of a*b are
//     a_h*b_h, a_h*b_m, a_m*b_h, a_h*b_l, a_l*b_h, a_m*b_m        (dropped: a_m*b_l, a_l*b_m ~2^-24, a_l*b_l ~2^-32),
// i.e. SIX bf16 MFMAs with FP32 accumulation reproduce the fp32 product to ~2^-23 relative — the size of ONE rounding of the
// fp32 MFMA's own accumulation chain, far inside the reduction-order noise of any fp32 convolution (tests/test_gpu_f32s.py
// measures both kernels against float64).  6/16 of the fp32 pipe time, and operand reuse comes for free: per (tap, 16-channel
// chunk) a wave reads 3 x 2 A fragments and 3 x NT B fragments for 6 x 2 x NT MFMAs (0.5 loads per MFMA, the bf16 kernel above
// needs 1.0), so neither the L1 (B) nor the LDS (A) path limits it.
//
// Same tiling as conv3d_bf16_kernel<NT, 1, 3>: block = 4 waves, tile 4 x 8 x 8 voxels x 32*NT channels, halo tile of a chunk in
// LDS as [part h|m|l][channel half][hz][hy][hx pad 12][8 bf16] (6 planes, 69 KB: ONE buffer, two blocks per CU; while one block
// restages, the other owns the MFMA pipe — a chunk is 27 x 12*NT MFMAs = 10-20 k cycles against ~3 k cycles of staging).
// Activations are split while staging (GroupNorm affine in fp32 first, zero padding after it), weights by
// u3d_pack_weights_f32s from the fp32 master copy into three images in pack_weights_bf16_kernel's layout.
namespace {

template <int NT, int ABL = 0>
__global__ __launch_bounds__(256, 2) void conv3d_f32s_kernel(const bf16_conv_params p) {
    constexpr bool g_interleave_off = (ABL & 16) != 0;
    using G = tile_geom<1, 3>;
    constexpr int HY = G::HY, NTAPS = 27;
    constexpr int PART = 2 * G::PLANE;  // bytes per (h | m | l) part = two channel-half planes
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int nblk = p.K / (32 * NT);
    const int bid = u3d_xcd_remap(blockIdx.x, gridDim.x);
    const int nb = bid % nblk;
    int tile = bid / nblk;
    const int split = tile % p.ksplit;
    tile /= p.ksplit;
    const int txi = tile % p.tx;
    tile /= p.tx;
    const int tyi = tile % p.ty;
    tile /= p.ty;
    const int tzi = tile % p.tz;
    const int n = tile / p.tz;
    const int z0 = tzi * G::TZ, y0 = tyi * 8, x0 = txi * 8;
    const int nch_all = p.C >> 4;
    const int cps = (nch_all + p.ksplit - 1) / p.ksplit;
    const int cbeg = split * cps, nch = min(nch_all, cbeg + cps);
    const int ntiles = p.K >> 5;
    const int q = t & 3;
    const int r = lane & 31, kh = lane >> 5;
    const int a_base = kh * G::PLANE + ((w * HY + (r & 3)) * HS + (r >> 2)) * 16;

    // staging descriptors (as in conv3d_bf16_kernel): element offset from the halo origin, validity bit, LDS byte offset
    int rel[G::ITERS], lo[G::ITERS];
    unsigned okmask = 0;
    {
        const int hv0 = t >> 2;
        const int bz = hv0 / (G::HY * G::HX), brem = hv0 - bz * (G::HY * G::HX), by = brem / G::HX, bx = brem - by * G::HX;
        const int rel_safe = ((p.H + 1) * p.W + 1) * p.C;  // the tile's first output voxel: always inside the volume
#pragma unroll
        for (int it = 0; it < G::ITERS; ++it) {
            constexpr int HYX = G::HY * G::HX;
            const int dz = (64 * it) / HYX, dy = ((64 * it) % HYX) / G::HX, dx = (64 * it) % G::HX;
            int hx = bx + dx, hy = by + dy, hz = bz + dz;
            if (hx >= G::HX) hx -= G::HX, hy += 1;
            if (hy >= G::HY) hy -= G::HY, hz += 1;
            const int z = z0 - 1 + hz, y = y0 - 1 + hy, xx = x0 - 1 + hx;
            const bool ok = hz < G::HZ && (unsigned)z < (unsigned)p.D && (unsigned)y < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
            rel[it] = ok ? ((hz * p.H + hy) * p.W + hx) * p.C : rel_safe;
            okmask |= (ok ? 1u : 0u) << it;
            lo[it] = hz < G::HZ ? (q >> 1) * G::PLANE + ((hz * G::HY + hy) * HS + hx) * 16 + (q & 1) * 8 : G::PLANE - 64 + (t & 7) * 8;
        }
    }
    const float* xo = p.x + ((((long long)n * p.D + (z0 - 1)) * p.H + (y0 - 1)) * p.W + (x0 - 1)) * (long long)p.C + 4 * q;

    f32x16 acc[2][NT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][j][e] = 0.f;

    // B fragments one tap ahead (a tap is 12*NT MFMAs = 384-768 pipe cycles, well above the L2 latency); the ring runs across
    // chunk boundaries (the images are linear in (chunk, tap) and carry tail padding)
    constexpr int BR = 3;  // B ring: fragments BR - 1 taps ahead (27 % BR == 0 keeps the slots aligned across chunks)
    bf16x8 bq[BR][3][NT];
    // the images are block-contiguous, [channel block nb][chunk][tap][j < NT][lane]: one running pointer per image, advanced by the
    // compile-time tap stride, so that the unrolled loop needs no table of scalar offsets
    const bf16x8* wq[3];
#pragma unroll
    for (int pb = 0; pb < 3; ++pb) wq[pb] = p.wpk + pb * p.wpart + ((size_t)nb * (nch_all * NTAPS + 2) + (size_t)cbeg * NTAPS) * (NT * 64);
    auto load_b = [&](int slot) {  // the next tap's fragments; advances the pointers
#pragma unroll
        for (int pb = 0; pb < 3; ++pb) {
#pragma unroll
            for (int j = 0; j < NT; ++j) bq[slot][pb][j] = wq[pb][j * 64 + lane];
            if constexpr (!(ABL & 8)) wq[pb] += NT * 64;
        }
    };
    if (cbeg < nch) {
#pragma unroll
        for (int d = 0; d < BR - 1; ++d) load_b(d);
    }

    for (int c = cbeg; c < nch; ++c) {
        // ---- stage chunk c: fp32 -> affine -> (h, m, l) bf16 planes
        if (c > cbeg) {
            __syncthreads();  // everyone is done reading the previous chunk
            // The bf16 MFMA rounds its accumulation toward -infinity (measured: tools/f32s_error_probe.py — a drift linear in
            // the number of accumulations, which sums over millions of voxels do not average out).  The weight images carry
            // the sign (-1)^chunk and the accumulator is negated between chunks (exact), so that the drift alternates in sign
            // and cancels: result = (-1)^(last chunk) * acc.
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[m][j][e] = -acc[m][j][e];
        }
        if (!(ABL & 1) || c == cbeg) {
            f32x4 ga, gb;
            u3d_load_affine(p.affine, n, p.C, (c << 4) + 4 * q, true, ga, gb);
            f32x4 v[G::ITERS];
#pragma unroll
            for (int it = 0; it < G::ITERS; ++it) v[it] = *reinterpret_cast<const f32x4*>(xo + rel[it] + (c << 4));
#pragma unroll
            for (int it = 0; it < G::ITERS; ++it) {
                const bool ok = (okmask >> it) & 1u;
                bf16x4 oh, om, ol;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float g = ok ? fmaf(v[it][e], ga[e], gb[e]) : 0.f;
                    const __bf16 h = (__bf16)g;
                    const float r1 = g - (float)h;
                    const __bf16 mm = (__bf16)r1;
                    oh[e] = h, om[e] = mm, ol[e] = (__bf16)(r1 - (float)mm);
                }
                *reinterpret_cast<bf16x4*>(lds + lo[it]) = oh;
                *reinterpret_cast<bf16x4*>(lds + PART + lo[it]) = om;
                *reinterpret_cast<bf16x4*>(lds + 2 * PART + lo[it]) = ol;
            }
        }
        __syncthreads();
        bf16x8 aq[2][3][2] = {};
        auto load_a = [&](int slot, int tap) {
            const int tzz = tap / 9, tyy = (tap / 3) % 3, txx = tap % 3;
#pragma unroll
            for (int pa = 0; pa < 3; ++pa)
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    aq[slot][pa][m] = *reinterpret_cast<const bf16x8*>(lds + pa * PART + a_base + ((tzz * HY + (m * 4 + tyy)) * HS + txx) * 16);
        };
        load_a(0, 0);
#pragma unroll
        for (int tap = 0; tap < NTAPS; ++tap) {
            const int cur = tap & 1, nxt = cur ^ 1;
            if constexpr (!(ABL & 2)) load_b((tap + BR - 1) % BR);  // (the last taps fetch the next chunk's first ones, or the tail padding)
            if (!(ABL & 4) && tap + 1 < NTAPS) load_a(nxt, tap + 1);
            if (g_interleave_off) __builtin_amdgcn_sched_barrier(0);
            // the six product classes, smallest first; consecutive MFMAs go to different accumulators
#pragma unroll
            for (int cls = 0; cls < 6; ++cls) {
                constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[cur][PA[cls]][m], bq[tap % BR][PB[cls]][j], acc[m][j], 0, 0, 0);
            }
            if (!g_interleave_off) {
                // a VMEM issue costs tens of cycles (MI355X_MICROARCH.md: ~60 cyc per 1 KiB load among MFMAs): spread the tap's
                // prefetches between its MFMAs instead of a burst in front of them
#pragma unroll
                for (int i = 0; i < 6; ++i) {  // next tap's A fragments first (needed soonest)
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
                    __builtin_amdgcn_sched_group_barrier(0x008, NT, 0); // NT MFMA
                }
#pragma unroll
                for (int i = 0; i < 3 * NT; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // 1 VMEM read
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // 2 MFMA
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (nch > cbeg && ((nch - 1) & 1)) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[m][j][e] = -acc[m][j][e];
    }
    __syncthreads();  // the LDS tile is reused by the epilogue's block reduction
    conv_tile_epilogue<NT, 1, 3>(p, acc, lds, n, nb, split, z0, y0, x0, t, lane, w);
}

// image part: 0 = bf16(w), 1 = bf16(w - h), 2 = bf16(w - h - m); source element (row = produced/contraction channel as in
// pack_weights_bf16_kernel) read with row stride `ld` input channels and input-channel offset `ci_off` (channel slices of a
// wider weight tensor: the skip half of a decoder's first convolution)
__global__ void pack_weights_f32s_kernel(const float* __restrict__ w, int Cout, int Cin, int mode, int ld, int ci_off,
                                         __bf16* __restrict__ out, long long per_part, long long total) {
    const int Kc = mode == 0 ? Cin : Cout, Nc = mode == 0 ? Cout : Cin;
    const int NT = Nc % 64 == 0 ? 2 : 1, nchunks = Kc >> 4;
    // image [nb][chunk (+ one tap of tail padding per nb)][tap][j][lane][8]
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(i & 7);
        long long rest = i >> 3;
        const int l = (int)(rest & 63);
        rest >>= 6;
        const int j = (int)(rest % NT);
        rest /= NT;
        const long long per_nb = (long long)nchunks * 27 + 2;
        const int ct = (int)(rest % per_nb);  // chunk*27 + tap, or the padding slot
        const int nb = (int)(rest / per_nb);
        float v = 0.f;
        if (ct < nchunks * 27) {
            const int c = ct / 27, tap = ct - c * 27;
            const int k = c * 16 + 8 * (l >> 5) + e, col = (nb * NT + j) * 32 + (l & 31);
            if (mode == 0)
                v = w[((size_t)col * ld + ci_off + k) * 27 + tap];
            else
                v = w[((size_t)k * ld + ci_off + col) * 27 + (26 - tap)];
            if (c & 1) v = -v;  // sign (-1)^chunk: see the accumulator negation in conv3d_f32s_kernel
        }
        const __bf16 h = (__bf16)v;
        const float r1 = v - (float)h;
        const __bf16 m = (__bf16)r1;
        out[i] = h;
        out[per_part + i] = m;
        out[2 * per_part + i] = (__bf16)(r1 - (float)m);
    }
}

}  // namespace

static long long f32s_part_elems(int Cin, int Cout, int mode) {
    const int Kc = mode == 0 ? Cin : Cout, Nc = mode == 0 ? Cout : Cin;
    if (Kc <= 0 || Nc <= 0 || Kc % 16 != 0 || Nc % 32 != 0) return 0;
    return ((long long)(Kc / 16) * 27 + 2) * (Nc / 32) * 64 * 8;  // two taps of tail padding per channel block (prefetched, never used)
}

extern "C" long long u3d_packed_weight_f32s_elems(int Cin, int Cout, int mode) { return 3 * f32s_part_elems(Cin, Cout, mode); }

extern "C" int u3d_pack_weights_f32s(int device, u3d_stream_t stream, const float* w, int Cout, int Cin, int mode, int ld,
                                     int ci_off, void* packed) {
    U3D_ENTER(device);
    const long long per = f32s_part_elems(Cin, Cout, mode);
    U3D_REQUIRE(w && packed && (mode == 0 || mode == 1) && per > 0 && ld >= Cin && ci_off >= 0 && ci_off + Cin <= ld,
                "u3d_pack_weights_f32s: needs contraction channels %% 16 == 0, produced channels %% 32 == 0 (Cin %d, Cout %d) and a "
                "channel slice inside the row (ld %d, offset %d)", Cin, Cout, ld, ci_off);
    const long long total = per;
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_weights_f32s_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, Cout, Cin, mode, ld,
                       ci_off, reinterpret_cast<__bf16*>(packed), per, total);
    U3D_LAUNCH_CHECK();
    return 0;
}

template <int NT, int ABL = 0>
static int launch_f32s(const bf16_conv_params& p, hipStream_t stream) {
    using G = tile_geom<1, 3>;
    bf16_conv_params q = p;
    q.tz = (p.D + G::TZ - 1) / G::TZ;
    q.ty = (p.H + 7) / 8;
    q.tx = (p.W + 7) / 8;
    const long long blocks = (long long)p.N * q.tz * q.ty * q.tx * (p.K / (32 * NT)) * p.ksplit;
    if (blocks > 0x7fffffffLL) return u3d_set_err(U3D_EINVAL, "u3d_conv3d_f32s: grid too large");
    const size_t shmem = 6 * (size_t)G::PLANE;
    U3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_f32s_kernel<NT, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)shmem));
    hipLaunchKernelGGL((conv3d_f32s_kernel<NT, ABL>), dim3((unsigned)blocks), dim3(256), shmem, stream, q);
    U3D_LAUNCH_CHECK();
    if (p.ksplit > 1) {
        const long long V = (long long)p.D * p.H * p.W;
        const int vper = 64;
        hipLaunchKernelGGL(splitk_bf16_reduce_kernel, dim3((unsigned)((V + vper - 1) / vper), (unsigned)((p.K + 63) / 64), (unsigned)p.N),
                           dim3(256), 0, stream, q, V, vper);
        U3D_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int u3d_conv3d_f32s(int device, u3d_stream_t stream, const float* x, const float* affine, const void* packed_w, float* out,
                               int N, int D, int H, int W, int C, int K, int relu, double* out_stats, const float* gx,
                               double* gstats, const float* residual, float* workspace, long long workspace_floats) {
    U3D_ENTER(device);
    U3D_REQUIRE(x && packed_w && out && N > 0 && D > 0 && H > 0 && W > 0, "u3d_conv3d_f32s: bad argument");
    U3D_REQUIRE(u3d_conv3d_bf16_supported(C, K), "u3d_conv3d_f32s: needs Cin %% 16 == 0 and Cout %% 32 == 0 (got %d, %d)", C, K);
    U3D_REQUIRE(!(out_stats && gstats), "u3d_conv3d_f32s: out_stats and gstats are mutually exclusive");
    U3D_REQUIRE(!gstats || gx, "u3d_conv3d_f32s: gstats needs gx");
    U3D_REQUIRE(!(residual && gstats), "u3d_conv3d_f32s: residual and gx/gstats are mutually exclusive");
    U3D_REQUIRE((((uintptr_t)x | (uintptr_t)packed_w | (uintptr_t)affine) & 15) == 0, "u3d_conv3d_f32s: 16-byte alignment");
    bf16_conv_params p{x, affine, reinterpret_cast<const bf16x8*>(packed_w), out, residual, gx, out_stats, gstats,
                       N, D, H, W, C, K, relu, 0, 0, 0, 1, nullptr, 1, nullptr, 0};
    p.wpart = f32s_part_elems(C, K, 0) / 8;
    const int ks = bf16_ksplit(N, D, H, W, C, K);
    if (ks > 1 && workspace && workspace_floats >= (long long)ks * N * D * H * W * K) {
        p.ksplit = ks;
        p.ws = workspace;
    }
    hipStream_t s = (hipStream_t)stream;
    if (K % 64 == 0 && g_u3d_tune[6] >= 300) {  // TIMING-ONLY ablations (wrong results; DESIGN_HISTORY.md 4.9 quotes them)
        switch (g_u3d_tune[6] - 300) {
            case 2: return launch_f32s<2, 2>(p, s);    // no B-fragment loads in the loop
            case 16: return launch_f32s<2, 16>(p, s);  // prefetches as a burst in front of each tap's MFMAs
        }
    }
    return K % 64 == 0 ? launch_f32s<2>(p, s) : launch_f32s<1>(p, s);
}
