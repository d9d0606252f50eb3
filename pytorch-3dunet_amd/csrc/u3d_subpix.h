// u3d_subpix.h — tiling constants and the packed-weight layout of the sub-pixel convolution (csrc/u3d_subpix.hip); shared with
// the batch weight packer in u3d_conv.hip.
#pragma once
#include <type_traits>

#include "u3d_common.h"

namespace sp {
constexpr int TZ = 4, TY = 4, TX = 8;
constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2;
constexpr int CC = 16, CS = 16;
constexpr int RS = HX * CS + 4;   // 164: the conflict-free row stride of u3d_conv.hip
constexpr int PS = HY * RS;       // 984
constexpr int TILE_FLOATS = HZ * PS + 4;
constexpr int NITEMS = HZ * HY * HX * (CC / 4);  // 1440 float4 items per chunk
constexpr int NIT = (NITEMS + 255) / 256;        // 6
constexpr int NSTEP = 54;
constexpr int PACK_PAD = 9;  // zero fragments appended to the packed image (fetch overrun, >= RING)

// Which output parity classes use which halo offset h (0,1,2 = low-res offset -1,0,+1), per dimension.
//   Nearest2x: conv3x3x3 over a nearest-2x-upsampled tensor — parity 0 reads offsets {-1,0}, parity 1 reads {0,+1}.
//   Deconv3s2: ConvTranspose3d(k=3, stride=2, padding=1) (buildingblocks.py:653-662) — output 2j reads input j (tap 1),
//              output 2j+1 reads inputs j (tap 2) and j+1 (tap 0); offset -1 is never used.
struct Nearest2x {
    static constexpr int RING = 8;  // B ring slots; fragments are fetched RING-1 ahead (NFRAG % RING == 0)
    __host__ __device__ static constexpr int ncls1(int h) { return h == 1 ? 2 : 1; }
    __host__ __device__ static constexpr int cls1(int h, int i) { return h == 0 ? 0 : (h == 2 ? 1 : i); }
};
struct Deconv3s2 {
    static constexpr int RING = 9;
    __host__ __device__ static constexpr int ncls1(int h) { return h == 0 ? 0 : (h == 1 ? 2 : 1); }
    __host__ __device__ static constexpr int cls1(int h, int i) { return h == 1 ? i : 1; }
};
template <class S>
struct Sch {
    __host__ __device__ static constexpr int ncls(int tap) {
        return S::ncls1(tap / 9) * S::ncls1((tap / 3) % 3) * S::ncls1(tap % 3);
    }
    // i-th class (pz*4 + py*2 + px) using halo offset `tap`
    __host__ __device__ static constexpr int cls(int tap, int i) {
        const int ny = S::ncls1((tap / 3) % 3), nx = S::ncls1(tap % 3);
        const int ix = i % nx, iy = (i / nx) % ny, iz = i / (nx * ny);
        return S::cls1(tap / 9, iz) * 4 + S::cls1((tap / 3) % 3, iy) * 2 + S::cls1(tap % 3, ix);
    }
    __host__ __device__ static constexpr int prefix(int st) {  // fragments of a chunk before k-step st
        int s = 0;
        for (int k = 0; k < st; ++k) s += ncls(k >> 1);
        return s;
    }
    __host__ __device__ static constexpr int nact() {  // k-steps with at least one class
        int n = 0;
        for (int st = 0; st < NSTEP; ++st) n += ncls(st >> 1) > 0 ? 1 : 0;
        return n;
    }
    __host__ __device__ static constexpr int act(int k) {  // the k-th such step
        int n = 0;
        for (int st = 0; st < NSTEP; ++st)
            if (ncls(st >> 1) > 0) {
                if (n == k) return st;
                ++n;
            }
        return NSTEP;
    }
};
template <class S>
constexpr int NFRAG_OF = Sch<S>::prefix(NSTEP);  // B fragments per chunk
template <class S>
constexpr int NACT_OF = Sch<S>::nact();
constexpr int NFRAG = NFRAG_OF<Nearest2x>;  // 128
constexpr int RING = Nearest2x::RING;
static_assert(NFRAG == 128 && NFRAG % RING == 0, "fragment count");
static_assert(NFRAG_OF<Deconv3s2> == 54 && NFRAG_OF<Deconv3s2> % Deconv3s2::RING == 0 && NACT_OF<Deconv3s2> == 16, "deconv scheme");
__host__ __device__ constexpr int ncls(int tap) { return Sch<Nearest2x>::ncls(tap); }
__host__ __device__ constexpr int cls(int tap, int i) { return Sch<Nearest2x>::cls(tap, i); }
__host__ __device__ constexpr int prefix(int st) { return Sch<Nearest2x>::prefix(st); }

template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// fragment f of a chunk -> (k-step, index of the class within the step), as a compile-time table
template <class S>
struct FragTabT {
    unsigned char st[NFRAG_OF<S>], i[NFRAG_OF<S>];
};
template <class S>
constexpr FragTabT<S> make_frag_tab() {
    FragTabT<S> t{};
    int f = 0;
    for (int st = 0; st < NSTEP; ++st)
        for (int i = 0; i < Sch<S>::ncls(st >> 1); ++i) {
            t.st[f] = (unsigned char)st;
            t.i[f] = (unsigned char)i;
            ++f;
        }
    return t;
}
using FragTab = FragTabT<Nearest2x>;
__device__ __forceinline__ const FragTab& frag_tab() {
    static constexpr FragTab tab = make_frag_tab<Nearest2x>();
    return tab;
}
__device__ __forceinline__ const FragTabT<Deconv3s2>& frag_tab_deconv() {
    static constexpr FragTabT<Deconv3s2> tab = make_frag_tab<Deconv3s2>();
    return tab;
}

// ---- weight packing: f32x4 index (((ch*128 + f)*ncb + cb)*64 + lane), element j; fragment f = (k-step st, i-th class of
// the step's halo offset); k-channel = ch*16 + 8*(st&1) + 4*(lane>>5) + j, n-channel = cb*32 + (lane&31); the value is the
// SUM of the original taps of that class which read this low-res voxel.  `w` points at the first packed input channel of
// the (Cout, cstride, 3,3,3) weight.
__host__ __device__ inline long long packed_floats(int C1, int Cout) {
    return ((long long)((C1 + 15) / 16) * NFRAG + PACK_PAD) * ((Cout + 31) / 32) * 256;
}

// value of fragment f for ONE (n-channel, k-channel) pair whose 27 original taps start at wr (global memory or an LDS copy)
__device__ __forceinline__ float frag_value(const float* wr, int f) {
    const FragTab& ft = frag_tab();
    const int st = ft.st[f];
    const int tap = st >> 1, ci = cls(tap, ft.i[f]);
    const int hh[3] = {tap / 9, (tap / 3) % 3, tap % 3};
    const int pp[3] = {ci >> 2, (ci >> 1) & 1, ci & 1};
    int lo[3], num[3];  // original taps t in [lo, lo+num) of parity p read halo offset h
    for (int d = 0; d < 3; ++d) {
        if (pp[d] == 0) {
            lo[d] = hh[d] == 0 ? 0 : 1;
            num[d] = hh[d] == 0 ? 1 : 2;
        } else {
            lo[d] = hh[d] == 1 ? 0 : 2;
            num[d] = hh[d] == 1 ? 2 : 1;
        }
    }
    float v = 0.f;
    for (int a = 0; a < num[0]; ++a)
        for (int b = 0; b < num[1]; ++b)
            for (int c = 0; c < num[2]; ++c) v += wr[((lo[0] + a) * 3 + lo[1] + b) * 3 + lo[2] + c];
    return v;
}

__device__ __forceinline__ float pack_elem(const float* __restrict__ w, int Cout, int cstride, int C1, int nchunks, int ncb,
                                           long long idx) {
    const int j = (int)(idx & 3);
    const int lane = (int)((idx >> 2) & 63);
    long long r = idx >> 8;
    const int cb = (int)(r % ncb);
    r /= ncb;
    const int f = (int)(r % NFRAG);
    const int ch = (int)(r / NFRAG);
    if (ch >= nchunks) return 0.f;  // the trailing zero fragments
    const int st = frag_tab().st[f];
    const int kc = ch * 16 + 8 * (st & 1) + 4 * (lane >> 5) + j;
    const int nc = cb * 32 + (lane & 31);
    if (kc >= C1 || nc >= Cout) return 0.f;
    return frag_value(w + ((size_t)nc * cstride + kc) * 27, f);
}

// ConvTranspose3d(k=3, s=2, p=1) weight (Cin, Cout, 3,3,3) in the same fragment layout (54 fragments per chunk): every (class,
// offset) pair is a single tap — per dimension (parity 0, offset 0) -> tap 1, (1, 0) -> tap 2, (1, +1) -> tap 0.
__host__ __device__ inline long long packed_floats_deconv(int Cin, int Cout) {
    return ((long long)((Cin + 15) / 16) * NFRAG_OF<Deconv3s2> + PACK_PAD) * ((Cout + 31) / 32) * 256;
}
__device__ __forceinline__ float pack_elem_deconv(const float* __restrict__ w, int Cin, int Cout, int nchunks, int ncb,
                                                  long long idx) {
    constexpr int NF = NFRAG_OF<Deconv3s2>;
    const int j = (int)(idx & 3);
    const int lane = (int)((idx >> 2) & 63);
    long long r = idx >> 8;
    const int cb = (int)(r % ncb);
    r /= ncb;
    const int f = (int)(r % NF);
    const int ch = (int)(r / NF);
    if (ch >= nchunks) return 0.f;
    const FragTabT<Deconv3s2>& ft = frag_tab_deconv();
    const int st = ft.st[f];
    const int tap = st >> 1, ci = Sch<Deconv3s2>::cls(tap, ft.i[f]);
    const int kc = ch * 16 + 8 * (st & 1) + 4 * (lane >> 5) + j;
    const int nc = cb * 32 + (lane & 31);
    if (kc >= Cin || nc >= Cout) return 0.f;
    const int hh[3] = {tap / 9, (tap / 3) % 3, tap % 3};
    const int pp[3] = {ci >> 2, (ci >> 1) & 1, ci & 1};
    int t3[3];
    for (int d = 0; d < 3; ++d) t3[d] = pp[d] == 0 ? 1 : (hh[d] == 1 ? 2 : 0);
    return w[((size_t)kc * Cout + nc) * 27 + (t3[0] * 3 + t3[1]) * 3 + t3[2]];
}
}  // namespace sp

// ---- data gradient of the same operator with respect to the LOW-RES tensor (csrc/u3d_subpix.hip, subpixel_dgrad_kernel):
//   dlow[i][c] = sum_{d in {-1,0,1,2}^3} sum_k dz[2i + d][k] * Wd[d][k][c],
// Wd[d] = the sum of the original taps t with (v + t - 1) >> 1 == i for v = 2i + d: per dimension d=-1 -> {2}, d=0 -> {1,2},
// d=1 -> {0,1}, d=2 -> {0}.  64 taps at stride 2 instead of 8 x 27.
namespace spd {
constexpr int TZ = 2, TY = 4, TX = 8;                          // low-res output tile of a block
constexpr int RZ = 2 * TZ + 2, RY = 2 * TY + 2, RX = 2 * TX + 2;  // full-res dz region 6 x 10 x 18
constexpr int ROW = 2 * (RX / 2) * 16 + 4;                     // 292: [x parity][9][16 ch] + pad; ROW/4 = 1 (mod 4): conflict-free
constexpr int REGION_FLOATS = RZ * RY * ROW;                   // [rz][y parity][5][ROW] = 17520 floats (70 KB)
constexpr int NITEMS = RZ * RY * RX * 4;                       // 4320 float4 items per 16-channel chunk
constexpr int NIT = (NITEMS + 255) / 256;                      // 17
constexpr int NFRAG = 128;                                     // 64 taps x 2 channel octets
constexpr int RING = 8, PACK_PAD = 8;

__host__ __device__ inline long long packed_floats(int K, int C1) {
    return ((long long)((K + 15) / 16) * NFRAG + PACK_PAD) * ((C1 + 31) / 32) * 256;
}

// f32x4 index (((ch*128 + f)*ntot + ntg)*64 + lane), element j; f = tap*2 + octet, tap = (dz+1)*16 + (dy+1)*4 + (dx+1);
// k (dz channel) = ch*16 + 8*octet + 4*(lane>>5) + j, n (low-res channel) = ntg*32 + (lane&31).  `w` points at the first
// upsampled input channel of the (K, cstride, 3,3,3) weight.
// value of fragment f (tap = f >> 1) for ONE (dz channel k, low-res channel n) pair whose 27 original taps start at wr
__device__ __forceinline__ float frag_value(const float* wr, int f) {
    const int tap = f >> 1;
    const int di[3] = {tap >> 4, (tap >> 2) & 3, tap & 3};
    int lo[3], num[3];
    for (int d = 0; d < 3; ++d) {
        lo[d] = di[d] == 0 ? 2 : (di[d] == 1 ? 1 : 0);
        num[d] = (di[d] == 1 || di[d] == 2) ? 2 : 1;
    }
    float v = 0.f;
    for (int a = 0; a < num[0]; ++a)
        for (int b = 0; b < num[1]; ++b)
            for (int c = 0; c < num[2]; ++c) v += wr[((lo[0] + a) * 3 + lo[1] + b) * 3 + lo[2] + c];
    return v;
}

__device__ __forceinline__ float pack_elem(const float* __restrict__ w, int K, int cstride, int C1, int nchunks, int ntot,
                                           long long idx) {
    const int j = (int)(idx & 3);
    const int lane = (int)((idx >> 2) & 63);
    long long r = idx >> 8;
    const int ntg = (int)(r % ntot);
    r /= ntot;
    const int f = (int)(r % NFRAG);
    const int ch = (int)(r / NFRAG);
    if (ch >= nchunks) return 0.f;
    const int kc = ch * 16 + 8 * (f & 1) + 4 * (lane >> 5) + j;
    const int nc = ntg * 32 + (lane & 31);
    if (kc >= K || nc >= C1) return 0.f;
    return frag_value(w + ((size_t)kc * cstride + nc) * 27, f);
}
}  // namespace spd
