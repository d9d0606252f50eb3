// u3d_common.h — shared device helpers for the gfx950 3D U-Net kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "u3d.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- activation storage type: fp32 (default) or bf16 (`activation_dtype: bf16`, csrc/u3d_b16 entry points).  All arithmetic
// is fp32; only the bytes in HBM change.  Stores round to nearest even (v_cvt_pk_bf16_f32).
typedef __bf16 u3d_bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 u3d_ldq(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 u3d_ldq(const __bf16* p) {
    const u3d_bf16x4 v = *reinterpret_cast<const u3d_bf16x4*>(p);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
__device__ __forceinline__ void u3d_stq(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void u3d_stq(__bf16* p, const f32x4& v) {
    *reinterpret_cast<u3d_bf16x4*>(p) = u3d_bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
}
__device__ __forceinline__ float u3d_ld(const float* p) { return *p; }
__device__ __forceinline__ float u3d_ld(const __bf16* p) { return (float)*p; }
__device__ __forceinline__ void u3d_st(float* p, float v) { *p = v; }
__device__ __forceinline__ void u3d_st(__bf16* p, float v) { *p = (__bf16)v; }
// what a stored value reads back as (statistics are taken over the STORED tensor: the consumer's GroupNorm normalises that)
__device__ __forceinline__ float u3d_stored(float v, const float*) { return v; }
__device__ __forceinline__ float u3d_stored(float v, const __bf16*) { return (float)(__bf16)v; }
template <typename T>
struct u3d_vec_align {  // bytes a 4-channel quad access needs
    static constexpr uintptr_t mask = sizeof(T) == 4 ? 15 : 7;
};

int u3d_set_err(int code, const char* fmt, ...);
// Every entry point runs with `device` current and RESTORES the caller's current device on return: the library never
// changes the calling thread's HIP device behind PyTorch (autograd worker threads, nn.DataParallel replica threads).
struct u3d_device_guard {
    int prev = -1;
    bool switched = false;
    int enter(int device);
    ~u3d_device_guard() {
        if (switched) (void)hipSetDevice(prev);
    }
};
#define U3D_ENTER(device)           \
    u3d_device_guard _u3d_guard;    \
    if (int _e = _u3d_guard.enter(device)) return _e

#define U3D_HIP(call)                                                                       \
    do {                                                                                    \
        hipError_t _e = (call);                                                             \
        if (_e != hipSuccess)                                                               \
            return u3d_set_err(U3D_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, \
                               __LINE__);                                                   \
    } while (0)

#define U3D_REQUIRE(cond, ...)                              \
    do {                                                    \
        if (!(cond)) return u3d_set_err(U3D_EINVAL, __VA_ARGS__); \
    } while (0)

#define U3D_LAUNCH_CHECK() U3D_HIP(hipGetLastError())

// ---- (virtual) source tensor access ------------------------------------------------------------
// voxel indices of (n,z,y,x) in the full-res source and (through the nearest maps) in the low-res one
__device__ __forceinline__ void u3d_vox_index(const u3d_src_t& s, int n, int z, int y, int x, int D, int H, int W,
                                              int& v0, int& v1) {
    v0 = ((n * D + z) * H + y) * W + x;
    if (s.C1 > 0) {
        const int z1 = s.zmap[z], y1 = s.ymap[y], x1 = s.xmap[x];
        v1 = ((n * s.D1 + z1) * s.H1 + y1) * s.W1 + x1;
    } else {
        v1 = 0;
    }
}

// same, with the index tables bypassed for an exact 2x nearest upsampling (D == 2*D1, H == 2*H1, W == 2*W1), where
// PyTorch's float32 formula min(floor(dst * 0.5f), in-1) is i >> 1 for every i: no dependent loads
__device__ __forceinline__ void u3d_vox_index_x(const u3d_src_t& s, bool exact2x, int n, int z, int y, int x, int D, int H,
                                                int W, int& v0, int& v1) {
    v0 = ((n * D + z) * H + y) * W + x;
    v1 = 0;
    if (s.C1 > 0) {
        const int z1 = exact2x ? z >> 1 : s.zmap[z], y1 = exact2x ? y >> 1 : s.ymap[y], x1 = exact2x ? x >> 1 : s.xmap[x];
        v1 = ((n * s.D1 + z1) * s.H1 + y1) * s.W1 + x1;
    }
}

__device__ __forceinline__ float u3d_load_elem(const u3d_src_t& s, int v0, int v1, int c) {
    if (c < s.C0) return s.p0[(size_t)v0 * s.C0 + c];
    if (c < s.C0 + s.C1) return s.p1[(size_t)v1 * s.C1 + (c - s.C0)];
    return 0.f;
}

// 4 consecutive channels starting at cq (cq % 4 == 0).  vec: C0 % 4 == 0 && C1 % 4 == 0 and 16-B aligned bases.
__device__ __forceinline__ f32x4 u3d_load_quad(const u3d_src_t& s, int v0, int v1, int cq, bool vec) {
    f32x4 r = {0.f, 0.f, 0.f, 0.f};
    if (vec) {
        if (cq < s.C0)
            r = *reinterpret_cast<const f32x4*>(s.p0 + (size_t)v0 * s.C0 + cq);
        else if (cq < s.C0 + s.C1)
            r = *reinterpret_cast<const f32x4*>(s.p1 + (size_t)v1 * s.C1 + (cq - s.C0));
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = u3d_load_elem(s, v0, v1, cq + e);
    }
    return r;
}

// GroupNorm affine (a,b) for 4 consecutive channels of sample n; table [N][Ctot][2].
__device__ __forceinline__ void u3d_load_affine(const float* aff, int n, int Ctot, int cq, bool vec, f32x4& a,
                                                f32x4& b) {
    a = f32x4{0.f, 0.f, 0.f, 0.f};
    b = f32x4{0.f, 0.f, 0.f, 0.f};
    if (aff == nullptr) {
        a = f32x4{1.f, 1.f, 1.f, 1.f};
        return;
    }
    const float* p = aff + ((size_t)n * Ctot + cq) * 2;
    if (vec) {
        if (cq < Ctot) {
            const f32x4 lo = *reinterpret_cast<const f32x4*>(p);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(p + 4);
            a = f32x4{lo[0], lo[2], hi[0], hi[2]};
            b = f32x4{lo[1], lo[3], hi[1], hi[3]};
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (cq + e < Ctot) {
                a[e] = p[2 * e];
                b[e] = p[2 * e + 1];
            }
    }
}

__device__ __forceinline__ void u3d_atomic_add_f64(double* p, double v) {
    // hardware global_atomic_add_f64 on coarse-grained (hipMalloc / torch caching allocator) memory
    unsafeAtomicAdd(p, v);
}

// Bijective XCD-aware remap (cdna_hip_programming.md T1): block b runs on XCD b % 8; give every XCD a
// contiguous run of logical ids so neighbouring tiles (shared halos / shared weights) hit one L2.
__device__ __forceinline__ int u3d_xcd_remap(int b, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = b & 7, idx = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
