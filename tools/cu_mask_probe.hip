// tools/cu_mask_probe.hip — which CUs does a CU-masked HIP stream run on?  Establishes the bit -> (XCD, CU) layout that
// u3d_streams_create_reserved (csrc/u3d_ops.hip) relies on: for a few masks, launch a grid of short-spinning blocks and record
// (XCC_ID, HW_ID) per block.     hipcc --offload-arch=gfx950 -O2 tools/cu_mask_probe.hip -o tools/bin/cu_mask_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <map>
#include <set>
#include <vector>

__global__ void where_kernel(unsigned* out, int spin) {
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x + 0] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
        out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID
    }
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(32);
}

#define CK(x)                                                                       \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            printf("%s failed: %s\n", #x, hipGetErrorString(e_));                   \
            return 1;                                                               \
        }                                                                           \
    } while (0)

static int run(const char* name, hipStream_t s, unsigned* dbuf, int blocks) {
    std::vector<unsigned> h(2 * blocks);
    hipLaunchKernelGGL(where_kernel, dim3(blocks), dim3(64), 0, s, dbuf, 2000);
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h.data(), dbuf, h.size() * 4, hipMemcpyDeviceToHost));
    std::map<unsigned, std::set<unsigned>> cus;  // xcc -> {(se, cu)}
    for (int b = 0; b < blocks; ++b) {
        const unsigned xcc = h[2 * b] & 0xf, hw = h[2 * b + 1];
        cus[xcc].insert((hw >> 8) & 0xff);  // CU_ID [11:8], SH_ID [12], SE_ID [15:13]
    }
    int total = 0;
    printf("%-28s:", name);
    for (auto& kv : cus) {
        printf(" xcc%u=%zu", kv.first, kv.second.size());
        total += (int)kv.second.size();
    }
    printf("  -> %d distinct CUs\n", total);
    return 0;
}

__global__ void empty_kernel() {}
__global__ void busy_kernel(float* p, int n) {
    float a = p[threadIdx.x & 63];
    for (int i = 0; i < n; ++i) a = a * 1.0001f + 0.5f;
    if (a == 123.f) p[0] = a;
}

// per-launch cost of back-to-back launches on a stream: `n` empty kernels, and `n` kernels of ~50 us that fill every CU
static int launch_cost(const char* name, hipStream_t s, float* buf) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int busy = 0; busy < 2; ++busy) {
        const int n = busy ? 200 : 2000;
        for (int rep = 0; rep < 2; ++rep) {  // first repetition warms up
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < n; ++i) {
                if (busy)
                    hipLaunchKernelGGL(busy_kernel, dim3(1024), dim3(256), 0, s, buf, 20000);
                else
                    hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s);
            }
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("%-28s: %s %.2f us per launch\n", name, busy ? "chip-filling kernel," : "empty kernel,       ", 1e3 * ms / n);
        }
    }
    return 0;
}

int main() {
    int ncu = 0;
    CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    printf("multiProcessorCount %d\n", ncu);
    const int blocks = 8192, words = (ncu + 31) / 32;
    unsigned* dbuf;
    CK(hipMalloc(&dbuf, 2 * blocks * 4));
    hipStream_t plain;
    CK(hipStreamCreate(&plain));
    run("unmasked", plain, dbuf, blocks);
    {
        float* fb;
        CK(hipMalloc(&fb, 4096));
        CK(hipMemset(fb, 0, 4096));
        hipStream_t nb, masked, maskedall;
        CK(hipStreamCreateWithFlags(&nb, hipStreamNonBlocking));
        std::vector<uint32_t> m8(words, 0), mall(words, 0);
        for (int i = 0; i < ncu; ++i) {
            mall[i >> 5] |= 1u << (i & 31);
            if (i < ncu - 8) m8[i >> 5] |= 1u << (i & 31);
        }
        CK(hipExtStreamCreateWithCUMask(&masked, words, m8.data()));
        CK(hipExtStreamCreateWithCUMask(&maskedall, words, mall.data()));
        launch_cost("hipStreamCreate (blocking)", plain, fb);
        launch_cost("hipStreamNonBlocking", nb, fb);
        launch_cost("CU mask, all bits set", maskedall, fb);
        launch_cost("CU mask, 8 CUs reserved", masked, fb);
    }
    struct Case {
        const char* name;
        int lo, hi;  // bits [lo, hi) SET
    } cases[] = {{"bits [0, ncu-8)", 0, ncu - 8},   {"bits [ncu-8, ncu)", ncu - 8, ncu}, {"bits [0, 8)", 0, 8},
                 {"bits [0, 32)", 0, 32},           {"bits [0, ncu-16)", 0, ncu - 16},   {"bits [0, ncu-1)", 0, ncu - 1},
                 {"bit 0", 0, 1},                   {"bit 1", 1, 2},                     {"bit 8", 8, 9}};
    for (auto& c : cases) {
        std::vector<uint32_t> mask(words, 0);
        for (int i = c.lo; i < c.hi; ++i) mask[i >> 5] |= 1u << (i & 31);
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, words, mask.data());
        if (e != hipSuccess) {
            printf("%-28s: hipExtStreamCreateWithCUMask failed: %s\n", c.name, hipGetErrorString(e));
            continue;
        }
        run(c.name, s, dbuf, blocks);
        std::vector<uint32_t> back(words, 0);
        if (hipExtStreamGetCUMask(s, words, back.data()) == hipSuccess) {
            int bits = 0;
            for (auto w : back) bits += __builtin_popcount(w);
            printf("%-28s  hipExtStreamGetCUMask: %d bits set\n", "", bits);
        }
        CK(hipStreamDestroy(s));
    }
    return 0;
}
