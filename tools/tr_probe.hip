// Probe of ds_read_b64_tr_b16 (gfx950 LDS transpose read) semantics: every lane supplies its own 8-byte address; the probe
// prints, for each (lane, element), which (source lane, source element) it received — with identity addresses and with
// permuted addresses — to confirm it is a pure 16-lane-group shuffle: out[l][j] = in[16*(l>>4) + 4*j + ((l&15)>>2)][(l&15)&3].
//   hipcc --offload-arch=gfx950 -O2 tools/tr_probe.hip -o tools/bin/tr_probe && tools/bin/tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const int* slot_of_lane, short* out) {
    __shared__ short lds[64 * 4];
    int t = threadIdx.x;
    for (int e = 0; e < 4; ++e) lds[t * 4 + e] = (short)(t * 4 + e);  // value = slot*4 + element
    __syncthreads();
    const int slot = slot_of_lane[t];
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + slot * 4));
    for (int j = 0; j < 4; ++j) out[t * 4 + j] = v[j];
}
int main() {
    int h_slot[64];
    short h_out[256];
    int* d_slot;
    short* d_out;
    hipMalloc(&d_slot, sizeof(h_slot));
    hipMalloc(&d_out, sizeof(h_out));
    for (int variant = 0; variant < 2; ++variant) {
        for (int l = 0; l < 64; ++l) h_slot[l] = variant == 0 ? l : (l * 37 + 5) % 64;  // identity / a permutation
        hipMemcpy(d_slot, h_slot, sizeof(h_slot), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_slot, d_out);
        hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 4; ++j) {
                const int src_lane = 16 * (l >> 4) + 4 * j + ((l & 15) >> 2), src_e = (l & 15) & 3;
                const int expect = h_slot[src_lane] * 4 + src_e;
                if (h_out[l * 4 + j] != expect) ++bad;
            }
        printf("variant %d: %d mismatches vs out[l][j] = in[16*(l>>4) + 4*j + ((l&15)>>2)][(l&15)&3]\n", variant, bad);
        if (bad) {
            for (int l = 0; l < 64; ++l) {
                printf("lane %2d:", l);
                for (int j = 0; j < 4; ++j) printf(" (slot %2d, e %d)", h_out[l * 4 + j] / 4, h_out[l * 4 + j] % 4);
                printf("\n");
            }
        }
    }
    return 0;
}
