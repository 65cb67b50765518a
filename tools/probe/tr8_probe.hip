// Probe: lane mapping of ds_read_b64_tr_b8 (gfx950).  Lane l supplies the LDS address of an 8-byte chunk; the output
// shows, for every lane, which (source lane, byte) each of its 8 result bytes came from.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v2i __attribute__((ext_vector_type(2)));
__global__ void probe(uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t tag[64 * 8];     // tag per byte would need 9 bits: use two passes
    __shared__ __attribute__((aligned(16))) uint8_t lo[64 * 8], hi[64 * 8];
    const int l = threadIdx.x;
    for (int e = 0; e < 8; ++e) { lo[l * 8 + e] = (uint8_t)((l * 8 + e) & 0xff); hi[l * 8 + e] = (uint8_t)((l * 8 + e) >> 8); }
    __syncthreads();
    v2i a = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i*)(lo + l * 8));
    v2i b = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i*)(hi + l * 8));
    for (int e = 0; e < 8; ++e) {
        const int va = (a[e >> 2] >> ((e & 3) * 8)) & 0xff, vb = (b[e >> 2] >> ((e & 3) * 8)) & 0xff;
        out[l * 8 + e] = (uint16_t)(va | (vb << 8));
    }
}
int main() {
    uint16_t* d; hipMalloc(&d, 64 * 8 * 2);
    probe<<<1, 64>>>(d);
    uint16_t h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int e = 0; e < 8; ++e) printf("  (L%2d,b%d)", h[l * 8 + e] >> 3, h[l * 8 + e] & 7);
        printf("\n");
    }
    return 0;
}
