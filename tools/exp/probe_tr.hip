// probe: semantics of ds_read_b64_tr_b16 on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void probe(uint16_t* out, int mode) {
    __shared__ uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x;
    unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) uint16_t*)lds;
    if (mode == 0) addr += l * 8;                      // what a plain b64 read would use
    if (mode == 1) addr += 0;                          // uniform
    if (mode == 2) addr += (l & 15) * 2 + (l >> 4) * 128;   // column l&15 of a [4][16] block per 16-lane group
    if (mode == 3) addr += (l & 15) * 64 + (l >> 4) * 8;    // row l&15 (stride 64 B), 4 consecutive elements, groups shift 4 elems
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[l * 4 + 0] = v.x & 0xffff; out[l * 4 + 1] = v.x >> 16; out[l * 4 + 2] = v.y & 0xffff; out[l * 4 + 3] = v.y >> 16;
}
int main() {
    uint16_t* d; hipMalloc(&d, 512);
    uint16_t h[256];
    for (int mode = 0; mode < 4; ++mode) {
        probe<<<1, 64>>>(d, mode); hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) { printf("  l%02d: %4d %4d %4d %4d", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]); if (l % 4 == 3) printf("\n"); }
    }
    return 0;
}
