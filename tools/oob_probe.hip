// Developer probe: does the range check of a raw buffer descriptor include the SGPR offset on gfx950?
// A 256-byte descriptor over a 1-KiB array of 0xABABABAB words; loads at voffset / soffset inside and outside the range.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void lds_void;
__global__ void k(const unsigned *p, unsigned *out) {
    __shared__ unsigned lds[256];
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, 256, 0x00020000);
    lds[threadIdx.x] = 0x11111111u; lds[threadIdx.x + 64] = 0x11111111u; lds[threadIdx.x + 128] = 0x11111111u; lds[threadIdx.x + 192] = 0x11111111u;
    __syncthreads();
    out[0] = __builtin_amdgcn_raw_buffer_load_b32(r, 0, 0, 0);        // inside
    out[1] = __builtin_amdgcn_raw_buffer_load_b32(r, 512, 0, 0);      // voffset outside
    out[2] = __builtin_amdgcn_raw_buffer_load_b32(r, 0, 512, 0);      // soffset outside
    out[3] = __builtin_amdgcn_raw_buffer_load_b32(r, 128, 256, 0);    // sum outside
    // the same through the LDS-DMA form: lane 0's dword lands at lds[0] / lds[64]
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void *)lds, 4, 0, 512, 0, 0);          // soffset outside
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void *)(lds + 64), 4, 0, 0, 0, 0);     // inside
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    out[4] = lds[0]; out[5] = lds[64];
}
int main() {
    unsigned *p, *o, h[256], ho[8];
    for (int i = 0; i < 256; ++i) h[i] = 0xABABABABu;
    hipMalloc(&p, 1024); hipMalloc(&o, 32); hipMemcpy(p, h, 1024, hipMemcpyHostToDevice);
    k<<<1, 64>>>(p, o); hipMemcpy(ho, o, 32, hipMemcpyDeviceToHost);
    printf("inside %08x | voffset out %08x | soffset out %08x | voffset+soffset out %08x | DMA soffset out %08x | DMA inside %08x\n", ho[0], ho[1], ho[2], ho[3], ho[4], ho[5]);
    printf("(0 = dropped by the range check, abababab = read past the descriptor, 11111111 = LDS untouched)\n");
    return 0;
}
