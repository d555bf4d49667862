// Probe: do ds_read_b64 / ds_read_b128 / ds_write_b64 honour byte-unaligned LDS addresses on
// this GPU + driver?  (Unsupported hardware silently rounds the address down: no fault.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
__global__ void k(unsigned long long *out64, unsigned *out128, unsigned long long *outw) {
    __shared__ __attribute__((aligned(16))) unsigned char s[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) s[i] = (unsigned char)(i * 7 + 3);
    __syncthreads();
    unsigned addr = (unsigned)(size_t)(s) + threadIdx.x;  // LDS byte address, unaligned for most lanes
    unsigned long long v;
    asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out64[threadIdx.x] = v;
    typedef unsigned v4 __attribute__((ext_vector_type(4)));
    v4 q;
    asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(addr) : "memory");
    out128[threadIdx.x * 4 + 0] = q.x; out128[threadIdx.x * 4 + 1] = q.y;
    out128[threadIdx.x * 4 + 2] = q.z; out128[threadIdx.x * 4 + 3] = q.w;
    __syncthreads();
    // unaligned 8-byte store: lane 5 only
    if (threadIdx.x == 5) {
        unsigned a2 = (unsigned)(size_t)(s) + 512 + 3;
        unsigned long long w = 0x1122334455667788ull;
        asm volatile("ds_write_b64 %0, %1\n s_waitcnt lgkmcnt(0)" :: "v"(a2), "v"(w) : "memory");
    }
    __syncthreads();
    if (threadIdx.x < 4) { unsigned long long r; memcpy(&r, s + 512 + threadIdx.x * 8, 8); outw[threadIdx.x] = r; }
}
int main() {
    unsigned long long *d64, *dw; unsigned *d128;
    hipMalloc(&d64, 64 * 8); hipMalloc(&d128, 64 * 16); hipMalloc(&dw, 32);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d64, d128, dw);
    unsigned long long h64[64], hw[4]; unsigned h128[256];
    hipMemcpy(h64, d64, sizeof h64, hipMemcpyDeviceToHost);
    hipMemcpy(h128, d128, sizeof h128, hipMemcpyDeviceToHost);
    hipMemcpy(hw, dw, sizeof hw, hipMemcpyDeviceToHost);
    int bad64 = 0, bad128 = 0;
    for (int l = 0; l < 64; ++l) {
        unsigned char e[16]; for (int i = 0; i < 16; ++i) e[i] = (unsigned char)((l + i) * 7 + 3);
        if (memcmp(e, &h64[l], 8)) ++bad64;
        if (memcmp(e, &h128[l * 4], 16)) ++bad128;
    }
    unsigned char exp[32]; for (int i = 0; i < 32; ++i) exp[i] = (unsigned char)((512 + i) * 7 + 3);
    unsigned long long w = 0x1122334455667788ull; memcpy(exp + 3, &w, 8);
    printf("ds_read_b64 unaligned: %s (%d bad lanes)\n", bad64 ? "NOT honoured" : "ok", bad64);
    printf("ds_read_b128 unaligned: %s (%d bad lanes)\n", bad128 ? "NOT honoured" : "ok", bad128);
    printf("ds_write_b64 unaligned: %s\n", memcmp(exp, hw, 32) ? "NOT honoured" : "ok");
    return 0;
}
