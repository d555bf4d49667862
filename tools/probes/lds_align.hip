// Microbenchmark: LDS pipeline cycles per wave instruction for reads of 4, 8 and 16 bytes at
// addresses aligned to 16, 8, 4 bytes and at odd addresses (gfx950 takes unaligned LDS accesses;
// this measures what they cost).  Build: hipcc --offload-arch=gfx950 -O3 lds_align.hip -o lds_align
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <int BYTES>
__global__ __launch_bounds__(256) void probe(uint32_t *out, uint32_t stride, uint32_t off, int iters) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    for (uint32_t i = threadIdx.x; i < 16384; i += 256) ((uint32_t *)lds)[i] = i * 2654435761u;
    __syncthreads();
    uint32_t acc = 0;
    uint32_t p = (threadIdx.x * stride + off) & 0x7fff;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t a = (p + k * 1040) & 0xffff;
            const uint32_t addr = (uint32_t)(size_t)lds + a;
            if (BYTES == 4) {
                uint32_t v;
                asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr) : "memory");
                asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                acc += v;
            } else if (BYTES == 8) {
                u32x2 v;
                asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr) : "memory");
                asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                acc += v[0] ^ v[1];
            } else if (BYTES == 16) {
                u32x4 v;
                asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
                asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                acc += v[0] ^ v[1] ^ v[2] ^ v[3];
            } else if (BYTES == 32) {
                u32x2 v;
                asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(v) : "v"(addr) : "memory");
                asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                acc += v[0] ^ v[1];
            } else if (BYTES == 12) {
                typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
                u32x3 v;
                asm volatile("ds_read_b96 %0, %1" : "=v"(v) : "v"(addr) : "memory");
                asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                acc += v[0] ^ v[1] ^ v[2];
            } else if (BYTES == 104) {
                asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(acc) : "memory");
                asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            } else if (BYTES == 101) {
                asm volatile("ds_write_b8 %0, %1" :: "v"(addr), "v"(acc) : "memory");
                asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            } else if (BYTES == 108) {
                asm volatile("ds_write2_b32 %0, %1, %1 offset1:1" :: "v"(addr), "v"(acc) : "memory");
                asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            } else {
                uint32_t v;
                asm volatile("ds_read_u8 %0, %1" : "=v"(v) : "v"(addr) : "memory");
                asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                acc += v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        p = (p + 64 + (acc & 0)) & 0x7fff;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int BYTES>
static double run(uint32_t *d_out, uint32_t stride, uint32_t off) {
    const int blocks = 256 * 2, iters = 2000;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    probe<BYTES><<<blocks, 256, 65536 + 64>>>(d_out, stride, off, 10);
    hipEventRecord(a);
    probe<BYTES><<<blocks, 256, 65536 + 64>>>(d_out, stride, off, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    // wave instructions per CU: blocks/256 CUs * 4 waves * iters * 8
    const double per_cu = (double)blocks / 256 * 4 * iters * 8;
    return ms * 1e-3 * 2.4e9 / per_cu; // cycles of a 2.4 GHz clock per wave instruction per CU
}

int main() {
    uint32_t *d_out;
    hipMalloc(&d_out, 256 * 2 * 256 * 4);
    hipFuncSetAttribute((const void *)probe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 64);
    hipFuncSetAttribute((const void *)probe<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 64);
    hipFuncSetAttribute((const void *)probe<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 64);
    hipFuncSetAttribute((const void *)probe<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 64);
    hipFuncSetAttribute((const void *)probe<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 64);
    hipFuncSetAttribute((const void *)probe<12>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 64);
    hipFuncSetAttribute((const void *)probe<104>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 64);
    hipFuncSetAttribute((const void *)probe<101>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 64);
    hipFuncSetAttribute((const void *)probe<108>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 64);
    const uint32_t strides[] = {16, 172, 4, 7};
    for (uint32_t st : strides) {
        printf("lane stride %u bytes\n", st);
        for (uint32_t off : {0u, 8u, 4u, 2u, 1u}) {
            printf("  offset %u: u8 %.1f  b32 %.1f  b64 %.1f  b128 %.1f  read2_b32 %.1f  b96 %.1f | write b8 %.1f  b32 %.1f  write2_b32 %.1f cycles/instr\n",
                   off, run<1>(d_out, st, off), run<4>(d_out, st, off), run<8>(d_out, st, off), run<16>(d_out, st, off),
                   run<32>(d_out, st, off), run<12>(d_out, st, off), run<101>(d_out, st, off), run<104>(d_out, st, off),
                   run<108>(d_out, st, off));
            fflush(stdout);
        }
    }
    return 0;
}
