// L2 -> CU delivery rate by path, on an L2-resident working set (what the H x H GEMM operands are): which path feeds a CU fastest?
//   dma : global_load_lds_dwordx4 into an LDS ring (what gemm16* does today)
//   reg : global_load_dwordx4 into VGPRs (consumed by a cheap xor)
//   rds : global_load_dwordx4 -> VGPR -> ds_write_b128 (register-staged LDS fill)
//   mix : waves 0,1 dma + waves 2,3 reg (do the two paths add up?)
// build: hipcc --offload-arch=gfx950 -O3 -o fill_bench fill_bench.hip ; run: ./fill_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int PIECE = 1024;              // bytes per wave-instruction (64 lanes x 16 B)

// each workgroup streams `bytes_per_wg` from buf (wrapping inside `span` bytes), wave w taking pieces w, w+4, ...
template <int MODE, int DEPTH, int NWV = 4>        // NWV waves per workgroup
__global__ __launch_bounds__(64 * NWV) void fill_kernel(const unsigned char* __restrict__ buf, size_t span, size_t bytes_per_wg, unsigned* sink, int lds_bytes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t npieces = bytes_per_wg / PIECE / NWV;        // per wave
    size_t off = ((size_t)blockIdx.x * 7919 * PIECE * NWV + (size_t)wave * PIECE) % span;
    uint4 acc = {0, 0, 0, 0};
    const bool dma = MODE == 0 || (MODE == 3 && wave < 2);
    const bool stage = MODE == 2;
    unsigned char* ring = smem + wave * (lds_bytes / NWV);
    const int ring_pieces = lds_bytes / NWV / PIECE;
    int slot = 0;
    if (dma) {
        for (size_t i = 0; i < npieces; i += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                __builtin_amdgcn_global_load_lds((const void*)(buf + off + lane * 16), (lds_void*)(ring + slot * PIECE), 16, 0, 0);
                off += NWV * PIECE; if (off >= span) off -= span;
                slot = slot + 1 == ring_pieces ? 0 : slot + 1;
            }
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DEPTH / 2) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc.x = *reinterpret_cast<unsigned*>(ring + lane * 4);
    } else {
        for (size_t i = 0; i < npieces; i += DEPTH) {
            uint4 v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                v[d] = *reinterpret_cast<const uint4*>(buf + off + lane * 16);
                off += NWV * PIECE; if (off >= span) off -= span;
            }
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                if (stage) {
                    *reinterpret_cast<uint4*>(ring + slot * PIECE + lane * 16) = v[d];
                    slot = slot + 1 == ring_pieces ? 0 : slot + 1;
                } else {
                    acc.x ^= v[d].x; acc.y ^= v[d].y; acc.z ^= v[d].z; acc.w ^= v[d].w;
                }
            }
        }
        if (stage) { __syncthreads(); acc.x = *reinterpret_cast<unsigned*>(ring + lane * 4); }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

template <int MODE, int DEPTH, int NWV = 4>
static void run(const char* name, const unsigned char* buf, size_t span, int wgs_per_cu, int lds_bytes, unsigned* sink) {
    const int cus = 256, grid = cus * wgs_per_cu;
    const size_t per_wg = (size_t)16 << 20;      // 16 MB streamed per workgroup
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&fill_kernel<MODE, DEPTH, NWV>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((fill_kernel<MODE, DEPTH, NWV>), dim3(grid), dim3(64 * NWV), lds_bytes, 0, buf, span, per_wg, sink, lds_bytes);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
    }
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double gbs = (double)per_wg * grid / (ms * 1e-3) / 1e9;
    printf("%-4s depth %2d  %d waves/wg  %d wg/cu  lds %3d KB/wg  span %5.1f MB : %7.1f GB/s per CU  (%6.2f TB/s chip)  %.3f ms\n", name, DEPTH, NWV, wgs_per_cu, lds_bytes >> 10,
           span / 1048576.0, gbs / cus, gbs / 1e3, ms);
}

int main() {
    unsigned char* buf;
    unsigned* sink;
    const size_t cap = (size_t)512 << 20;
    CK(hipMalloc(&buf, cap)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 1, cap)); CK(hipMemset(sink, 0, 64));
    for (size_t span : {(size_t)2 << 20, (size_t)24 << 20, (size_t)400 << 20}) {     // L2-resident, Infinity-Cache-resident, HBM
        printf("---- working set %zu MB\n", span >> 20);
        run<0, 8>("dma", buf, span, 1, 64 << 10, sink);
        run<0, 16>("dma", buf, span, 1, 128 << 10, sink);           // one workgroup per CU: more pieces in flight per wave ...
        run<0, 32>("dma", buf, span, 1, 128 << 10, sink);
        run<0, 8, 8>("dma", buf, span, 1, 128 << 10, sink);         // ... or more waves issuing
        run<0, 16, 8>("dma", buf, span, 1, 128 << 10, sink);
        run<0, 8, 16>("dma", buf, span, 1, 128 << 10, sink);
        run<0, 8>("dma", buf, span, 2, 64 << 10, sink);
        run<0, 16>("dma", buf, span, 2, 64 << 10, sink);
        run<0, 16>("dma", buf, span, 4, 32 << 10, sink);
        run<1, 8>("reg", buf, span, 1, 16 << 10, sink);
        run<1, 8>("reg", buf, span, 2, 16 << 10, sink);
        run<1, 16>("reg", buf, span, 2, 16 << 10, sink);
        run<1, 8>("reg", buf, span, 4, 16 << 10, sink);
        run<2, 8>("rds", buf, span, 2, 64 << 10, sink);
        run<2, 16>("rds", buf, span, 2, 64 << 10, sink);
        run<3, 8>("mix", buf, span, 2, 64 << 10, sink);
        run<3, 16>("mix", buf, span, 2, 64 << 10, sink);
    }
    return 0;
}
