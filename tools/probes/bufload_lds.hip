// Probe (gfx950): does `buffer_load_dwordx4 ... offen lds` write ZEROS to LDS for lanes whose offset is out of the buffer's range?
// The out-of-range offsets stay inside the real allocation, so a missing range check shows as sentinel bytes, not as a fault.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/bufload_lds.hip -o tools/probes/bufload_probe && tools/probes/bufload_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef int rsrc_t __attribute__((ext_vector_type(4)));

__global__ void probe(const char* src, uint32_t records, uint32_t* out, int mode) {
    __shared__ __attribute__((aligned(16))) char lds[2048];
    const int lane = threadIdx.x;
    for (int i = lane; i < 512; i += 64) reinterpret_cast<uint32_t*>(lds)[i] = 0xdeadbeefu;
    __syncthreads();
    const uint64_t b = (uint64_t)(uintptr_t)src;
    rsrc_t r;
    r[0] = __builtin_amdgcn_readfirstlane((uint32_t)b);
    r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32) & 0xffff);      // stride 0, no swizzle
    r[2] = __builtin_amdgcn_readfirstlane(records);
    r[3] = 0x00020000;
    uint32_t voff;
    if (mode == 0) voff = lane < 32 ? lane * 16 : 8192 + lane * 16;            // half the lanes out of range (but inside the allocation)
    else if (mode == 1) voff = (lane & 1) ? 0x40000000u + lane * 16 : lane * 16;   // far out of range: never dereferenced if the check works
    else voff = lane * 16 + 3 * 1024;                                          // all in range, 3 KB in
    const uint32_t m = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds + 1024);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds\n\ts_waitcnt vmcnt(0)" ::"v"(voff), "s"(r), "s"(m) : "memory", "m0");
    __syncthreads();
    for (int i = lane; i < 512; i += 64) out[i] = reinterpret_cast<uint32_t*>(lds)[i];
}

int main(int argc, char** argv) {
    const int modes = argc > 1 ? atoi(argv[1]) : 1;      // 1: only the safe mode 0 (+2); 3: also the far offsets
    char* src; uint32_t* out;
    const size_t bytes = 1 << 20;
    hipMalloc(&src, bytes); hipMalloc(&out, 2048);
    std::vector<uint32_t> h(bytes / 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x11000000u | (uint32_t)i;     // dword i holds its own index
    hipMemcpy(src, h.data(), bytes, hipMemcpyHostToDevice);
    std::vector<uint32_t> o(512);
    for (int mode : {0, 2, 1}) {
        if (mode == 1 && modes < 3) continue;
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, src, 4096u, out, mode);
        hipDeviceSynchronize();
        hipMemcpy(o.data(), out, 2048, hipMemcpyDeviceToHost);
        printf("mode %d: untouched first KB %s\n", mode, o[0] == 0xdeadbeefu && o[255] == 0xdeadbeefu ? "yes" : "NO");
        for (int lane : {0, 1, 2, 31, 32, 33, 62, 63}) printf("  lane %2d -> %08x %08x %08x %08x\n", lane, o[256 + lane * 4], o[256 + lane * 4 + 1], o[256 + lane * 4 + 2], o[256 + lane * 4 + 3]);
    }
    return 0;
}
