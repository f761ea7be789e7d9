// Probe (gfx950): lane mapping of v_permlane16_swap_b32 as hipcc's __builtin_amdgcn_permlane16_swap(a, b, fi, bc) returns it.
// Expected (CDNA4 ISA): the ODD 16-lane rows of the first operand are exchanged with the EVEN rows of the second --
//   r[0]: rows 0, 2 = a rows 0, 2;  rows 1, 3 = b rows 0, 2        r[1]: rows 0, 2 = a rows 1, 3;  rows 1, 3 = b rows 1, 3
// which is what the halo GEMM's epilogue uses to turn two fragments' 8-byte (pixel, 4 channels) pieces into 16-byte stores.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/permlane16_swap.hip -o /tmp/pl16 && /tmp/pl16
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void probe(unsigned* out) {
    const unsigned lane = threadIdx.x;
    const unsigned a = 0xA000u | lane, b = 0xB000u | lane;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[lane] = r[0];
    out[64 + lane] = r[1];
}

int main() {
    unsigned* out; unsigned h[128];
    hipMalloc(&out, sizeof(h));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out);
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int row = l >> 4, fr = l & 15;
        const unsigned e0 = (row & 1) ? (0xB000u | ((row - 1) * 16 + fr)) : (0xA000u | l);
        const unsigned e1 = (row & 1) ? (0xB000u | l) : (0xA000u | ((row + 1) * 16 + fr));
        bad += h[l] != e0 || h[64 + l] != e1;
    }
    for (int l : {0, 15, 16, 31, 32, 47, 48, 63}) printf("lane %2d: r0 %04x r1 %04x\n", l, h[l], h[64 + l]);
    printf("permlane16_swap mapping %s\n", bad ? "DIFFERS from the expected one" : "as expected");
    return bad ? 1 : 0;
}
