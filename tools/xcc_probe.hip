// Which XCD does workgroup b of a 1-D launch land on?  Reads the XCC_ID hardware register per workgroup and prints the
// (blockIdx % 8) x XCC_ID histogram.  Evidence for the blockIdx -> XCD mapping that xcd_remap() (common.h) relies on.
//   hipcc --offload-arch=gfx950 -O2 tools/xcc_probe.hip -o gpurun_out/xcc_probe && gpurun_out/xcc_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void probe(int* xcc, int spin) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    if (threadIdx.x == 0) xcc[blockIdx.x] = (int)(id & 0xF);
    // keep the workgroup alive for a while so that large grids really occupy the chip
    float x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;
    if (x == 12345.f) xcc[0] = -1;
}

int main() {
    for (int nblk : {48, 256, 768, 2048}) {
        for (int threads : {256, 1024}) {
            int* d;
            hipMalloc(&d, nblk * sizeof(int));
            hipLaunchKernelGGL(probe, dim3(nblk), dim3(threads), 0, 0, d, 20000);
            hipDeviceSynchronize();
            int* h = new int[nblk];
            hipMemcpy(h, d, nblk * sizeof(int), hipMemcpyDeviceToHost);
            int hist[8][16] = {};
            for (int b = 0; b < nblk; ++b) hist[b & 7][h[b] & 15]++;
            int off = 0;
            for (int m = 0; m < 8; ++m)
                for (int x = 0; x < 16; ++x)
                    if (x != m) off += hist[m][x];
            printf("grid %5d x %4d threads: workgroups with XCC_ID != blockIdx %% 8: %d of %d\n", nblk, threads, off, nblk);
            if (off) {
                for (int m = 0; m < 8; ++m) {
                    printf("  b%%8=%d:", m);
                    for (int x = 0; x < 8; ++x) printf(" %4d", hist[m][x]);
                    printf("\n");
                }
            }
            delete[] h;
            hipFree(d);
        }
    }
    return 0;
}
