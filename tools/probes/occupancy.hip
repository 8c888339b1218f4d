// tools/probes/occupancy.hip -- resident workgroups per CU as a function of LDS / VGPR use on this device
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDS_KB, int REGS>
__global__ __launch_bounds__(256) void k(float* out) {
    __shared__ float lds[LDS_KB * 256];
    float r[REGS];
    for (int i = 0; i < REGS; i++) r[i] = out[i + threadIdx.x];
    lds[threadIdx.x] = r[0];
    __syncthreads();
    float s = lds[(threadIdx.x + 1) & 255];
    for (int i = 0; i < REGS; i++) s += r[i] * r[(i + 1) % REGS];
    out[threadIdx.x] = s;
}
template <int LDS_KB, int REGS>
void q() {
    int n = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k<LDS_KB, REGS>, 256, 0);
    hipFuncAttributes a;
    hipFuncGetAttributes(&a, (const void*)k<LDS_KB, REGS>);
    printf("LDS %3d KB, ~%3d regs (numRegs %d): %d workgroups of 256 per CU\n", LDS_KB, REGS, a.numRegs, n);
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("%s: sharedMemPerBlock %zu, maxSharedMemoryPerMultiProcessor %zu, regsPerBlock %d, CUs %d\n", p.gcnArchName,
           p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor, p.regsPerBlock, p.multiProcessorCount);
    q<16, 32>(); q<32, 32>(); q<48, 32>(); q<64, 32>(); q<40, 32>(); q<32, 100>(); q<32, 120>(); q<48, 120>(); q<20,120>();
    return 0;
}
