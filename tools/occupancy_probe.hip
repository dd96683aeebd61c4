// Do two workgroups with ~69 KB of LDS each share a CU on gfx950?  Queries the occupancy calculator and measures it: 2 x #CU
// workgroups that each spin for a fixed time take 1x that time if two are co-resident per CU, 2x if not.
//   hipcc --offload-arch=gfx950 -O2 tools/occupancy_probe.hip -o tools/occupancy_probe && tools/occupancy_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int LDS_DOUBLES, int NT>
__global__ __launch_bounds__(NT) void spin_kernel(long long cycles, double *out) {
    __shared__ double buf[LDS_DOUBLES];
    buf[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) { }
    if (threadIdx.x == 0) out[blockIdx.x] = buf[(blockIdx.x * 7) % LDS_DOUBLES];
}

template <int LDS_DOUBLES, int NT>
void probe(const char *label, int ncu, double *out) {
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, spin_kernel<LDS_DOUBLES, NT>, NT, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const long long cycles = 20000;                                 // wall_clock64 ticks at 100 MHz: 200 us
    float ms[2];
    for (int mult = 1; mult <= 2; ++mult) {
        hipLaunchKernelGGL((spin_kernel<LDS_DOUBLES, NT>), dim3(ncu * mult), dim3(NT), 0, 0, cycles, out);   // warm
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((spin_kernel<LDS_DOUBLES, NT>), dim3(ncu * mult), dim3(NT), 0, 0, cycles, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[mult - 1], e0, e1);
    }
    printf("%-44s occupancy calculator: %d blocks/CU;  %d blocks: %.3f ms, %d blocks: %.3f ms  -> %s\n", label, occ, ncu, ms[0], 2 * ncu, ms[1],
           ms[1] < 1.5f * ms[0] ? "two workgroups share a CU" : "ONE workgroup per CU");
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount;
    printf("%s: %d CUs, sharedMemPerBlock %zu B, maxSharedMemoryPerMultiProcessor %zu B\n", prop.gcnArchName, ncu, prop.sharedMemPerBlock,
           prop.maxSharedMemoryPerMultiProcessor);
    double *out; hipMalloc(&out, sizeof(double) * ncu * 4);
    probe<4096, 256>("32 KB LDS, 256 threads", ncu, out);
    probe<8000, 256>("62.5 KB LDS, 256 threads", ncu, out);
    probe<8192 + 512, 256>("68 KB LDS, 256 threads", ncu, out);
    probe<8192 + 512, 512>("68 KB LDS, 512 threads", ncu, out);
    probe<10000, 256>("78 KB LDS, 256 threads", ncu, out);
    probe<10240, 256>("80 KB LDS, 256 threads", ncu, out);
    return 0;
}
