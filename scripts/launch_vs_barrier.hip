// launch_vs_barrier.hip -- what does a persistent single-window kernel have to beat?  (VERDICT r3 task 4: "build the persistent
// kernel or show with measurements why it cannot beat 85 launches")
//
// One isv_batch_optimize of ONE window is a chain of 85 dependent kernel launches; its rocprofv3 kernel trace shows ~1.8 us between
// the end of a kernel and the start of the next.  A persistent kernel replaces a launch boundary by a device-side barrier: between
// phases that use ONE workgroup that is a __syncthreads (free), but the phases that can use many CUs (linearise + Gram products,
// rank-1 downdates, back-substitution, candidate evaluation) need a GRID barrier across their workgroups, twice per phase (fan out,
// fan in).  This program measures both on the device it runs on:
//   (a) the cost of a kernel boundary: K back-to-back launches of a kernel that does nothing, G workgroups x 256 threads;
//   (b) the cost of a grid barrier: ONE launch of G resident workgroups that cross K barriers (monotonic arrive counter +
//       spin on it, release / acquire at agent scope -- what a hand-written phase flag has to do);
//   (c) the same with 64 bytes written before and read after every barrier by a different workgroup (the hand-over of a phase's
//       result through L2: at agent scope the writer's L2 write-back and the reader's invalidate are part of the price).
// build:  hipcc -O3 --offload-arch=gfx950 -o launch_vs_barrier scripts/launch_vs_barrier.hip      run: ./launch_vs_barrier
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_nothing(int *p) { if (p && threadIdx.x == 1000) *p = 1; }

// K grid barriers across gridDim.x resident workgroups
__global__ __launch_bounds__(256) void k_barriers(unsigned *counter, double *buf, int K, int handover) {
    const unsigned G = gridDim.x;
    double acc = 0;
    for (int k = 0; k < K; k++) {
        if (handover && threadIdx.x < 8) buf[((size_t)k & 1) * 8 * G + blockIdx.x * 8 + threadIdx.x] = acc + k;     // this phase's "result"
        __syncthreads();
        if (threadIdx.x == 0) {
            __atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE);                      // arrive (agent scope: system-coherent atomics on global memory)
            const unsigned target = (unsigned)(k + 1) * G;
            while (__atomic_load_n(counter, __ATOMIC_ACQUIRE) < target) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        if (handover && threadIdx.x < 8) acc += buf[((size_t)k & 1) * 8 * G + ((blockIdx.x + 1) % G) * 8 + threadIdx.x];   // the neighbour's
    }
    if (acc == 12345.678 && threadIdx.x == 0) buf[0] = acc;
}

int main() {
    int dev = 0, cus = 0;
    CHK(hipGetDevice(&dev));
    CHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    hipStream_t st;
    CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned *counter; double *buf; int *flag;
    CHK(hipMalloc(&counter, 4)); CHK(hipMalloc(&buf, sizeof(double) * 2 * 8 * 256)); CHK(hipMalloc(&flag, 4));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int K = 2000;
    printf("{\"device_cus\": %d, \"K\": %d, \"rows\": [\n", cus, K);
    const int Gs[] = {1, 4, 16, 32, 64};
    for (int gi = 0; gi < 5; gi++) {
        const int G = Gs[gi];
        float ms_launch = 0, ms_bar = 0, ms_bar_h = 0;
        for (int rep = 0; rep < 3; rep++) {             // (the last repetition is kept)
            CHK(hipEventRecord(e0, st));
            for (int k = 0; k < K; k++) hipLaunchKernelGGL(k_nothing, dim3(G), dim3(256), 0, st, flag);
            CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms_launch, e0, e1));
            for (int h = 0; h < 2; h++) {
                CHK(hipMemsetAsync(counter, 0, 4, st));
                CHK(hipEventRecord(e0, st));
                hipLaunchKernelGGL(k_barriers, dim3(G), dim3(256), 0, st, counter, buf, K, h);
                CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(h ? &ms_bar_h : &ms_bar, e0, e1));
            }
        }
        printf("  {\"workgroups\": %d, \"us_per_kernel_boundary\": %.3f, \"us_per_grid_barrier\": %.3f, \"us_per_grid_barrier_with_64B_handover\": %.3f}%s\n",
               G, 1e3 * ms_launch / K, 1e3 * ms_bar / K, 1e3 * ms_bar_h / K, gi < 4 ? "," : "");
    }
    printf("]}\n");
    return 0;
}
