// Microbenchmark: cost of a device-wide barrier inside a persistent kernel on MI355X, against back-to-back dependent launches.
// Build: hipcc --offload-arch=gfx950 -O3 -o grid_barrier_mb grid_barrier_mb.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ bool barrier_flat(unsigned* ctr, unsigned target) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) { ok = false; break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return ok;
}

// hierarchical: 8 group counters (blockIdx.x & 7), last arriver of a group bumps the global counter
__device__ __forceinline__ bool barrier_hier(unsigned* ctr, unsigned step, unsigned per_group_n) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        const unsigned g = blockIdx.x & 7;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned old = __hip_atomic_fetch_add(ctr + 32 * (1 + g), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == step * per_group_n) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < step * 8) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) { ok = false; break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return ok;
}

template <int MODE>
__global__ __launch_bounds__(256) void persist(unsigned* ctr, int nsteps, float* data, int work, unsigned* fail) {
    float acc = 0.f;
    for (int s = 1; s <= nsteps; ++s) {
        // a little dependent memory traffic per step, as a conv step would have: write own slot, read a neighbour's after the barrier
        if (work) data[(size_t)blockIdx.x * 256 + threadIdx.x] = acc + (float)s;
        bool ok;
        if (MODE == 0) ok = barrier_flat(ctr, (unsigned)s * gridDim.x);
        else ok = barrier_hier(ctr, (unsigned)s, gridDim.x / 8);
        if (!ok) { if (threadIdx.x == 0) atomicAdd(fail, 1u); return; }
        if (work) acc += data[(size_t)((blockIdx.x + 37) % gridDim.x) * 256 + threadIdx.x];
    }
    if (work) data[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void tiny(float* data, int s) {
    float v = data[(size_t)((blockIdx.x + 37) % gridDim.x) * 256 + threadIdx.x];
    data[(size_t)blockIdx.x * 256 + threadIdx.x + (size_t)gridDim.x * 256 * (s & 1 ? 0 : 0)] = v + (float)s;
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs %d\n", p.name, p.multiProcessorCount);
    unsigned* ctr; float* data; unsigned* fail;
    CK(hipMalloc(&ctr, 4096 * 4)); CK(hipMalloc(&data, 4096 * 256 * 4)); CK(hipMalloc(&fail, 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int nsteps = 2000;
    for (int work = 0; work < 2; ++work)
        for (int mode = 0; mode < 2; ++mode)
            for (int grid : {256, 512, 1024}) {
                float best = 1e9f;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipMemsetAsync(ctr, 0, 4096 * 4, st)); CK(hipMemsetAsync(fail, 0, 4, st)); CK(hipMemsetAsync(data, 0, 4096 * 256 * 4, st));
                    CK(hipEventRecord(e0, st));
                    if (mode == 0) hipLaunchKernelGGL(persist<0>, dim3(grid), dim3(256), 0, st, ctr, nsteps, data, work, fail);
                    else hipLaunchKernelGGL(persist<1>, dim3(grid), dim3(256), 0, st, ctr, nsteps, data, work, fail);
                    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
                }
                unsigned f; CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
                printf("persistent mode %s work %d grid %4d: %.2f us per barrier step (fail %u)\n", mode ? "hier" : "flat", work, grid, best * 1000.f / nsteps, f);
                if (f) { printf("barrier timed out: stopping\n"); return 2; }
            }
    for (int grid : {256, 1024}) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, st));
            for (int s = 0; s < nsteps; ++s) hipLaunchKernelGGL(tiny, dim3(grid), dim3(256), 0, st, data, s);
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("back-to-back launches grid %4d: %.2f us per launch\n", grid, best * 1000.f / nsteps);
    }
    return 0;
}
