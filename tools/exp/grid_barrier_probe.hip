// Cost and cross-XCD visibility of a grid-wide barrier (one workgroup per CU, device-scope atomics + fences) on one GPU.
// Every spin is bounded: a workgroup that waits too long raises a flag and every workgroup leaves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target, int* abort_flag) {
    __syncthreads();
    __shared__ int ok;
    if (threadIdx.x == 0) {
        __threadfence();
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        int good = 1;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > (1u << 20) || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                good = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        __threadfence();
        ok = good;
    }
    __syncthreads();
    return ok != 0;
}

__global__ __launch_bounds__(256) void probe(unsigned* counter, int* abort_flag, float* buf, int nb, int work, unsigned* errors) {
    const unsigned n = gridDim.x, b = blockIdx.x;
    unsigned bad = 0;
    for (int p = 0; p < nb; ++p) {
        float* cur = buf + (size_t)(p & 1) * n * work;
        for (int i = threadIdx.x; i < work; i += 256) cur[(size_t)b * work + i] = (float)(p * 1000 + b);
        if (!grid_barrier(counter, (unsigned)(p + 1) * n, abort_flag)) return;
        const unsigned nb_ = (b + 1) % n;
        for (int i = threadIdx.x; i < work; i += 256)
            if (cur[(size_t)nb_ * work + i] != (float)(p * 1000 + nb_)) ++bad;
    }
    if (bad) atomicAdd(errors, bad);
}

__global__ void tiny(float* x) { if (threadIdx.x == 0 && blockIdx.x == 0) x[0] += 1.f; }

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("CUs %d\n", cus);
    unsigned* counter; int* abort_flag; float* buf; unsigned* errors;
    const int work = 4096;
    CHECK(hipMalloc(&counter, 4)); CHECK(hipMalloc(&abort_flag, 4)); CHECK(hipMalloc(&errors, 4));
    CHECK(hipMalloc(&buf, sizeof(float) * 2 * cus * work));
    hipEvent_t a, e; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&e));
    for (int nblk : {cus / 4, cus}) {
        for (int w : {256, 4096}) {
            for (int rep = 0; rep < 2; ++rep) {
                CHECK(hipMemset(counter, 0, 4)); CHECK(hipMemset(abort_flag, 0, 4)); CHECK(hipMemset(errors, 0, 4));
                const int nb = 200;
                CHECK(hipEventRecord(a, 0));
                hipLaunchKernelGGL(probe, dim3(nblk), dim3(256), 0, 0, counter, abort_flag, buf, nb, w, errors);
                CHECK(hipEventRecord(e, 0));
                CHECK(hipDeviceSynchronize());
                float ms; CHECK(hipEventElapsedTime(&ms, a, e));
                int ab; unsigned er;
                CHECK(hipMemcpy(&ab, abort_flag, 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&er, errors, 4, hipMemcpyDeviceToHost));
                if (rep) printf("workgroups %d, %d floats each per phase: %.2f us per phase (write + barrier + read), aborted %d, stale reads %u\n",
                                nblk, w, ms * 1e3 / nb, ab, er);
            }
        }
    }
    // dependent chain of tiny kernels: the per-launch floor
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(a, 0));
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(tiny, dim3(256), dim3(256), 0, 0, buf);
        CHECK(hipEventRecord(e, 0));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, a, e));
        if (rep) printf("chain of 200 tiny kernels (256 workgroups): %.2f us per launch\n", ms * 1e3 / 200);
    }
    return 0;
}
