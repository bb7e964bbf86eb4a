// How long does the host take to learn that a small kernel has finished?  (a) hipStreamSynchronize, (b) hipEventSynchronize on an
// event recorded behind it, (c) the kernel's LAST action is a system-scope store of a sequence number into pinned host memory and
// the host spins on it.  Median of 2000 launches each of a one-block kernel that writes 10 results + the flag.
//   hipcc -O2 --offload-arch=gfx950 scripts/probe_flag_wait.hip -o /tmp/probe_flag_wait && /tmp/probe_flag_wait
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void work_kernel(float* out, volatile uint32_t* flag, uint32_t seq, int spin) {
    float v = threadIdx.x;
    for (int i = 0; i < spin; i++) v = v * 1.0001f + 0.5f;
    if (threadIdx.x < 10) out[threadIdx.x] = v + seq;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0 && flag) __hip_atomic_store(const_cast<uint32_t*>(flag), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

static double med(std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2] * 1e6; }

int main() {
    float* out; uint32_t* flag;
    CK(hipHostMalloc(reinterpret_cast<void**>(&out), 4096, hipHostMallocDefault));
    CK(hipHostMalloc(reinterpret_cast<void**>(&flag), 4096, hipHostMallocDefault));
    *flag = 0;
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const int N = 2000;
    for (int spin : {0, 2000, 20000}) {
        std::vector<double> a, b, c;
        uint32_t seq = *flag;
        for (int mode = 0; mode < 3; mode++)
            for (int i = 0; i < N + 50; i++) {
                auto t0 = std::chrono::steady_clock::now();
                seq++;
                hipLaunchKernelGGL(work_kernel, dim3(1), dim3(256), 0, s, out, mode == 2 ? flag : nullptr, seq, spin);
                if (mode == 0) CK(hipStreamSynchronize(s));
                else if (mode == 1) { CK(hipEventRecord(ev, s)); CK(hipEventSynchronize(ev)); }
                else { while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {} }
                double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (mode == 2 && out[3] != 0 && i == 0) {}
                if (i >= 50) (mode == 0 ? a : mode == 1 ? b : c).push_back(dt);
                if (mode == 2 && (i % 64) == 63) CK(hipStreamSynchronize(s));  // keep the queue short
            }
        CK(hipStreamSynchronize(s));
        printf("kernel spin %6d:  hipStreamSynchronize %6.1f us   event record + synchronize %6.1f us   flag in pinned memory %6.1f us\n", spin, med(a), med(b), med(c));
    }
    return 0;
}
