// Can a stream wait for a word a KERNEL of another stream writes (hipStreamWaitValue32), instead of for an event recorded behind that
// kernel?  An event record is a packet on the producer's queue: the producer's next kernel starts only after it (0.011-0.013 ms per
// scan in profiles/r05/shard_step_queues.txt).  A wait for a value puts nothing on the producer's queue.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/wait_value profiles/micro/wait_value.hip && timeout -k 5 60 /tmp/wait_value
// The producer is always enqueued BEFORE the wait, so a shared hardware queue serialises the two instead of deadlocking them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void producer(unsigned long long ticks, unsigned *data, unsigned n, unsigned *flag, unsigned value) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    for (unsigned i = threadIdx.x; i < n; i += blockDim.x) data[i] = value + i;          // what the consumer must see
    __syncthreads();
    if (threadIdx.x == 0) {
        __atomic_thread_fence(__ATOMIC_RELEASE);                                            // (system scope)
        __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ void consumer(const unsigned *data, unsigned n, unsigned value, unsigned *bad, unsigned long long *when) {
    if (threadIdx.x == 0) when[0] = wall_clock64();
    unsigned b = 0;
    for (unsigned i = threadIdx.x; i < n; i += blockDim.x) b += data[i] != value + i;
    if (b) atomicAdd(bad, b);
    if (threadIdx.x == 0) when[1] = wall_clock64();
}
__global__ void stamp(unsigned long long *when) { *when = wall_clock64(); }

int main() {
    int can = -1;
    CHECK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    std::printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    if (can != 1) return 0;
    for (int kind = 0; kind < 2; ++kind) {                  // 0: signal memory, 1: plain device memory
        unsigned *flag = nullptr, *data = nullptr, *bad = nullptr;
        unsigned long long *when = nullptr;
        if (kind == 0) CHECK(hipExtMallocWithFlags((void **)&flag, 8, hipMallocSignalMemory));
        else CHECK(hipMalloc((void **)&flag, 8));
        const unsigned n = 1u << 20;
        CHECK(hipMalloc((void **)&data, n * 4));
        CHECK(hipMalloc((void **)&bad, 4));
        CHECK(hipMalloc((void **)&when, 32));
        CHECK(hipMemset(flag, 0, 8)); CHECK(hipMemset(data, 0, n * 4)); CHECK(hipMemset(bad, 0, 4)); CHECK(hipMemset(when, 0, 32));
        hipStream_t a, b;
        CHECK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
        CHECK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
        CHECK(hipDeviceSynchronize());
        for (unsigned round = 1; round <= 3; ++round) {
            hipLaunchKernelGGL(stamp, dim3(1), dim3(1), 0, a, when + 0);
            hipLaunchKernelGGL(producer, dim3(1), dim3(256), 0, a, 50000ull /* 0.5 ms */, data, n, flag, round * 1000u);
            hipLaunchKernelGGL(stamp, dim3(1), dim3(1), 0, a, when + 1);           // right behind the producer: the queue was not held up
            const hipError_t e = hipStreamWaitValue32(b, flag, round * 1000u, hipStreamWaitValueGte, 0xFFFFFFFFu);
            if (e != hipSuccess) { std::printf("kind %d: hipStreamWaitValue32 -> %s\n", kind, hipGetErrorString(e)); break; }
            hipLaunchKernelGGL(consumer, dim3(1), dim3(256), 0, b, data, n, round * 1000u, bad, when + 2);
            CHECK(hipDeviceSynchronize());
            unsigned long long w[4]; unsigned nbad = 0;
            CHECK(hipMemcpy(w, when, 32, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(&nbad, bad, 4, hipMemcpyDeviceToHost));
            std::printf("kind %d (%s) round %u: producer began 0, stamp behind it %+.1f us, consumer began %+.1f us and ended %+.1f us, words the consumer saw stale: %u\n",
                        kind, kind == 0 ? "signal memory" : "hipMalloc", round, (double)(w[1] - w[0]) / 100.0, (double)(w[2] - w[0]) / 100.0, (double)(w[3] - w[0]) / 100.0, nbad);
        }
        (void)hipStreamDestroy(a); (void)hipStreamDestroy(b);
        (void)hipFree(flag); (void)hipFree(data); (void)hipFree(bad); (void)hipFree(when);
    }
    return 0;
}
