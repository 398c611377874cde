// Does a kernel dispatched behind hipStreamWaitValue32 see what a kernel of ANOTHER stream, running on other XCDs, wrote before it set
// the word?  Each XCD has its own L2; data written through one XCD's L2 reaches another's only by write-back + invalidate, which kernel
// boundaries ordered by events are known to perform.  The test makes stale lines certain if they are possible: a 1 MB buffer that EVERY
// workgroup of the consumer reads in full (so every XCD's L2 holds all of it), rewritten each round by a producer whose waves each
// release (agent scope) and count themselves, the last one setting the word (system scope) — the protocol a scan kernel would use.
// Three orderings: "event" (hipEventRecord behind the producer + hipStreamWaitEvent: the control that must be clean), "value"
// (hipStreamWaitValue32 on the word), "none" (no ordering at all: shows what stale / early reads look like on this box).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/wait_value_xcd profiles/micro/wait_value_xcd.hip && timeout -k 5 120 /tmp/wait_value_xcd
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr unsigned kWords = 1u << 18;          // 1 MB
constexpr unsigned kProducerWgs = 512, kConsumerWgs = 512;

__global__ __launch_bounds__(256) void producer(unsigned *data, unsigned value, unsigned long long *counter, unsigned long long target,
                                                unsigned *flag, unsigned long long spin_ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(8);          // (lets the consumer's dispatch arrive first)
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < kWords; i += gridDim.x * 256u) data[i] = value ^ i;
    // every wave: release what it wrote, then count itself; the wave that completes the count sets the word
    if ((threadIdx.x & 63u) == 0u) {
        const unsigned long long old = __hip_atomic_fetch_add(counter, 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1ull == target) __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ __launch_bounds__(256) void consumer(const unsigned *data, unsigned value, unsigned *bad) {
    unsigned b = 0;
    for (unsigned i = threadIdx.x; i < kWords; i += 256u) b += data[i] != (value ^ i);      // every workgroup reads everything
    if (b) atomicAdd(bad, b);
}

int main() {
    int can = 0;
    CHECK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    unsigned *data, *flag, *bad; unsigned long long *counter;
    CHECK(hipMalloc((void **)&data, kWords * 4)); CHECK(hipMalloc((void **)&flag, 8)); CHECK(hipMalloc((void **)&bad, 4));
    CHECK(hipMalloc((void **)&counter, 8));
    hipStream_t a, b; hipEvent_t ev;
    CHECK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const char *names[3] = {"event", "value", "none"};
    for (int mode = 0; mode < 3; ++mode) {
        if (mode == 1 && !can) { std::printf("value: hipStreamWaitValue32 not supported here\n"); continue; }
        CHECK(hipMemset(data, 0, kWords * 4)); CHECK(hipMemset(flag, 0, 8)); CHECK(hipMemset(bad, 0, 4)); CHECK(hipMemset(counter, 0, 8));
        CHECK(hipDeviceSynchronize());
        const unsigned rounds = 400;
        unsigned long long target = 0;
        unsigned stale_rounds = 0, total_bad = 0;
        for (unsigned r = 1; r <= rounds; ++r) {
            const unsigned value = (mode + 1) * 0x01000000u + r * 0x10001u;
            target += (unsigned long long)kProducerWgs * 4ull;                 // waves per launch
            hipLaunchKernelGGL(producer, dim3(kProducerWgs), dim3(256), 0, a, data, value, counter, target, flag, (r & 3u) ? 0ull : 2000ull);
            if (mode == 0) { CHECK(hipEventRecord(ev, a)); CHECK(hipStreamWaitEvent(b, ev, 0)); }
            if (mode == 1) CHECK(hipStreamWaitValue32(b, flag, value, hipStreamWaitValueEq, 0xFFFFFFFFu));
            hipLaunchKernelGGL(consumer, dim3(kConsumerWgs), dim3(256), 0, b, data, value, bad);
            // the next producer must not overwrite what this consumer still reads: order a behind b (an event on b: not what is tested)
            CHECK(hipEventRecord(ev, b)); CHECK(hipStreamWaitEvent(a, ev, 0));
            if ((r % 50u) == 0u) {
                CHECK(hipDeviceSynchronize());
                unsigned nb = 0; CHECK(hipMemcpy(&nb, bad, 4, hipMemcpyDeviceToHost)); CHECK(hipMemset(bad, 0, 4));
                if (nb) { ++stale_rounds; total_bad += nb; }
            }
        }
        CHECK(hipDeviceSynchronize());
        std::printf("%-5s: %u rounds, %u of 8 checkpoints saw wrong words (%u words x workgroups in all)\n", names[mode], rounds, stale_rounds, total_bad);
    }
    return 0;
}
