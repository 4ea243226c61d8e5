/*
 * TEST INFRASTRUCTURE ONLY - a stand-in for <hip/hip_runtime.h> with which the host runtime of the product
 * (nfc-laboratory_amd/csrc/nfcgpu.hip, unchanged) compiles for a box without a GPU: device memory is host memory,
 * the stream is synchronous, a kernel launch is a call of the CPU twin of that kernel (tests/hostsim/emu_kernels.cpp).
 * Only tests/hostsim/build_emulated.sh uses it; see there for what the resulting library is for.
 */
#ifndef NFC_FAKE_HIP_RUNTIME_H
#define NFC_FAKE_HIP_RUNTIME_H

#include <cstdint>
#include <cstdlib>
#include <cstring>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline

typedef int hipError_t;
#define hipSuccess 0
#define hipErrorUnknown 999

#include <functional>
#include <vector>

/* A stream is synchronous - a launch is a call - unless it was created with a priority while NFC_EMU_DEFER_LOW is set: then
 * its launches are kept and run when somebody waits for the stream (hipStreamSynchronize, or hipStreamWaitEvent /
 * hipEventSynchronize on an event recorded on it). The product gives its lowest-priority stream the walk that writes the
 * front-end planes beside the rounds of second walks: with the switch that walk runs after every rewrite of the rounds instead
 * of before them - the other order of the two the device may take (tests/test_time_parallel.py). */
struct fakeHipStream
{
   bool deferred;
   std::vector<std::function<void()>> queue;
};

struct fakeHipEvent
{
   fakeHipStream *on;
};

typedef struct fakeHipStream *hipStream_t;
typedef struct fakeHipEvent *hipEvent_t;

enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
#define hipStreamNonBlocking 1u
#define hipHostMallocDefault 0u

struct dim3
{
   uint32_t x, y, z;
   dim3(uint32_t x_ = 1, uint32_t y_ = 1, uint32_t z_ = 1) : x(x_), y(y_), z(z_) {}
};

struct float2
{
   float x, y;
};

struct float4
{
   float x, y, z, w;
};

#include <mutex>

namespace fakehip {
extern dim3 launchGrid, launchBlock;
/* the CPU twins keep their working storage in statics (one wave's LDS, the fibres of a wave): one launch at a time,
 * whatever host thread it comes from (the shards of nfcgpu.hip launch from threads of their own) */
extern std::recursive_mutex launchMutex;
/* the device the calling thread last made current (hipSetDevice): every entry point of the C ABI has to set its context's own
 * before it touches the runtime - a host with a context per GPU calls them in any order (tests/test_bench_multi_rank_dry.py) */
extern thread_local int currentDevice;
}

static inline const char *hipGetErrorString(hipError_t) { return "emulated HIP error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
/* (NFCGPU_FAKE_DEVICES: how many devices the stand-in shows - a rehearsal of one process per GPU asks for device LOCAL_RANK) */
static inline hipError_t hipGetDeviceCount(int *count)
{
   const char *v = std::getenv("NFCGPU_FAKE_DEVICES");
   *count = v && v[0] ? std::atoi(v) : 1;
   return hipSuccess;
}
static inline hipError_t hipSetDevice(int device) { fakehip::currentDevice = device; return hipSuccess; }
namespace fakehip {
static inline void flush(fakeHipStream *s)
{
   while (s && !s->queue.empty())
   {
      std::vector<std::function<void()>> now;
      now.swap(s->queue);
      for (auto &f: now)
         f();
   }
}
}

static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = new fakeHipStream {false, {}}; return hipSuccess; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int)
{
   const char *v = std::getenv("NFC_EMU_DEFER_LOW");
   *s = new fakeHipStream {v && v[0] && v[0] != '0', {}};
   return hipSuccess;
}
static inline hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest) { *least = 0; *greatest = -1; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { fakehip::flush(s); delete s; return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t s) { fakehip::flush(s); return hipSuccess; }
static inline hipError_t hipMalloc(void **p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorUnknown; }
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { return hipMalloc(p, n); }
static inline hipError_t hipFree(void *p) { std::free(p); return hipSuccess; }
static inline hipError_t hipMemGetInfo(size_t *freeBytes, size_t *totalBytes) { *freeBytes = 0; *totalBytes = 0; return hipSuccess; }
static inline hipError_t hipHostFree(void *p) { std::free(p); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) { std::memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { std::memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { std::memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new fakeHipEvent {nullptr}; return hipSuccess; }
#define hipEventDisableTiming 2u
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new fakeHipEvent {nullptr}; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
/* (an event stands for everything queued on its stream so far; whoever waits for it has that stream's queue run) */
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) { if (e) e->on = s; return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t e) { if (e) fakehip::flush(e->on); return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t e, unsigned) { if (e) fakehip::flush(e->on); return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return hipSuccess; }

/* a launch is a call: the CPU twin walks the grid itself */
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
   do                                                              \
   {                                                               \
      const dim3 fakeGrid = (grid), fakeBlock = (block);           \
      auto fakeLaunch = [=]() {                                    \
         std::lock_guard<std::recursive_mutex> launchLock(fakehip::launchMutex); \
         fakehip::launchGrid = fakeGrid;                           \
         fakehip::launchBlock = fakeBlock;                         \
         (kernel)(__VA_ARGS__);                                    \
      };                                                           \
      fakeHipStream *fakeOn = (stream);                            \
      if (fakeOn && fakeOn->deferred)                              \
         fakeOn->queue.push_back(fakeLaunch);                      \
      else                                                         \
         fakeLaunch();                                             \
   } while (0)

#endif
