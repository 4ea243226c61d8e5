/*
 * nfc_wave.hip — the wave decoder's kernel (nfc_wave.hpp) for gfx950: a workgroup is one wave, a wave is one lane of
 * work of the time-parallel path. The decoder's step machine is compiled here a second time with one stream's rings in
 * LDS (history 1024 deep, no lane pitch).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NFC_HIST 1024u
#define NFC_RING_STRIDE 1u
#define NFC_WAVE_LDS __attribute__((address_space(3)))
#define NFC_RING_FLOAT NFC_WAVE_LDS float

#define NFC_DEV __device__ __forceinline__

/* the step runs wave-uniform: one lane appends to the frame sink for all */
__device__ __forceinline__ uint32_t nfc_wave_atomic_add(uint32_t *p, uint32_t v)
{
   uint32_t old = 0;
   if (threadIdx.x == 0)
      old = atomicAdd(p, v);
   return (uint32_t)__builtin_amdgcn_readfirstlane((int)old);
}

#define NFC_ATOMIC_ADD(ptr, value) nfc_wave_atomic_add((ptr), (value))
#define NFC_ANY(predicate) (predicate)

#include "nfc_core.hpp"

__device__ __forceinline__ float nfc_iq_magnitude(float i, float q)
{
   return __builtin_sqrtf(__fadd_rn(__fmul_rn(i, i), __fmul_rn(q, q)));
}

__device__ __forceinline__ float nfc_sample_at(const uint8_t *data, uint32_t stride, uint32_t index)
{
   if (stride == 2)
   {
      const float2 iq = reinterpret_cast<const float2 *>(data)[index];
      return nfc_iq_magnitude(iq.x, iq.y);
   }
   return reinterpret_cast<const float *>(data)[index];
}

#define NFC_SAMPLE_AT(data, stride, index) nfc_sample_at((data), (stride), (index))
#define NFC_FENCE() __threadfence()
#include "nfc_scan.hpp"

#define NFC_WAVE_LANE() (threadIdx.x)
#define NFC_WAVE_BARRIER() __syncthreads()
#define NFC_WAVE_BALLOT(p) ((uint64_t)__ballot(p))
#define NFC_WAVE_SHFL_UP_F(v, d) __shfl_up((v), (d), 64)
#define NFC_WAVE_SHFL_XOR_F(v, d) __shfl_xor((v), (d), 64)
#define NFC_WAVE_UNIFORM_BEGIN(u) {
#define NFC_WAVE_UNIFORM_END(u) }
#define NFC_WAVE_STAT_ADD(p, v) atomicAdd((p), (v))
#define NFC_WAVE_STAT_MAX(p, v) atomicMax((p), (v))

#include "nfc_wave.hpp"

#define NFC_FIXED_FN __device__ __forceinline__
#include "nfc_config_fixed.inc"

namespace {

/* run-time part of the configuration on top of the compiled-in table (the path is only taken at that sample rate) */
__device__ __forceinline__ void nfc_wave_config(const NfcConfig *cfgPtr, NfcConfig &cc)
{
   typedef __attribute__((address_space(4))) const NfcConfig ConstConfig;
   ConstConfig *cp = (ConstConfig *)cfgPtr;

   nfc_fixed_config(cc);
   cc.enabled = cp->enabled;
   cc.powerThreshold = cp->powerThreshold;
   cc.lowThreshold = cp->lowThreshold;
   cc.highThreshold = cp->highThreshold;
   for (int t = 0; t < 4; t++)
   {
      cc.corrThreshold[t] = cp->corrThreshold[t];
      cc.minDepth[t] = cp->minDepth[t];
      cc.maxDepth[t] = cp->maxDepth[t];
   }
}

} // namespace

__global__ __launch_bounds__(64) void nfc_wave_kernel(const NfcConfig *__restrict__ cfgPtr, NfcLaunch L, NfcScanArgs A, uint32_t mode)
{
   __shared__ NfcWaveLds lds;

   NfcConfig cc;
   nfc_wave_config(cfgPtr, cc);

   nfc_wave_run(cfgPtr, cc, L, A, mode, blockIdx.x, (NFC_WAVE_LDS NfcWaveLds *)&lds);
}
