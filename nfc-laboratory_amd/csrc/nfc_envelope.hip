/*
 * nfc_envelope.hip — gfx950 kernel of the envelope tracker's second walks (nfc_envelope.hpp): one lane per listed chunk,
 * 64 chunks per wave, no LDS, no barrier; a lane reads its own chunk (8 B of IQ or 4 B of magnitude per sample, a group of
 * sixteen samples ahead of the one it walks) and carries two words of state.
 *
 * Bound: the latency of one lane's dependent arithmetic (the tracker is a recurrence: sample k needs the envelope sample
 * k - 1 left), a chunk of 32768 samples per round of a large submission, 4096 of a capture; HBM traffic is the listed
 * chunks' samples once. The path is only taken at the sample rate whose constants are compiled in (nfcgpu.hip:
 * windowed_eligible), so the tracker's three constants are literals.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NFC_DEV __device__ __forceinline__
#define NFC_ATOMIC_ADD(ptr, value) atomicAdd((ptr), (value))
#define NFC_ANY(predicate) (__any(predicate) != 0)

#include "nfc_core.hpp"
#include "nfc_scan.h"
#include "nfc_launch.h"
#include "nfc_scan_launch.h"

/* magnitude of one IQ sample, the reference's scalar formula (RadioDeviceTask.cpp:626-642): products and sum rounded
 * separately (no contraction), correctly rounded square root - as nfc_kernels.hip forms it */
__device__ __forceinline__ float nfc_envelope_sample_at(const uint8_t *data, uint32_t stride, uint32_t index)
{
   if (stride == 2)
   {
      const float2 iq = reinterpret_cast<const float2 *>(data)[index];
      return __builtin_sqrtf(__fadd_rn(__fmul_rn(iq.x, iq.x), __fmul_rn(iq.y, iq.y)));
   }
   return reinterpret_cast<const float *>(data)[index];
}

#define NFC_SAMPLE_AT(data, stride, index) nfc_envelope_sample_at((data), (stride), (index))

/* sample-rate-derived constants of the most common configuration as literals (generated at build time) */
#define NFC_FIXED_FN __device__ __forceinline__
#include "nfc_config_fixed.inc"

#include "nfc_envelope.hpp"

__global__ __launch_bounds__(64) void nfc_envelope_kernel(const NfcConfig *__restrict__ cfgPtr, NfcScanArgs A)
{
   const uint32_t listed = blockIdx.x * NFC_LANES + threadIdx.x;

   if (listed >= A.nChunks)
      return;

   (void)cfgPtr; /* (the thresholds of the run-time configuration play no part in the tracker) */

   NfcConfig cc;
   nfc_fixed_config(cc);

   nfc_envelope_rewalk(cc, A, A.chunks[listed]);
}
