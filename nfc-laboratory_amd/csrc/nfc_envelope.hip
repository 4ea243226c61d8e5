/*
 * nfc_envelope.hip — gfx950 kernel of the envelope tracker's second walks (nfc_envelope.hpp): one wavefront per listed chunk
 * (round 5; round 4: one lane per chunk), no LDS, no barrier; lane j loads sample j of a tile (8 B of IQ or 4 B of magnitude,
 * one coalesced load per tile, two tiles ahead of the one being walked), every lane walks the tile's 64 steps on values handed
 * round with v_readlane.
 *
 * Bound: the latency of the tracker's dependent arithmetic (a recurrence: sample k needs the envelope sample k - 1 left), a
 * chunk of 32768 samples per round of a large submission, 4096 of a capture; HBM traffic is the listed chunks' samples once. The path is only taken at the sample rate whose constants are compiled in (nfcgpu.hip:
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

#define NFC_ENVELOPE_WAVE
#define NFC_ENVELOPE_READLANE_F(v, j) __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, (v)), (int)(j)))
/* (a value the compiler may know nothing about: in a vector register, and not to be reasoned about as uniform) */
#define NFC_ENVELOPE_OPAQUE_F(v) asm volatile("" : "+v"(v))
#define NFC_ENVELOPE_OPAQUE_U(v) asm volatile("" : "+v"(v))

#define NFC_ENVELOPE_BALLOT(p) ((uint64_t)__ballot(p))
/* (a value every lane holds alike: is it positive? as a uniform condition) */
#define NFC_ENVELOPE_UNIFORM_POSITIVE(v) (__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, (v)))) > 0.0f)

#ifdef NFC_ENVELOPE_VERIFY_BUILD
/* -DNFC_ENVELOPE_VERIFY_BUILD (not a product build): every tile the groups have walked is walked again sample by sample by the
 * statement (nfc_envelope_step) from the same state and the two results compared bit for bit; every launch says what the launches
 * before it have found (stdout: "[envelope verify] tiles differing") */
static __device__ uint32_t nfcEnvelopeVerifyTiles, nfcEnvelopeVerifyDiffering;

__device__ __forceinline__ void nfc_envelope_verify(const NfcConfig &c, float env, uint32_t pulseFilter, float x0, float e, uint32_t pf, float l, float h)
{
   float ve = env, vl = 3.0e38f, vh = -3.0e38f;
   uint32_t vpf = pulseFilter;
   uint32_t clock = 1u << 20; /* (past the stream's first symbol, as the tile is) */

   for (uint32_t j = 0; j < 64u; j++)
   {
      const float x = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x0), (int)j));
      ++clock;
      ++vpf;
      nfc_envelope_step(c, clock, vpf, ve, x);
      vl = ve < vl ? ve : vl;
      vh = ve > vh ? ve : vh;
   }

   if (threadIdx.x == 0)
   {
      atomicAdd(&nfcEnvelopeVerifyTiles, 1u);
      if (__builtin_bit_cast(uint32_t, ve) != __builtin_bit_cast(uint32_t, e) || vpf != pf || __builtin_bit_cast(uint32_t, vl) != __builtin_bit_cast(uint32_t, l) ||
          __builtin_bit_cast(uint32_t, vh) != __builtin_bit_cast(uint32_t, h))
      {
         if (atomicAdd(&nfcEnvelopeVerifyDiffering, 1u) < 8u)
            printf("[envelope verify] differs: env %a -> %a / %a, counter %u -> %u / %u, min %a / %a, max %a / %a\n", (double)env, (double)e, (double)ve, pulseFilter, pf, vpf,
                   (double)l, (double)vl, (double)h, (double)vh);
      }
   }
}
#define NFC_ENVELOPE_VERIFY(c, env, pf0, x0, limit, e, pf, l, h) nfc_envelope_verify((c), (env), (pf0), (x0), (e), (pf), (l), (h))
#endif

#include "nfc_envelope.hpp"

/* one wavefront per listed chunk (nfc_envelope_rewalk_wave) */
__global__ __launch_bounds__(64) void nfc_envelope_kernel(const NfcConfig *__restrict__ cfgPtr, NfcScanArgs A)
{
   const uint32_t listed = blockIdx.x;

#ifdef NFC_ENVELOPE_VERIFY_BUILD
   if (listed == 0u && threadIdx.x == 0u)
      printf("[envelope verify] %u %u\n", nfcEnvelopeVerifyTiles, nfcEnvelopeVerifyDiffering);
#endif

   if (listed >= A.nChunks)
      return;

   (void)cfgPtr; /* (the thresholds of the run-time configuration play no part in the tracker) */

   NfcConfig cc;
   nfc_fixed_config(cc);

   nfc_envelope_rewalk_wave(cc, A, A.chunks[listed], threadIdx.x);
}
