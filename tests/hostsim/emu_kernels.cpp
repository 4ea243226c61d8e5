/*
 * TEST INFRASTRUCTURE ONLY - CPU twins of the kernels of nfc_kernels.hip for the emulated build of the host runtime
 * (tests/hostsim/build_emulated.sh). They keep what the host runtime relies on: which stream slots a launch covers, which
 * rows of the work table / uniform layout they read, which of the common / exact-modulo kernels takes a stream block
 * (decided from the device-side clocks exactly like nfc_demod_body), where state, rings and frames go. The per-sample work
 * is the product's device step machine (nfc_core.hpp) run lane after lane instead of 64 lanes in lock step.
 */
#include <hip/hip_runtime.h>

#include <cmath>
#include <vector>

#define NFC_DEV static inline
static inline uint32_t emu_add(uint32_t *p, uint32_t v) { uint32_t old = *p; *p += v; return old; }
#define NFC_ATOMIC_ADD(ptr, value) emu_add((ptr), (value))
#define NFC_ANY(predicate) (predicate)
#include "../../nfc-laboratory_amd/csrc/nfc_core.hpp"
#include "../../nfc-laboratory_amd/csrc/nfc_launch.h"

namespace fakehip {
dim3 launchGrid, launchBlock;
}

namespace {

/* nfc_kernels.hip: nfc_exact_span */
bool exact_span(uint32_t clock, uint32_t count)
{
   const uint32_t start = clock + 1u + 1024u;
   const uint32_t untilWrap = 0u - start;
   return count != 0 && (start < 2048u || untilWrap < count);
}

struct Row
{
   const uint8_t *data;
   uint32_t count;
};

/* nfc_kernels.hip: nfc_row + the lane's own count */
Row row_of(const NfcLaunch &L, uint32_t slot)
{
   Row r {nullptr, 0};
   if (slot >= L.firstSlot && slot < L.firstSlot + L.slotCount)
   {
      if (L.works)
      {
         r.data = L.works[slot].data;
         r.count = L.works[slot].count;
      }
      else
      {
         r.data = L.uniformBase + (uint64_t)(slot - L.firstSlot) * L.uniformPitch;
         r.count = L.uniformCount;
      }
   }
   return r;
}

void demod(const NfcConfig *cfgPtr, const NfcLaunch &L, bool exactKernel)
{
   for (uint32_t b = 0; b < fakehip::launchGrid.x; b++)
   {
      const uint32_t block = L.firstBlock + b;
      uint32_t longest = 0;
      bool anyExact = false;
      bool served = false;

      for (uint32_t lane = 0; lane < NFC_LANES; lane++)
      {
         const uint32_t slot = block * NFC_LANES + lane;
         const Row r = row_of(L, slot);
         longest = r.count > longest ? r.count : longest;
         anyExact = anyExact || exact_span(L.states[slot].clock, r.count);
         served = served || (r.count != 0 && L.states[slot].served == L.launchSeq);
      }

      if (longest == 0 || served)
         continue;

      if ((L.forceExact != 0 || anyExact) != exactKernel)
         continue;

      for (uint32_t lane = 0; lane < NFC_LANES; lane++)
      {
         const uint32_t slot = block * NFC_LANES + lane;
         const Row r = row_of(L, slot);

         if (!r.count)
            continue;

         NfcStreamState s = L.states[slot];

         NfcLaneMem mem;
         mem.ring = L.rings + (uint64_t)block * L.ringBlockFloats;
         mem.lane = lane;
         mem.exact = false;
         mem.bytes = L.bytes + (uint64_t)slot * NFC_STREAM_BYTES;
         mem.sink = L.sink;
         mem.sinkCursor = L.sinkCtl;
         mem.sinkDropped = L.sinkCtl + 1;
         mem.sinkWords = L.sinkWords;
         mem.streamId = slot;
         mem.cold = L.cold + slot;
         mem.tables = cfgPtr;

         const float *p = reinterpret_cast<const float *>(r.data);

         for (uint32_t k = 0; k < r.count; k++)
         {
            float v;
            if (L.uniformStride == 2)
            {
               volatile float ii = p[2 * k] * p[2 * k];
               volatile float qq = p[2 * k + 1] * p[2 * k + 1];
               v = __builtin_sqrtf(ii + qq);
            }
            else
               v = p[k];

            nfc_step(*cfgPtr, s, mem, v, exactKernel);
         }

         s.served = L.launchSeq;
         L.states[slot] = s;
      }
   }
}

}

void nfc_demod_kernel(const NfcConfig *__restrict__ cfgPtr, NfcLaunch L) { demod(cfgPtr, L, false); }
void nfc_demod_fixed_kernel(const NfcConfig *__restrict__ cfgPtr, NfcLaunch L) { demod(cfgPtr, L, false); }
void nfc_demod_exact_kernel(const NfcConfig *__restrict__ cfgPtr, NfcLaunch L) { demod(cfgPtr, L, true); }
void nfc_demod_fixed_exact_kernel(const NfcConfig *__restrict__ cfgPtr, NfcLaunch L) { demod(cfgPtr, L, true); }

void nfc_init_kernel(const NfcConfig *__restrict__ cfgPtr, NfcLaunch L, uint32_t keepFrontEnd)
{
   for (uint32_t idx = 0; idx < L.slotCount; idx++)
   {
      const uint32_t slot = L.firstSlot + idx;
      const uint32_t block = slot / NFC_LANES;
      const uint32_t lane = slot % NFC_LANES;

      NfcStreamState s = L.states[slot];
      NfcStreamCold cold;
      nfc_state_init(*cfgPtr, s, cold, keepFrontEnd != 0);
      L.states[slot] = s;
      L.cold[slot] = cold;

      float *ring = L.rings + (uint64_t)block * L.ringBlockFloats + lane;
      const uint32_t from = keepFrontEnd ? 4 * NFC_HIST : 0;
      const uint32_t total = L.ringBlockFloats / NFC_LANES;

      for (uint32_t i = from; i < total; i++)
         ring[i * NFC_LANES] = 0.0f;
   }
}

void nfc_magnitude_kernel(const float2 *__restrict__ iq, float *__restrict__ out, uint64_t n)
{
   for (uint64_t i = 0; i < n; i++)
   {
      volatile float ii = iq[i].x * iq[i].x;
      volatile float qq = iq[i].y * iq[i].y;
      out[i] = __builtin_sqrtf(ii + qq);
   }
}

/* the radio branch of the adaptive resampler (SignalResamplingTask.cpp:168-226), one buffer after the other */
void nfc_resample_radio_kernel(const float *__restrict__ in, uint64_t pitchFloats, uint32_t nBuffers, uint32_t n,
                               float *__restrict__ out, uint64_t outPitchFloats, uint32_t capacityPairs, uint32_t *__restrict__ counts)
{
   const int32_t window = 51, interval = 255;
   const float filter = 0.005f;

   for (uint32_t buffer = 0; buffer < nBuffers; buffer++)
   {
      const float *x = in + (uint64_t)buffer * pitchFloats;
      float *dst = out + (uint64_t)buffer * outPitchFloats;
      uint32_t count = 0;

      auto put = [&](float value, float offset) {
         if (count < capacityPairs)
         {
            dst[2 * count] = value;
            dst[2 * count + 1] = offset;
         }
         count++;
      };

      float avrg = 0.0f;
      for (int32_t k = 0; k < window / 2; k++)
         avrg += x[k];

      float last = x[0];
      put(x[0], 0.0f);

      int32_t c = 0, p = -1;

      for (int32_t i = 0; i < (int32_t)n; ++i, ++p)
      {
         const float value = x[i];

         if ((uint32_t)(i + window / 2) < n)
            avrg += x[i + window / 2];

         if (i - window / 2 - 1 >= 0)
            avrg -= x[i - window / 2 - 1];

         const float stdev = std::fabs(value - (avrg / (float)window));

         if (stdev > filter || (i - c) >= interval)
         {
            if (stdev > filter && c < p)
               put(last, (float)p);

            put(value, (float)i);
            c = i;
         }

         last = value;
      }

      if (c < p)
         put(last, (float)p);

      counts[buffer] = count;
   }
}
