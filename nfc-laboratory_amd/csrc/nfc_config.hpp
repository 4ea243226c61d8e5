/*
 * nfc_config.hpp — host-side derivation of NfcConfig from (sampleRate, thresholds).
 *
 * Mirrors the arithmetic of the reference's initialize() chain so that every derived constant is
 * bit-identical: NfcDecoder.cpp:295-360 (sampleTimeUnit, elementaryTimeUnit, EMA weights, carrier
 * thresholds), NfcA.cpp:140-164 / NfcB.cpp:149-173 / NfcF.cpp:132-158 / NfcV.cpp:150-166 (symbol
 * periods via std::round on double), NfcV.cpp:220-234 (pulse slots).
 */
#ifndef NFC_AMD_CONFIG_HPP
#define NFC_AMD_CONFIG_HPP

#include <cmath>
#include <cstring>

#include "nfc_types.h"

struct NfcHostParams
{
   uint32_t sampleRate = 0;
   uint32_t enabled = 0xF;
   float powerLevelThreshold = 0.01f;
   float corrThreshold[4] = {0.75f, 0.50f, 0.50f, 0.50f};
   float minDepth[4] = {0.90f, 0.10f, 0.10f, 0.90f};
   float maxDepth[4] = {1.00f, 0.90f, 0.90f, 1.00f};
};

static inline void nfc_fill_rate(NfcRate &rt, double stu, int rate, uint32_t delay)
{
   const float fc = 13.56E6f;
   rt.symbolsPerSecond = (uint32_t) static_cast<int>(std::round(fc / static_cast<float>(128 >> rate)));
   rt.p0 = (uint32_t) static_cast<int>(std::round(stu * (256 >> rate)));
   rt.p1 = (uint32_t) static_cast<int>(std::round(stu * (128 >> rate)));
   rt.p2 = (uint32_t) static_cast<int>(std::round(stu * (64 >> rate)));
   rt.p4 = (uint32_t) static_cast<int>(std::round(stu * (32 >> rate)));
   rt.p8 = (uint32_t) static_cast<int>(std::round(stu * (16 >> rate)));
   rt.delay = delay;
   rt.preamble = (uint32_t) static_cast<int>(std::round(stu * (128 >> rate) * 48));
}

/* returns false when the sample rate cannot be decoded with the fixed-depth history rings */
static inline bool nfc_build_config(const NfcHostParams &p, NfcConfig &c)
{
   std::memset(&c, 0, sizeof(c));

   c.sampleRate = p.sampleRate;
   c.enabled = p.enabled;

   if (p.sampleRate == 0)
      return false;

   const float fc = 13.56E6f;

   c.stu = static_cast<double>(p.sampleRate) / static_cast<double>(fc);
   c.etu = static_cast<int>(c.stu * 128);
   c.iirA = static_cast<float>(0.9);
   c.envW0 = static_cast<float>(1 - 5E5 / p.sampleRate);
   c.envW1 = static_cast<float>(1 - c.envW0);
   c.mdevW0 = static_cast<float>(1 - 2E5 / p.sampleRate);
   c.mdevW1 = static_cast<float>(1 - c.mdevW0);
   c.meanW0 = static_cast<float>(1 - 5E4 / p.sampleRate);
   c.meanW1 = static_cast<float>(1 - c.meanW0);
   c.powerThreshold = p.powerLevelThreshold;
   c.lowThreshold = p.powerLevelThreshold / 1.25f;
   c.highThreshold = p.powerLevelThreshold * 1.25f;

   for (int t = 0; t < 4; t++)
   {
      c.corrThreshold[t] = p.corrThreshold[t];
      c.minDepth[t] = p.minDepth[t];
      c.maxDepth[t] = p.maxDepth[t];
   }

   uint32_t delay = 0;

   for (int r = 0; r < 3; r++)
   {
      nfc_fill_rate(c.a[r], c.stu, r, delay);
      nfc_fill_rate(c.b[r], c.stu, r, delay);
      nfc_fill_rate(c.f[r], c.stu, r, 0);
      delay += c.a[r].p1;
   }

   c.v.symbolsPerSecond = (uint32_t) static_cast<int>(std::round(fc / 256));
   c.v.p0 = (uint32_t) static_cast<int>(std::round(c.stu * 512));
   c.v.p1 = (uint32_t) static_cast<int>(std::round(c.stu * 256));
   c.v.p2 = (uint32_t) static_cast<int>(std::round(c.stu * 128));
   c.v.p4 = (uint32_t) static_cast<int>(std::round(c.stu * 64));
   c.v.p8 = (uint32_t) static_cast<int>(std::round(c.stu * 32));
   c.v.delay = c.v.p0;
   c.v.preamble = 0;

   c.vLen2 = static_cast<int>(std::round(4 * c.stu * 256));
   c.vLen8 = static_cast<int>(std::round(256 * c.stu * 256));

   for (int i = 0; i < 4; i++)
      c.vSlotEnd2[i] = static_cast<int>(std::round((i + 1) * c.stu * 256));
   for (int i = 0; i < 256; i++)
      c.vSlotEnd8[i] = static_cast<int>(std::round((i + 1) * c.stu * 256));

   /* correlation ring layout inside the stream block: A106 A212 A424 F212 F424 V(p0) */
   uint32_t off = 0;
   for (int r = 0; r < 3; r++)
   {
      c.corrOffset[r] = off;
      off += c.a[r].p1;
   }
   for (int r = 1; r < 3; r++)
   {
      c.corrOffset[2 + r] = off;
      off += c.f[r].p1;
   }
   c.corrOffset[5] = off;
   off += c.v.p0;
   c.corrTotal = off;

   /* every look-back must fit the history rings; the periods must be usable as ring moduli */
   const uint32_t deepest = c.v.delay + c.v.p2 + 1;
   const uint32_t deepestProd = c.v.p1 + 1;
   bool ok = deepest < NFC_HIST_STORED && deepestProd < NFC_PROD && (c.a[2].delay + c.a[2].p1 + 1) < NFC_HIST_STORED;
   /* ring moduli must be usable: p2 + 1 < p1 keeps the three ring points of a correlator distinct */
   for (int r = 0; r < 3; r++)
      ok = ok && c.a[r].p8 > 0 && c.a[r].p2 >= 2 && c.a[r].p2 + 1 < c.a[r].p1 && c.f[r].p2 + 1 < c.f[r].p1;
   ok = ok && c.v.p2 + 1 < c.v.p1 && c.v.p1 < c.v.p0 && c.etu > 0;

   return ok;
}


#endif
