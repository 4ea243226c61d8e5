/*
 * TEST INFRASTRUCTURE ONLY - property check of the rule by which the chain kernel lets a lane of the time-parallel path
 * stand although the NFC-F pulse memory it assumed (pulse counter, threshold of the last pulse) was not the true one
 * (nfc_fbound_admits, nfc_scan.hpp): two copies of the preamble tracker (nfcf_track_preamble, the product's text) are fed
 * the same random pulse trains from two different memories; the first records its bounds as a lane does. Whenever the
 * bounds admit the second memory the two must have decided the same at every call, hold the same record at every call up
 * to the counter's constant offset (until the record starts over) and the threshold (until the first pulse that sets it),
 * and end as nfc_chain_follow's correction predicts.
 *
 * usage: fbound_check <seed> <trains>   -> prints the number of trains and how many were admitted; exit 1 on a violation
 */
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

#define NFC_DEV static inline
static inline uint32_t emu_add(uint32_t *p, uint32_t v) { uint32_t old = *p; *p += v; return old; }
#define NFC_ATOMIC_ADD(ptr, value) emu_add((ptr), (value))
#define NFC_ANY(predicate) (predicate)
#include "../../nfc-laboratory_amd/csrc/nfc_core.hpp"
static inline float sample_at(const uint8_t *, uint32_t, uint32_t) { return 0.0f; }
#define NFC_SAMPLE_AT(data, stride, index) sample_at((data), (stride), (index))
#define NFC_FENCE() ((void)0)
#include "../../nfc-laboratory_amd/csrc/nfc_scan.h"
#include "../../nfc-laboratory_amd/csrc/nfc_scan.hpp"

static bool same_but_memory(const NfcDetF &a, const NfcDetF &b)
{
   NfcDetF x = a, y = b;
   x.pulses = y.pulses = 0;
   x.thr = y.thr = 0.0f;
   return std::memcmp(&x, &y, sizeof(x)) == 0;
}

int main(int argc, char **argv)
{
   const uint32_t seed = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 1u;
   const uint32_t trains = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 100000u;

   std::mt19937 rng(seed);
   std::uniform_real_distribution<float> unit(0.0f, 1.0f);

   NfcRate rt;
   std::memset(&rt, 0, sizeof(rt));
   rt.p0 = 47; rt.p1 = 24; rt.p2 = 12; rt.p4 = 6; rt.p8 = 3; rt.preamble = 1133; /* NFC-F 424 at 10 MS/s */

   uint64_t admitted = 0, evaluated = 0, different = 0, offset = 0, threshold = 0, completed = 0, startedOver = 0;

   for (uint32_t n = 0; n < trains; n++)
   {
      NfcStreamState sa, sb;
      std::memset(&sa, 0, sizeof(sa));
      std::memset(&sb, 0, sizeof(sb));

      NfcDetF a, b;
      std::memset(&a, 0, sizeof(a));

      /* a record at rest with some memory, as a partial reset leaves it; both copies the same but for the memory */
      a.syncValue = unit(rng); a.c0 = unit(rng) - 0.5f; a.lastPhase = unit(rng) - 0.5f; a.lastValue = unit(rng) - 0.5f;
      b = a;

      const uint32_t kinds = rng() % 4u;
      a.pulses = kinds == 0 ? 0u : (kinds == 1 ? rng() % 8u : (kinds == 2 ? 85u + rng() % 20u : rng() % 200u));
      b.pulses = (rng() % 3u) == 0 ? a.pulses : (rng() % 2u ? rng() % 8u : 85u + rng() % 20u);
      a.thr = (rng() % 3u) == 0 ? 0.0f : unit(rng) * 0.5f;
      b.thr = (rng() % 4u) == 0 ? a.thr : ((rng() % 3u) == 0 ? 0.0f : unit(rng) * 0.5f);

      const uint32_t hadPulses = a.pulses, havePulses = b.pulses;
      const float hadThr = a.thr, haveThr = b.thr;

      NfcFBound bound;
      std::memset(&bound, 0, sizeof(bound));
      uint32_t clearedA = 0, clearedB = 0, usedA = 0, usedB = 0;

      /* the trajectory, kept to be judged once the bounds are complete */
      enum { STEPS = 4000 };
      static NfcDetF ra[STEPS], rb[STEPS];
      static uint8_t reta[STEPS], retb[STEPS], cla[STEPS], clb[STEPS], owna[STEPS];

      uint32_t clock = 1000u + rng() % 1000u;
      const float level = 0.05f + unit(rng) * 0.3f;
      uint32_t steps = 0;
      bool locked = false;

      /* (short trains too: lanes that end before the record has started over, with the counter still off) */
      const uint32_t limit = (rng() % 2u) ? STEPS : 20u + rng() % 400u;

      for (; steps < limit && !locked; steps++)
      {
         clock++;
         sa.clock = sb.clock = clock;

         /* pulses: bursts of correlation above the level roughly every half symbol, with gaps (the detector's partial
          * reset: nfcf_detect_decide) */
         const bool burst = ((clock / 12u) % 2u) == 0u && (rng() % 8u) != 0u;
         const float sd = burst ? level + unit(rng) * 0.4f : unit(rng) * level * 0.9f;
         const float s0 = (rng() % 2u ? 1.0f : -1.0f) * sd;

         if ((rng() % 600u) == 0u)
         {
            for (NfcDetF *m: {&a, &b})
            {
               m->symStart = 0; m->symEnd = 0; m->winStart = 0; m->winEnd = 0; m->sync = 0; m->peakTime = 0; m->peak = 0;
            }
         }

         uint32_t pa = 0, pb = 0;
         const bool qa = clock >= a.winStart ? nfcf_track_preamble(sa, a, rt, sd, s0, sd > level, pa, &clearedA, &usedA, 1u, &bound) : false;
         const bool qb = clock >= b.winStart ? nfcf_track_preamble(sb, b, rt, sd, s0, sd > level, pb, &clearedB, &usedB, 1u, nullptr) : false;

         if (qa && !clearedA)
            bound.flags |= NFC_FBOUND_EXACT; /* (nfcf_detect_decide: a preamble completed on the memory the lane was given) */

         ra[steps] = a; rb[steps] = b;
         reta[steps] = qa ? 1 + (uint8_t)pa : 0; retb[steps] = qb ? 1 + (uint8_t)pb : 0;
         cla[steps] = (uint8_t)clearedA; clb[steps] = (uint8_t)clearedB;
         owna[steps] = (bound.flags & NFC_FBOUND_THR_OWN) ? 1 : 0;

         locked = qa || qb;
      }

      evaluated += usedA ? 1u : 0u;

      if (!nfc_fbound_admits(bound, hadPulses, hadThr, havePulses, haveThr, false, false))
         continue;

      admitted++;

      const uint32_t d = havePulses - hadPulses;

      different += (d != 0u || std::memcmp(&hadThr, &haveThr, 4) != 0) ? 1u : 0u;
      offset += d != 0u ? 1u : 0u;
      threshold += std::memcmp(&hadThr, &haveThr, 4) != 0 ? 1u : 0u;
      completed += locked ? 1u : 0u;
      startedOver += clearedA ? 1u : 0u;

      for (uint32_t k = 0; k < steps; k++)
      {
         bool ok = reta[k] == retb[k] && cla[k] == clb[k] && same_but_memory(ra[k], rb[k]);
         ok = ok && (cla[k] ? ra[k].pulses == rb[k].pulses : ra[k].pulses + d == rb[k].pulses);
         ok = ok && (owna[k] ? std::memcmp(&ra[k].thr, &rb[k].thr, 4) == 0 : (std::memcmp(&ra[k].thr, &hadThr, 4) == 0 && std::memcmp(&rb[k].thr, &haveThr, 4) == 0));

         if (!ok)
         {
            std::fprintf(stderr, "violation: seed %u train %u step %u: assumed (%u, %g) true (%u, %g), bounds lowMax %u highMin %u above %g below %g flags %x\n", seed, n, k,
                         hadPulses, hadThr, havePulses, haveThr, bound.lowMax, bound.highMin, bound.thrAbove, bound.thrBelow, bound.flags);
            std::fprintf(stderr, "   A: ret %u cleared %u pulses %u thr %g own %u | B: ret %u cleared %u pulses %u thr %g\n", reta[k], cla[k], ra[k].pulses, ra[k].thr, owna[k], retb[k],
                         clb[k], rb[k].pulses, rb[k].thr);
            return 1;
         }
      }
   }

   std::printf("{\"seed\": %u, \"trains\": %u, \"with_an_evaluation\": %llu, \"admitted\": %llu, \"admitted_with_another_memory\": %llu, \"of_them_counter_off\": %llu, "
               "\"of_them_threshold_off\": %llu, \"admitted_that_started_over\": %llu, \"admitted_that_completed_a_preamble\": %llu, \"violations\": 0}\n", seed, trains,
               (unsigned long long)evaluated, (unsigned long long)admitted, (unsigned long long)different, (unsigned long long)offset, (unsigned long long)threshold,
               (unsigned long long)startedOver, (unsigned long long)completed);
   return 0;
}
